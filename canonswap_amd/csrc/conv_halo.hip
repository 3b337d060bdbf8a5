// Convolution with LDS-staged input patches for gfx950 (CDNA4).
//
// Same GEMM view, packed-weight format and epilogues as conv_igemm.hip, different data movement:
//   * activations: the (tile + kernel-1) input "halo" box of a CK-channel chunk is staged ONCE in LDS and every
//     tap of the kernel reads its shifted window from there -- 9x (3x3), 27x (3x3x3) or 343x (7x7x7) fewer
//     global/L2 reads, address computations and bounds checks than gathering a tile per tap;
//   * weights: never touch LDS. The packed layout [kstep][Cout][32] makes one MFMA A-operand fragment
//     (16 rows x 32 k) a contiguous 1 KiB, so each wave streams its own fragments global -> VGPR with a
//     two-step software prefetch; no LDS traffic, no staging pass;
//   * therefore no workgroup barrier inside the tap loop: one __syncthreads() per channel chunk (when the
//     double-buffered halo is swapped), waves otherwise run free and overlap each other's latencies.
// Per K-step (32 channels of one tap) a wave issues WPX ds_read_b128 + WCH global_load_dwordx4 + WPX*WCH MFMAs.
//
// LDS image: [halo voxel][CK channels] fp16, voxel stride CK*2+16 bytes (the 16-byte pad makes the 16 voxels
// of an MFMA operand fetch land on distinct bank groups).
#include "common.h"
#include "conv_epilogue.h"

template <int CK, int WPX, int WCH, int WVP, int WVC, int MODE, bool DB>
__global__ void __launch_bounds__(256) conv_halo_kernel(const ConvParams p)
{
    constexpr int BM = WPX * 16 * WVP;
    constexpr int BN = WCH * 16 * WVC;
    constexpr int SL = CK / 8;           // 16-byte slots per voxel
    constexpr int VS = CK * 2 + 16;      // LDS bytes per halo voxel
    constexpr int HI = 8;                // halo pieces a thread can hold in flight (double-buffered mode)
    constexpr int KH32 = CK / 32;        // 32-channel K-steps per tap and chunk
    static_assert(WVP * WVC == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wpx = wave % WVP, wch = wave / WVP;
    const int l15 = lane & 15, l4 = lane >> 4;

    int t = blockIdx.x;
    const int tw = t % p.nTW; t /= p.nTW;
    const int th = t % p.nTH; t /= p.nTH;
    const int td = t % p.nTD; t /= p.nTD;
    const int tn = t;
    const int n0 = blockIdx.y * BN;
    const int lgS = p.lgTW + p.lgTH + p.lgTD;
    const int mW = (1 << p.lgTW) - 1, mH = (1 << p.lgTH) - 1, mD = (1 << p.lgTD) - 1;
    const int TN = BM >> lgS;
    const int HW = (1 << p.lgTW) + p.KW - 1, HH = (1 << p.lgTH) + p.KH - 1, HD = (1 << p.lgTD) + p.KD - 1;
    const int HV = TN * HD * HH * HW;
    const int nitems = HV * SL;
    const int w0 = tw << p.lgTW, h0 = th << p.lgTH, d0 = td << p.lgTD, nb = tn * TN;

    // ---- halo staging: piece q = (voxel q / SL, 16-byte slot q % SL)
    auto piece_src = [&](int q, int c0, bool& ok) -> const half_t* {
        const int hv = q / SL, slot = q % SL;
        const int hw = hv % HW; int r = hv / HW;
        const int hh = r % HH; r /= HH;
        const int hd = r % HD;
        const int hn = r / HD;
        const int n = nb + hn, id = d0 + hd - p.PD, ih = h0 + hh - p.PH, iw = w0 + hw - p.PW, c = c0 + slot * 8;
        ok = q < nitems && n < p.N && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W &&
             c < p.Cin;
        return p.in + (long)n * p.in_sN + (long)id * p.in_sD + (long)(ih >> p.up_shift) * p.in_sH +
               (long)(iw >> p.up_shift) * p.in_sW + c;
    };
    auto piece_dst = [&](int q, int buf) -> uint4* {
        return (uint4*)(smem + (size_t)buf * HV * VS + (size_t)(q / SL) * VS + (q % SL) * 16);
    };
    auto fill_halo = [&](int buf, int c0) {      // synchronous fill, batches of 8 loads in flight per thread
        for (int q0 = tid; q0 < nitems; q0 += 256 * 8) {
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bool ok;
                const half_t* src = piece_src(q0 + 256 * j, c0, ok);
                v[j] = make_uint4(0, 0, 0, 0);
                if (ok) v[j] = *(const uint4*)src;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (q0 + 256 * j < nitems) *piece_dst(q0 + 256 * j, buf) = v[j];
        }
    };
    uint4 hreg[DB ? HI : 1];
    auto prefetch_issue = [&](int c0) {
#pragma unroll
        for (int j = 0; j < (DB ? HI : 1); ++j) {
            bool ok;
            const half_t* src = piece_src(tid + 256 * j, c0, ok);
            hreg[j] = make_uint4(0, 0, 0, 0);
            if (ok) hreg[j] = *(const uint4*)src;
        }
    };
    auto prefetch_commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < (DB ? HI : 1); ++j)
            if (tid + 256 * j < nitems) *piece_dst(tid + 256 * j, buf) = hreg[j];
    };

    // ---- per-lane constants of the MFMA operand fetches
    int abase[WPX];                      // LDS byte offset of this lane's position in the un-shifted halo window
#pragma unroll
    for (int pi = 0; pi < WPX; ++pi) {
        int m = wpx * WPX * 16 + pi * 16 + l15;
        const int wl = m & mW; m >>= p.lgTW;
        const int hl = m & mH; m >>= p.lgTH;
        const int dl = m & mD; m >>= p.lgTD;
        abase[pi] = (((m * HD + dl) * HH + hl) * HW + wl) * VS + l4 * 16;
    }
    // weights: fragment ci of K-step kidx = 1 KiB at wgt + (kidx*Cout_pad + n0 + wch*WCH*16 + ci*16)*32; lane = (row l15, k l4*8)
    const half_t* wlane = p.wgt + ((long)(n0 + wch * WCH * 16) * 32 + l15 * 32 + l4 * 8);
    const long wstep = (long)p.Cout_pad * 32;
    const int ntaps = p.KD * p.KH * p.KW;
    const int nks = p.nchunks * ntaps;   // 32-channel K-steps in total

    // producer iterator (weights prefetch) in consumption order: chunk -> kd -> kh -> kw -> 32-channel half
    int pc = 0, ptap = 0, phalf = 0, pcount = 0;
    auto next_kidx = [&]() -> int {      // returns packed K-step index of the next step, -1 when exhausted
        if (pcount >= nks) return -1;
        const int c32 = pc * KH32 + phalf;
        const int kidx = c32 * ntaps + ptap;
        ++pcount;
        if (++phalf == KH32 || pc * KH32 + phalf >= p.nchunks) { phalf = 0; if (++ptap == ntaps) { ptap = 0; ++pc; } }
        return kidx;
    };
    auto wload = [&](h8_t (&dst)[WCH]) {
        const int kidx = next_kidx();
#pragma unroll
        for (int ci = 0; ci < WCH; ++ci) {
            if (kidx >= 0) dst[ci] = *(const h8_t*)(wlane + kidx * wstep + ci * 512);
            else dst[ci] = (h8_t){0, 0, 0, 0, 0, 0, 0, 0};
        }
    };

    f4_t acc[WCH][WPX];
#pragma unroll
    for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
        for (int pi = 0; pi < WPX; ++pi) acc[ci][pi] = (f4_t){0.f, 0.f, 0.f, 0.f};

    h8_t wa[WCH], wb[WCH];
    wload(wa);
    wload(wb);

    const int nck = (p.Cin + CK - 1) / CK;
    fill_halo(0, 0);
    __syncthreads();
    int cur = 0;
    for (int cc = 0; cc < nck; ++cc) {
        if (DB && cc + 1 < nck) prefetch_issue((cc + 1) * CK);
        const unsigned char* hb = smem + (size_t)cur * HV * VS;
        const int nhalf = (p.nchunks - cc * KH32) < KH32 ? (p.nchunks - cc * KH32) : KH32;
        for (int kd = 0; kd < p.KD; ++kd)
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw) {
                    const int toff = ((kd * HH + kh) * HW + kw) * VS;
                    for (int half = 0; half < nhalf; ++half) {
                        h8_t af[WPX];
#pragma unroll
                        for (int pi = 0; pi < WPX; ++pi) af[pi] = *(const h8_t*)(hb + abase[pi] + toff + half * 64);
#pragma unroll
                        for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                            for (int pi = 0; pi < WPX; ++pi)
                                acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ci], af[pi], acc[ci][pi], 0, 0, 0);
#pragma unroll
                        for (int ci = 0; ci < WCH; ++ci) wa[ci] = wb[ci];
                        wload(wb);
                    }
                }
        if (cc + 1 < nck) {
            if (DB) {
                prefetch_commit(cur ^ 1);     // buffer cur^1 was last read in chunk cc-1 (behind the previous barrier)
                __syncthreads();
                cur ^= 1;
            } else {
                __syncthreads();              // everyone is done reading the single buffer
                fill_halo(0, (cc + 1) * CK);
                __syncthreads();
            }
        }
    }

    CONV_EPILOGUE()
}

template <int CK, int WPX, int WCH, int WVP, int WVC, int MODE>
static int launch_halo_cfg(const ConvParams& p, hipStream_t st)
{
    constexpr int BM = WPX * 16 * WVP, BN = WCH * 16 * WVC;
    constexpr int VS = CK * 2 + 16, SL = CK / 8;
    if (p.Cout_pad % BN != 0) { cs_set_error("conv_halo: Cout_pad %d not a multiple of the channel tile %d", p.Cout_pad, BN); return -1; }
    if (p.inD != p.D) { cs_set_error("conv_halo: depth-collapsing convs are not supported"); return -1; }
    const int lgS = p.lgTW + p.lgTH + p.lgTD;
    if ((1 << lgS) > BM) { cs_set_error("conv_halo: spatial tile exceeds BM"); return -1; }
    const int TN = BM >> lgS;
    const long HV = (long)TN * ((1 << p.lgTD) + p.KD - 1) * ((1 << p.lgTH) + p.KH - 1) * ((1 << p.lgTW) + p.KW - 1);
    const int nck = (p.Cin + CK - 1) / CK;
    const bool db = nck > 1 && HV * SL <= 256 * 8 && 2 * HV * VS <= 64 * 1024;
    const size_t lds = (size_t)(db ? 2 : 1) * HV * VS;
    if (lds > 160 * 1024) { cs_set_error("conv_halo: halo of %ld voxels does not fit LDS", HV); return -1; }
    dim3 grid((unsigned)(p.nTW * p.nTH * p.nTD * p.nTN), (unsigned)(p.Cout_pad / BN));
    hipError_t e;
    if (db) {
        auto k = conv_halo_kernel<CK, WPX, WCH, WVP, WVC, MODE, true>;
        if (lds > 64 * 1024) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); (void)e; }
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
    } else {
        auto k = conv_halo_kernel<CK, WPX, WCH, WVP, WVC, MODE, false>;
        if (lds > 64 * 1024) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); (void)e; }
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
    }
    e = hipGetLastError();
    if (e != hipSuccess) { cs_set_error("conv_halo launch: %s", hipGetErrorString(e)); return -1; }
    return 0;
}

// cfg: CFG_H_* (common.h); ck: 32 or 64
int launch_conv_halo(const ConvParams& p, int cfg, int ck, int mode, hipStream_t st)
{
#define HALO_CASE(CFG, WPX, WCH, WVP, WVC, MODE)                                              \
    if (cfg == CFG && mode == MODE) {                                                         \
        if (ck == 64) return launch_halo_cfg<64, WPX, WCH, WVP, WVC, MODE>(p, st);           \
        return launch_halo_cfg<32, WPX, WCH, WVP, WVC, MODE>(p, st);                          \
    }
    HALO_CASE(CFG_H_128x128, 8, 2, 1, 4, MODE_STD)
    HALO_CASE(CFG_H_128x128, 8, 2, 1, 4, MODE_TBLEND)
    HALO_CASE(CFG_H_128x128, 8, 2, 1, 4, MODE_SPADE)
    HALO_CASE(CFG_H_128x64, 4, 2, 2, 2, MODE_STD)
    HALO_CASE(CFG_H_128x64, 4, 2, 2, 2, MODE_SPADE)
    HALO_CASE(CFG_H_256x32, 4, 2, 4, 1, MODE_STD)
    HALO_CASE(CFG_H_128x32, 2, 2, 4, 1, MODE_STD)
    HALO_CASE(CFG_H_128x16, 2, 1, 4, 1, MODE_STD)
    HALO_CASE(CFG_H_256x16, 4, 1, 4, 1, MODE_PIXSHUF)
#undef HALO_CASE
    cs_set_error("conv_halo: unsupported cfg/mode %d/%d", cfg, mode);
    return -1;
}
