// Convolution with LDS-staged input patches for gfx950 (CDNA4).
//
// Same GEMM view, packed-weight format and epilogues as conv_igemm.hip, different data movement:
//   * activations: the (tile + kernel-1) input "halo" box of a CK-channel chunk is staged ONCE in LDS and every
//     tap of the kernel reads its shifted window from there -- 9x (3x3), 27x (3x3x3) or 343x (7x7x7) fewer
//     global/L2 reads, address computations and bounds checks than gathering a tile per tap;
//   * weights: never touch LDS. The packed layout [kstep][Cout][32] makes one MFMA A-operand fragment
//     (16 rows x 32 k) a contiguous 1 KiB, so each wave streams its own fragments global -> VGPR through a
//     PFD-deep, statically indexed register ring (the loads of K-step s+PFD are issued right after step s has
//     consumed its registers) -- deep enough to cover L2 / Infinity-Cache latency under load;
//   * therefore no workgroup barrier inside the tap loop: one __syncthreads() per channel chunk (when the
//     halo is swapped), waves otherwise run free and overlap each other's latencies;
//   * SK variants (narrow Cout: the 7x7x7 mask conv): the 4 waves split the K-steps of every chunk instead of the
//     positions, so each still issues 16 MFMAs per 8 LDS reads, and reduce their accumulators through LDS once.
// Per K-step (32 channels of one tap) a wave issues WPX ds_read_b128 + WCH global_load_dwordx4 + WPX*WCH MFMAs.
//
// LDS image: [halo voxel][CK channels] fp16, voxel stride CK*2+16 bytes, followed by a tap -> byte-offset table.
#include "common.h"
#include "conv_epilogue.h"

template <int CK, int WPX, int WCH, int WVP, int WVC, int MODE, bool DB, bool SK>
__global__ void __launch_bounds__(256) conv_halo_kernel(const ConvParams p)
{
    constexpr int BM = SK ? WPX * 16 : WPX * 16 * WVP;
    constexpr int BN = WCH * 16 * WVC;
    constexpr int SL = CK / 8;           // 16-byte slots per voxel
    constexpr int VS = CK * 2 + 16;      // LDS bytes per halo voxel
    constexpr int HI = 6;                // halo pieces a thread holds in flight (double-buffered mode)
    constexpr int KH32 = CK / 32;        // 32-channel K-steps per tap and chunk
    constexpr int PFD = 4;               // weight prefetch depth in K-steps
    constexpr int SKS = SK ? 4 : 1;      // K-step stride of one wave
    static_assert(WVP * WVC == 4, "4 waves per workgroup");
    static_assert(!SK || (WVC == 1 && WPX == 8), "split-K variants: every wave covers all 128 positions");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpx = SK ? 0 : wave % WVP, wch = SK ? 0 : wave / WVP;
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
    const int ntaps = p.KD * p.KH * p.KW;
    int* tofftab = (int*)(smem + (size_t)(DB ? 2 : 1) * HV * VS);     // tap -> LDS byte offset of the shifted window
    for (int i = tid; i < ntaps; i += 256) {
        const int kw = i % p.KW, r = i / p.KW;
        tofftab[i] = (((r / p.KH) * HH + (r % p.KH)) * HW + kw) * VS;
    }

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
    const int nck = (p.Cin + CK - 1) / CK;
    const int j0 = SK ? wave : 0;

    // This wave's K-step sequence: for every chunk cc, local steps j = j0, j0+SKS, ... < ntaps*nhalf(cc), with
    // tap = j / nhalf, half = j % nhalf, packed index kidx = (cc*KH32 + half)*ntaps + tap.
    auto chunk_steps = [&](int cc) -> int {
        const int rem = p.nchunks - cc * KH32;
        return ntaps * (rem < KH32 ? rem : KH32);
    };
    int pcc = 0, pj = j0, psteps = chunk_steps(0);        // producer (weight prefetch) position
    auto wload = [&](h8_t (&dst)[WCH]) {
        while (pcc < nck && pj >= psteps) { ++pcc; pj = j0; psteps = pcc < nck ? chunk_steps(pcc) : 0; }
        if (pcc < nck) {
            const bool two = psteps == 2 * ntaps;
            const int tap = two ? pj >> 1 : pj, half = two ? pj & 1 : 0;
            const long off = (long)((pcc * KH32 + half) * ntaps + tap) * wstep;
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci) dst[ci] = *(const h8_t*)(wlane + off + ci * 512);
            pj += SKS;
        } else {
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci) dst[ci] = (h8_t){0, 0, 0, 0, 0, 0, 0, 0};
        }
    };

    f4_t acc[WCH][WPX];
#pragma unroll
    for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
        for (int pi = 0; pi < WPX; ++pi) acc[ci][pi] = (f4_t){0.f, 0.f, 0.f, 0.f};

    h8_t wr[PFD][WCH];
#pragma unroll
    for (int i = 0; i < PFD; ++i) wload(wr[i]);

    fill_halo(0, 0);
    __syncthreads();
    if (DB && nck > 1) prefetch_issue(CK);
    int cur = 0, cc = 0, cj = j0, csteps = chunk_steps(0);
    const unsigned char* hb = smem;
    bool done = false;
    while (!done) {
#pragma unroll
        for (int i = 0; i < PFD; ++i) {
            while (!done && cj >= csteps) {                // this wave finished its share of chunk cc
                if (cc + 1 >= nck) { done = true; break; }
                if (DB) {
                    prefetch_commit(cur ^ 1);              // buffer cur^1 was last read in chunk cc-1 (behind the previous barrier)
                    __syncthreads();
                    cur ^= 1;
                    if (cc + 2 < nck) prefetch_issue((cc + 2) * CK);
                } else {
                    __syncthreads();                       // everyone is done reading the single buffer
                    fill_halo(0, (cc + 1) * CK);
                    __syncthreads();
                }
                ++cc; cj = j0; csteps = chunk_steps(cc);
                hb = smem + (size_t)cur * HV * VS;
            }
            if (done) break;
            const bool two = csteps == 2 * ntaps;
            const int tap = two ? cj >> 1 : cj, half = two ? cj & 1 : 0;
            const int toff = tofftab[tap] + half * 64;
            h8_t af[WPX];
#pragma unroll
            for (int pi = 0; pi < WPX; ++pi) af[pi] = *(const h8_t*)(hb + abase[pi] + toff);
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                for (int pi = 0; pi < WPX; ++pi)
                    acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[i][ci], af[pi], acc[ci][pi], 0, 0, 0);
            wload(wr[i]);
            cj += SKS;
        }
    }

    if constexpr (!SK) {
        constexpr int EP_WPX = WPX;
        const int ep_wpx = wpx;
        auto& ep_acc = acc;
        CONV_EPILOGUE()
    } else {
        // reduce the four waves' partial accumulators through LDS; wave w finishes position blocks 2w, 2w+1
        __syncthreads();                                   // halo no longer needed: reuse it
        float* red = (float*)smem;                         // [4 waves][WCH*WPX frags][4][64 lanes]
#pragma unroll
        for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
            for (int pi = 0; pi < WPX; ++pi)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((wave * WCH * WPX + ci * WPX + pi) * 4 + r) * 64 + lane] = acc[ci][pi][r];
        __syncthreads();
        constexpr int EP_WPX = 2;
        const int ep_wpx = wave;
        f4_t ep_acc[WCH][EP_WPX];
#pragma unroll
        for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
            for (int q = 0; q < EP_WPX; ++q) {
                const int pi = wave * EP_WPX + q;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) s += red[((w * WCH * WPX + ci * WPX + pi) * 4 + r) * 64 + lane];
                    ep_acc[ci][q][r] = s;
                }
            }
        CONV_EPILOGUE()
    }
}

template <int CK, int WPX, int WCH, int WVP, int WVC, int MODE, bool SK>
static int launch_halo_cfg(const ConvParams& p, hipStream_t st)
{
    constexpr int BM = SK ? WPX * 16 : WPX * 16 * WVP, BN = WCH * 16 * WVC;
    constexpr int VS = CK * 2 + 16, SL = CK / 8;
    if (p.Cout_pad % BN != 0) { cs_set_error("conv_halo: Cout_pad %d not a multiple of the channel tile %d", p.Cout_pad, BN); return -1; }
    if (p.inD != p.D) { cs_set_error("conv_halo: depth-collapsing convs are not supported"); return -1; }
    const int lgS = p.lgTW + p.lgTH + p.lgTD;
    if ((1 << lgS) > BM) { cs_set_error("conv_halo: spatial tile exceeds BM"); return -1; }
    const int TN = BM >> lgS;
    const long HV = (long)TN * ((1 << p.lgTD) + p.KD - 1) * ((1 << p.lgTH) + p.KH - 1) * ((1 << p.lgTW) + p.KW - 1);
    const int nck = (p.Cin + CK - 1) / CK;
    const bool db = nck > 1 && HV * SL <= 256 * 6 && 2 * HV * VS <= 64 * 1024;
    size_t lds = (size_t)(db ? 2 : 1) * HV * VS + (size_t)p.KD * p.KH * p.KW * sizeof(int);
    if (SK && lds < (size_t)4 * WCH * WPX * 4 * 64 * sizeof(float)) lds = (size_t)4 * WCH * WPX * 4 * 64 * sizeof(float);
    if (lds > 160 * 1024) { cs_set_error("conv_halo: halo of %ld voxels does not fit LDS", HV); return -1; }
    dim3 grid((unsigned)(p.nTW * p.nTH * p.nTD * p.nTN), (unsigned)(p.Cout_pad / BN));
    hipError_t e;
    if (db) {
        auto k = conv_halo_kernel<CK, WPX, WCH, WVP, WVC, MODE, true, SK>;
        if (lds > 64 * 1024) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); (void)e; }
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
    } else {
        auto k = conv_halo_kernel<CK, WPX, WCH, WVP, WVC, MODE, false, SK>;
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
#define HALO_CASE(CFG, WPX, WCH, WVP, WVC, MODE, SK)                                          \
    if (cfg == CFG && mode == MODE) {                                                         \
        if (ck == 64) return launch_halo_cfg<64, WPX, WCH, WVP, WVC, MODE, SK>(p, st);       \
        return launch_halo_cfg<32, WPX, WCH, WVP, WVC, MODE, SK>(p, st);                      \
    }
    HALO_CASE(CFG_H_128x128, 8, 2, 1, 4, MODE_STD, false)
    HALO_CASE(CFG_H_128x128, 8, 2, 1, 4, MODE_TBLEND, false)
    HALO_CASE(CFG_H_128x128, 8, 2, 1, 4, MODE_SPADE, false)
    HALO_CASE(CFG_H_128x64, 4, 2, 2, 2, MODE_STD, false)
    HALO_CASE(CFG_H_256x32, 4, 2, 4, 1, MODE_STD, false)
    HALO_CASE(CFG_H_128x32, 2, 2, 4, 1, MODE_STD, false)
    HALO_CASE(CFG_H_128x16, 2, 1, 4, 1, MODE_STD, false)
    HALO_CASE(CFG_H_256x16, 4, 1, 4, 1, MODE_PIXSHUF, false)
    HALO_CASE(CFG_H_SK128x32, 8, 2, 4, 1, MODE_STD, true)
#undef HALO_CASE
    cs_set_error("conv_halo: unsupported cfg/mode %d/%d", cfg, mode);
    return -1;
}
