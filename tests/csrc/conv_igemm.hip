// TEST-ONLY cross-check kernel (built into libcanonswap_test.so, never into the product library).
// Implicit-GEMM convolution for gfx950 (CDNA4): fp16 operands, fp32 accumulate on
// v_mfma_f32_16x16x32_f16, channels-last activations with arbitrary position strides.
//
// Replaces every F.conv2d / F.conv3d / nn.Conv{2,3}d call site on the CanonSwap generator path
// (util.py:80-344, appearance_feature_extractor.py:38-48, dense_motion.py:67-104,
// warping_network.py:64-71, adaptive_modulate.py:128-193, spade_generator.py:41-59) with one kernel
// family whose epilogue carries the surrounding element-wise work (folded eval-BatchNorm, activation,
// residual add, occlusion multiply, pre-activation of the next layer, the T blend, SPADE modulation,
// PixelShuffle + sigmoid).
//
// Work decomposition (64-lane wavefronts, 4 waves per workgroup):
//   workgroup tile = BM output positions x BN output channels, K-step = 32 input channels of one tap
//   LDS: activations [BM][32] and weights [BN][32] fp16, double buffered, 16-byte slots XOR-swizzled so
//        that the ds_read_b128 lane groups of the MFMA operand fetch are bank-conflict free
//   MFMA operand roles are swapped (A = weights -> rows = channels, B = activations -> columns =
//   positions) so each lane ends up with 4 consecutive channels of one position: 8/16-byte stores.
#include "../../canonswap_amd/csrc/common.h"
#include "../../canonswap_amd/csrc/conv_epilogue.h"

__device__ __forceinline__ int swz_slot(int row, int seg)
{
    // f(q) = {0,2,3,1}[q], q = (row>>2)&3 : makes the four 16-lane groups of ds_read_b128 hit 16 distinct slots
    return seg ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
}

template <int WPX, int WCH, int WVP, int WVC, int MODE>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvParams p)
{
    constexpr int BM = WPX * 16 * WVP;
    constexpr int BN = WCH * 16 * WVC;
    constexpr int AR = BM / 64;                // activation rows staged per thread
    constexpr int BL = (BN * 4 + 255) / 256;   // weight 16-byte pieces staged per thread
    static_assert(WVP * WVC == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* As = (half_t*)smem;          // [2][BM][32]
    half_t* Bs = As + 2 * BM * 32;       // [2][BN][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wpx = wave % WVP, wch = wave / WVP;
    const int l15 = lane & 15, l4 = lane >> 4;

    const int tile_lin = blockIdx.x;
    int t = tile_lin;
    const int tw = t % p.nTW; t /= p.nTW;
    const int th = t % p.nTH; t /= p.nTH;
    const int td = t % p.nTD; t /= p.nTD;
    const int tn = t;
    const int n0 = blockIdx.y * BN;
    const int lgTW = p.lgTW, lgTH = p.lgTH, lgTD = p.lgTD;
    const int lgS = p.lgTW + p.lgTH + p.lgTD;
    const int mW = (1 << p.lgTW) - 1, mH = (1 << p.lgTH) - 1, mD = (1 << p.lgTD) - 1;

    // position of the rows this thread stages
    const int seg = tid & 3;
    int rn[AR], rd[AR], rh[AR], rw[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int m = (tid >> 2) + 64 * i;
        rw[i] = (tw << p.lgTW) + (m & mW); m >>= p.lgTW;
        rh[i] = (th << p.lgTH) + (m & mH); m >>= p.lgTH;
        rd[i] = (td << p.lgTD) + (m & mD); m >>= p.lgTD;
        rn[i] = tn * (BM >> lgS) + m;
    }

    uint4 areg[AR], breg[BL];
    auto gload = [&](int kd, int kh, int kw, int c0, int ks) {
        const int dd = kd - p.PD, dh = kh - p.PH, dw = kw - p.PW;
        const int cc = c0 + seg * 8;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int id = rd[i] + dd, ih = rh[i] + dh, iw = rw[i] + dw;
            const bool ok = rn[i] < p.N && (unsigned)id < (unsigned)p.inD && (unsigned)ih < (unsigned)p.H &&
                            (unsigned)iw < (unsigned)p.W && cc < p.Cin;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                const half_t* src = p.in + (long)rn[i] * p.in_sN + (long)id * p.in_sD +
                                    (long)(ih >> p.up_shift) * p.in_sH + (long)(iw >> p.up_shift) * p.in_sW + cc;
                v = *(const uint4*)src;
            }
            areg[i] = v;
        }
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            const int idx = tid + 256 * j;
            if (idx < BN * 4)
                breg[j] = *(const uint4*)(p.wgt + ((long)ks * p.Cout_pad + n0 + (idx >> 2)) * 32 + (idx & 3) * 8);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int row = (tid >> 2) + 64 * i;
            *(uint4*)(As + buf * BM * 32 + row * 32 + swz_slot(row, seg) * 8) = areg[i];
        }
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            const int idx = tid + 256 * j;
            if (idx < BN * 4) {
                const int row = idx >> 2;
                *(uint4*)(Bs + buf * BN * 32 + row * 32 + swz_slot(row, idx & 3) * 8) = breg[j];
            }
        }
    };

    f4_t acc[WCH][WPX];
#pragma unroll
    for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
        for (int pi = 0; pi < WPX; ++pi) acc[ci][pi] = (f4_t){0.f, 0.f, 0.f, 0.f};

    const int nks = p.nchunks * p.KD * p.KH * p.KW;
    int kd = 0, kh = 0, kw = 0, c0 = 0;
    gload(0, 0, 0, 0, 0);
    sstore(0);
    __syncthreads();

    for (int ks = 0; ks < nks; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < nks;
        if (more) {
            if (++kw == p.KW) { kw = 0; if (++kh == p.KH) { kh = 0; if (++kd == p.KD) { kd = 0; c0 += 32; } } }
            gload(kd, kh, kw, c0, ks + 1);
        }
        h8_t wf[WCH], af[WPX];
#pragma unroll
        for (int ci = 0; ci < WCH; ++ci) {
            const int row = wch * WCH * 16 + ci * 16 + l15;
            wf[ci] = *(const h8_t*)(Bs + cur * BN * 32 + row * 32 + swz_slot(row, l4) * 8);
        }
#pragma unroll
        for (int pi = 0; pi < WPX; ++pi) {
            const int row = wpx * WPX * 16 + pi * 16 + l15;
            af[pi] = *(const h8_t*)(As + cur * BM * 32 + row * 32 + swz_slot(row, l4) * 8);
        }
#pragma unroll
        for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
            for (int pi = 0; pi < WPX; ++pi)
                acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ci], af[pi], acc[ci][pi], 0, 0, 0);
        if (more) sstore(cur ^ 1);
        __syncthreads();
    }

    constexpr int EP_WPX = WPX;
    constexpr int EP_PAIR = 0;           // weights go through LDS here: plain row order
    constexpr bool EP_EARLY = false;
    constexpr bool EP_FAST = false;
    const int l15p = l15;
    ep_u2_t ep_xpre[1][1];
    constexpr bool EP_HEAVY = true;      // the cross-check kernel carries every activation
    const int ep_wpx = wpx;
    auto& ep_acc = acc;
    CONV_EPILOGUE()
}

template <int WPX, int WCH, int WVP, int WVC, int MODE>
static int launch_cfg(const ConvParams& p, hipStream_t st)
{
    constexpr int BM = WPX * 16 * WVP, BN = WCH * 16 * WVC;
    if (p.Cout_pad % BN != 0) { cs_set_error("conv: Cout_pad %d not a multiple of the channel tile %d", p.Cout_pad, BN); return -1; }
    if ((1 << (p.lgTW + p.lgTH + p.lgTD)) > BM) { cs_set_error("conv: spatial tile exceeds BM"); return -1; }
    if (MODE == MODE_SPADE && (1 << (p.lgTW + p.lgTH + p.lgTD)) != BM) { cs_set_error("conv: SPADE launches must tile within one sample"); return -1; }
    const size_t lds = (size_t)2 * (BM + BN) * 32 * sizeof(half_t);
    dim3 grid((unsigned)(p.nTW * p.nTH * p.nTD * p.nTN), (unsigned)(p.Cout_pad / BN));
    hipLaunchKernelGGL((conv_igemm_kernel<WPX, WCH, WVP, WVC, MODE>), grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cs_set_error("conv launch: %s", hipGetErrorString(e)); return -1; }
    return 0;
}

int launch_conv(const ConvParams& p, int cfg, int mode, hipStream_t st)
{
    if (p.wslot) { cs_set_error("conv_igemm: per-sample weight sets are only implemented by conv_halo"); return -1; }
    if (p.act1 >= ACT_SIGMOID) { cs_set_error("conv_igemm: the second output takes none / ReLU / LeakyReLU only"); return -1; }
    if (ep_check_extents(p, "conv_igemm")) return -1;
    if (mode == MODE_STD && p.stat_out) {
        switch (cfg) {
        case CFG_128x128: return launch_cfg<4, 4, 2, 2, MODE_STDSTAT>(p, st);
        case CFG_128x64: return launch_cfg<2, 4, 4, 1, MODE_STDSTAT>(p, st);
        case CFG_256x32: return launch_cfg<4, 2, 4, 1, MODE_STDSTAT>(p, st);
        }
    } else if (mode == MODE_STD) {
        switch (cfg) {
        case CFG_128x128: return launch_cfg<4, 4, 2, 2, MODE_STD>(p, st);
        case CFG_128x64: return launch_cfg<2, 4, 4, 1, MODE_STD>(p, st);
        case CFG_256x32: return launch_cfg<4, 2, 4, 1, MODE_STD>(p, st);
        case CFG_256x16: return launch_cfg<4, 1, 4, 1, MODE_STD>(p, st);
        }
    } else if (mode == MODE_TBLEND) {
        if (cfg == CFG_128x128) return launch_cfg<4, 4, 2, 2, MODE_TBLEND>(p, st);
    } else if (mode == MODE_SPADE) {
        if (cfg == CFG_128x128) return launch_cfg<4, 4, 2, 2, MODE_SPADE>(p, st);
        if (cfg == CFG_128x64) return launch_cfg<2, 4, 4, 1, MODE_SPADE>(p, st);
    } else if (mode == MODE_PIXSHUF) {
        if (cfg == CFG_256x16) return launch_cfg<4, 1, 4, 1, MODE_PIXSHUF>(p, st);
    }
    cs_set_error("conv: unsupported cfg/mode %d/%d", cfg, mode);
    return -1;
}
