#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <cstdint>
#define __device__
#define __forceinline__ inline
struct float4 { float x, y, z, w; }; struct float2 { float x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return {a, b}; }
#define __expf expf
static inline float __shfl_xor(float a, int, int) { return a; }
static inline int __mul24(int a, int b) { return (int)(((long long)(a << 8 >> 8)) * (b << 8 >> 8)); }      // v_mul_i32_i24: low 24 bits of each
static void cs_set_error(const char*, ...) {}
#define hipStream_t void*
#include "common_stub.h"
#include EPH

static uint32_t hsh(uint32_t a) { a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a; }
static float rnd(uint32_t a) { return (float)(hsh(a) % 20001) / 10000.f - 1.f; }

// EP_FAST as in conv_halo_kernel.h (the branch-free copies of the epilogue for the hot tensor combinations): HARNESS_NO_FAST runs
// every case through the general path only; both builds must print the same checksums
template <int WPX, int WCH, int WVP, int WVC, int MODE, bool EP_HEAVY>
void run(const ConvParams& p, int BM)
{
#ifdef HARNESS_NO_FAST
    constexpr bool EP_FAST = false;
#else
    constexpr bool EP_FAST = (WCH == 4 || WCH == 5 || (WCH == 2 && WPX == 8));
#endif
    constexpr int BN = WCH * 16 * WVC;
    const int ntiles = p.nTW * p.nTH * p.nTD * p.nTN;
    for (int cb = 0; cb < p.Cout_pad / BN; ++cb)
    for (int tile = 0; tile < ntiles; ++tile)
    for (int wave = 0; wave < 4; ++wave)
    for (int lane = 0; lane < 64; ++lane) {
        const int tile_lin = tile;
        int t = tile_lin;
        const int tw = t % p.nTW; t /= p.nTW; const int th = t % p.nTH; t /= p.nTH; const int td = t % p.nTD; t /= p.nTD; const int tn = t;
        const int n0 = cb * BN;
        const int lgTW = p.lgTW, lgTH = p.lgTH, lgTD = p.lgTD;
        const int lgS = p.lgTW + p.lgTH + p.lgTD;
        const int mW = (1 << p.lgTW) - 1, mH = (1 << p.lgTH) - 1, mD = (1 << p.lgTD) - 1;
        const int wpx = wave % WVP, wch = wave / WVP;
        const int l15 = lane & 15, l4 = lane >> 4;
#ifdef HARNESS_NO_PAIRING
        const int l15p = l15;
#else
        const int l15p = ((l15 & 1) << 1) | ((l15 & 2) << 1) | ((l15 >> 2) & 1) | (l15 & 8);      // any bijection: the volume kernels' one
#endif
#ifdef HARNESS_NO_PAIRING
        constexpr int EP_PAIR = 0;
#else
        constexpr int EP_PAIR = ep_pair_of(MODE, WCH);
#endif
        // the accumulators as the kernel's MFMAs would leave them: a function of (packed weight row, position) only, so the
        // checksums do not depend on which lane / fragment a row is assigned to (EP_PAIR)
        f4_t acc[WCH][WPX];
        for (int ci = 0; ci < WCH; ++ci) for (int pi = 0; pi < WPX; ++pi) for (int r = 0; r < 4; ++r) {
            const int row = n0 + wch * WCH * 16 + ep_frag_row(EP_PAIR, ci) + ep_lane_row(EP_PAIR, l4 * 4 + r);
            const int m = (wpx * WPX + pi) * 16 + l15p;
            acc[ci][pi][r] = rnd((uint32_t)((tile * 1024 + m) * 2053 + row * 31 + 7));
        }
#ifdef HARNESS_NO_PAIRING
        constexpr bool EP_EARLY = false;
#else
        constexpr bool EP_EARLY = (MODE == MODE_SPADE) && WCH == 2 && WPX == 8;      // as conv_halo_kernel.h
#endif
        constexpr int EP_WPX0 = WPX;
        const int ep_wpx0 = wpx;
        CONV_EPILOGUE_EARLY_FETCH()
        constexpr int EP_WPX = WPX;
        const int ep_wpx = wpx;
        auto& ep_acc = acc;
        CONV_EPILOGUE()
    }
}

static std::vector<float> F(size_t n, uint32_t seed) { std::vector<float> v(n); for (size_t i = 0; i < n; ++i) v[i] = rnd(seed + (uint32_t)i); return v; }
static std::vector<half_t> H(size_t n, uint32_t seed) { std::vector<half_t> v(n); for (size_t i = 0; i < n; ++i) v[i] = (half_t)rnd(seed + (uint32_t)i); return v; }
static uint64_t crc(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; uint64_t h = 1469598103934665603ULL; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ULL; } return h; }

static void tile_of(ConvParams& p, int BM, int tw, int th) {
    int td = BM / (tw * th); if (td > p.D) td = p.D; int tn = BM / (tw * th * td);
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    p.lgTW = lg(tw); p.lgTH = lg(th); p.lgTD = lg(td); p.nTW = p.W / tw; p.nTH = p.H / th; p.nTD = p.D / td; p.nTN = (p.N + tn - 1) / tn;
}

int main()
{
    // case 1: 2-D 16x8 tile, 128x128 STD, f32 residual, out1 with affine, pixscale, Cout 200 of 256 (ragged channels)
    {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 3; p.D = 1; p.H = 16; p.W = 32; p.Cout = 200; p.Cout_pad = 256; tile_of(p, 128, 16, 8);
        const size_t P = (size_t)p.N * p.H * p.W;
        auto res = F(P * 256, 1), ps = F(P * 4, 2), bias = F(256, 3), s2 = F(256, 4), t2 = F(256, 5);
        std::vector<float> out0(P * 300, -9.f); std::vector<half_t> out1(P * 256, (half_t)-9.f);
        p.res = TDesc{res.data(), (long)p.H * p.W * 256, 0, (long)p.W * 256, 256}; p.res_f32 = 1;
        p.out0 = TDesc{out0.data(), (long)p.H * p.W * 300, 0, (long)p.W * 300, 300}; p.out0_f32 = 1;
        p.out1 = TDesc{out1.data(), (long)p.H * p.W * 256, 0, (long)p.W * 256, 256};
        p.pixscale = ps.data(); p.ps_stride = 4; p.bias = bias.data(); p.s2 = s2.data(); p.t2 = t2.data(); p.act0 = ACT_LRELU; p.slope0 = 0.2f; p.act1 = ACT_RELU;
        run<8, 2, 1, 4, MODE_STD, false>(p, 128);
        printf("case1 %016llx %016llx\n", (unsigned long long)crc(out0.data(), out0.size() * 4), (unsigned long long)crc(out1.data(), out1.size() * 2));
    }
    // case 2: volume 4x4x16 tile (256 positions), 256x32, f32 residual, hwdc strides, N = 2
    {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 2; p.D = 16; p.H = 8; p.W = 8; p.Cout = 32; p.Cout_pad = 32; tile_of(p, 256, 4, 4);
        const size_t V = (size_t)p.N * p.H * p.W * p.D * 32;
        auto res = F(V, 11), bias = F(32, 12); std::vector<float> out0(V, -9.f); std::vector<half_t> out1(V, (half_t)-9.f);
        TDesc d{nullptr, (long)p.H * p.W * p.D * 32, 32, (long)p.W * p.D * 32, (long)p.D * 32};
        p.res = d; p.res.p = res.data(); p.res_f32 = 1; p.out0 = d; p.out0.p = out0.data(); p.out0_f32 = 1; p.out1 = d; p.out1.p = out1.data();
        p.bias = bias.data(); p.ps_stride = 1; p.act1 = ACT_RELU;
        std::vector<float> st((size_t)p.N * 4 * 4 * 32 * 2 * 8, -9.f); p.stat_out = st.data();
        run<4, 2, 4, 1, MODE_STDSTAT, false>(p, 256);
        printf("case2 %016llx %016llx %016llx\n", (unsigned long long)crc(out0.data(), out0.size() * 4), (unsigned long long)crc(out1.data(), out1.size() * 2), (unsigned long long)crc(st.data(), st.size() * 4));
    }
    // case 3: SPADE with up-sampled x (res_shift 1), f16 x, 128x128
    {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 2; p.D = 1; p.H = 16; p.W = 16; p.Cout = 64; p.Cout_pad = 128; tile_of(p, 128, 16, 8);
        const size_t P = (size_t)p.N * p.H * p.W, Px = (size_t)p.N * 8 * 8;
        auto x = H(Px * 64, 21); auto bias = F(64, 22), bias2 = F(64, 23), stats = F(p.N * 64 * 2, 24);
        std::vector<half_t> out0(P * 64, (half_t)-9.f);
        p.res = TDesc{x.data(), 8L * 8 * 64, 0, 8L * 64, 64}; p.res_f32 = 0; p.res_shift = 1;
        p.out0 = TDesc{out0.data(), (long)p.H * p.W * 64, 0, (long)p.W * 64, 64};
        p.bias = bias.data(); p.bias2 = bias2.data(); p.stats = stats.data(); p.act0 = ACT_LRELU; p.slope0 = 0.2f; p.ps_stride = 1;
        run<8, 2, 1, 4, MODE_SPADE, false>(p, 128);
        printf("case3 %016llx\n", (unsigned long long)crc(out0.data(), out0.size() * 2));
    }
    // case 4: T blend 128x256, pixscale stride 4, f32 residual + outputs
    {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 2; p.D = 1; p.H = 16; p.W = 16; p.Cout = 128; p.Cout_pad = 256; tile_of(p, 128, 16, 8);
        const size_t P = (size_t)p.N * p.H * p.W;
        auto res = F(P * 128, 31), ps = F(P * 4, 32), bias = F(128, 33);
        std::vector<float> out0(P * 128, -9.f); std::vector<half_t> out1(P * 128, (half_t)-9.f);
        TDesc d{nullptr, (long)p.H * p.W * 128, 0, (long)p.W * 128, 128};
        p.res = d; p.res.p = res.data(); p.res_f32 = 1; p.out0 = d; p.out0.p = out0.data(); p.out0_f32 = 1; p.out1 = d; p.out1.p = out1.data();
        p.pixscale = ps.data(); p.ps_stride = 4; p.bias = bias.data(); p.act0 = ACT_NONE;
        run<8, 4, 1, 4, MODE_TBLEND, false>(p, 128);
        printf("case4 %016llx %016llx\n", (unsigned long long)crc(out0.data(), out0.size() * 4), (unsigned long long)crc(out1.data(), out1.size() * 2));
    }
    // case 5: pixel shuffle + sigmoid, 256x16
    {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 2; p.D = 1; p.H = 16; p.W = 16; p.Cout = 16; p.Cout_pad = 16; tile_of(p, 256, 16, 16);
        auto bias = F(16, 41); std::vector<float> img((size_t)p.N * 3 * 32 * 32, -9.f);
        p.out0 = TDesc{img.data(), 0, 0, 0, 0}; p.out0_f32 = 1; p.bias = bias.data(); p.act0 = ACT_SIGMOID; p.ps_stride = 1;
        run<4, 1, 4, 1, MODE_PIXSHUF, true>(p, 256);
        printf("case5 %016llx\n", (unsigned long long)crc(img.data(), img.size() * 4));
    }
    // case 6: tiny spatial (2x2x16 = 64 positions per sample, two samples per 128-position tile), odd batch, mask-like 2x8x8 tile too
    {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 3; p.D = 16; p.H = 2; p.W = 2; p.Cout = 128; p.Cout_pad = 128; tile_of(p, 128, 2, 2);
        const size_t P = (size_t)p.N * p.D * p.H * p.W;
        auto res = H(P * 128, 51); auto bias = F(128, 52); std::vector<half_t> out0(P * 160, (half_t)-9.f);
        p.res = TDesc{res.data(), (long)p.D * p.H * p.W * 128, (long)p.H * p.W * 128, (long)p.W * 128, 128}; p.res_f32 = 0;
        p.out0 = TDesc{out0.data(), (long)p.D * p.H * p.W * 160, (long)p.H * p.W * 160, (long)p.W * 160, 160};
        p.bias = bias.data(); p.act0 = ACT_RELU; p.ps_stride = 1;
        run<8, 2, 1, 4, MODE_STD, false>(p, 128);
        printf("case6 %016llx\n", (unsigned long long)crc(out0.data(), out0.size() * 2));
        ConvParams q; memset(&q, 0, sizeof q);
        q.N = 2; q.D = 16; q.H = 16; q.W = 4; q.Cout = 160; q.Cout_pad = 160; tile_of(q, 128, 2, 8);
        const size_t Q = (size_t)q.N * q.D * q.H * q.W; std::vector<float> o(Q * 160, -9.f);
        q.out0 = TDesc{o.data(), (long)q.D * q.H * q.W * 160, (long)q.H * q.W * 160, (long)q.W * 160, 160}; q.out0_f32 = 1; q.ps_stride = 1;
        run<4, 5, 2, 2, MODE_STD, false>(q, 128);
        printf("case6b %016llx\n", (unsigned long long)crc(o.data(), o.size() * 4));
    }
    // case 7: channel count that ends in the middle of a lane's 8-channel pair (196 = 24 * 8 + 4), fp16 residual, both outputs fp16;
    // 7b: the same with an fp32 first output and the T-blend kernel shape on 100 of 128 output channels (2 x 128 packed rows)
    {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 2; p.D = 1; p.H = 8; p.W = 32; p.Cout = 196; p.Cout_pad = 256; tile_of(p, 128, 16, 8);
        const size_t P = (size_t)p.N * p.H * p.W;
        auto res = H(P * 200, 61); auto bias = F(256, 62), s2 = F(256, 63), t2 = F(256, 64);
        std::vector<half_t> out0(P * 200, (half_t)-9.f), out1(P * 196, (half_t)-9.f);
        p.res = TDesc{res.data(), (long)p.H * p.W * 200, 0, (long)p.W * 200, 200}; p.res_f32 = 0;
        p.out0 = TDesc{out0.data(), (long)p.H * p.W * 200, 0, (long)p.W * 200, 200};
        p.out1 = TDesc{out1.data(), (long)p.H * p.W * 196, 0, (long)p.W * 196, 196};
        p.bias = bias.data(); p.s2 = s2.data(); p.t2 = t2.data(); p.act0 = ACT_LRELU; p.slope0 = 0.1f; p.act1 = ACT_RELU; p.ps_stride = 1;
        run<8, 2, 1, 4, MODE_STD, false>(p, 128);
        printf("case7 %016llx %016llx\n", (unsigned long long)crc(out0.data(), out0.size() * 2), (unsigned long long)crc(out1.data(), out1.size() * 2));
        ConvParams q; memset(&q, 0, sizeof q);
        q.N = 1; q.D = 1; q.H = 16; q.W = 16; q.Cout = 100; q.Cout_pad = 256; tile_of(q, 128, 16, 8);
        const size_t Q = (size_t)q.N * q.H * q.W;
        auto r2 = F(Q * 100, 71), ps = F(Q, 72), b2 = F(128, 73);
        std::vector<float> o0(Q * 100, -9.f); std::vector<half_t> o1(Q * 100, (half_t)-9.f);
        q.res = TDesc{r2.data(), (long)q.H * q.W * 100, 0, (long)q.W * 100, 100}; q.res_f32 = 1;
        q.out0 = TDesc{o0.data(), (long)q.H * q.W * 100, 0, (long)q.W * 100, 100}; q.out0_f32 = 1;
        q.out1 = TDesc{o1.data(), (long)q.H * q.W * 100, 0, (long)q.W * 100, 100};
        q.bias = b2.data(); q.pixscale = ps.data(); q.ps_stride = 1; q.act1 = ACT_RELU;
        run<8, 4, 1, 4, MODE_TBLEND, false>(q, 128);
        printf("case7b %016llx %016llx\n", (unsigned long long)crc(o0.data(), o0.size() * 4), (unsigned long long)crc(o1.data(), o1.size() * 2));
    }
    // case 8: statistics are emitted per 64 positions: the 128x128 kernel shape (a wave owns 128 positions) and the 128x64 shape
    // (64 per wave) must leave the same partial sums and the same output
    {
        uint64_t c0[2], c1[2];
        for (int v = 0; v < 2; ++v) {
            ConvParams p; memset(&p, 0, sizeof p);
            p.N = 2; p.D = 1; p.H = 16; p.W = 32; p.Cout = 120; p.Cout_pad = 128; tile_of(p, 128, 16, 8);
            const size_t P = (size_t)p.N * p.H * p.W;
            auto bias = F(128, 81); auto res = H(P * 120, 82);
            std::vector<half_t> out0(P * 120, (half_t)-9.f);
            std::vector<float> st((size_t)p.N * (p.H * p.W / 64) * 120 * 2, -9.f);
            p.res = TDesc{res.data(), (long)p.H * p.W * 120, 0, (long)p.W * 120, 120};
            p.out0 = TDesc{out0.data(), (long)p.H * p.W * 120, 0, (long)p.W * 120, 120};
            p.bias = bias.data(); p.act0 = ACT_RELU; p.ps_stride = 1; p.stat_out = st.data();
            if (v == 0) run<8, 2, 1, 4, MODE_STDSTAT, false>(p, 128); else run<4, 2, 2, 2, MODE_STDSTAT, false>(p, 128);
            c0[v] = crc(out0.data(), out0.size() * 2); c1[v] = crc(st.data(), st.size() * 4);
        }
        printf("case8 %016llx %016llx same=%d\n", (unsigned long long)c0[0], (unsigned long long)c1[0], (int)(c0[0] == c0[1] && c1[0] == c1[1]));
    }
    // cases 9-12: the tensor combinations that have a branch-free copy (EP_FAST builds take it, HARNESS_NO_FAST builds the general path)
    // case 9: SPADE 128x256, fp16 x at the same and at half resolution, every channel valid
    for (int sh = 0; sh < 2; ++sh) {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 2; p.D = 1; p.H = 16; p.W = 16; p.Cout = 256; p.Cout_pad = 512; tile_of(p, 128, 16, 8);
        const int Sx = 16 >> sh;
        const size_t P = (size_t)p.N * p.H * p.W, Px = (size_t)p.N * Sx * Sx;
        auto x = H(Px * 256, 91); auto bias = F(256, 92), bias2 = F(256, 93), stats = F(p.N * 256 * 2, 94);
        std::vector<half_t> out0(P * 256, (half_t)-9.f);
        p.res = TDesc{x.data(), (long)Sx * Sx * 256, 0, (long)Sx * 256, 256}; p.res_f32 = 0; p.res_shift = sh;
        p.out0 = TDesc{out0.data(), (long)p.H * p.W * 256, 0, (long)p.W * 256, 256};
        p.bias = bias.data(); p.bias2 = bias2.data(); p.stats = stats.data(); p.act0 = ACT_LRELU; p.slope0 = 0.2f; p.ps_stride = 1;
        run<8, 4, 1, 4, MODE_SPADE, false>(p, 128);
        printf("case9.%d %016llx\n", sh, (unsigned long long)crc(out0.data(), out0.size() * 2));
    }
    // case 10: 128x256 and 128x128 STD / STDSTAT, fp16 output, without and with an fp16 residual
    for (int v = 0; v < 4; ++v) {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 3; p.D = 1; p.H = 16; p.W = 32; p.Cout = 256; p.Cout_pad = 256; tile_of(p, 128, 16, 8);
        const size_t P = (size_t)p.N * p.H * p.W;
        auto bias = F(256, 101); auto res = H(P * 256, 102);
        std::vector<half_t> out0(P * 256, (half_t)-9.f);
        std::vector<float> st((size_t)p.N * (p.H * p.W / 64) * 256 * 2, -9.f);
        if (v & 1) p.res = TDesc{res.data(), (long)p.H * p.W * 256, 0, (long)p.W * 256, 256};
        p.out0 = TDesc{out0.data(), (long)p.H * p.W * 256, 0, (long)p.W * 256, 256};
        p.bias = bias.data(); p.act0 = ACT_LRELU; p.slope0 = 0.2f; p.ps_stride = 1;
        if (v & 2) { p.stat_out = st.data(); run<8, 4, 1, 4, MODE_STDSTAT, false>(p, 128); run<8, 2, 1, 4, MODE_STDSTAT, false>(p, 128); }
        else { run<8, 4, 1, 4, MODE_STD, false>(p, 128); }
        printf("case10.%d %016llx %016llx\n", v, (unsigned long long)crc(out0.data(), out0.size() * 2), (unsigned long long)crc(st.data(), st.size() * 4));
    }
    // case 11: 128x256 STD, fp32 residual stream in and out + fp16 copy through an affine and LeakyReLU (R's 2-D blocks)
    {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 2; p.D = 1; p.H = 16; p.W = 16; p.Cout = 256; p.Cout_pad = 256; tile_of(p, 128, 16, 8);
        const size_t P = (size_t)p.N * p.H * p.W;
        auto res = F(P * 256, 111), bias = F(256, 112), s2 = F(256, 113), t2 = F(256, 114);
        std::vector<float> out0(P * 256, -9.f); std::vector<half_t> out1(P * 256, (half_t)-9.f);
        TDesc d{nullptr, (long)p.H * p.W * 256, 0, (long)p.W * 256, 256};
        p.res = d; p.res.p = res.data(); p.res_f32 = 1; p.out0 = d; p.out0.p = out0.data(); p.out0_f32 = 1; p.out1 = d; p.out1.p = out1.data();
        p.bias = bias.data(); p.s2 = s2.data(); p.t2 = t2.data(); p.act1 = ACT_LRELU; p.slope1 = 0.01f; p.ps_stride = 1;
        run<8, 4, 1, 4, MODE_STD, false>(p, 128);
        printf("case11 %016llx %016llx\n", (unsigned long long)crc(out0.data(), out0.size() * 4), (unsigned long long)crc(out1.data(), out1.size() * 2));
    }
    // case 12: 256x160 volume tile (8x8x4), 144 of 160 channels (the second channel wave takes the ragged copy), fp16 output, ReLU
    {
        ConvParams p; memset(&p, 0, sizeof p);
        p.N = 2; p.D = 8; p.H = 8; p.W = 8; p.Cout = 144; p.Cout_pad = 160; tile_of(p, 256, 8, 8);
        const size_t V = (size_t)p.N * p.D * p.H * p.W;
        auto bias = F(160, 121); std::vector<half_t> out0(V * 144, (half_t)-9.f);
        p.out0 = TDesc{out0.data(), (long)p.D * p.H * p.W * 144, (long)p.H * p.W * 144, (long)p.W * 144, 144};
        p.bias = bias.data(); p.act0 = ACT_RELU; p.ps_stride = 1;
        run<8, 5, 2, 2, MODE_STD, false>(p, 256);
        printf("case12 %016llx\n", (unsigned long long)crc(out0.data(), out0.size() * 2));
    }
    return 0;
}
