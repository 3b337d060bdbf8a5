// Shared epilogue of the gfx950 conv kernels (conv_igemm.hip, conv_halo.hip).
// Accumulator layout (v_mfma_f32_16x16x32_f16 with weights as the A operand): acc[ci][pi][r] is
// channel (16-block ci, row l4*4 + r) of position (16-block pi, column l15).
#pragma once
#include "common.h"

__device__ __forceinline__ float apply_act(float v, int act, float slope)
{
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    if (act == ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    return v;
}

__device__ __forceinline__ void load4(const TDesc& t, int is_f32, long off, float v[4])
{
    if (is_f32) {
        const float4 x = *(const float4*)((const float*)t.p + off);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    } else {
        const h4_t x = *(const h4_t*)((const half_t*)t.p + off);
        v[0] = (float)x[0]; v[1] = (float)x[1]; v[2] = (float)x[2]; v[3] = (float)x[3];
    }
}

__device__ __forceinline__ void store4(const TDesc& t, int is_f32, long off, const float v[4])
{
    if (is_f32) {
        *(float4*)((float*)t.p + off) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        h4_t x;
        x[0] = (half_t)v[0]; x[1] = (half_t)v[1]; x[2] = (half_t)v[2]; x[3] = (half_t)v[3];
        *(h4_t*)((half_t*)t.p + off) = x;
    }
}


// Expects in scope: p, ep_acc[WCH][EP_WPX] (f4_t), ep_wpx (position-block index of this wave), EP_WPX, n0, tw, th, td, tn,
// lgS, mW, mH, mD, wch, l15, l4 and the template constants WCH, BM, MODE.
#define CONV_EPILOGUE() \
    constexpr int CSTEP = (MODE == MODE_TBLEND || MODE == MODE_SPADE) ? 2 : 1; \
    /* per-channel constants of this lane's 4 channels, loaded once (16-byte loads), not once per position block */ \
    float4 ep_bias[WCH], ep_bias2[WCH], ep_s2[WCH], ep_t2[WCH], ep_mean[WCH], ep_rstd[WCH]; \
_Pragma("unroll") \
    for (int ci = 0; ci < WCH; ci += CSTEP) { \
        const int pb = (n0 + wch * WCH * 16) / 16 + ci; \
        const int cb = (CSTEP == 2 ? (pb >> 1) : pb) * 16 + l4 * 4; \
        const bool cok = cb < p.Cout; \
        ep_bias[ci] = (cok && p.bias) ? *(const float4*)(p.bias + cb) : make_float4(0.f, 0.f, 0.f, 0.f); \
        ep_bias2[ci] = (cok && MODE == MODE_SPADE) ? *(const float4*)(p.bias2 + cb) : make_float4(0.f, 0.f, 0.f, 0.f); \
        ep_s2[ci] = (cok && p.s2) ? *(const float4*)(p.s2 + cb) : make_float4(1.f, 1.f, 1.f, 1.f); \
        ep_t2[ci] = (cok && p.s2) ? *(const float4*)(p.t2 + cb) : make_float4(0.f, 0.f, 0.f, 0.f); \
        if (MODE == MODE_SPADE && cok) { /* SPADE launches tile within one sample: n == tn */ \
            const float4 q0 = *(const float4*)(p.stats + ((long)tn * p.Cout + cb) * 2); \
            const float4 q1 = *(const float4*)(p.stats + ((long)tn * p.Cout + cb) * 2 + 4); \
            ep_mean[ci] = make_float4(q0.x, q0.z, q1.x, q1.z); ep_rstd[ci] = make_float4(q0.y, q0.w, q1.y, q1.w); \
        } else { ep_mean[ci] = make_float4(0.f, 0.f, 0.f, 0.f); ep_rstd[ci] = make_float4(1.f, 1.f, 1.f, 1.f); } \
    } \
    constexpr bool EP_STAT = (MODE == MODE_STDSTAT); /* compile-time: the accumulators cost occupancy otherwise */ \
    constexpr int EP_SC = EP_STAT ? WCH : 1; \
    float ep_sum[EP_SC][4], ep_sq[EP_SC][4]; \
_Pragma("unroll") \
    for (int ci = 0; ci < EP_SC; ++ci) \
_Pragma("unroll") \
        for (int r = 0; r < 4; ++r) { ep_sum[ci][r] = 0.f; ep_sq[ci][r] = 0.f; } \
    /* The residual / modulated tensor is fetched for EP_G position blocks at a time BEFORE any of their stores: inside the \
       per-block loop every load would wait for its own round trip (hipcc may not hoist a load above the previous block's \
       stores, res and out0 may alias) - 8 to 16 serialised memory latencies per wave.  In-place use (res == out0) stays \
       correct: a lane reads exactly the elements it later writes. */ \
    /* Compiled in only for the 256x32 STD kernel (the residual convs of the ResBlock3d chains, -9 % there): elsewhere the extra \
       live registers cost an occupancy step (mask / tail convs +30..40 %, SPADE +5 %), measured per layer. */ \
    constexpr bool EP_PF = (MODE == MODE_STD) && (WCH == 2) && (EP_WPX == 4); \
    constexpr int EP_G = EP_PF ? 4 : 1; \
    constexpr int EP_NCI = EP_PF ? (WCH + CSTEP - 1) / CSTEP : 1; \
    const bool ep_fetch = EP_PF && p.res.p != nullptr; \
    const int ep_rshift = 0; \
_Pragma("unroll") \
    for (int pg = 0; pg < EP_WPX; pg += EP_G) { \
    f4_t ep_raw32[EP_G][EP_NCI]; h4_t ep_raw16[EP_G][EP_NCI]; float ep_ps[EP_G]; \
    if (EP_PF && (ep_fetch || p.pixscale)) { \
_Pragma("unroll") \
        for (int g = 0; g < EP_G; ++g) { \
            int m = ep_wpx * EP_WPX * 16 + (pg + g) * 16 + l15; \
            const int w = (tw << p.lgTW) + (m & mW); m >>= p.lgTW; \
            const int h = (th << p.lgTH) + (m & mH); m >>= p.lgTH; \
            const int d = (td << p.lgTD) + (m & mD); m >>= p.lgTD; \
            const int n = tn * (BM >> lgS) + m; \
            if (n >= p.N) continue; \
            if (p.pixscale) ep_ps[g] = p.pixscale[((((long)n * p.D + d) * p.H + h) * p.W + w) * p.ps_stride]; \
            if (ep_fetch) { \
_Pragma("unroll") \
                for (int ci = 0; ci < WCH; ci += CSTEP) { \
                    const int pb = (n0 + wch * WCH * 16) / 16 + ci; \
                    const int cb = (CSTEP == 2 ? (pb >> 1) : pb) * 16 + l4 * 4; \
                    if (cb >= p.Cout) continue; \
                    const long xo = (long)n * p.res.sN + (long)d * p.res.sD + (long)(h >> ep_rshift) * p.res.sH + \
                                    (long)(w >> ep_rshift) * p.res.sW + cb; \
                    if (p.res_f32) ep_raw32[g][(EP_PF ? ci / CSTEP : 0)] = *(const f4_t*)((const float*)p.res.p + xo); \
                    else ep_raw16[g][(EP_PF ? ci / CSTEP : 0)] = *(const h4_t*)((const half_t*)p.res.p + xo); \
                } \
            } \
        } \
    } \
_Pragma("unroll") \
    for (int g = 0; g < EP_G; ++g) { \
        const int pi = pg + g; \
        int m = ep_wpx * EP_WPX * 16 + pi * 16 + l15; \
        const int w = (tw << p.lgTW) + (m & mW); m >>= p.lgTW; \
        const int h = (th << p.lgTH) + (m & mH); m >>= p.lgTH; \
        const int d = (td << p.lgTD) + (m & mD); m >>= p.lgTD; \
        const int n = tn * (BM >> lgS) + m; \
        if (n >= p.N) continue; \
        float ps = 1.f; \
        if (p.pixscale) ps = EP_PF ? ep_ps[g] : p.pixscale[((((long)n * p.D + d) * p.H + h) * p.W + w) * p.ps_stride]; \
_Pragma("unroll") \
        for (int ci = 0; ci < WCH; ci += CSTEP) { \
            const int pb = (n0 + wch * WCH * 16) / 16 + ci; \
            const int cb = (CSTEP == 2 ? (pb >> 1) : pb) * 16 + l4 * 4; \
            if (cb >= p.Cout) continue; \
            float v[4]; \
            if (MODE == MODE_TBLEND) { \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) \
                    v[r] = ps * (ep_acc[ci + CSTEP - 1][pi][r] + ((const float*)&ep_bias[ci])[r]) + (1.f - ps) * ep_acc[ci][pi][r]; \
            } else if (MODE == MODE_SPADE) { \
                float x[4]; \
                const long xo = (long)n * p.res.sN + (long)d * p.res.sD + (long)(h >> p.res_shift) * p.res.sH + \
                                (long)(w >> p.res_shift) * p.res.sW + cb; \
                load4(p.res, p.res_f32, xo, x); \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) { \
                    const float g = ep_acc[ci][pi][r] + ((const float*)&ep_bias[ci])[r]; \
                    const float b = ep_acc[ci + CSTEP - 1][pi][r] + ((const float*)&ep_bias2[ci])[r]; \
                    v[r] = (x[r] - ((const float*)&ep_mean[ci])[r]) * ((const float*)&ep_rstd[ci])[r] * (1.f + g) + b; \
                } \
            } else { \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) v[r] = ep_acc[ci][pi][r] + ((const float*)&ep_bias[ci])[r]; \
            } \
_Pragma("unroll") \
            for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act0, p.slope0); \
            if (MODE == MODE_PIXSHUF) { \
                const int c = cb >> 2; \
                if (c < 3) { \
                    float* o = (float*)p.out0.p; \
                    const long W2 = 2L * p.W, H2 = 2L * p.H; \
                    const long base = (((long)n * 3 + c) * H2 + 2 * h) * W2 + 2 * w; \
                    *(float2*)(o + base) = make_float2(v[0], v[1]); \
                    *(float2*)(o + base + W2) = make_float2(v[2], v[3]); \
                } \
                continue; \
            } \
            if (MODE != MODE_SPADE && p.res.p) { \
                if (EP_PF) { \
_Pragma("unroll") \
                    for (int r = 0; r < 4; ++r) v[r] += p.res_f32 ? ep_raw32[g][(EP_PF ? ci / CSTEP : 0)][r] : (float)ep_raw16[g][(EP_PF ? ci / CSTEP : 0)][r]; \
                } else { \
                    float rr[4]; \
                    load4(p.res, p.res_f32, (long)n * p.res.sN + (long)d * p.res.sD + (long)h * p.res.sH + (long)w * p.res.sW + cb, rr); \
_Pragma("unroll") \
                    for (int r = 0; r < 4; ++r) v[r] += rr[r]; \
                } \
            } \
            if ((MODE == MODE_STD || MODE == MODE_STDSTAT) && p.pixscale) { \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) v[r] *= ps; \
            } \
            if (p.out0.p) \
                store4(p.out0, p.out0_f32, (long)n * p.out0.sN + (long)d * p.out0.sD + (long)h * p.out0.sH + (long)w * p.out0.sW + cb, v); \
            if (EP_STAT) { /* statistics of the values as stored (fp16-rounded when out0 is fp16) */ \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) { \
                    const float vs = p.out0_f32 ? v[r] : (float)(half_t)v[r]; \
                    ep_sum[EP_STAT ? ci : 0][r] += vs; ep_sq[EP_STAT ? ci : 0][r] = fmaf(vs, vs, ep_sq[EP_STAT ? ci : 0][r]); \
                } \
            } \
            if (p.out1.p) { \
                float u[4]; \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) { \
                    const float a = v[r] * ((const float*)&ep_s2[ci])[r] + ((const float*)&ep_t2[ci])[r]; \
                    u[r] = apply_act(a, p.act1, p.slope1); \
                } \
                store4(p.out1, 0, (long)n * p.out1.sN + (long)d * p.out1.sD + (long)h * p.out1.sH + (long)w * p.out1.sW + cb, u); \
            } \
        } \
    } \
    } \
    if (EP_STAT) { /* fixed-order butterfly over the 16 position lanes, then one partial per (tile, wave, channel) */ \
        const int ep_tiles = p.nTW * p.nTH * p.nTD; \
        const int ep_nblk = ep_tiles * (BM / (EP_WPX * 16)); \
        const int ep_blk = (blockIdx.x % ep_tiles) * (BM / (EP_WPX * 16)) + ep_wpx; \
_Pragma("unroll") \
        for (int ci = 0; ci < WCH; ci += CSTEP) { \
            const int pb = (n0 + wch * WCH * 16) / 16 + ci; \
            const int cb = (CSTEP == 2 ? (pb >> 1) : pb) * 16 + l4 * 4; \
_Pragma("unroll") \
            for (int r = 0; r < 4; ++r) { \
                float a = ep_sum[EP_STAT ? ci : 0][r], b = ep_sq[EP_STAT ? ci : 0][r]; \
_Pragma("unroll") \
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); } \
                if (l15 == 0 && cb < p.Cout) { \
                    float* dst = p.stat_out + (((long)tn * ep_nblk + ep_blk) * p.Cout + cb + r) * 2; \
                    dst[0] = a; dst[1] = b; \
                } \
            } \
        } \
    }


