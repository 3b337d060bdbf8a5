// Shared epilogue of the gfx950 conv kernels (conv_halo.hip, vol32.hip; also the test-only tests/csrc/conv_igemm.hip).
// Accumulator layout (v_mfma_f32_16x16x32_f16 with weights as the A operand): acc[ci][pi][r] is
// channel (16-block ci, row l4*4 + r) of position (16-block pi, column l15).
#pragma once
#include "common.h"

#ifndef EP_TL
#define EP_TL(i) do { } while (0)      /* phase stamps of instrumented builds (conv_halo_kernel.h, -DCS_TIMELINE) */
#endif
typedef unsigned int ep_u4_t __attribute__((ext_vector_type(4)));
typedef unsigned int ep_u2_t __attribute__((ext_vector_type(2)));

// Host side: the epilogue addresses its tensors with 32-bit element offsets and 24-bit per-axis multiplies; both conv launchers
// call this before launching.
static inline int ep_check_extents(const ConvParams& p, const char* who)
{
    auto span = [&](const TDesc& t) -> long {
        return (long)(p.N - 1) * t.sN + (long)(p.D - 1) * t.sD + (long)(p.H - 1) * t.sH + (long)(p.W - 1) * t.sW + p.Cout;
    };
    auto wide = [&](const TDesc& t) { return t.sD >= (1L << 23) || t.sH >= (1L << 23) || t.sW >= (1L << 23); };
    const long lim = 1L << 31;
    long ph_max = 0;          // a grouped launch (ConvParams::nphase) adds its phase's element offset to every out0 address
    for (int z = 0; z < p.nphase && z < 4; ++z) ph_max = p.ph_ooff[z] > ph_max ? (long)p.ph_ooff[z] : ph_max;
    if ((p.out0.p && span(p.out0) + ph_max >= lim) || (p.out1.p && span(p.out1) >= lim) || (p.res.p && span(p.res) >= lim)) {
        cs_set_error("%s: a tensor of this launch spans 2^31 elements or more (32-bit in-tensor offsets)", who);
        return -1;
    }
    if ((p.out0.p && wide(p.out0)) || (p.out1.p && wide(p.out1)) || (p.res.p && wide(p.res)) || p.D >= (1 << 22) || p.H >= (1 << 22) ||
        p.W >= (1 << 22)) {
        cs_set_error("%s: an axis stride of 2^23 elements or more (24-bit multiplies in the addressing)", who);
        return -1;
    }
    return 0;
}

__device__ __forceinline__ float gelu_act(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }      // nn.GELU (exact erf form)
__device__ __forceinline__ float apply_act(float v, int act, float slope)
{
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    if (act == ACT_GELU) return gelu_act(v);
    return v;
}

// ReLU / LeakyReLU / identity as ONE branch-free formula: slope 0 / s / 1 (bit-identical to the switch above for those three).
// The epilogue is fully unrolled; a runtime switch per element (with the erf polynomial of GELU inlined each time) made it 18 000
// instructions and ~1 800 branches long, fetch-bound at 13-38 % of a wave's lifetime (profiles/r02_timeline_*.txt).
// NaN: fminf / fmaxf return the non-NaN operand, so a NaN accumulator (inf - inf after an fp16 overflow upstream) is stored as 0 by
// this formula instead of propagating.  Kept: `v > 0 ? v : v * slope` would propagate NaN but turns relu(-inf) into
// -inf * 0 = NaN, and the two-instruction forms cost an issue slot more in a VALU-issue-bound epilogue.  Overflow is watched where it
// would start: CANONSWAP_AMAX=1 reports the largest |value| every conv stored in fp16 (inf shows up there).
__device__ __forceinline__ float lin_act(float v, float slope) { return fmaf(slope, fminf(v, 0.f), fmaxf(v, 0.f)); }
__device__ __forceinline__ float lin_slope(int act, float slope) { return act == ACT_NONE ? 1.f : (act == ACT_LRELU ? slope : 0.f); }

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

// Ablation build (tools/build_variant.py, not the product): -DCS_EP_NOSTORE puts every epilogue store behind a runtime condition
// that is never true (what the stores cost: profiles/r02_store_ablation.txt).
// fp32 -> fp16 of values the epilogue computed: round-to-nearest-even of the fp32 RESULT.  Left to itself hipcc folds a preceding fma
// into v_fma_mix{lo,hi}_f16 (one rounding, straight from the exact product-sum to fp16; an instruction-selection pattern, not governed by
// the contraction pragma) for some elements and not for others, depending on the code around it - the branch-free copies of the epilogue
// then differ from the general one in the last bit of values that sit on an fp16 rounding boundary (tests/test_gpu_epilogue_fast.py).
// The conversion is therefore written as the instruction the compiler uses anyway, where it cannot be folded into.
__device__ __forceinline__ unsigned ep_pk(float a, float b)          // (fp16(a), fp16(b)) in one register
{
#ifdef EP_HOST_EMULATION          /* tests/epilogue_host: the macro executed on the host */
    const half_t ha = (half_t)a, hb = (half_t)b;
    return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
#else
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
__device__ __forceinline__ half_t ep_h(float v)
{
    const unsigned r = ep_pk(v, v);
    return __builtin_bit_cast(half_t, (unsigned short)(r & 0xFFFFu));
}

__device__ __forceinline__ void store4(const TDesc& t, int is_f32, long off, const float v[4])
{
    if (is_f32) {
        *(float4*)((float*)t.p + off) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        ep_u2_t x;
        x[0] = ep_pk(v[0], v[1]); x[1] = ep_pk(v[2], v[3]);
        *(ep_u2_t*)((half_t*)t.p + off) = x;
    }
}

__device__ __forceinline__ void store8(const TDesc& t, int is_f32, long off, const float a[4], const float b[4])
{
    if (is_f32) {
        *(float4*)((float*)t.p + off) = make_float4(a[0], a[1], a[2], a[3]);
        *(float4*)((float*)t.p + off + 4) = make_float4(b[0], b[1], b[2], b[3]);
    } else {
        ep_u4_t x;
        x[0] = ep_pk(a[0], a[1]); x[1] = ep_pk(a[2], a[3]); x[2] = ep_pk(b[0], b[1]); x[3] = ep_pk(b[2], b[3]);
        *(ep_u4_t*)((half_t*)t.p + off) = x;
    }
}

// Channel pairing (EP_PAIR).  An MFMA output fragment gives a lane 4 consecutive rows (l4 * 4 + r), i.e. 4 consecutive output
// channels: 8-byte fp16 stores, 16 separate 32-byte pieces per store instruction.  Which packed weight row a lane feeds into row i
// of a fragment is free, so kernels with an even number of output fragments per wave permute the rows of each fragment PAIR:
//   EP_PAIR 1 (one output per fragment, WCH even):      fragment 2q + h, row 4a + j  <-  row 32q + 8a + 4h + j of the wave's rows
//   EP_PAIR 2 (T blend / SPADE: fragments come as (plain, modulated) / (gamma, beta) twins over 16-row blocks, WCH % 4 == 0):
//              fragment 4q + 2h + t, row 4a + j  <-  twin t of output channel 32q + 8a + 4h + j of the wave's output channels
// A lane then owns 8 consecutive output channels per pair: ONE 16-byte fp16 store / residual load (two adjacent 16-byte ones for
// fp32) - half the store instructions, which is what the epilogue's store tail is bound by (MI355X_MICROARCH: store-issue-bound;
// profiles/r02_store_ablation.txt).  The weight fetch of a fragment still covers 8 whole 128-byte lines.
// ep_frag_row / ep_lane_row: the conv kernel's weight addressing; ep_chan: first of the 4 channels of (fragment ci, row group l4).
__device__ __forceinline__ constexpr int ep_frag_row(int pair, int ci)
{
    return pair == 1 ? (ci >> 1) * 32 + (ci & 1) * 4 : pair == 2 ? (ci >> 2) * 64 + ((ci >> 1) & 1) * 4 + (ci & 1) * 16 : ci * 16;
}
__device__ __forceinline__ constexpr int ep_lane_row(int pair, int i)
{
    const int c = (i >> 2) * 8 + (i & 3);
    return pair == 1 ? c : pair == 2 ? (c >> 4) * 32 + (c & 15) : i;
}
__device__ __forceinline__ constexpr int ep_chan(int pair, int cstep, int wave_row0, int ci, int l4)
{
    if (pair == 1) return wave_row0 + (ci >> 1) * 32 + l4 * 8 + (ci & 1) * 4;
    if (pair == 2) return wave_row0 / 2 + (ci >> 2) * 32 + l4 * 8 + ((ci >> 1) & 1) * 4;
    const int pb = wave_row0 / 16 + ci;
    return (cstep == 2 ? (pb >> 1) : pb) * 16 + l4 * 4;
}
__device__ __forceinline__ constexpr bool ep_second(int pair, int ci) { return pair == 1 ? (ci & 1) != 0 : pair == 2 ? ((ci >> 1) & 1) != 0 : false; }
// 8 fp16 channels at a multiple-of-8 channel offset are 16-byte aligned in this view
__device__ __forceinline__ bool ep_al8(const TDesc& t)
{
    return (((unsigned long long)t.p & 15ull) == 0) && (((t.sN | t.sD | t.sH | t.sW) & 7) == 0);
}
// EP_PAIR of a kernel instantiation
__device__ __forceinline__ constexpr int ep_pair_of(int mode, int wch)
{
    return (mode == MODE_PIXSHUF) ? 0 : (mode == MODE_TBLEND || mode == MODE_SPADE) ? (wch % 4 == 0 ? 2 : 0) : (wch % 2 == 0 ? 1 : 0);
}

// EP_HEAVY (constexpr bool, in scope): this instantiation also carries sigmoid / GELU for act0 (launchers refuse those activations
// on the others); act1 is always one of none / ReLU / LeakyReLU.
// Expects in scope: p, ep_acc[WCH][EP_WPX] (f4_t), ep_wpx (position-block index of this wave), EP_WPX, n0, tw, th, td, tn,
// l15p (the position 0..15 of its blocks this lane works on: l15, or the conv_halo kernels' bank-conflict permutation of it),
// lgTW, lgTH, lgTD, lgS, mW, mH, mD (the tile decomposition: compile-time constants in the static-shape kernels, which turns the
// per-block coordinates below into constants), wch, l15, l4, tile_lin (linear index of the position tile) and the template constants
// WCH, BM, MODE, EP_EARLY (+ ep_xpre from CONV_EPILOGUE_EARLY_FETCH; false elsewhere), EP_PAIR (ep_pair_of(MODE, WCH) where the kernel permutes its weight rows accordingly, else 0).
#if defined(CS_EP_NOSTORE)
#define EP_STORE_COND && (p.N < 0)
#else
#define EP_STORE_COND
#endif
// SPADE kernels fetch the tensor they modulate (x, fp16) for the wave's WHOLE tile at kernel start, where the round trip hides behind
// the first halo wait, instead of after the main loop where every wave of the CU would sit through it at the same time (the
// workgroups of a CU run in lock step; profiles/r02_timeline_*.txt: "epi first block").  16 more live registers in the main loop:
// only where the kernel has them to spare (EP_EARLY, set by the kernel).  Same addresses / validity rules as CONV_EPILOGUE's fetch
// round.  Expects the kernel's tile decomposition in scope (see CONV_EPILOGUE) and EP_WPX0 = position blocks per wave, ep_wpx0 = the
// wave's block index; defines ep_xpre.
#define CONV_EPILOGUE_EARLY_FETCH() \
    ep_u2_t ep_xpre[EP_EARLY ? EP_WPX0 : 1][EP_EARLY ? (WCH + 1) / 2 : 1]; \
    if constexpr (EP_EARLY) { \
        int xlw, xlh, xld, xln; \
        { int t = l15p; xlw = t & mW; t >>= lgTW; xlh = t & mH; t >>= lgTH; xld = t & mD; t >>= lgTD; xln = t; } \
        const int xrs = p.res_shift, xnb = tn * (BM >> lgS); \
        const unsigned xlane = (unsigned)((xnb + xln) * (int)p.res.sN + __mul24((td << lgTD) + xld, (int)p.res.sD) + \
                                          __mul24(((th << lgTH) + xlh) >> xrs, (int)p.res.sH) + __mul24(((tw << lgTW) + xlw) >> xrs, (int)p.res.sW)); \
_Pragma("unroll") \
        for (int pi = 0; pi < EP_WPX0; ++pi) { \
            int bw, bh, bd, bn; \
            { int t = (ep_wpx0 * EP_WPX0 + pi) << 4; bw = t & mW; t >>= lgTW; bh = t & mH; t >>= lgTH; bd = t & mD; t >>= lgTD; bn = t; } \
            const unsigned xb = xlane + (unsigned)(bn * (int)p.res.sN + bd * (int)p.res.sD + (bh >> xrs) * (int)p.res.sH + (bw >> xrs) * (int)p.res.sW); \
_Pragma("unroll") \
            for (int ci = 0; ci < WCH; ci += 2) { \
                const int cb = ep_chan(EP_PAIR, 2, n0 + wch * WCH * 16, ci, l4); \
                ep_u2_t q2; q2[0] = 0u; q2[1] = 0u; \
                if (p.res.p && xnb + xln + bn < p.N && cb < p.Cout) q2 = *(const ep_u2_t*)((const half_t*)p.res.p + (xb + (unsigned)cb)); \
                ep_xpre[pi][ci / 2] = q2; \
            } \
        } \
    }

// value of lane i + N of the lane's row of 16 (row_shl:N; 0 beyond the row)
template <int N> __device__ __forceinline__ float ep_row_shl(float a)
{
#ifdef EP_HOST_EMULATION          /* the host harness runs one lane at a time and never takes the pooled path */
    (void)a;
    return 0.f;
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x100 | N, 0xf, 0xf, true));
#endif
}

// Sum over the 16 lanes of a row (the 16 positions of an MFMA block), valid in the row's lane 0, with the bits of the xor butterfly
// a += shfl_xor(a, 1), 2, 4, 8 there: lane 0 of that butterfly only ever combines values from lanes i and i + o (i a multiple of 2 o), which
// is what a row shift by o delivers.  As DPP operands of the adds these are 4 VALU instructions; the shuffles were 4 ds_bpermute round
// trips per value (the statistics epilogue of a 128 x 128 wave tile: 27 000 of 34 000 cycles, profiles/r04_d_wide_probe.txt).
__device__ __forceinline__ float ep_row_sum16(float a)
{
#ifdef EP_HOST_EMULATION
    for (int o = 1; o < 16; o <<= 1) a += __shfl_xor(a, o, 64);
    return a;
#else
    // row_shl:n (dpp_ctrl 0x100 + n): lane i reads lane i + n of its row of 16; lanes that would read beyond the row get 0 (bound_ctrl)
    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x101, 0xf, 0xf, true));
    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x102, 0xf, 0xf, true));
    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x104, 0xf, 0xf, true));
    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x108, 0xf, 0xf, true));
    return a;
#endif
}

// partial statistics of segment sg (EP_SG position blocks of this wave): fixed-order butterfly over the 16 position lanes, one
// partial per (tile, segment, channel); executed by all lanes
#define EP_STAT_FLUSH(sg) \
    if (EP_STAT) { \
        const int ep_tiles = p.nTW * p.nTH * p.nTD; \
        const int ep_nblk = EP_NBLK_V; \
        const int ep_blk = EP_BLK_V(sg); \
_Pragma("unroll") \
        for (int ci = 0; ci < WCH; ci += CSTEP) { \
            const int cb = ep_chan(EP_PAIR, CSTEP, n0 + wch * WCH * 16, ci, l4); \
            float ep_pa[4], ep_pb[4]; \
_Pragma("unroll") \
            for (int r = 0; r < 4; ++r) { \
                ep_pa[r] = ep_row_sum16(ep_sum[EP_STAT ? ci : 0][r]); ep_pb[r] = ep_row_sum16(ep_sq[EP_STAT ? ci : 0][r]); \
                ep_sum[EP_STAT ? ci : 0][r] = 0.f; ep_sq[EP_STAT ? ci : 0][r] = 0.f; \
            } \
            /* the (sum, sum of squares) pairs of the lane's 4 channels are 32 contiguous bytes: two 16-byte stores instead of eight 4-byte ones \
               (p.Cout is a multiple of 4 and cb of 4: all four channels exist or none; the launchers refuse anything else) */ \
            if (l15 == 0 && cb < p.Cout) { \
                float4* dst = (float4*)(p.stat_out + (((long)tn * ep_nblk + ep_blk) * p.Cout + cb) * 2); \
                dst[0] = make_float4(ep_pa[0], ep_pb[0], ep_pa[1], ep_pb[1]); \
                dst[1] = make_float4(ep_pa[2], ep_pb[2], ep_pa[3], ep_pb[3]); \
            } \
        } \
    }

#ifndef EP_GSEL_V
#define EP_GSEL_V 0       /* a kernel may fix the position blocks fetched per round itself (conv_wide.hip: a constexpr in its scope) */
#endif
#ifndef EP_BLK_V          /* partial-statistics block of (this wave, segment sg): kernels whose tile is not the one the block order was defined on override it */
#define EP_BLK_V(sg) ((tile_lin % ep_tiles) * (BM / (EP_SG * 16)) + ep_wpx * (EP_WPX / EP_SG) + (sg))
#define EP_NBLK_V (ep_tiles * (BM / (EP_SG * 16)))
#endif
#ifndef EP_SLICE_FENCE    /* executed after every position block of the epilogue (conv_wide.hip: a scheduling fence, so that the 256 accumulators \
                             of its waves are read block by block instead of all at once) */
#define EP_SLICE_FENCE
#endif
#ifndef EP_FG
#define EP_FG 8           /* position blocks fetched per round in the fast paths of the 128x256 kernels (the general path: 2 / 1) */
#endif
#ifndef EP_FG_F32
#define EP_FG_F32 4         /* ... with an fp32 residual (16 registers per block) */
#endif
#ifndef EP_FG_STAT
#define EP_FG_STAT 4
#endif
#ifndef EP_PIPE_V
#define EP_PIPE_V 0
#endif
#ifndef EP_O0_EXTRA_V
#define EP_O0_EXTRA_V 0u      /* element offset added to every out0 address (conv_halo: the output phase of a grouped launch, ConvParams::nphase) */
#endif
/* AvgPool(1,2,2) of out0 inside the epilogue (ConvParams::pool_hw; DownBlock3d, util.py:185-190; DownBlock2d, util.py:150-165): the four
   positions of a window are four lanes of one 16-position block; a kernel that supports it names the two DPP row shifts (in lanes) that reach
   the w + 1 and h + 1 neighbours under its lane -> position map.  WSH > 0 with HSH == 0: the blocks are rows of 16 columns (2-D 16 x 8
   tiles) and the h + 1 neighbour is the same lane of the next block.  WSH == 0: the kernel has no pooling epilogue. */
#ifndef EP_POOL_WSH_V
#define EP_POOL_WSH_V 0
#define EP_POOL_HSH_V 0
#endif
#define EP_POOLB 128          /* EP_CODE bit: pooled out0 */
/* ConvParams::spmul (mode STD): out0 = act0(IN(res) * (1 + conv + bias)) with (mean, rstd) from stats - SPADE's modulation without its beta
   half (util.py:295-302 where beta is applied elsewhere: the learned shortcut of SPADEResnetBlock, engine.hip run_G).  The general epilogue
   always knows it; a kernel that wants a branch-free copy sets EP_SPMUL_V. */
#ifndef EP_SPMUL_V
#define EP_SPMUL_V 0
#endif
/* a kernel whose launches use the form "fp16 residual, no out0, second output only" (out1 = act1(conv + bias + res): the last conv of G's
   up_1 block, whose only consumer is conv_img behind leaky_relu) and wants a branch-free copy of it sets EP_O1ONLY_V. */
#ifndef EP_O1ONLY_V
#define EP_O1ONLY_V 0
#endif
/* a kernel that runs the motion extractor's linear layers (1x1 convs in split precision, convnextv2.py:39-45: fp32 out0 with GELU; fp32 out0 +
   fp32 residual in place; fp32 out0) and wants branch-free copies of those three forms - and of F.second's (fp32 out0 + fp16 out1) - sets EP_MLIN_V. */
#ifndef EP_MLIN_V
#define EP_MLIN_V 0
#endif
/* a kernel whose launches write the same fp16 values twice (out0 and, through the identity second affine, out1: the duplicate output rows of
   mlp_shared on a x4 up-sampled map, engine.hip run_G) and wants a branch-free copy of that form sets EP_DUP_V. */
#ifndef EP_DUP_V
#define EP_DUP_V 0
#endif
#define EP_SPMULB 256         /* EP_CODE bit: spmul */
#define EP_GELUB 512          /* EP_CODE bit: act0 is GELU (kernels with EP_HEAVY: the motion extractor's pwconv1) */
/* one fetch round of the epilogue: residual / modulated tensor / per-position scale of the position blocks PG0 .. PG0 + EP_G - 1 -> register set BI */
#define EP_FETCH_ROUND(PG0, BI) \
    if (EP_PF && !EP_EARLY && (ep_fetch || ep_has_ps)) { \
_Pragma("unroll") \
        for (int g = 0; g < EP_G; ++g) { \
            int bw, bh, bd, bn; \
            { int t = (ep_wpx * EP_WPX + (PG0) + g) << 4; bw = t & mW; t >>= lgTW; bh = t & mH; t >>= lgTH; bd = t & mD; t >>= lgTD; bn = t; } \
            if (!EPFAST && ep_nb + ep_ln + bn >= p.N) continue; \
            if (ep_has_ps) ep_ps2[BI][g] = p.pixscale[ep_lane_ps + (unsigned)((((bn * p.D + bd) * p.H + bh) * p.W + bw) * p.ps_stride)]; \
            if (ep_fetch) { \
                const unsigned xb = ep_lane_res + (unsigned)(bn * (int)p.res.sN + bd * (int)p.res.sD + (bh >> ep_rs) * (int)p.res.sH + \
                                                            (bw >> ep_rs) * (int)p.res.sW); \
_Pragma("unroll") \
                for (int ci = 0; ci < WCH; ci += CSTEP) { \
                    const int cb = ep_chan(EP_PAIR, CSTEP, n0 + wch * WCH * 16, ci, l4); \
                    if (!EPALL && cb >= p.Cout) continue; \
                    if (ep_res32) ep_raw2[BI][g][(EP_PF ? ci / CSTEP : 0)] = *(const ep_u4_t*)((const float*)p.res.p + (xb + (unsigned)cb)); \
                    else if (ep_mr && ep_second(EP_PAIR, ci)) { /* came with the pair's first half */ } \
                    else if (ep_mr && (EPALL || cb + 4 < p.Cout)) ep_raw2[BI][g][(EP_PF ? ci / CSTEP : 0)] = *(const ep_u4_t*)((const half_t*)p.res.p + (xb + (unsigned)cb)); \
                    else { \
                        const ep_u2_t q2 = *(const ep_u2_t*)((const half_t*)p.res.p + (xb + (unsigned)cb)); \
                        ep_raw2[BI][g][(EP_PF ? ci / CSTEP : 0)][0] = q2[0]; ep_raw2[BI][g][(EP_PF ? ci / CSTEP : 0)][1] = q2[1]; \
                    } \
                } \
            } \
        } \
    }
#define CONV_EPILOGUE_IMPL(EPCODE) \
    constexpr int EPF = (EPCODE); \
    constexpr bool EPFAST = EPF >= 0; \
    constexpr bool EPALL = EPFAST && ((EPF >> 6) & 1) == 0;      /* every channel of the wave exists */ \
    constexpr bool EP_POOL = EPFAST && ((EPF >> 7) & 1) != 0 && (EP_POOL_WSH_V) > 0;      /* out0 = AvgPool(1,2,2) of the activated values, on the pooled grid */ \
    constexpr int EP_PS = EP_POOL ? 1 : 0; \
    constexpr bool EP_GELU = EPFAST && ((EPF >> 9) & 1) != 0;          /* act0 = GELU (EP_GELUB) */ \
    const bool ep_spm = EPFAST ? (((EPF >> 8) & 1) != 0) : ((MODE == MODE_STD) && p.spmul != 0); \
    constexpr bool EP_POOL_HB = EP_POOL && (EP_POOL_HSH_V) == 0;      /* 2-D tiles: the h + 1 neighbour is the same lane of the NEXT position block */ \
    float ep_phold[EP_POOL_HB ? WCH : 1][4];                           /* activated values of the even block, until the odd one arrives */ \
    const bool ep_has_res = EPFAST ? ((EPF & 3) != 0) : (p.res.p != nullptr); \
    const bool ep_res32 = EPFAST ? ((EPF & 3) == 2) : (p.res_f32 != 0); \
    const bool ep_has_o0 = EPFAST ? (((EPF >> 2) & 1) != 0) : (p.out0.p != nullptr); \
    const bool ep_o032 = EPFAST ? (((EPF >> 3) & 1) != 0) : (p.out0_f32 != 0); \
    const bool ep_has_o1 = EPFAST ? (((EPF >> 4) & 1) != 0) : (p.out1.p != nullptr); \
    const bool ep_has_ps = EPFAST ? (((EPF >> 5) & 1) != 0) : (p.pixscale != nullptr); \
    constexpr int CSTEP = (MODE == MODE_TBLEND || MODE == MODE_SPADE) ? 2 : 1; \
    /* per-channel constants of this lane's 4 channels, loaded once (16-byte loads), not once per position block */ \
    float4 ep_bias[WCH], ep_bias2[WCH], ep_s2[WCH], ep_t2[WCH], ep_mean[WCH], ep_rstd[WCH]; \
_Pragma("unroll") \
    for (int ci = 0; ci < WCH; ci += CSTEP) { \
        const int cb = ep_chan(EP_PAIR, CSTEP, n0 + wch * WCH * 16, ci, l4); \
        const bool cok = EPALL || cb < p.Cout; \
        ep_bias[ci] = (cok && p.bias) ? *(const float4*)(p.bias + cb) : make_float4(0.f, 0.f, 0.f, 0.f); \
        ep_bias2[ci] = (cok && MODE == MODE_SPADE) ? *(const float4*)(p.bias2 + cb) : make_float4(0.f, 0.f, 0.f, 0.f); \
        ep_s2[ci] = (cok && p.s2) ? *(const float4*)(p.s2 + cb) : make_float4(1.f, 1.f, 1.f, 1.f); \
        ep_t2[ci] = (cok && p.s2) ? *(const float4*)(p.t2 + cb) : make_float4(0.f, 0.f, 0.f, 0.f); \
        if ((MODE == MODE_SPADE || ep_spm) && cok) { /* SPADE launches tile within one sample: n == tn */ \
            const float4 q0 = *(const float4*)(p.stats + ((long)tn * p.Cout + cb) * 2); \
            const float4 q1 = *(const float4*)(p.stats + ((long)tn * p.Cout + cb) * 2 + 4); \
            ep_mean[ci] = make_float4(q0.x, q0.z, q1.x, q1.z); ep_rstd[ci] = make_float4(q0.y, q0.w, q1.y, q1.w); \
        } else { ep_mean[ci] = make_float4(0.f, 0.f, 0.f, 0.f); ep_rstd[ci] = make_float4(1.f, 1.f, 1.f, 1.f); } \
    } \
    const float ep_sl0 = lin_slope(p.act0, p.slope0), ep_sl1 = lin_slope(p.act1, p.slope1); \
    constexpr bool EP_STAT = (MODE == MODE_STDSTAT); /* compile-time: the accumulators cost occupancy otherwise */ \
    constexpr int EP_SC = EP_STAT ? WCH : 1; \
    /* statistics are emitted per 64 positions (4 blocks) also by the waves that own 128: the partial sums - and with them every \
       bit downstream - do not depend on whether a layer ran as 128x128 or 128x64 tiles (the engine narrows the tiles of launches \
       that would leave CUs idle) */ \
    constexpr int EP_SG = EP_WPX > 4 ? 4 : EP_WPX; \
    float ep_sum[EP_SC][4], ep_sq[EP_SC][4]; \
_Pragma("unroll") \
    for (int ci = 0; ci < EP_SC; ++ci) \
_Pragma("unroll") \
        for (int r = 0; r < 4; ++r) { ep_sum[ci][r] = 0.f; ep_sq[ci][r] = 0.f; } \
    /* Every fetch of the epilogue (residual / the tensor being modulated / per-position scale) for the WHOLE tile of the wave is \
       issued before its first store.  vmcnt counts loads and stores alike and hipcc may not hoist a load above a store that might \
       alias, so a load inside the per-block loop waits for its own round trip AND for the acknowledgement of every store issued \
       before it: 8-16 serialised memory latencies per wave, 13-38 % of a wave's lifetime (profiles/r02_timeline_before.txt). \
       The main loop's operand registers are dead here, so the staging registers are free.  In-place use (res == out0) stays \
       correct: a lane reads exactly the elements it later writes.  Not compiled into the 160-wide tiles (WCH == 5: they never \
       carry a residual and 80 more live registers would cost them an occupancy step). */ \
    constexpr bool EP_PF = (WCH != 5); \
    /* position blocks fetched per round: the whole tile where the register budget allows (no scratch, same occupancy step - \
       checked with tools/kernel_resources.py), else groups of 4 (128x128 tiles held to 168 registers), 2 (128x256 non-blend) or 1 (128x256 with 32 statistics accumulators) */ \
    constexpr int EP_G = (EP_GSEL_V) > 0 ? (EP_GSEL_V) : !EP_PF ? 1 : (EP_WPX < 8 ? EP_WPX : (MODE == MODE_TBLEND ? 8 : (WCH == 4 ? (EPFAST ? (EP_STAT ? EP_FG_STAT : ((EPF & 3) == 2 ? EP_FG_F32 : EP_FG)) : (EP_STAT ? 1 : 2)) : 4))); \
    constexpr int EP_NCI = EP_PF ? (WCH + CSTEP - 1) / CSTEP : 1; \
    const bool ep_fetch = EP_PF && ep_has_res; \
    /* channel pairs (EP_PAIR): fp16 tensors whose pointer and strides keep 8 channels 16-byte aligned get one access per pair */ \
    const bool ep_m0 = EP_PAIR != 0 && !ep_o032 && (EPFAST || ep_al8(p.out0)); \
    const bool ep_m1 = EP_PAIR != 0 && (EPFAST || ep_al8(p.out1)); \
    const bool ep_mr = EP_PAIR != 0 && EP_PF && !ep_res32 && (EPFAST || ep_al8(p.res)); \
    const int ep_rshift = (MODE == MODE_SPADE || ep_spm) ? p.res_shift : 0; \
    /* Addressing.  A position of the tile is m = blk * 16 + l15 with blk = ep_wpx * EP_WPX + pi uniform over the wave; the tile's \
       (w, h, d, n) are disjoint bit fields of m, so every coordinate - also after the >> of an up-sampled operand - is the sum of a \
       lane part (bits of l15) and a block part (bits of blk), and every element offset (a linear form in the coordinates) is \
       lane offset + block offset: the lane offsets are computed once per wave (VALU), the block offsets are scalar arithmetic. \
       Offsets are 32-bit (launchers refuse tensors of 2^31 elements) and unsigned, which lets the loads / stores use the \
       SGPR-base + VGPR-offset form. */ \
    int ep_lw, ep_lh, ep_ld, ep_ln; \
    { int t = l15p; ep_lw = t & mW; t >>= lgTW; ep_lh = t & mH; t >>= lgTH; ep_ld = t & mD; t >>= lgTD; ep_ln = t; } \
    const int ep_w0 = tw << lgTW, ep_h0 = th << lgTH, ep_d0 = td << lgTD, ep_nb = tn * (BM >> lgS); \
    const int ep_rs = (MODE == MODE_SPADE || ep_spm) ? p.res_shift : 0; \
    /* per-axis products as 24-bit multiplies (full rate; coordinates are small, launchers refuse axis strides >= 2^23); the \
       sample term only where a tile spans several samples */ \
    const bool ep_tn1 = (BM >> lgS) == 1; \
    const unsigned ep_lane_res = (unsigned)(ep_nb * (int)p.res.sN + (ep_tn1 ? 0 : ep_ln * (int)p.res.sN) + __mul24(ep_d0 + ep_ld, (int)p.res.sD) + \
                                            __mul24((ep_h0 + ep_lh) >> ep_rs, (int)p.res.sH) + __mul24((ep_w0 + ep_lw) >> ep_rs, (int)p.res.sW)); \
    const unsigned ep_lane_o0 = (EP_O0_EXTRA_V) + (unsigned)(ep_nb * (int)p.out0.sN + (ep_tn1 ? 0 : ep_ln * (int)p.out0.sN) + __mul24(ep_d0 + ep_ld, (int)p.out0.sD) + \
                                           __mul24((ep_h0 + ep_lh) >> EP_PS, (int)p.out0.sH) + __mul24((ep_w0 + ep_lw) >> EP_PS, (int)p.out0.sW)); \
    /* pooled: the lanes whose position is the (even h, even w) corner of a window hold the window's sum and store it */ \
    const bool ep_pool_lane = !EP_POOL || (lane & ((EP_POOL_WSH_V) | (EP_POOL_HSH_V))) == 0; \
    const unsigned ep_lane_o1 = (EP_O0_EXTRA_V) + (unsigned)(ep_nb * (int)p.out1.sN + (ep_tn1 ? 0 : ep_ln * (int)p.out1.sN) + __mul24(ep_d0 + ep_ld, (int)p.out1.sD) + \
                                           __mul24(ep_h0 + ep_lh, (int)p.out1.sH) + __mul24(ep_w0 + ep_lw, (int)p.out1.sW)); \
    const unsigned ep_lane_ps = (unsigned)(((((ep_nb + ep_ln) * p.D + ep_d0 + ep_ld) * p.H + ep_h0 + ep_lh) * p.W + ep_w0 + ep_lw) * p.ps_stride); \
    /* pixel shuffle: out[n][c][2h + i][2w + j], H2 = 2H, W2 = 2W */ \
    const unsigned ep_lane_px = (unsigned)((((ep_nb + ep_ln) * 3) * 2 * p.H + 2 * (ep_h0 + ep_lh)) * 2 * p.W + 2 * (ep_w0 + ep_lw)); \
    /* EP_PIPE_V (a kernel's choice; 0 elsewhere): the fetch round of group k + 1 is issued BEFORE the stores of group k, into a second set of \
       registers.  vmcnt retires in order, so the wait for group k + 1's operands then leaves group k's stores in flight; fetched after them \
       (the plain order) every round waits for the acknowledgement of the previous round's stores plus its own round trip - 4 rounds of \
       the fp32-residual forms of conv_wide: 54 000 cycles per tile instead of 8 000 (profiles/r04_d_wide_probe.txt). */ \
    constexpr bool EP_PIPE = (EP_PIPE_V) != 0 && EP_G < EP_WPX; \
    ep_u4_t ep_raw2[EP_PIPE ? 2 : 1][EP_G][EP_NCI]; float ep_ps2[EP_PIPE ? 2 : 1][EP_G]; \
    if (EP_PIPE) { EP_FETCH_ROUND(0, 0) } \
_Pragma("unroll") \
    for (int pg = 0; pg < EP_WPX; pg += EP_G) { \
    const int ep_bi = EP_PIPE ? ((pg / EP_G) & 1) : 0; \
    if (!EP_PIPE) { EP_FETCH_ROUND(pg, 0) } \
    else if (pg + EP_G < EP_WPX) { EP_FETCH_ROUND(pg + EP_G, EP_PIPE ? (((pg / EP_G) + 1) & 1) : 0) } \
    ep_u4_t (&ep_raw)[EP_G][EP_NCI] = ep_raw2[ep_bi]; float (&ep_ps)[EP_G] = ep_ps2[ep_bi]; \
    if (pg == 0) EP_TL(6); \
_Pragma("unroll") \
    for (int g = 0; g < EP_G; ++g) { \
        const int pi = pg + g; \
        if (pi == 1) EP_TL(7); \
        if (pi > 0 && pi % EP_SG == 0) { EP_STAT_FLUSH(pi / EP_SG - 1) } \
        int bw, bh, bd, bn; \
        { int t = (ep_wpx * EP_WPX + pi) << 4; bw = t & mW; t >>= lgTW; bh = t & mH; t >>= lgTH; bd = t & mD; t >>= lgTD; bn = t; } \
        if (!EPFAST && ep_nb + ep_ln + bn >= p.N) continue; \
        float ps = 1.f; \
        if (ep_has_ps) ps = EP_PF ? ep_ps[g] : p.pixscale[ep_lane_ps + (unsigned)((((bn * p.D + bd) * p.H + bh) * p.W + bw) * p.ps_stride)]; \
        const unsigned xb = ep_lane_res + (unsigned)(bn * (int)p.res.sN + bd * (int)p.res.sD + (bh >> ep_rs) * (int)p.res.sH + (bw >> ep_rs) * (int)p.res.sW); \
        const unsigned ob0 = ep_lane_o0 + (unsigned)(bn * (int)p.out0.sN + bd * (int)p.out0.sD + (bh >> EP_PS) * (int)p.out0.sH + (bw >> EP_PS) * (int)p.out0.sW); \
        const unsigned ob1 = ep_lane_o1 + (unsigned)(bn * (int)p.out1.sN + bd * (int)p.out1.sD + bh * (int)p.out1.sH + bw * (int)p.out1.sW); \
        float ep_vh[4] = {0.f, 0.f, 0.f, 0.f}, ep_uh[4] = {0.f, 0.f, 0.f, 0.f};      /* first half of a channel pair, held for the joint store */ \
_Pragma("unroll") \
        for (int ci = 0; ci < WCH; ci += CSTEP) { \
            const int cb = ep_chan(EP_PAIR, CSTEP, n0 + wch * WCH * 16, ci, l4); \
            if (!EPALL && cb >= p.Cout) continue; \
            float rr[4] = {0.f, 0.f, 0.f, 0.f};     /* residual (STD / TBLEND) or the modulated tensor x (SPADE) */ \
            if (ep_has_res) { \
                if (EP_EARLY) { \
                    const h4_t hx = __builtin_bit_cast(h4_t, ep_xpre[EP_EARLY ? pi : 0][EP_EARLY ? ci / 2 : 0]); \
_Pragma("unroll") \
                    for (int r = 0; r < 4; ++r) rr[r] = (float)hx[r]; \
                } else if (EP_PF) { \
                    const bool ep_hi = ep_mr && ep_second(EP_PAIR, ci);      /* upper half of the pair's 16-byte fetch */ \
                    const ep_u4_t q4 = ep_raw[g][(EP_PF ? ci / CSTEP : 0)]; \
                    const ep_u4_t qp = ep_raw[g][(EP_PF && ep_second(EP_PAIR, ci) ? ci / CSTEP - 1 : 0)]; \
                    if (ep_res32) { \
                        const f4_t qf = __builtin_bit_cast(f4_t, q4);     /* whole-vector cast: bit_cast of q4[r] reads element 0 */ \
_Pragma("unroll") \
                        for (int r = 0; r < 4; ++r) rr[r] = qf[r]; \
                    } else { \
                        ep_u2_t q2; q2[0] = ep_hi ? qp[2] : q4[0]; q2[1] = ep_hi ? qp[3] : q4[1]; \
                        const h4_t hx = __builtin_bit_cast(h4_t, q2); \
_Pragma("unroll") \
                        for (int r = 0; r < 4; ++r) rr[r] = (float)hx[r]; \
                    } \
                } else { \
                    load4(p.res, ep_res32, (long)(xb + (unsigned)cb), rr); \
                } \
            } \
            float v[4]; \
            if (MODE == MODE_TBLEND) { \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) \
                    v[r] = ps * (ep_acc[ci + CSTEP - 1][pi][r] + ((const float*)&ep_bias[ci])[r]) + (1.f - ps) * ep_acc[ci][pi][r]; \
            } else if (MODE == MODE_SPADE) { \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) { \
                    const float gm = ep_acc[ci][pi][r] + ((const float*)&ep_bias[ci])[r]; \
                    const float bt = ep_acc[ci + CSTEP - 1][pi][r] + ((const float*)&ep_bias2[ci])[r]; \
                    v[r] = fmaf((rr[r] - ((const float*)&ep_mean[ci])[r]) * ((const float*)&ep_rstd[ci])[r], 1.f + gm, bt); \
                } \
            } else { \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) v[r] = ep_acc[ci][pi][r] + ((const float*)&ep_bias[ci])[r]; \
                if (ep_spm) { \
_Pragma("unroll") \
                    for (int r = 0; r < 4; ++r) v[r] = ((rr[r] - ((const float*)&ep_mean[ci])[r]) * ((const float*)&ep_rstd[ci])[r]) * (1.f + v[r]); \
                } \
            } \
_Pragma("unroll") \
            /* branch-free copies know their activation: the linear family as one formula, or GELU (EP_GELUB); only the general epilogue of a \
               kernel that carries sigmoid / GELU (EP_HEAVY) switches per element */ \
            for (int r = 0; r < 4; ++r) v[r] = EP_GELU ? gelu_act(v[r]) : ((EP_HEAVY && !EPFAST) ? apply_act(v[r], p.act0, p.slope0) : lin_act(v[r], ep_sl0)); \
            if (MODE == MODE_PIXSHUF) { \
                const int c = cb >> 2; \
                if (c < 3) { \
                    float* o = (float*)p.out0.p; \
                    const unsigned W2 = 2u * (unsigned)p.W, H2 = 2u * (unsigned)p.H; \
                    const unsigned base = ep_lane_px + ((unsigned)(bn * 3 + c) * H2 + 2u * (unsigned)bh) * W2 + 2u * (unsigned)bw; \
                    *(float2*)(o + base) = make_float2(v[0], v[1]); \
                    *(float2*)(o + base + W2) = make_float2(v[2], v[3]); \
                } \
                continue; \
            } \
            /* no residual: rr == 0; no per-position scale: ps == 1 - applied unconditionally (a select per element costs more issue \
               slots than the add / multiply it would skip; the epilogue is VALU-issue-bound, profiles/r02_store_ablation.txt) */ \
            if (MODE != MODE_SPADE && !ep_spm) { \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) v[r] += rr[r]; \
            } \
            if (MODE == MODE_STD || MODE == MODE_STDSTAT) { \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) v[r] *= ps; \
            } \
            if (EP_POOL_HB) { /* (a + c) + (b + d): the block pair in registers, the column pair as a DPP operand; x 0.25 */ \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) { \
                    if ((pi & 1) == 0) ep_phold[EP_POOL_HB ? ci : 0][r] = v[r]; \
                    else { \
                        float t_ = ep_phold[EP_POOL_HB ? ci : 0][r] + v[r]; \
                        t_ += ep_row_shl<(EP_POOL ? (EP_POOL_WSH_V) : 1)>(t_); \
                        v[r] = t_ * 0.25f; \
                    } \
                } \
            } else if (EP_POOL) { /* (a + b) + (c + d) over the window, as DPP operands of the adds; x 0.25 */ \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) { \
                    float t_ = v[r]; \
                    t_ += ep_row_shl<(EP_POOL ? (EP_POOL_WSH_V) : 1)>(t_); \
                    t_ += ep_row_shl<(EP_POOL ? (EP_POOL_HSH_V) : 1)>(t_); \
                    v[r] = t_ * 0.25f; \
                } \
            } \
            const bool ep_hold = EP_PAIR != 0 && !ep_second(EP_PAIR, ci) && (EPALL || cb + 4 < p.Cout);      /* first half of a complete pair */ \
            if (ep_has_o0 && ep_pool_lane && (!EP_POOL_HB || (pi & 1)) EP_STORE_COND) { \
                if (ep_m0 && ep_hold) { \
_Pragma("unroll") \
                    for (int r = 0; r < 4; ++r) ep_vh[r] = v[r]; \
                } else if (ep_m0 && ep_second(EP_PAIR, ci)) store8(p.out0, 0, (long)(ob0 + (unsigned)(cb - 4)), ep_vh, v); \
                else store4(p.out0, ep_o032, (long)(ob0 + (unsigned)cb), v); \
            } \
            if (EP_STAT) { /* statistics of the values as stored (fp16-rounded when out0 is fp16) */ \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) { \
                    const float vs = ep_o032 ? v[r] : (float)ep_h(v[r]); \
                    ep_sum[EP_STAT ? ci : 0][r] += vs; ep_sq[EP_STAT ? ci : 0][r] = fmaf(vs, vs, ep_sq[EP_STAT ? ci : 0][r]); \
                } \
            } \
            if (ep_has_o1) { \
                float u[4]; \
_Pragma("unroll") \
                for (int r = 0; r < 4; ++r) { \
                    const float a = fmaf(v[r], ((const float*)&ep_s2[ci])[r], ((const float*)&ep_t2[ci])[r]);   /* explicit: every copy of the epilogue fuses alike */ \
                    u[r] = lin_act(a, ep_sl1); \
                } \
                if (true EP_STORE_COND) { \
                    if (ep_m1 && ep_hold) { \
_Pragma("unroll") \
                        for (int r = 0; r < 4; ++r) ep_uh[r] = u[r]; \
                    } else if (ep_m1 && ep_second(EP_PAIR, ci)) store8(p.out1, 0, (long)(ob1 + (unsigned)(cb - 4)), ep_uh, u); \
                    else store4(p.out1, 0, (long)(ob1 + (unsigned)cb), u); \
                } \
            } \
        } \
        EP_SLICE_FENCE \
    } \
    } \
    EP_STAT_FLUSH((EP_WPX - 1) / EP_SG)

// Fast paths.  The general epilogue decides per element, at run time, whether a residual / second output / per-position scale exists, in
// which precision, whether a channel pair can go out as one 16-byte access and whether the lane's sample and channels exist.  hipcc turns
// that into ~300 branches with the fetches inside them and an `s_waitcnt vmcnt(0)` at every join (78-90 per wave in the 128x256 kernels of
// modes STD / SPADE / STDSTAT; the T blend kernel, whose tensors happen to take one path, has 2): every fetch waits for its own round trip
// and - vmcnt counts stores too - for the acknowledgement of every store before it.  The epilogue was 44 % of a SPADE workgroup's life
// (profiles/r03_o_timeline_spade.txt).  EP_FAST kernels therefore test ONCE per wave whether the launch is one of the combinations the
// engine's hot layers use (code: bits 1:0 residual none / fp16 / fp32, bit 2 out0, bit 3 out0 fp32, bit 4 out1, bit 5 per-position
// scale) with every channel of the wave and every sample of the tile valid and the fp16 tensors 16-byte aligned, and run a copy of the
// same epilogue with those facts as compile-time constants: no branches around fetches or stores, counted waits.  Same arithmetic, same
// bits.  Expects EP_FAST (constexpr bool) in scope.
#define EP_CODE(res, o0, o0f32, o1, ps) ((res) | ((o0) << 2) | ((o0f32) << 3) | ((o1) << 4) | ((ps) << 5))
#define EP_RAGGED 64          /* not every channel of the wave exists (kernels without channel pairs only: the 160-wide tiles) */
#define CONV_EPILOGUE() \
    { \
        int ep_code = -1; \
        if constexpr (EP_FAST || (EP_POOL_WSH_V) > 0) { \
            constexpr int CST = (MODE == MODE_TBLEND || MODE == MODE_SPADE) ? 2 : 1; \
            const int ep_r0 = n0 + wch * WCH * 16; \
            const int ep_chi = (ep_r0 + WCH * 16) / CST;              /* one past the wave's last output channel */ \
            const bool ep_call = ep_chi <= p.Cout && (p.Cout & 7) == 0; \
            const bool ep_ok = !p.ep_general && (ep_call || (EP_PAIR == 0 && (p.Cout & 3) == 0)) && (tn + 1) * (BM >> lgS) <= p.N && \
                               (!EP_HEAVY || p.act0 <= ACT_LRELU || p.act0 == ACT_GELU) &&      /* (the copies carry the linear family or GELU, not sigmoid) */ \
                               (!p.res.p || p.res_f32 || EP_PAIR == 0 || ep_al8(p.res)) && \
                               (!p.out0.p || p.out0_f32 || EP_PAIR == 0 || ep_al8(p.out0)) && (!p.out1.p || EP_PAIR == 0 || ep_al8(p.out1)); \
            if (ep_ok) ep_code = EP_CODE(p.res.p ? (p.res_f32 ? 2 : 1) : 0, p.out0.p ? 1 : 0, p.out0_f32 ? 1 : 0, p.out1.p ? 1 : 0, p.pixscale ? 1 : 0) | \
                                 (ep_call ? 0 : EP_RAGGED) | (p.pool_hw ? EP_POOLB : 0) | ((MODE == MODE_STD && p.spmul) ? EP_SPMULB : 0) | \
                                 ((EP_HEAVY && p.act0 == ACT_GELU) ? EP_GELUB : 0); \
        } \
        bool ep_done = false; \
        if constexpr (EP_FAST && MODE == MODE_SPADE) { \
            if (ep_code == EP_CODE(1, 1, 0, 0, 0)) { CONV_EPILOGUE_IMPL(EP_CODE(1, 1, 0, 0, 0)); ep_done = true; } \
        } \
        if constexpr (EP_FAST && (MODE == MODE_STD || MODE == MODE_STDSTAT)) { \
            if (ep_code == EP_CODE(0, 1, 0, 0, 0)) { CONV_EPILOGUE_IMPL(EP_CODE(0, 1, 0, 0, 0)); ep_done = true; } \
        } \
        if constexpr (MODE == MODE_STD && (EP_POOL_WSH_V) > 0) { \
            if (ep_code == (EP_CODE(0, 1, 0, 0, 0) | EP_POOLB)) { CONV_EPILOGUE_IMPL(EP_CODE(0, 1, 0, 0, 0) | EP_POOLB); ep_done = true; } \
        } \
        if constexpr (EP_FAST && MODE == MODE_STD && (EP_SPMUL_V)) { \
            if (ep_code == (EP_CODE(1, 1, 0, 0, 0) | EP_SPMULB)) { CONV_EPILOGUE_IMPL(EP_CODE(1, 1, 0, 0, 0) | EP_SPMULB); ep_done = true; } \
        } \
        if constexpr (EP_FAST && (MODE == MODE_STD || MODE == MODE_STDSTAT) && WCH != 5) { \
            if (ep_code == EP_CODE(1, 1, 0, 0, 0)) { CONV_EPILOGUE_IMPL(EP_CODE(1, 1, 0, 0, 0)); ep_done = true; } \
        } \
        if constexpr (EP_FAST && MODE == MODE_STD && (EP_MLIN_V)) { \
            if (ep_code == (EP_CODE(0, 1, 1, 0, 0) | EP_GELUB)) { CONV_EPILOGUE_IMPL(EP_CODE(0, 1, 1, 0, 0) | EP_GELUB); ep_done = true; } \
            if (ep_code == EP_CODE(0, 1, 1, 0, 0)) { CONV_EPILOGUE_IMPL(EP_CODE(0, 1, 1, 0, 0)); ep_done = true; } \
            if (ep_code == EP_CODE(2, 1, 1, 0, 0)) { CONV_EPILOGUE_IMPL(EP_CODE(2, 1, 1, 0, 0)); ep_done = true; } \
            if (ep_code == EP_CODE(0, 1, 1, 1, 0)) { CONV_EPILOGUE_IMPL(EP_CODE(0, 1, 1, 1, 0)); ep_done = true; }      /* F.second: fp32 volume + its fp16 copy */ \
        } \
        if constexpr (EP_FAST && MODE == MODE_STD && (EP_DUP_V)) { \
            if (ep_code == EP_CODE(0, 1, 0, 1, 0)) { CONV_EPILOGUE_IMPL(EP_CODE(0, 1, 0, 1, 0)); ep_done = true; } \
        } \
        if constexpr (EP_FAST && MODE == MODE_STD && (EP_O1ONLY_V)) { \
            if (ep_code == EP_CODE(1, 0, 0, 1, 0)) { CONV_EPILOGUE_IMPL(EP_CODE(1, 0, 0, 1, 0)); ep_done = true; } \
        } \
        if constexpr (EP_FAST && MODE == MODE_STD && WCH == 4) { \
            if (ep_code == EP_CODE(2, 1, 1, 1, 0)) { CONV_EPILOGUE_IMPL(EP_CODE(2, 1, 1, 1, 0)); ep_done = true; } \
        } \
        if constexpr (EP_FAST && MODE == MODE_STD && WCH == 5) { \
            if (ep_code == (EP_CODE(0, 1, 0, 0, 0) | EP_RAGGED)) { CONV_EPILOGUE_IMPL(EP_CODE(0, 1, 0, 0, 0) | EP_RAGGED); ep_done = true; } \
        } \
        if (!ep_done) { CONV_EPILOGUE_IMPL(-1); } \
    }
