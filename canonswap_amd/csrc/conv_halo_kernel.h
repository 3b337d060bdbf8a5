// Convolution with LDS-staged input patches for gfx950 (CDNA4).
//
// GEMM view: M = positions, N = Cout, K = taps x Cin; packed weights [kstep][Cout_pad][32]; data movement:
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
// LDS image: [halo voxel][CK channels] fp16, voxel stride CK*2+16 bytes.
#pragma once
#include <type_traits>
#include <utility>
#include <cstdlib>
#include "common.h"
#ifdef CS_TIMELINE
#define EP_TL(i) TL_STAMP(i)
#endif
#define EP_POOL_WSH_V EP_POOL_WSH_K      /* lane shifts of the pooling epilogue: per instantiation, from the lane -> position map (below) */
#define EP_POOL_HSH_V EP_POOL_HSH_K
#define EP_SPMUL_V EP_SPMUL_K            /* kernels that carry a branch-free copy of the spmul epilogue (below) */
#define EP_DUP_V EP_DUP_K                /* kernels that carry a branch-free copy of the two-identical-outputs epilogue (below) */
#define EP_MLIN_V EP_MLIN_K              /* kernels that carry branch-free copies of the motion extractor's three epilogue forms (below) */
#define EP_O1ONLY_V EP_O1ONLY_K          /* kernels that carry a branch-free copy of the residual + second-output-only epilogue (below) */
#define EP_O0_EXTRA_V ep_o0_extra        /* the output phase of a grouped launch (ConvParams::nphase; 0 otherwise) */
#include "conv_epilogue.h"

// Weight fragments are streamed with loads the compiler does not track (inline asm) and are waited for with an
// explicit counted s_waitcnt: hipcc drains vmcnt to 0 at every loop back-edge for loads it tracks, which would
// collapse the PFD-deep register ring to an effective depth of one K-step.
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned mdiv(unsigned u, unsigned magic) { return magic ? __umulhi(u, magic) : u; }

template <int OFF>
__device__ __forceinline__ void wfrag_load(u4_t& dst, const half_t* ptr)
{
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_le()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifdef CS_TIMELINE
extern unsigned long long* g_cs_tl;
extern long g_cs_tl_cap;
// stamp i of this wave (lane 0 stores at the end); sched barriers keep the phases where they are written
#define TL_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); tl_t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TL_STAMP(i) do { } while (0)
#endif

// Static shapes (ST): the kernel extent and the position tile are compile-time constants, so the tap loop of a chunk is
// fully unrolled: LDS window offsets become instruction immediates and all iterator arithmetic disappears.
//   ST 0: everything dynamic     ST 1: 1x3x3, tile 16x8        ST 2: 3x3x3, tile 8x8x2       ST 3: 3x3x3, tile 4x4x16
//   ST 4: 7x7x1 (the kw-split mask conv), tile 8x8x2            ST 5: 3x3x3, tile 4x4x8 (4x4 hourglass level)
//   ST 6: 7x7x1, tile 2x8x8 (no halo along W: 392 halo voxels instead of 896, double-buffered)
//   ST 16: 1x3x3, tile 16x16 (256 positions x 64 channels: the 64-channel 3x3 convs of G's last up block and F's first down block)
//   ST 12 / 13 / 17: 3x2x2, tiles 4x4x16, 8x8x2 and 4x4x8 (the hourglass up-blocks per output phase on the source grid)
//   ST 10 / 11 / 14 / 15: 1x2x2, 1x2x1, 1x1x2, 1x1x1, tile 16x8 (the per-phase convs of mlp_shared on the up-sampled seg, run_G)
template <int ST> struct StaticShape { static constexpr int KD = 0, KH = 0, KW = 0, LW = 0, LH = 0, LD = 0; };
template <> struct StaticShape<1> { static constexpr int KD = 1, KH = 3, KW = 3, LW = 4, LH = 3, LD = 0; };
template <> struct StaticShape<2> { static constexpr int KD = 3, KH = 3, KW = 3, LW = 3, LH = 3, LD = 1; };
template <> struct StaticShape<3> { static constexpr int KD = 3, KH = 3, KW = 3, LW = 2, LH = 2, LD = 4; };
template <> struct StaticShape<4> { static constexpr int KD = 7, KH = 7, KW = 1, LW = 3, LH = 3, LD = 1; };
template <> struct StaticShape<5> { static constexpr int KD = 3, KH = 3, KW = 3, LW = 2, LH = 2, LD = 3; };
template <> struct StaticShape<6> { static constexpr int KD = 7, KH = 7, KW = 1, LW = 1, LH = 3, LD = 3; };
template <> struct StaticShape<7> { static constexpr int KD = 3, KH = 3, KW = 3, LW = 3, LH = 3, LD = 2; };   // 256 positions: 8x8x4
template <> struct StaticShape<8> { static constexpr int KD = 7, KH = 7, KW = 1, LW = 1, LH = 3, LD = 4; };   // 256 positions: 2x8x16
template <> struct StaticShape<9> { static constexpr int KD = 7, KH = 7, KW = 1, LW = 2, LH = 3, LD = 3; };   // 256 positions: 4x8x8 (the mask conv with the 4-column kw sum)
template <> struct StaticShape<16> { static constexpr int KD = 1, KH = 3, KW = 3, LW = 4, LH = 4, LD = 0; };   // 256 positions: 16x16 (2-D)
template <> struct StaticShape<12> { static constexpr int KD = 3, KH = 2, KW = 2, LW = 2, LH = 2, LD = 4; };
template <> struct StaticShape<13> { static constexpr int KD = 3, KH = 2, KW = 2, LW = 3, LH = 3, LD = 1; };
template <> struct StaticShape<17> { static constexpr int KD = 3, KH = 2, KW = 2, LW = 2, LH = 2, LD = 3; };   // 3x2x2, tile 4x4x8 (up-block 0 of the hourglass per phase: 4x4 source grid)
template <> struct StaticShape<10> { static constexpr int KD = 1, KH = 2, KW = 2, LW = 4, LH = 3, LD = 0; };
template <> struct StaticShape<11> { static constexpr int KD = 1, KH = 2, KW = 1, LW = 4, LH = 3, LD = 0; };
template <> struct StaticShape<14> { static constexpr int KD = 1, KH = 1, KW = 2, LW = 4, LH = 3, LD = 0; };
template <> struct StaticShape<15> { static constexpr int KD = 1, KH = 1, KW = 1, LW = 4, LH = 3, LD = 0; };

// ---- LDS image of the halo and the bank conflicts of the fragment reads (model and search: tools/lds_bank_search.py).
// ds_read_b128 serves a wave in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) over 64 banks of 4 bytes.  A lane
// (l15 = position, l4 = k slot) reads voxel(position) * VS + l4 * 16, so with one pad slot (voxel stride 5 / 9 slots of 16 bytes) every
// group hits half of its bank quads twice - three times where a position block is rows of 8 or 4 voxels - which is what the counters
// show (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.50-0.67, profiles/r02_g_wave_state.txt).  Two levers:
//   * which of the 16 positions of a block a lane works on is free (the epilogue uses the same map): a bit permutation of the lane
//     index takes the volume tiles from 3 to 2 cycles per group at the same stride (halo_lane_pos);
//   * a voxel stride of 6 / 10 slots (two pad slots) is conflict-free for 16 consecutive voxels at any alignment and, with the
//     permutation, for the volume tiles; it costs 20 % / 11 % more LDS and is used where the occupancy stays the same (halo_pad).
template <int CK, int WPX, int WCH, int WVP, int ST, int MODE> constexpr int halo_pad()
{
#ifdef CS_LDS_V1
    return 1;
#else
    if (CK == 64 && WCH == 4 && WPX == 8 && ST == 1) return 2;                       // 128x256 tiles (T, wide G / R convs): 2 workgroups per CU
    if (CK == 32 && WCH == 5 && WPX == 8 && (ST == 7 || ST == 8 || ST == 9)) return 2;           // 256x160 tiles: 1 workgroup per CU
    if (CK == 64 && WCH == 2 && WPX == 8 && WVP == 2 && ST == 16) return 2;           // 256x64 2-D tiles: 2 workgroups per CU (2 x 52 KB)
    // SPADE gamma/beta convs (128x128, three workgroups per CU): the 64-channel image would not fit three times with two pad slots
    // (173 KB) and measured slower at two workgroups; with 32-channel chunks it does (104 KB) - the engine launches them that way
    if (CK == 32 && WCH == 2 && WPX == 8 && ST == 1 && MODE == MODE_SPADE) return 2;
#ifdef CS_LDS_V32
    if (CK == 32 && WCH == 2 && WPX == 4 && WVP == 4 && ST == 3) return 2;            // 256x32 volume tiles: 3 -> 2 workgroups per CU
#endif
    return 1;
#endif
}
// pieces per thread whose source offsets are kept in registers = ceil(halo voxels * slots per voxel / 256)
template <int ST, int PAD> constexpr int halo_hi()
{
    if (PAD == 2) return ST == 3 ? 16 : (ST == 9 ? 19 : ((ST == 7 || ST == 8) ? 15 : (ST == 16 ? 13 : 8)));
    return ST == 4 ? 18 : ((ST == 3 || ST == 7 || ST == 8 || ST == 9) ? 13 : (ST == 12 ? 9 : 8));
}
// position (0..15 within its block) that lane l15 works on
template <int ST, int PAD> __host__ __device__ constexpr int halo_lane_pos(int l)
{
#ifdef CS_LDS_V1
    return l;
#else
    using SS = StaticShape<ST>;
    if (ST == 0 || SS::KW != 3 || SS::LW > 3 || SS::LW < 2) return l;
    if (SS::LW == 2 && PAD == 2) return ((l & 1) << 1) | ((l & 2) << 2) | ((l >> 2) & 1) | ((l & 8) >> 1);      // bits (1,3,0,2)
    return ((l & 1) << 1) | ((l & 2) << 1) | ((l >> 2) & 1) | (l & 8);                                          // bits (1,2,0,3)
#endif
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}) - the K-steps of the ASMR kernels need
// their index as a constant expression (the counted waits are instruction immediates)
template <class F, int... I> __device__ __forceinline__ void halo_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void halo_static_for(F&& f) { halo_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// ASMR kernels: VMEM instructions younger than the ring loads of K-step `st` at the moment that step starts.  Issue order of a chunk of L steps:
// step u >= 1 reloads (WCH loads) the slot step u - 1 used, then - if the chunk stages a next one and ST0 <= u < ST0 + HI - issues one DMA
// piece; behind the last step the slot it used is reloaded.  A reload at step u fetches step u - 1 + PFS of the same chunk, or - past its end -
// the step of the NEXT chunk that slot serves there (L % PFS == 0: steps 0, 1, ... in order; else step (u - 1) % PFS).  The same code runs as
// the first chunk of a tile (ring primed with steps 0 .. PFS - 1 in order, behind the staging burst) and behind a full chunk of NP steps
// that staged pieces: the smaller count (the stronger wait) of the two cases is used.
template <int WCH, int PFS, int HI, int ST0>
constexpr int halo_ring_young(int st, int L, int NP, bool dma_cur)
{
    auto dma = [&](int u, bool on) { return (on && u >= ST0 && u < ST0 + HI) ? 1 : 0; };
    // events of the current chunk before step st starts, younger than a given point
    auto cur_before = [&](int from_step /* events of steps from_step .. st - 1 */) {
        int n = 0;
        for (int u = from_step; u < st; ++u) n += (u >= 1 ? WCH : 0) + dma(u, dma_cur);
        return n;
    };
    if (st >= PFS) {
        const int u0 = st - PFS + 1;                      // reload group of step u0 fetched step st
        return dma(u0, dma_cur) + cur_before(u0 + 1);
    }
    // fetched before the chunk: (a) primed, (b) carried from the previous chunk
    const int prime = (PFS - 1 - st) * WCH + cur_before(0);
    int g = 0;                                            // reload group (1 .. NP) of the previous chunk that fetched this chunk's step st
    if (NP % PFS == 0) g = st + NP - PFS + 1;
    else { for (int u = NP - PFS + 1; u <= NP; ++u) if ((u - 1) % PFS == st) g = u; }
    int carried = dma(g, true);
    for (int u = g + 1; u <= NP; ++u) carried += WCH + (u < NP ? dma(u, true) : 0);
    carried += cur_before(0);
    return prime < carried ? prime : carried;
}

// lane shift (1, 2, 4, 8) that moves a lane to the neighbour whose position differs in bit `bit` of the 16-position block index
template <int ST, int PAD> constexpr int halo_pool_shift(int bit)
{
    for (int b = 0; b < 4; ++b) if (halo_lane_pos<ST, PAD>(1 << b) == (1 << bit)) return 1 << b;
    return 0;
}

// kernels that can run a grouped launch (ConvParams::nphase)
template <int WCH, int ST> constexpr bool halo_phase_group() { return !(ST == 0 && WCH >= 4); }

// kernels that carry the 2-D pooling epilogue (conv_epilogue.h, EP_POOL_HB)
template <int CK, int WPX, int WCH, int WVP, int MODE, bool SK, int ST> constexpr bool halo_pool2d()
{
    using SS = StaticShape<ST>;
    return ST != 0 && !SK && MODE == MODE_STD && CK == 32 && WPX == 8 && WVP == 1 && (WCH == 2 || WCH == 4) && SS::KD == 1 && SS::KH == 3 && SS::KW == 3 &&
           SS::LW == 4 && SS::LH == 3 && SS::LD == 0;
}

template <int CK, int WPX, int WCH, int WVP, int WVC, int MODE, bool DB, bool SK, int ST>
// Resident workgroups per CU the register budget is held to: 2 for the 128x256 tiles (256 VGPRs); 3 for the statically
// unrolled 128x128 tiles (<= 168 VGPRs, no scratch) -- for short-K convs a third workgroup hides the prologue / epilogue
// of the other two (gamma/beta convs K = 9 x 128: +15 % over the 128x256 tile, tools/ab_conv.sh); the dynamic-shape
// variants would spill at that budget and stay unconstrained.
__global__ void __launch_bounds__(256, (WCH == 4 ? 2 : ((WCH == 2 && WPX == 8 && !SK && ST != 0) ? (WVP == 2 ? 2 : 3) : 1))) conv_halo_kernel(const ConvParams p)
{
    using SS = StaticShape<ST>;
    static_assert(ST == 0 || !SK, "static shapes are not combined with split-K");
    constexpr int BM = SK ? WPX * 16 : WPX * 16 * WVP;
    constexpr int BN = WCH * 16 * WVC;
    constexpr int SL = CK / 8;           // 16-byte slots per voxel
    constexpr int PAD = halo_pad<CK, WPX, WCH, WVP, ST, MODE>();
    constexpr int SLP = SL + PAD;        // ... plus the pad slot(s) (bank spreading; also fetched, from the zero page)
    // pooling epilogue (ConvParams::pool_hw): 3-D static tiles of 8 or 4 columns - the w and h neighbours of a position lie in its 16-position block
    constexpr bool EP_POOLK = ST != 0 && !SK && MODE == MODE_STD && WCH == 2 && SS::KD == 3 && SS::KH == 3 && SS::KW == 3 && (SS::LW == 3 || SS::LW == 2) && SS::LH >= 1;
    // ... and the 2-D 16 x 8 tiles of F's down blocks (32-channel chunks: their weights are the [W_hi | W_lo] groups): a block is a row of 16 columns
    constexpr bool EP_POOLK2 = halo_pool2d<CK, WPX, WCH, WVP, MODE, SK, ST>();
    // ConvParams::spmul: the 128 x 256 tiles the gamma convs of G's two learned shortcuts run on
    constexpr bool XSK = ST == 1 && !SK && MODE == MODE_STD && CK == 64 && WCH == 4 && WPX == 8 && WVP == 1;       // ConvParams::xs_w (the fused shortcut epilogue)
#ifdef CS_NO_SPMUL_FAST
    constexpr bool EP_SPMUL_K = false;
#else
    constexpr bool EP_SPMUL_K = ST == 1 && !SK && MODE == MODE_STD && CK == 64 && WCH == 4 && WPX == 8 && WVP == 1;
#endif
    // G.up_1.conv_1 (64 -> 64 at 256 x 256; res + out1 = lrelu(.), no out0) ran the general epilogue: 21 000 of a wave's 41 000 cycles against
    // 10 000 of main loop (profiles/r06_q_timeline_b64.txt); only the 2-D 256x64 kernel carries the copy
    constexpr bool EP_O1ONLY_K = ST == 16 && !SK && MODE == MODE_STD;
    // the 1x1 kernels (M's 39 linear layers: K of 9 - 288 steps per tile, so the epilogue is most of a tile's life; they ran the general epilogue with
    // the per-element activation switch of EP_HEAVY)
    constexpr bool EP_MLIN_K = ST == 15 && !SK && MODE == MODE_STD;
    // the per-phase convs of mlp_shared with one source row (ST 14 / 15: 1x1x2 and 1x1x1 taps): output rows 4i + 1 and 4i + 2 of the x4 level are the
    // same values - one launch writes both (out1 through the identity second affine) instead of two launches computing them
    // (ST 11, 1x2x1 taps: the two middle column phases of a two-row phase - one value for two neighbouring pixels)
    constexpr bool EP_DUP_K = (ST == 11 || ST == 14 || ST == 15) && !SK && MODE == MODE_STD && WCH == 2 && WPX == 8;
    constexpr int EP_POOL_WSH_K = EP_POOLK ? halo_pool_shift<ST, PAD>(0) : (EP_POOLK2 ? 1 : 0);
    constexpr int EP_POOL_HSH_K = EP_POOLK ? halo_pool_shift<ST, PAD>(SS::LW) : 0;
    constexpr int VS = SLP * 16;         // LDS bytes per halo voxel
    // halo pieces per thread whose source offsets are kept in registers (the big halos of the mask conv / 4x4x16 tiles too)
    constexpr int HI = halo_hi<ST, PAD>();
    constexpr int KH32 = CK / 32;        // 32-channel K-steps per tap and chunk
    constexpr int PFD = 4;               // weight prefetch depth in K-steps
    constexpr int SKS = SK ? 4 : 1;      // K-step stride of one wave
    static_assert(WVP * WVC == 4, "4 waves per workgroup");
    static_assert(!SK || (WVC == 1 && WPX == 8), "split-K variants: every wave covers all 128 positions");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef CS_TIMELINE
    unsigned long long tl_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tl_real0 = __builtin_amdgcn_s_memrealtime();     // constant 100 MHz clock
    TL_STAMP(0);
#endif

#ifdef CS_DEPHASE        // experiment: workgroups of the first dispatch round in an odd wave slot start CS_DEPHASE x 1024 cycles late
    if (MODE == MODE_SPADE && WCH == 4) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if ((hwid & 1u) && (int)(blockIdx.x + blockIdx.y * gridDim.x) < 512)
            for (int i = 0; i < CS_DEPHASE; ++i) __builtin_amdgcn_s_sleep(16);
    }
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // grouped launch (ConvParams::nphase): blockIdx.z picks the phase's weights, leading padding and output offset
    // (not in the dynamic-shape 128 x 256 kernels: with anything added hipcc re-allocates their hand-counted weight ring - _lib.isa_check - and no
    // grouped layer runs on them)
    constexpr bool PHK = halo_phase_group<WCH, ST>();
    const int phz = (PHK && p.nphase) ? (int)blockIdx.z : 0;
    const int PHv = (PHK && p.nphase) ? p.ph_PH[phz] : p.PH, PWv = (PHK && p.nphase) ? p.ph_PW[phz] : p.PW;
    const unsigned ep_o0_extra = (PHK && p.nphase) ? p.ph_ooff[phz] : 0u;
    const int wpx = SK ? 0 : wave % WVP, wch = SK ? 0 : wave / WVP;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int l15p = halo_lane_pos<ST, PAD>(l15);      // the position of its 16-position blocks this lane works on
#ifdef CS_NO_PAIR                               // A/B builds (tools/build_variant.py)
    constexpr int EP_PAIR = 0;
#else
    constexpr int EP_PAIR = ep_pair_of(MODE, WCH);     // weight-row permutation / joint stores of channel pairs (conv_epilogue.h)
#endif

    // ---- persistent mode (the ASMR kernels, ConvParams::persist_total > 0; see the ASMR block below): the workgroup walks a list of tiles.
    // XCD x owns persist_total / 8 consecutive entries of the launch order (as in xcd_map 2), its workgroups take them round robin, so the
    // tiles in flight in an XCD are neighbours; the state below lives across tiles: the weight ring (the last steps of a tile fetch the
    // first steps of the next one: the weights are the same), the staging offsets / buffer of the tile being staged (the next tile's first
    // chunk is staged under the last chunk's MFMAs), the LDS buffer parity.
#ifdef CS_NO_ASMRING
    constexpr bool ASMRK = false;
#else
    constexpr bool ASMRK = DB && !SK && CK == 32 && WCH == 5 && WPX == 8 && WVP == 2 && (ST == 7 || ST == 8 || ST == 9) && MODE == MODE_STD;
#endif
    u4_t wr_p[ASMRK ? 3 : 1][ASMRK ? WCH : 1];
    unsigned poffb[ASMRK ? HI : 1];
    __amdgpu_buffer_rsrc_t rsrc_p;
    int gbuf = 0;
    bool staged = false;                          // this tile's first chunk and ring were staged by the previous tile
    const bool persist = ASMRK && p.persist_total > 0;
    int p_base = 0, p_cnt = 0, p_j = 0, p_i = 0, p_per = 1;
    if (persist) {
        const int total = p.persist_total, G = (int)gridDim.x;
        const int xcd = blockIdx.x & 7, q = total >> 3, r = total & 7;
        p_i = blockIdx.x >> 3; p_per = G >> 3;
        p_base = xcd * q + (xcd < r ? xcd : r); p_cnt = q + (xcd < r ? 1 : 0);
        if (p_i >= p_cnt) return;
    }
    for (;;) {
    // position tile / channel block of this workgroup (ConvParams::xcd_map)
    int tile_lin = blockIdx.x, cblk = blockIdx.y;
    const bool has_next = persist && p_i + p_per * (p_j + 1) < p_cnt;
    auto u_to_tile = [&](int u, int& tl_, int& cb_) {
        const int ncb = p.Cout_pad / BN;
        tl_ = (int)mdiv((unsigned)u, p.mg_ncb); cb_ = u - tl_ * ncb;
    };
    if (persist) u_to_tile(p_base + p_i + p_per * p_j, tile_lin, cblk);
    else if (p.xcd_map != 0) {
        const int flat = p.xcd_map == 2;
        const int total = (int)gridDim.x;                               // mode 1: tiles; mode 2: tiles x channel blocks
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        const int u = xcd * q + (xcd < r ? xcd : r) + i;                // bijection: XCD x owns q + (x < r) consecutive entries
        if (flat) u_to_tile(u, tile_lin, cblk);
        else tile_lin = u;
    }
    // (mg_*: multiply-high constants of launch_halo_st for these divisions)
    int t = tile_lin, tq;
    tq = (int)mdiv((unsigned)t, p.mg_tw); const int tw = t - tq * p.nTW; t = tq;
    tq = (int)mdiv((unsigned)t, p.mg_th); const int th = t - tq * p.nTH; t = tq;
    tq = (int)mdiv((unsigned)t, p.mg_td); const int td = t - tq * p.nTD; t = tq;
    const int tn = t;
    const int n0 = cblk * BN;
    const int lgTW = ST ? SS::LW : p.lgTW, lgTH = ST ? SS::LH : p.lgTH, lgTD = ST ? SS::LD : p.lgTD;
    const int KD = ST ? SS::KD : p.KD, KH = ST ? SS::KH : p.KH, KW = ST ? SS::KW : p.KW;
    const int lgS = lgTW + lgTH + lgTD;
    const int mW = (1 << lgTW) - 1, mH = (1 << lgTH) - 1, mD = (1 << lgTD) - 1;
    const int TN = BM >> lgS;
    const int HW = (1 << lgTW) + KW - 1, HH = (1 << lgTH) + KH - 1, HD = (1 << lgTD) + KD - 1;
    const int HV = TN * HD * HH * HW;
    const int nitems = HV * SLP;
    const int w0 = tw << lgTW, h0 = th << lgTH, d0 = td << lgTD, nb = tn * TN;
    const int ntaps = KD * KH * KW;

    // ---- halo staging, global -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass).
    // The LDS image is linear in the piece index q = voxel*SLP + slot (SLP = SL data slots + 1 pad slot), which is
    // what the instruction requires (wave-uniform base + lane*16). Out-of-range / pad pieces read the zero page.
    // 32-bit offsets; the per-axis products are 24-bit multiplies (full rate; v_mul_lo_u32 / the 64-bit forms are quarter rate and
    // this runs 13 times per thread in the volume kernels): coordinates are small and launch_halo_st refuses axis strides >= 2^23
    const int isN = (int)p.in_sN, isD = (int)p.in_sD, isH = (int)p.in_sH, isW = (int)p.in_sW;
    const int in_nb = nb * isN;                                                   // scalar
    auto piece_off = [&](int q, bool& inb) -> int {       // element offset of piece q's voxel (without the channel part)
        const int hv = q / SLP;
        const int hw = hv % HW; int r = hv / HW;
        const int hh = r % HH; r /= HH;
        const int hd = r % HD;
        const int hn = TN == 1 ? 0 : r / HD;
        const int n = nb + hn, id = d0 + hd - p.PD, ih = h0 + hh - PHv, iw = w0 + hw - PWv;
        inb = q < nitems && (q % SLP) < SL && (TN == 1 ? r < HD : true) && n < p.N && (unsigned)id < (unsigned)p.D &&
              (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        return in_nb + (TN == 1 ? 0 : hn * isN) + __mul24(id, isD) + __mul24(ih >> p.up_shift, isH) + __mul24(iw >> p.up_shift, isW) +
               (q % SLP) * 8;
    };
    // few pieces per thread: keep their offsets in registers (always the case in double-buffered mode, see launcher)
    const bool pre = DB || nitems <= 256 * HI;
    int poff[HI];
    unsigned pmask = 0;
    if (pre) {
#pragma unroll
        for (int j = 0; j < HI; ++j) {
            bool inb;
            const int o = piece_off(tid + 256 * j, inb);
            poff[j] = inb ? o : 0;
            pmask |= inb ? (1u << j) : 0u;
        }
    }
    auto glds = [&](const half_t* src, int buf, int q_wave_base) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + (size_t)buf * HV * VS + (size_t)q_wave_base * 16),
                                         16, 0, 0);
    };
    auto stage_halo = [&](int buf, int c0) {     // asynchronous: completion is awaited by the next __syncthreads()
        // channel part of the source offset; grouped mode: chunk j -> group j / cg at stride in_sG
        long coff = c0;
        int climit = p.Cin - c0;                 // channels of this chunk that exist
        // (64-channel chunks: the 1x1 kernels and the dynamic-shape ones they fall back to on small maps, with an even cg - the two
        // 32-channel halves of a chunk are then neighbours in memory; the launcher checks.  Compiled into nothing else: the 128 x 256
        // kernels with hand-counted rings tolerate no addition, DESIGN 5.6 rule 9)
        if ((CK == 32 || ST == 15 || (ST == 0 && WCH != 4)) && p.cg > 0) {
            const int j = c0 >> 5;
            coff = (long)(j / p.cg) * p.in_sG + (j % p.cg) * 32;
            climit = p.cg_cin - (j % p.cg) * 32;
        }
        if (pre) {
#pragma unroll
            for (int j = 0; j < HI; ++j) {
                const int q = tid + 256 * j;
                if (q < nitems) {
                    const bool ok = ((pmask >> j) & 1u) && ((q % SLP) * 8 < climit);
                    glds(ok ? p.in + poff[j] + coff : p.zero, buf, 256 * j + wave * 64);
                }
            }
        } else if constexpr (!DB) {
            for (int q0 = 0; q0 < nitems; q0 += 256) {
                const int q = q0 + tid;
                if (q < nitems) {
                    bool inb;
                    const int o = piece_off(q, inb);
                    const bool ok = inb && ((q % SLP) * 8 < climit);
                    glds(ok ? p.in + o + coff : p.zero, buf, q0 + wave * 64);
                }
            }
        }
    };

    // ---- per-lane constants of the MFMA operand fetches
    int abase[WPX];                      // LDS byte offset of this lane's position in the un-shifted halo window
#pragma unroll
    for (int pi = 0; pi < WPX; ++pi) {
        int m = wpx * WPX * 16 + pi * 16 + l15p;
        const int wl = m & mW; m >>= lgTW;
        const int hl = m & mH; m >>= lgTH;
        const int dl = m & mD; m >>= lgTD;
        abase[pi] = (((m * HD + dl) * HH + hl) * HW + wl) * VS + l4 * 16;
    }
    const int nck = (p.Cin + CK - 1) / CK;
    const int j0 = SK ? wave : 0;
    // cross-workgroup split-K (ConvParams::sk_out): blockIdx.z owns the channel chunks [cc_lo, cc_hi) and leaves raw partial sums
    int cc_lo = 0, cc_hi = nck;
    if (p.sk_out) { cc_lo = (int)(((long)nck * blockIdx.z) / gridDim.z); cc_hi = (int)(((long)nck * (blockIdx.z + 1)) / gridDim.z); }

    // weights: fragment ci of K-step kidx = 1 KiB at wgt + (kidx*Cout_pad + n0 + wch*WCH*16 + ci*16)*32; lane = (row l15, k l4*8)
    const half_t* wbase = p.wgt;
    if (p.wslot) wbase += p.wofs[p.wslot[nb < p.N ? nb : p.N - 1]];         // per-sample weight set (uniform over the tile)
    if (PHK && p.nphase) wbase += p.ph_wofs[phz];
    // EP_PAIR (conv_epilogue.h): the rows of a fragment pair are permuted so that a lane ends up with 8 consecutive output channels
    const half_t* wlane = wbase + ((long)(n0 + wch * WCH * 16) * 32 + ep_lane_row(EP_PAIR, l15) * 32 + l4 * 8);
    const long wstep = (long)p.Cout_pad * 32;

    // SPADE: fetch the modulated tensor of the whole tile now (conv_epilogue.h); only the 3-waves-per-SIMD 128x128 kernel has the
    // 16 registers to spare, and launch_halo_st refuses an fp32 operand for it
#ifdef CS_NO_EARLY
    constexpr bool EP_EARLY = false;
#else
    constexpr bool EP_EARLY = (MODE == MODE_SPADE) && !SK && ST != 0 && WCH == 2 && WPX == 8;
#endif
    // branch-free copies of the epilogue for the tensor combinations of the engine's hot layers (conv_epilogue.h, CONV_EPILOGUE)
#ifdef CS_NO_EPFAST
    constexpr bool EP_FAST = false;
#else
    constexpr bool EP_FAST = !SK && ST != 0 && (WCH == 4 || WCH == 5 || (WCH == 2 && WPX == 8) || ST == 15);
#endif
    constexpr int EP_WPX0 = WPX;
    const int ep_wpx0 = wpx;
    CONV_EPILOGUE_EARLY_FETCH()

    f4_t acc[WCH][WPX];
#pragma unroll
    for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
        for (int pi = 0; pi < WPX; ++pi) acc[ci][pi] = (f4_t){0.f, 0.f, 0.f, 0.f};

    if constexpr (ST != 0) {
        // ---------------- static shape: the K-steps of a chunk are fully unrolled
        constexpr int NT = SS::KD * SS::KH * SS::KW, NS = NT * KH32;
        // kernels that carry the paired-tap body for a ragged last chunk (the 256-position tiles of the hourglass tail, the mask
        // conv and the first encoder block)
        constexpr bool RAGK = (CK == 32) && (WPX == 8) && (ST == 7 || ST == 8 || ST == 9) && MODE == MODE_STD;
        // weight ring depth; the ring is re-primed at every chunk.  Three steps cover an L2 round trip where a step is 16-40 MFMAs; the
        // 16- and 32-channel tiles run 2-4 MFMAs per step and were bound by that latency (T's mask conv: 80 us for 134 MB): 8 steps
        // LATK: the 128 x 64 tiles of the 512-channel 3x3 convs (64-channel chunks, 4 x 2 fragments per wave) only run below three frames
        // per launch: one workgroup per CU, one wave per SIMD, and the weights come from HBM (every conv's set is read once per frame).  A ring
        // of 6 steps carried over the chunk boundary: R's 2-D convs 0.251 -> 0.215 ms, G's 512 -> 512 convs -9 % on the one-frame step
        // (profiles/r05_e_ab_lat_ring.txt).  The 128 x 128 tiles (T, SPADE; 168-register budget) lose with the deeper ring: not taken there.
        constexpr bool LATK = CK == 64 && WCH == 2 && WPX == 4 && ST == 1 && !SK;
#ifdef CS_PFS3
        constexpr int PFS = NS < 3 ? NS : 3;
#else
        constexpr int PFS = NS < 3 ? NS : (WCH * WPX <= 4 ? (NS < 8 ? NS : 8) : (LATK ? (NS < 6 ? NS : 6) : 3));
#endif
        // The ring is carried over the chunk boundary in the kernels that run one wave per SIMD (256-position tiles), where a chunk is a
        // whole number of ring turns (3x3x3 x 32 channels: 27 steps): the last PFS steps of a chunk fetch the first PFS steps of the next
        // one into the slots these expect, so the first MFMA behind the chunk barrier does not wait for an L2 round trip (wload_at clamps
        // the chunk index: behind the last chunk the fetches repeat its first steps and are dropped).  The ragged chunk has another step
        // count, but it is the last one.  Hourglass tail and first encoder block -4 %, +0.25 % on the step; with two or three waves per
        // SIMD the others fill that gap already and the longer live ranges cost the 128x256 kernels 12-24 bytes of scratch: -0.2 %
        // (profiles/r03_n_ab_wcarry.txt; -DCS_NO_WCARRY is the A/B switch).  Where a chunk is not a whole number of ring turns (the mask conv:
        // 49 steps) the slot a step s of the last turn frees takes the next chunk's step s % PFS - the step that slot serves there.
#ifdef CS_NO_WCARRY
        constexpr bool WCARRY = false;
#else
#ifdef CS_WCARRY_MULT
        constexpr bool WCARRY = (NS % PFS == 0) && NS >= 2 * PFS && WPX == 8 && WVP == 2;
#else
        constexpr bool WCARRY = NS >= 2 * PFS && ((WPX == 8 && WVP == 2) || LATK);
#endif
#endif
        constexpr int SHW = (1 << SS::LW) + SS::KW - 1, SHH = (1 << SS::LH) + SS::KH - 1;
        // ---- ASMR: the 256 x 160 tiles (hourglass tail, mask conv: one workgroup per CU, one wave per SIMD) stream their weight ring with
        // loads the compiler does not track and stage the next chunk's halo piece by piece UNDER the MFMAs, through buffer-addressed LDS
        // DMA - conv_wide.hip's scheme (see there) inside this kernel's tile / epilogue machinery.  With compiler-tracked loads every wait
        // for a weight fragment becomes vmcnt(0) while an LDS DMA is in flight, so the halo could only be staged as a burst of 15 pieces
        // at the chunk's head (13 % of a tail chunk) followed by a full drain.  Same K order, same bits.
#ifdef CS_NO_ASMRING
        constexpr bool ASMR = false;
#else
        constexpr bool ASMR = DB && WCH == 5 && WPX == 8 && WVP == 2 && (ST == 7 || ST == 8 || ST == 9) && MODE == MODE_STD && PFS == 3;
#endif
        if constexpr (ASMR) {
            static_assert(SK == false && KH32 == 1, "ASMR: 32-channel chunks");
            constexpr int HSTRIDE = HI * 4096;       // LDS bytes between the two buffers: the last piece's lanes beyond the image land in the gap
            constexpr int ST0 = 1;                   // K-steps ST0 .. ST0 + HI - 1 of a chunk each issue one DMA piece of the next chunk
            constexpr unsigned OOB = 0x80000000u;    // byte offset outside the buffer (the launcher keeps a sample below 2^31 bytes): the lane reads zeros
            // per-sample buffer: pad slots, voxels outside the volume, lanes beyond the image and channels beyond Cin carry OOB
            static_assert(ASMRK, "the kernel-level and the block-level conditions of ASMR agree");
            auto (&wr) = wr_p;
            // staging offsets of a tile (set_stage: this tile at the first tile of the workgroup, the NEXT tile at the head of every last chunk)
            auto set_stage = [&](int n_, int d0_, int h0_, int w0_) {
                rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (long)n_ * p.in_sN), 0, (int)p.in_sample_bytes, 0x00020000);
                int tid_o = tid;
                asm volatile("" : "+v"(tid_o));              // (opaque: the tile-invariant part of this addressing is not to be hoisted out of the tile loop)
#pragma unroll
                for (int j = 0; j < HI; ++j) {
                    const int q = tid_o + 256 * j;
                    const int hv = q / SLP, sl = q % SLP;
                    const int hw = hv % HW; int r = hv / HW;
                    const int hh = r % HH; r /= HH;
                    const int id = d0_ + r - p.PD, ih = h0_ + hh - PHv, iw = w0_ + hw - PWv;
                    const bool inb = q < nitems && sl < SL && r < HD && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    poffb[j] = inb ? (unsigned)(__mul24(id, isD) + __mul24(ih >> p.up_shift, isH) + __mul24(iw >> p.up_shift, isW) + sl * 8) * 2u : OOB;
                }
            };
            // piece j of chunk cn of the tile the staging offsets belong to; cn >= cc_hi: the first chunk of the NEXT tile (the offsets were
            // switched at the head of this tile's last chunk) - or, behind the workgroup's last tile, nothing (every lane out of range)
            auto stage_piece = [&](int buf, int cn, int j) {
                const bool nxt = cn >= cc_hi;
                const int c0 = (nxt ? cc_lo : cn) * CK;
                const int climit = (nxt && !has_next) ? 0 : p.Cin - c0;          // channels of that chunk that exist
                const unsigned off = (((tid + 256 * j) % SLP) * 8 < climit) ? poffb[j] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_p, (__attribute__((address_space(3))) void*)(smem + (size_t)buf * HSTRIDE + (size_t)(256 * j + wave * 64) * 16),
                                                         16, (int)off, c0 * 2, 0, 0);
            };
            auto wsrc_of = [&](int cc, int st, int nsc) -> const half_t* {       // st may run past the chunk (the carried fetches)
                int c2 = cc, s2 = st;
                if (st >= nsc) { c2 = cc + 1; s2 = (nsc % PFS == 0) ? st - nsc : (st - PFS) % PFS; }      // the slot a step frees serves that step of the next chunk
                const int ccl = c2 < cc_hi ? c2 : cc_lo;                         // behind a tile's last chunk: the next tile's first chunk (the same weights; dropped behind the last tile)
                return wlane + (long)(ccl * NT + s2) * wstep;
            };
            auto wload1 = [&](u4_t& dst, const half_t* src, int ci) {            // ci: compile-time after unrolling; EP_PAIR == 0 here: rows ci * 16
                if (ci == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src));
                else if (ci == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(dst) : "v"(src));
                else if (ci == 2) asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(dst) : "v"(src));
                else if (ci == 3) asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(dst) : "v"(src));
                else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src + 4 * 16 * 32));
            };
            static_assert(EP_PAIR == 0, "ASMR: plain weight row order");
            TL_STAMP(1);
            // (the 4-column mask tile, 19 staging offsets and the kw-sum epilogue: the ring does not stay live across the epilogue - that spilled the
            // epilogue's tables, and every scratch reload is a full drain of the in-order memory queue, output stores included - but is primed again)
            constexpr bool CARRY = ST != 9;
            if (!staged) {                                   // the workgroup's first tile: staging burst
                set_stage(nb, d0, h0, w0);
#pragma unroll
                for (int j = 0; j < HI; ++j) stage_piece(gbuf, cc_lo, j);
            }
            if (!staged || !CARRY) {                         // ring prime
#pragma unroll
                for (int st = 0; st < PFS; ++st)
#pragma unroll
                    for (int ci = 0; ci < WCH; ++ci) wload1(wr[st][ci], wsrc_of(cc_lo, st, NS), ci);
            }
            TL_STAMP(2);
            auto run_chunk_a = [&](int cc, auto rag_t) {
                constexpr bool RAG = decltype(rag_t)::value;
                // head: this chunk's halo has landed - this wave's pieces by the counted wait (the PFS * WCH ring fetches are younger), everyone's by
                // the barrier, which also says that everyone has left the other buffer
                wait_vmcnt_le<PFS * WCH>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const unsigned char* hb = smem + (size_t)gbuf * HSTRIDE;
                const int nbuf = gbuf ^ 1;
                gbuf ^= 1;
                if (cc + 1 == cc_hi && has_next) {           // the last chunk: from here on the staging offsets are the next tile's
                    int ntl, ncb_, t2 = 0, q2;
                    u_to_tile(p_base + p_i + p_per * (p_j + 1), ntl, ncb_);
                    t2 = ntl;
                    q2 = (int)mdiv((unsigned)t2, p.mg_tw); const int tw2 = t2 - q2 * p.nTW; t2 = q2;
                    q2 = (int)mdiv((unsigned)t2, p.mg_th); const int th2 = t2 - q2 * p.nTH; t2 = q2;
                    q2 = (int)mdiv((unsigned)t2, p.mg_td); const int td2 = t2 - q2 * p.nTD; t2 = q2;
                    set_stage(t2 * TN, td2 << lgTD, th2 << lgTH, tw2 << lgTW);
                }
                constexpr int HA = WPX / 2;
                constexpr int PK = SS::KW > 1 ? SS::KW : SS::KH, SPR = PK / 2 + PK % 2;
                constexpr int NSC = RAG ? (NT / PK) * SPR : NS;
                auto tap_of = [&](int st) -> int { return RAG ? (st / SPR) * PK + 2 * (st % SPR) : st; };
                auto paired = [&](int st) -> bool { return RAG && (st % SPR) < PK / 2; };
                auto toff_of = [&](int st) -> int {
                    const int tap = tap_of(st);
                    return (((tap / (SS::KW * SS::KH)) * SHH + (tap / SS::KW) % SS::KH) * SHW + tap % SS::KW) * VS;
                };
                int abP[RAG ? WPX : 1];
                if constexpr (RAG) {
                    const int pd = (l4 >= 2) ? (SS::KW > 1 ? VS : SHW * VS) - 32 : 0;
#pragma unroll
                    for (int pi = 0; pi < WPX; ++pi) abP[pi] = abase[pi] + pd;
                }
                auto ab_of = [&](int pi, int st) -> int { return paired(st) ? abP[RAG ? pi : 0] : abase[pi]; };
                h8_t afA[HA], afB[WPX - HA];
#pragma unroll
                for (int pi = 0; pi < HA; ++pi) afA[pi] = *(const h8_t*)(hb + ab_of(pi, 0) + toff_of(0));
                halo_static_for<NSC>([&](auto stc) {
                    constexpr int st = decltype(stc)::value;
                    {   // wait for this step's ring slot; the count: halo_ring_young (below the kernel)
#ifdef CS_ASMR_DEBUG0
                        constexpr int NY = 0;
#else
                        constexpr int NY = halo_ring_young<WCH, PFS, HI, ST0>(st, NSC, NS, true);
#endif
                        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(wr[st % PFS][0]), "+v"(wr[st % PFS][1]), "+v"(wr[st % PFS][2]), "+v"(wr[st % PFS][3]),
                                                             "+v"(wr[st % PFS][4]) : "n"(NY));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const half_t* rsrc_w = wsrc_of(cc, st - 1 + PFS, NSC);      // step st reloads the slot of step st - 1
                    const int toff = toff_of(st);
#pragma unroll
                    for (int k = 0; k < 2 * WCH; ++k) {                          // pairs of MFMAs on position fragments 0-3
                        const int ci = k >> 1, p0 = (k & 1) * 2;
                        acc[ci][p0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % PFS][ci]), afA[p0], acc[ci][p0], 0, 0, 0);
                        acc[ci][p0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % PFS][ci]), afA[p0 + 1], acc[ci][p0 + 1], 0, 0, 0);
                        if (k >= 2 && k < 6) afB[k - 2] = *(const h8_t*)(hb + ab_of(HA + k - 2, st) + toff);
                        if constexpr (st >= 1) { if (k >= 5) wload1(wr[(st - 1) % PFS][k - 5], rsrc_w, k - 5); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (st >= ST0 && st < ST0 + HI) { stage_piece(nbuf, cc + 1, st - ST0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                    for (int k = 0; k < 2 * WCH; ++k) {                          // ... on fragments 4-7, with the LDS reads of the next step's 0-3
                        const int ci = k >> 1, p0 = (k & 1) * 2;
                        acc[ci][HA + p0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % PFS][ci]), afB[p0], acc[ci][HA + p0], 0, 0, 0);
                        acc[ci][HA + p0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % PFS][ci]), afB[p0 + 1], acc[ci][HA + p0 + 1], 0, 0, 0);
                        if (k >= 2 && k < 6 && st + 1 < NSC) afA[k - 2] = *(const h8_t*)(hb + ab_of(k - 2, st + 1) + toff_of(st + 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                {   // the slot of the last step serves the next chunk
                    const half_t* rsrc_w = wsrc_of(cc, NSC - 1 + PFS, NSC);
#pragma unroll
                    for (int ci = 0; ci < WCH; ++ci) wload1(wr[(NSC - 1) % PFS][ci], rsrc_w, ci);
                }
            };
            const bool rag = RAGK && p.ragged && cc_hi == nck;
            for (int cc = cc_lo; cc < cc_hi - (rag ? 1 : 0); ++cc) run_chunk_a(cc, std::false_type{});
            if constexpr (RAGK) {
                if (rag) run_chunk_a(cc_hi - 1, std::true_type{});
            }
            staged = has_next;
            if (!has_next || !CARRY) {
                wait_vmcnt_le<0>();      // the ring's last (dropped) fetches: loads the compiler does not know of
                __builtin_amdgcn_sched_barrier(0);
            }
            // ... and whose results nobody reads: without a use BEHIND the drain hipcc treats their destination registers as free from the load
            // on - in the straight-line ragged chunk it gave them to LDS fragments and addresses, which the landing data then overwrote.  (With a
            // next tile the ring is live across the epilogue: its fetches are the next tile's first steps.)
#pragma unroll
            for (int st = 0; st < PFS; ++st)
                asm volatile("" :: "v"(wr[st][0]), "v"(wr[st][1]), "v"(wr[st][2]), "v"(wr[st][3]), "v"(wr[st][4]));
        } else {
        u4_t wr[PFS][WCH];
        // In this fully unrolled body hipcc counts vmcnt / lgkmcnt exactly (the only conservative drain sits at the chunk
        // loop's back-edge, next to the barrier), so the weight fragments are ordinary loads here.
        auto wload_at = [&](u4_t (&dst)[WCH], int cc, int st) {          // st: compile-time after unrolling
            const int ccl = cc < nck ? cc : nck - 1;
            const half_t* src = wlane + (long)((ccl * KH32 + st % KH32) * NT + st / KH32) * wstep;
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci) dst[ci] = *(const u4_t*)(src + ep_frag_row(EP_PAIR, ci) * 32);
        };
        TL_STAMP(1);
        stage_halo(0, cc_lo * CK);
        __syncthreads();
        TL_STAMP(2);
        if (DB && cc_lo + 1 < cc_hi) stage_halo(1, (cc_lo + 1) * CK);
        // head of a chunk: prime the weight ring, then wait for / re-issue the halo staging; returns the chunk's LDS buffer
        auto chunk_head = [&](int cc) -> const unsigned char* {
            // prime the ring with this chunk's first steps before waiting on the halo: both latencies overlap
            if (!WCARRY || cc == cc_lo) {
#pragma unroll
                for (int st = 0; st < PFS; ++st) wload_at(wr[st], cc, st);
            }
            if (cc > cc_lo && !(p.hilo && cc == 1)) {      // hilo: weight chunk 1 reuses the staged hi halo
                if (DB) {
                    __syncthreads();                       // chunk cc has landed in buffer (cc-cc_lo)&1; everyone left the other one
                    if (cc + 1 < cc_hi) stage_halo((cc + 1 - cc_lo) & 1, (cc + 1) * CK);
                } else {
                    __syncthreads();
                    stage_halo(0, (p.hilo ? 1 : cc) * CK);     // hilo: weight chunk 2 multiplies activation chunk 1 (lo)
                    __syncthreads();
                }
            }
            return smem + (size_t)(DB ? ((cc - cc_lo) & 1) : 0) * HV * VS;
        };
        // Software-pipelined operand fetch: the position fragments are split in two halves; while the MFMAs of one half
        // run, the ds_reads of the other half (of this step or of the next one) are in flight, so no LDS round trip is
        // exposed in steady state and no extra registers are needed.
        // RAG (the ragged last chunk of a layer with Cin % 32 == 16, ConvParams::ragged): the 16 real channels of two taps that
        // are neighbours along the row share one 32-deep MFMA - lanes l4 < 2 read tap t, lanes l4 >= 2 the same two slots of tap
        // t + 1 (their fragment address is shifted by one voxel / halo row minus two slots); the weights of this chunk were
        // re-packed the same way (pair_ragged_kernel).  The odd tap of a row runs as an ordinary step (its upper 16 channels are
        // the staged zeros).
        auto run_chunk = [&](int cc, const unsigned char* hb, auto rag_t) {
            constexpr bool RAG = decltype(rag_t)::value;
            constexpr int HA = WPX / 2;                    // fragments in the first half
            constexpr int PK = SS::KW > 1 ? SS::KW : SS::KH, SPR = PK / 2 + PK % 2;      // taps per row, steps per row
            constexpr int NSC = RAG ? (NT / PK) * SPR : NS;
            auto tap_of = [&](int st) -> int { return RAG ? (st / SPR) * PK + 2 * (st % SPR) : st / KH32; };
            auto paired = [&](int st) -> bool { return RAG && (st % SPR) < PK / 2; };
            auto toff_of = [&](int st) -> int {
                const int tap = tap_of(st), half = RAG ? 0 : st % KH32;
                return (((tap / (SS::KW * SS::KH)) * SHH + (tap / SS::KW) % SS::KH) * SHW + tap % SS::KW) * VS + half * 64;
            };
            // lane part of a paired fragment's address
            int abP[RAG ? WPX : 1];
            if constexpr (RAG) {
                const int pd = (l4 >= 2) ? (SS::KW > 1 ? VS : SHW * VS) - 32 : 0;
#pragma unroll
                for (int pi = 0; pi < WPX; ++pi) abP[pi] = abase[pi] + pd;
            }
            auto ab_of = [&](int pi, int st) -> int { return paired(st) ? abP[RAG ? pi : 0] : abase[pi]; };
            h8_t afA[HA > 0 ? HA : 1], afB[WPX - HA];
#pragma unroll
            for (int pi = 0; pi < HA; ++pi) afA[pi] = *(const h8_t*)(hb + ab_of(pi, 0) + toff_of(0));
            // Interleaved issue (one wave per SIMD kernels: the 256-position tiles of the hourglass tail, the mask conv and the first encoder
            // block): the LDS reads of the other half and the reload of the weight slot the PREVIOUS step used go out between the MFMAs of a
            // half instead of as a block ahead of them.  A block of 4 ds_reads + 5 global loads is 50 - 100 cycles of idle matrix pipe per
            // 20 MFMAs when no other wave shares the SIMD: tail 2.25 -> 2.15 ms, mask 3.46 -> 3.33 ms per 32 frames, +1 % on the step
            // (profiles/r03_n_ab_ilv.txt; -DCS_NO_ILV is the A/B switch, -DCS_ILV_ALL extends it to every static kernel)
#if defined(CS_NO_ILV)
            constexpr bool ILV = false;
#elif defined(CS_ILV_ALL)
            constexpr bool ILV = (WPX >= 2);
#elif defined(CS_ILV_128)
            constexpr bool ILV = (WPX == 8 && WVP == 2) || (WPX == 8 && WCH == 2 && WVP == 1);
#else
            constexpr bool ILV = (WPX == 8 && WVP == 2);
#endif
#pragma unroll
            for (int st = 0; st < NSC; ++st) {
                const int toff = toff_of(st);
#pragma unroll
                for (int pi = HA; pi < WPX; ++pi) afB[pi - HA] = *(const h8_t*)(hb + ab_of(pi, st) + toff);
                if (ILV && st >= 1) {
                    if (st - 1 + PFS < NSC) wload_at(wr[(st - 1) % PFS], cc, st - 1 + PFS);
                    else if (WCARRY && !RAG) wload_at(wr[(st - 1) % PFS], cc + 1, (st - 1) % PFS);
                }
                if (!ILV) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                    for (int pi = 0; pi < HA; ++pi)
                        acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % PFS][ci]), afA[pi],
                                                                             acc[ci][pi], 0, 0, 0);
                if (ILV) {
#pragma unroll
                    for (int i = 0; i < WPX - HA; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
                    for (int i = 0; i < WCH; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < NSC) {
#pragma unroll
                    for (int pi = 0; pi < HA; ++pi)
                        afA[pi] = *(const h8_t*)(hb + ab_of(pi, st + 1) + toff_of(st + 1));
                }
                if (!ILV) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                    for (int pi = HA; pi < WPX; ++pi)
                        acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % PFS][ci]), afB[pi - HA],
                                                                             acc[ci][pi], 0, 0, 0);
                if (ILV) {
#pragma unroll
                    for (int i = 0; i < HA; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!ILV) {
                    if (st + PFS < NSC) wload_at(wr[st % PFS], cc, st + PFS);
                    else if (WCARRY && !RAG) wload_at(wr[st % PFS], cc + 1, st % PFS);
                }
            }
            if (ILV && WCARRY && !RAG) wload_at(wr[(NSC - 1) % PFS], cc + 1, (NSC - 1) % PFS);
        };
        // the ragged chunk is peeled off the loop (inside it, the two bodies together cost the 160-wide kernels 500-650 bytes of scratch)
        const bool rag = RAGK && p.ragged && cc_hi == nck;
        for (int cc = cc_lo; cc < cc_hi - (rag ? 1 : 0); ++cc) { const unsigned char* hb = chunk_head(cc); run_chunk(cc, hb, std::false_type{}); }
        if constexpr (RAGK) {
            if (rag) { const unsigned char* hb = chunk_head(cc_hi - 1); run_chunk(cc_hi - 1, hb, std::true_type{}); }
        }
        }
    } else {
        // This wave's K-step sequence: for every chunk cc, "fine" steps j = j0, j0+SKS, ... < ntaps*nhalf(cc) with
        // tap = j / nhalf, half = j % nhalf; packed weight index kidx = (cc*KH32 + half)*ntaps + tap. Two scalar iterators
        // walk it: the producer (weight prefetch, PFD steps ahead) and the consumer.
        auto chunk_nhalf = [&](int cc) -> int {
            const int rem = p.nchunks - cc * KH32;
            return rem < KH32 ? rem : KH32;
        };
        struct It { int cc, nh, tap, half; };
        auto it_init = [&](It& it) {
            it.cc = cc_lo; it.nh = chunk_nhalf(cc_lo); it.tap = 0; it.half = j0;
            while (it.half >= it.nh) { it.half -= it.nh; ++it.tap; }
        };
        auto it_chunk_done = [&](const It& it) -> bool { return it.tap >= ntaps; };
        auto it_next_chunk = [&](It& it) {
            ++it.cc; it.nh = it.cc < cc_hi ? chunk_nhalf(it.cc) : 1; it.tap = 0; it.half = j0;
            while (it.half >= it.nh) { it.half -= it.nh; ++it.tap; }
        };
        auto it_advance = [&](It& it) {
            it.half += SKS;
            while (it.half >= it.nh) { it.half -= it.nh; ++it.tap; }
        };
        It P;                                // producer
        it_init(P);
        long last_off = 0;
        auto wload = [&](u4_t (&dst)[WCH]) {
            while (P.cc < cc_hi && it_chunk_done(P)) it_next_chunk(P);
            if (P.cc < cc_hi) { last_off = (long)((P.cc * KH32 + P.half) * ntaps + P.tap) * wstep; it_advance(P); }
            // past the end the last valid fragment is re-read (never consumed): every step issues exactly WCH loads, so the
            // counted wait below is exact in steady state and conservative otherwise
            const half_t* src = wlane + last_off;
            wfrag_load<0>(dst[0], src);
            if constexpr (WCH > 1) wfrag_load<ep_frag_row(EP_PAIR, 1) * 64>(dst[1], src);
            if constexpr (WCH > 2) wfrag_load<ep_frag_row(EP_PAIR, 2) * 64>(dst[2], src);
            if constexpr (WCH > 3) wfrag_load<ep_frag_row(EP_PAIR, 3) * 64>(dst[3], src);
            if constexpr (WCH > 4) wfrag_load<0>(dst[4], src + ep_frag_row(EP_PAIR, 4) * 32);
        };

        u4_t wr[PFD][WCH];
    #pragma unroll
        for (int i = 0; i < PFD; ++i) wload(wr[i]);

        TL_STAMP(1);
        stage_halo(0, cc_lo * CK);
        __syncthreads();
        TL_STAMP(2);
        if (DB && cc_lo + 1 < cc_hi) stage_halo(1, (cc_lo + 1) * CK);
        int cur = 0;
        It C;                                // consumer
        it_init(C);
        // (kd, kh, kw) of C.tap, kept incrementally
        int ckw = C.tap % KW, ckh = (C.tap / KW) % KH, ckd = C.tap / (KW * KH), ctap = C.tap;
        const unsigned char* hb = smem;
        bool done = false;
        while (!done) {
    #pragma unroll
            for (int i = 0; i < PFD; ++i) {
                while (!done && it_chunk_done(C)) {            // this wave finished its share of chunk C.cc
                    if (C.cc + 1 >= cc_hi) { done = true; break; }
                    if (DB) {
                        __syncthreads();                       // chunk cc+1 has landed in buffer cur^1; everyone left buffer cur
                        if (C.cc + 2 < cc_hi) stage_halo(cur, (C.cc + 2) * CK);
                        cur ^= 1;
                    } else {
                        __syncthreads();                       // everyone is done reading the single buffer
                        stage_halo(0, (C.cc + 1) * CK);
                        __syncthreads();
                    }
                    it_next_chunk(C);
                    ctap = C.tap; ckw = ctap % KW; ckh = (ctap / KW) % KH; ckd = ctap / (KW * KH);
                    hb = smem + (size_t)cur * HV * VS;
                }
                if (done) break;
                while (ctap < C.tap) { ++ctap; if (++ckw == KW) { ckw = 0; if (++ckh == KH) { ckh = 0; ++ckd; } } }
                const int toff = ((ckd * HH + ckh) * HW + ckw) * VS + C.half * 64;
                h8_t af[WPX];
    #pragma unroll
                for (int pi = 0; pi < WPX; ++pi) af[pi] = *(const h8_t*)(hb + abase[pi] + toff);
                wait_vmcnt_le<WCH*(PFD - 1)>();               // the WCH loads of this step's slot are older than the last WCH*(PFD-1)
                __builtin_amdgcn_sched_barrier(0);             // keep the MFMAs below the wait (hipcc would hoist them)
    #pragma unroll
                for (int ci = 0; ci < WCH; ++ci)
    #pragma unroll
                    for (int pi = 0; pi < WPX; ++pi)
                        acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[i][ci]), af[pi], acc[ci][pi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);             // ... and the reload of the slot below its last reader
                wload(wr[i]);
                it_advance(C);
            }
        }
    }

    if constexpr (ST == 0) {
        // The asm ring still has untracked loads in flight (re-reads issued by the last PFD steps): drain them before the
        // compiler reuses those VGPRs in the epilogue.
        wait_vmcnt_le<0>();
        __builtin_amdgcn_sched_barrier(0);
    }

    // sigmoid / GELU epilogues exist in: the pixel-shuffle kernel (conv_img), the 16-channel tiles (T's mask conv), the 1x1 kernels
    // (the motion extractor's linear layers) and the dynamic-shape fallbacks (small maps of the same layers)
    constexpr bool EP_HEAVY = (MODE == MODE_PIXSHUF) || (WCH == 1) || (ST == 15) || (ST == 0);
    constexpr bool KWSUM = !SK && MODE == MODE_STD && WCH == 5 && WPX == 8 && WVP == 2 && (ST == 8 || ST == 9);      // the kw-split mask conv (ConvParams::kw_out)
    TL_STAMP(3);
    if constexpr (!SK) {
        if (p.sk_out) {      // split-K: this workgroup's partial sums, fp32, [split][position][channel]; finished by splitk_finish_kernel
            const long mtot = (long)p.N * p.D * p.H * p.W;
#pragma unroll
            for (int pi = 0; pi < WPX; ++pi) {
                int m = wpx * WPX * 16 + pi * 16 + l15p;
                const int w = (tw << lgTW) + (m & mW); m >>= lgTW;
                const int h = (th << lgTH) + (m & mH); m >>= lgTH;
                const int d = (td << lgTD) + (m & mD); m >>= lgTD;
                const int n = tn * TN + m;
                if (n >= p.N) continue;
                const long pos = (((long)n * p.D + d) * p.H + h) * p.W + w;
#pragma unroll
                for (int ci = 0; ci < WCH; ++ci)
                    *(f4_t*)(p.sk_out + ((long)blockIdx.z * mtot + pos) * p.Cout_pad + ep_chan(EP_PAIR, 1, n0 + wch * WCH * 16, ci, l4)) = acc[ci][pi];
            }
        } else if (XSK && p.xs_w) {
            // ---- learned shortcut of a SPADEResnetBlock, fused (ConvParams::xs_w; util.py:329-344): the tile holds ALL 256 channels of
            // h = IN(x)(1 + gamma) for its 128 positions - wave wch the 64 channels wch * 64 ..., a lane 8 consecutive ones per fragment pair
            // (EP_PAIR 1), which is exactly the B operand of a 32-deep MFMA step (k = l4 * 8 ...).  So conv_s (1x1, 256 -> 64) runs on the
            // values in registers: per 16-position block 4 (output fragments) x 2 (32-channel groups) MFMAs per wave, the four waves' partial
            // sums meet in LDS (fixed order: wave 0 + 1 + 2 + 3), wave w adds conv_s(beta) (xs_res) to output fragment w and stores 16 x 16
            // fp16 values.  h never goes to HBM (2.1 GB per 64-frame launch at 256^2) and the conv_s launch disappears.
            if constexpr (XSK) {
                __syncthreads();                               // every wave has left the halo: the LDS is free
                float* part = (float*)smem;                    // [2 buffers][4 waves][4 fragments][64 lanes] float4
                const int cw0 = n0 + wch * 64;                 // this wave's first channel
                h8_t wsA[4][2];
#pragma unroll
                for (int of = 0; of < 4; ++of)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        wsA[of][q] = *(const h8_t*)(p.xs_w + ((long)((cw0 >> 5) + q) * 64 + of * 16 + l15) * 32 + l4 * 8);
                float gb[2][8], mu[2][8], rs[2][8];            // gamma bias, mean, rstd of the lane's 2 x 8 channels
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int cb = cw0 + q * 32 + l4 * 8;
#pragma unroll
                    for (int j = 0; j < 8; j += 4) {
                        const float4 b4 = p.bias ? *(const float4*)(p.bias + cb + j) : make_float4(0.f, 0.f, 0.f, 0.f);
                        gb[q][j] = b4.x; gb[q][j + 1] = b4.y; gb[q][j + 2] = b4.z; gb[q][j + 3] = b4.w;
                        const float4 q0 = *(const float4*)(p.stats + ((long)tn * p.Cout + cb + j) * 2);
                        const float4 q1 = *(const float4*)(p.stats + ((long)tn * p.Cout + cb + j) * 2 + 4);
                        mu[q][j] = q0.x; rs[q][j] = q0.y; mu[q][j + 1] = q0.z; rs[q][j + 1] = q0.w;
                        mu[q][j + 2] = q1.x; rs[q][j + 2] = q1.y; mu[q][j + 3] = q1.z; rs[q][j + 3] = q1.w;
                    }
                }
                const int rsh = p.res_shift;
                const half_t* xrow = (const half_t*)p.res.p + (long)tn * p.res.sN + cw0 + l4 * 8;
                const int ow = (tw << lgTW) + l15p, oh0 = th << lgTH;          // 16 x 8 tile: block pi is row oh0 + pi
                auto xfetch = [&](int pi, int q) -> u4_t {
                    return *(const u4_t*)(xrow + (long)((oh0 + pi) >> rsh) * p.res.sH + (long)(ow >> rsh) * p.res.sW + q * 32);
                };
                u4_t xr[2] = {xfetch(0, 0), xfetch(0, 1)};
                const int oc = wave * 16 + l4 * 4;             // reduction phase: this lane's 4 output channels (fragment = wave)
                const bool has_b = p.xs_res.p != nullptr && oc < p.xs_cout;
                auto bfetch = [&](int pi) -> h4_t {            // conv_s(beta) of this lane's outputs, fetched one block ahead like x
                    return *(const h4_t*)((const half_t*)p.xs_res.p + ((long)tn * p.xs_res.sN + (long)(oh0 + pi) * p.xs_res.sH + (long)ow * p.xs_res.sW + oc));
                };
                h4_t br = (h4_t){0, 0, 0, 0};
                if (has_b) br = bfetch(0);
#pragma unroll
                for (int pi = 0; pi < WPX; ++pi) {
                    u4_t xn[2] = {xr[0], xr[1]};
                    h4_t bn = br;
                    if (pi + 1 < WPX) { xn[0] = xfetch(pi + 1, 0); xn[1] = xfetch(pi + 1, 1); if (has_b) bn = bfetch(pi + 1); }
                    h8_t hB[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const h8_t xh = __builtin_bit_cast(h8_t, xr[q]);
                        unsigned pk[4];
#pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            float v2[2];
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int jj = j + e;
                                const float g = acc[2 * q + (jj >> 2)][pi][jj & 3] + gb[q][jj];
                                v2[e] = (((float)xh[jj] - mu[q][jj]) * rs[q][jj]) * (1.f + g);        // the spmul epilogue's arithmetic (conv_epilogue.h)
                            }
                            pk[j >> 1] = ep_pk(v2[0], v2[1]);
                        }
                        u4_t t4; t4[0] = pk[0]; t4[1] = pk[1]; t4[2] = pk[2]; t4[3] = pk[3];
                        hB[q] = __builtin_bit_cast(h8_t, t4);
                    }
                    float* pb = part + (size_t)(pi & 1) * (4 * 4 * 64 * 4);
#pragma unroll
                    for (int of = 0; of < 4; ++of) {
                        f4_t a = (f4_t){0.f, 0.f, 0.f, 0.f};
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wsA[of][0], hB[0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wsA[of][1], hB[1], a, 0, 0, 0);
                        *(f4_t*)(pb + ((wave * 4 + of) * 64 + lane) * 4) = a;
                    }
                    __syncthreads();                           // (two LDS buffers: the next block's writes do not wait for this block's reads)
                    f4_t sum = *(const f4_t*)(pb + ((0 * 4 + wave) * 64 + lane) * 4);
#pragma unroll
                    for (int w2 = 1; w2 < 4; ++w2) {
                        const f4_t t = *(const f4_t*)(pb + ((w2 * 4 + wave) * 64 + lane) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) sum[r] += t[r];
                    }
                    if (oc < p.xs_cout) {
                        const long po = (long)tn * p.xs_out.sN + (long)(oh0 + pi) * p.xs_out.sH + (long)ow * p.xs_out.sW + oc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) sum[r] += (float)br[r];
                        ep_u2_t o2; o2[0] = ep_pk(sum[0], sum[1]); o2[1] = ep_pk(sum[2], sum[3]);
                        *(ep_u2_t*)((half_t*)p.xs_out.p + po) = o2;
                    }
                    xr[0] = xn[0]; xr[1] = xn[1]; br = bn;
                }
            }
        } else if (KWSUM && p.kw_out) {
            // ---- mask conv: in-tile sum over kw (ConvParams::kw_out).  Tile 2 (w) x 8 (h) x 16 (d); a 16-position block of a wave is the
            // depth slice d = wpx * 8 + pi with l15p = (h << 1) | w'.  Per depth slice the two channel waves of a position half lay their
            // 160 channels into LDS ([16 positions][160] fp32), then its 128 threads add, for each of the 8 output columns j (w0 - 3 + j)
            // the tile touches, P[w' = 0][kw = 6 - j] + P[w' = 1][kw = 7 - j] and store 8 x 22 logits per (d, h) - one 704-byte run.
            if constexpr (KWSUM && ST == 9) {
                // Tile 4 (w) x 8 (h) x 8 (d): a 16-position block is 4 columns x 4 rows of one depth slice (l15p = (h & 3) << 2 | w'), block pi of wave
                // wpx = (depth slice wpx * 4 + pi / 2, row half pi & 1).  The tile's 4 columns reach the 10 output columns j = w0 - 3 + j; column j
                // sums P[w'][kw = w' + 6 - j] over the w' with 0 <= kw <= 6 (1 to 4 terms, w' ascending: a fixed order).  Per block: both channel
                // waves of the position half lay 16 x 160 floats into LDS, its 128 threads finish 4 rows x 10 x 22 logits = 220 float4 and store
                // them as one 880-byte run per (d, h): kw_out[((n D + d) H + h) (W / 4) + tile][j][22] - 10 vectors per 4 voxels instead of 8 per 2.
                __syncthreads();                               // every wave has left the halo: the LDS is free
                constexpr int ZERO = 16 * 160;
                float* buf = (float*)(smem + (ASMRK ? (size_t)(gbuf ^ 1) * (HI * 4096) : 0)) + wpx * (16 * 160 + 4);      // (ASMR: the buffer the last chunk read)
                const int th128 = wch * 64 + lane;
                if (th128 == 0) buf[ZERO] = 0.f;
                int offT[2][4][4]; int goff[2]; bool on[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int fi = q * 128 + th128;            // float4 index, 220 per block
                    on[q] = fi < 220;
                    const int fic = on[q] ? fi : 0;
                    goff[q] = (fic / 55) * (p.nTW * 220) + (fic % 55) * 4;      // row h' of the block (55 float4 per row), position inside the row's 220 floats
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = (fic % 55) * 4 + e, c = idx % 22, j = idx / 22, h = fic / 55;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {          // term t: column w' = t
                            const int kw = t + 6 - j;
                            offT[q][e][t] = (kw >= 0 && kw <= 6) ? ((h << 2) | t) * 160 + kw * 22 + c : ZERO;
                        }
                    }
                }
                const long gslice = (long)p.H * p.nTW * 220;  // one depth slice further
                float* const gbase = p.kw_out + ((((long)tn * p.D + (td << lgTD) + wpx * 4) * p.H + (th << lgTH)) * p.nTW + tw) * 220;
#pragma unroll
                for (int pi = 0; pi < WPX; ++pi) {
#pragma unroll
                    for (int ci = 0; ci < WCH; ++ci) *(f4_t*)(buf + l15p * 160 + wch * (WCH * 16) + ci * 16 + l4 * 4) = acc[ci][pi];
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (on[q]) {
                            f4_t v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = ((buf[offT[q][e][0]] + buf[offT[q][e][1]]) + buf[offT[q][e][2]]) + buf[offT[q][e][3]];
                            *(f4_t*)(gbase + (pi >> 1) * gslice + (long)(pi & 1) * 4 * (p.nTW * 220) + goff[q]) = v;
                        }
                    }
                    __syncthreads();
                }
            } else if constexpr (KWSUM) {
                __syncthreads();                               // every wave has left the halo: the LDS is free
                constexpr int ZERO = 16 * 160;                 // a zero float behind the image of a position half: the missing term at j = 0 / 7
                float* buf = (float*)(smem + (ASMRK ? (size_t)(gbuf ^ 1) * (HI * 4096) : 0)) + wpx * (16 * 160 + 4);      // (ASMR: the buffer the last chunk read)
                const int th128 = wch * 64 + lane;
                if (th128 == 0) buf[ZERO] = 0.f;
                // every thread finishes 3 float4 (12 logits) of the 8 (h) x 176 (j, c) block of a depth slice; the LDS offsets of their two
                // terms and the global offset do not depend on the slice
                int offA[12], offB[12], goff[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int fi = q * 128 + th128;            // float4 index, 352 per slice
                    goff[q] = (fi * 4 / 176) * (p.nTW * 176) + (fi * 4) % 176;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = fi * 4 + e, c = idx % 22, j = (idx / 22) & 7, h = (idx / 176) & 7;
                        offA[q * 4 + e] = j <= 6 ? (h << 1) * 160 + (6 - j) * 22 + c : ZERO;
                        offB[q * 4 + e] = j >= 1 ? ((h << 1) | 1) * 160 + (7 - j) * 22 + c : ZERO;
                    }
                }
                float* const gbase = p.kw_out + ((((long)tn * p.D + (td << lgTD) + wpx * WPX) * p.H + (th << lgTH)) * p.nTW + tw) * 176;
                const long gslice = (long)p.H * p.nTW * 176;  // one depth slice further
#pragma unroll
                for (int pi = 0; pi < WPX; ++pi) {             // unrolled: the accumulators are registers, no dynamic index
#pragma unroll
                    for (int ci = 0; ci < WCH; ++ci) *(f4_t*)(buf + l15p * 160 + wch * (WCH * 16) + ci * 16 + l4 * 4) = acc[ci][pi];
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        if (q * 128 + th128 < 352) {
                            f4_t v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = buf[offA[q * 4 + e]] + buf[offB[q * 4 + e]];
                            *(f4_t*)(gbase + pi * gslice + goff[q]) = v;
                        }
                    }
                    __syncthreads();
                }
            }
        } else {
        constexpr int EP_WPX = WPX;
        const int ep_wpx = wpx;
        auto& ep_acc = acc;
        CONV_EPILOGUE()
        }
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
    if (!ASMRK || !has_next) break;
    ++p_j;
    }       // tile loop (persistent mode)
#ifdef CS_TIMELINE
    TL_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TL_STAMP(5);
    const unsigned long long tl_real5 = __builtin_amdgcn_s_memrealtime();
    if (p.tl && lane == 0) {
        const long wi = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
        if (wi < p.tl_cap) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o = p.tl + wi * 12;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = tl_t[i];
            o[8] = tl_real0; o[9] = tl_real5; o[10] = hwid; o[11] = xcc;
        }
    }
#endif
}

static inline int lgS_of(const ConvParams& p) { return p.lgTW + p.lgTH + p.lgTD; }
template <int CK, int WPX, int WCH, int WVP, int WVC, int MODE, bool SK, int ST>
static int launch_halo_st(const ConvParams& p, hipStream_t st)
{
    constexpr int BM = SK ? WPX * 16 : WPX * 16 * WVP, BN = WCH * 16 * WVC;
    constexpr int PAD = halo_pad<CK, WPX, WCH, WVP, ST, MODE>(), HI = halo_hi<ST, PAD>();
    constexpr int SLP = CK / 8 + PAD, VS = SLP * 16;
    if (p.Cout_pad % BN != 0) { cs_set_error("conv_halo: Cout_pad %d not a multiple of the channel tile %d", p.Cout_pad, BN); return -1; }
    {   // in-tensor element offsets are kept in 32 bits inside the kernel, per-axis products in 24 (halo piece offsets; the epilogue's
        // tensors: ep_check_extents)
        const long in_span = (long)(p.N - 1) * p.in_sN + (long)(p.D - 1) * p.in_sD + (long)(p.H - 1) * p.in_sH + (long)(p.W - 1) * p.in_sW +
                             p.Cin + (p.cg > 0 ? (long)(p.nchunks / p.cg) * p.in_sG : 0);
        if (in_span >= (1L << 31)) { cs_set_error("conv_halo: the input spans 2^31 elements or more (32-bit in-tensor offsets)"); return -1; }
        if (p.in_sD >= (1L << 23) || p.in_sH >= (1L << 23) || p.in_sW >= (1L << 23)) {
            cs_set_error("conv_halo: an input axis stride of 2^23 elements or more (24-bit multiplies in the addressing)");
            return -1;
        }
        if (ep_check_extents(p, "conv_halo")) return -1;
    }
    constexpr bool heavy_ok = (MODE == MODE_PIXSHUF) || (WCH == 1) || (ST == 15) || (ST == 0);
    if ((!heavy_ok && p.act0 >= ACT_SIGMOID) || p.act1 >= ACT_SIGMOID) {
        cs_set_error("conv_halo: this tile configuration carries no sigmoid / GELU epilogue (act0 %d, act1 %d)", p.act0, p.act1);
        return -1;
    }
    if (p.inD != p.D) { cs_set_error("conv_halo: depth-collapsing convs are not supported"); return -1; }
    if (p.pool_hw && !p.sk_out) {      // (a split-K launch leaves raw partial sums: its finishing launch pools)
        using SSL = StaticShape<ST>;
        constexpr bool poolk = ST != 0 && !SK && MODE == MODE_STD && WCH == 2 && SSL::KD == 3 && SSL::KH == 3 && SSL::KW == 3 && (SSL::LW == 3 || SSL::LW == 2) && SSL::LH >= 1;
        constexpr bool poolk2 = halo_pool2d<CK, WPX, WCH, WVP, MODE, SK, ST>();
        if (!(poolk || poolk2) || p.res.p || p.out1.p || p.pixscale || p.stat_out || p.out0_f32 || !p.out0.p || p.sk_out || p.ep_general || (p.H & 1) || (p.W & 1) ||
            p.Cout % 8 || p.Cout != p.Cout_pad || (1 << lgS_of(p)) != BM || ((unsigned long long)p.out0.p & 15ull) ||
            ((p.out0.sN | p.out0.sD | p.out0.sH | p.out0.sW) & 7)) {
            cs_set_error("conv_halo: pool_hw (AvgPool(1,2,2) in the epilogue) needs a static 3x3x3 tile of 8 or 4 columns with two channel fragments per wave or a 2-D 16x8 tile with 32-channel chunks, fp16 out0 only, every packed channel real");
            return -1;
        }
    }
    if (p.spmul) {
        if (MODE != MODE_STD || SK || !p.res.p || p.res_f32 || !p.stats || p.out1.p || p.pixscale || p.stat_out || p.pool_hw || p.sk_out || p.kw_out ||
            (1 << lgS_of(p)) != BM) {
            cs_set_error("conv_halo: spmul (out0 = act0(IN(res) (1 + conv))) is a mode-STD epilogue with an fp16 res, stats, one output and tiles within one sample");
            return -1;
        }
    }
    if (p.xs_w) {
        constexpr bool xsk = ST == 1 && !SK && MODE == MODE_STD && CK == 64 && WCH == 4 && WPX == 8 && WVP == 1;
        if (!xsk || !p.spmul || p.out0.p || p.Cout != 256 || p.Cout_pad != 256 || p.xs_cout < 4 || p.xs_cout > 64 || p.xs_cout % 4 || !p.xs_out.p || p.act0 != ACT_NONE ||
            p.lgTW != 4 || p.lgTH != 3 || ((unsigned long long)p.res.p & 15ull) || ((p.res.sN | p.res.sH | p.res.sW) & 7) || ((unsigned long long)p.xs_w & 15ull) ||
            ((unsigned long long)p.xs_out.p & 7ull) || ((p.xs_out.sN | p.xs_out.sH | p.xs_out.sW) & 3) ||
            (p.xs_res.p && (((unsigned long long)p.xs_res.p & 7ull) || ((p.xs_res.sN | p.xs_res.sH | p.xs_res.sW) & 3)))) {
            cs_set_error("conv_halo: xs_w (conv_s inside the spmul epilogue) needs the 128 x 256 tile kernel, 256 modulated channels, no out0, at most 64 shortcut channels");
            return -1;
        }
    }
    if (p.kw_out) {
        constexpr bool kwsum = !SK && MODE == MODE_STD && WCH == 5 && WPX == 8 && WVP == 2 && (ST == 8 || ST == 9);
        if (!kwsum || p.Cout_pad != 160 || p.lgTH != 3 || !((p.lgTW == 1 && p.lgTD == 4) || (p.lgTW == 2 && p.lgTD == 3)) || p.sk_out || p.W % (1 << p.lgTW) ||
            p.nTN != p.N) {
            cs_set_error("conv_halo: kw_out is the mask conv's epilogue (7x7x1 taps, 160 packed channels, 2x8x16 or 4x8x8 tiles of one sample)");
            return -1;
        }
    }
    const int lgS = p.lgTW + p.lgTH + p.lgTD;
    if ((1 << lgS) > BM) { cs_set_error("conv_halo: spatial tile exceeds BM"); return -1; }
    if (MODE == MODE_SPADE && (1 << lgS) != BM) { cs_set_error("conv_halo: SPADE launches must tile within one sample"); return -1; }
    if (MODE == MODE_SPADE && p.res_f32) { cs_set_error("conv_halo: the tensor a SPADE launch modulates is fp16"); return -1; }
    if (p.ragged) {
        constexpr bool ragk = (CK == 32) && (WPX == 8) && (ST == 7 || ST == 8 || ST == 9) && MODE == MODE_STD;
        if (!ragk || SK || p.Cin % 32 != 16 || p.sk_out || p.hilo || p.cg > 0) {
            cs_set_error("conv_halo: paired-tap weights (ragged) need the 256-position 32-channel static kernels and Cin %% 32 == 16");
            return -1;
        }
    }
    if (p.wslot && (1 << lgS) != BM) { cs_set_error("conv_halo: per-sample weight sets need tiles within one sample"); return -1; }
    if (p.cg > 0 && CK != 32 && (!(ST == 15 || (ST == 0 && WCH != 4)) || (p.cg & 1) || (p.in_sG & 63) || p.KD * p.KH * p.KW != 1)) {
        cs_set_error("conv_halo: grouped input channels run 32-channel chunks (64-channel chunks: 1x1 convs with an even number of 32-channel pieces per group)");
        return -1;
    }
    const int TN = BM >> lgS;
    const long HV = (long)TN * ((1 << p.lgTD) + p.KD - 1) * ((1 << p.lgTH) + p.KH - 1) * ((1 << p.lgTW) + p.KW - 1);
    const int nck = (p.Cin + CK - 1) / CK;
    // double-buffered halos: up to 64 KB per workgroup (two or three workgroups stay resident); the 256x160 tiles run one workgroup
    // per CU and take what they need
    constexpr bool big = (BM == 256 && WCH == 5);
    // (double-buffering the three 52 KB chunks of the split-precision volume convs - one workgroup per CU instead of three single-
    // buffered ones - measured 3 % slower on the whole step: profiles/r02_notes.md)
#ifndef CS_NO_ASMRING
    constexpr bool asmr = WCH == 5 && WPX == 8 && WVP == 2 && (ST == 7 || ST == 8 || ST == 9) && MODE == MODE_STD && !SK;      // see the kernel (ASMR)
#else
    constexpr bool asmr = false;
#endif
    // (the ASMR kernels' two buffers are HI DMA pieces each: the 4 x 8 x 8 mask tile's 2 x 76 KB fit the 160 KB of a CU)
    const bool db = nck > 1 && (asmr ? 2 * HI * 4096 <= 160 * 1024 : 2 * HV * VS <= (big || (WCH == 4 && PAD == 2) ? 128 : 64) * 1024) && HV * SLP <= 256 * HI;
    size_t lds = (size_t)(db ? 2 : 1) * HV * VS + 16;
    ConvParams kp = p;
    if (asmr && db) {
        lds = (size_t)2 * HI * 4096;          // buffer stride of a whole number of DMA pieces
        const long span = (long)(p.inD - 1) * p.in_sD + (long)((p.H - 1) >> p.up_shift) * p.in_sH + (long)((p.W - 1) >> p.up_shift) * p.in_sW + p.Cin + 32;
        if (span >= (1L << 30) || p.cg > 0 || p.hilo || (1 << lgS) != BM) {
            cs_set_error("conv_halo: the 256 x 160 tiles address one sample of the input through a buffer of less than 2^31 bytes (no grouped / split-precision input)");
            return -1;
        }
        kp.in_sample_bytes = (unsigned)(span * 2);
    }
    kp.persist_total = 0;
    if (SK && lds < (size_t)4 * WCH * WPX * 4 * 64 * sizeof(float)) lds = (size_t)4 * WCH * WPX * 4 * 64 * sizeof(float);
    if (p.xs_w && lds < (size_t)32768) lds = 32768;        // the fused-shortcut epilogue's part[2][4][4][64] float4 lives in the halo LDS
    if (lds > 160 * 1024) { cs_set_error("conv_halo: halo of %ld voxels does not fit LDS", HV); return -1; }
    dim3 grid((unsigned)(p.nTW * p.nTH * p.nTD * p.nTN), (unsigned)(p.Cout_pad / BN));
    if (p.hilo && (ST == 0 || CK != 32 || nck != 3 || db || p.sk_out)) {
        cs_set_error("conv_halo: the hi/lo split-precision mode needs a static-shape 32-channel kernel with three single-buffered chunks");
        return -1;
    }
    if (p.nphase) {
        if (SK || !halo_phase_group<WCH, ST>() || p.nphase < 1 || p.nphase > 4 || p.sk_out || p.kw_out || p.xs_w || p.persist_total < 0) { cs_set_error("conv_halo: bad grouped launch (%d phases)", p.nphase); return -1; }
        // the phase's output offset is applied to out0 and out1: every other tensor of the epilogue would be read / written at the same addresses by all phases
        if (p.res.p || p.stat_out || p.pixscale || p.spmul || p.pool_hw) { cs_set_error("conv_halo: a grouped launch carries out0 / out1 only (no res / statistics / pixel scale / spmul / pooling)"); return -1; }
        for (int z = 0; z < p.nphase; ++z)
            if (p.ph_ooff[z] % 8u) { cs_set_error("conv_halo: output offset of phase %d (%u elements) is not a multiple of 8 (16-byte stores)", z, p.ph_ooff[z]); return -1; }
        grid.z = (unsigned)p.nphase;
    }
    if (p.sk_out) {
        if (SK || p.sk_splits < 1 || p.sk_splits > nck || p.xcd_map == 2) { cs_set_error("conv_halo: bad split-K launch (%d splits, %d chunks)", p.sk_splits, nck); return -1; }
        grid.z = (unsigned)p.sk_splits;
    }
    if (p.xcd_map == 2) grid = dim3(grid.x * grid.y, 1, grid.z);
    hipError_t e;
    {   // exact division of the workgroup index by the tile counts as one multiply-high each (u / d == umulhi(u, 2^32 / d + 1) while
        // u * d < 2^32; 0 encodes d == 1): four runtime integer divisions cost every workgroup about a hundred issue slots
        const unsigned long long umax = (unsigned long long)grid.x * grid.y;
        const unsigned ds[4] = {(unsigned)(p.Cout_pad / BN), (unsigned)p.nTW, (unsigned)p.nTH, (unsigned)p.nTD};
        unsigned mg[4];
        for (int i = 0; i < 4; ++i) {
            if (ds[i] == 0 || umax * ds[i] >= (1ull << 32)) { cs_set_error("conv_halo: launch of %llu workgroups / tile count %u too large", umax, ds[i]); return -1; }
            mg[i] = ds[i] == 1 ? 0u : (unsigned)((1ull << 32) / ds[i]) + 1u;
        }
        kp.mg_ncb = mg[0]; kp.mg_tw = mg[1]; kp.mg_th = mg[2]; kp.mg_td = mg[3];
    }
#ifdef CS_TIMELINE
    kp.tl = g_cs_tl; kp.tl_cap = g_cs_tl_cap;
#endif
    if (asmr && db && p.persist_total != 0 && !p.sk_out && !p.nphase) {          // persistent: one workgroup per CU (LDS: one resident), XCD x walks entries [x * total / 8 ...)
        const unsigned total = grid.x * grid.y;
        kp.persist_total = (int)total;
        grid = dim3(total < 256u ? (total + 7u) & ~7u : 256u, 1);
    }
    if (db) {
        auto k = conv_halo_kernel<CK, WPX, WCH, WVP, WVC, MODE, true, SK, ST>;
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { cs_set_error("conv_halo: opting into %zu bytes of LDS failed: %s", lds, hipGetErrorString(e)); return -1; }
        }
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, kp);
    } else {
        auto k = conv_halo_kernel<CK, WPX, WCH, WVP, WVC, MODE, false, SK, ST>;
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { cs_set_error("conv_halo: opting into %zu bytes of LDS failed: %s", lds, hipGetErrorString(e)); return -1; }
        }
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, kp);
    }
    e = hipGetLastError();
    if (e != hipSuccess) { cs_set_error("conv_halo launch: %s", hipGetErrorString(e)); return -1; }
    return 0;
}

