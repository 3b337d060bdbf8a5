// Persistent 3x3 convolution for the wide (>= 256 packed output channels) 2-D layers of the path, gfx950 (CDNA4).
//
// Replaces conv_halo's 128 x 256 tile (conv_halo_kernel.h) for: the 14 T blend convs (adaptive_modulate.py:128-193: 512 -> 2 x 512,
// shared + identity-modulated weight on one activation tile, blended in the epilogue), and - as further instantiations - the 3x3 convs
// 512 -> 512 of G / R and the SPADE gamma / beta convs (util.py:282-344, 105-128).  VERDICT r3 item 1.  What changes against conv_halo:
//
//   * ONE WORKGROUP PER CU, ONE WAVE PER SIMD, 8 x 8 ACCUMULATOR FRAGMENTS PER WAVE.  A workgroup owns 256 positions (16 x 16) x 256 packed
//     channels; wave (wp, wc) computes 128 positions (8 rows of 16) x 128 channels = 64 MFMA fragments (256 accumulator registers).  Per
//     32-deep K-step a wave issues 8 ds_read_b128 (activation fragments) + 8 global_load_dwordx4 (weight fragments) for 64 MFMAs: half the
//     LDS reads per MFMA of the 128 x 256 tile (8 + 4 per 32), and a workgroup's unique weight bytes per MFMA are halved (the two position
//     halves read the same fragments: the second read hits the CU's L1).  tools/mfma_power2.hip prices this operand mix at 0.625 of peak
//     against 0.58 for the 8 x 4 wave.
//   * PERSISTENT, CROSS-TILE PIPELINED.  256 workgroups walk their own list of (tile, channel block) items.  The halo of item k + 1's first
//     channel chunk is DMA-staged into the free LDS buffer while item k's last chunk computes, its addressing is done under those MFMAs,
//     and the stores of item k's epilogue drain under item k + 1's main loop (the chunk barriers wait with a counted vmcnt that leaves the
//     stores in flight).  With one workgroup per CU a non-persistent kernel would expose prologue + first halo wait + epilogue store tail
//     per tile (the round-2 experiment with a 256 x 256 tile measured +-0 for that reason).
//   * XCD-AWARE ITEM ORDER.  Hardware places workgroup b on XCD b % 8.  Every XCD owns a contiguous range of tiles; inside an XCD the
//     workgroups form groups of `ncb` (one per 256-channel block of the layer) that work on the SAME tile at the same time, so the halo of
//     a tile is fetched into that XCD's L2 once for all its channel blocks, and concurrently processed tiles are neighbours (their halos
//     overlap).  conv_halo re-fetched a tile per channel block at different times (FETCH 5x the input).
//
// K order per output element: 64-channel chunk -> tap (kh, kw) -> 32-channel half, one v_mfma_f32_16x16x32_f16 each - exactly
// conv_halo_kernel's order for CK = 64, and the epilogue is the same macro (conv_epilogue.h), so both kernels give the same bits
// (tests/test_gpu_wide.py compares them with torch.equal).
//
// LDS image of a halo chunk: [18 x 18 voxels][8 data slots + 2 pad slots] x 16 bytes (voxel stride 160 B: conflict-free fragment reads for
// 16 consecutive voxels at any alignment, tools/lds_bank_search.py), two buffers of 51 840 B.
#define EP_GSEL_V EP_GSEL_K      /* position blocks whose residual / mask the epilogue fetches per round: set per instantiation below */
#define EP_SLICE_FENCE __builtin_amdgcn_sched_barrier(0);
#include "conv_epilogue.h"

namespace {

constexpr int W_SLP = 10, W_VS = W_SLP * 16;          // slots per voxel (8 data + 2 pad), bytes per voxel
constexpr int W_HW = 18, W_HV = W_HW * W_HW;          // halo of a 16 x 16 tile
constexpr int W_BUF = W_HV * W_VS;                    // 51 840 bytes per buffer
constexpr int W_NPIECE = W_HV * W_SLP;                // 3240 16-byte pieces per chunk
constexpr int W_HI = (W_NPIECE + 255) / 256;          // 13 pieces per thread
constexpr int W_PFS = 3;                              // weight ring depth in K-steps (a K-step is 64 MFMAs ~ 1 000 cycles)
constexpr int W_NT = 9, W_NS = 18;                    // taps, K-steps per 64-channel chunk

typedef unsigned int u4w_t __attribute__((ext_vector_type(4)));

// The epilogue reads the accumulators through this proxy: a VALU instruction cannot take an AGPR source, and left to itself the register
// allocator splits the live range of all 256 accumulators at the loop exit (256 v_accvgpr_read into VGPRs at once, which then spill).  An
// explicit read with an AGPR-constrained operand keeps every accumulator where the MFMAs left it until the element is needed.
struct WideAcc3 { const f4_t& v; __device__ __forceinline__ float operator[](int r) const { float x; asm("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(v[r])); return x; } };
struct WideAcc2 { const f4_t (&row)[8]; __device__ __forceinline__ WideAcc3 operator[](int pi) const { return WideAcc3{row[pi]}; } };
struct WideAcc1 { const f4_t (&a)[8][8]; __device__ __forceinline__ WideAcc2 operator[](int ci) const { return WideAcc2{a[ci]}; } };

template <int N> __device__ __forceinline__ void wide_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// schedule of the persistent grid (set by the launcher)
struct WideSched {
    int ntiles;        // N * (H / 16) * (W / 16)
    int nTW, nTH;      // tiles per row / column of a sample
    int ncb;           // 256-channel blocks of the layer (1, 2, 4)
    int tg;            // tile groups per XCD = (G / 8) / ncb
    int t8;            // tiles per XCD (contiguous range)
};

// EPC: the tensor combination of the launch as a compile-time constant (EP_CODE of conv_epilogue.h): straight-line epilogue, counted waits
template <int MODE, int EPC>
__global__ void __launch_bounds__(256, 1) conv_wide_kernel(const ConvParams p, const WideSched s)
{
    constexpr int WCH = 8, WPX = 8, BM = 256;
    constexpr int EP_PAIR = ep_pair_of(MODE, WCH);
    constexpr int CSTEP_W = (MODE == MODE_TBLEND || MODE == MODE_SPADE) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wch = wave >> 1;        // position half (rows wp * 8 ..), channel half (packed rows wch * 128 ..)
    const int l15 = lane & 15, l4 = lane >> 4;
    const int l15p = l15;

    // ---- this workgroup's item list: channel block cb (fixed), tiles xcd * t8 + tgi + tg * j
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int cblk = slot % s.ncb, tgi = slot / s.ncb;
    const int n0 = cblk * 256;
    auto tile_of = [&](int j) -> int {              // -1: no such item
        const int r = tgi + s.tg * j;
        const int t = xcd * s.t8 + r;
        return (r < s.t8 && t < s.ntiles) ? t : -1;
    };

    const int isN = (int)p.in_sN, isH = (int)p.in_sH, isW = (int)p.in_sW;
    // ---- halo staging: piece q = tid + 256 * j of a chunk <-> (voxel q / 10, slot q % 10); global -> LDS directly, pad slots are not fetched
    int poff[W_HI];
    unsigned pmask = 0;                              // bit j: piece j lies inside the image (else it reads the zero page)
    unsigned pdata = 0;                              // bit j: piece j is a data slot of an existing halo voxel (else not issued at all)
#pragma unroll
    for (int j = 0; j < W_HI; ++j) {
        const int q = tid + 256 * j;
        if (q < W_NPIECE && (q % W_SLP) < 8) pdata |= 1u << j;
    }
    auto setup_item = [&](int tile) {
        int t = tile;
        const int tw = t % s.nTW; t /= s.nTW;
        const int th = t % s.nTH; t /= s.nTH;
        const int base = t * isN;
        pmask = 0;
        // (the thread index goes through an opaque move: otherwise the item-invariant part of this addressing - 3 values per piece - is hoisted
        // out of the item loop and held in registers across the main loop, and the kernel spills)
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int j = 0; j < W_HI; ++j) {
            const int q = tid_o + 256 * j;
            const int hv = q / W_SLP, sl = q % W_SLP;
            const int hh = hv / W_HW, hw = hv % W_HW;
            const int ih = th * 16 + hh - 1, iw = tw * 16 + hw - 1;
            const bool inb = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            poff[j] = inb ? base + __mul24(ih, isH) + __mul24(iw, isW) + sl * 8 : 0;
            pmask |= inb ? (1u << j) : 0u;
        }
    };
    auto stage = [&](int buf, int c0) {              // asynchronous (vmcnt): awaited by the counted wait + barrier at the chunk's head
#pragma unroll
        for (int j = 0; j < W_HI; ++j) {
            if ((pdata >> j) & 1u) {
                const half_t* src = ((pmask >> j) & 1u) ? p.in + poff[j] + c0 : p.zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem + (size_t)buf * W_BUF + (size_t)(256 * j + wave * 64) * 16),
                                                 16, 0, 0);
            }
        }
    };

    // ---- operand addressing.  Activation fragment pi of this wave = row wp * 8 + pi of the tile, 16 positions along w (l15), k slot l4;
    // its LDS byte offset in the un-shifted window is abase0 + pi * 18 * 160: every fragment address of the loop is base + immediate.
    const int abase0 = ((wp * 8) * W_HW + l15p) * W_VS + l4 * 16;
    const int nck = p.Cin / 64;
    const long wstep = (long)p.Cout_pad * 32;
    // weights: fragment ci of K-step kidx = 1 KiB at wgt + (kidx * Cout_pad + n0 + wch * 128 + ci * 16) * 32, rows permuted per EP_PAIR
    const long wlane_off = (long)(n0 + wch * WCH * 16) * 32 + ep_lane_row(EP_PAIR, l15) * 32 + l4 * 8;

    int j_item = 0;
    int tile = tile_of(0);
    if (tile < 0) return;
    setup_item(tile);
    stage(0, 0);
    int gbuf = 0;                                    // buffer of the chunk about to be computed

    while (tile >= 0) {
        // coordinates of the item being computed (the staging registers move on to the next item during its last chunk)
        const int tile_lin = tile;
        int tdec = tile;
        const int tw = tdec % s.nTW; tdec /= s.nTW;
        const int th = tdec % s.nTH; tdec /= s.nTH;
        const int tn = tdec, td = 0;
        const int next_tile = tile_of(j_item + 1);
        const half_t* wlane = p.wgt + wlane_off;
        if (p.wslot) wlane += p.wofs[p.wslot[tn]];   // per-sample weight set (element offset from the kernel argument: stays global_load)

        f4_t acc[WCH][WPX];
#pragma unroll
        for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
            for (int pi = 0; pi < WPX; ++pi) acc[ci][pi] = (f4_t){0.f, 0.f, 0.f, 0.f};

        u4w_t wr[W_PFS][WCH];
        auto wload_at = [&](u4w_t (&dst)[WCH], int cc, int st) {          // st: compile-time after unrolling
            const int ccl = cc < nck ? cc : nck - 1;                      // behind the last chunk the carried fetches repeat and are dropped
            const half_t* src = wlane + (long)((ccl * 2 + st % 2) * W_NT + st / 2) * wstep;
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci) dst[ci] = *(const u4w_t*)(src + ep_frag_row(EP_PAIR, ci) * 32);
        };
#pragma unroll
        for (int st = 0; st < W_PFS; ++st) wload_at(wr[st], 0, st);

        for (int cc = 0; cc < nck; ++cc) {
            // ---- head of a chunk: its halo (staged one chunk ago by every wave) has landed: this wave's pieces by the counted wait - the
            // ring fetches (and, at an item's first chunk, the previous epilogue's stores) are younger and stay in flight -, everyone's by
            // the barrier, which also says that everyone has left the other buffer.  Then the next chunk (or the next item's first one,
            // whose addressing is computed here, under the MFMAs that follow) goes into that buffer.
            wide_wait_vm<W_PFS * WCH>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (cc + 1 < nck) stage(gbuf ^ 1, (cc + 1) * 64);
            const unsigned char* hb = smem + (size_t)gbuf * W_BUF + abase0;
            gbuf ^= 1;

            // ---- 18 K-steps, fully unrolled.  Position fragments in two halves: while the MFMAs of one half run, the LDS reads of the
            // other half (of this step or the next) and the reload of the ring slot the previous step used go out between them.
            constexpr int HA = WPX / 2;
            auto toff_of = [&](int st) -> int {
                const int tap = st / 2, half = st % 2;
                return ((tap / 3) * W_HW + tap % 3) * W_VS + half * 64;
            };
            h8_t afA[HA], afB[WPX - HA];
#pragma unroll
            for (int pi = 0; pi < HA; ++pi) afA[pi] = *(const h8_t*)(hb + pi * (W_HW * W_VS) + toff_of(0));
#pragma unroll
            for (int st = 0; st < W_NS; ++st) {
                const int toff = toff_of(st);
#pragma unroll
                for (int pi = HA; pi < WPX; ++pi) afB[pi - HA] = *(const h8_t*)(hb + pi * (W_HW * W_VS) + toff);
                if (st >= 1) {
                    if (st - 1 + W_PFS < W_NS) wload_at(wr[(st - 1) % W_PFS], cc, st - 1 + W_PFS);
                    else wload_at(wr[(st - 1) % W_PFS], cc + 1, (st - 1 + W_PFS) - W_NS);
                }
#pragma unroll
                for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                    for (int pi = 0; pi < HA; ++pi)
                        acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % W_PFS][ci]), afA[pi], acc[ci][pi], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < WPX - HA; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
                for (int i = 0; i < WCH; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < W_NS) {
#pragma unroll
                    for (int pi = 0; pi < HA; ++pi) afA[pi] = *(const h8_t*)(hb + pi * (W_HW * W_VS) + toff_of(st + 1));
                }
#pragma unroll
                for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                    for (int pi = HA; pi < WPX; ++pi)
                        acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % W_PFS][ci]), afB[pi - HA], acc[ci][pi], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < HA; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_barrier(0);
            }
            // W_NS % W_PFS == 0: step W_NS - 1 used slot (W_NS - 1) % W_PFS, which serves step W_PFS - 1 of the next chunk
            wload_at(wr[(W_NS - 1) % W_PFS], cc + 1, W_PFS - 1);
        }

        // ---- the next item's first chunk goes into the buffer the last chunk did not read (everyone left it before that chunk's barrier): it
        // lands while the epilogue runs
        if (next_tile >= 0) { setup_item(next_tile); stage(gbuf, 0); }
        // ---- epilogue (conv_epilogue.h): the tensor combination is a compile-time constant.  Stores are not waited for here.
        {
            constexpr bool EP_HEAVY = false, EP_EARLY = false;
            constexpr int EP_GSEL_K = 8;      // every fetch of the wave's tile before its first store (2 / 4 blocks per round spill: hipcc carries more addressing then)
            constexpr int EP_WPX = WPX;
            constexpr int lgTW = 4, lgTH = 4, lgTD = 0, lgS = 8, mW = 15, mH = 15, mD = 0;
            const int ep_wpx = wp;
            ep_u2_t ep_xpre[1][1];
            (void)ep_xpre; (void)td; (void)CSTEP_W;
            const WideAcc1 ep_acc{acc};
            CONV_EPILOGUE_IMPL(EPC);
        }
        tile = next_tile;
        ++j_item;
    }
}

template <int MODE, int EPC>
int launch_wide_inst(const ConvParams& p, const WideSched& s, int grid, hipStream_t st)
{
    auto k = conv_wide_kernel<MODE, EPC>;
    const size_t lds = 2 * (size_t)W_BUF;
    static bool attr_done = false;                   // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { cs_set_error("conv_wide: opting into %zu bytes of LDS failed: %s", lds, hipGetErrorString(e)); return -1; }
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, p, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cs_set_error("conv_wide launch: %s", hipGetErrorString(e)); return -1; }
    return 0;
}

int wide_ep_code(const ConvParams& p)
{
    return EP_CODE(p.res.p ? (p.res_f32 ? 2 : 1) : 0, p.out0.p ? 1 : 0, p.out0_f32 ? 1 : 0, p.out1.p ? 1 : 0, p.pixscale ? 1 : 0);
}

}  // namespace

// Which launches this kernel takes: 3x3, 2-D, 16 x 16 tiles within a sample, Cin a multiple of 64, every packed channel a real one, the
// tensor combinations compiled below; enough tiles to give every workgroup of the persistent grid the same number of items.
bool conv_wide_supported(const ConvParams& p, int mode)
{
    if (p.KD != 1 || p.KH != 3 || p.KW != 3 || p.D != 1 || p.inD != 1 || p.up_shift || p.cg || p.hilo || p.ragged || p.sk_out || p.kw_out || p.xf_kind) return false;
    if (p.H % 16 || p.W % 16 || p.Cin % 64 || p.Cout_pad % 256 || p.Cout_pad > 1024) return false;
    if (p.stat_out || p.ep_general) return false;
    const int cstep = (mode == MODE_TBLEND || mode == MODE_SPADE) ? 2 : 1;
    if (p.Cout * cstep != p.Cout_pad || (p.Cout & 7)) return false;
    const int ncb = p.Cout_pad / 256;
    if (ncb != 1 && ncb != 2 && ncb != 4) return false;
    if (p.act0 >= ACT_SIGMOID || p.act1 >= ACT_SIGMOID) return false;
    auto al8 = [](const TDesc& t) { return (((unsigned long long)t.p & 15ull) == 0) && (((t.sN | t.sD | t.sH | t.sW) & 7) == 0); };
    if ((p.res.p && !p.res_f32 && !al8(p.res)) || (p.out0.p && !p.out0_f32 && !al8(p.out0)) || (p.out1.p && !al8(p.out1))) return false;
    if (((unsigned long long)p.in & 15ull) || ((p.in_sN | p.in_sH | p.in_sW) & 7)) return false;
    const int code = wide_ep_code(p);
    if (mode == MODE_TBLEND) return code == EP_CODE(0, 1, 0, 0, 1) || code == EP_CODE(2, 1, 1, 1, 1);
    return false;
}

int launch_conv_wide(const ConvParams& p, int mode, hipStream_t st)
{
    if (!conv_wide_supported(p, mode)) { cs_set_error("conv_wide: this launch is not one of the kernel's shapes / tensor combinations"); return -1; }
    {
        const long in_span = (long)(p.N - 1) * p.in_sN + (long)(p.H - 1) * p.in_sH + (long)(p.W - 1) * p.in_sW + p.Cin;
        if (in_span >= (1L << 31) || p.in_sH >= (1L << 23) || p.in_sW >= (1L << 23)) { cs_set_error("conv_wide: input too large for 32-bit offsets / 24-bit axis products"); return -1; }
        if (ep_check_extents(p, "conv_wide")) return -1;
    }
    WideSched s;
    s.nTW = p.W / 16; s.nTH = p.H / 16;
    s.ntiles = p.N * s.nTW * s.nTH;
    s.ncb = p.Cout_pad / 256;
    const int G = 256;                                // one workgroup per CU; 32 per XCD
    s.tg = (G / 8) / s.ncb;
    s.t8 = (s.ntiles + 7) / 8;
    ConvParams kp = p;
    const int code = wide_ep_code(p);
    if (mode == MODE_TBLEND) {
        if (code == EP_CODE(0, 1, 0, 0, 1)) return launch_wide_inst<MODE_TBLEND, EP_CODE(0, 1, 0, 0, 1)>(kp, s, G, st);
        if (code == EP_CODE(2, 1, 1, 1, 1)) return launch_wide_inst<MODE_TBLEND, EP_CODE(2, 1, 1, 1, 1)>(kp, s, G, st);
    }
    cs_set_error("conv_wide: no instantiation for mode %d / epilogue code %d", mode, code);
    return -1;
}
