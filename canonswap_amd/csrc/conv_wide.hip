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
#define EP_PIPE_V EP_PIPE_K       /* residual fetch rounds issued ahead of the previous round's stores (conv_epilogue.h): set per instantiation below */
// partial-statistics blocks (64 positions = 16 columns x 4 rows) in the order conv_halo's 16 x 8 tiles give them: ((row block of 8) * tiles per
// row + tile column) * 2 + half - the finishing kernel adds the partials in block order, so the (mean, rstd) bits are the same on either kernel
#define EP_BLK_V(sg) ((((th * 2 + ep_wpx) * s.nTW + tw) * 2) + (sg))
#define EP_NBLK_V (s.nTW * s.nTH * 4)
#include "conv_epilogue.h"
#include <cstdlib>

namespace {

constexpr int W_SLP = 10, W_VS = W_SLP * 16;          // slots per voxel (8 data + 2 pad), bytes per voxel
constexpr int W_HW = 18, W_HV = W_HW * W_HW;          // halo of a 16 x 16 tile
constexpr int W_BUF = W_HV * W_VS;                    // 51 840 bytes of a chunk's image
constexpr int W_NPIECE = W_HV * W_SLP;                // 3240 16-byte pieces per chunk
constexpr int W_HI = (W_NPIECE + 255) / 256;          // 13 pieces per thread
constexpr int W_BSTRIDE = W_HI * 4096;                // LDS bytes between the two buffers: the last piece's lanes beyond the image land in the gap
constexpr int W_PFS = 3;                              // weight ring depth in K-steps (a K-step is 64 MFMAs ~ 1 000 cycles)
constexpr int W_NT = 9, W_NS = 18;                    // taps, K-steps per 64-channel chunk
constexpr int W_ST0 = 1;                              // K-steps W_ST0 .. W_ST0 + 12 of a chunk each issue one DMA piece of the next chunk

typedef unsigned int u4w_t __attribute__((ext_vector_type(4)));

// The epilogue reads the accumulators through this proxy: a VALU instruction cannot take an AGPR source, and left to itself the register
// allocator splits the live range of all 256 accumulators at the loop exit (256 v_accvgpr_read into VGPRs at once, which then spill).  An
// explicit read with an AGPR-constrained operand keeps every accumulator where the MFMAs left it until the element is needed.
struct WideAcc3 { const f4_t& v; __device__ __forceinline__ float operator[](int r) const { float x; asm("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(v[r])); return x; } };
struct WideAcc2 { const f4_t (&row)[8]; __device__ __forceinline__ WideAcc3 operator[](int pi) const { return WideAcc3{row[pi]}; } };
struct WideAcc1 { const f4_t (&a)[8][8]; __device__ __forceinline__ WideAcc2 operator[](int ci) const { return WideAcc2{a[ci]}; } };

// -DW_TL (tools/wide_probe.py, instrumented A/B build only): every wave accumulates s_memtime cycles per phase of its life
#ifdef W_TL
unsigned long long* g_wide_tl = nullptr;
long g_wide_tl_cap = 0;
#define WTL(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_acc[i] += t_ - tl_last; tl_last = t_; \
                    __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WTL(i) do { } while (0)
#endif

template <int N> __device__ __forceinline__ void wide_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// schedule of the persistent grid (set by the launcher)
struct WideSched {
    int ntiles;        // N * (H / 16) * (W / 16)
    int nTW, nTH;      // tiles per row / column of a sample
    int ncb;           // 256-channel blocks of the layer (1, 2, 4)
    int tg;            // tile groups per XCD = (G / 8) / ncb
    int t8;            // tiles per XCD (contiguous range)
    int in_bytes;      // bytes one sample of the input spans (< 2^31): the DMA's buffer range
#ifdef W_TL
    unsigned long long* tl; long tl_cap;     // 12 x u64 per wave: cycles in [startup, ring prime, vm wait, barrier, stage, main loop, next-item stage, epilogue], items, chunks, hw id, end time
#endif
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
#ifdef W_TL
    unsigned long long tl_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime();
#endif

    // ---- this workgroup's item list: channel block cb (fixed), tiles xcd * t8 + tgi + tg * j
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int cblk = slot % s.ncb, tgi = slot / s.ncb;
    const int n0 = cblk * 256;
    auto tile_of = [&](int j) -> int {              // -1: no such item
        const int r = tgi + s.tg * j;
        const int t = xcd * s.t8 + r;
        return (r < s.t8 && t < s.ntiles) ? t : -1;
    };

    const int isH = (int)p.in_sH, isW = (int)p.in_sW;
    // ---- halo staging: piece q = tid + 256 * j of a chunk <-> (voxel q / 10, slot q % 10); global -> LDS directly, pad slots are not fetched
    // Buffer-addressed DMA (buffer_load_dwordx4 ... lds): a lane whose byte offset lies outside the buffer gets zeros, so pad slots, voxels
    // outside the image and the lanes of the last piece beyond the image carry a sentinel offset - no zero page, no per-piece address select
    // (two instructions per piece in the K loop: the LDS base into M0 and the load; the chunk's channel offset is the scalar offset).
    __amdgpu_buffer_rsrc_t in_rsrc;                  // the buffer is the sample of the item being staged (set by setup_item)
    constexpr unsigned W_OOB = 0x80000000u;          // launcher: the input spans less than 2^31 bytes
    unsigned poff[W_HI];                             // byte offset of piece j's 16 bytes at channel 0 of the chunk
    auto setup_item = [&](int tile) {                // tile < 0: nothing to stage (every piece is out of range)
        int t = tile < 0 ? 0 : tile;
        const int tw = t % s.nTW; t /= s.nTW;
        const int th = t % s.nTH; t /= s.nTH;
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (long)t * p.in_sN), 0, s.in_bytes, 0x00020000);
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
            const bool inb = tile >= 0 && q < W_NPIECE && sl < 8 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            poff[j] = inb ? (unsigned)(__mul24(ih, isH) + __mul24(iw, isW) + sl * 8) * 2u : W_OOB;
        }
    };
    // piece j of a chunk, asynchronous (vmcnt): awaited by the counted wait + barrier at the chunk's head.  Every lane takes part (no exec
    // mask, no branch: the K-steps that carry a piece stay one scheduling region).
    auto stage_piece = [&](int buf, int c0, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (__attribute__((address_space(3))) void*)(smem + (size_t)buf * W_BSTRIDE + (size_t)(256 * j + wave * 64) * 16),
                                                 16, (int)poff[j], c0 * 2, 0, 0);
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
#pragma unroll
    for (int j = 0; j < W_HI; ++j) stage_piece(0, 0, j);
    int gbuf = 0;                                    // buffer of the chunk about to be computed

    WTL(0);
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

        // ---- weight ring: W_PFS K-steps x 8 fragments, streamed global -> VGPR with loads the compiler does not track (inline asm) and
        // waited for with counted s_waitcnt.  With compiler-tracked loads every wait for a fragment degenerates to vmcnt(0) while an LDS DMA
        // of the halo is in flight (hipcc treats the two kinds of vmcnt events as returning out of order), i.e. no halo piece could be
        // staged under the MFMAs at all; it also drains vmcnt at the chunk loop's back-edge.  The waits name the slot's registers as
        // read-write operands, so no consumer can be scheduled above them; tools/../_lib.isa_check_wide verifies in the disassembly that every
        // MFMA's weight operand was written by a ring load (never a copy of one).
        u4w_t wr[W_PFS][WCH];
        auto wsrc_of = [&](int cc, int st) -> const half_t* {            // st: compile-time after unrolling (may run into the next chunk)
            const int c2 = cc + st / W_NS, s2 = st % W_NS;
            const int ccl = c2 < nck ? c2 : nck - 1;                      // behind the last chunk the carried fetches repeat and are dropped
            return wlane + (long)((ccl * 2 + s2 % 2) * W_NT + s2 / 2) * wstep;
        };
        auto wload1 = [&](u4w_t& dst, const half_t* src, int ci) {        // ci: compile-time after unrolling
            // two base pointers: the fragments' row offsets (ep_frag_row) reach beyond the 12-bit instruction offset
            const half_t* b = src + ep_frag_row(EP_PAIR, ci & 4) * 32;
            const int off = (ep_frag_row(EP_PAIR, ci) - ep_frag_row(EP_PAIR, ci & 4)) * 64;
            if (off == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(b));
            else if (off == 256) asm volatile("global_load_dwordx4 %0, %1, off offset:256" : "=v"(dst) : "v"(b));
            else if (off == 1024) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(dst) : "v"(b));
            else if (off == 1280) asm volatile("global_load_dwordx4 %0, %1, off offset:1280" : "=v"(dst) : "v"(b));
            else if (off == 2048) asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(dst) : "v"(b));
            else if (off == 2304) asm volatile("global_load_dwordx4 %0, %1, off offset:2304" : "=v"(dst) : "v"(b));
            else __builtin_trap();
        };
#define WIDE_RING_WAIT(N, S) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(wr[S][0]), "+v"(wr[S][1]), "+v"(wr[S][2]), "+v"(wr[S][3]), \
                                                                     "+v"(wr[S][4]), "+v"(wr[S][5]), "+v"(wr[S][6]), "+v"(wr[S][7]))
        static_assert(W_PFS == 3 && W_NS % W_PFS == 0, "the counted waits below are written for a three-slot ring");
#pragma unroll
        for (int st = 0; st < W_PFS; ++st)
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci) wload1(wr[st][ci], wsrc_of(0, st), ci);
        WTL(1);

        for (int cc = 0; cc < nck; ++cc) {
            // ---- head of a chunk: its halo (staged one chunk ago by every wave) has landed: this wave's pieces by the counted wait - the
            // ring fetches (and, at an item's first chunk, the previous epilogue's stores) are younger and stay in flight -, everyone's by
            // the barrier, which also says that everyone has left the other buffer.  Then the next chunk (or the next item's first one,
            // whose addressing is computed here, under the MFMAs that follow) goes into that buffer.
            wide_wait_vm<(W_PFS - 1) * WCH>();
            WTL(2);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            WTL(3);
            // the next chunk - or the next item's first one, whose addressing replaces this item's here - is staged piece by piece under the
            // K-steps below
            int c0n = (cc + 1) * 64;
            if (cc + 1 == nck) { setup_item(next_tile); c0n = 0; }
            const int nbuf = gbuf ^ 1;
            WTL(4);
            const unsigned char* hb = smem + (size_t)gbuf * W_BSTRIDE + abase0;
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
            // Issue order of a step, pinned by scheduling fences (the asm loads are opaque to the scheduler's instruction classes):
            //   wait for the step's ring slot | 16 MFMA pairs on position fragments 0-3, with pair k: LDS read of fragment 2 + k (2 <= k < 6) or
            //   reload of fragment k - 6 of the slot the previous step used (6 <= k < 14) | one DMA piece of the next chunk | 16 MFMA pairs on
            //   fragments 4-7, with pairs 2-5 the LDS reads of the next step's fragments 0-3.
            // (Pairs 0 and 1 of a half touch all four fragments the previous half read and carry no LDS read: hipcc waits for LDS data with
            // lgkmcnt(0) while it believes an LDS DMA is pending - it never sees the counted vmcnt waits -, so a read issued before those first
            // uses would be waited for right away: 100 cycles of idle matrix pipe per half.)
            // VMEM instructions younger than the loads of step st's slot when the step starts: the 8 reloads of step st - 1 plus the DMA pieces
            // of steps st - 2 and st - 1 (step 0: the 16 loads of the two other slots; they were issued behind the previous chunk's step 16).
#pragma unroll
            for (int st = 0; st < W_NS; ++st) {
                const int toff = toff_of(st);
                constexpr int S0 = 0;
                (void)S0;
                {
                    const bool d2 = st - 2 >= W_ST0 && st - 2 < W_ST0 + W_HI, d1 = st - 1 >= W_ST0 && st - 1 < W_ST0 + W_HI;
                    const int nyoung = st == 0 ? 16 : 8 + (d2 ? 1 : 0) + (d1 ? 1 : 0);
                    const int sl = st % W_PFS;
                    // (the count must be an instruction immediate: one asm per (count, slot))
                    if (nyoung == 16) { if (sl == 0) WIDE_RING_WAIT(16, 0); else if (sl == 1) WIDE_RING_WAIT(16, 1); else WIDE_RING_WAIT(16, 2); }
                    else if (nyoung == 8) { if (sl == 0) WIDE_RING_WAIT(8, 0); else if (sl == 1) WIDE_RING_WAIT(8, 1); else WIDE_RING_WAIT(8, 2); }
                    else if (nyoung == 9) { if (sl == 0) WIDE_RING_WAIT(9, 0); else if (sl == 1) WIDE_RING_WAIT(9, 1); else WIDE_RING_WAIT(9, 2); }
                    else { if (sl == 0) WIDE_RING_WAIT(10, 0); else if (sl == 1) WIDE_RING_WAIT(10, 1); else WIDE_RING_WAIT(10, 2); }
                }
                __builtin_amdgcn_sched_barrier(0);
                const half_t* rsrc = wsrc_of(cc, st - 1 + W_PFS);       // step st reloads the slot of step st - 1
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int ci = k >> 1, p0 = (k & 1) * 2;
                    acc[ci][p0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % W_PFS][ci]), afA[p0], acc[ci][p0], 0, 0, 0);
                    acc[ci][p0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % W_PFS][ci]), afA[p0 + 1], acc[ci][p0 + 1], 0, 0, 0);
                    if (k >= 2 && k < 6) afB[k - 2] = *(const h8_t*)(hb + (HA + k - 2) * (W_HW * W_VS) + toff);
                    else if (k >= 6 && k < 14 && st >= 1) wload1(wr[(st - 1) % W_PFS][k - 6], rsrc, k - 6);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // one DMA piece of the next chunk per K-step (a burst of 13 at the chunk's head idled the matrix pipe for 1 200 cycles of a
                // 20 000-cycle chunk, and for 4 000 at an item's end; profiles/r04_b_wide_probe.txt)
                if (st >= W_ST0 && st < W_ST0 + W_HI) { stage_piece(nbuf, c0n, st - W_ST0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int ci = k >> 1, p0 = (k & 1) * 2;
                    acc[ci][HA + p0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % W_PFS][ci]), afB[p0], acc[ci][HA + p0], 0, 0, 0);
                    acc[ci][HA + p0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[st % W_PFS][ci]), afB[p0 + 1], acc[ci][HA + p0 + 1], 0, 0, 0);
                    if (k >= 2 && k < 6 && st + 1 < W_NS) afA[k - 2] = *(const h8_t*)(hb + (k - 2) * (W_HW * W_VS) + toff_of(st + 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            {   // step W_NS - 1 used slot (W_NS - 1) % W_PFS, which serves step W_PFS - 1 of the next chunk
                const half_t* rsrc = wsrc_of(cc, W_NS - 1 + W_PFS);
#pragma unroll
                for (int ci = 0; ci < WCH; ++ci) wload1(wr[(W_NS - 1) % W_PFS][ci], rsrc, ci);
            }
            WTL(5);
#ifdef W_TL
            tl_acc[9] += 1;
#endif
        }

        wide_wait_vm<0>();      // the ring's last (dropped) fetches are loads the compiler does not know of: they must have landed before it reuses their registers
        __builtin_amdgcn_sched_barrier(0);
        // (a use behind the drain: the destination registers of those loads stay reserved until the data has landed)
#pragma unroll
        for (int st = 0; st < W_PFS; ++st)
            asm volatile("" :: "v"(wr[st][0]), "v"(wr[st][1]), "v"(wr[st][2]), "v"(wr[st][3]), "v"(wr[st][4]), "v"(wr[st][5]), "v"(wr[st][6]), "v"(wr[st][7]));
        WTL(6);
        // ---- epilogue (conv_epilogue.h): the tensor combination is a compile-time constant.  Stores are not waited for here.
        {
            constexpr bool EP_HEAVY = false, EP_EARLY = false;
            // position blocks whose residual / mask is fetched per round, and whether the rounds are pipelined (the next round's fetch ahead of
            // this round's stores, two register sets): the whole tile in one round where no residual is read; with a residual 2 blocks per
            // round (1 for the fp32 residual of the non-blend form: 32 registers per block), pipelined.  No scratch in any instantiation.
            constexpr bool EP_HASRES = (EPC & 3) != 0;
            constexpr int EP_GSEL_K = !EP_HASRES ? 8 : (MODE == MODE_STD && (EPC & 3) == 2) ? 1 : (MODE == MODE_SPADE ? 8 : (MODE == MODE_STDSTAT ? 1 : 2));
            constexpr int EP_PIPE_K = EP_HASRES && MODE != MODE_SPADE ? 1 : 0;
            constexpr int EP_WPX = WPX;
            constexpr int lgTW = 4, lgTH = 4, lgTD = 0, lgS = 8, mW = 15, mH = 15, mD = 0;
            const int ep_wpx = wp;
            ep_u2_t ep_xpre[1][1];
            (void)ep_xpre; (void)td; (void)CSTEP_W;
            const WideAcc1 ep_acc{acc};
            CONV_EPILOGUE_IMPL(EPC);
        }
        WTL(7);
#ifdef W_TL
        tl_acc[8] += 1;
#endif
        tile = next_tile;
        ++j_item;
    }
#ifdef W_TL
    if (s.tl && lane == 0) {
        const long wi = (long)blockIdx.x * 4 + wave;
        if (wi < s.tl_cap) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned long long* o = s.tl + wi * 12;
#pragma unroll
            for (int i = 0; i < 10; ++i) o[i] = tl_acc[i];
            o[10] = hwid; o[11] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

template <int MODE, int EPC>
int launch_wide_inst(const ConvParams& p, const WideSched& s, int grid, hipStream_t st)
{
    auto k = conv_wide_kernel<MODE, EPC>;
    const size_t lds = 2 * (size_t)W_BSTRIDE;
    {   // per launch: the attribute belongs to the current device, and engines may live on several GPUs of one process
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { cs_set_error("conv_wide: opting into %zu bytes of LDS failed: %s", lds, hipGetErrorString(e)); return -1; }
    }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, p, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cs_set_error("conv_wide launch: %s", hipGetErrorString(e)); return -1; }
    return 0;
}

// the instantiations: (mode, tensor combination) of the engine's wide 3x3 layers
//   T blend conv1 / conv2 (adaptive_modulate.py:337-349) | R's 2-D pair (util.py:120-128), W.third | G 3x3 convs with the next norm's statistics,
//   without / with the block's fp16 residual (util.py:329-344) | SPADE gamma / beta convs (util.py:295-302)
constexpr int WIDE_NINST = 7;
constexpr int WIDE_INST[WIDE_NINST][2] = {
    {MODE_TBLEND, EP_CODE(0, 1, 0, 0, 1)}, {MODE_TBLEND, EP_CODE(2, 1, 1, 1, 1)},
    {MODE_STD, EP_CODE(0, 1, 0, 0, 0)}, {MODE_STD, EP_CODE(2, 1, 1, 1, 0)},
    {MODE_STDSTAT, EP_CODE(0, 1, 0, 0, 0)}, {MODE_STDSTAT, EP_CODE(1, 1, 0, 0, 0)},
    {MODE_SPADE, EP_CODE(1, 1, 0, 0, 0)},
};
int wide_inst_of(int mode, int code)
{
    for (int i = 0; i < WIDE_NINST; ++i) if (WIDE_INST[i][0] == mode && WIDE_INST[i][1] == code) return i;
    return -1;
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
    if (p.KD != 1 || p.KH != 3 || p.KW != 3 || p.D != 1 || p.inD != 1 || p.up_shift || p.cg || p.hilo || p.ragged || p.sk_out || p.kw_out || p.xf_kind || p.spmul || p.pool_hw || p.nphase || p.PH != 1 || p.PW != 1) return false;
    if (p.H % 16 || p.W % 16 || p.Cin % 64 || p.Cout_pad % 256 || p.Cout_pad > 1024) return false;
    if (p.ep_general) return false;
    if (mode == MODE_STD && p.stat_out) mode = MODE_STDSTAT;
    if ((mode == MODE_STDSTAT) != (p.stat_out != nullptr)) return false;
    if (mode == MODE_SPADE && (!p.res.p || p.res_f32 || !p.stats || !p.bias || !p.bias2)) return false;
    const int cstep = (mode == MODE_TBLEND || mode == MODE_SPADE) ? 2 : 1;
    if (p.Cout * cstep != p.Cout_pad || (p.Cout & 7)) return false;
    const int ncb = p.Cout_pad / 256;
    if (ncb != 1 && ncb != 2 && ncb != 4) return false;
    if (p.act0 >= ACT_SIGMOID || p.act1 >= ACT_SIGMOID) return false;
    auto al8 = [](const TDesc& t) { return (((unsigned long long)t.p & 15ull) == 0) && (((t.sN | t.sD | t.sH | t.sW) & 7) == 0); };
    if ((p.res.p && !p.res_f32 && !al8(p.res)) || (p.out0.p && !p.out0_f32 && !al8(p.out0)) || (p.out1.p && !al8(p.out1))) return false;
    if (((unsigned long long)p.in & 15ull) || ((p.in_sN | p.in_sH | p.in_sW) & 7)) return false;
    // a sample of the input is addressed with 31-bit byte offsets, its axis products are 24-bit multiplies
    if ((long)(p.H - 1) * p.in_sH + (long)(p.W - 1) * p.in_sW + p.Cin >= (1L << 30) || p.in_sH >= (1L << 23) || p.in_sW >= (1L << 23)) return false;
    return wide_inst_of(mode, wide_ep_code(p)) >= 0;
}

int launch_conv_wide(const ConvParams& p, int mode, hipStream_t st)
{
    if (!conv_wide_supported(p, mode)) { cs_set_error("conv_wide: this launch is not one of the kernel's shapes / tensor combinations"); return -1; }
    {
        if (ep_check_extents(p, "conv_wide")) return -1;
    }
    const long in_span = (long)(p.H - 1) * p.in_sH + (long)(p.W - 1) * p.in_sW + p.Cin;      // one sample
    WideSched s;
    s.in_bytes = (int)(in_span * 2);
    s.nTW = p.W / 16; s.nTH = p.H / 16;
    s.ntiles = p.N * s.nTW * s.nTH;
    s.ncb = p.Cout_pad / 256;
    const int G = 256;                                // one workgroup per CU; 32 per XCD
    s.tg = (G / 8) / s.ncb;
    s.t8 = (s.ntiles + 7) / 8;
#ifdef W_TL
    s.tl = g_wide_tl; s.tl_cap = g_wide_tl_cap;
#endif
    ConvParams kp = p;
    if (mode == MODE_STD && p.stat_out) mode = MODE_STDSTAT;
    const int code = wide_ep_code(p);
    switch (wide_inst_of(mode, code)) {
#define WIDE_CASE(i) case i: return launch_wide_inst<WIDE_INST[i][0], WIDE_INST[i][1]>(kp, s, G, st);
        WIDE_CASE(0) WIDE_CASE(1) WIDE_CASE(2) WIDE_CASE(3) WIDE_CASE(4) WIDE_CASE(5) WIDE_CASE(6)
#undef WIDE_CASE
        default: break;
    }
    cs_set_error("conv_wide: no instantiation for mode %d / epilogue code %d", mode, code);
    return -1;
}

#ifdef W_TL
// instrumented builds only: device buffer of cap x 12 u64 that every following conv_wide launch fills (nullptr: off)
extern "C" void cs_debug_set_wide_tl(void* buf, long cap) { g_wide_tl = (unsigned long long*)buf; g_wide_tl_cap = cap; }
#endif
