// 3x3x3 convolutions 32 -> 32 channels on the canonical feature volume [N][H][W][D = 16][C = 32] for gfx950 (CDNA4).
//
// Replaces, for the 36 such convolutions of a frame (ResBlock3d of F and T, util.py:80-102; ResBlock3D_stage3_leak of R,
// util.py:528-544), the generic LDS-halo kernel (conv_halo_kernel.h).  At 32 output channels that kernel was bound by what it
// streams per MFMA - every wave re-fetched the 55 KB of weights from L1/L2 for each 256-position tile (2 KB per 8 MFMAs against a
// measured 26-36 B/clk/CU), two-way LDS bank conflicts on top - and by work outside its main loop (13-piece halo prologue, first
// halo wait, epilogue: 51-59 % of a wave's life; profiles/r02_i_timeline.txt).  This kernel is built around the shape instead:
//
//   * WEIGHTS LIVE IN REGISTERS.  27 taps x 2 fragments x 16 bytes per lane = 216 VGPRs per wave, loaded once per workgroup; one
//     wave per SIMD (a 512-register budget).  Nothing but activations moves during the main loop.
//   * THE 16 POSITIONS OF AN MFMA BLOCK ARE THE 16 DEPTH SLICES OF ONE (h, w) COLUMN.  D = 16 is the whole depth, so a column
//     (16 voxels x 32 channels fp16) is one contiguous KiB in HBM on both sides, the depth taps are shifts inside a column and the
//     row / column taps are whole-column moves.
//   * PERSISTENT WORKGROUPS MARCH ALONG H.  A workgroup owns a strip of 8 columns (w) and walks its rows: per step it computes one
//     output row of the strip (8 columns x 27 taps x 2 fragments = 432 MFMAs, two columns per wave) from the three input rows
//     resident in an LDS ring of four row slabs, while the DMA (global_load_lds) of the row after next lands in the fourth slot,
//     the stores of the previous step drain and the residual of this step is in flight: staging, epilogue and main loop of
//     neighbouring steps overlap instead of running back to back per tile, and the halo is re-fetched along W only (10 columns per
//     8), never along H or D.
//   * CONFLICT-FREE LDS IMAGE WITHOUT PAD SLOTS.  Per row slab and 8-channel group q (= one 16-byte slot of a voxel) a plane of
//     10 columns x 17 slots: slot 0 of a column is zero (the d = -1 tap; the next column's zero slot is this column's d = 16 tap),
//     slots 1..16 the voxels.  The planes of the four q lie a multiple of 256 bytes apart, so the 16 lanes ds_read_b128 serves per
//     cycle ({8 depth slices of q} + {the other 8 of q + 1}) always cover 16 distinct bank quads, for every column and every depth
//     shift (the voxel-major image of conv_halo needs two pad slots per voxel for that: +50 % LDS).  The DMA writes the image
//     directly: the LDS side of global_load_lds is linear in the lane index, the global side is free per lane.
//
// Accumulation order per output element: taps in (kd, kh, kw) raster order, one 32-channel MFMA K-step each - the same as
// conv_halo_kernel, so both kernels give the same bits (tests/test_gpu_vol32.py compares them with torch.equal).
#include "conv_epilogue.h"

namespace {

constexpr int V_TW = 8;                      // output columns (w) of a strip
constexpr int V_NC = V_TW + 2;               // columns of a row slab (one halo column on either side)
constexpr int V_CS = 17 * 16;                // bytes of one column in one q plane: zero slot + 16 voxels x 16 bytes
constexpr int V_RS = V_NC * V_CS;            // bytes of a row slab in one q plane (2720)
constexpr int V_WLO = 27 * 2 * 1024;         // split precision: the W_lo fragments, [tap][fragment][lane] x 16 bytes
constexpr int V_SROWS = 8;                   // statistics: one partial per (column pair, 8 rows) - independent of the work decomposition
// epilogue forms (the three uses on the path; each has a fixed number of memory instructions per step, which the counted waits rely on)
enum { V_EPI_F16 = 0,    // out0 fp16 = act0(conv + bias)                                              (conv1 of a ResBlock3d)
       V_EPI_RES = 1,    // out0 fp32 = conv + bias + res (fp32), out1 fp16 = act1(out0 * s2 + t2)      (conv2 of a ResBlock3d)
       V_EPI_STAT = 2 }; // out0 fp32 = conv + bias, partial statistics of it                            (R's GroupNorm convs)

// ring of row slabs: rows h-1, h, h+1 feed step h; rows up to h + K are staged ahead.  HBM latency under load is 1.5-2 us and a step
// of the plain kernel 0.7 us of MFMA work: with one row of look-ahead the first build of this kernel ran at the rate of one DMA round
// trip per step (profiles/r03_b_layers_b32.csv: 124 / 190 us per conv against 46 us of MFMA time).
template <bool SPLIT, int EPI = 0> struct VRing {
#ifdef V32_K6
    static constexpr int K = SPLIT ? 2 : (EPI == 0 ? 6 : 4);
#else
    static constexpr int K = SPLIT ? 2 : 4;
#endif                     // look-ahead in rows (split: 143 KB of LDS at K = 2; its step is 3x longer)
    static constexpr int RING = K + 2;
    static constexpr int QS = ((RING * V_RS + 16 + 255) / 256) * 256;      // q plane stride: a multiple of 256 bytes (bank-conflict freedom)
    static constexpr int IMG = 4 * QS;                          // one image (four q planes)
    static constexpr int NIMG = SPLIT ? 2 : 1;
    static constexpr int AUX = NIMG * IMG;                      // W_lo fragments (split) or the residual ring (V_EPI_RES) follow the images
    static constexpr int RROWS = K + 1;                         // residual ring: row h is staged with input row h, K steps before step h uses it
    static constexpr int RSLOT = 4 * 4096;                      // one residual row of a strip: 4 waves x 2 columns x 2 KB fp32
    static constexpr int lds(int epi) { return AUX + (SPLIT ? V_WLO : 0) + (epi == 1 ? RROWS * RSLOT : 0); }
};

struct Vol32Params {
    const half_t* in; const half_t* zero;
    int in_sN, in_sH, in_sW;                 // element strides; a column (16 voxels x 32 [x 2: hi | lo] channels) is contiguous
    int N, H, W;
    const half_t* wgt;                       // packed [chunk][27 taps][32 rows][32 k]; split precision: chunks W_hi | W_lo | W_hi
    const float* bias;
    float sl0, sl1;                          // lin_act slopes of act0 / act1
    const float* res; int res_sN, res_sH, res_sW;        // fp32 residual
    void* out0; int o0_sN, o0_sH, o0_sW;
    half_t* out1; int o1_sN, o1_sH, o1_sW;
    const float* s2; const float* t2;
    float* stat_out; int stat_nblk;          // [N][stat_nblk][32][2] partial (sum, sum of squares) of the stored out0 values
    int srows;                               // rows of a partial-statistics block (V_SROWS; 2 in latency mode)
    // transform staging (XF kernels; ConvParams::xf_*)
    const float* xf_y; int xy_sN, xy_sH, xy_sW;
    const float* xf_res; int xr_sN, xr_sH, xr_sW;
    float* xf_out; int xo_sN, xo_sH, xo_sW;
    const float* xf_stats; const float* xf_gamma; const float* xf_beta; float xf_slope;
    int nstrips, nseg, seg_rows, items;
#ifdef V32_TL
    unsigned long long* tl; long tl_cap;     // 12 x u64 per wave: cycles in [startup, prologue wait, DMA issue, stores, main loop, epilogue, vm wait, barrier, tail], steps, hw id
#endif
};

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

// -DV32_TL (tools/vol32_probe.py, instrumented A/B build only): every wave accumulates s_memtime cycles per phase of its life
#ifdef V32_TL
#define TLS(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_acc[i] += t_ - tl_last; tl_last = t_; \
                    __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TLS(i) do { } while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// SPLIT: split-precision convolution (ConvParams::hilo): activations [hi | lo] per voxel, out = W_hi x_hi + W_lo x_hi + W_hi x_lo, the 81
// K-steps of an output element in exactly that order (= conv_halo_kernel's chunk order).  W_hi stays in registers; W_lo lives in LDS as
// ready-made A fragments (54 KiB, linear in the lane index: conflict-free) and is read per tap - 10 LDS reads per 12 MFMAs in that pass,
// 4 per 12 in the other two: the LDS runs at half its rate.
// XF (transform staging, split-precision kernels only): 0 - the input is DMA-staged from a [hi | lo] fp16 volume;  1 - it is an fp32 volume,
// split on the way into LDS;  2 .. 5 - it is lrelu(GroupNorm(y) [+ res]) of fp32 volumes, computed, split and (with xf_out) written back
// while it is staged (2 + (residual ? 1 : 0) + (write-back ? 2 : 0): compile-time variants, the staging code has no run-time branch).  Raw rows travel global -> registers -> VALU -> ds_write within one step (fetched early, converted late), spread
// over the MFMA groups, where the VALU work hides under the matrix pipe.
template <int EPI, bool SPLIT, int XF>
__global__ void __launch_bounds__(256, 1) vol32_kernel(const Vol32Params p)
{
    static_assert(XF == 0 || (SPLIT && EPI == V_EPI_STAT), "transform staging exists for the split-precision statistics kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using R = VRing<SPLIT, EPI>;
    constexpr int K = R::K, RING = R::RING, QS = R::QS, IMG = R::IMG, NIMG = R::NIMG;
    constexpr int VOX = SPLIT ? 64 : 32;     // input elements per voxel
    static_assert(!(SPLIT && EPI == V_EPI_RES), "the residual ring and the W_lo fragments share the LDS region after the images");
    // vector-memory instructions per step, in issue order: DMA of input row h + K (and of residual row h + K), stores of row h - 1.  No
    // load of the loop returns into registers: hipcc drains vmcnt to 0 wherever a loop-carried load result is used, which would put a
    // full memory round trip into every step (the residual therefore goes through LDS as well).
    constexpr int ND = XF ? 0 : 3 * NIMG + (EPI == V_EPI_RES ? 4 : 0), NR = 0, NS = EPI == V_EPI_F16 ? 2 : (EPI == V_EPI_RES ? 6 : 4);
    // End of step h: the DMA of row h + 2 (issued K - 2 steps ago, first thing of its step) must have landed.  vmcnt counts in order
    // (loads, LDS DMA and stores share the queue on gfx9: the compiler's own waits rest on the same fact), so "at most NW outstanding"
    // does it when NW never exceeds what was issued after that DMA: the residual loads and stores of its own step (compiler fences keep
    // the three groups of a step in program order; the first step of an item has no stores) and K - 2 whole steps.  More instructions
    // than that (the statistics partials) only make the wait earlier than necessary.
    constexpr int NW0 = NR + (K - 2) * (ND + NR + NS), NW = NW0 + NS;
    static_assert(NW <= 63, "vmcnt is a 6-bit counter");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
#ifdef V32_TL
    unsigned long long tl_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime();
#endif

    // ---- once per workgroup: zero the LDS images (zero slots are never written again), weights into registers [and LDS]
    for (int i = tid * 16; i < NIMG * IMG; i += 256 * 16) *(u4v*)(smem + i) = (u4v){0u, 0u, 0u, 0u};
    h8_t wr[27][2];
    {
        // EP_PAIR 1 row permutation (conv_epilogue.h): a lane ends up with 8 consecutive output channels l4 * 8 .. + 7
        const half_t* wl = p.wgt + ep_lane_row(1, l15) * 32 + l4 * 8;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            wr[t][0] = *(const h8_t*)(wl + t * 1024 + ep_frag_row(1, 0) * 32);
            wr[t][1] = *(const h8_t*)(wl + t * 1024 + ep_frag_row(1, 1) * 32);
        }
        if constexpr (SPLIT) {               // W_lo (chunk 1): fragment (tap, ci) of this lane -> LDS at ((tap * 2 + ci) * 64 + lane) * 16
            for (int f = wave; f < 54; f += 4) {
                const h8_t v = *(const h8_t*)(wl + (27 + (f >> 1)) * 1024 + ep_frag_row(1, f & 1) * 32);
                *(h8_t*)(smem + R::AUX + f * 1024 + lane * 16) = v;
            }
        }
    }
    float bias_v[8], s2_v[8], t2_v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        bias_v[r] = p.bias ? p.bias[l4 * 8 + r] : 0.f;
        s2_v[r] = p.s2 ? p.s2[l4 * 8 + r] : 1.f;
        t2_v[r] = p.s2 ? p.t2[l4 * 8 + r] : 0.f;
    }
    __syncthreads();

    // ---- staging constants.  Wave q stages q plane q of a row slab (of both images): 170 slots = 3 DMA instructions of 64 lanes; slot idx ->
    // (column idx / 17, depth idx % 17 - 1); slot 0 of a column stays zero (lanes masked off)
    int s_off[3]; bool s_on[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int idx = j * 64 + lane, col = idx / 17, ds = idx % 17;
        s_on[j] = idx < V_NC * 17 && ds != 0;
        s_off[j] = col * p.in_sW + (ds - 1) * VOX + wave * 8;
#ifdef V32_EXP_COAL        /* timing experiment only (wrong results): consecutive lanes fetch consecutive 16-byte pieces */
        s_off[j] = col * p.in_sW + ((ds - 1) * 8 + wave * 128) % 512;
#endif
    }
    const int lanebase = l4 * QS + l15 * 16 + (2 * wave) * V_CS;         // fragment reads: q plane l4, depth l15, this wave's first column
    const unsigned lane_el = (unsigned)(l15 * 32 + l4 * 8);              // epilogue: element offset of this lane inside a column
    const unsigned char* wlo = smem + R::AUX + lane * 16;               // split: W_lo fragments; V_EPI_RES: the residual ring (same lane-linear form)

    // XCD-aware item order: hardware places workgroup b on XCD b % 8; every XCD walks a contiguous range of items (neighbouring strips
    // share halo columns: they hit in that XCD's L2)
    const int G = (int)gridDim.x;
    int u = (int)blockIdx.x;
    if ((G & 7) == 0) u = (u & 7) * (G >> 3) + (u >> 3);

    for (int item = u; item < p.items; item += G) {
        int t = item;
        const int strip = t % p.nstrips; t /= p.nstrips;
        const int seg = t % p.nseg;
        const int n = t / p.nseg;
        const int w0 = strip * V_TW;
        const int h0 = seg * p.seg_rows, h1 = (h0 + p.seg_rows < p.H) ? h0 + p.seg_rows : p.H;
        bool c_ok[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int col = (j * 64 + lane) / 17;
            c_ok[j] = s_on[j] && (unsigned)(w0 - 1 + col) < (unsigned)p.W;
        }
        const half_t* in_n = p.in + (long)n * p.in_sN + (long)(w0 - 1) * p.in_sW;
        const unsigned nb_o0 = (unsigned)(n * p.o0_sN + (w0 + 2 * wave) * p.o0_sW);
        const unsigned nb_o1 = (unsigned)(n * p.o1_sN + (w0 + 2 * wave) * p.o1_sW);
        const unsigned nb_rs = (unsigned)(n * p.res_sN + (w0 + 2 * wave) * p.res_sW);
        // asynchronous; input row r -> ring slot `slot` [and the residual of row r -> residual slot rslot].  Rows outside the volume or beyond
        // what this item reads (r > h1) are zero rows: every step issues the same ND instructions
        auto stage_piece = [&](int r, int slot, int rslot, int q) {      // q: compile-time after unrolling, 0 .. ND - 1
            if (q < 3 * NIMG) {
                const int im = q / 3, j = q % 3;
                const bool rok = (unsigned)r < (unsigned)p.H && r <= h1;
                const half_t* rp = in_n + (long)r * p.in_sH;
                unsigned char* dst = smem + wave * QS + slot * V_RS;
                if (s_on[j]) {
                    const half_t* src = (rok && c_ok[j]) ? rp + s_off[j] + im * 32 : p.zero;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + im * IMG + j * 1024), 16, 0, 0);
                }
            } else if constexpr (EPI == V_EPI_RES) {
                // this wave's two columns of residual row r (2 x 2 KB fp32): lane (d, l4) fetches the two 16-byte halves of ITS 8 channels, each
                // half of a column as one instruction -> the LDS image is linear in the lane index for the reader too (conflict-free)
                const int j = q - 3 * NIMG;
                const int rc = (unsigned)r < (unsigned)p.H ? r : 0;          // rows outside are never consumed: any valid address
                const float* rsrc = p.res + (nb_rs + (unsigned)(rc * p.res_sH) + lane_el);
                unsigned char* rdst = smem + R::AUX + rslot * R::RSLOT + wave * 4096;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rsrc + (j >> 1) * p.res_sW + (j & 1) * 4),
                                                 (__attribute__((address_space(3))) void*)(rdst + j * 1024), 16, 0, 0);
            }
        };
        auto stage_row = [&](int r, int slot, int rslot) {
#pragma unroll
            for (int q = 0; q < ND; ++q) stage_piece(r, slot, rslot, q);
        };
        // ---- transform staging (XF): thread t handles pieces idx = it * 256 + t of a row slab (640 pieces of 8 channels: column idx / 64, depth
        // (idx / 4) % 16, channel group idx % 4 = t % 4 - consecutive threads read consecutive 32 bytes)
        constexpr bool XRES = XF >= 2 && ((XF - 2) & 1) != 0, XWB = XF >= 2 && ((XF - 2) & 2) != 0;     // kind 2 variants: + residual, + write-back
        float4 xr[XF ? 3 : 1][2], xs[XRES ? 3 : 1][2];       // raw row in flight: y [, res]
        float xsc[XF >= 2 ? 8 : 1], xsh[XF >= 2 ? 8 : 1];
        const int xq = tid & 3;
        if constexpr (XF >= 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int c = xq * 8 + r;
                const float mean = p.xf_stats[((long)n * 32 + c) * 2], rstd = p.xf_stats[((long)n * 32 + c) * 2 + 1];
                xsc[r] = gn_scale(rstd, p.xf_gamma[c]);
                xsh[r] = gn_shift(mean, xsc[r], p.xf_beta[c]);
            }
        }
        // per-item constants of a piece: element offset inside the sample's row (the three fp32 volumes share their strides: the launcher
        // checks), LDS byte offset inside a ring slot, validity of its column, ownership of its column (write-back)
        int xoff[XF ? 3 : 1]; bool xcok[XF ? 3 : 1];
        if constexpr (XF != 0) {
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                // 640 pieces on 768 thread slots: the upper half of the third round repeats the lower half's pieces (the same loads, the same
                // values to the same LDS and global addresses) - no test of the piece index inside the K loop
                int idx = it * 256 + tid;
                if (idx >= V_NC * 64) idx -= 128;
                const int col = idx >> 6, d = (idx >> 2) & 15;
                xoff[it] = n * p.xy_sN + (w0 - 1 + col) * p.xy_sW + d * 32 + xq * 8;
                xcok[it] = (unsigned)(w0 - 1 + col) < (unsigned)p.W;
            }
        }
        // Branch-free by construction: a fetch behind a bounds test makes hipcc drain vmcnt to 0 where the branches join (every load and
        // store of the step), and a branch inside the MFMA groups ends the scheduling region the conversion is interleaved in.  Pieces
        // outside the volume fetch a valid address of their own sample (first row and column of the strip) and are zeroed after the
        // transform; their write-back goes to the zero page - as zeros.
        const unsigned xsafe = (unsigned)(n * p.xy_sN + w0 * p.xy_sW + h0 * p.xy_sH + ((tid >> 2) & 15) * 32 + xq * 8);
        auto xf_load_to = [&](int r, int it, float4 (&dr)[2], float4 (&ds)[2]) {       // raw piece it of row r -> registers
            if constexpr (XF != 0) {
                const bool okl = xcok[it] && (unsigned)r < (unsigned)p.H && r <= h1;
                const unsigned o = okl ? (unsigned)(xoff[it] + r * p.xy_sH) : xsafe;
                const float* y = p.xf_y + o;
                dr[0] = *(const float4*)y; dr[1] = *(const float4*)(y + 4);
                if constexpr (XRES) {
                    const float* x = p.xf_res + o;
                    ds[0] = *(const float4*)x; ds[1] = *(const float4*)(x + 4);
                }
            }
        };
        auto xf_load = [&](int r, int it) { xf_load_to(r, it, xr[it], xs[XRES ? it : 0]); };
        // part 0 / 1: the transform of channels 0..3 / 4..7 of the piece; part 2: split, ds_writes [, write-back]; part < 0: all of it.  In the
        // K loop the three parts of a piece go into three consecutive MFMA groups: a wave issues about three VALU instructions in the
        // shadow of one MFMA (one wave per SIMD), a whole piece (~110 instructions) inside one group of 12 MFMAs left the matrix pipe idle
        // for half of it.
        float xa[8];
        auto xf_write_from = [&](int r, int slot, int it, int part, const float4 (&xr)[2], const float4 (&xs)[2]) {     // registers -> transform -> [hi | lo] images of ring slot `slot`
            if constexpr (XF != 0) {
                int idx = it * 256 + tid;
                if (idx >= V_NC * 64) idx -= 128;
                const bool ok = xcok[it] && (unsigned)r < (unsigned)p.H && r <= h1;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (part >= 0 && (k >> 2) != part) continue;
                    const float v = ((const float*)&xr[k >> 2])[k & 3];
                    if constexpr (XRES) xa[k] = gn_lrelu(v, xsc[k], xsh[k], ((const float*)&xs[k >> 2])[k & 3], p.xf_slope);
                    else if constexpr (XF >= 2) xa[k] = gn_lrelu(v, xsc[k], xsh[k], 0.f, p.xf_slope);
                    else xa[k] = v;
                    if (!ok) xa[k] = 0.f;                // zero padding applies to the conv's input, i.e. after the transform
                }
                if (part >= 0 && part != 2) return;
                h8_t hi, lo;
#pragma unroll
                for (int k = 0; k < 8; ++k) { hi[k] = (half_t)xa[k]; lo[k] = (half_t)(xa[k] - (float)hi[k]); }
                unsigned char* dst = smem + xq * QS + (idx >> 6) * V_CS + (((idx >> 2) & 15) + 1) * 16 + slot * V_RS;
                *(h8_t*)dst = hi; *(h8_t*)(dst + IMG) = lo;
                if constexpr (XWB) {
                    // the transformed tensor is the new residual stream.  Every strip writes all the pieces it staged: the halo columns and
                    // rows are its neighbours' too, who write the same bits there (nobody reads xf_out during this launch)
                    float* o = ok ? p.xf_out + (unsigned)(xoff[it] + r * p.xy_sH) : (float*)p.zero;
                    *(float4*)o = make_float4(xa[0], xa[1], xa[2], xa[3]); *(float4*)(o + 4) = make_float4(xa[4], xa[5], xa[6], xa[7]);
                }
            }
        };
        auto xf_write = [&](int r, int slot, int it, int part) { xf_write_from(r, slot, it, part, xr[it], xs[XRES ? it : 0]); };
        if constexpr (XF != 0) {
            // rows h0 - 1, h0, h0 + 1 into slots 0 .. 2 (row h + 2 is fetched, transformed and written within step h: no load result is
            // carried over the loop's back-edge, where hipcc would drain vmcnt to 0 - the step's own stores included - at every step).
            // All nine pieces are fetched before the first is converted: one round trip instead of three (a 2-row item of the one-frame launch
            // is mostly this prologue)
            float4 pr[3][3][2], ps[XRES ? 3 : 1][3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int it = 0; it < 3; ++it) xf_load_to(h0 - 1 + i, it, pr[i][it], ps[XRES ? i : 0][it]);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int it = 0; it < 3; ++it) xf_write_from(h0 - 1 + i, i, it, -1, pr[i][it], ps[XRES ? i : 0][it]);
        } else {
            // row h0 - 1 + i lives in slot i (mod RING) of this item
#pragma unroll
            for (int i = 0; i <= K; ++i) stage_row(h0 - 1 + i, i, i == 0 ? K : i - 1);       // residual rows h0 .. h0 + K - 1 in slots 0 .. K - 1 (row h0 - 1: dummy)
        }
        TLS(0);
        __syncthreads();
        TLS(1);

        // output of the previous step, held across the barrier and stored while the next step computes
        float ov[2][8]; h8_t ou[2];
        bool have = false;
        int hprev = 0;
        auto store_piece = [&](int k) {          // store k of row hprev: compile-time after unrolling, 0 .. NS - 1
            constexpr int PC = NS / 2;           // stores per column
            const int c = k / PC, w = k % PC;
            const unsigned o = nb_o0 + (unsigned)(hprev * p.o0_sH + c * p.o0_sW) + lane_el;
            if constexpr (EPI != V_EPI_F16) {
                if (w == 0) *(float4*)((float*)p.out0 + o) = make_float4(ov[c][0], ov[c][1], ov[c][2], ov[c][3]);
                if (w == 1) *(float4*)((float*)p.out0 + o + 4) = make_float4(ov[c][4], ov[c][5], ov[c][6], ov[c][7]);
                if constexpr (EPI == V_EPI_RES) {
                    if (w == 2) *(h8_t*)(p.out1 + (nb_o1 + (unsigned)(hprev * p.o1_sH + c * p.o1_sW) + lane_el)) = ou[c];
                }
            } else {
                h8_t x;
#pragma unroll
                for (int r = 0; r < 8; ++r) x[r] = ep_h(ov[c][r]);
                *(h8_t*)((half_t*)p.out0 + o) = x;
            }
        };
        auto flush = [&]() {                     // stores of row hprev: NS instructions
#pragma unroll
            for (int k = 0; k < NS; ++k) store_piece(k);
        };
        float st_s[8], st_q[8];                  // V_EPI_STAT: running partial sums of this lane's 8 channels
#pragma unroll
        for (int r = 0; r < 8; ++r) { st_s[r] = 0.f; st_q[r] = 0.f; }

        int s0 = 0, r0 = 0;                      // ring slot of input row h - 1; residual slot of row h
        for (int h = h0; h < h1; ++h) {
            // input row h + K into the slot row h - 2 left at the last barrier; residual row h + K into the slot this wave read last step
            int sk = s0 + K + 1; sk -= sk >= RING ? RING : 0;
            int rk = r0 + K; rk -= rk >= R::RROWS ? R::RROWS : 0;
#ifdef V32_NOSPREAD
            stage_row(h + K, sk, rk);
            asm volatile("" ::: "memory");
            TLS(2);
            if (have) flush();
            asm volatile("" ::: "memory");
            TLS(3);
#endif
            f4_t rr[2][2];                           // residual of this step's outputs, from LDS (landed at least two steps ago)
            if constexpr (EPI == V_EPI_RES) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i) rr[c][i] = *(const f4_t*)(wlo + r0 * R::RSLOT + wave * 4096 + (c * 2 + i) * 1024);
            }
            // ---- main loop: per pass 9 (kd, kh) groups x 3 kw; the four input columns of a group serve both output columns.
            // Passes (SPLIT): W_hi x_hi, W_lo x_hi, W_hi x_lo.
            int rb[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) { int sl = s0 + kh; sl -= sl >= RING ? RING : 0; rb[kh] = lanebase + sl * V_RS; }
            f4_t acc[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) acc[c][ci] = (f4_t){0.f, 0.f, 0.f, 0.f};
            constexpr int NG = SPLIT ? 27 : 9;
            constexpr int XL0 = 4, XW0 = 17;       // transform staging: first load group, first conversion group (of NG = 27)
            h8_t fa[4], fb[4];
            h8_t wa[SPLIT ? 3 : 1][2], wb[SPLIT ? 3 : 1][2];           // W_lo fragments of a group (pass 1)
            // group sequence.  Plain: 9 (kd, kh) groups.  Split precision (the halo kernel's chunk order): three passes of 9
            // groups - W_hi x_hi, W_lo x_hi, W_hi x_lo.  Merged order (-DV32_MERGE): (W_hi x_hi, W_lo x_hi) alternate per group on the SAME four activation
            // fragments, then the W_hi x_lo pass - 126 instead of 162 LDS reads per step; the pass that streams W_lo from LDS drops from 10 to
            // 6 reads per 12 MFMAs.  Another (fixed) summation order per output element than conv_halo's hilo kernel: equal to ~1e-7.
#ifdef V32_MERGE         /* A/B switch: measured 467.7 vs 467.2 frames/s (profiles/r03_l_ab.txt) - not worth another summation order */
            constexpr bool MERGE = SPLIT;
#else
            constexpr bool MERGE = false;
#endif
            auto g_pass = [](int G2) { return MERGE ? (G2 < 18 ? (G2 & 1) : 2) : G2 / 9; };
            auto g_idx = [](int G2) { return MERGE ? (G2 < 18 ? (G2 >> 1) : G2 - 18) : G2 % 9; };
            auto g_buf = [](int G2) { return MERGE ? (G2 < 18 ? ((G2 >> 1) & 1) : ((G2 - 17) & 1)) : (G2 & 1); };
            auto g_newb = [](int G2) { return MERGE ? !(G2 < 18 && (G2 & 1)) : true; };
            auto rd = [&](h8_t (&f)[4], h8_t (&wl2)[SPLIT ? 3 : 1][2], int G2) {      // G2: compile-time after unrolling
                const int pass = g_pass(G2), g = g_idx(G2), kd = g / 3, kh = g % 3;
                const int img = (SPLIT && pass == 2) ? IMG : 0;
                if (g_newb(G2)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) f[i] = *(const h8_t*)(smem + rb[kh] + img + i * V_CS + kd * 16);
                }
                if constexpr (SPLIT) {
                    if (pass == 1) {
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                            for (int ci = 0; ci < 2; ++ci) wl2[kw][ci] = *(const h8_t*)(wlo + ((g * 3 + kw) * 2 + ci) * 1024);
                    }
                }
            };
            rd(fa, wa, 0);
#pragma unroll
            for (int G2 = 0; G2 < NG; ++G2) {
                h8_t (&cur)[4] = g_buf(G2) ? fb : fa;
                h8_t (&wcur)[SPLIT ? 3 : 1][2] = (G2 & 1) ? wb : wa;
                h8_t (&wnxt)[SPLIT ? 3 : 1][2] = (G2 & 1) ? wa : wb;
#ifndef V32_XF_NOIL
                // transform staging: the conversion of piece G2 / 2 of row h + 2 (VALU + two ds_writes) shares the scheduling region of this
                // group's MFMAs and is spread between them (below)
                if constexpr (XF != 0) { if (G2 >= XW0 && G2 < XW0 + 9) xf_write(h + 2, sk, (G2 - XW0) / 3, (G2 - XW0) % 3); }
#endif
                if (G2 + 1 < NG) {
                    h8_t (&nxt)[4] = g_buf(G2 + 1) ? fb : fa;
                    rd(nxt, wnxt, G2 + 1);
                }
#ifdef V32_NOINTERLEAVE
                __builtin_amdgcn_sched_barrier(0);
#endif
                const int pass = g_pass(G2), g = g_idx(G2);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci) {
                            const h8_t a = (SPLIT && pass == 1) ? wcur[SPLIT ? kw : 0][ci] : wr[g * 3 + kw][ci];
                            acc[c][ci] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, cur[kw + c], acc[c][ci], 0, 0, 0);
                        }
#ifndef V32_NOINTERLEAVE
                // the LDS reads of the next group go out BETWEEN this group's 12 MFMAs (one wave per SIMD: a block of reads ahead of the
                // MFMAs is 30 - 80 cycles of idle matrix pipe per group): one read per MFMA until they are out, then the remaining MFMAs
                if (G2 + 1 < NG) {
                    const int nrd = (g_newb(G2 + 1) ? 4 : 0) + ((SPLIT && g_pass(G2 + 1) == 1) ? 6 : 0);
#ifndef V32_XF_NOIL
                    const bool conv_here = XF != 0 && G2 >= XW0 && G2 < XW0 + 9;
#else
                    const bool conv_here = false;
#endif
#pragma unroll
                    for (int i = 0; i < 11; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                        if (i < nrd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one DS read
                        if (conv_here) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // a few VALU instructions of the conversion
                    }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
#ifndef V32_NOSPREAD
                // the step's memory instructions (DMA pieces of row h + K first, then the stores of row h - 1: the order the counted wait
                // assumes) are spread over the MFMA groups: a wave that issues them back to back waits for the address pipeline to take
                // each one (150 - 350 cycles apiece at the head of a step) with nothing else to run; spread out, the residual-conv and
                // split-precision kernels lose 18 % / 11 % of their cycles (profiles/r03_d_vol32_probe_spread.txt; -DV32_NOSPREAD is the A/B switch)
                if constexpr (XF == 0) {
                    constexpr int NM = ND + NS;
#pragma unroll
                    for (int m = G2 * NM / NG; m < (G2 + 1) * NM / NG; ++m) {
                        asm volatile("" ::: "memory");
                        if (m < ND) stage_piece(h + K, sk, rk, m);
                        else if (have) store_piece(m - ND);
                        asm volatile("" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    // transform staging: the stores of row h - 1 first (groups 0 .. NS - 1), then the loads of row h + 2 (groups XL0, +2, +4), its
                    // conversion into slot sk late in the step (groups XW0 .. XW0 + 8, three parts per piece): the only memory instructions
                    // younger than a piece's loads at its conversion are the other pieces' loads and write-backs - the compiler counts them
#ifdef V32_XF_NOIL
                    if (G2 >= XW0 && G2 < XW0 + 9) xf_write(h + 2, sk, (G2 - XW0) / 3, (G2 - XW0) % 3);
#endif
                    if (G2 < NS && have) store_piece(G2);
                    if (G2 >= XL0 && G2 < XL0 + 6 && ((G2 - XL0) & 1) == 0) xf_load(h + 2, (G2 - XL0) >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            }
            TLS(4);
            // ---- epilogue arithmetic (conv_epilogue.h formulas, same order of operations); the stores follow after the barrier
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float v = acc[c][r >> 2][r & 3] + bias_v[r];
                    v = lin_act(v, p.sl0);
                    if constexpr (EPI == V_EPI_RES) v += rr[c][r >> 2][r & 3];
                    ov[c][r] = v;
                    if constexpr (EPI == V_EPI_STAT) { st_s[r] += v; st_q[r] = fmaf(v, v, st_q[r]); }
                    if constexpr (EPI == V_EPI_RES) {
                        const float a = v * s2_v[r] + t2_v[r];
                        ou[c][r] = ep_h(lin_act(a, p.sl1));
                    }
                }
            }
            if constexpr (EPI == V_EPI_STAT) {
                if (((h + 1) % p.srows) == 0 || h + 1 == h1) {      // the block (column pair, rows [8k, 8k + 8)) is complete: fixed-order butterfly
                    const int blk = (h / p.srows) * (p.W / 2) + (w0 / 2 + wave);
                    float pa[8], pb[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        float a = st_s[r], b = st_q[r];
                        st_s[r] = 0.f; st_q[r] = 0.f;
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                        pa[r] = a; pb[r] = b;
                    }
                    if (l15 == 0) {              // 8 channels x (sum, sum of squares) = 64 contiguous bytes: four 16-byte stores
                        float4* dst = (float4*)(p.stat_out + (((long)n * p.stat_nblk + blk) * 32 + l4 * 8) * 2);
#pragma unroll
                        for (int r = 0; r < 4; ++r) dst[r] = make_float4(pa[2 * r], pb[2 * r], pa[2 * r + 1], pb[2 * r + 1]);
                    }
                }
            }
            have = true; hprev = h;
            s0 = s0 + 1 == RING ? 0 : s0 + 1;
            r0 = r0 + 1 == R::RROWS ? 0 : r0 + 1;
            // row h + 2 has landed in this wave's part of the ring (counted wait: younger stores / DMA stay in flight); after the barrier in
            // everyone's, and everyone is done reading row h - 1
            TLS(5);
            if constexpr (XF != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's ds_writes of row h + 2 are in the LDS
            else if (h > h0 + K - 2) wait_vm<NW>();
            else wait_vm<NW0>();
            TLS(6);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            TLS(7);
#ifdef V32_TL
            tl_acc[9] += 1;
#endif
        }
        if (have) flush();
        __syncthreads();                         // the next item's prologue overwrites the ring
        TLS(8);
    }
#ifdef V32_TL
    if (p.tl && lane == 0) {
        const long wi = (long)blockIdx.x * 4 + wave;
        if (wi < p.tl_cap) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned long long* o = p.tl + wi * 12;
#pragma unroll
            for (int i = 0; i < 10; ++i) o[i] = tl_acc[i];
            o[10] = hwid; o[11] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

int g_ncu = 0;
#ifdef V32_TL
unsigned long long* g_v32_tl = nullptr;
long g_v32_cap = 0;
#endif

}  // namespace

#ifdef V32_TL
extern "C" void cs_debug_set_vol32_tl(void* buf, long cap) { g_v32_tl = (unsigned long long*)buf; g_v32_cap = cap; }
#endif

// Does the convolution described by p have the shape this kernel is built for?
//   plain:  Cin 32, input voxel = 32 channels;  split precision (ConvParams::hilo): Cin 96 = three weight chunks, input voxel = [hi | lo]
bool vol32_supported(const ConvParams& p)
{
    if (p.KD != 3 || p.KH != 3 || p.KW != 3 || p.Cout_pad != 32 || p.Cout != 32 || p.D != 16 || p.inD != 16 || (p.W % V_TW) != 0) return false;
    if (p.xf_kind) {     // transform staging: fp32 source volumes instead of `in`
        if (!p.hilo || p.Cin != 96 || !p.stat_out || (p.xf_kind != 1 && p.xf_kind != 2) || !p.xf_y.p || p.xf_y.sD != 32) return false;
        if (p.xf_kind == 2 && (!p.xf_stats || !p.xf_gamma || !p.xf_beta)) return false;
        if (p.xf_res.p && (p.xf_kind != 2 || p.xf_res.sD != 32)) return false;
        if (p.xf_out.p && (p.xf_kind != 2 || p.xf_out.sD != 32)) return false;
    } else if (p.hilo ? (p.Cin != 96 || p.in_sD != 64) : (p.Cin != 32 || p.in_sD != 32)) return false;
    if (p.up_shift || p.cg || p.wslot || p.sk_out || p.ragged || p.pixscale || p.stats || p.pool_hw || p.spmul) return false;
    if (p.act0 > ACT_LRELU || p.act1 > ACT_LRELU) return false;
    if (p.res.p && (!p.res_f32 || p.res.sD != 32)) return false;
    if (p.out0.p && p.out0.sD != 32) return false;
    if (p.out1.p && p.out1.sD != 32) return false;
    // the three epilogue forms on the path (V_EPI_*)
    if (!p.out0.p) return false;
    if (p.hilo && !p.stat_out) return false;
    if (p.stat_out) return p.out0_f32 && !p.res.p && !p.out1.p;
    if (p.res.p) return p.out0_f32 && p.out1.p && !p.hilo;
    return !p.out0_f32 && !p.out1.p;
}

// partial-statistics blocks per sample of a vol32 launch with ConvParams::stat_out
static int vol32_srows(const ConvParams& p) { return p.v32_srows > 0 ? p.v32_srows : V_SROWS; }
int vol32_stat_nblk(const ConvParams& p) { const int sr = vol32_srows(p); return ((p.H + sr - 1) / sr) * (p.W / 2); }

int launch_vol32(const ConvParams& p, hipStream_t st)
{
    if (!vol32_supported(p)) { cs_set_error("vol32: not a 3x3x3 32 -> 32 convolution on a [N][H][W][16][32] volume this kernel supports"); return -1; }
    if (ep_check_extents(p, "vol32")) return -1;
    const long in_span = (long)(p.N - 1) * p.in_sN + (long)(p.H - 1) * p.in_sH + (long)(p.W - 1) * p.in_sW + 1024;
    if (!p.xf_kind && in_span >= (1L << 31)) { cs_set_error("vol32: the input spans 2^31 elements or more"); return -1; }
    if (!g_ncu) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        g_ncu = n;
    }
    Vol32Params k;
    k.in = p.in; k.zero = p.zero; k.in_sN = (int)p.in_sN; k.in_sH = (int)p.in_sH; k.in_sW = (int)p.in_sW;
    k.N = p.N; k.H = p.H; k.W = p.W;
    k.wgt = p.wgt; k.bias = p.bias;
    k.sl0 = p.act0 == ACT_NONE ? 1.f : (p.act0 == ACT_LRELU ? p.slope0 : 0.f);
    k.sl1 = p.act1 == ACT_NONE ? 1.f : (p.act1 == ACT_LRELU ? p.slope1 : 0.f);
    k.res = (const float*)p.res.p; k.res_sN = (int)p.res.sN; k.res_sH = (int)p.res.sH; k.res_sW = (int)p.res.sW;
    k.out0 = p.out0.p; k.o0_sN = (int)p.out0.sN; k.o0_sH = (int)p.out0.sH; k.o0_sW = (int)p.out0.sW;
    k.out1 = (half_t*)p.out1.p; k.o1_sN = (int)p.out1.sN; k.o1_sH = (int)p.out1.sH; k.o1_sW = (int)p.out1.sW;
    k.s2 = p.s2; k.t2 = p.t2;
    k.stat_out = p.stat_out; k.stat_nblk = vol32_stat_nblk(p); k.srows = vol32_srows(p);
    k.xf_y = (const float*)p.xf_y.p; k.xy_sN = (int)p.xf_y.sN; k.xy_sH = (int)p.xf_y.sH; k.xy_sW = (int)p.xf_y.sW;
    k.xf_res = (const float*)p.xf_res.p; k.xr_sN = (int)p.xf_res.sN; k.xr_sH = (int)p.xf_res.sH; k.xr_sW = (int)p.xf_res.sW;
    k.xf_out = (float*)p.xf_out.p; k.xo_sN = (int)p.xf_out.sN; k.xo_sH = (int)p.xf_out.sH; k.xo_sW = (int)p.xf_out.sW;
    k.xf_stats = p.xf_stats; k.xf_gamma = p.xf_gamma; k.xf_beta = p.xf_beta; k.xf_slope = p.xf_slope;
    if (p.xf_kind) {
        auto same = [&](const TDesc& t) { return !t.p || (t.sN == p.xf_y.sN && t.sH == p.xf_y.sH && t.sW == p.xf_y.sW); };
        if (!same(p.xf_res) || !same(p.xf_out)) { cs_set_error("vol32: the transform-staging volumes must share their strides"); return -1; }
        const long lim = 1L << 31;
        auto span = [&](const TDesc& t) { return (long)(p.N - 1) * t.sN + (long)(p.H - 1) * t.sH + (long)(p.W - 1) * t.sW + 512; };
        if (span(p.xf_y) >= lim || (p.xf_res.p && span(p.xf_res) >= lim) || (p.xf_out.p && span(p.xf_out) >= lim)) {
            cs_set_error("vol32: a transform-staging tensor spans 2^31 elements or more"); return -1;
        }
    }
#ifdef V32_TL
    k.tl = g_v32_tl; k.tl_cap = g_v32_cap;
#endif
    k.nstrips = p.W / V_TW;
    // rows per item: whole strips when they fill the chip, else the strips are cut along H (a segment re-stages two halo rows, it never
    // recomputes anything) down to 8 rows (= one statistics block).  Per output element nothing depends on the decomposition.
    int nseg = 1;
    while ((long)p.N * k.nstrips * nseg < g_ncu && (p.H % (nseg * 2 * k.srows)) == 0) nseg *= 2;
    k.seg_rows = (p.H + nseg - 1) / nseg;
    k.nseg = (p.H + k.seg_rows - 1) / k.seg_rows;
    k.items = p.N * k.nstrips * k.nseg;
    int grid = k.items < g_ncu ? k.items : g_ncu;
    if (grid >= 8) grid &= ~7;
    const bool split = p.hilo != 0;
    const int epi = p.stat_out ? V_EPI_STAT : (p.res.p ? V_EPI_RES : V_EPI_F16);
    auto go = [&](auto kern, size_t lds) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { cs_set_error("vol32: opting into %zu bytes of LDS failed: %s", lds, hipGetErrorString(e)); return -1; }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, k);
        return 0;
    };
    int r = -1;
    if (split) {
        if (epi != V_EPI_STAT) { cs_set_error("vol32: the split-precision kernel exists with the statistics epilogue only"); return -1; }
        if (p.xf_kind == 2) {       // compile-time variants: + residual, + write-back (no run-time test inside the K loop)
            const int v = (p.xf_res.p ? 1 : 0) | (p.xf_out.p ? 2 : 0);
            if (v == 0) r = go(vol32_kernel<V_EPI_STAT, true, 2>, VRing<true, V_EPI_STAT>::lds(V_EPI_STAT));
            else if (v == 1) r = go(vol32_kernel<V_EPI_STAT, true, 3>, VRing<true, V_EPI_STAT>::lds(V_EPI_STAT));
            else if (v == 2) r = go(vol32_kernel<V_EPI_STAT, true, 4>, VRing<true, V_EPI_STAT>::lds(V_EPI_STAT));
            else r = go(vol32_kernel<V_EPI_STAT, true, 5>, VRing<true, V_EPI_STAT>::lds(V_EPI_STAT));
        }
        else if (p.xf_kind == 1) r = go(vol32_kernel<V_EPI_STAT, true, 1>, VRing<true, V_EPI_STAT>::lds(V_EPI_STAT));
        else r = go(vol32_kernel<V_EPI_STAT, true, 0>, VRing<true, V_EPI_STAT>::lds(V_EPI_STAT));
    } else {
        if (epi == V_EPI_STAT) r = go(vol32_kernel<V_EPI_STAT, false, 0>, VRing<false, V_EPI_STAT>::lds(V_EPI_STAT));
        else if (epi == V_EPI_RES) r = go(vol32_kernel<V_EPI_RES, false, 0>, VRing<false, V_EPI_RES>::lds(V_EPI_RES));
        else r = go(vol32_kernel<V_EPI_F16, false, 0>, VRing<false, V_EPI_F16>::lds(V_EPI_F16));
    }
    if (r) return r;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cs_set_error("vol32 launch: %s", hipGetErrorString(e)); return -1; }
    return 0;
}
