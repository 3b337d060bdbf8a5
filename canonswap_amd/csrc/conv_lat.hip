// Single-frame (latency mode) 3x3 convolution for the 512-channel 2-D layers at 64 x 64, gfx950 (CDNA4).
//
// One frame gives these layers - the 14 T blend convs (adaptive_modulate.py:128-193, 337-349), G's twelve 512 -> 512 convs of G_middle and
// up_0 (util.py:305-344), R's 2-D pair (util.py:120-128), W.third (warping_network.py:64-71) - 4096 positions: as 16 x 8 position tiles x
// 64 (128 packed) output channels that is exactly 256 workgroups, one per CU, and conv_halo's four-wave workgroup then runs ONE wave per
// SIMD through 144 dependent K-steps of 8 - 16 MFMAs: the launch is bound by the latencies of that chain (LDS fragment reads, weight
// fragments from L2 / HBM, a barrier per chunk), 37 - 44 us for 8 - 15 us of MFMA work (profiles/r05_d_layers_b1_lat.txt).  conv_wide's
// persistent 256 x 256 tiles need 512 items and have 32 - 64 here.  This kernel keeps the tile and splits its K loop INSIDE the workgroup:
//
//   * 12 WAVES, THREE PER SIMD, EVERY WAVE 128 POSITIONS x 32 ROWS.  The K-steps of a 64-channel chunk (3 x 3 taps x 2 halves of 32 channels) are
//     dealt to K-groups: T blend (128 packed rows per workgroup) has three groups of four waves, group g = kernel row kh (6 K-steps per chunk
//     and wave); the 64-channel layers have six groups of two waves, group g = (kh, half) (3 K-steps).  All groups read the same LDS halo of the
//     chunk (18 x 10 voxels, staged once, double buffered); each SIMD interleaves three independent MFMA chains.  A wave issues 8 LDS reads +
//     2 weight fragments per 16 MFMAs - the operand mix of conv_halo's 128 x 128 tile: the CU's vector-memory path delivers about 46 bytes per
//     clock of L2 hits, and waves of 4 x 4 / 4 x 2 fragments (this kernel's first form: twice the weight bytes per MFMA) ran at its rate,
//     5 000 / 10 300 cycles per chunk for 2 304 / 4 608 cycles of MFMA issue per SIMD (tools/lat_probe.py, profiles/r05_h_lat_probe.txt).
//   * THE HALO DMA IS HIDDEN FROM THE COMPILER, THE WEIGHT RING IS NOT.  hipcc turns every wait for a register load into vmcnt(0) while it
//     knows an LDS DMA to be pending (DESIGN 5.6 rule 1); twelve barrier-locked waves would all sit through the next chunk's halo round
//     trip at the head of every chunk.  conv_wide hides the weight loads (inline asm, hand-counted waits); here it is the DMA that goes
//     through inline asm (m0 + buffer_load_dwordx4 ... lds), issued at the head of a chunk for the next one, and the weight fragments stay
//     ordinary loads whose waits the compiler counts in the fully unrolled K loop.  vmcnt retires in order, so the compiler's counts -
//     which do not include the DMA pieces - can only wait longer than needed, never shorter; with a ring of RS = 3 K-steps a wait first
//     reaches behind the DMA three steps (of three waves per SIMD) after it was issued, by when it has landed.  The chunk head waits for
//     the DMA with a counted vmcnt((RS - 1) x WCH): everything but the ring's youngest RS - 1 steps - the DMA pieces are older than all
//     RS x WCH reloads of the chunk before.
//   * REDUCTION THROUGH LDS in a fixed order ((k0 + k1) + k2 over the kernel rows; the halves of a row first where they are split), into the
//     waves of kernel rows 0 and 1, which then own tile rows 0-3 / 4-7 (4 x 2 fragments) and run the shared epilogue (conv_epilogue.h).
//
// Another summation order per output element than conv_halo / conv_wide (K-group -> chunk -> kw [-> half] instead of chunk -> tap -> half):
// equal to ~1e-7 relative, not bit for bit - which is why the engine only takes this kernel behind cs_set_latency_mode, like split-K.
// Statistics partials: per 64 positions in the 16 x 8 tiles' order, as on every kernel these layers run on.
#define EP_GSEL_V EP_GSEL_K      /* position blocks whose residual / mask the epilogue fetches per round: set per instantiation below */
#include "conv_epilogue.h"
#include <cstdlib>

namespace {


constexpr int L_SLP = 10, L_VS = L_SLP * 16;          // slots per voxel (8 data + 2 pad: conflict-free fragment reads, tools/lds_bank_search.py), bytes per voxel
constexpr int L_HW = 18, L_HH = 10, L_HV = L_HW * L_HH;
constexpr int L_NPIECE = L_HV * L_SLP;                // 1800 16-byte pieces per chunk
constexpr int L_NW = 12;                              // waves per workgroup
constexpr int L_HI = 3;                               // DMA instructions per wave and chunk: 36 x 1 KiB >= 28 800 bytes
constexpr int L_BSTRIDE = L_HI * L_NW * 1024;         // bytes between the two halo buffers (the last instructions' lanes beyond the image land in the gap)
constexpr int L_NCK = 8;                              // 64-channel chunks: Cin = 512
constexpr unsigned L_OOB = 0x80000000u;               // byte offset outside the buffer: the lane reads zeros

typedef int l_i4_t __attribute__((ext_vector_type(4)));
typedef unsigned int l_u4_t __attribute__((ext_vector_type(4)));

// one LDS-DMA instruction the compiler does not know of: 64 lanes x 16 bytes -> LDS at m0v + lane * 16.  m0 cannot be
// named as a clobber (hipcc: "reserved register, may not be preserved" - the clobber is ignored with a warning per expansion), so the guard is in
// the build: _lib._isa_check_lat fails on any instruction of the kernel that names m0 outside this three-instruction sequence
__device__ __forceinline__ void lat_dma16(unsigned m0v, unsigned voff, l_i4_t rsrc, int soff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// -DLAT_TL (tools/lat_probe.py, instrumented A/B build only): every wave accumulates s_memtime cycles per phase of its life
#ifdef LAT_TL
__device__ unsigned long long* g_lat_tl = nullptr;
__device__ long g_lat_tl_cap = 0;
#define LTL(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_acc[i] += t_ - tl_last; tl_last = t_; \
                    __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LTL(i) do { } while (0)
#endif
template <int N> __device__ __forceinline__ void lat_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NG: K-groups of the workgroup.  3 (T blend, 128 packed rows per workgroup): group g multiplies kernel row kh = g, 6 K-steps (kw x 32-channel half) per
// chunk, four waves per group (32 packed rows each).  6 (64 output channels per workgroup): group g multiplies (kh = g / 2, half = g % 2), 3 K-steps
// (kw) per chunk, two waves per group.  Every wave owns ALL 128 positions of the tile x 32 rows: 8 x 2 fragments, 8 LDS reads + 2 weight fragments per
// 16 MFMAs - the CU's vector-memory path (about 46 B / clk of L2 hits) carries 12 waves x 2 KB per K-step; with 4 x 4 / 4 x 2 fragment waves (the first
// form of this kernel) it carried twice that and was the bound (tools/lat_probe.py: 5 000 / 10 300 cycles per chunk for 2 304 / 4 608 of MFMA issue).
// EPC: the tensor combination of the launch as a compile-time constant (EP_CODE of conv_epilogue.h): one straight-line copy of the epilogue
template <int MODE, int EPC, int NG>
__global__ void __launch_bounds__(768, 1) conv_lat_kernel(const ConvParams p)
{
    constexpr int WPX = 8, WCH = 2, WVC = L_NW / NG, BM = 128, BN = WCH * 16 * WVC;
    constexpr int SPC = 18 / NG;                       // K-steps per chunk and wave
    constexpr int RS = 3;                              // weight ring depth in K-steps
    constexpr int EP_PAIR = ep_pair_of(MODE, WCH);
    static_assert(NG == 3 || NG == 6, "kernel rows, or kernel rows x 32-channel halves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / WVC, wch = wave % WVC;      // K-group; channel slice (packed rows wch * 32 ..)
    const int kh = NG == 3 ? grp : grp >> 1, khalf = NG == 3 ? 0 : grp & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int l15p = l15;
#ifdef LAT_TL
    unsigned long long tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime();      // [startup, head wait, barrier, stage, K-steps, reduction, epilogue]
#endif

    // ---- workgroup -> (tile, channel block): XCD x (= blockIdx.x % 8) owns a contiguous range of the launch order, the channel blocks of a tile adjacent
    int tile_lin, cblk;
    {
        const int total = (int)gridDim.x, xcd = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        const int u = xcd * q + (xcd < r ? xcd : r) + i;
        const int ncb = p.Cout_pad / BN;
        tile_lin = u / ncb; cblk = u - tile_lin * ncb;
    }
    int t = tile_lin;
    const int tw = t % p.nTW; t /= p.nTW;
    const int th = t % p.nTH; t /= p.nTH;
    const int tn = t, td = 0;
    const int n0 = cblk * BN;
    constexpr int lgTW = 4, lgTH = 3, lgTD = 0, lgS = 7, mW = 15, mH = 7, mD = 0;

    // ---- halo staging: DMA instruction (j, wave) writes the 1 KiB of pieces q = ((j * 12 + wave) * 64 + lane); piece q <-> (voxel q / 10, slot q % 10)
    l_i4_t rsrc;
    {
        const unsigned long long base = (unsigned long long)(p.in + (long)tn * p.in_sN);
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        rsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)p.in_sample_bytes);
        rsrc[3] = 0x00020000;
    }
    unsigned poff[L_HI];
    {
        const int isH = (int)p.in_sH, isW = (int)p.in_sW;
#pragma unroll
        for (int j = 0; j < L_HI; ++j) {
            const int q = ((j * L_NW + wave) << 6) + lane;
            const int hv = q / L_SLP, sl = q - hv * L_SLP;
            const int hh = hv / L_HW, hw = hv - hh * L_HW;
            const int ih = th * 8 + hh - 1, iw = tw * 16 + hw - 1;
            const bool inb = q < L_NPIECE && sl < 8 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            poff[j] = inb ? (unsigned)(__mul24(ih, isH) + __mul24(iw, isW) + sl * 8) * 2u : L_OOB;
        }
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    auto stage = [&](int buf, int cc) {       // chunk cc -> buffer buf (asynchronous)
#pragma unroll
        for (int j = 0; j < L_HI; ++j) {
            const int slot = j * L_NW + wave;      // wave-uniform; the slots beyond the image (29 .. 35) write zeros into the gap behind it: no branch
            lat_dma16((unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(buf * L_BSTRIDE + slot * 1024))), poff[j], rsrc, cc * 128);
        }
    };

    // ---- operands.  Activation fragment pi of a wave = tile row pi, 16 positions along w (l15), k slot l4; K-step s of a chunk reads it at
    // hb + pi * 18 * 160 + kw * 160 [+ half * 64], hb including the kernel row (and, NG = 6, the half) of the group
    const int abase = (kh * L_HW + l15p) * L_VS + l4 * 16 + khalf * 64;
    const half_t* wbase = p.wgt;
    if (p.wslot) wbase += p.wofs[p.wslot[tn]];         // per-sample weight set (uniform over the tile)
    const half_t* wlane = wbase + ((long)(n0 + wch * WCH * 16) * 32 + ep_lane_row(EP_PAIR, l15) * 32 + l4 * 8);
    const long wstep = (long)p.Cout_pad * 32;
    // packed K-step index of (chunk cc, kernel row kh, column kw, half): ((cc * 2 + half) * 9 + kh * 3 + kw)
    const half_t* wrow = wlane + (long)(kh * 3 + khalf * 9) * wstep;
    auto step_kw = [](int s) { return NG == 3 ? s >> 1 : s; };
    auto step_half = [](int s) { return NG == 3 ? s & 1 : 0; };
    auto wload = [&](l_u4_t (&dst)[WCH], int cc, int s) {        // cc, s: compile-time after unrolling
        const half_t* src = wrow + (long)((cc * 2 + step_half(s)) * 9 + step_kw(s)) * wstep;
#pragma unroll
        for (int ci = 0; ci < WCH; ++ci) dst[ci] = *(const l_u4_t*)(src + ep_frag_row(EP_PAIR, ci) * 32);
    };

    f4_t acc[WCH][WPX];
#pragma unroll
    for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
        for (int pi = 0; pi < WPX; ++pi) acc[ci][pi] = (f4_t){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    l_u4_t wr[RS][WCH];
#pragma unroll
    for (int s = 0; s < RS; ++s) wload(wr[s], 0, s);
    LTL(0);

#pragma unroll
    for (int cc = 0; cc < L_NCK; ++cc) {
        // head of a chunk: this wave's DMA pieces of the chunk have landed (counted: only the ring's youngest RS - 1 steps may still be in
        // flight), everyone's after the barrier, which also says that everyone has left the other buffer
        lat_wait_vm<(RS - 1) * WCH>();
        LTL(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        LTL(2);
        if (cc + 1 < L_NCK) stage((cc + 1) & 1, cc + 1);
        LTL(3);
        const unsigned char* hb = smem + (cc & 1) * L_BSTRIDE + abase;
        auto aoff = [&](int pi, int s) { return pi * (L_HW * L_VS) + step_kw(s) * L_VS + step_half(s) * 64; };
        // position fragments in two halves: while the MFMAs of one half run, the LDS reads of the other half (of this step or the next) are in flight
        constexpr int HA = WPX / 2;
        h8_t afA[HA], afB[HA];
#pragma unroll
        for (int pi = 0; pi < HA; ++pi) afA[pi] = *(const h8_t*)(hb + aoff(pi, 0));
#pragma unroll
        for (int s = 0; s < SPC; ++s) {
#pragma unroll
            for (int pi = 0; pi < HA; ++pi) afB[pi] = *(const h8_t*)(hb + aoff(HA + pi, s));
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                for (int pi = 0; pi < HA; ++pi)
                    acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[s % RS][ci]), afA[pi], acc[ci][pi], 0, 0, 0);
            if (s + 1 < SPC) {
#pragma unroll
                for (int pi = 0; pi < HA; ++pi) afA[pi] = *(const h8_t*)(hb + aoff(pi, s + 1));
            }
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                for (int pi = 0; pi < HA; ++pi)
                    acc[ci][HA + pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wr[s % RS][ci]), afB[pi], acc[ci][HA + pi], 0, 0, 0);
            // the slot this step used serves step s + RS: of this chunk, or of the next one
            if (s + RS < SPC) wload(wr[s % RS], cc, s + RS);
            else if (cc + 1 < L_NCK) wload(wr[s % RS], cc + 1, s + RS - SPC);
            // the reload goes out HERE: left alone, hipcc sinks the ring's loads to the end of the chunk (nothing needs them earlier), where the
            // chunk-head wait then sits through their whole round trip (tools/lat_probe.py: 690 cycles per chunk)
            __builtin_amdgcn_sched_barrier(0);
        }
        LTL(4);
    }

    // ---- reduction over the K-groups through LDS (the halo region is free: every DMA was waited for at its chunk's head).  Fixed order per
    // output element: NG = 3: (k0 + k1) + k2 over the kernel rows; NG = 6: ((k0h0 + k0h1) + (k1h0 + k1h1)) + (k2h0 + k2h1).  The kernel-row
    // groups 0 and 1 end up owning tile rows 0-3 / 4-7 (4 x 2 fragments per wave) and run the epilogue.
    f4_t* red = (f4_t*)smem;
    __syncthreads();
    if constexpr (NG == 6) {
        // the half-1 groups hand everything to their half-0 partner: region [kh][slice][ci * 8 + pi]
        if (khalf) {
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                for (int pi = 0; pi < WPX; ++pi) red[(((kh * WVC + wch) * WCH + ci) * WPX + pi) * 64 + lane] = acc[ci][pi];
        }
        __syncthreads();
        if (!khalf) {
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                for (int pi = 0; pi < WPX; ++pi) {
                    const f4_t b = red[(((kh * WVC + wch) * WCH + ci) * WPX + pi) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ci][pi][r] += b[r];
                }
        }
        __syncthreads();
    }
    // kernel-row group `kh` (NG = 6: its half-0 waves) -> owner q (tile rows 4 q .. 4 q + 3): region [q][source index][slice][ci * 4 + pi'];
    // the sources of owner 0 are rows 1, 2 (index 0, 1), those of owner 1 rows 0, 2
    constexpr int HB = WPX / 2;
    const bool live = NG == 3 || !khalf;
    if (live) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (kh == q) continue;
            const int si = q == 0 ? kh - 1 : (kh == 0 ? 0 : 1);
#pragma unroll
            for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
                for (int pi = 0; pi < HB; ++pi) red[(((((q * 2 + si) * WVC + wch) * WCH + ci) * HB) + pi) * 64 + lane] = acc[ci][q * HB + pi];
        }
    }
    __syncthreads();
    LTL(5);
    if (live && kh < 2) {
        f4_t ep_acc[WCH][HB];
#pragma unroll
        for (int ci = 0; ci < WCH; ++ci)
#pragma unroll
            for (int pi = 0; pi < HB; ++pi) {
                const f4_t a = red[(((((kh * 2 + 0) * WVC + wch) * WCH + ci) * HB) + pi) * 64 + lane];
                const f4_t b = red[(((((kh * 2 + 1) * WVC + wch) * WCH + ci) * HB) + pi) * 64 + lane];
                const f4_t own = kh == 0 ? acc[ci][pi] : acc[ci][HB + pi];       // (kh is wave-uniform)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep_acc[ci][pi][r] = kh == 0 ? (own[r] + a[r]) + b[r] : (a[r] + own[r]) + b[r];
            }
        // ---- epilogue (conv_epilogue.h): this wave's 64 positions (tile rows 4 kh ..) x 32 packed rows
        constexpr bool EP_HEAVY = false, EP_EARLY = false;
        constexpr int EP_GSEL_K = 0;
        constexpr int EP_WPX = HB;
        const int ep_wpx = kh;
        ep_u2_t ep_xpre[1][1];
        (void)ep_xpre; (void)td;
        CONV_EPILOGUE_IMPL(EPC);
        LTL(6);
    }
#ifdef LAT_TL
    if (g_lat_tl && lane == 0) {
        const long wi = (long)blockIdx.x * 12 + wave;
        if (wi < g_lat_tl_cap) { for (int i = 0; i < 8; ++i) g_lat_tl[wi * 8 + i] = tl_acc[i]; }
    }
#endif
}

template <int MODE, int EPC, int NG>
int launch_lat_inst(const ConvParams& p, hipStream_t st)
{
    auto k = conv_lat_kernel<MODE, EPC, NG>;
    constexpr int WVC = L_NW / NG, BN = WVC * 32;
    size_t lds = 2 * (size_t)L_BSTRIDE;
    // reduction regions (1 KiB per wave fragment): NG = 6: [3][WVC][16]; then [2 owners][2 sources][WVC][8]
    const size_t red = (size_t)(NG == 6 ? 3 * WVC * 16 : 2 * 2 * WVC * 8) * 1024;
    if (red > lds) lds = red;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      // per launch: the attribute belongs to the current device
    if (e != hipSuccess) { cs_set_error("conv_lat: opting into %zu bytes of LDS failed: %s", lds, hipGetErrorString(e)); return -1; }
    const unsigned grid = (unsigned)(p.nTW * p.nTH * p.nTN * (p.Cout_pad / BN));
    hipLaunchKernelGGL(k, dim3(grid), dim3(768), lds, st, p);
    e = hipGetLastError();
    if (e != hipSuccess) { cs_set_error("conv_lat launch: %s", hipGetErrorString(e)); return -1; }
    return 0;
}

// the instantiations: (mode, tensor combination) of the engine's 512-channel 3x3 layers (the same list as conv_wide's, without SPADE)
//   T blend conv1 / conv2 (adaptive_modulate.py:337-349) | R's 2-D pair (util.py:120-128), W.third | G 3x3 convs with the next norm's statistics,
//   without / with the block's fp16 residual (util.py:329-344)
constexpr int LAT_NINST = 6;
constexpr int LAT_INST[LAT_NINST][2] = {
    {MODE_TBLEND, EP_CODE(0, 1, 0, 0, 1)}, {MODE_TBLEND, EP_CODE(2, 1, 1, 1, 1)},
    {MODE_STD, EP_CODE(0, 1, 0, 0, 0)}, {MODE_STD, EP_CODE(2, 1, 1, 1, 0)},
    {MODE_STDSTAT, EP_CODE(0, 1, 0, 0, 0)}, {MODE_STDSTAT, EP_CODE(1, 1, 0, 0, 0)},
};
int lat_inst_of(int mode, int code)
{
    for (int i = 0; i < LAT_NINST; ++i) if (LAT_INST[i][0] == mode && LAT_INST[i][1] == code) return i;
    return -1;
}
int lat_ep_code(const ConvParams& p)
{
    return EP_CODE(p.res.p ? (p.res_f32 ? 2 : 1) : 0, p.out0.p ? 1 : 0, p.out0_f32 ? 1 : 0, p.out1.p ? 1 : 0, p.pixscale ? 1 : 0);
}

}  // namespace

// Which launches this kernel takes: 3x3, 2-D, Cin = 512, 16 x 8 tiles within a sample, 64 (T blend: 128 packed) output channels per workgroup, every
// packed channel a real one, the tensor combinations of conv_epilogue.h for modes STD / STDSTAT / TBLEND with ReLU-family activations.
bool conv_lat_supported(const ConvParams& p, int mode)
{
    if (p.KD != 1 || p.KH != 3 || p.KW != 3 || p.D != 1 || p.inD != 1 || p.up_shift || p.cg || p.hilo || p.ragged || p.sk_out || p.kw_out || p.xf_kind || p.spmul ||
        p.pool_hw || p.xs_w || p.nphase || p.PH != 1 || p.PW != 1) return false;
    if (p.Cin != 64 * L_NCK || p.H % 8 || p.W % 16) return false;
    if (mode == MODE_STD && p.stat_out) mode = MODE_STDSTAT;
    if (mode != MODE_STD && mode != MODE_STDSTAT && mode != MODE_TBLEND) return false;
    if ((mode == MODE_STDSTAT) != (p.stat_out != nullptr)) return false;
    const int cstep = mode == MODE_TBLEND ? 2 : 1, bn = mode == MODE_TBLEND ? 128 : 64;
    if (p.Cout_pad % bn || p.Cout * cstep != p.Cout_pad || (p.Cout & 7)) return false;
    if (p.act0 >= ACT_SIGMOID || p.act1 >= ACT_SIGMOID) return false;
    if (((unsigned long long)p.in & 15ull) || ((p.in_sN | p.in_sH | p.in_sW) & 7)) return false;
    // a sample of the input is addressed with 31-bit byte offsets, its axis products are 24-bit multiplies
    if ((long)(p.H - 1) * p.in_sH + (long)(p.W - 1) * p.in_sW + p.Cin >= (1L << 30) || p.in_sH >= (1L << 23) || p.in_sW >= (1L << 23)) return false;
    if (p.ep_general) return false;
    auto al8 = [](const TDesc& t) { return (((unsigned long long)t.p & 15ull) == 0) && (((t.sN | t.sD | t.sH | t.sW) & 7) == 0); };
    if ((p.res.p && !p.res_f32 && !al8(p.res)) || (p.out0.p && !p.out0_f32 && !al8(p.out0)) || (p.out1.p && !al8(p.out1))) return false;
    return lat_inst_of(mode, lat_ep_code(p)) >= 0;
}

#ifdef LAT_TL
// instrumented builds only: device buffer of cap x 8 u64 that every following conv_lat launch fills (nullptr: off)
extern "C" void cs_debug_set_lat_tl(void* buf, long cap)
{
    unsigned long long* b = (unsigned long long*)buf;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lat_tl), &b, sizeof(b));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lat_tl_cap), &cap, sizeof(cap));
}
#endif

int launch_conv_lat(const ConvParams& p0, int mode, hipStream_t st)
{
    if (!conv_lat_supported(p0, mode)) { cs_set_error("conv_lat: this launch is not one of the kernel's shapes / tensor combinations"); return -1; }
    if (ep_check_extents(p0, "conv_lat")) return -1;
    ConvParams p = p0;
    p.lgTW = 4; p.lgTH = 3; p.lgTD = 0;
    p.nTW = p.W / 16; p.nTH = p.H / 8; p.nTD = 1; p.nTN = p.N;
    p.in_sample_bytes = (unsigned)(((long)(p.H - 1) * p.in_sH + (long)(p.W - 1) * p.in_sW + p.Cin) * 2);
    if (mode == MODE_STD && p.stat_out) mode = MODE_STDSTAT;
    switch (lat_inst_of(mode, lat_ep_code(p))) {
#define LAT_CASE(i) case i: return launch_lat_inst<LAT_INST[i][0], LAT_INST[i][1], LAT_INST[i][0] == MODE_TBLEND ? 3 : 6>(p, st);
        LAT_CASE(0) LAT_CASE(1) LAT_CASE(2) LAT_CASE(3) LAT_CASE(4) LAT_CASE(5)
#undef LAT_CASE
        default: break;
    }
    cs_set_error("conv_lat: no instantiation for mode %d / epilogue code %d", mode, lat_ep_code(p));
    return -1;
}
