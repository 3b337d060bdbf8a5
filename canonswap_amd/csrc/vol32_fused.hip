// One ResBlock3d of the feature volume as ONE kernel for gfx950 (CDNA4): out = conv2(relu(bn2(conv1(a)))) + x, a = relu(bn1(x))
// (util.py:80-102; the six blocks of the appearance feature extractor and the six of the identity-transfer module).
//
// As two launches of vol32.hip the block moved 1.14 GB per 32 frames and both were bound by their memory instructions (conv2 runs at
// 3.7 TB/s, 65 % of what a CU's share of HBM allows; profiles/r03_d_vol32_probe_spread.txt).  Fused, conv1's output never leaves the CU:
//
//   * a workgroup owns a strip of 8 columns and marches along H like vol32.hip.  Per step it computes one row of h = relu(conv1(a) + b1)
//     for 10 columns (one halo column each side) INTO AN LDS RING in the same conflict-free q-plane image the a-rows are staged in, and one
//     output row of conv2 from the three h-rows before it.  conv1 runs two rows ahead of conv2; one barrier per step.
//   * EIGHT WAVES, TWO PER SIMD, each with ONE 16-channel output fragment of ONE conv in registers (27 taps x 4 = 108 VGPRs):
//     wave = (conv, output-channel half, column half).  A SIMD hosts a conv1 wave (5 columns x 27 taps = 135 MFMAs per step) and a conv2
//     wave (4 x 27 = 108): while one waits for its DMA / store instructions to be taken by the address pipeline (150 cycles apiece at one
//     wave per SIMD) the other issues MFMAs.  An LDS fragment now feeds one MFMA per wave instead of two; the LDS runs at about half rate.
//   * the two waves of a SIMD run half a step out of phase (conv2: epilogue + stores, then MFMAs; conv1: MFMAs, then epilogue + DMA), so one
//     wave's VALU / memory phase lies under the other's MFMAs.
//   * per step and CU: 12 KB of a (fp16, DMA), 16 KB of x (fp32 residual, DMA into a ring read back by the lane that fetched it), 16 + 8 KB
//     of stores - what conv2 alone moved before.
//
// Per output element both convs accumulate their 27 taps in (kd, kh, kw) order, one 32-channel MFMA step each, h is rounded to fp16
// exactly where the two-launch path stores it, and the epilogue formulas are those of conv_epilogue.h: the fused block gives the same
// bits as conv1 -> conv2 on vol32.hip / conv_halo (tests/test_gpu_vol32.py).
#include "conv_epilogue.h"

namespace {

constexpr int F_TW = 8;                      // output columns of a strip
constexpr int F_ANC = F_TW + 4;              // a slab: columns w0 - 2 .. w0 + 9
constexpr int F_HNC = F_TW + 2;              // h slab: columns w0 - 1 .. w0 + 8
constexpr int F_CS = 17 * 16;                // bytes of one column in one q plane: zero slot + 16 voxels x 16 bytes
constexpr int F_ARS = F_ANC * F_CS, F_HRS = F_HNC * F_CS;
constexpr int F_KA = 2;                      // a rows staged ahead of conv1's row + 1
constexpr int F_ARING = F_KA + 3, F_HRING = 4, F_RRING = 3;
constexpr int F_AQS = ((F_ARING * F_ARS + 16 + 255) / 256) * 256;      // q plane strides: multiples of 256 bytes (bank-conflict freedom)
constexpr int F_HQS = ((F_HRING * F_HRS + 16 + 255) / 256) * 256;
constexpr int F_AIMG = 4 * F_AQS, F_HIMG = 4 * F_HQS;
constexpr int F_RSLOT = 4 * 4096;            // residual ring slot: 4 conv2 waves x 4 columns x 16 voxels x 16 channels fp32
constexpr int F_LDS = F_AIMG + F_HIMG + F_RRING * F_RSLOT;
static_assert(F_LDS <= 160 * 1024, "LDS budget");

struct FusedParams {
    const half_t* a; const half_t* zero; int a_sN, a_sH, a_sW;           // fp16 [N][H][W][16][32]
    const float* x; int x_sN, x_sH, x_sW;                                // fp32 residual stream
    float* out0; int o0_sN, o0_sH, o0_sW;                                // fp32 = conv2 + b2 + x
    half_t* out1; int o1_sN, o1_sH, o1_sW;                               // fp16 = act1(out0 * s2 + t2)
    const half_t* w1; const half_t* w2;                                  // packed [27][32][32]
    const float* b1; const float* b2; const float* s2; const float* t2;
    float sl1;                                                           // lin_act slope of act1
    int N, H, W;
    int nstrips, nseg, seg_rows, items;
#ifdef V32_TL
    unsigned long long* tl; long tl_cap;     // 12 x u64 per wave: [startup, prologue, compute (MFMA groups), memory instructions, epilogue, lgkm wait, vm wait, barrier, idle step, tail], steps, role
#endif
};

typedef unsigned int u4v __attribute__((ext_vector_type(4)));
#ifdef V32_TL      /* instrumented A/B build (tools/vol32_probe.py): cycles per phase of a wave's life */
#define FTL(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_acc[i] += t_ - tl_last; tl_last = t_; \
                    __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FTL(i) do { } while (0)
#endif
template <int N> __device__ __forceinline__ void f_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ void __launch_bounds__(512, 2) vol32_fused_kernel(const FusedParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const himg = smem + F_AIMG;
    unsigned char* const rring = smem + F_AIMG + F_HIMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2;              // 0: conv1 (waves 0-3), 1: conv2 (waves 4-7): waves w and w + 4 share a SIMD
    const int hf = (wave >> 1) & 1;          // output-channel half: channels hf * 16 .. + 15
    const int chf = wave & 1;                // column half
    const int l15 = lane & 15, l4 = lane >> 4;
#ifdef V32_TL
    unsigned long long tl_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime();
    unsigned long long tl_steps = 0;
#endif

    for (int i = tid * 16; i < F_AIMG + F_HIMG; i += 512 * 16) *(u4v*)(smem + i) = (u4v){0u, 0u, 0u, 0u};
    // this wave's A fragments: rows hf * 16 + l15 of the conv's 27 taps
    h8_t wr[27];
    {
        const half_t* wl = (role ? p.w2 : p.w1) + (hf * 16 + l15) * 32 + l4 * 8;
#pragma unroll
        for (int t = 0; t < 27; ++t) wr[t] = *(const h8_t*)(wl + t * 1024);
    }
    const int ch0 = hf * 16 + l4 * 4;        // first of this lane's 4 output channels
    float bias_v[4], s2_v[4], t2_v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float* b = role ? p.b2 : p.b1;
        bias_v[r] = b ? b[ch0 + r] : 0.f;
        s2_v[r] = (role && p.s2) ? p.s2[ch0 + r] : 1.f;
        t2_v[r] = (role && p.s2) ? p.t2[ch0 + r] : 0.f;
    }
    __syncthreads();

    // a-row staging by the conv1 waves (they have the fewer memory instructions): wave q stages q plane q, 204 slots (12 columns x 17) = 4 pieces
    int s_off[4]; bool s_on[4];
    const int sq = wave & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = j * 64 + lane, col = idx / 17, ds = idx % 17;
        s_on[j] = idx < F_ANC * 17 && ds != 0;
        s_off[j] = col * p.a_sW + (ds - 1) * 32 + sq * 8;
    }
    // fragment reads: lane (depth l15, q plane l4); conv1 reads a-slab columns 5 chf .. + 6, conv2 h-slab columns 4 chf .. + 5
    const int lanebase = role ? (l4 * F_HQS + l15 * 16 + (4 * chf) * F_CS) : (l4 * F_AQS + l15 * 16 + (5 * chf) * F_CS);
    // conv1's epilogue writes h (4 channels = 8 bytes of the 16-byte slot of channel group q) at depth l15 of h-slab columns 5 chf .. + 4
    const int hq = ch0 >> 3, hsub = (ch0 & 4) * 2;
    const int hw_base = hq * F_HQS + (l15 + 1) * 16 + hsub + (5 * chf) * F_CS;
    const unsigned lane_el = (unsigned)(l15 * 32 + ch0);            // element offset of this lane's channels inside a column

    const int G = (int)gridDim.x;
    int u = (int)blockIdx.x;
    if ((G & 7) == 0) u = (u & 7) * (G >> 3) + (u >> 3);

    for (int item = u; item < p.items; item += G) {
        int t = item;
        const int strip = t % p.nstrips; t /= p.nstrips;
        const int seg = t % p.nseg;
        const int n = t / p.nseg;
        const int w0 = strip * F_TW;
        const int h0 = seg * p.seg_rows, h1 = (h0 + p.seg_rows < p.H) ? h0 + p.seg_rows : p.H;
        bool c_ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c_ok[j] = s_on[j] && (unsigned)(w0 - 2 + (j * 64 + lane) / 17) < (unsigned)p.W;
        const half_t* a_n = p.a + (long)n * p.a_sN + (long)(w0 - 2) * p.a_sW;
        const int oc0 = w0 + 4 * chf;            // conv2 waves: first output column
        const unsigned nb_o0 = (unsigned)(n * p.o0_sN + oc0 * p.o0_sW), nb_o1 = (unsigned)(n * p.o1_sN + oc0 * p.o1_sW);
        const unsigned nb_x = (unsigned)(n * p.x_sN + oc0 * p.x_sW);
        // piece j (0 .. 3) of a row r -> a-ring slot; rows outside the volume or beyond this item's reach are zero rows
        auto stage_a = [&](int r, int slot, int j) {
            const bool rok = (unsigned)r < (unsigned)p.H && r <= h1 + 1;
            const half_t* rp = a_n + (long)r * p.a_sH;
            unsigned char* dst = smem + sq * F_AQS + slot * F_ARS;
            if (s_on[j]) {
                const half_t* src = (rok && c_ok[j]) ? rp + s_off[j] : p.zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
            }
        };
        // column c (0 .. 3) of residual row r for the conv2 wave with the same (channel half, column half) -> residual slot; lane-linear: the
        // conv2 lane (depth, channel group) reads back exactly the 16 bytes the conv1 lane of the same index fetched
        auto stage_x = [&](int r, int slot, int c) {
            const int rc = (unsigned)r < (unsigned)p.H ? r : 0;          // rows outside are never consumed
            const float* src = p.x + (nb_x + (unsigned)(rc * p.x_sH) + lane_el);
            unsigned char* dst = rring + slot * F_RSLOT + (wave & 3) * 4096;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * p.x_sW),
                                             (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
        };
        // prologue: a rows h0 - 2 .. h0 + 1 (slot of row r is (r - (h0 - 2)) mod 5); the first step stages row h0 + 2
        if (!role) {
#pragma unroll
            for (int i = 0; i < F_ARING - 1; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) stage_a(h0 - 2 + i, i, j);
        }
        FTL(0);
        __syncthreads();
        FTL(1);

        // The two waves of a SIMD run their steps HALF A STEP OUT OF PHASE: the conv2 wave opens a step with the epilogue and the 8 stores of
        // the output row it accumulated in the step before (its accumulators wait across the barrier) and then runs its MFMAs, the conv1 wave
        // opens with its MFMAs and closes with its epilogue (h row into LDS) and the step's 8 DMA instructions.  In lock step (both MFMA
        // phases, then both epilogues) the matrix pipe idled through the epilogues and memory instructions of both: 52 % busy, 225 us per
        // block (profiles/r03_i_vol32_fused_phases.txt).
        f4_t acc2[4];                            // conv2: accumulators of output row hprev, finished at the head of the next step
        bool pend = false;
        int hprev = 0;
        int sa = 0;                              // a-ring slot of row t - 1 (conv1 at step t reads a rows t - 1, t, t + 1)
        float ov[4][4]; ep_u2_t ou[4];           // conv2: finished values of row hrow, stored one instruction per MFMA group
        int hrow = 0;
        auto finish_math = [&]() {               // conv2: epilogue arithmetic of row hprev (conv_epilogue.h formulas, same order of operations)
            f4_t rr[4];                          // residual, staged by the conv1 lane of the same index
            const int xsl = ((hprev % F_RRING) + F_RRING) % F_RRING;
#pragma unroll
            for (int c = 0; c < 4; ++c) rr[c] = *(const f4_t*)(rring + xsl * F_RSLOT + (wave & 3) * 4096 + c * 1024 + lane * 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                h4_t uu;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc2[c][r] + bias_v[r];
                    v = lin_act(v, 1.f);
                    v += rr[c][r];
                    ov[c][r] = v;
                    const float a2 = v * s2_v[r] + t2_v[r];
                    uu[r] = ep_h(lin_act(a2, p.sl1));
                }
                ou[c] = __builtin_bit_cast(ep_u2_t, uu);
            }
            hrow = hprev;
        };
        auto store_k = [&](int k) {              // store k (0 .. 7) of row hrow
            asm volatile("" ::: "memory");
            const int c = k >> 1;
            if ((k & 1) == 0) *(float4*)(p.out0 + (nb_o0 + (unsigned)(hrow * p.o0_sH + c * p.o0_sW) + lane_el)) = make_float4(ov[c][0], ov[c][1], ov[c][2], ov[c][3]);
            else *(ep_u2_t*)(p.out1 + (nb_o1 + (unsigned)(hrow * p.o1_sH + c * p.o1_sW) + lane_el)) = ou[c];
            asm volatile("" ::: "memory");
        };
        auto dma_k = [&](int tt, int k) {        // conv1: DMA instruction k (0 .. 7) of step tt: a row tt + 3 into the slot row tt - 2 left, residual row tt - 1
            asm volatile("" ::: "memory");
            int sk = sa + F_KA + 2; sk -= sk >= F_ARING ? F_ARING : 0;
            if (k < 4) stage_a(tt + 3, sk, k);
            else stage_x(tt - 1, (((tt - 1) % F_RRING) + F_RRING) % F_RRING, k - 4);
            asm volatile("" ::: "memory");
        };
        // step t: conv1 -> h row t (rows h0 - 1 .. h1), conv2 -> output row t - 2 (rows h0 .. h1 - 1)
        for (int tt = h0 - 1; tt <= h1 + 1; ++tt) {
            const bool c1_on = tt <= h1, c2_on = tt >= h0 + 2;
            FTL(8);
            if (!role) {
                // ---------------- conv1: h row tt for h-slab columns 5 chf .. + 4 from a rows tt - 1 .. tt + 1
                if (c1_on) {
                    int rb[3];
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) { int sl = sa + kh; sl -= sl >= F_ARING ? F_ARING : 0; rb[kh] = lanebase + sl * F_ARS; }
                    f4_t acc[5];
#pragma unroll
                    for (int c = 0; c < 5; ++c) acc[c] = (f4_t){0.f, 0.f, 0.f, 0.f};
                    h8_t fa[7], fb[7];
#pragma unroll
                    for (int i = 0; i < 7; ++i) fa[i] = *(const h8_t*)(smem + rb[0] + i * F_CS);
#pragma unroll
                    for (int g = 0; g < 9; ++g) {
                        h8_t (&cur)[7] = (g & 1) ? fb : fa;
                        h8_t (&nxt)[7] = (g & 1) ? fa : fb;
                        if (g + 1 < 9) {
                            const int kd = (g + 1) / 3, kh = (g + 1) % 3;
#pragma unroll
                            for (int i = 0; i < 7; ++i) nxt[i] = *(const h8_t*)(smem + rb[kh] + i * F_CS + kd * 16);
                        }
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                            for (int c = 0; c < 5; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[g * 3 + kw], cur[kw + c], acc[c], 0, 0, 0);
                        if (g + 1 < 9) {     // the 7 LDS reads of the next group interleaved with this group's MFMAs, two MFMAs per read
#pragma unroll
                            for (int i = 0; i < 7; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#ifdef F_SPREAD
                        if (g < 8) { dma_k(tt, g); __builtin_amdgcn_sched_barrier(0); }
#endif
                    }
                    FTL(2);
                    // h = relu(conv1 + b1) as fp16 (what the two-launch path stores); outside the volume h is conv2's zero padding
                    const bool row_in = (unsigned)tt < (unsigned)p.H;
                    unsigned char* hrow = himg + hw_base + (tt & 3) * F_HRS;
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        const bool in = row_in && (unsigned)(w0 - 1 + 5 * chf + c) < (unsigned)p.W;
                        h4_t hv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) hv[r] = in ? ep_h(lin_act(acc[c][r] + bias_v[r], 0.f)) : (half_t)0.f;
                        *(h4_t*)(hrow + c * F_CS) = hv;
                    }
                    FTL(4);
                }
#ifdef F_SPREAD
                if (!c1_on) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) dma_k(tt, k);
                }
                // the 8 DMA instructions of the LAST step must have landed (rows / residuals the next step reads); this step's may stay in flight
                f_wait_vm<8>();
                FTL(6);
#else
                // the DMA instructions issued at the end of the last step are a whole step old: everything this wave has in flight may drain
                f_wait_vm<0>();
                FTL(6);
#pragma unroll
                for (int k = 0; k < 8; ++k) dma_k(tt, k);
                FTL(3);
#endif
            } else {
                // ---------------- conv2: finish output row hprev, then accumulate output row tt - 2 from h rows tt - 3 .. tt - 1
                const bool st_on = pend;
                if (pend) { finish_math(); pend = false; }
#ifndef F_SPREAD
                if (st_on) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) store_k(k);
                }
#endif
                FTL(3);
                if (c2_on) {
                    const int ro = tt - 2;
                    int rb[3];
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) rb[kh] = lanebase + ((ro - 1 + kh) & 3) * F_HRS;
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc2[c] = (f4_t){0.f, 0.f, 0.f, 0.f};
                    h8_t fa[6], fb[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) fa[i] = *(const h8_t*)(himg + rb[0] + i * F_CS);
#pragma unroll
                    for (int g = 0; g < 9; ++g) {
                        h8_t (&cur)[6] = (g & 1) ? fb : fa;
                        h8_t (&nxt)[6] = (g & 1) ? fa : fb;
                        if (g + 1 < 9) {
                            const int kd = (g + 1) / 3, kh = (g + 1) % 3;
#pragma unroll
                            for (int i = 0; i < 6; ++i) nxt[i] = *(const h8_t*)(himg + rb[kh] + i * F_CS + kd * 16);
                        }
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc2[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[g * 3 + kw], cur[kw + c], acc2[c], 0, 0, 0);
                        if (g + 1 < 9) {     // the 6 LDS reads of the next group interleaved with this group's MFMAs, two MFMAs per read
#pragma unroll
                            for (int i = 0; i < 6; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#ifdef F_SPREAD
                        if (g < 8 && st_on) { store_k(g); __builtin_amdgcn_sched_barrier(0); }
#endif
                    }
                    pend = true; hprev = ro;
                    FTL(2);
                }
#ifdef F_SPREAD
                else if (st_on) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) store_k(k);
                }
#endif
            }
            sa = sa + 1 == F_ARING ? 0 : sa + 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS traffic (the h row) is done
            FTL(5);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            FTL(7);
#ifdef V32_TL
            tl_steps += 1;
#endif
        }
        if (role && pend) {
            finish_math();
#pragma unroll
            for (int k = 0; k < 8; ++k) store_k(k);
        }
        __syncthreads();                         // the next item's prologue overwrites the rings
        FTL(9);
    }
#ifdef V32_TL
    if (p.tl && lane == 0) {
        const long wi = (long)blockIdx.x * 8 + wave;
        if (wi < p.tl_cap) {
            unsigned long long* o = p.tl + wi * 12;
#pragma unroll
            for (int i = 0; i < 10; ++i) o[i] = tl_acc[i];
            o[10] = tl_steps; o[11] = role;
        }
    }
#endif
}

int g_ncu_f = 0;
#ifdef V32_TL
unsigned long long* g_v32f_tl = nullptr;
long g_v32f_cap = 0;
#endif

}  // namespace

#ifdef V32_TL
extern "C" void cs_debug_set_vol32f_tl(void* buf, long cap) { g_v32f_tl = (unsigned long long*)buf; g_v32f_cap = cap; }
#endif

bool vol32_fused_supported(const ResBlock3dCall& c)
{
    return c.N >= 1 && c.H >= 1 && c.W >= F_TW && (c.W % F_TW) == 0 && c.a && c.x && c.out0 && c.out1 && c.w1 && c.w2 && c.act1 <= ACT_LRELU &&
           c.a != c.out1 && c.x != c.out0;
}

int launch_vol32_fused(const ResBlock3dCall& c, hipStream_t st)
{
    if (!vol32_fused_supported(c)) { cs_set_error("vol32_fused: unsupported ResBlock3d call (W %% 8, distinct in / out buffers, act1 <= LeakyReLU)"); return -1; }
    const long span = (long)(c.N - 1) * c.sN + (long)(c.H - 1) * c.sH + (long)(c.W - 1) * c.sW + 512;
    if (span >= (1L << 31)) { cs_set_error("vol32_fused: a volume spans 2^31 elements or more"); return -1; }
    if (!g_ncu_f) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        g_ncu_f = n;
    }
    FusedParams k;
    k.a = c.a; k.zero = cs_zero_page(); k.a_sN = (int)c.sN; k.a_sH = (int)c.sH; k.a_sW = (int)c.sW;
    k.x = c.x; k.x_sN = (int)c.sN; k.x_sH = (int)c.sH; k.x_sW = (int)c.sW;
    k.out0 = c.out0; k.o0_sN = (int)c.sN; k.o0_sH = (int)c.sH; k.o0_sW = (int)c.sW;
    k.out1 = c.out1; k.o1_sN = (int)c.sN; k.o1_sH = (int)c.sH; k.o1_sW = (int)c.sW;
    k.w1 = c.w1; k.w2 = c.w2; k.b1 = c.b1; k.b2 = c.b2; k.s2 = c.s2; k.t2 = c.t2;
    k.sl1 = c.act1 == ACT_NONE ? 1.f : (c.act1 == ACT_LRELU ? c.slope1 : 0.f);
    k.N = c.N; k.H = c.H; k.W = c.W;
#ifdef V32_TL
    k.tl = g_v32f_tl; k.tl_cap = g_v32f_cap;
#endif
    k.nstrips = c.W / F_TW;
    int nseg = 1;
    while ((long)c.N * k.nstrips * nseg < g_ncu_f && (c.H % (nseg * 2 * 8)) == 0) nseg *= 2;      // a segment re-stages 4 a rows and recomputes 2 h rows
    // one or two frames: down to 2-row segments while every item still gets a CU of its own (64 x 64 volume, one frame: 256 items of 2 rows
    // instead of 64 of 8 - 6 row steps per workgroup instead of 18).  Per output element nothing depends on the decomposition: same bits.
    while ((long)c.N * k.nstrips * nseg * 2 <= g_ncu_f && (c.H % (nseg * 2 * 2)) == 0) nseg *= 2;
    k.seg_rows = (c.H + nseg - 1) / nseg;
    k.nseg = (c.H + k.seg_rows - 1) / k.seg_rows;
    k.items = c.N * k.nstrips * k.nseg;
    int grid = k.items < g_ncu_f ? k.items : g_ncu_f;
    if (grid >= 8) grid &= ~7;
    hipError_t e = hipFuncSetAttribute((const void*)vol32_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
    if (e != hipSuccess) { cs_set_error("vol32_fused: opting into %d bytes of LDS failed: %s", F_LDS, hipGetErrorString(e)); return -1; }
    hipLaunchKernelGGL(vol32_fused_kernel, dim3((unsigned)grid), dim3(512), F_LDS, st, k);
    e = hipGetLastError();
    if (e != hipSuccess) { cs_set_error("vol32_fused launch: %s", hipGetErrorString(e)); return -1; }
    return 0;
}
