// Micro-benchmark (GPU box), round 5: what the OPERAND MIX of each conv_halo tile family sustains on the whole chip under the power cap -
// nothing but the family's fragment loads (LDS reads of dense activations, weight fragments from an L2-resident set) and its MFMAs, at the family's
// occupancy, about a second per case (DVFS settled).  A family's measured executed fraction is to be read against its row, not against 2.5 PFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power4.hip -o /tmp/mfma_power4 && /tmp/mfma_power4 [seconds per case]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

// MF activation fragments (LDS) x NF weight fragments (global, L2-resident set; SHARE: wave pairs read the same fragments) per wave and K-step;
// IMG: KB of LDS per workgroup (sets the occupancy together with the grid)
template <int MF, int NF, bool LDS, bool L2, bool SHARE, int IMG, int OCC, bool BLDS = false>
__global__ void __launch_bounds__(256, OCC) mix(const h8_t* __restrict__ adata, const h8_t* __restrict__ wdata, unsigned wmask, float* out, int iters)
{
    constexpr int NI = IMG * 64;                   // h8_t elements
    __shared__ h8_t img[NI];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NI; i += 256) img[i] = adata[((blockIdx.x & 63) * 4096 + i) & 262143];
    __syncthreads();
    f4_t acc[MF][NF];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = (f4_t){0.f, 0.f, 0.f, 0.f};
    h8_t a[2][MF], b[2][NF];
    const int wsel = SHARE ? (wave >> 1) : wave;
    auto fetch = [&](int s, h8_t (&av)[MF], h8_t (&bv)[NF]) {
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            if (LDS) av[m] = img[((s * MF + m) * 64 + lane + wave * 17) & (NI - 1)];
            else if (s < 2) av[m] = adata[(tid * 8 + m) & 4095];
        }
#pragma unroll
        for (int n = 0; n < NF; ++n)
            if (BLDS) bv[n] = img[((s * NF + n + 777) * 64 + lane + wsel * 23) & (NI - 1)];      // weight fragments from LDS too (what a DMA-staged weight tile would cost to read)
            else if (L2) bv[n] = wdata[(unsigned)(((s * NF + n) * 4 + wsel) * 64 + lane) & wmask];
            else if (s < 2) bv[n] = wdata[tid * NF + n];
    };
    fetch(0, a[0], b[0]);
    int s = 0;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (LDS || L2 || BLDS || (it == 0 && u == 0)) fetch(++s, a[u ^ 1], b[u ^ 1]);
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < NF; ++n) MFMA(acc[m][n], a[u][m], b[u][n]);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) r += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = r;
}

// conv_wide with its weights THROUGH LDS: per K-step a workgroup's four waves DMA the step's 16 weight fragments (16 KB, 4 pieces per wave,
// global_load_lds from an L2-resident set) into a ring of three steps, wait for their own pieces of the step about to be read (counted vmcnt),
// meet at a barrier, and every wave reads 8 activation + 8 weight fragments from LDS for its 64 MFMAs.
__global__ void __launch_bounds__(256, 1) mixdma(const h8_t* __restrict__ adata, const h8_t* __restrict__ wdata, unsigned wmask, float* out, int iters)
{
    extern __shared__ h8_t dyn[];
    h8_t* img = dyn;                               // 64 KB of activations
    h8_t* ring = dyn + 4096;                       // 3 x 16 KB of weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) img[i] = adata[((blockIdx.x & 63) * 4096 + i) & 262143];
    __syncthreads();
    f4_t acc[8][8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[m][n] = (f4_t){0.f, 0.f, 0.f, 0.f};
    // (the DMA goes through inline asm: hipcc waits with vmcnt(0) before any LDS read behind a DMA it knows of - DESIGN 5.6 rules 1 and 7)
    typedef int i4_t __attribute__((ext_vector_type(4)));
    i4_t rsrc;
    {
        const unsigned long long base = (unsigned long long)wdata;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        rsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu));
        rsrc[2] = 0x7fffffff; rsrc[3] = 0x00020000;
    }
    const unsigned lds_ring = (unsigned)(unsigned long long)(__attribute__((address_space(3))) h8_t*)ring;
    auto dma = [&](int s) {                        // this wave's 4 pieces of step s -> ring slot s % 3
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_ring + (unsigned)(((s % 3) * 16 + wave * 4 + q) * 1024)));
            const unsigned voff = (((unsigned)((s * 16 + wave * 4 + q) * 64 + lane)) & wmask) * 16u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory");
        }
    };
    dma(0); dma(1);
    for (int s = 0; s < iters; ++s) {
        dma(s + 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // the pieces of step s have landed (those of s + 1, s + 2 may be in flight)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        h8_t a[8], b[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = img[((s * 8 + m) * 64 + lane + wave * 17) & 4095];
#pragma unroll
        for (int n = 0; n < 8; ++n) b[n] = ring[((s % 3) * 16 + (wave >> 1) * 8 + n) * 64 + lane];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n) MFMA(acc[m][n], a[m], b[n]);
    }
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) r += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = r;
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    const size_t wfrags = 1 << 18;                 // 4 MB of weights; the mask picks the working set
    std::vector<_Float16> ha((size_t)64 * 4096 * 8), hw(wfrags * 8);
    srand(1);
    auto rnd = [] { const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
                    return 0.05 * sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2); };
    for (auto& v : ha) { const double x = rnd() * 20; v = (_Float16)(x > 0 ? x : 0); }       // post-ReLU activations: half of them zero
    for (auto& v : hw) v = (_Float16)rnd();
    h8_t *da, *dw; float* out;
    hipMalloc(&da, ha.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&out, 768 * 256 * 4);
    hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char* name; int mf, nf, nb, id; };
    const Case cases[] = {
        {"8x5 fragments, registers only, 1 workgroup per CU", 8, 5, 256, 0},
        {"256x160 tile (hourglass tail, mask conv): 8x5 per wave, 8 lds + 5 l2 (pairs share), 1 per CU, 1 MB set", 8, 5, 256, 1},
        {"256x64 tile (first encoder block): 8x2 per wave, 8 lds + 2 l2 (pairs share), 2 per CU, 0.5 MB set", 8, 2, 512, 2},
        {"128x256 tile (SPADE gamma / beta): 8x4 per wave, 8 lds + 4 l2 unshared, 2 per CU, 2 MB set", 8, 4, 512, 3},
        {"128x128 tile (hourglass enc / dec, 64-channel SPADE): 8x2 per wave, 8 lds + 2 l2 unshared, 3 per CU, 4 MB set", 8, 2, 768, 4},
        {"8x8 fragments (conv_wide): 8 lds + 8 l2 (pairs share), 1 per CU, 4 MB set", 8, 8, 256, 5},
        {"8x8 fragments, BOTH operands from LDS (16 ds_read_b128 per 64 MFMAs), 1 per CU", 8, 8, 256, 6},
        {"8x8 fragments, registers only", 8, 8, 256, 7},
        {"8x8 fragments, weights DMA-staged into an LDS ring (16 KB per step and workgroup, barrier per step), 16 lds reads", 8, 8, 256, 8},
    };
    for (const Case& c : cases) {
        int iters = 400; float ms = 0;
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
#define MIX(...) hipLaunchKernelGGL((mix<__VA_ARGS__>), dim3(c.nb), dim3(256), 0, 0, da, dw, wm, out, iters)
            unsigned wm = (1u << 18) - 1;
            switch (c.id) {
            case 0: MIX(8, 5, false, false, false, 1, 1); break;
            case 1: wm = (1u << 16) - 1; MIX(8, 5, true, true, true, 64, 1); break;
            case 2: wm = (1u << 15) - 1; MIX(8, 2, true, true, true, 64, 2); break;
            case 3: wm = (1u << 17) - 1; MIX(8, 4, true, true, false, 64, 2); break;
            case 4: MIX(8, 2, true, true, false, 32, 3); break;
            case 5: MIX(8, 8, true, true, true, 64, 1); break;
            case 6: MIX(8, 8, true, false, true, 64, 1, true); break;
            case 7: MIX(8, 8, false, false, false, 1, 1); break;
            case 8: {
                static bool once = false;
                if (!once) { hipFuncSetAttribute((const void*)mixdma, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 49152); once = true; }
                hipLaunchKernelGGL(mixdma, dim3(c.nb), dim3(256), 65536 + 49152, 0, da, dw, wm, out, iters);
                break;
            }
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", c.name); break; }
            if (pass == 0) iters = (int)(iters * (secs * 1e3 / ms)) & ~1;
        }
        const double F = 2.0 * 16 * 16 * 32 * c.nb * 4;
        const double r = F * c.mf * c.nf * iters / (ms * 1e-3) / 1e12;
        printf("%-112s %8.1f TFLOP/s over %.2f s (%.3f of 2500)\n", c.name, r, ms * 1e-3, r / 2500);
    }
    return 0;
}
