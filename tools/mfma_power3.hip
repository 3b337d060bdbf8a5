// Micro-benchmark (GPU box), round 5 pricing: sustained MFMA rate of the whole chip, one wave per SIMD,
// about a second per case (DVFS settled), for the operand mixes of
//   * the direct 3x3 conv as conv_wide runs it (8 x 8 fragments per wave, 8 ds_read_b128 + 8 global_load_dwordx4 per 64 MFMAs, weight
//     fragments shared by wave pairs, 4.7 MB weight set),
//   * Winograd F(2,3) along W (4 xi, one xi per wave, 8 x 8 fragments: the same instruction mix but no weight sharing between waves, a
//     6.3 MB weight set, and the input transform V = d[j1] +- d[j2] done in registers: 16 ds_read_b128 + 32 v_pk_add_f16 per 64 MFMAs),
//   * Winograd F(2x2,3x3) (16 xi, four per wave, 4 x 4 fragments per xi: 4 + 4 loads per 16 MFMAs, 8.4 MB weight set; with the transform
//     in registers: 8 ds_read_b128 and 8 x 4 v_pk_add_f16 per xi step and 4 tile fragments),
//   * the block-scaled narrow MFMA v_mfma_scale_f32_16x16x128_f8f6f4 on fp8 / fp6 / fp4 operands from registers, and the split-precision mix
//     (one fp16 pass + two narrow correction passes).
// "equiv" = sustained rate x the algorithmic FLOPs one executed FLOP stands for (1.5 for F(2,3), 2.25 for F(2x2,3x3)).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power3.hip -o /tmp/mfma_power3 && /tmp/mfma_power3 [seconds per case]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef int i8_t __attribute__((ext_vector_type(8)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

// NXI xi groups per wave, MF x NF fragments each.  Per step and xi: MF A fragments (LDS if LDS, and a second read + packed add each if XF),
// NF B fragments (global if L2; SHARE: wave pairs read the same fragments).
template <int NXI, int MF, int NF, bool LDS, bool L2, bool SHARE, bool XF>
__global__ void __launch_bounds__(256) mix(const h8_t* __restrict__ adata, const h8_t* __restrict__ wdata, unsigned wmask, float* out, int iters)
{
    __shared__ h8_t img[4096];                     // 64 KB of activations
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) img[i] = adata[(blockIdx.x & 63) * 4096 + i];
    __syncthreads();
    f4_t acc[NXI][MF][NF];
#pragma unroll
    for (int x = 0; x < NXI; ++x)
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n) acc[x][m][n] = (f4_t){0.f, 0.f, 0.f, 0.f};
    h8_t a[2][MF], b[2][NF];
    const int wsel = SHARE ? (wave >> 1) : wave;
    auto fetch = [&](int s, h8_t (&av)[MF], h8_t (&bv)[NF]) {
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            if (LDS) {
                av[m] = img[((s * MF + m) * 64 + lane + wave * 17) & 4095];
                if (XF) {
                    const h8_t o = img[((s * MF + m) * 64 + lane + wave * 17 + 2048) & 4095];
                    av[m] = av[m] + o;             // 4 v_pk_add_f16
                }
            } else if (s < 2) av[m] = adata[(tid * 8 + m) & 4095];
        }
#pragma unroll
        for (int n = 0; n < NF; ++n)
            if (L2) bv[n] = wdata[(unsigned)(((s * NF + n) * 4 + wsel) * 64 + lane) & wmask];
            else if (s < 2) bv[n] = wdata[tid * NF + n];
    };
    fetch(0, a[0], b[0]);
    int s = 0;
    for (int it = 0; it < iters; it += 2) {        // two steps per trip: the operand buffers alternate with compile-time indices
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int x = 0; x < NXI; ++x) {
                const int h = NXI == 1 ? u : (x & 1);
                if (LDS || L2 || (it == 0 && u == 0 && x == 0)) fetch(++s, a[h ^ 1], b[h ^ 1]);
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int n = 0; n < NF; ++n) MFMA(acc[x][m][n], a[h][m], b[h][n]);
            }
    }
    float r = 0.f;
#pragma unroll
    for (int x = 0; x < NXI; ++x)
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n) r += acc[x][m][n][0] + acc[x][m][n][1] + acc[x][m][n][2] + acc[x][m][n][3];
    out[blockIdx.x * 256 + tid] = r;
}

// the block-scaled instruction from registers: FA / FB = format codes (0 fp8 e4m3, 2 fp6 e2m3, 4 fp4 e2m1); F16 = fp16 MFMAs issued with it
// (F16 = 4, NARROW = 2: one K = 128 fp16 pass (4 instructions) + two narrow correction instructions, R's split-precision mix)
template <int FA, int FB, int NARROW, int F16>
__global__ void __launch_bounds__(256) narrow(const i8_t* __restrict__ d8, const h8_t* __restrict__ d16, float* out, int iters)
{
    const int tid = threadIdx.x;
    f4_t acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f4_t){0.f, 0.f, 0.f, 0.f};
    i8_t a8[8], b8[4]; h8_t a16[8], b16[4];
#pragma unroll
    for (int m = 0; m < 8; ++m) { a8[m] = d8[(tid * 8 + m) & 4095]; a16[m] = d16[(tid * 8 + m) & 4095]; }
#pragma unroll
    for (int n = 0; n < 4; ++n) { b8[n] = d8[(tid * 4 + n + 1234) & 4095]; b16[n] = d16[(tid * 4 + n + 1234) & 4095]; }
    const int sc = 0x70707070;                     // E8M0 2^-15 for every block
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
#pragma unroll
                for (int r = 0; r < F16; ++r) MFMA(acc[m][n], a16[m], b16[n]);
#pragma unroll
                for (int r = 0; r < NARROW; ++r)
                    acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[m], b8[n], acc[m][n], FA, FB, 0, sc, 0, sc);
            }
        asm volatile("" ::: "memory");
    }
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) r += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = r;
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    const int nb = 256, nt = nb * 256;
    const size_t wfrags = 1 << 20;                 // 16 MB of weights; the mask picks the working set
    std::vector<_Float16> ha((size_t)64 * 4096 * 8), hw(wfrags * 8);
    std::vector<unsigned char> h8b((size_t)4096 * 32);
    srand(1);
    auto rnd = [] { const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
                    return 0.05 * sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2); };
    for (auto& v : ha) v = (_Float16)(rnd() * 20);                 // dense (G's inputs pass a leaky ReLU; Winograd-transformed data are dense)
    for (auto& v : hw) v = (_Float16)rnd();
    for (auto& v : h8b) { v = (unsigned char)(rand() & 0xFF); if ((v & 0x78) == 0x78) v ^= 0x40; }     // random narrow payloads, no NaN / large exponents
    h8_t *da, *dw; i8_t* d8; float* out;
    hipMalloc(&da, ha.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&d8, h8b.size()); hipMalloc(&out, nt * 4);
    hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d8, h8b.data(), h8b.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char* name; double flop_per_it; double equiv; int id; };
    const double F = 2.0 * 16 * 16 * 32 * nb * 4;                  // one fp16 MFMA per wave, whole chip
    const Case cases[] = {
        {"fp16 8x8 fragments, registers only (dense data)", 64 * F, 1, 0},
        {"direct conv_wide mix: 8x8, 8 lds + 8 l2 (pairs share), 4 MB set", 64 * F, 1, 1},
        {"direct mix, weights NOT shared by wave pairs, 4 MB", 64 * F, 1, 2},
        {"F(2,3) mix: 8x8 per xi-wave, 8 lds + 8 l2 unshared, 8 MB set", 64 * F, 1.5, 3},
        {"F(2,3) mix + transform in registers (16 lds, 32 pk_add)", 64 * F, 1.5, 4},
        {"F(2x2,3x3) mix: 4 xi x 4x4 per wave, 4 lds + 4 l2 per xi, 8 MB", 64 * F, 2.25, 5},
        {"F(2x2,3x3) mix + transform in registers (8 lds, 16 pk_add per xi)", 64 * F, 2.25, 6},
        {"F(2x2,3x3) mix, 16 MB set", 64 * F, 2.25, 7},
        {"F(2,3) all 4 xi per wave: 4x4 per xi, 16 lds + 16 l2 (pairs share), 8 MB", 64 * F, 1.5, 8},
        {"F(2,3) all 4 xi per wave + transform (32 lds, 64 pk_add)", 64 * F, 1.5, 9},
        {"scaled 16x16x128 fp8 x fp8, registers only", 32 * 4 * F, 1, 10},
        {"scaled 16x16x128 fp6 x fp6, registers only", 32 * 4 * F, 1, 11},
        {"scaled 16x16x128 fp4 x fp4, registers only", 32 * 4 * F, 1, 12},
        {"scaled 16x16x128 fp8 x fp6", 32 * 4 * F, 1, 13},
        {"split mix: 4 fp16 (K=128) + 2 fp8 correction instr; fp16-pass FLOPs", 32 * 4 * F, 1, 14},
        {"split mix: 4 fp16 (K=128) + 2 fp6 correction instr; fp16-pass FLOPs", 32 * 4 * F, 1, 15},
        {"three fp16 passes (R today); one pass's FLOPs", 32 * 4 * F, 1, 16},
        {"one fp16 pass (32 fragments)", 32 * 4 * F, 1, 17},
    };
    for (const Case& c : cases) {
        int iters = 400; float ms = 0;
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
#define MIX(...) hipLaunchKernelGGL((mix<__VA_ARGS__>), dim3(nb), dim3(256), 0, 0, da, dw, wm, out, iters)
#define NAR(...) hipLaunchKernelGGL((narrow<__VA_ARGS__>), dim3(nb), dim3(256), 0, 0, d8, da, out, iters)
            unsigned wm = (1u << 18) - 1;          // 4 MB
            switch (c.id) {
            case 0: MIX(1, 8, 8, false, false, false, false); break;
            case 1: MIX(1, 8, 8, true, true, true, false); break;
            case 2: MIX(1, 8, 8, true, true, false, false); break;
            case 3: wm = (1u << 19) - 1; MIX(1, 8, 8, true, true, false, false); break;
            case 4: wm = (1u << 19) - 1; MIX(1, 8, 8, true, true, false, true); break;
            case 5: wm = (1u << 19) - 1; MIX(4, 4, 4, true, true, false, false); break;
            case 6: wm = (1u << 19) - 1; MIX(4, 4, 4, true, true, false, true); break;
            case 7: wm = (1u << 20) - 1; MIX(4, 4, 4, true, true, false, false); break;
            case 8: wm = (1u << 19) - 1; MIX(4, 4, 4, true, true, true, false); break;
            case 9: wm = (1u << 19) - 1; MIX(4, 4, 4, true, true, true, true); break;
            case 10: NAR(0, 0, 1, 0); break;
            case 11: NAR(2, 2, 1, 0); break;
            case 12: NAR(4, 4, 1, 0); break;
            case 13: NAR(0, 2, 1, 0); break;
            case 14: NAR(0, 0, 2, 4); break;
            case 15: NAR(2, 2, 2, 4); break;
            case 16: NAR(0, 0, 0, 12); break;
            case 17: NAR(0, 0, 0, 4); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", c.name); break; }
            if (pass == 0) iters = (int)(iters * (secs * 1e3 / ms)) & ~1;
        }
        const double r = c.flop_per_it * iters / (ms * 1e-3) / 1e12;
        printf("%-72s %8.1f TFLOP/s over %.2f s (%.3f of 2500; x%.2f algorithmic = %.3f)\n", c.name, r, ms * 1e-3, r / 2500, c.equiv, r / 2500 * c.equiv);
    }
    return 0;
}
