// Micro-benchmark (GPU box), companion of mfma_power.hip: the MFMA stream of a 128 x 256 conv tile's wave (8 x 4 fragments of
// v_mfma_f32_16x16x32_f16 per 32-channel step, one wave per SIMD) with and without the operand traffic of the real kernel:
//   +lds : 8 ds_read_b128 per step (the activation fragments; conflict-free 16-byte slots)
//   +l2  : 4 global_load_dwordx4 per step from a 4 MB buffer every workgroup walks (the weight stream: L2 hits)
// Whole chip, about a second per case (DVFS settled), post-ReLU-like data (A side half zeros).  Tells how much of the chip's power
// budget the operand delivery of such a tile costs, i.e. what conv_halo's T kernel (0.578 of 2.5 PFLOP/s) could reach at best.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power2.hip -o /tmp/mfma_power2 && /tmp/mfma_power2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <int LDS, int L2, int NB = 4>      // fragments per step that are really fetched (the others stay in registers); NB: n fragments of the wave (8 x NB accumulators)
__global__ void __launch_bounds__(256) tile(const h8_t* __restrict__ adata, const h8_t* __restrict__ wdata, int wfrags, float* out, int iters)
{
    __shared__ h8_t img[4096];                     // 64 KB of activations
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) img[i] = adata[(blockIdx.x & 63) * 4096 + i];
    __syncthreads();
    f4_t acc[8][NB];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = (f4_t){0.f, 0.f, 0.f, 0.f};
    h8_t a[2][8], b[2][NB];
    auto fetch = [&](int s, h8_t (&av)[8], h8_t (&bv)[NB]) {
#pragma unroll
        for (int m = 0; m < 8; ++m) if (m < LDS) av[m] = img[((s * 8 + m) * 64 + lane + wave * 17) & 4095]; else if (s < 2) av[m] = adata[(tid * 8 + m) & 4095];
#pragma unroll
        for (int n = 0; n < NB; ++n) if (n < L2) bv[n] = wdata[(unsigned)((s * NB + n) * 256 + wave * 64 + lane) & (unsigned)(wfrags - 1)]; else if (s < 2) bv[n] = wdata[tid * NB + n];
    };
    fetch(0, a[0], b[0]);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (LDS || L2 || it == 0) fetch(it + h + 1, a[h ^ 1], b[h ^ 1]);
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) MFMA(acc[m][n], a[h][m], b[h][n]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = s;
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    const int nb = 256, nt = nb * 256;
    const int wfrags = 1 << 18;                    // 4 MB of weights (a power of two: the index is a mask), about one T layer's rows of a 256-channel block
    std::vector<_Float16> ha((size_t)64 * 4096 * 8), hw((size_t)wfrags * 8);
    srand(1);
    auto rnd = [] { const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
                    return 0.05 * sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2); };
    for (auto& v : ha) { const double x = rnd() * 20; v = (_Float16)(x < 0 ? 0 : x); }
    for (auto& v : hw) v = (_Float16)rnd();
    h8_t *da, *dw; float* out;
    hipMalloc(&da, ha.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&out, nt * 4);
    hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int zeros = argc > 2 ? atoi(argv[2]) : 0;
    if (zeros) { hipMemset(da, 0, ha.size() * 2); hipMemset(dw, 0, hw.size() * 2); }
    const char* names[9] = {"registers only", "+lds 4 ds_read_b128 / step", "+lds 8", "+l2 2 global_load_dwordx4 / step", "+l2 4", "+lds 8 +l2 4 (the 128 x 256 tile)", "+lds 4 +l2 2",
                             "8 x 8 fragments, registers only", "8 x 8 fragments +lds 8 +l2 8 (256 accumulator registers)"};
    for (int v = 0; v < 9; ++v) {
        const double flop_per_it = (v >= 7 ? 64.0 : 32.0) * 2 * 16 * 16 * 32 * nb * 4;
        int iters = 2000; float ms = 0;
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
#define GO(A, B) hipLaunchKernelGGL((tile<A, B>), dim3(nb), dim3(256), 0, 0, da, dw, wfrags, out, iters)
            if (v == 0) GO(0, 0); if (v == 1) GO(4, 0); if (v == 2) GO(8, 0); if (v == 3) GO(0, 2); if (v == 4) GO(0, 4); if (v == 5) GO(8, 4); if (v == 6) GO(4, 2);
            if (v == 7) hipLaunchKernelGGL((tile<0, 0, 8>), dim3(nb), dim3(256), 0, 0, da, dw, wfrags, out, iters);
            if (v == 8) hipLaunchKernelGGL((tile<8, 8, 8>), dim3(nb), dim3(256), 0, 0, da, dw, wfrags, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (pass == 0) iters = (int)(iters * (secs * 1e3 / ms)) & ~1;
        }
        printf("%s%-36s %8.1f TFLOP/s sustained over %.2f s (%.3f of 2500)\n", zeros ? "[all-zero operands] " : "", names[v], flop_per_it * iters / (ms * 1e-3) / 1e12, ms * 1e-3,
               flop_per_it * iters / (ms * 1e-3) / 2.5e15);
    }
    return 0;
}
