// Micro-benchmark (GPU box): issue rate of v_mfma_f32_16x16x32_f16 vs the carried-forward v_mfma_f32_16x16x16_f16 on gfx950,
// one wave per SIMD, 8 independent accumulators back to back.   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <int K32>
__global__ void __launch_bounds__(256) rate(float* out, unsigned long long* cyc, int iters)
{
    f4_t acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (f4_t){0.f, 0.f, 0.f, 0.f};
    h8_t a8, b8; h4_t a4, b4;
    for (int k = 0; k < 8; ++k) { a8[k] = (_Float16)(threadIdx.x * 0.001f + k); b8[k] = (_Float16)(k * 0.01f + threadIdx.x * 0.002f); }
    for (int k = 0; k < 4; ++k) { a4[k] = a8[k]; b4[k] = b8[k]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // inline asm: the builtin form made hipcc rotate the accumulators through overlapping AGPR ranges (a dependent chain)
            if (K32) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a8), "v"(b8));
            else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a4), "v"(b4));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main()
{
    const int nb = 8, iters = 2000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nb * 256 * sizeof(float)); hipMalloc(&cyc, nb * 4 * sizeof(unsigned long long));
    unsigned long long h[nb * 4];
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode) hipLaunchKernelGGL(rate<1>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(rate<0>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < nb * 4; ++i) s += (double)h[i];
        printf("%s: %.2f cycles per MFMA (one wave per SIMD, 8 independent accumulators)\n", mode ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_16x16x16_f16",
               s / (nb * 4) / (iters * 8.0));
    }
    return 0;
}
