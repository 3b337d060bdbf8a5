// Micro-benchmark (GPU box): what a pure MFMA stream sustains on the WHOLE chip for about a second (DVFS settled), by operand data and by
// MFMA shape.  No LDS, no memory in the loop: the ceiling any convolution kernel of this engine can approach on the same data.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
// Operands: 8 A and 8 B fragments per lane from a buffer (random normal / post-ReLU (half zeros) / all zeros), every MFMA takes another pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));

template <int SHAPE>      // 0: 16x16x32 (8 accumulators), 1: 32x32x16 (4 accumulators)
__global__ void __launch_bounds__(256) burn(const h8_t* __restrict__ data, float* out, int iters)
{
    h8_t a[8], b[8];
    const int t = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = data[(size_t)t * 16 + k]; b[k] = data[(size_t)t * 16 + 8 + k]; }
    float s = 0.f;
    if (SHAPE == 0) {
        f4_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (f4_t){0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a[j]), "v"(b[(j + r) & 7]));      // asm: the builtin form makes hipcc rotate the accumulators through overlapping AGPR ranges
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        f16_t acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a[(j + 4 * (r & 1)) & 7]), "v"(b[(j + r) & 7]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) s += acc[j][q];
    }
    out[t] = s;
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    const int wpc = argc > 2 ? atoi(argv[2]) : 1;             // workgroups (4 waves) per CU: 1 = one wave per SIMD
    const int nb = 256 * wpc, nt = nb * 256;
    std::vector<_Float16> h((size_t)nt * 128);
    h8_t* d; float* out;
    hipMalloc(&d, h.size() * 2); hipMalloc(&out, nt * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"random normal x 0.05", "post-ReLU (half zeros)", "all zeros"};
    for (int kind = 0; kind < 3; ++kind) {
        srand(1);
        for (size_t i = 0; i < h.size(); ++i) {
            const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
            double v = 0.05 * sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2);
            // fragments 0..7 of a lane's 16 are the A side: post-ReLU zeroes the A side's negatives
            if (kind == 1 && ((i / 8) % 16) < 8 && v < 0) v = 0;
            if (kind == 2) v = 0;
            h[i] = (_Float16)v;
        }
        hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int shape = 0; shape < 2; ++shape) {
            const double flop_per_it = (shape ? 32.0 * 2 * 32 * 32 * 16 : 64.0 * 2 * 16 * 16 * 32) * nb * 4;   // MFMAs per iteration per wave x flops
            // calibrate: a short launch, then one sized for `secs`
            int iters = 2000; float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {
                hipEventRecord(e0);
                if (shape) hipLaunchKernelGGL(burn<1>, dim3(nb), dim3(256), 0, 0, d, out, iters);
                else hipLaunchKernelGGL(burn<0>, dim3(nb), dim3(256), 0, 0, d, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                if (pass == 0) iters = (int)(iters * (secs * 1e3 / ms));
            }
            printf("%-24s %s  %d wave(s)/SIMD: %8.1f TFLOP/s sustained over %.2f s (%.3f of 2500)\n", names[kind],
                   shape ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", wpc, flop_per_it * iters / (ms * 1e-3) / 1e12, ms * 1e-3,
                   flop_per_it * iters / (ms * 1e-3) / 2.5e15);
        }
    }
    return 0;
}
