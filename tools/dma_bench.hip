// Micro-benchmark (GPU box): how fast does a workgroup get a conv halo into LDS?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_bench tools/dma_bench.hip && /tmp/dma_bench
// Each workgroup (256 threads) repeatedly fetches one "halo" of V voxels x 64 bytes from a 64 MB fp16 tensor
// [H=64][W=64][D=16][C=32] (the feature-volume layout) and waits for it; no compute.  Variants:
//   path  : LDS-DMA (global_load_lds_dwordx4) or global_load_dwordx4 + ds_write_b128
//   order : halo voxels enumerated w-fastest (64-byte pieces 1 KiB apart) or d-fastest (1 KiB contiguous runs)
//   wgs   : resident workgroups per CU (set through the dynamic LDS size)
// Output: GB/s over the whole chip and bytes per ns per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

constexpr int HW = 6, HH = 6, HD = 18, V = HW * HH * HD;     // 4x4x16 tile + halo
constexpr int PIECES = V * 4;                                 // 16-byte pieces

template <bool DMA, bool DFAST>
__global__ void __launch_bounds__(256) halo_fetch(const half_t* __restrict__ x, int iters, int ntiles, unsigned* sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const int t = (blockIdx.x + it * gridDim.x) % ntiles;
        const int tw = t % 16, th = (t / 16) % 16, n = t / 256;
        const half_t* base = x + (long)n * 64 * 64 * 16 * 32 + ((long)(th * 4) * 64 + tw * 4) * 16 * 32;
        for (int q0 = 0; q0 < PIECES; q0 += 256) {
            const int q = q0 + tid;
            if (q < PIECES) {
                const int slot = q & 3, hv = q >> 2;
                int hw, hh, hd;
                if (DFAST) { hd = hv % HD; hw = (hv / HD) % HW; hh = hv / (HD * HW); }
                else { hw = hv % HW; hh = (hv / HW) % HH; hd = hv / (HW * HH); }
                const int ih = th * 4 + hh - 1, iw = tw * 4 + hw - 1, id = hd - 1;
                const bool ok = (unsigned)ih < 64u && (unsigned)iw < 64u && (unsigned)id < 16u;
                const half_t* src = ok ? base + (((long)(hh - 1) * 64 + (hw - 1)) * 16 + (hd - 1)) * 32 + slot * 8 : x;
                if (DMA) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(smem + (size_t)(q0 + wave * 64) * 16), 16, 0, 0);
                } else {
                    const u4_t v = *(const u4_t*)src;
                    *(u4_t*)(smem + (size_t)q * 16) = v;
                }
            }
        }
        __syncthreads();
        acc += *(const unsigned*)(smem + (tid & 63) * 16);
        __syncthreads();
    }
    if (acc == 0x12345678u) *sink = acc;
}

// Store side: each wave writes its 64 positions x 32 fp16 channels of a 4x4x16 tile (what the c1 epilogue of the 32->32 convs
// does: 8 instructions of 8 bytes per lane, lanes l4 -> 4 x 8 bytes inside a 64-byte row, l15 -> 16 rows) or the same bytes as
// 2 contiguous 16-byte stores per lane; then waits for the acknowledgements (what the end of a workgroup does).
template <int PATTERN>
__global__ void __launch_bounds__(256) tile_store(half_t* __restrict__ y, int iters, int ntiles)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    for (int it = 0; it < iters; ++it) {
        const int t = (blockIdx.x + it * gridDim.x) % ntiles;
        const int tw = t % 16, th = (t / 16) % 16, n = t / 256;
        half_t* base = y + (long)n * 64 * 64 * 16 * 32 + ((long)(th * 4) * 64 + tw * 4) * 16 * 32;
        if (PATTERN == 0) {
            for (int pi = 0; pi < 4; ++pi) {
                const int m = wave * 64 + pi * 16 + l15;            // position inside the tile: w(2) h(2) d(4)
                const int wl = m & 3, hl = (m >> 2) & 3, dl = m >> 4;
                half_t* o = base + (((long)hl * 64 + wl) * 16 + dl) * 32;
                for (int ci = 0; ci < 2; ++ci) *(uint2*)(o + ci * 16 + l4 * 4) = make_uint2(it, tid);
            }
        } else {
            for (int k = 0; k < 2; ++k) {                           // 4 (w,h) columns per wave, 1 KiB contiguous each
                const int col = wave * 4 + (k * 2 + (lane >> 5)), wl = col & 3, hl = col >> 2;
                half_t* o = base + (((long)hl * 64 + wl) * 16) * 32 + (lane & 31) * 16;
                *(u4_t*)o = (u4_t){(unsigned)it, (unsigned)tid, 0u, 0u};
                *(u4_t*)(o + 8) = (u4_t){(unsigned)it, (unsigned)tid, 1u, 1u};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <int PATTERN>
static void run_store(half_t* y, int wgs_per_cu, int ncu, const char* name)
{
    const int iters = 64, ntiles = 256 * 16;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(tile_store<PATTERN>, dim3(ncu * wgs_per_cu), dim3(256), 0, 0, y, 4, ntiles);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(tile_store<PATTERN>, dim3(ncu * wgs_per_cu), dim3(256), 0, 0, y, iters, ntiles);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)ncu * wgs_per_cu * iters * 256 * 64;
    printf("%-34s wgs/CU %d : %7.1f GB/s chip  %6.2f B/ns/CU  %6.2f us per 16 KB tile per WG\n", name, wgs_per_cu, bytes / ms / 1e6,
           bytes / ms / 1e6 / ncu, ms * 1e3 / iters);
}

template <bool DMA, bool DFAST>
static void run(const half_t* x, unsigned* sink, int wgs_per_cu, int ncu, const char* name)
{
    const size_t lds = 160 * 1024 / wgs_per_cu - 1024;     // occupies 1/wgs of the LDS: exactly wgs workgroups fit
    auto k = halo_fetch<DMA, DFAST>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int iters = 64, ntiles = 256 * 16;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(ncu * wgs_per_cu), dim3(256), lds, 0, x, 4, ntiles, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(ncu * wgs_per_cu), dim3(256), lds, 0, x, iters, ntiles, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)ncu * wgs_per_cu * iters * PIECES * 16;
    printf("%-34s wgs/CU %d : %7.1f GB/s chip  %6.2f B/ns/CU  %6.2f us per halo per WG\n", name, wgs_per_cu, bytes / ms / 1e6,
           bytes / ms / 1e6 / ncu, ms * 1e3 / iters);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    half_t* x;
    unsigned* sink;
    const size_t n = (size_t)16 * 64 * 64 * 16 * 32;
    hipMalloc(&x, n * sizeof(half_t)); hipMemset(x, 0, n * sizeof(half_t)); hipMalloc(&sink, 4);
    printf("CUs %d, halo %d voxels = %d KB\n", ncu, V, V * 64 / 1024);
    for (int w : {1, 2, 3, 4, 6}) {
        run<true, false>(x, sink, w, ncu, "LDS-DMA, w-fastest (64 B @ 1 KiB)");
        run<true, true>(x, sink, w, ncu, "LDS-DMA, d-fastest (1 KiB runs)");
        run<false, false>(x, sink, w, ncu, "load+ds_write, w-fastest");
        run<false, true>(x, sink, w, ncu, "load+ds_write, d-fastest");
    }
    for (int w : {1, 2, 3, 4, 8}) {
        run_store<0>(x, w, ncu, "store 8 B/lane, epilogue pattern");
        run_store<1>(x, w, ncu, "store 16 B/lane, 1 KiB runs");
    }
    return 0;
}
