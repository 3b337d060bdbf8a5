// HBM-bound kernels of the CanonSwap generator path for gfx950: trilinear grid-sample of the
// feature volume, the fused sparse-motion sampler, normalisation statistics / application, pooling,
// layout conversion at the API boundary, and the per-identity weight modulation.
// Each kernel cites the reference lines it replaces. Internal layouts:
//   feature volumes  [N][H][W][D=16][C=32]   ("HWDC": the reference's view(bs, c*d, h, w) is free)
//   dense-motion     [N][D][H][W][C]
//   2D feature maps  [N][H][W][C]
#include "common.h"

#define LAUNCH_CHECK(name)                                                        \
    do {                                                                          \
        hipError_t _e = hipGetLastError();                                        \
        if (_e != hipSuccess) { cs_set_error(name ": %s", hipGetErrorString(_e)); return -1; } \
    } while (0)

const half_t* cs_zero_page()
{
    static void* zp[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!zp[dev]) {
        if (hipMalloc(&zp[dev], 256) != hipSuccess) return nullptr;
        (void)hipMemset(zp[dev], 0, 256);
    }
    return (const half_t*)zp[dev];
}

static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// first encoder layer: conv 3x3 3->64 + folded BN + ReLU (appearance_feature_extractor.py:22,39;
// util.py:207-211). K = 27 is too small for MFMA: direct convolution, fp32 NCHW in, fp16 NHWC out.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv_first_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                         const float* __restrict__ b, half_t* __restrict__ out, int N, int H, int W)
{
    __shared__ float ws[64 * 27 + 64];
    for (int i = threadIdx.x; i < 64 * 27 + 64; i += 256) ws[i] = i < 64 * 27 ? w[i] : b[i - 64 * 27];
    __syncthreads();
    const long pix = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int cg = threadIdx.x >> 6;  // 16 output channels per thread
    if (pix >= (long)N * H * W) return;
    const int x = pix % W, y = (pix / W) % H, n = pix / ((long)W * H);
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xx = x + kx - 1;
                in[c * 9 + ky * 3 + kx] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                                              ? img[(((long)n * 3 + c) * H + yy) * W + xx] : 0.f;
            }
    half_t o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int oc = cg * 16 + j;
        float a = ws[64 * 27 + oc];
#pragma unroll
        for (int k = 0; k < 27; ++k) a = fmaf(in[k], ws[oc * 27 + k], a);
        o[j] = (half_t)fmaxf(a, 0.f);
    }
    uint4* dst = (uint4*)(out + pix * 64 + cg * 16);
    dst[0] = *(uint4*)&o[0];
    dst[1] = *(uint4*)&o[8];
}

int launch_conv_first(const float* img, const float* w, const float* b, half_t* out, int N, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(conv_first_kernel, dim3(cdiv((long)N * H * W, 64)), dim3(256), 0, st, img, w, b, out, N, H, W);
    LAUNCH_CHECK("conv_first");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// AvgPool over (H,W) 2x2 (nn.AvgPool2d(2) util.py:158,165; nn.AvgPool3d((1,2,2)) util.py:183,189)
// in: contiguous [N][D][H][W][C] fp16; out: strided view (so it can land inside a concat buffer)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) avgpool_kernel(const half_t* __restrict__ in, int N, int D, int H, int W, int C, TDesc out)
{
    const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * D * Ho * Wo * C8;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = i % C8; i /= C8;
    const int x = i % Wo; i /= Wo;
    const int y = i % Ho; i /= Ho;
    const int d = i % D;
    const int n = i / D;
    const half_t* src = in + ((((long)n * D + d) * H + 2 * y) * W + 2 * x) * C + c8 * 8;
    const h8_t a = *(const h8_t*)src, b = *(const h8_t*)(src + C), c = *(const h8_t*)(src + (long)W * C),
               e = *(const h8_t*)(src + (long)W * C + C);
    h8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(((float)a[j] + (float)b[j] + (float)c[j] + (float)e[j]) * 0.25f);
    *(h8_t*)((half_t*)out.p + (long)n * out.sN + (long)d * out.sD + (long)y * out.sH + (long)x * out.sW + c8 * 8) = o;
}

int launch_avgpool(const half_t* in, int N, int D, int H, int W, int C, TDesc out, hipStream_t st)
{
    const long total = (long)N * D * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(avgpool_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, in, N, D, H, W, C, out);
    LAUNCH_CHECK("avgpool");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// dense motion front end
// ------------------------------------------------------------------------------------------------
// compress 1x1x1 conv 32->4 + BN3d(eval, folded) + ReLU  (dense_motion.py:70-72)
// in: fp32 HWDC feature; out: fp16 [N][D][H][W][4]
__global__ void __launch_bounds__(256) dm_compress_kernel(const float* __restrict__ f, const float* __restrict__ w,
                                                          const float* __restrict__ b, half_t* __restrict__ comp, int N, int D, int H, int W)
{
    __shared__ float ws[4 * 32 + 4];
    if (threadIdx.x < 132) ws[threadIdx.x] = threadIdx.x < 128 ? w[threadIdx.x] : b[threadIdx.x - 128];
    __syncthreads();
    long i = (long)blockIdx.x * 256 + threadIdx.x;  // over HWDC voxel order (coalesced reads)
    const long total = (long)N * H * W * D;
    if (i >= total) return;
    const float4* src = (const float4*)(f + i * 32);
    float a0 = ws[128], a1 = ws[129], a2 = ws[130], a3 = ws[131];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 v = src[j];
        const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a0 = fmaf(x[r], ws[0 * 32 + j * 4 + r], a0);
            a1 = fmaf(x[r], ws[1 * 32 + j * 4 + r], a1);
            a2 = fmaf(x[r], ws[2 * 32 + j * 4 + r], a2);
            a3 = fmaf(x[r], ws[3 * 32 + j * 4 + r], a3);
        }
    }
    const int d = i % D; long r = i / D;
    const int x = r % W; r /= W;
    const int y = r % H;
    const int n = r / H;
    h4_t o;
    o[0] = (half_t)fmaxf(a0, 0.f); o[1] = (half_t)fmaxf(a1, 0.f); o[2] = (half_t)fmaxf(a2, 0.f); o[3] = (half_t)fmaxf(a3, 0.f);
    *(h4_t*)(comp + ((((long)n * D + d) * H + y) * W + x) * 4) = o;
}

int launch_dm_compress(const float* f, const float* w, const float* b, half_t* comp, int N, int D, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(dm_compress_kernel, dim3(cdiv((long)N * D * H * W, 256)), dim3(256), 0, st, f, w, b, comp, N, D, H, W);
    LAUNCH_CHECK("dm_compress");
    return 0;
}

__device__ __forceinline__ float sqdist3(float a, float b, float c)
{
#pragma clang fp contract(off)
    return (a * a + b * b) + c * c;
}
__device__ __forceinline__ float grid_coord(int i, int n) { return 2.f * ((float)i / (float)(n - 1)) - 1.f; }  // util.py:48-50

// Fused create_sparse_motions + create_deformed_feature + create_heatmap_representations + concat
// (dense_motion.py:29-65,77-84). The 22x(16x64x64x3) grid tensor is never materialised: each thread
// rebuilds its sampling point from the key-points. Output: 112 fp16 channels per voxel
// (channel k*5 = heat-map_k, k*5+1..4 = deformed feature_k, 110/111 = zero pad) at pixel stride `ostride`.
// A workgroup owns one x-row (W <= 64 voxels).  Lanes run along x for a fixed key-point slot, so the 8 corner fetches of the
// trilinear sample are contiguous across the wave (with the slot as the fastest index every lane sampled a different motion: 64
// separate lines per fetch instruction, the kernel ran at the L1's line rate, 0.49 ms per call at 32 frames).  The 112 fp16 values of
// a voxel are assembled in an LDS row image and leave as 16-byte stores.
__global__ void __launch_bounds__(256) dm_sparse_kernel(const half_t* __restrict__ comp, const float* __restrict__ kp_d,
                                                        const float* __restrict__ kp_s, half_t* __restrict__ out, int ostride,
                                                        int N, int D, int H, int W, long comp_sN, int kps_sN)
{
    // comp_sN / kps_sN: sample strides of the compressed volume and of kp_s (0: one volume / one key-point set shared by all samples - the
    // v2i body, can_swap_pipeline_v2i.py:311-312)
    constexpr int RS = 120;                                  // LDS row stride in halfs (112 used; 240 bytes: 16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) half_t tile[64 * RS];
    const int t = threadIdx.x, x = t & 63;
    int r = blockIdx.x;
    const int y = r % H; r /= H;
    const int d = r % D;
    const int n = r / D;
    const long v0 = (((long)n * D + d) * H + y) * W;         // first voxel of the row
    const float gx = grid_coord(x, W), gy = grid_coord(y, H), gz = grid_coord(d, D);
    const half_t* base = comp + (long)n * comp_sN;
    // Round 6 (second pass): what a slot's sample shares over the row - its key-points, the y and z coordinates with their corner weights,
    // clamps and offsets - is formed ONCE per wave, by lane k for slot k, and handed to the slot loop through v_readlane (the slot is
    // wave-uniform): the loop keeps the x axis, the heat map and the gather.  The compiler cannot do this itself (no scalar float unit on
    // gfx950: wave-uniform float arithmetic runs on all 64 lanes); the kernel is VALU-bound (198 vector instructions per slot and row before,
    // 0.58 ms per 64-frame call).  The same operations on the same values in the same order: same bits (tools/cmp_libs.py).
    const int lane = t & 63;
    float u_pd[3] = {0.f, 0.f, 0.f}, u_ps[3] = {0.f, 0.f, 0.f};
    if (lane >= 1 && lane <= 21) {
        const float* pd = kp_d + ((long)n * 21 + (lane - 1)) * 3;
        const float* ps = kp_s + (long)n * kps_sN + (lane - 1) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) { u_pd[a] = pd[a]; u_ps[a] = ps[a]; }
    }
    float u_wy[2], u_wz[2]; unsigned u_oyz[4];
    {
        float sy = gy, sz = gz;
        if (lane > 0) { sy = (gy - u_pd[1]) + u_ps[1]; sz = (gz - u_pd[2]) + u_ps[2]; }
        const float iy = ((sy + 1.f) * H - 1.f) * 0.5f, iz = ((sz + 1.f) * D - 1.f) * 0.5f;
        const float fy = floorf(iy), fz = floorf(iz);
        const int y0 = (int)fy, z0 = (int)fz;
        const float ty = iy - fy, tz = iz - fz;
        unsigned oy[2], oz[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int yc = y0 + q, zc = z0 + q;
            u_wy[q] = (unsigned)yc < (unsigned)H ? (q ? ty : 1.f - ty) : 0.f;
            u_wz[q] = (unsigned)zc < (unsigned)D ? (q ? tz : 1.f - tz) : 0.f;
            oy[q] = (unsigned)(min(max(yc, 0), H - 1) * (W * 8));           // byte offsets (4 fp16 channels per voxel)
            oz[q] = (unsigned)(min(max(zc, 0), D - 1) * (H * W * 8));
        }
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) u_oyz[dz * 2 + dy] = oz[dz] + oy[dy];
    }
    for (int k = __builtin_amdgcn_readfirstlane(t >> 6); k < 23; k += 4) {                   // wave-uniform slot
        // (all lanes take part in the lane reads; lanes beyond the row only skip the stores)
        float pd[3], ps[3], wy[2], wz[2]; unsigned oyz[4];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            pd[a] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u_pd[a]), k));
            ps[a] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u_ps[a]), k));
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wy[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u_wy[q]), k));
            wz[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u_wz[q]), k));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) oyz[q] = (unsigned)__builtin_amdgcn_readlane((int)u_oyz[q], k);
        if (x >= W) continue;
        half_t* o = tile + x * RS + k * 5;
        if (k == 22) { o[0] = (half_t)0.f; o[1] = (half_t)0.f; continue; }
        float sx = gx, heat = 0.f;
        if (k > 0) {
            sx = (gx - pd[0]) + ps[0];
            // (the squared distances in the reference's order with every product rounded - torch's (grid - kp) ** 2 summed over the last axis,
            // util.py:31-35: left to the compiler, which of the products fused into an FMA depended on how it happened to pair the axes)
            const float dd = sqdist3(gx - pd[0], gy - pd[1], gz - pd[2]);
            const float ds = sqdist3(gx - ps[0], gy - ps[1], gz - ps[2]);
            heat = __expf(-0.5f * dd / 0.01f) - __expf(-0.5f * ds / 0.01f);   // util.py:36 kp_variance = 0.01
        }
        // F.grid_sample(..., align_corners=False), trilinear, zeros padding (dense_motion.py:50)
        const float ix = ((sx + 1.f) * W - 1.f) * 0.5f;
        const float fx = floorf(ix);
        const int x0 = (int)fx;
        const float tx = ix - fx;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        // The eight corners are fetched unconditionally (clamped address, weight 0 outside the volume: fmaf(0, c, a) == a, the same bits as
        // skipping the corner) so that the loads go out back to back; behind a bounds test each one waited for its own round trip.  Bounds,
        // clamps and address terms are formed per AXIS (six of each, not twenty-four); a corner's weight is ((wx * wy) * wz) with the
        // out-of-range factor replaced by 0 - the same bits as the product of the three factors followed by the bounds select (all
        // factors are >= 0, so the zero is +0 either way; tools/dbg_hash.py printed the same hashes).  Offsets are unsigned 32-bit byte
        // counts from the sample's base (a scalar): the loads take the SGPR-base + VGPR-offset form, no 64-bit address per corner.
        float wx[2]; unsigned ox[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int xc = x0 + q;
            wx[q] = (unsigned)xc < (unsigned)W ? (q ? tx : 1.f - tx) : 0.f;
            ox[q] = (unsigned)(min(max(xc, 0), W - 1) * 8);
        }
        h4_t cv[8]; float wv[8];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    cv[dz * 4 + dy * 2 + dx] = *(const h4_t*)((const char*)base + (oyz[dz * 2 + dy] + ox[dx]));
                    wv[dz * 4 + dy * 2 + dx] = (wx[dx] * wy[dy]) * wz[dz];
                }
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = fmaf(wv[c8], (float)cv[c8][j], a[j]);
        o[0] = (half_t)heat;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[1 + j] = (half_t)a[j];      // (packed as three 4-byte stores hipcc folds the last fma of a[j] into the fp16 conversion -
                                                                  // one rounding instead of two, other bits - for no measurable time: five 2-byte stores stay)
    }
    __syncthreads();
    for (int q = t; q < W * 14; q += 256) {                  // 14 pieces of 16 bytes per voxel
        const int pv = q / 14, pc = q % 14;
        *(uint4*)(out + (v0 + pv) * ostride + pc * 8) = *(const uint4*)(tile + pv * RS + pc * 8);
    }
}

int launch_dm_sparse(const half_t* comp, const float* kp_d, const float* kp_s, half_t* out, int out_stride, int N, int D, int H,
                     int W, hipStream_t st, bool shared_comp, bool shared_kps)
{
    if (((uintptr_t)out & 15) || (out_stride & 7)) { cs_set_error("dm_sparse: output rows must be 16-byte aligned"); return -1; }
    if (W > 64) { cs_set_error("dm_sparse: rows of at most 64 voxels"); return -1; }
    hipLaunchKernelGGL(dm_sparse_kernel, dim3((unsigned)((long)N * D * H)), dim3(256), 0, st, comp, kp_d, kp_s, out, out_stride, N, D, H, W,
                       shared_comp ? 0L : (long)D * H * W * 4, shared_kps ? 0 : 63);
    LAUNCH_CHECK("dm_sparse");
    return 0;
}

// mask conv finish + softmax over the 22 mask logits + deformation = sum_k mask_k * sparse_motion_k
// (dense_motion.py:88-94). The 7x7x7 mask conv runs on the MFMA kernel as a (7,7,1)-tap conv whose 154 output
// channels are (kw, c): part[voxel][kw*22+c] = sum_{kd,kh,cin} pred[d+kd-3][h+kh-3][w][cin] * Wm[c][cin][kd][kh][kw]
// (row stride 160 floats). This kernel adds the seven horizontally shifted partials:
//     logit_c(d,h,w) = bias_c + sum_kw part[(d,h,w+kw-3)][kw*22+c]   (zero padding in w),
// then softmax and the motion blend. deformation out: fp32 [N][D][H][W][3]; optional mask out [N][22][D][H][W].
// the 22 mask logits of voxel v = (n, d, y, x): bias + the mask conv's partials.  compact 0: part[voxel][kw * 22 + c] of the 7 voxels
// x - 3 .. x + 3 (ConvParams::out0 of the kw-split conv);  compact 1: the in-tile sums part[((n D + d) H + y) * (W / 2) + T][j][22] of the
// (up to) four 2-column tiles T whose 8 output columns 2 T - 3 .. 2 T + 4 include x (ConvParams::kw_out), T ascending: a fixed order
__device__ __forceinline__ void dm_logits(float (&l)[22], const float* __restrict__ part, const float* __restrict__ bias, long v, int x, int W,
                                          int compact)
{
#pragma unroll
    for (int k = 0; k < 22; ++k) l[k] = bias[k];
    if (compact == 2) {
        // the 4-column tiles' sums: part[((n D + d) H + y) * (W / 4) + T][j][22], j <-> output column 4 T - 3 + j (10 per tile); x lies in the
        // reach of (up to) three tiles, T ascending: a fixed order
        const long row = (v - x) / 4 * 220;
        const int T0 = (x + 1) / 4 - 1;                      // = ceil((x - 6) / 4): the smallest T with 4 T + 6 >= x
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int T = T0 + dt, j = x - 4 * T + 3;
            if ((unsigned)T < (unsigned)(W >> 2) && (unsigned)j < 10u) {
                const float2* src = (const float2*)(part + row + (long)T * 220 + j * 22);
#pragma unroll
                for (int k = 0; k < 11; ++k) { const float2 q = src[k]; l[2 * k] += q.x; l[2 * k + 1] += q.y; }
            }
        }
        return;
    }
    if (compact) {
        const long row = (v - x) / 2 * 176;                  // first tile of this (n, d, y) row: (v - x) is the row's first voxel, W / 2 tiles x 176
        const int T0 = (x >> 1) - 2 + (x & 1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int T = T0 + dt;
            if ((unsigned)T < (unsigned)(W >> 1)) {
                const float2* src = (const float2*)(part + row + (long)T * 176 + (x - 2 * T + 3) * 22);
#pragma unroll
                for (int j = 0; j < 11; ++j) { const float2 q = src[j]; l[2 * j] += q.x; l[2 * j + 1] += q.y; }
            }
        }
        return;
    }
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
        const int xx = x + kw - 3;
        if ((unsigned)xx < (unsigned)W) {
            const float2* src = (const float2*)(part + (v + kw - 3) * 160 + kw * 22);
#pragma unroll
            for (int j = 0; j < 11; ++j) { const float2 q = src[j]; l[2 * j] += q.x; l[2 * j + 1] += q.y; }
        }
    }
}

__global__ void __launch_bounds__(256) dm_softmax_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                         const float* __restrict__ kp_d, const float* __restrict__ kp_s,
                                                         float* __restrict__ deform, float* __restrict__ mask_out, int N, int D, int H, int W,
                                                         int compact, int kps_sN)
{
    const long total = (long)N * D * H * W;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= total) return;
    const int x = v % W; long r = v / W;
    const int y = r % H; r /= H;
    const int d = r % D;
    const int n = r / D;
    float l[22];
    dm_logits(l, part, bias, v, x, W, compact);
    float mx = l[0];
#pragma unroll
    for (int k = 1; k < 22; ++k) mx = fmaxf(mx, l[k]);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 22; ++k) { l[k] = __expf(l[k] - mx); sum += l[k]; }
    const float inv = 1.f / sum;
    const float gx = grid_coord(x, W), gy = grid_coord(y, H), gz = grid_coord(d, D);
    float ox = gx * (l[0] * inv), oy = gy * (l[0] * inv), oz = gz * (l[0] * inv);
#pragma unroll
    for (int k = 1; k < 22; ++k) {
        const float* pd = kp_d + ((long)n * 21 + (k - 1)) * 3;
        const float* ps = kp_s + (long)n * kps_sN + (k - 1) * 3;
        const float m = l[k] * inv;
        ox = fmaf(m, (gx - pd[0]) + ps[0], ox);
        oy = fmaf(m, (gy - pd[1]) + ps[1], oy);
        oz = fmaf(m, (gz - pd[2]) + ps[2], oz);
    }
    float* o = deform + v * 3;
    o[0] = ox; o[1] = oy; o[2] = oz;
    if (mask_out) {
        const long dhw = (long)D * H * W, sp = ((long)d * H + y) * W + x;
#pragma unroll
        for (int k = 0; k < 22; ++k) mask_out[((long)n * 22 + k) * dhw + sp] = l[k] * inv;
    }
}

int launch_dm_softmax(const float* part, const float* bias, const float* kp_d, const float* kp_s, float* deform, float* mask_out,
                      int N, int D, int H, int W, hipStream_t st, int compact, bool shared_kps)
{
    if ((compact == 1 && (W & 1)) || (compact == 2 && (W & 3))) { cs_set_error("dm_softmax: the compact partial layouts need a width that is a multiple of the tile's columns"); return -1; }
    hipLaunchKernelGGL(dm_softmax_kernel, dim3(cdiv((long)N * D * H * W, 256)), dim3(256), 0, st, part, bias, kp_d, kp_s,
                       deform, mask_out, N, D, H, W, compact, shared_kps ? 0 : 63);
    LAUNCH_CHECK("dm_softmax");
    return 0;
}

// dm_softmax + the feature warp that consumes its deformation in ONE kernel (dense_motion.py:88-94 -> warping_network.py:46-62): the
// sampling grid never goes through HBM (SURVEY 8d prices it at 0 bytes then) and one dependent launch disappears.  A workgroup owns the
// 16 (w) x 16 (d) voxels of one (n, h): phase 1, one thread per voxel, is dm_softmax_kernel's arithmetic verbatim (same order of
// operations: same bits) and leaves (x, y, z) in LDS; phase 2 gathers with 8 lanes x float4 per voxel exactly like grid_sample_kernel,
// voxels in (w, d) order so that a wave writes 8 consecutive depth slices of one column = 1 KiB contiguous.  XCD-aware block order:
// every XCD walks a contiguous eighth of the output (neighbouring voxels share source lines: they meet in one L2).
__global__ void __launch_bounds__(256) dm_softmax_warp_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                              const float* __restrict__ kp_d, const float* __restrict__ kp_s,
                                                              const float* __restrict__ in, float* __restrict__ out32, half_t* __restrict__ out16,
                                                              float* __restrict__ deform, int N, int D, int H, int W, int compact,
                                                              long in_sN, int kps_sN)
{
    __shared__ float defs[256 * 3];
    long blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) blk = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int nwb = W >> 4;
    const int wb = (int)(blk % nwb); long r = blk / nwb;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    const int t = threadIdx.x;
    {   // ---- phase 1: softmax over the 22 mask logits and the motion blend for voxel (d, x) = (t >> 4, wb * 16 + (t & 15))
        const int d = t >> 4, x = wb * 16 + (t & 15);
        const long v = (((long)n * D + d) * H + y) * W + x;
        float l[22];
        dm_logits(l, part, bias, v, x, W, compact);
        float mx = l[0];
#pragma unroll
        for (int k = 1; k < 22; ++k) mx = fmaxf(mx, l[k]);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 22; ++k) { l[k] = __expf(l[k] - mx); sum += l[k]; }
        const float inv = 1.f / sum;
        const float gx = grid_coord(x, W), gy = grid_coord(y, H), gz = grid_coord(d, D);
        float ox = gx * (l[0] * inv), oy = gy * (l[0] * inv), oz = gz * (l[0] * inv);
#pragma unroll
        for (int k = 1; k < 22; ++k) {
            const float* pd = kp_d + ((long)n * 21 + (k - 1)) * 3;
            const float* ps = kp_s + (long)n * kps_sN + (k - 1) * 3;
            const float m = l[k] * inv;
            ox = fmaf(m, (gx - pd[0]) + ps[0], ox);
            oy = fmaf(m, (gy - pd[1]) + ps[1], oy);
            oz = fmaf(m, (gz - pd[2]) + ps[2], oz);
        }
        defs[t * 3] = ox; defs[t * 3 + 1] = oy; defs[t * 3 + 2] = oz;
        if (deform) { float* o = deform + v * 3; o[0] = ox; o[1] = oy; o[2] = oz; }
    }
    __syncthreads();
    // ---- phase 2: trilinear gather (grid_sample_kernel's arithmetic), voxel j = (w, d) = (j >> 4, j & 15) -> 8 lanes x float4
    const int cg = t & 7;
    const float* base = in + (long)n * in_sN + cg * 4;         // in_sN = 0: every sample warps the one shared volume (v2i)
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const int j = it * 32 + (t >> 3), wl = j >> 4, d = j & 15;
        const float* g = defs + ((d << 4) | wl) * 3;
        const float ix = ((g[0] + 1.f) * W - 1.f) * 0.5f, iy = ((g[1] + 1.f) * H - 1.f) * 0.5f, iz = ((g[2] + 1.f) * D - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
        const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
        const float tx = ix - fx, ty = iy - fy, tz = iz - fz;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        // all eight corners fetched back to back (clamped address, weight 0 outside: the same bits as skipping them; see dm_sparse_kernel)
        float4 cv[8]; float wv[8];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int xc = x0 + dx, yc = y0 + dy, zc = z0 + dz;
                    const bool inb = (unsigned)xc < (unsigned)W && (unsigned)yc < (unsigned)H && (unsigned)zc < (unsigned)D;
                    const float wgt = (dx ? tx : 1.f - tx) * (dy ? ty : 1.f - ty) * (dz ? tz : 1.f - tz);
                    const int xq = min(max(xc, 0), W - 1), yq = min(max(yc, 0), H - 1), zq = min(max(zc, 0), D - 1);
                    cv[dz * 4 + dy * 2 + dx] = *(const float4*)(base + (((long)yq * W + xq) * D + zq) * 32);
                    wv[dz * 4 + dy * 2 + dx] = inb ? wgt : 0.f;
                }
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            a[0] = fmaf(wv[c8], cv[c8].x, a[0]); a[1] = fmaf(wv[c8], cv[c8].y, a[1]);
            a[2] = fmaf(wv[c8], cv[c8].z, a[2]); a[3] = fmaf(wv[c8], cv[c8].w, a[3]);
        }
        const long vo = ((((long)n * H + y) * W + wb * 16 + wl) * D + d) * 32 + cg * 4;
        if (out32) *(float4*)(out32 + vo) = make_float4(a[0], a[1], a[2], a[3]);
        if (out16) {
            h4_t o; o[0] = (half_t)a[0]; o[1] = (half_t)a[1]; o[2] = (half_t)a[2]; o[3] = (half_t)a[3];
            *(h4_t*)(out16 + vo) = o;
        }
    }
}

int launch_dm_softmax_warp(const float* part, const float* bias, const float* kp_d, const float* kp_s, const float* in, float* out32,
                           half_t* out16, float* deform, int N, int D, int H, int W, hipStream_t st, int compact, bool shared_in, bool shared_kps)
{
    if (D != 16 || (W & 15)) { cs_set_error("dm_softmax_warp: depth 16 and a width that is a multiple of 16"); return -1; }
    hipLaunchKernelGGL(dm_softmax_warp_kernel, dim3((unsigned)((long)N * H * (W >> 4))), dim3(256), 0, st, part, bias, kp_d, kp_s, in,
                       out32, out16, deform, N, D, H, W, compact, shared_in ? 0L : (long)H * W * D * 32, shared_kps ? 0 : 63);
    LAUNCH_CHECK("dm_softmax_warp");
    return 0;
}

// occlusion map, second half (dense_motion.py:98-102). The 7x7 conv over the (c,d)-flattened prediction is run
// on the MFMA conv kernel as a depth-collapsing (16 x 7 x 1)-tap conv whose 7 output channels are the 7
// horizontal taps: part[n][y][xin][kx] = sum_{d,ky,c} pred[d][y+ky-3][xin][c] * w[c*16+d][ky][kx].
// This kernel finishes: occ[y][x] = sigmoid(bias + sum_kx part[y][x+kx-3][kx]).
__global__ void __launch_bounds__(256) occ_finish_kernel(const float* __restrict__ part, float bias, float* __restrict__ occ, int N, int H, int W)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * H * W) return;
    const int x = i % W;
    float s = bias;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
        const int xx = x + kx - 3;
        if ((unsigned)xx < (unsigned)W) s += part[(i + kx - 3) * 16 + kx];
    }
    occ[i] = 1.f / (1.f + __expf(-s));
}

// the same for the 49-tap form (run_dense_motion): part[n][y][x][ky * 7 + kx] (row stride 64 floats) = the 1x1 conv of input position
// (y, x) with tap (ky, kx); occ[y][x] = sigmoid(bias + sum_ky sum_kx part[y + ky - 3][x + kx - 3][ky * 7 + kx]), zero padding, fixed order
__global__ void __launch_bounds__(256) occ_finish49_kernel(const float* __restrict__ part, float bias, float* __restrict__ occ, int N, int H, int W)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * H * W) return;
    const int x = i % W, y = (i / W) % H;
    float s = bias;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
        const int yy = y + ky - 3;
        const bool yok = (unsigned)yy < (unsigned)H;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const int xx = x + kx - 3;
            const bool ok = yok && (unsigned)xx < (unsigned)W;
            const float v = part[(ok ? i + (long)(ky - 3) * W + (kx - 3) : i) * 64 + ky * 7 + kx];
            s += ok ? v : 0.f;
        }
    }
    occ[i] = 1.f / (1.f + __expf(-s));
}

int launch_occ_finish49(const float* part, float bias, float* occ, int N, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(occ_finish49_kernel, dim3(cdiv((long)N * H * W, 256)), dim3(256), 0, st, part, bias, occ, N, H, W);
    LAUNCH_CHECK("occ_finish49");
    return 0;
}

int launch_occ_finish(const float* part, float bias, float* occ, int N, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(occ_finish_kernel, dim3(cdiv((long)N * H * W, 256)), dim3(256), 0, st, part, bias, occ, N, H, W);
    LAUNCH_CHECK("occ_finish");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Feature warp: F.grid_sample(feature, deformation, align_corners=False) (warping_network.py:46-47)
// in/out: fp32 HWDC [N][H][W][D][32]; grid: fp32 [N][D][H][W][3] (x,y,z). 8 lanes cover one voxel's
// 32 channels with float4 accesses, so every tap is one 128-byte coalesced read.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) grid_sample_kernel(const float* __restrict__ in, const float* __restrict__ grid,
                                                          float* __restrict__ out32, half_t* __restrict__ out16, int N, int D, int H, int W, long in_sN)
{
    const long total = (long)N * H * W * D * 8;
    // XCD-aware block order: hardware places workgroup b on XCD b % 8 (each with its own L2); every XCD walks a contiguous eighth of the
    // output, so the source lines that neighbouring output voxels share (the 8 corners of adjacent voxels overlap) are fetched into ONE
    // L2 instead of up to eight (r02: 2.1x over-fetch on the read side, profiles/r02_i_pmc_summary.csv)
    long blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) blk = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    long i = blk * 256 + threadIdx.x;
    if (i >= total) return;
    const int cg = i & 7; long v = i >> 3;     // v: voxel index in HWDC order
    const int d = v % D; long r = v / D;
    const int x = r % W; r /= W;
    const int y = r % H;
    const int n = r / H;
    const float* g = grid + ((((long)n * D + d) * H + y) * W + x) * 3;
    const float ix = ((g[0] + 1.f) * W - 1.f) * 0.5f, iy = ((g[1] + 1.f) * H - 1.f) * 0.5f, iz = ((g[2] + 1.f) * D - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float tx = ix - fx, ty = iy - fy, tz = iz - fz;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const float* base = in + (long)n * in_sN + cg * 4;
    float4 cv[8]; float wv[8];           // all eight corners fetched back to back (see dm_sparse_kernel)
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int xc = x0 + dx, yc = y0 + dy, zc = z0 + dz;
                const bool inb = (unsigned)xc < (unsigned)W && (unsigned)yc < (unsigned)H && (unsigned)zc < (unsigned)D;
                const float wgt = (dx ? tx : 1.f - tx) * (dy ? ty : 1.f - ty) * (dz ? tz : 1.f - tz);
                const int xq = min(max(xc, 0), W - 1), yq = min(max(yc, 0), H - 1), zq = min(max(zc, 0), D - 1);
                cv[dz * 4 + dy * 2 + dx] = *(const float4*)(base + (((long)yq * W + xq) * D + zq) * 32);
                wv[dz * 4 + dy * 2 + dx] = inb ? wgt : 0.f;
            }
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        a[0] = fmaf(wv[c8], cv[c8].x, a[0]); a[1] = fmaf(wv[c8], cv[c8].y, a[1]);
        a[2] = fmaf(wv[c8], cv[c8].z, a[2]); a[3] = fmaf(wv[c8], cv[c8].w, a[3]);
    }
    if (out32) *(float4*)(out32 + v * 32 + cg * 4) = make_float4(a[0], a[1], a[2], a[3]);
    if (out16) {
        h4_t o; o[0] = (half_t)a[0]; o[1] = (half_t)a[1]; o[2] = (half_t)a[2]; o[3] = (half_t)a[3];
        *(h4_t*)(out16 + v * 32 + cg * 4) = o;
    }
}

int launch_grid_sample(const float* in, const float* grid, float* out32, half_t* out16, int N, int D, int H, int W, hipStream_t st, bool shared_in)
{
    hipLaunchKernelGGL(grid_sample_kernel, dim3(cdiv((long)N * D * H * W * 8, 256)), dim3(256), 0, st, in, grid, out32, out16, N, D, H, W,
                       shared_in ? 0L : (long)H * W * D * 32);
    LAUNCH_CHECK("grid_sample");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-(n, channel) mean and 1/sqrt(var+eps) over P positions of a [N][P][C] tensor (biased variance)
// (InstanceNorm2d util.py:286,296; GroupNorm(32,32) util.py:521-523). Two deterministic passes: fixed-order
// partial sums per workgroup, then a fixed-order fp64 finish -- no atomics, so results do not depend on
// scheduling or on the batch size.
// ------------------------------------------------------------------------------------------------
template <bool F32>
__global__ void __launch_bounds__(256) chan_stats_kernel(const void* __restrict__ xin, long P, int C, int ppb, float* __restrict__ partials)
{
    __shared__ float red[256 * 8];
    const int G = C >> 2;            // channel groups of 4
    const int g = threadIdx.x % G, pl = threadIdx.x / G, PL = 256 / G;
    const int n = blockIdx.y;
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < P ? p0 + ppb : P;
    float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (long p = p0 + pl; p < p1; p += PL) {
        float v[4];
        if (F32) {
            const float4 q = *(const float4*)((const float*)xin + ((long)n * P + p) * C + g * 4);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            const h4_t q = *(const h4_t*)((const half_t*)xin + ((long)n * P + p) * C + g * 4);
            v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r] += v[r]; ss[r] = fmaf(v[r], v[r], ss[r]); }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[threadIdx.x * 8 + r] = s[r]; red[threadIdx.x * 8 + 4 + r] = ss[r]; }
    __syncthreads();
    if (pl == 0) {
        for (int j = 1; j < PL; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[r] += red[(j * G + g) * 8 + r]; ss[r] += red[(j * G + g) * 8 + 4 + r]; }
        float* o = partials + (((long)n * gridDim.x + blockIdx.x) * C + g * 4) * 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r * 2] = s[r]; o[r * 2 + 1] = ss[r]; }
    }
}

__global__ void __launch_bounds__(256) chan_stats_finish_kernel(const float* __restrict__ partials, int nblk, int C, int NC, double cnt_inv,
                                                                float eps, float* __restrict__ stats)
{
    // workgroup = 4 channels x 64 lanes over the partials (the kernel is latency-bound: the deeper the fan-out over the partial
    // blocks, the fewer dependent loads per thread); fixed-order fp64 reduction -> deterministic
    __shared__ double red[2][64][4];
    const int cl = threadIdx.x & 3, bl = threadIdx.x >> 2;
    const int i = blockIdx.x * 4 + cl;       // (n, c) flat index, NC multiple of 4
    const int n = i / C, c = i % C;
    double s = 0, ss = 0;
    for (int b = bl; b < nblk; b += 64) {
        const float2 q = *(const float2*)(partials + (((long)n * nblk + b) * C + c) * 2);
        s += q.x; ss += q.y;
    }
    red[0][bl][cl] = s; red[1][bl][cl] = ss;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {       // pairwise tree over the 64 block lanes, same shape every time
        if (bl < o) { red[0][bl][cl] += red[0][bl + o][cl]; red[1][bl][cl] += red[1][bl + o][cl]; }
        __syncthreads();
    }
    if (bl == 0) {
        s = red[0][0][cl]; ss = red[1][0][cl];
        const double mean = s * cnt_inv;
        double var = ss * cnt_inv - mean * mean;
        if (var < 0) var = 0;
        stats[(long)i * 2] = (float)mean;
        stats[(long)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

static inline int stats_ppb(long P, int C) { const int PL = 256 / (C / 4); return P <= 16384 ? PL * 16 : PL * 32; }

long chan_stats_partial_floats(int N, long P, int C) { return (long)N * cdiv(P, stats_ppb(P, C)) * C * 2; }

int launch_chan_stats_finish(const float* partials, int nblk, int N, int C, double cnt_inv, float eps, float* stats, hipStream_t st)
{
    if (C % 16) { cs_set_error("chan_stats_finish: unsupported C=%d", C); return -1; }
    hipLaunchKernelGGL(chan_stats_finish_kernel, dim3((unsigned)((long)N * C / 4)), dim3(256), 0, st, partials, nblk, C, N * C, cnt_inv, eps, stats);
    LAUNCH_CHECK("chan_stats_finish");
    return 0;
}

int launch_chan_stats(const void* x, int is_f32, int N, long P, int C, float eps, float* partials, float* stats, hipStream_t st)
{
    const int G = C / 4;
    if (C % 16 || G > 256 || 256 % G) { cs_set_error("chan_stats: unsupported C=%d", C); return -1; }
    const int ppb = stats_ppb(P, C);
    dim3 grid(cdiv(P, ppb), (unsigned)N);
    if (is_f32) hipLaunchKernelGGL(chan_stats_kernel<true>, grid, dim3(256), 0, st, x, P, C, ppb, partials);
    else hipLaunchKernelGGL(chan_stats_kernel<false>, grid, dim3(256), 0, st, x, P, C, ppb, partials);
    LAUNCH_CHECK("chan_stats");
    hipLaunchKernelGGL(chan_stats_finish_kernel, dim3((unsigned)((long)N * C / 4)), dim3(256), 0, st, partials, (int)grid.x, C, N * C,
                       1.0 / (double)P, eps, stats);
    LAUNCH_CHECK("chan_stats_finish");
    return 0;
}

__device__ __forceinline__ float act_f(float v, int act, float slope)
{
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));       // same form as the conv epilogue (conv_epilogue.h apply_act)
    return v;
}

// GroupNorm(32,32) apply + optional residual + LeakyReLU (util.py:531-540), on fp32 HWDC volumes.
// out32 = lrelu(gn(y) [+ res]); out16 = act2(out32 * s2[i % period2] + t2[i % period2]) (next conv's input).
// Round 6: grid (chunks of a sample, sample) and NU float4 per thread at a stride of 1024 elements (the same 4 channels: their scale / shift are
// formed once), all loads of a thread issued before its arithmetic.  The first form - one float4 per thread, the sample and the period-2 index by
// 64-bit division per thread - ran at 3.8 TB/s on 1.88 GB; the arithmetic per element is unchanged (same bits).
constexpr int NORM_ACT_NU = 4;
__global__ void __launch_bounds__(256) norm_act_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ res, float slope, float* __restrict__ out32,
                                                       half_t* __restrict__ out16, const float* __restrict__ s2,
                                                       const float* __restrict__ t2, unsigned period2, int act2, float slope2, unsigned per_n,
                                                       int split)
{
    const int n = blockIdx.y;
    const unsigned j0 = (blockIdx.x * (256u * NORM_ACT_NU) + threadIdx.x) * 4u;      // element index within the sample of this thread's first float4
    const int c = j0 & 31;
    const long base = (long)n * per_n;
    float4 q[NORM_ACT_NU], t[NORM_ACT_NU];
#pragma unroll
    for (int u = 0; u < NORM_ACT_NU; ++u) {
        const unsigned j = j0 + u * 1024u;
        const long i = base + (j < per_n ? j : 0u);
        q[u] = *(const float4*)(y + i);
        t[u] = res ? *(const float4*)(res + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float* st = stats + ((long)n * 32 + c + r) * 2;
        sc[r] = gn_scale(st[1], gamma[c + r]);
        sh[r] = gn_shift(st[0], sc[r], beta[c + r]);
    }
#pragma unroll
    for (int u = 0; u < NORM_ACT_NU; ++u) {
        const unsigned j = j0 + u * 1024u;
        if (j >= per_n) break;
        const long i = base + j;
        float v[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
        const float rr[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
        h4_t o16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = gn_lrelu(v[r], sc[r], sh[r], rr[r], slope);      // common.h: the same sequence as vol32's transform staging
            v[r] = a;
            if (s2) { const unsigned k = (j + r) % period2; a = a * s2[k] + t2[k]; }     // (per_n is a multiple of period2: the launcher checks)
            o16[r] = (half_t)act_f(a, act2, slope2);
        }
        if (out32) *(float4*)(out32 + i) = make_float4(v[0], v[1], v[2], v[3]);
        if (out16 && !split) *(h4_t*)(out16 + i) = o16;
        if (out16 && split) {      // split precision for the next conv: voxel-wise [hi(32) | lo(32)], hi + lo == value to 2^-22
            h4_t lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) lo[r] = (half_t)(v[r] - (float)o16[r]);     // split mode carries no second affine: o16 = fp16(v)
            half_t* o = out16 + (i >> 5) * 64 + c;
            *(h4_t*)o = o16; *(h4_t*)(o + 32) = lo;
        }
    }
}

// fp32 volume -> split-precision fp16 [hi | lo] per voxel of 32 channels
__global__ void __launch_bounds__(256) split16_kernel(const float* __restrict__ x, half_t* __restrict__ out, long total4)
{
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= total4) return;
    const long i = i4 * 4;
    const float4 q = *(const float4*)(x + i);
    const float v[4] = {q.x, q.y, q.z, q.w};
    h4_t hi, lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) { hi[r] = (half_t)v[r]; lo[r] = (half_t)(v[r] - (float)hi[r]); }
    half_t* o = out + (i >> 5) * 64 + (i & 31);
    *(h4_t*)o = hi; *(h4_t*)(o + 32) = lo;
}

int launch_split16(const float* x, half_t* out, long n, hipStream_t st)
{
    hipLaunchKernelGGL(split16_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, x, out, n / 4);
    LAUNCH_CHECK("split16");
    return 0;
}

int launch_norm_act(const float* y, const float* stats, const float* gamma, const float* beta,
                    const float* res, float slope, float* out32, half_t* out16, const float* s2, const float* t2, int period2,
                    int act2, float slope2, int N, long per_n, hipStream_t st, int split)
{
    if (split && s2) { cs_set_error("norm_act: the split-precision output carries no second affine"); return -1; }
    if (per_n % 32 || per_n >= (1L << 31) || N > 65535 || (s2 && (period2 < 1 || per_n % period2))) {
        cs_set_error("norm_act: %ld elements per sample (whole 32-channel voxels, a multiple of the second affine's period %d), %d samples", per_n, period2, N);
        return -1;
    }
    hipLaunchKernelGGL(norm_act_kernel, dim3(cdiv(per_n / 4, 256 * NORM_ACT_NU), (unsigned)N), dim3(256), 0, st, y, stats, gamma, beta, res, slope,
                       out32, out16, s2, t2, (unsigned)(period2 > 0 ? period2 : 1), act2, slope2, (unsigned)per_n, split);
    LAUNCH_CHECK("norm_act");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// API-boundary layout conversion (the reference passes fp32 NCDHW / NCHW tensors between stages)
// ------------------------------------------------------------------------------------------------
// fp32 NCDHW [N][C][D][H][W] -> fp32 HWDC (+ optional fp16 pre-activation copy act2(x*s2[c]+t2[c]))
__global__ void __launch_bounds__(256) ncdhw_to_hwdc_kernel(const float* __restrict__ in, float* __restrict__ out32,
                                                            half_t* __restrict__ out16, const float* __restrict__ s2,
                                                            const float* __restrict__ t2, int act2, float slope2, int N, int C, int D,
                                                            int H, int W)
{
    // tile: one (n, d, y) row: transpose [C][W] -> [W][C] through LDS
    __shared__ float tile[32][65];
    const int y = blockIdx.x % H, d = (blockIdx.x / H) % D, n = blockIdx.x / (H * D);
    for (int x0 = 0; x0 < W; x0 += 64) {
        for (int j = threadIdx.x; j < C * 64; j += 256) {
            const int c = j / 64, x = j % 64;
            if (x0 + x < W) tile[c][x] = in[((((long)n * C + c) * D + d) * H + y) * W + x0 + x];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < C * 64; j += 256) {
            const int x = j / C, c = j % C;
            if (x0 + x < W) {
                const float v = tile[c][x];
                const long o = ((((long)n * H + y) * W + x0 + x) * D + d) * C + c;
                if (out32) out32[o] = v;
                if (out16) out16[o] = (half_t)act_f(s2 ? v * s2[c] + t2[c] : v, act2, slope2);
            }
        }
        __syncthreads();
    }
}

int launch_ncdhw_to_hwdc(const float* in, float* out32, half_t* out16, const float* s2, const float* t2, int act2, float slope2,
                         int N, int C, int D, int H, int W, hipStream_t st)
{
    if (C != 32) { cs_set_error("ncdhw_to_hwdc: C must be 32"); return -1; }
    hipLaunchKernelGGL(ncdhw_to_hwdc_kernel, dim3((unsigned)((long)N * D * H)), dim3(256), 0, st, in, out32, out16, s2, t2, act2, slope2,
                       N, C, D, H, W);
    LAUNCH_CHECK("ncdhw_to_hwdc");
    return 0;
}

__global__ void __launch_bounds__(256) hwdc_to_ncdhw_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int D,
                                                            int H, int W)
{
    __shared__ float tile[32][65];
    const int y = blockIdx.x % H, d = (blockIdx.x / H) % D, n = blockIdx.x / (H * D);
    for (int x0 = 0; x0 < W; x0 += 64) {
        for (int j = threadIdx.x; j < C * 64; j += 256) {
            const int x = j / C, c = j % C;
            if (x0 + x < W) tile[c][x] = in[((((long)n * H + y) * W + x0 + x) * D + d) * C + c];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < C * 64; j += 256) {
            const int c = j / 64, x = j % 64;
            if (x0 + x < W) out[((((long)n * C + c) * D + d) * H + y) * W + x0 + x] = tile[c][x];
        }
        __syncthreads();
    }
}

int launch_hwdc_to_ncdhw(const float* in, float* out, int N, int C, int D, int H, int W, hipStream_t st)
{
    if (C != 32) { cs_set_error("hwdc_to_ncdhw: C must be 32"); return -1; }
    hipLaunchKernelGGL(hwdc_to_ncdhw_kernel, dim3((unsigned)((long)N * D * H)), dim3(256), 0, st, in, out, N, C, D, H, W);
    LAUNCH_CHECK("hwdc_to_ncdhw");
    return 0;
}

// fp32 NCHW -> fp16 NHWC and back (HW = H*W, multiple of 64; C multiple of 32)
__global__ void __launch_bounds__(256) nchw_to_nhwc16_kernel(const float* __restrict__ in, half_t* __restrict__ out, int N, int C, int HW)
{
    __shared__ float tile[32][65];
    const int pt = blockIdx.x % (HW / 64), ct = (blockIdx.x / (HW / 64)) % (C / 32), n = blockIdx.x / ((HW / 64) * (C / 32));
    for (int j = threadIdx.x; j < 32 * 64; j += 256) {
        const int c = j / 64, p = j % 64;
        tile[c][p] = in[((long)n * C + ct * 32 + c) * HW + pt * 64 + p];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 32 * 64; j += 256) {
        const int p = j / 32, c = j % 32;
        out[((long)n * HW + pt * 64 + p) * C + ct * 32 + c] = (half_t)tile[c][p];
    }
}

int launch_nchw_to_nhwc16(const float* in, half_t* out, int N, int C, int HW, hipStream_t st)
{
    if (C % 32 || HW % 64) { cs_set_error("nchw_to_nhwc16: bad shape"); return -1; }
    hipLaunchKernelGGL(nchw_to_nhwc16_kernel, dim3((unsigned)((long)N * (C / 32) * (HW / 64))), dim3(256), 0, st, in, out, N, C, HW);
    LAUNCH_CHECK("nchw_to_nhwc16");
    return 0;
}

__global__ void __launch_bounds__(256) nhwc16_to_nchw_kernel(const half_t* __restrict__ in, float* __restrict__ out, int N, int C, int HW)
{
    __shared__ float tile[32][65];
    const int pt = blockIdx.x % (HW / 64), ct = (blockIdx.x / (HW / 64)) % (C / 32), n = blockIdx.x / ((HW / 64) * (C / 32));
    for (int j = threadIdx.x; j < 32 * 64; j += 256) {
        const int p = j / 32, c = j % 32;
        tile[c][p] = (float)in[((long)n * HW + pt * 64 + p) * C + ct * 32 + c];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 32 * 64; j += 256) {
        const int c = j / 64, p = j % 64;
        out[((long)n * C + ct * 32 + c) * HW + pt * 64 + p] = tile[c][p];
    }
}

int launch_nhwc16_to_nchw(const half_t* in, float* out, int N, int C, int HW, hipStream_t st)
{
    if (C % 32 || HW % 64) { cs_set_error("nhwc16_to_nchw: bad shape"); return -1; }
    hipLaunchKernelGGL(nhwc16_to_nchw_kernel, dim3((unsigned)((long)N * (C / 32) * (HW / 64))), dim3(256), 0, st, in, out, N, C, HW);
    LAUNCH_CHECK("nhwc16_to_nchw");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-identity precompute for T (adaptive_modulate.py:148-155), once per source identity
// ------------------------------------------------------------------------------------------------
// fc: per layer [W1 512x512][b1 512][W2 512x512][b2 512] fp32; style out [nlayers][512] (reference channel order)
__global__ void __launch_bounds__(512) t_style_kernel(const float* __restrict__ id, const float* __restrict__ fc, float* __restrict__ style)
{
    __shared__ float v[512], hbuf[512];
    const int j = threadIdx.x;
    const float* L = fc + (long)blockIdx.x * (2 * (512 * 512 + 512));
    v[j] = id[j];
    __syncthreads();
    float a = L[512 * 512 + j];
    for (int i = 0; i < 512; ++i) a = fmaf(L[(long)j * 512 + i], v[i], a);
    hbuf[j] = a > 0.f ? a : 0.2f * a;
    __syncthreads();
    const float* L2 = L + 512 * 512 + 512;
    float b = L2[512 * 512 + j];
    for (int i = 0; i < 512; ++i) b = fmaf(L2[(long)j * 512 + i], hbuf[i], b);
    style[(long)blockIdx.x * 512 + j] = b;
}

// T's mask conv (adaptive_modulate.py:118-121,176: 3x3 conv 512 -> 1 + sigmoid on the 64x64 feature map) as a VALU kernel.  On the
// MFMA conv kernel this layer uses one of sixteen output rows and is bound by per-chunk latencies (65 us for 134 MB at 32 frames, 14
// launches per frame batch); it is a memory-bound dot product.  Layout x [N][H][W][512] fp16: a lane owns 8 channels (one 16-byte load
// per position: a wave reads a position's 1 KiB in one instruction), a wave owns SEG = 16 output positions of one row and streams the
// 3 x 18 input positions that reach them; every input position feeds the three outputs w - 1 .. w + 1 of its row offset (9 taps x 4
// v_dot2_f32_f16 with the lane's 72 weights in registers).  The 16 per-lane partial sums are reduced over the 64 lanes by recursive
// halving (17 exchanges; a fixed order).  Weights are read from the conv packing [j * 9 + tap][16][32] (row 0, j = channel / 32).
// out: tmask[(n * H * W + h * W + w) * 4] = sigmoid(sum + bias[0]) (the blend epilogue reads it at stride 4).
typedef _Float16 tm_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float tm_dot8(const uint4& v, const uint4& q, float a)
{
    a = __builtin_amdgcn_fdot2(__builtin_bit_cast(tm_h2, v.x), __builtin_bit_cast(tm_h2, q.x), a, false);
    a = __builtin_amdgcn_fdot2(__builtin_bit_cast(tm_h2, v.y), __builtin_bit_cast(tm_h2, q.y), a, false);
    a = __builtin_amdgcn_fdot2(__builtin_bit_cast(tm_h2, v.z), __builtin_bit_cast(tm_h2, q.z), a, false);
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(tm_h2, v.w), __builtin_bit_cast(tm_h2, q.w), a, false);
}
// SEG: output columns a wave marches over (16; 4 for launches of fewer than 1024 waves - one or two frames -, where 64 x 4 waves of 18 dependent
// column steps leave three quarters of the CUs idle: 10 -> 5 us per launch at one frame).  Per output the same sums in the same order: same bits.
template <int SEG>
__global__ void __launch_bounds__(256, 4) t_mask_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wp, const float* __restrict__ bias,
                                                        float* __restrict__ tmask, int N, int H, int W)
{
    constexpr int RS = 65;                                  // LDS row stride 65 floats: the reduction's reads are conflict-free
    __shared__ float part[4][SEG * RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = W / SEG;
    // XCD-aware order: hardware places workgroup b on XCD b % 8; every XCD walks a contiguous range of rows (the three input rows of
    // neighbouring output rows are then fetched into one L2)
    long blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) blk = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long item = blk * 4 + wave;                       // (n, h, segment); the launcher makes the item count a multiple of 4
    const int sg = (int)(item % nseg); long r = item / nseg;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    const int w0 = sg * SEG;
    // this lane's weights: tap t, channels lane * 8 .. + 7 = packed chunk j = lane / 4, k = (lane % 4) * 8
    uint4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *(const uint4*)(wp + ((long)((lane >> 2) * 9 + t) * 16) * 32 + (lane & 3) * 8);
    const half_t* xb = x + (long)n * H * W * 512 + lane * 8;
    const bool hok0 = h > 0, hok2 = h + 1 < H;
    const long r0 = (long)(hok0 ? h - 1 : h) * W, r1 = (long)h * W, r2 = (long)(hok2 ? h + 1 : h) * W;
    auto fetch = [&](int c, uint4 (&v)[3]) {               // input column w0 - 1 + c of the three rows (zero outside the map)
        const int wi = w0 - 1 + c;
        const bool cin = (unsigned)wi < (unsigned)W;
        const int wq = cin ? wi : w0;
        v[0] = *(const uint4*)(xb + (r0 + wq) * 512);
        v[1] = *(const uint4*)(xb + (r1 + wq) * 512);
        v[2] = *(const uint4*)(xb + (r2 + wq) * 512);
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        if (!cin || !hok0) v[0] = z;
        if (!cin) v[1] = z;
        if (!cin || !hok2) v[2] = z;
    };
    // rolling accumulators: a2 = output c (kw = 0 so far), a1 = output c - 1, a0 = output c - 2 (complete after this column)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    uint4 cur[3], nxt[3];
    fetch(0, cur);
#pragma unroll 2
    for (int c = 0; c < SEG + 2; ++c) {
        if (c + 1 < SEG + 2) fetch(c + 1, nxt);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            a2 = tm_dot8(cur[kh], wt[kh * 3 + 0], a2);
            a1 = tm_dot8(cur[kh], wt[kh * 3 + 1], a1);
            a0 = tm_dot8(cur[kh], wt[kh * 3 + 2], a0);
        }
        if (c >= 2) part[wave][(c - 2) * RS + lane] = a0;
        a0 = a1; a1 = a2; a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) cur[k] = nxt[k];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the wave's own LDS writes (no other wave reads them)
    // lane = output o (lane / 4), quarter q of the 64 channel lanes: 16 partials in a fixed order, then the four quarters
    const int o = lane >> 2, q = lane & 3;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += part[wave][(o < SEG ? o : 0) * RS + q * 16 + i];
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    if (q == 0 && o < SEG) {
        const float y = sum + bias[0];
        tmask[(((long)n * H + h) * W + w0 + o) * 4] = 1.f / (1.f + __expf(-y));
    }
}

// The same layer with vertical reuse for launches of many frames: a wave owns RH output rows of a SEG-column segment and marches over the columns
// of its RH + 2 input rows, so a fetched 16 bytes feed up to nine dot products instead of three and the read out of the L2 per output drops from
// 3 x 18 / 16 = 3.4 KiB to (RH + 2) / RH x (SEG + 2) / SEG = 1.9 KiB (RH = 4, SEG = 8): t_mask_kernel runs at the L2's rate (10.5 TB/s of reads at
// 64 frames; more columns in flight made it slower, profiles/r06_v_dec_phases.txt).  Per output the same dot products in the same order as in
// t_mask_kernel (column by column, kh ascending within a column; the lanes' partial sums reduced the same way): the same bits, so that the
// launcher may choose by launch size (tests/test_gpu_ops.py).
template <int SEG, int RH>
__global__ void __launch_bounds__(256, 2) t_mask_rows_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wp, const float* __restrict__ bias,
                                                             float* __restrict__ tmask, int N, int H, int W)
{
    constexpr int RS = 65;
    __shared__ float part[4][RH][SEG * RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = W / SEG, nhb = H / RH;
    long blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) blk = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long item = blk * 4 + wave;
    const int sg = (int)(item % nseg); long r = item / nseg;
    const int h0 = (int)(r % nhb) * RH;
    const int n = (int)(r / nhb);
    const int w0 = sg * SEG;
    uint4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *(const uint4*)(wp + ((long)((lane >> 2) * 9 + t) * 16) * 32 + (lane & 3) * 8);
    const half_t* xb = x + (long)n * H * W * 512 + lane * 8;
    long roff[RH + 2]; bool rok[RH + 2];
#pragma unroll
    for (int i = 0; i < RH + 2; ++i) {
        const int row = h0 - 1 + i;
        rok[i] = (unsigned)row < (unsigned)H;
        roff[i] = (long)(rok[i] ? row : h0) * W;
    }
    auto fetch = [&](int c, uint4 (&v)[RH + 2]) {          // input column w0 - 1 + c of the RH + 2 rows (zero outside the map)
        const int wi = w0 - 1 + c;
        const bool cin = (unsigned)wi < (unsigned)W;
        const int wq = cin ? wi : w0;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < RH + 2; ++i) v[i] = *(const uint4*)(xb + (roff[i] + wq) * 512);
#pragma unroll
        for (int i = 0; i < RH + 2; ++i) if (!cin || !rok[i]) v[i] = z;
    };
    float a0[RH], a1[RH], a2[RH];
#pragma unroll
    for (int j = 0; j < RH; ++j) { a0[j] = 0.f; a1[j] = 0.f; a2[j] = 0.f; }
    uint4 cur[RH + 2], nxt[RH + 2];
    fetch(0, cur);
#pragma unroll 2
    for (int c = 0; c < SEG + 2; ++c) {
        if (c + 1 < SEG + 2) fetch(c + 1, nxt);
#pragma unroll
        for (int j = 0; j < RH; ++j) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                a2[j] = tm_dot8(cur[j + kh], wt[kh * 3 + 0], a2[j]);
                a1[j] = tm_dot8(cur[j + kh], wt[kh * 3 + 1], a1[j]);
                a0[j] = tm_dot8(cur[j + kh], wt[kh * 3 + 2], a0[j]);
            }
            if (c >= 2) part[wave][j][(c - 2) * RS + lane] = a0[j];
            a0[j] = a1[j]; a1[j] = a2[j]; a2[j] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < RH + 2; ++k) cur[k] = nxt[k];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the wave's own LDS writes (no other wave reads them)
    const int o = lane >> 2, q = lane & 3;
#pragma unroll
    for (int j = 0; j < RH; ++j) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += part[wave][j][(o < SEG ? o : 0) * RS + q * 16 + i];
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        if (q == 0 && o < SEG) {
            const float y = sum + bias[0];
            tmask[(((long)n * H + h0 + j) * W + w0 + o) * 4] = 1.f / (1.f + __expf(-y));
        }
    }
}

int launch_t_mask(const half_t* x, const half_t* wpacked, const float* bias, float* tmask, int N, int H, int W, hipStream_t st)
{
    // (a row-marching variant - a wave owns 8 output rows of a 16-column segment, every fetched 16 bytes feed nine dot products: 86 -> 75 us per
    // 64-frame launch, but 32 waves for one frame - was built in round 4 and removed in round 5: profiles/HISTORY.md)
    if (W % 16 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)wpacked & 15)) { cs_set_error("t_mask: width a multiple of 16, 16-byte aligned tensors"); return -1; }
    const long items = (long)N * H * (W / 16);
    if (items % 4 != 0) { cs_set_error("t_mask: N * H * W / 16 must be a multiple of 4"); return -1; }
    // (the segment length may depend on N: it does not change a bit of the result, tests/test_gpu_ops.py)
    // from four frames of a 64 x 64 map up: 4 output rows x 8 columns per wave (the same bits; 76 -> 58 us per 64-frame launch, and faster at
    // 4, 8, 16 and 32 frames too: profiles/r06_x_t_mask_rows.txt)
    const long ritems = (long)N * (H / 4) * (W / 8);
    if (items >= 1024 && H % 4 == 0 && W % 8 == 0 && ritems % 4 == 0) {
        hipLaunchKernelGGL((t_mask_rows_kernel<8, 4>), dim3((unsigned)(ritems / 4)), dim3(256), 0, st, x, wpacked, bias, tmask, N, H, W);
        LAUNCH_CHECK("t_mask_rows");
        return 0;
    }
    if (items < 1024) hipLaunchKernelGGL(t_mask_kernel<4>, dim3((unsigned)items), dim3(256), 0, st, x, wpacked, bias, tmask, N, H, W);
    else hipLaunchKernelGGL(t_mask_kernel<16>, dim3((unsigned)cdiv(items, 4)), dim3(256), 0, st, x, wpacked, bias, tmask, N, H, W);
    LAUNCH_CHECK("t_mask");
    return 0;
}

int launch_t_style(const float* id, const float* fc, float* style, int nlayers, hipStream_t st)
{
    hipLaunchKernelGGL(t_style_kernel, dim3(nlayers), dim3(512), 0, st, id, fc, style);
    LAUNCH_CHECK("t_style");
    return 0;
}

// wraw: fp32 [512 o][9 taps][512 i] in MEMORY channel order, style already permuted to memory order.
// Writes the demodulated fp16 rows into the fused packed weight [kstep = chunk*9+tap][1024][32],
// packed row of memory out-channel o (kind 1 = modulated) = ((o/16)*2 + 1)*16 + o%16.
__global__ void __launch_bounds__(256) t_modulate_kernel(const float* __restrict__ wraw, const float* __restrict__ style,
                                                         half_t* __restrict__ packed)
{
    __shared__ float red[256];
    const int o = blockIdx.x;
    const float* src = wraw + (long)o * 9 * 512;
    float wv[18];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        const int e = threadIdx.x + 256 * j;     // e = tap*512 + i
        const float m = src[e] * style[e & 511];
        wv[j] = m;
        ss = fmaf(m, m, ss);
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float demod = rsqrtf(red[0] + 1e-8f);
    const int prow = ((o >> 4) * 2 + 1) * 16 + (o & 15);
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        const int e = threadIdx.x + 256 * j;
        const int tap = e >> 9, i = e & 511;
        packed[((long)((i >> 5) * 9 + tap) * 1024 + prow) * 32 + (i & 31)] = (half_t)(wv[j] * demod);
    }
}

int launch_t_modulate(const float* wraw, const float* style, half_t* packed, int layer, hipStream_t st)
{
    (void)layer;
    hipLaunchKernelGGL(t_modulate_kernel, dim3(512), dim3(256), 0, st, wraw, style, packed);
    LAUNCH_CHECK("t_modulate");
    return 0;
}

// parse_output on device (can_swap_e2e.py:314-322): NCHW fp32 -> NHWC u8, clip, *255, truncate
__global__ void __launch_bounds__(256) pack_u8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, int N, int C, int H, int W)
{
    const long total = (long)N * H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long n = i / ((long)H * W), p = i % ((long)H * W);
    for (int c = 0; c < C; ++c) {
        float v = img[(n * C + c) * (long)H * W + p];
        v = fminf(fmaxf(v, 0.f), 1.f) * 255.f;
        v = fminf(fmaxf(v, 0.f), 255.f);
        out[i * C + c] = (uint8_t)v;
    }
}

int launch_pack_u8(const float* img, uint8_t* out, int N, int C, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(pack_u8_kernel, dim3(cdiv((long)N * H * W, 256)), dim3(256), 0, st, img, out, N, C, H, W);
    LAUNCH_CHECK("pack_u8");
    return 0;
}

// prepare_source / prepare_videos on device (can_swap_e2e.py:126-163): NHWC u8 -> NCHW fp32 / 255 (already inside [0,1])
__global__ void __launch_bounds__(256) unpack_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int N, int C, int H, int W)
{
    const long total = (long)N * H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long n = i / ((long)H * W), p = i % ((long)H * W);
    for (int c = 0; c < C; ++c) out[(n * C + c) * (long)H * W + p] = (float)in[i * C + c] / 255.f;
}

int launch_unpack_u8(const uint8_t* in, float* out, int N, int C, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(unpack_u8_kernel, dim3(cdiv((long)N * H * W, 256)), dim3(256), 0, st, in, out, N, C, H, W);
    LAUNCH_CHECK("unpack_u8");
    return 0;
}

__global__ void __launch_bounds__(256) lrelu16_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, long n8, float slope)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    h8_t v = ((const h8_t*)in)[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; v[j] = (half_t)(f > 0.f ? f : f * slope); }
    ((h8_t*)out)[i] = v;
}

int launch_lrelu16(const half_t* in, half_t* out, long n, float slope, hipStream_t st)
{
    hipLaunchKernelGGL(lrelu16_kernel, dim3(cdiv(n / 8, 256)), dim3(256), 0, st, in, out, n / 8, slope);
    LAUNCH_CHECK("lrelu16");
    return 0;
}


// ------------------------------------------------------------------------------------------------
// finish of a cross-workgroup split-K convolution (ConvParams::sk_out): sum of the splits in split order, bias, activation,
// store to out0 (fp16 or fp32, arbitrary position strides)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) splitk_finish_kernel(const ConvParams p, long mtot)
{
    const int cq = p.Cout / 4;                                  // 4 channels per thread
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mtot * cq) return;
    const long pos = i / cq;
    const int c = (int)(i % cq) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.sk_splits; ++k) {
        const float4 q = *(const float4*)(p.sk_out + ((long)k * mtot + pos) * p.Cout_pad + c);
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
    }
    long r = pos;
    const int w = (int)(r % p.W); r /= p.W;
    const int h = (int)(r % p.H); r /= p.H;
    const int d = (int)(r % p.D); r /= p.D;
    const long o = r * p.out0.sN + (long)d * p.out0.sD + (long)h * p.out0.sH + (long)w * p.out0.sW + c;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = act_f(v[k] + (p.bias ? p.bias[c + k] : 0.f), p.act0, p.slope0);
    if (p.out0_f32) *(float4*)((float*)p.out0.p + o) = make_float4(v[0], v[1], v[2], v[3]);
    else {
        h4_t x;
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = (half_t)v[k];
        *(h4_t*)((half_t*)p.out0.p + o) = x;
    }
}

// ... with ConvParams::pool_hw: out0 = AvgPool(1,2,2) of the activated values, on the pooled grid (DownBlock3d, util.py:185-190): a thread finishes
// the four positions of a window and averages their fp32 values in the epilogue's order, (a + b) + (c + d) with b the w + 1 and c the h + 1
// neighbour (conv_epilogue.h, EP_POOL), rounded once - no full-resolution tensor, no pooling launch in latency mode either
__global__ void __launch_bounds__(256) splitk_finish_pool_kernel(const ConvParams p, long mtot)
{
    const int cq = p.Cout / 4;
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (mtot >> 2) * cq) return;
    long r = i / cq;
    const int c = (int)(i % cq) * 4;
    const int w2 = (int)(r % W2); r /= W2;
    const int h2 = (int)(r % H2); r /= H2;
    const int d = (int)(r % p.D); r /= p.D;
    float a[4][4];
    const float* src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long pos = ((r * p.D + d) * p.H + 2 * h2 + (j >> 1)) * p.W + 2 * w2 + (j & 1);
        src[j] = p.sk_out + pos * p.Cout_pad + c;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[j][k] = 0.f;
    }
    // (splits outermost: the four window positions' fetches of a split go out together; per position the splits still add in split order)
    for (int k = 0; k < p.sk_splits; ++k) {
        float4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = *(const float4*)(src[j] + (long)k * mtot * p.Cout_pad);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j][0] += q[j].x; a[j][1] += q[j].y; a[j][2] += q[j].z; a[j][3] += q[j].w; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[j][k] = act_f(a[j][k] + (p.bias ? p.bias[c + k] : 0.f), p.act0, p.slope0);
    const long o = r * p.out0.sN + (long)d * p.out0.sD + (long)h2 * p.out0.sH + (long)w2 * p.out0.sW + c;
    h4_t x;
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = (half_t)(((a[0][k] + a[1][k]) + (a[2][k] + a[3][k])) * 0.25f);
    *(h4_t*)((half_t*)p.out0.p + o) = x;
}

int launch_splitk_finish(const ConvParams& p, hipStream_t st)
{
    if (p.res.p || p.pixscale || p.out1.p || p.stat_out || p.s2) { cs_set_error("split-K finish: only bias + activation + one output"); return -1; }
    const long mtot = (long)p.N * p.D * p.H * p.W;
    if (p.pool_hw) {
        if (p.out0_f32 || (p.H & 1) || (p.W & 1)) { cs_set_error("split-K finish: the pooled form stores fp16 on an even grid"); return -1; }
        hipLaunchKernelGGL(splitk_finish_pool_kernel, dim3(cdiv((mtot >> 2) * (p.Cout / 4), 256)), dim3(256), 0, st, p, mtot);
    } else
        hipLaunchKernelGGL(splitk_finish_kernel, dim3(cdiv(mtot * (p.Cout / 4), 256)), dim3(256), 0, st, p, mtot);
    LAUNCH_CHECK("splitk_finish");
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Ragged last chunk (Cin % 32 == 16): rows of the chunk are re-packed so that K-step s of the chunk multiplies the 16 real channels
// of tap t(s) (k 0..15) and of tap t(s) + 1 (k 16..31) where both lie in one row of PK taps; the odd tap of a row keeps its row
// (k 16..31 are the zero pad channels).  Step order as conv_halo_kernel.h walks it: row-major, pairs first.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pair_ragged_kernel(const half_t* __restrict__ src, half_t* __restrict__ dst, int Cout_pad, int NT,
                                                          int PK)
{
    const int SPR = PK / 2 + PK % 2, NSR = (NT / PK) * SPR;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)NSR * Cout_pad * 32) return;
    const int k = (int)(i & 31);
    const int c = (int)((i >> 5) % Cout_pad);
    const int s = (int)((i >> 5) / Cout_pad);
    const int tapA = (s / SPR) * PK + 2 * (s % SPR);
    const bool paired = (s % SPR) < PK / 2;
    const long a = ((long)tapA * Cout_pad + c) * 32, b = ((long)(tapA + 1) * Cout_pad + c) * 32;
    dst[i] = !paired ? src[a + k] : (k < 16 ? src[a + k] : src[b + k - 16]);
}

int launch_pair_ragged(half_t* w, int Cout_pad, int nchunks, int KD, int KH, int KW, hipStream_t st)
{
    const int NT = KD * KH * KW, PK = KW > 1 ? KW : KH;
    if (nchunks < 1 || PK < 2 || NT % PK) { cs_set_error("pair_ragged: unsupported shape"); return -1; }
    const size_t n = (size_t)NT * Cout_pad * 32;
    half_t* last = w + (size_t)(nchunks - 1) * n;
    half_t* tmp = nullptr;
    CS_CHECK_HIP(hipMalloc((void**)&tmp, n * sizeof(half_t)));
    hipError_t r = hipMemcpyAsync(tmp, last, n * sizeof(half_t), hipMemcpyDeviceToDevice, st);
    if (r == hipSuccess) {
        const int NSR = (NT / PK) * (PK / 2 + PK % 2);
        hipLaunchKernelGGL(pair_ragged_kernel, dim3(cdiv((long)NSR * Cout_pad * 32, 256)), dim3(256), 0, st, tmp, last, Cout_pad, NT, PK);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipStreamSynchronize(st);
    hipFree(tmp);
    if (r != hipSuccess) { cs_set_error("pair_ragged: %s", hipGetErrorString(r)); return -1; }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// debug: largest magnitude of an fp16 channels-last tensor view (CANONSWAP_AMAX=1 in a profiled step: how close every layer's
// stored activations come to the fp16 range limit 65504).  slot: float bits, maximum by integer compare (values are >= 0)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) absmax16_kernel(const half_t* __restrict__ x, TDesc t, int N, int D, int H, int W, int C, unsigned* slot)
{
    const long npos = (long)N * D * H * W;
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npos * (C / 4); i += (long)gridDim.x * 256) {
        long pos = i / (C / 4);
        const int c = (int)(i % (C / 4)) * 4;
        const int w = (int)(pos % W); pos /= W;
        const int h = (int)(pos % H); pos /= H;
        const int d = (int)(pos % D); pos /= D;
        const h4_t v = *(const h4_t*)(x + pos * t.sN + (long)d * t.sD + (long)h * t.sH + (long)w * t.sW + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float a = fabsf((float)v[r]); m = (a == a) ? fmaxf(m, a) : INFINITY; }     // NaN counts as overflow
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(slot, __float_as_uint(m));
}

int launch_absmax16(const half_t* x, TDesc t, int N, int D, int H, int W, int C, unsigned* slot, hipStream_t st)
{
    if (C % 4) return 0;
    hipLaunchKernelGGL(absmax16_kernel, dim3(512), dim3(256), 0, st, x, t, N, D, H, W, C, slot);
    LAUNCH_CHECK("absmax16");
    return 0;
}
