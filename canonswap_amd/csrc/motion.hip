// Motion extractor M: the non-GEMM pieces of ConvNeXtV2-tiny for gfx950.
// Reference: src/modules/convnextv2.py:15-144 (Block :34-46, stem / downsample :64-76, heads :98-106),
// util.py:356-396 (GRN, LayerNorm).  The pointwise convs (pwconv1/2, the 2x2 stride-2 downsample convs after a
// space-to-depth) run on conv_halo; everything here is bandwidth / latency bound and tiny (the whole network is
// 11.6 GFLOP per frame against 2374 for the generator).
//
// Layout: residual stream x fp32 [N][H][W][C].  One wavefront owns one position and its lanes stride over the channels, so
// LayerNorm is a wave reduction and every load is a contiguous 256-byte row.
//
// Precision: the key-points steer the generator's warps, and with fp16 GEMM operands M's outputs are only good to 1e-3
// (41 dB on the generated frame).  M is 0.5 % of the frame's FLOPs, so its GEMMs run in split precision on the same fp16
// MFMA kernel: an activation row of C' values is stored as the 3C' fp16 vector [hi | lo | hi] (hi = fp16(v), lo =
// fp16(v - hi)) and the weights are packed as [W_hi | W_hi | W_lo] along the input-channel axis (pack._pack_M), so one
// conv computes W_hi v_hi + W_hi v_lo + W_lo v_hi with fp32 accumulation (the dropped W_lo v_lo term is 2^-22 relative).
#include "common.h"

namespace {

// store v as the split-precision triple [hi | lo | hi] at channel c of a row of C values (row stride 3C)
__device__ __forceinline__ void store_split(half_t* row, int C, int c, float v)
{
    const half_t hi = (half_t)v;
    row[c] = hi; row[C + c] = (half_t)(v - (float)hi); row[2 * C + c] = hi;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int MAXK = 12;   // channels per lane: 768 / 64

// LayerNorm over C values spread as v[k] = channel lane + 64 k (biased variance, eps inside the sqrt: util.py:388-396)
template <int K>
__device__ __forceinline__ void wave_layernorm(float (&v)[K], int C, int lane, float eps)
{
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) if (lane + 64 * k < C) s += v[k];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) if (lane + 64 * k < C) { const float d = v[k] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = (v[k] - mean) * rstd;
}

// stem: Conv2d(3, 96, k=4, s=4) + LayerNorm(channels_first) (convnextv2.py:64-68). img fp32 NCHW -> x fp32 NHWC.
// w: [48][96] with k = ci*16 + dy*4 + dx.
__global__ void __launch_bounds__(256) m_stem_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ b,
                                                     const float* __restrict__ g, const float* __restrict__ be, float* __restrict__ x,
                                                     int N, int HI, int WI)
{
    const int lane = threadIdx.x & 63;
    const int HO = HI / 4, WO = WI / 4;
    const long pos = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pos >= (long)N * HO * WO) return;
    const int wo = pos % WO, ho = (pos / WO) % HO, n = pos / ((long)WO * HO);
    float v[2] = {0.f, 0.f};
    for (int k = 0; k < 48; ++k) {
        const int ci = k >> 4, dy = (k >> 2) & 3, dx = k & 3;
        const float a = img[(((long)n * 3 + ci) * HI + ho * 4 + dy) * WI + wo * 4 + dx];      // wave-uniform -> scalar load
        v[0] = fmaf(a, w[k * 96 + lane], v[0]);
        if (lane < 32) v[1] = fmaf(a, w[k * 96 + 64 + lane], v[1]);
    }
    v[0] += b[lane];
    if (lane < 32) v[1] += b[64 + lane];
    wave_layernorm<2>(v, 96, lane, 1e-6f);
    float* o = x + pos * 96;
    o[lane] = v[0] * g[lane] + be[lane];
    if (lane < 32) o[64 + lane] = v[1] * g[64 + lane] + be[64 + lane];
}

// Block front half: depth-wise 7x7 conv (padding 3) + LayerNorm (convnextv2.py:36-38); x fp32 -> y split fp16 [N][H][W][3C].
// wt: [49][C] (tap-major so lanes read consecutive channels).  One wavefront owns a run of DWP consecutive positions of a row:
// every input element of the 7 x (DWP+6) window is loaded once and scattered into the outputs it feeds (4x fewer loads than
// a window per position), the 7 weights of the current kernel row sit in registers.
constexpr int DWP = 8;
template <int K>
__global__ void __launch_bounds__(256) m_dwln_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ b,
                                                     const float* __restrict__ g, const float* __restrict__ be, half_t* __restrict__ y,
                                                     int N, int H, int W, int C)
{
    const int lane = threadIdx.x & 63;
    const int runs_per_row = W / DWP;                                   // W is 64, 32, 16 or 8
    const long run = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (run >= (long)N * H * runs_per_row) return;
    const int w0 = (int)(run % runs_per_row) * DWP, h0 = (int)((run / runs_per_row) % H), n = (int)(run / ((long)runs_per_row * H));
    float acc[DWP][K];
#pragma unroll
    for (int p = 0; p < DWP; ++p)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[p][k] = (lane + 64 * k < C) ? b[lane + 64 * k] : 0.f;
    for (int dy = 0; dy < 7; ++dy) {
        const int h = h0 + dy - 3;
        if ((unsigned)h >= (unsigned)H) continue;
        float wk[7][K];
#pragma unroll
        for (int dx = 0; dx < 7; ++dx)
#pragma unroll
            for (int k = 0; k < K; ++k) wk[dx][k] = (lane + 64 * k < C) ? wt[(dy * 7 + dx) * C + lane + 64 * k] : 0.f;
        const float* xrow = x + ((long)n * H + h) * W * C;
#pragma unroll
        for (int j = 0; j < DWP + 6; ++j) {                             // input column w0 - 3 + j feeds output p with tap dx = j - p
            const int ww = w0 - 3 + j;
            if ((unsigned)ww >= (unsigned)W) continue;
            float xv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) xv[k] = (lane + 64 * k < C) ? xrow[(long)ww * C + lane + 64 * k] : 0.f;
#pragma unroll
            for (int p = 0; p < DWP; ++p) {
                const int dx = j - p;
                if (dx < 0 || dx > 6) continue;
#pragma unroll
                for (int k = 0; k < K; ++k) acc[p][k] = fmaf(xv[k], wk[dx][k], acc[p][k]);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < DWP; ++p) {
        float v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = acc[p][k];
        wave_layernorm<K>(v, C, lane, 1e-6f);
        half_t* o = y + ((((long)n * H + h0) * W) + w0 + p) * 3 * C;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (lane + 64 * k < C) store_split(o, C, lane + 64 * k, v[k] * g[lane + 64 * k] + be[lane + 64 * k]);
    }
}

// Downsample front half: LayerNorm(channels_first) + space-to-depth for the 2x2 stride-2 conv (convnextv2.py:70-75):
// x fp32 [N][H][W][C] -> y split fp16 [N][H/2][W/2][3 x 4C], inner index (dy*2+dx)*C + c.
template <int K>
__global__ void __launch_bounds__(256) m_ln_s2d_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ be,
                                                       half_t* __restrict__ y, int N, int H, int W, int C)
{
    const int lane = threadIdx.x & 63;
    const long pos = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pos >= (long)N * H * W) return;
    const int w0 = pos % W, h0 = (pos / W) % H, n = pos / ((long)W * H);
    float v[K];
    const float* xr = x + pos * C;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = (lane + 64 * k < C) ? xr[lane + 64 * k] : 0.f;
    wave_layernorm<K>(v, C, lane, 1e-6f);
    half_t* o = y + (((long)n * (H / 2) + (h0 >> 1)) * (W / 2) + (w0 >> 1)) * 12 * C;
    const int sub = ((h0 & 1) * 2 + (w0 & 1)) * C;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (lane + 64 * k < C) store_split(o, 4 * C, sub + lane + 64 * k, v[k] * g[lane + 64 * k] + be[lane + 64 * k]);
}

// GRN statistics (util.py:365-367): sumsq[n][c] = sum over the P positions of h[n][p][c]^2. Block = 64 channels x 4 position lanes.
__global__ void __launch_bounds__(256) m_grn_sumsq_kernel(const float* __restrict__ h, float* __restrict__ sumsq, int P, int C)
{
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
    float s = 0.f;
    for (int p = pl; p < P; p += 4) {
        const float v = h[((long)n * P + p) * C + c];
        s = fmaf(v, v, s);
    }
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0) sumsq[(long)n * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// scale[n][c] = 1 + gamma[c] * Gx / (mean_c(Gx) + 1e-6), Gx = sqrt(sumsq)  (util.py:366-368: gamma*(x*Nx) + beta + x)
__global__ void __launch_bounds__(256) m_grn_scale_kernel(const float* __restrict__ sumsq, const float* __restrict__ gamma,
                                                          float* __restrict__ scale, int C)
{
    __shared__ float red[256];
    const int n = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += sqrtf(sumsq[(long)n * C + c]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float inv = 1.f / (red[0] / (float)C + 1e-6f);
    for (int c = threadIdx.x; c < C; c += 256) scale[(long)n * C + c] = 1.f + gamma[c] * sqrtf(sumsq[(long)n * C + c]) * inv;
}

// out[pos] = split(h * scale[n][c] + beta[c]): fp32 [N][P][C] -> split fp16 [N][P][3C], 4 channels per thread
__global__ void __launch_bounds__(256) m_grn_apply_kernel(const float* __restrict__ h, const float* __restrict__ scale, const float* __restrict__ beta,
                                                          half_t* __restrict__ out, long per_n, int C, long total4)
{
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= total4) return;
    const long i = i4 * 4;
    const int n = i / per_n, c = i % C;
    const long pos = i / C;
    const float4 q = *(const float4*)(h + i);
    const float v[4] = {q.x, q.y, q.z, q.w};
    h4_t hi, lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a = v[r] * scale[(long)n * C + c + r] + beta[c + r];
        hi[r] = (half_t)a; lo[r] = (half_t)(a - (float)hi[r]);
    }
    half_t* o = out + pos * 3 * C + c;
    *(h4_t*)o = hi; *(h4_t*)(o + C) = lo; *(h4_t*)(o + 2 * C) = hi;
}

// Global average pool + final LayerNorm + the 7 linear heads (convnextv2.py:114-131). x fp32 [N][P][768] -> out fp32 [N][328]
// in the order kp(63) scale(1) pitch(66) yaw(66) roll(66) t(3) exp(63). hw: [328][768].
__global__ void __launch_bounds__(256) m_head_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ be,
                                                     const float* __restrict__ hw, const float* __restrict__ hb, float* __restrict__ out, int P)
{
    __shared__ float feat[768];
    __shared__ float red[256];
    const int n = blockIdx.x, t = threadIdx.x;
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += x[((long)n * P + p) * 768 + t + 256 * k];
        v[k] = s / (float)P;
    }
    red[t] = v[0] + v[1] + v[2];
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const float mean = red[0] / 768.f;
    __syncthreads();
    red[t] = (v[0] - mean) * (v[0] - mean) + (v[1] - mean) * (v[1] - mean) + (v[2] - mean) * (v[2] - mean);
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const float rstd = rsqrtf(red[0] / 768.f + 1e-6f);
#pragma unroll
    for (int k = 0; k < 3; ++k) feat[t + 256 * k] = (v[k] - mean) * rstd * g[t + 256 * k] + be[t + 256 * k];
    __syncthreads();
    for (int o = t; o < 328; o += 256) {
        const float* wr = hw + (long)o * 768;
        float s = hb[o];
        for (int c = 0; c < 768; ++c) s = fmaf(feat[c], wr[c], s);
        out[(long)n * 328 + o] = s;
    }
}


// Key-points of B frames from the raw head outputs, on the device ("transform_keypoint on-device"):
// can_swapper.get_kp_info's refinement (can_swap_e2e.py:192-197: 66-bin logits -> degrees, camera.py:14-28), get_rotation_matrix
// (camera.py:31-73: R = (Rz Ry Rx)^T of the angles in radians) and transform_keypoint (can_swap_e2e.py:228-256):
//     x_t = scale * (kp R + exp) + (t_x, t_y, 0),     x_can = scale * kp   (can_swap_pipeline_e2e.py:243)
// raw: [B][328] in the order kp(63) scale(1) pitch(66) yaw(66) roll(66) t(3) exp(63).  One wavefront per frame.
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float bins_to_degree(const float* logits, int lane)      // sum(softmax(logits) * idx) * 3 - 97.5
{
    const float a = logits[lane], b = lane < 2 ? logits[64 + lane] : -INFINITY;
    const float m = wave_max(fmaxf(a, b));
    const float ea = expf(a - m), eb = lane < 2 ? expf(b - m) : 0.f;
    const float den = wave_sum(ea + eb);
    const float num = wave_sum(ea / den * (float)lane + eb / den * (float)(64 + lane));
    return num * 3.f - 97.5f;
}

__global__ void __launch_bounds__(64) m_keypoints_kernel(const float* __restrict__ raw, float* __restrict__ x_t, float* __restrict__ x_can,
                                                         float* __restrict__ rot_out)
{
    const int n = blockIdx.x, lane = threadIdx.x;
    const float* r = raw + (long)n * 328;
    const float PI = 3.14159265358979323846f;
    const float x = bins_to_degree(r + 64, lane) / 180.f * PI, y = bins_to_degree(r + 130, lane) / 180.f * PI,
                z = bins_to_degree(r + 196, lane) / 180.f * PI;
    const float cx = cosf(x), sx = sinf(x), cy = cosf(y), sy = sinf(y), cz = cosf(z), sz = sinf(z);
    // rz @ ry, then @ rx (camera.py:53-71), row-major m[i][j]
    const float zy[3][3] = {{cz * cy, -sz, cz * sy}, {sz * cy, cz, sz * sy}, {-sy, 0.f, cy}};
    float m[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        m[i][0] = zy[i][0];
        m[i][1] = zy[i][1] * cx + zy[i][2] * sx;
        m[i][2] = zy[i][1] * -sx + zy[i][2] * cx;
    }
    // R = m^T: (kp R)_j = sum_i kp_i R[i][j] = sum_i kp_i m[j][i]
    if (rot_out && lane < 9) rot_out[(long)n * 9 + lane] = m[lane % 3][lane / 3];
    if (lane >= 21) return;
    const float scale = r[63];
    const float k0 = r[lane * 3], k1 = r[lane * 3 + 1], k2 = r[lane * 3 + 2];
    const float* ex = r + 265 + lane * 3;
    float* xt = x_t + ((long)n * 21 + lane) * 3;
    float* xc = x_can + ((long)n * 21 + lane) * 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float v = ((k0 * m[j][0] + k1 * m[j][1]) + k2 * m[j][2] + ex[j]) * scale;
        xt[j] = j < 2 ? v + r[262 + j] : v;
    }
    xc[0] = scale * k0; xc[1] = scale * k1; xc[2] = scale * k2;
}

}  // namespace

#define M_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { cs_set_error(name ": %s", hipGetErrorString(e_)); return -1; } } while (0)

int launch_m_stem(const float* img, const float* w, const float* b, const float* g, const float* be, float* x, int N, int HI, int WI, hipStream_t st)
{
    const long pos = (long)N * (HI / 4) * (WI / 4);
    hipLaunchKernelGGL(m_stem_kernel, dim3((unsigned)((pos + 3) / 4)), dim3(256), 0, st, img, w, b, g, be, x, N, HI, WI);
    M_LAUNCH_CHECK("m_stem");
    return 0;
}

int launch_m_dwln(const float* x, const float* wt, const float* b, const float* g, const float* be, half_t* y, int N, int H, int W, int C, hipStream_t st)
{
    if (W % DWP) { cs_set_error("m_dwln: width %d is not a multiple of %d", W, DWP); return -1; }
    const dim3 grid((unsigned)(((long)N * H * (W / DWP) + 3) / 4));
    switch ((C + 63) / 64) {
    case 2: hipLaunchKernelGGL(m_dwln_kernel<2>, grid, dim3(256), 0, st, x, wt, b, g, be, y, N, H, W, C); break;
    case 3: hipLaunchKernelGGL(m_dwln_kernel<3>, grid, dim3(256), 0, st, x, wt, b, g, be, y, N, H, W, C); break;
    case 6: hipLaunchKernelGGL(m_dwln_kernel<6>, grid, dim3(256), 0, st, x, wt, b, g, be, y, N, H, W, C); break;
    case 12: hipLaunchKernelGGL(m_dwln_kernel<12>, grid, dim3(256), 0, st, x, wt, b, g, be, y, N, H, W, C); break;
    default: cs_set_error("m_dwln: unsupported C=%d", C); return -1;
    }
    M_LAUNCH_CHECK("m_dwln");
    return 0;
}

int launch_m_ln_s2d(const float* x, const float* g, const float* be, half_t* y, int N, int H, int W, int C, hipStream_t st)
{
    const dim3 grid((unsigned)(((long)N * H * W + 3) / 4));
    switch ((C + 63) / 64) {
    case 2: hipLaunchKernelGGL(m_ln_s2d_kernel<2>, grid, dim3(256), 0, st, x, g, be, y, N, H, W, C); break;
    case 3: hipLaunchKernelGGL(m_ln_s2d_kernel<3>, grid, dim3(256), 0, st, x, g, be, y, N, H, W, C); break;
    case 6: hipLaunchKernelGGL(m_ln_s2d_kernel<6>, grid, dim3(256), 0, st, x, g, be, y, N, H, W, C); break;
    default: cs_set_error("m_ln_s2d: unsupported C=%d", C); return -1;
    }
    M_LAUNCH_CHECK("m_ln_s2d");
    return 0;
}

int launch_m_grn(const float* h, const float* gamma, const float* beta, float* sumsq, float* scale, half_t* out, int N, int P, int C, hipStream_t st)
{
    if (C % 64) { cs_set_error("m_grn: unsupported C=%d", C); return -1; }
    hipLaunchKernelGGL(m_grn_sumsq_kernel, dim3(C / 64, N), dim3(256), 0, st, h, sumsq, P, C);
    M_LAUNCH_CHECK("m_grn_sumsq");
    hipLaunchKernelGGL(m_grn_scale_kernel, dim3(N), dim3(256), 0, st, sumsq, gamma, scale, C);
    M_LAUNCH_CHECK("m_grn_scale");
    const long total4 = (long)N * P * C / 4;
    hipLaunchKernelGGL(m_grn_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, h, scale, beta, out, (long)P * C, C, total4);
    M_LAUNCH_CHECK("m_grn_apply");
    return 0;
}

int launch_m_head(const float* x, const float* g, const float* be, const float* hw, const float* hb, float* out, int N, int P, hipStream_t st)
{
    hipLaunchKernelGGL(m_head_kernel, dim3(N), dim3(256), 0, st, x, g, be, hw, hb, out, P);
    M_LAUNCH_CHECK("m_head");
    return 0;
}

int launch_m_keypoints(const float* raw, float* x_t, float* x_can, float* rot, int N, hipStream_t st)
{
    hipLaunchKernelGGL(m_keypoints_kernel, dim3(N), dim3(64), 0, st, raw, x_t, x_can, rot);
    M_LAUNCH_CHECK("m_keypoints");
    return 0;
}
