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
// MFMA kernel: the weights are packed as [W_hi | W_hi | W_lo] along the input-channel axis (pack._pack_M) and an activation row
// of C' values is stored as the 2C' fp16 vector [hi | lo] (hi = fp16(v), lo = fp16(v - hi)), which the conv reads as the 3C'-channel
// input [hi | lo | hi]: its 32-channel chunk j fetches channels (j mod 2C'/32) * 32 (ConvParams::cg at group stride 0, the addressing
// F's [W_hi | W_lo] convs use) - round 6; before, the third copy was stored and fetched (a third of the operand bytes of a memory-bound
// network).  One conv computes W_hi v_hi + W_hi v_lo + W_lo v_hi with fp32 accumulation (the dropped W_lo v_lo term is 2^-22 relative).
#include "common.h"

namespace {

// store v as the split-precision pair [hi | lo] at channel c of a row of C values (row stride 2C)
__device__ __forceinline__ void store_split(half_t* row, int C, int c, float v)
{
    const half_t hi = (half_t)v;
    row[c] = hi; row[C + c] = (half_t)(v - (float)hi);
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
// Round 6: a workgroup owns one output row (n, ho) of 64 positions: the 3 x 4 input rows (12 KB) and the weights (18 KB) are staged in LDS
// with coalesced loads; thread = (position, 24-channel group), the four groups of a position in four waves, so the weight reads are
// wave-uniform LDS broadcasts and an input patch row is one ds_read_b128.  (Before: one wavefront per position issuing 48 dependent
// scalar loads - 0.62 ms per 64 frames for 150 MB of traffic.)  LayerNorm: two-pass (mean, then deviations) through LDS, fixed order.
__global__ void __launch_bounds__(256) m_stem_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ b,
                                                     const float* __restrict__ g, const float* __restrict__ be, float* __restrict__ x,
                                                     int N, int HI, int WI)
{
    constexpr int WO = 64, CG = 24;
    __shared__ __attribute__((aligned(16))) float in_s[12][WO * 4];      // [ci*4 + dy][input column]
    __shared__ __attribute__((aligned(16))) float w_s[48 * 96];
    __shared__ float red[4][WO];
    const int HO = HI / 4;
    const int ho = blockIdx.x % HO, n = blockIdx.x / HO;
    const int t = threadIdx.x;
    for (int i = t; i < 12 * WO; i += 256) {                            // 12 rows of 256 floats, float4 per thread
        const int r = i / WO, c4 = i % WO, ci = r >> 2, dy = r & 3;
        ((float4*)in_s[r])[c4] = ((const float4*)(img + (((long)n * 3 + ci) * HI + ho * 4 + dy) * WI))[c4];
    }
    for (int i = t; i < 48 * 96 / 4; i += 256) ((float4*)w_s)[i] = ((const float4*)w)[i];
    __syncthreads();
    const int wo = t & 63, cg = t >> 6;
    float v[CG];
#pragma unroll
    for (int j = 0; j < CG; ++j) v[j] = b[cg * CG + j];
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        const float4 a4 = ((const float4*)in_s[r])[wo];
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const float4* wr = (const float4*)(w_s + (r * 4 + dx) * 96 + cg * CG);
#pragma unroll
            for (int j4 = 0; j4 < CG / 4; ++j4) {
                const float4 q = wr[j4];
                v[4 * j4] = fmaf(a[dx], q.x, v[4 * j4]); v[4 * j4 + 1] = fmaf(a[dx], q.y, v[4 * j4 + 1]);
                v[4 * j4 + 2] = fmaf(a[dx], q.z, v[4 * j4 + 2]); v[4 * j4 + 3] = fmaf(a[dx], q.w, v[4 * j4 + 3]);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CG; ++j) s += v[j];
    red[cg][wo] = s;
    __syncthreads();
    const float mean = ((red[0][wo] + red[1][wo]) + (red[2][wo] + red[3][wo])) / 96.f;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CG; ++j) { const float d = v[j] - mean; q += d * d; }
    red[cg][wo] = q;
    __syncthreads();
    const float rstd = rsqrtf(((red[0][wo] + red[1][wo]) + (red[2][wo] + red[3][wo])) / 96.f + 1e-6f);
    float* o = x + (((long)n * HO + ho) * WO + wo) * 96 + cg * CG;
#pragma unroll
    for (int j4 = 0; j4 < CG / 4; ++j4) {
        float4 r4;
        r4.x = (v[4 * j4] - mean) * rstd * g[cg * CG + 4 * j4] + be[cg * CG + 4 * j4];
        r4.y = (v[4 * j4 + 1] - mean) * rstd * g[cg * CG + 4 * j4 + 1] + be[cg * CG + 4 * j4 + 1];
        r4.z = (v[4 * j4 + 2] - mean) * rstd * g[cg * CG + 4 * j4 + 2] + be[cg * CG + 4 * j4 + 2];
        r4.w = (v[4 * j4 + 3] - mean) * rstd * g[cg * CG + 4 * j4 + 3] + be[cg * CG + 4 * j4 + 3];
        ((float4*)o)[j4] = r4;
    }
}

// Block front half: depth-wise 7x7 conv (padding 3) + LayerNorm (convnextv2.py:36-38); x fp32 -> y split fp16 [N][H][W][2C].
// wt: [49][C] (tap-major so lanes read consecutive channels).  One wavefront owns a run of DWP consecutive positions of a row:
// every input element of the 7 x (DWP+6) window is loaded once and scattered into the outputs it feeds (4x fewer loads than
// a window per position), the 7 weights of the current kernel row sit in registers.
constexpr int DWP = 8;
// Round 6: the loads of a kernel row (14 input columns + 7 weights per 64-channel group) are issued together, from clamped addresses with a
// select behind them, before the row's FMAs: the branchy form (a bounds test in front of every load) made hipcc wait for each load's round
// trip in turn - 98 x K dependent latencies per wavefront, 0.28 ms per stage-0 launch for 250 MB.  Channel groups go through in chunks of
// at most 3 (registers); an output's products are still added in the order (dy, dx) ascending: the same bits as before.
template <int K>
__global__ void __launch_bounds__(256) m_dwln_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ b,
                                                     const float* __restrict__ g, const float* __restrict__ be, half_t* __restrict__ y,
                                                     int N, int H, int W, int C)
{
    constexpr int KC = K > 3 ? 3 : K;
    static_assert(K % KC == 0, "channel groups per chunk");
    const int lane = threadIdx.x & 63;
    const int runs_per_row = W / DWP;                                   // W is 64, 32, 16 or 8
#ifdef DW_HROWS      /* A/B: the first form - the four waves of a workgroup own four consecutive runs of one row */
    const long run = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (run >= (long)N * H * runs_per_row) return;
    const int w0 = (int)(run % runs_per_row) * DWP, h0 = (int)((run / runs_per_row) % H), n = (int)(run / ((long)runs_per_row * H));
#else
    // the four waves of a workgroup own the same columns of four consecutive rows: their 7-row windows overlap (10 input rows instead of 28
    // through the CU's L1: 1.32 -> 1.26 ms per 64 frames, same bits); H is a multiple of 4 (launcher)
    const long item = blockIdx.x;
    if (item >= (long)N * (H >> 2) * runs_per_row) return;
    const int w0 = (int)(item % runs_per_row) * DWP, h0 = (int)((item / runs_per_row) % (H >> 2)) * 4 + (int)(threadIdx.x >> 6),
              n = (int)(item / ((long)runs_per_row * (H >> 2)));
#endif
    float acc[DWP][K];
#pragma unroll
    for (int p = 0; p < DWP; ++p)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[p][k] = (lane + 64 * k < C) ? b[lane + 64 * k] : 0.f;
#pragma unroll
    for (int kc = 0; kc < K; kc += KC) {
#pragma unroll 1
        for (int dy = 0; dy < 7; ++dy) {
            const int h = h0 + dy - 3;
            if ((unsigned)h >= (unsigned)H) continue;                   // wave-uniform
            float wk[7][KC], xv[DWP + 6][KC];
#pragma unroll
            for (int dx = 0; dx < 7; ++dx)
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const int c = lane + 64 * (kc + k);
                    wk[dx][k] = wt[(dy * 7 + dx) * C + (c < C ? c : 0)];
                }
            const float* xrow = x + ((long)n * H + h) * W * C;
#pragma unroll
            for (int j = 0; j < DWP + 6; ++j) {                         // input column w0 - 3 + j feeds output p with tap dx = j - p
                const int ww = w0 - 3 + j;
                const int wc = (unsigned)ww < (unsigned)W ? ww : w0;
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const int c = lane + 64 * (kc + k);
                    xv[j][k] = xrow[(long)wc * C + (c < C ? c : 0)];
                }
            }
            // The loads above must stay unconditional and together: with the selects below applied directly to the loaded values hipcc
            // sinks each load under its select's condition - a branch over the load and an s_waitcnt vmcnt(0) behind it, one round trip
            // after the other.  The opaque moves pin "loaded value" as an unconditional use AFTER all loads have been issued.
#pragma unroll
            for (int dx = 0; dx < 7; ++dx)
#pragma unroll
                for (int k = 0; k < KC; ++k) asm volatile("" : "+v"(wk[dx][k]));
#pragma unroll
            for (int j = 0; j < DWP + 6; ++j) {
                const bool in = (unsigned)(w0 - 3 + j) < (unsigned)W;
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    asm volatile("" : "+v"(xv[j][k]));
                    xv[j][k] = (in && lane + 64 * (kc + k) < C) ? xv[j][k] : 0.f;
                }
            }
#pragma unroll
            for (int p = 0; p < DWP; ++p)
#pragma unroll
                for (int dx = 0; dx < 7; ++dx)
#pragma unroll
                    for (int k = 0; k < KC; ++k) acc[p][kc + k] = fmaf(xv[p + dx][k], wk[dx][k], acc[p][kc + k]);
        }
    }
#pragma unroll
    for (int p = 0; p < DWP; ++p) {
        float v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = acc[p][k];
        wave_layernorm<K>(v, C, lane, 1e-6f);
        half_t* o = y + ((((long)n * H + h0) * W) + w0 + p) * 2 * C;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (lane + 64 * k < C) store_split(o, C, lane + 64 * k, v[k] * g[lane + 64 * k] + be[lane + 64 * k]);
    }
}

// Downsample front half: LayerNorm(channels_first) + space-to-depth for the 2x2 stride-2 conv (convnextv2.py:70-75):
// x fp32 [N][H][W][C] -> y split fp16 [N][H/2][W/2][2 x 4C], inner index (dy*2+dx)*C + c.
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
    half_t* o = y + (((long)n * (H / 2) + (h0 >> 1)) * (W / 2) + (w0 >> 1)) * 8 * C;
    const int sub = ((h0 & 1) * 2 + (w0 & 1)) * C;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (lane + 64 * k < C) store_split(o, 4 * C, sub + lane + 64 * k, v[k] * g[lane + 64 * k] + be[lane + 64 * k]);
}

// GRN statistics (util.py:365-367): sumsq[n][z][c] = sum over the z-th slice of the P positions of h[n][p][c]^2.
// Round 6: a workgroup = 64 channels (16 threads x float4) x 16 position lanes, four loads in flight per thread, and the positions of a
// sample are cut into gridDim.z slices (a fixed function of P): 384 workgroups of 1024 dependent loads each ran at 0.95 TB/s on stage 0.
// The slices are added by m_grn_scale_kernel in ascending order: deterministic.
constexpr int GRN_MAXZ = 16;
__global__ void __launch_bounds__(256) m_grn_sumsq_kernel(const float* __restrict__ h, float* __restrict__ sumsq, int P, int C)
{
    __shared__ float4 red[16][16];
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4, n = blockIdx.y, nz = gridDim.z;
    const int per = (P + nz - 1) / nz, p0 = blockIdx.z * per, p1 = min(P, p0 + per);
    const float* base = h + (long)n * P * C + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = p0 + pl;
    for (; p + 48 < p1; p += 64) {
        const float4 a0 = *(const float4*)(base + (long)p * C), a1 = *(const float4*)(base + (long)(p + 16) * C);
        const float4 a2 = *(const float4*)(base + (long)(p + 32) * C), a3 = *(const float4*)(base + (long)(p + 48) * C);
        s.x = fmaf(a0.x, a0.x, s.x); s.y = fmaf(a0.y, a0.y, s.y); s.z = fmaf(a0.z, a0.z, s.z); s.w = fmaf(a0.w, a0.w, s.w);
        s.x = fmaf(a1.x, a1.x, s.x); s.y = fmaf(a1.y, a1.y, s.y); s.z = fmaf(a1.z, a1.z, s.z); s.w = fmaf(a1.w, a1.w, s.w);
        s.x = fmaf(a2.x, a2.x, s.x); s.y = fmaf(a2.y, a2.y, s.y); s.z = fmaf(a2.z, a2.z, s.z); s.w = fmaf(a2.w, a2.w, s.w);
        s.x = fmaf(a3.x, a3.x, s.x); s.y = fmaf(a3.y, a3.y, s.y); s.z = fmaf(a3.z, a3.z, s.z); s.w = fmaf(a3.w, a3.w, s.w);
    }
    for (; p < p1; p += 16) {
        const float4 a0 = *(const float4*)(base + (long)p * C);
        s.x = fmaf(a0.x, a0.x, s.x); s.y = fmaf(a0.y, a0.y, s.y); s.z = fmaf(a0.z, a0.z, s.z); s.w = fmaf(a0.w, a0.w, s.w);
    }
    red[pl][cq] = s;
    __syncthreads();
    if (pl == 0) {
        float4 t = red[0][cq];
#pragma unroll
        for (int i = 1; i < 16; ++i) { const float4 q = red[i][cq]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
        *(float4*)(sumsq + ((long)n * nz + blockIdx.z) * C + c) = t;
    }
}

// scale[n][c] = 1 + gamma[c] * Gx / (mean_c(Gx) + 1e-6), Gx = sqrt(sumsq)  (util.py:366-368: gamma*(x*Nx) + beta + x)
__global__ void __launch_bounds__(256) m_grn_scale_kernel(const float* __restrict__ sumsq, const float* __restrict__ gamma,
                                                          float* __restrict__ scale, int C, int nz)
{
    __shared__ float red[256];
    __shared__ float gx[3072];
    const int n = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float q = 0.f;
        for (int z = 0; z < nz; ++z) q += sumsq[((long)n * nz + z) * C + c];
        const float r = sqrtf(q);
        gx[c] = r;
        s += r;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float inv = 1.f / (red[0] / (float)C + 1e-6f);
    for (int c = threadIdx.x; c < C; c += 256) scale[(long)n * C + c] = 1.f + gamma[c] * gx[c] * inv;
}

// out[pos] = split(h * scale[n][c] + beta[c]): fp32 [N][P][C] -> split fp16 [N][P][2C], 4 channels per thread
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
    half_t* o = out + pos * 2 * C + c;
    *(h4_t*)o = hi; *(h4_t*)(o + C) = lo;
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
    if (WI != 256 || HI % 4) { cs_set_error("m_stem: 256 input columns (got %d x %d)", HI, WI); return -1; }
    hipLaunchKernelGGL(m_stem_kernel, dim3((unsigned)((long)N * (HI / 4))), dim3(256), 0, st, img, w, b, g, be, x, N, HI, WI);
    M_LAUNCH_CHECK("m_stem");
    return 0;
}

int launch_m_dwln(const float* x, const float* wt, const float* b, const float* g, const float* be, half_t* y, int N, int H, int W, int C, hipStream_t st)
{
    if (W % DWP || H % 4) { cs_set_error("m_dwln: width %d is not a multiple of %d, or height %d not of 4", W, DWP, H); return -1; }
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
    if (C % 64 || C > 3072) { cs_set_error("m_grn: unsupported C=%d", C); return -1; }
    const int nz = P >= 4096 ? 16 : P >= 1024 ? 8 : P >= 256 ? 4 : 1;      // position slices per sample: a function of P only (sumsq holds N x GRN_MAXZ x C floats)
    hipLaunchKernelGGL(m_grn_sumsq_kernel, dim3(C / 64, N, nz), dim3(256), 0, st, h, sumsq, P, C);
    M_LAUNCH_CHECK("m_grn_sumsq");
    hipLaunchKernelGGL(m_grn_scale_kernel, dim3(N), dim3(256), 0, st, sumsq, gamma, scale, C, nz);
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
