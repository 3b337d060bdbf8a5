// Image-space steps on either side of the generator, on the device:
//   * SoftErosion of the face-parsing mask (src/utils/crop.py:21-47),
//   * paste-back of the generated crop into the original frame: cv2.warpAffine of crop and mask + blend
//     (src/utils/crop.py:49-63, 515-529, driven from can_swap_pipeline_e2e.py:267-283),
//   * input staging: cv2.resize(crop, (256,256), INTER_AREA) + /255 + HWC->CHW (src/utils/cropper.py:209, can_swap_e2e.py:126-163).
// All of them are HBM-bound byte / float work: one thread per output pixel, coalesced along the row.  The OpenCV steps follow
// OpenCV's published fixed-point algorithm (restated in oracle/cv_ref.py, which the tests compare against bit for bit).
#include "common.h"

// Bit-exact restatements of CPU arithmetic (numpy / OpenCV evaluate a * b + c with two roundings): no FMA contraction in this file.
// Plain operators on purpose: the __fmul_rn / __fadd_rn intrinsics are header functions compiled with contraction allowed, and
// once inlined their multiply and add fuse again.  The fp32 convolution of SoftErosion asks for fmaf explicitly.
#pragma clang fp contract(off)

#define LAUNCH_CHECK(name)                                                                       \
    do {                                                                                         \
        hipError_t _e = hipGetLastError();                                                       \
        if (_e != hipSuccess) { cs_set_error(name " launch: %s", hipGetErrorString(_e)); return -1; } \
    } while (0)

// ---------------------------------------------------------------------------------------------- SoftErosion
// one pass: out = conv2d(in, w, padding = r) [zero padding], optionally out = min(in, out)  (crop.py:38-41)
// Round 6: register tiled.  A workgroup owns 64 x 32 outputs, a thread 8 consecutive outputs of one row: per kernel row it reads the
// 8 + KS - 1 inputs it needs ONCE from LDS (seven ds_read_b128; the lane -> address map (row stride 84 floats, 32 B between lanes) is
// conflict free for that instruction: the 16 lanes of a group cover the 64 banks) and issues KS x 8 FMAs on them, the weights are
// wave-uniform scalar loads, the FMAs packed two per instruction (v_pk_fma_f32).  One LDS read per six FMAs instead of two per FMA: the pass is VALU-bound (0.69 GFLOP per frame and
// three passes, crop.py:29-35 - the kernel is a cone, not separable).  An output's products are added in the order (ky, kx) ascending.
// IN: float (0/1 masks as the reference's .float() makes them, or the previous pass) or uint8 (0/1 labels straight from the parser).
template <int KS, typename IN>
__global__ void __launch_bounds__(256) se_conv_kernel(const IN* __restrict__ in, const float* __restrict__ w, float* __restrict__ out,
                                                      int H, int W, int take_min)
{
    constexpr int R = KS / 2, NX = 8, TW = 64, TH = 32, LW = 84, LH = TH + 2 * R, NV = (NX + 2 * R + 3) / 4;
    static_assert(TW + 2 * R <= LW && LW % 8 == 4, "row stride: 16 B aligned and an odd multiple of 16 B");
    __shared__ __attribute__((aligned(16))) float tile[LH * LW + 4];
    const int n = blockIdx.z, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const IN* src = in + (long)n * H * W;
    {   // staging: thread -> (column lx, rows ly0, ly0 + 3, ...): one division per thread, not per element (252 of the 256 threads; LW = 84)
        const int lx = threadIdx.x % LW, ly0 = threadIdx.x / LW, x = x0 + lx - R;
        const bool xin = (unsigned)x < (unsigned)W;
        if (ly0 < 3) {
#pragma unroll 6
            for (int ly = ly0; ly < LH; ly += 3) {
                const int y = y0 + ly - R;
                tile[ly * LW + lx] = (xin && (unsigned)y < (unsigned)H) ? (float)src[(long)y * W + x] : 0.f;
            }
        }
    }
    __syncthreads();
    const int g = threadIdx.x & 7, r = threadIdx.x >> 3;
    // packed fp32 FMAs (v_pk_fma_f32: two lanes' worth of FMA per instruction slot): outputs in pairs (2p, 2p + 1); an even tap reads the
    // aligned input pairs E[j] = (v[2j], v[2j + 1]), an odd tap the shifted pairs O[j] = (v[2j + 1], v[2j + 2]) (one v_pk_mov_b32 each per
    // kernel row).  84 packed FMAs + 13 moves per kernel row instead of 168 FMAs; every output still adds its products in (ky, kx) order.
    typedef float f2_t __attribute__((ext_vector_type(2)));
    constexpr int NP = NX / 2, NE = NV * 2;
    f2_t acc2[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc2[p] = (f2_t){0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < KS; ++ky) {
        const float4* rp = (const float4*)(tile + (r + ky) * LW + g * NX);
        f2_t E[NE], O[NE - 1];
#pragma unroll
        for (int j = 0; j < NV; ++j) { const float4 q = rp[j]; E[2 * j] = (f2_t){q.x, q.y}; E[2 * j + 1] = (f2_t){q.z, q.w}; }
#pragma unroll
        for (int j = 0; j < NE - 1; ++j) O[j] = (f2_t){E[j].y, E[j + 1].x};
        const float* wr = w + ky * KS;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const float wv = wr[kx];
            const f2_t w2 = (f2_t){wv, wv};
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const f2_t in2 = (kx & 1) ? O[p + (kx - 1) / 2] : E[p + kx / 2];
                acc2[p] = __builtin_elementwise_fma(in2, w2, acc2[p]);
            }
        }
    }
    float acc[NX];
#pragma unroll
    for (int p = 0; p < NP; ++p) { acc[2 * p] = acc2[p].x; acc[2 * p + 1] = acc2[p].y; }
    const int y = y0 + r, xb = x0 + g * NX;
    if (y >= H) return;
    float res[NX];
#pragma unroll
    for (int o = 0; o < NX; ++o) {
        const float c = tile[(r + R) * LW + g * NX + R + o];
        res[o] = take_min ? fminf(c, acc[o]) : acc[o];
    }
    float* dst = out + (long)n * H * W + (long)y * W + xb;
    if (xb + NX <= W && (W & 3) == 0) {
        ((float4*)dst)[0] = make_float4(res[0], res[1], res[2], res[3]);
        ((float4*)dst)[1] = make_float4(res[4], res[5], res[6], res[7]);
    } else {
#pragma unroll
        for (int o = 0; o < NX; ++o) if (xb + o < W) dst[o] = res[o];
    }
}

// max over the pixels below the threshold (crop.py:45: `x[~mask].max()` runs over the WHOLE (N,1,H,W) tensor, not per sample -
// ADVICE r2); two deterministic stages (max is order independent): per-(sample, block) partials, then every workgroup of the final
// pass reduces all N x nparts of them once
__global__ void __launch_bounds__(256) se_max_kernel(const float* __restrict__ x, long P, float thr, float* __restrict__ part)
{
    const int n = blockIdx.y;
    float m = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long)gridDim.x * 256) {
        const float v = x[(long)n * P + i];
        if (!(v >= thr)) m = fmaxf(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[n * gridDim.x + blockIdx.x] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
}

__global__ void __launch_bounds__(256) se_final_kernel(float* __restrict__ x, long P, float thr, const float* __restrict__ part, int nparts,
                                                       unsigned char* __restrict__ hard, int per_sample)
{
    const int n = blockIdx.y;
    if (per_sample) part += (long)n * nparts;      // B independent calls of the module (the pipeline's loop): the maximum of this sample only
    // nparts = blocks x samples: the maximum over the whole batch.  One cooperative pass per workgroup (threads stride over the partials,
    // wave butterfly, four wave results through LDS) instead of every thread looping over all of them (4096 broadcast loads per thread at
    // 64 frames: ADVICE r3); max is order independent, so the value is the same.
    __shared__ float sm[4];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, part[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float v = x[(long)n * P + i];
    const bool hi = v >= thr;
    x[(long)n * P + i] = hi ? 1.f : v / m;          // x[mask] = 1; x[~mask] /= x[~mask].max()
    if (hard) hard[(long)n * P + i] = hi ? 1 : 0;
}

int launch_soft_erosion(const void* mask, int mask_u8, float* tmp_a, float* tmp_b, const float* w, float* part, float* soft, unsigned char* hard,
                        int B, int H, int W, int ksize, float thr, int iters, int per_sample, hipStream_t st)
{
    if (ksize != 21 && ksize != 15) { cs_set_error("soft_erosion: kernel size %d (21 and 15 are built)", ksize); return -1; }
    if (iters < 1) { cs_set_error("soft_erosion: iterations %d", iters); return -1; }
    const dim3 grid((W + 63) / 64, (H + 31) / 32, B);
    const float* src = nullptr;
    for (int it = 0; it < iters; ++it) {
        const bool last = it == iters - 1;
        float* dst = last ? soft : (it % 2 == 0 ? tmp_a : tmp_b);
        const int tm = last ? 0 : 1;
        if (it == 0 && mask_u8) {
            if (ksize == 21) hipLaunchKernelGGL((se_conv_kernel<21, unsigned char>), grid, dim3(256), 0, st, (const unsigned char*)mask, w, dst, H, W, tm);
            else hipLaunchKernelGGL((se_conv_kernel<15, unsigned char>), grid, dim3(256), 0, st, (const unsigned char*)mask, w, dst, H, W, tm);
        } else {
            const float* s_ = it == 0 ? (const float*)mask : src;
            if (ksize == 21) hipLaunchKernelGGL((se_conv_kernel<21, float>), grid, dim3(256), 0, st, s_, w, dst, H, W, tm);
            else hipLaunchKernelGGL((se_conv_kernel<15, float>), grid, dim3(256), 0, st, s_, w, dst, H, W, tm);
        }
        LAUNCH_CHECK("se_conv");
        src = dst;
    }
    const long P = (long)H * W;
    const int nparts = 64;
    hipLaunchKernelGGL(se_max_kernel, dim3(nparts, B), dim3(256), 0, st, soft, P, thr, part);
    LAUNCH_CHECK("se_max");
    hipLaunchKernelGGL(se_final_kernel, dim3((unsigned)((P + 255) / 256), B), dim3(256), 0, st, soft, P, thr, part, per_sample ? nparts : nparts * B, hard,
                       per_sample);
    LAUNCH_CHECK("se_final");
    return 0;
}

// ---------------------------------------------------------------------------------------------- input staging
// cv2.resize(INTER_AREA) by exactly 2 in both directions for 8-bit images: (a + b + c + d + 2) >> 2; then / 255, clip, CHW
__global__ void __launch_bounds__(256) prepare_crops_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, int B, int Hd, int Wd,
                                                            int factor)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long P = (long)Hd * Wd;
    if (i >= (long)B * P) return;
    const int n = (int)(i / P), y = (int)((i % P) / Wd), x = (int)(i % Wd);
    const int Ws = Wd * factor;
    const unsigned char* s = in + ((long)n * Hd * factor + (long)y * factor) * Ws * 3 + (long)x * factor * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v;
        if (factor == 2) v = (s[c] + s[3 + c] + s[(long)Ws * 3 + c] + s[(long)Ws * 3 + 3 + c] + 2) >> 2;
        else v = s[c];
        out[((long)n * 3 + c) * P + (long)y * Wd + x] = fminf(fmaxf((float)v / 255.f, 0.f), 1.f);
    }
}

int launch_prepare_crops(const unsigned char* in, float* out, int B, int Hc, int Wc, int factor, hipStream_t st)
{
    const long n = (long)B * (Hc / factor) * (Wc / factor);
    hipLaunchKernelGGL(prepare_crops_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, B, Hc / factor, Wc / factor, factor);
    LAUNCH_CHECK("prepare_crops");
    return 0;
}

// ---------------------------------------------------------------------------------------------- cv2.warpAffine, INTER_LINEAR
struct AffineInv { double m[6]; };      // destination -> source map (what warpAffine derives from M)

// OpenCV's fixed-point source coordinate of destination pixel (x, y): AB_BITS = 10, INTER_BITS = 5 (imgwarp.cpp warpAffine)
__device__ __forceinline__ void affine_coords(const AffineInv& A, int x, int y, int& sx, int& sy, int& fx, int& fy)
{
    const double AB = 1024.0;
    const long adelta = (long)rint(A.m[0] * (double)x * AB);
    const long bdelta = (long)rint(A.m[3] * (double)x * AB);
    const long X0 = (long)rint((A.m[1] * (double)y + A.m[2]) * AB) + 16;     // AB_SCALE / INTER_TAB_SIZE / 2
    const long Y0 = (long)rint((A.m[4] * (double)y + A.m[5]) * AB) + 16;
    const long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    long ix = X >> 5, iy = Y >> 5;
    ix = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);
    iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
    sx = (int)ix; sy = (int)iy; fx = (int)(X & 31); fy = (int)(Y & 31);
}

// crop (u8, 3 channels) warped into the destination frame; with a mask (warped from the crop frame, or given in the
// destination frame) and the original image: out = clip(mask * warped + (1 - mask) * ori, 0, 255) truncated (crop.py:523-529)
__global__ void __launch_bounds__(256) paste_kernel(const unsigned char* __restrict__ crop, const float* __restrict__ mask_crop,
                                                    const float* __restrict__ mask_ori, int Hc, int Wc, AffineInv A,
                                                    const unsigned char* __restrict__ ori, unsigned char* __restrict__ out, int Ho, int Wo)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)Ho * Wo) return;
    const int y = (int)(i / Wo), x = (int)(i % Wo);
    int sx, sy, fx, fy;
    affine_coords(A, x, y, sx, sy, fx, fy);
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;   // sum 1 << 15
    const bool y0 = (unsigned)sy < (unsigned)Hc, y1 = (unsigned)(sy + 1) < (unsigned)Hc;
    const bool x0 = (unsigned)sx < (unsigned)Wc, x1 = (unsigned)(sx + 1) < (unsigned)Wc;
    int res[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int t00 = (y0 && x0) ? crop[((long)sy * Wc + sx) * 3 + c] : 0;
        const int t01 = (y0 && x1) ? crop[((long)sy * Wc + sx + 1) * 3 + c] : 0;
        const int t10 = (y1 && x0) ? crop[((long)(sy + 1) * Wc + sx) * 3 + c] : 0;
        const int t11 = (y1 && x1) ? crop[((long)(sy + 1) * Wc + sx + 1) * 3 + c] : 0;
        res[c] = (t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11 + (1 << 14)) >> 15;
    }
    if (!ori) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[i * 3 + c] = (unsigned char)res[c];
        return;
    }
    float m;
    if (mask_ori) m = mask_ori[i];
    else {      // float image: table weights (1 - a)(1 - b) ... in float, left-to-right sum without contraction (remapBilinear, float)
        const float ax = (float)fx * 0.03125f, ay = (float)fy * 0.03125f;
        const float f00 = (1.f - ay) * (1.f - ax), f01 = (1.f - ay) * ax;
        const float f10 = ay * (1.f - ax), f11 = ay * ax;
        const float t00 = (y0 && x0) ? mask_crop[(long)sy * Wc + sx] : 0.f, t01 = (y0 && x1) ? mask_crop[(long)sy * Wc + sx + 1] : 0.f;
        const float t10 = (y1 && x0) ? mask_crop[(long)(sy + 1) * Wc + sx] : 0.f, t11 = (y1 && x1) ? mask_crop[(long)(sy + 1) * Wc + sx + 1] : 0.f;
        m = ((t00 * f00 + t01 * f01) + t10 * f10) + t11 * f11;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = m * (float)res[c] + (1.f - m) * (float)ori[i * 3 + c];
        out[i * 3 + c] = (unsigned char)fminf(fmaxf(v, 0.f), 255.f);       // np.clip(...).astype(np.uint8): truncation
    }
}

// ---- the paste-back step of B frames in one launch (can_swap_pipeline_e2e.py:273-283 per frame: SoftErosion mask -> prepare_paste_back ->
// paste_back).  blockIdx.y = frame; a thread owns four consecutive pixels of a row = 12 bytes = three dwords of the original and of the
// result (the single-frame kernel above moves them byte by byte).  A pixel whose source position lies outside the crop has mask 0 and
// warped value 0: v = 0 * 0 + (1 - 0) * ori = ori exactly, so it is copied - most of a 1080p frame.  Same arithmetic per pixel as
// paste_kernel with mask_crop (tests/test_gpu_chain.py holds the two bit-equal).
struct AffineBatch { AffineInv a[64]; };      // 3 KB of kernel arguments: no device-side staging buffer, nothing to race with

__global__ void __launch_bounds__(256) paste_batch_kernel(const unsigned char* __restrict__ crops, const float* __restrict__ masks, int Hc, int Wc,
                                                          AffineBatch AB, const unsigned char* __restrict__ oris,
                                                          unsigned char* __restrict__ outs, int Ho, int Wo)
{
    const int n = blockIdx.y;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;          // group of four pixels
    const long P = (long)Ho * Wo;
    if (q * 4 >= P) return;
    const AffineInv& A = AB.a[n];
    const unsigned char* crop = crops + (long)n * Hc * Wc * 3;
    const float* mask_crop = masks + (long)n * Hc * Wc;
    const unsigned* ori = (const unsigned*)(oris + (long)n * P * 3) + q * 3;
    unsigned* out = (unsigned*)(outs + (long)n * P * 3) + q * 3;
    unsigned wv[3] = {ori[0], ori[1], ori[2]};
    unsigned char* px = (unsigned char*)wv;
    const int y = (int)((q * 4) / Wo), xb = (int)((q * 4) % Wo);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int sx, sy, fx, fy;
        affine_coords(A, xb + k, y, sx, sy, fx, fy);
        if (sx < -1 || sy < -1 || sx >= Wc || sy >= Hc) continue;          // all four taps outside: the original pixel
        const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
        const bool y0 = (unsigned)sy < (unsigned)Hc, y1 = (unsigned)(sy + 1) < (unsigned)Hc;
        const bool x0 = (unsigned)sx < (unsigned)Wc, x1 = (unsigned)(sx + 1) < (unsigned)Wc;
        const long o00 = (long)sy * Wc + sx;
        const float ax = (float)fx * 0.03125f, ay = (float)fy * 0.03125f;
        const float f00 = (1.f - ay) * (1.f - ax), f01 = (1.f - ay) * ax;
        const float f10 = ay * (1.f - ax), f11 = ay * ax;
        const float m00 = (y0 && x0) ? mask_crop[o00] : 0.f, m01 = (y0 && x1) ? mask_crop[o00 + 1] : 0.f;
        const float m10 = (y1 && x0) ? mask_crop[o00 + Wc] : 0.f, m11 = (y1 && x1) ? mask_crop[o00 + Wc + 1] : 0.f;
        const float m = ((m00 * f00 + m01 * f01) + m10 * f10) + m11 * f11;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int t00 = (y0 && x0) ? crop[o00 * 3 + c] : 0;
            const int t01 = (y0 && x1) ? crop[(o00 + 1) * 3 + c] : 0;
            const int t10 = (y1 && x0) ? crop[(o00 + Wc) * 3 + c] : 0;
            const int t11 = (y1 && x1) ? crop[(o00 + Wc + 1) * 3 + c] : 0;
            const int res = (t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11 + (1 << 14)) >> 15;
            const float v = m * (float)res + (1.f - m) * (float)px[k * 3 + c];
            px[k * 3 + c] = (unsigned char)fminf(fmaxf(v, 0.f), 255.f);
        }
    }
    out[0] = wv[0]; out[1] = wv[1]; out[2] = wv[2];
}

// float image (one channel) warped into the destination frame (prepare_paste_back, crop.py:515-521)
__global__ void __launch_bounds__(256) warp_f32_kernel(const float* __restrict__ src, int Hs, int Ws, AffineInv A, float* __restrict__ dst,
                                                       int Hd, int Wd)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)Hd * Wd) return;
    const int y = (int)(i / Wd), x = (int)(i % Wd);
    int sx, sy, fx, fy;
    affine_coords(A, x, y, sx, sy, fx, fy);
    const bool y0 = (unsigned)sy < (unsigned)Hs, y1 = (unsigned)(sy + 1) < (unsigned)Hs;
    const bool x0 = (unsigned)sx < (unsigned)Ws, x1 = (unsigned)(sx + 1) < (unsigned)Ws;
    const float ax = (float)fx * 0.03125f, ay = (float)fy * 0.03125f;
    const float f00 = (1.f - ay) * (1.f - ax), f01 = (1.f - ay) * ax;
    const float f10 = ay * (1.f - ax), f11 = ay * ax;
    const float t00 = (y0 && x0) ? src[(long)sy * Ws + sx] : 0.f, t01 = (y0 && x1) ? src[(long)sy * Ws + sx + 1] : 0.f;
    const float t10 = (y1 && x0) ? src[(long)(sy + 1) * Ws + sx] : 0.f, t11 = (y1 && x1) ? src[(long)(sy + 1) * Ws + sx + 1] : 0.f;
    dst[i] = ((t00 * f00 + t01 * f01) + t10 * f10) + t11 * f11;
}

static AffineInv invert_affine(const double M[6])      // imgwarp.cpp warpAffine, !WARP_INVERSE_MAP
{
    AffineInv A;
    for (int i = 0; i < 6; ++i) A.m[i] = M[i];
    double D = A.m[0] * A.m[4] - A.m[1] * A.m[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = A.m[4] * D, A22 = A.m[0] * D;
    A.m[0] = A11; A.m[1] *= -D; A.m[3] *= -D; A.m[4] = A22;
    const double b1 = -A.m[0] * A.m[2] - A.m[1] * A.m[5];
    const double b2 = -A.m[3] * A.m[2] - A.m[4] * A.m[5];
    A.m[2] = b1; A.m[5] = b2;
    return A;
}

int launch_paste(const unsigned char* crop, const float* mask_crop, const float* mask_ori, int Hc, int Wc, const double M[6],
                 const unsigned char* ori, unsigned char* out, int Ho, int Wo, hipStream_t st)
{
    const long n = (long)Ho * Wo;
    hipLaunchKernelGGL(paste_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, crop, mask_crop, mask_ori, Hc, Wc, invert_affine(M),
                       ori, out, Ho, Wo);
    LAUNCH_CHECK("paste");
    return 0;
}

int launch_warp_f32(const float* src, int Hs, int Ws, const double M[6], float* dst, int Hd, int Wd, hipStream_t st)
{
    const long n = (long)Hd * Wd;
    hipLaunchKernelGGL(warp_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, Hs, Ws, invert_affine(M), dst, Hd, Wd);
    LAUNCH_CHECK("warp_f32");
    return 0;
}

int launch_paste_batch(const unsigned char* crops, const float* masks, int Hc, int Wc, const double* M, const unsigned char* oris,
                       unsigned char* outs, int B, int Ho, int Wo, hipStream_t st)
{
    const long P = (long)Ho * Wo;
    const bool vec = Wo % 4 == 0 && ((uintptr_t)oris & 3) == 0 && ((uintptr_t)outs & 3) == 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int nb = B - b0 < 64 ? B - b0 : 64;
        if (vec) {
            AffineBatch AB;
            for (int i = 0; i < nb; ++i) AB.a[i] = invert_affine(M + (long)(b0 + i) * 6);
            hipLaunchKernelGGL(paste_batch_kernel, dim3((unsigned)((P / 4 + 255) / 256), nb), dim3(256), 0, st, crops + (long)b0 * Hc * Wc * 3,
                               masks + (long)b0 * Hc * Wc, Hc, Wc, AB, oris + (long)b0 * P * 3, outs + (long)b0 * P * 3, Ho, Wo);
            LAUNCH_CHECK("paste_batch");
        } else {      // odd widths / unaligned buffers: the single-frame kernel per frame (same arithmetic)
            for (int i = b0; i < b0 + nb; ++i)
                if (launch_paste(crops + (long)i * Hc * Wc * 3, masks + (long)i * Hc * Wc, nullptr, Hc, Wc, M + (long)i * 6, oris + (long)i * P * 3,
                                 outs + (long)i * P * 3, Ho, Wo, st)) return -1;
        }
    }
    return 0;
}
