// Shared declarations for the CanonSwap gfx950 engine (kernels + host orchestration).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_SIGMOID = 3, ACT_GELU = 4 };   // GELU: exact erf form (nn.GELU default)
enum { MODE_STD = 0, MODE_TBLEND = 1, MODE_SPADE = 2, MODE_PIXSHUF = 3,
       MODE_STDSTAT = 4 };   // MODE_STD + per-tile channel statistics of the stored output (selected by the launchers when p.stat_out)
// tile configurations of the test-only cross-check kernel tests/csrc/conv_igemm.hip (pixels x channels per 256-thread workgroup)
enum { CFG_128x128 = 0, CFG_128x64 = 1, CFG_256x32 = 2, CFG_256x16 = 3 };
// tile configurations of conv_halo
enum { CFG_H_128x128 = 10, CFG_H_128x64 = 11, CFG_H_256x32 = 12, CFG_H_128x32 = 13, CFG_H_128x16 = 14, CFG_H_256x16 = 15,
       CFG_H_SK128x32 = 16 /* 4 waves split the K-steps, reduce through LDS */, CFG_H_128x256 = 17,
       CFG_H_128x160 = 18 /* 2x2 waves of 64 positions x 80 channels: the kw-split mask conv */,
       CFG_H_256x160 = 19 /* 2x2 waves of 128 positions x 80 channels, one workgroup per CU: half the weight bytes per MFMA of 128x160 */,
       CFG_H_256x64 = 20 /* 2x2 waves of 128 positions x 32 channels: 3x3x3 on 8x8x4 tiles (the 64-channel hourglass block) and 3x3 on 16x16 tiles (the 64-channel convs of G's last up block) */,
       CFG_VOL32 = 30 /* vol32.hip (cs_op_conv: force that kernel) */, CFG_WIDE = 31 /* conv_wide.hip (cs_op_conv: force that kernel) */,
       CFG_LAT = 32 /* conv_lat.hip (cs_op_conv: force that kernel) */ };

// A channels-last tensor view: element strides, channel stride is 1.
struct TDesc {
    void* p;
    long sN, sD, sH, sW;
};

// Parameters of one implicit-GEMM convolution launch ("same" padding, stride 1, 1-3 spatial dims).
// GEMM view: M = N*D*H*W output positions, N = Cout, K = taps * Cin.
struct ConvParams {
    // input activations, fp16, channels-last with arbitrary position strides
    const half_t* in;
    const half_t* zero;  // >= 16 bytes of zeros: source of out-of-range / padded pieces (address select, no data select)
    long in_sN, in_sD, in_sH, in_sW;
    int N, D, H, W;     // output extents (the input is addressed at (h>>up_shift, w>>up_shift))
    int inD;            // input depth extent (== D except for the depth-collapsing occlusion conv)
    int Cin;            // valid input channels (multiple of 8)
    // grouped input channels (conv_halo, CK=32 only; 0 = off): channel chunk j lives at in + (j / cg)*in_sG + (j % cg)*32 and
    // only the first cg_cin channels of a group exist. Used by the occlusion conv, whose "channels" are (depth, c).
    int cg, cg_cin;
    long in_sG;
    int nchunks;        // ceil(Cin / 32)
    int up_shift;       // nearest-neighbour up-sampling of the input folded into addressing
    int KD, KH, KW, PD, PH, PW;
    // packed weights [kstep][Cout_pad][32] fp16, kstep = ((chunk*KD+kd)*KH+kh)*KW+kw
    const half_t* wgt;
    // per-sample weight sets (T's identity-modulated convs with several identities in one batch, adaptive_modulate.py:157-167
    // groups=N): sample n uses the set at wgt + wofs[wslot[n]] (element offsets relative to wgt: plain pointer arithmetic on the
    // kernel argument keeps the weight stream global_load; a pointer fetched from a table would make it flat_load, whose completion
    // needs vmcnt(0) + lgkmcnt(0) and drains the prefetch ring at every K-step).  Tiles must lie within one sample.  nullptr: off.
    const long* wofs;
    const int* wslot;
    int Cout_pad;       // packed rows (multiple of the channel tile)
    int Cout;           // logical channels stored (multiple of 4)
    // M-tile decomposition: tile = TN x TD x TH x TW positions (all powers of two)
    int lgTW, lgTH, lgTD;
    int nTW, nTH, nTD, nTN;
    // ---- epilogue
    const float* bias;      // [Cout] (BatchNorm / spectral norm already folded into wgt, bias)
    const float* bias2;     // SPADE: beta bias
    int act0;
    float slope0;
    TDesc res;              // residual (STD/TBLEND) or the tensor being modulated (SPADE)
    int res_f32;
    int res_shift;          // SPADE: x lives at (h>>res_shift, w>>res_shift)
    const float* pixscale;  // per output position scalar: occlusion (STD) or blend mask (TBLEND)
    int ps_stride;
    TDesc out0;
    int out0_f32;
    const float* s2;        // second output: act1(v * s2[c] + t2[c]) as fp16 (next layer's pre-activation)
    const float* t2;
    int act1;
    float slope1;
    TDesc out1;
    const float* stats;     // SPADE: [N][C][2] = (mean, 1/sqrt(var+eps)) of x over its H*W
    // optional: per-(sample, tile, channel) partial (sum, sum of squares) of the stored out0 values, for the next layer's
    // Instance/GroupNorm: stat_out[((n*nblk + blk)*Cout + c)*2 + {0,1}], nblk = tiles per sample * waves along positions.
    // Finished by launch_chan_stats_finish (fixed order: deterministic). Requires tiles that lie within one sample.
    float* stat_out;
    // cross-workgroup split-K for launches with too few tiles to fill 256 CUs (single-frame latency): blockIdx.z = split; every
    // split writes fp32 partial sums [split][N*D*H*W][Cout_pad] instead of running the epilogue; launch_splitk_finish adds them in
    // split order (deterministic) and applies bias / activation / output conversion.  nullptr: off.
    float* sk_out;
    int sk_splits;
    // conv_halo, Cin % 32 == 16: the last 32-channel chunk holds 16 real channels; its weight rows were re-packed by
    // launch_pair_ragged so that two taps that are neighbours along the row share one 32-deep K-step (conv_halo_kernel.h, RAG)
    int ragged;
    // conv_halo: multiply-high constants for workgroup index / {channel blocks, nTW, nTH, nTD} (set by the launcher; 0: divisor 1)
    unsigned mg_ncb, mg_tw, mg_th, mg_td;
    // split-precision convs (activations [hi | lo], weights [W_hi | W_lo | W_hi] in three 32-channel chunks): weight chunks 0 and 1 both
    // multiply the activation chunk 0 (hi), chunk 2 multiplies activation chunk 1 (lo) - the hi halo is staged once.  0: off.
    int hilo;
    // workgroup -> (position tile, channel block) mapping (conv_halo): hardware places workgroup b on XCD b % 8 (each XCD has its
    // own L2).  0: blockIdx.x = tile, blockIdx.y = channel block.  1: the same grid, but every XCD walks a contiguous range of
    // tiles (halo overlaps of neighbouring tiles hit in that XCD's L2).  2: flat grid, contiguous range per XCD with the channel
    // blocks of one tile adjacent (the input tile is fetched into the L2 once for all of them).
    int xcd_map;
    // vol32 only - transform staging (xf_kind != 0): the conv's input is not read from `in` but computed while it is staged, from fp32
    // volumes [N][H][W][16][32] (strides of xf_y / xf_res / xf_out: sN, sH, sW; a column is contiguous):
    //   kind 1: a = xf_y                                                      (replaces the stand-alone split16 pass)
    //   kind 2: a = lrelu((xf_y - mean) * rstd * gamma + beta [+ xf_res])      (GroupNorm(32,32) apply of util.py:531-540 fused into the consumer conv;
    //           mean / rstd per (sample, channel) from xf_stats [N][32][2]; with xf_out the interior of a is also written back: the new
    //           residual stream)
    // and split into [hi | lo] fp16 on the way into LDS (the conv must be a split-precision one: hilo).
    // conv_halo, the kw-split mask conv only (7x7x1 taps, 7 x 22 = 154 output channels (kw, c), 2 x 8 x 16 tiles): instead of storing the 154
    // partials of every position (out0), the workgroup adds, through LDS, the partials its two columns contribute to the same output column
    // and stores 8 logit vectors per tile row: kw_out[((n * D + d) * H + h) * (W / 2) + tile][j][22], j <-> output column w0 - 3 + j
    // (dense_motion.py:88: logit(w) = sum_kw part[w + kw - 3][kw]).  45 % fewer bytes on both sides of the hand-over to the softmax.
    float* kw_out;
    // tests (cs_conv_desc::ep_general): run the general epilogue where the kernel also carries branch-free copies (conv_epilogue.h)
    int ep_general;
    // conv_halo, 3-D tiles of 8 or 4 columns, fp16 out0 only: out0 = AvgPool(1,2,2) of act0(conv + bias) computed in the epilogue (DownBlock3d,
    // util.py:185-190); out0's strides address the POOLED grid (h / 2, w / 2).  The average is taken over the fp32 values: one rounding.
    int pool_hw;
    // mode STD only: out0 = act0(IN(res) * (1 + conv + bias)), IN(res)[n][c] = (res - mean) * rstd with (mean, rstd) = stats[n][c][2] - SPADE's
    // modulation (util.py:295-302) without its beta half; res fp16 (res_shift as in SPADE), tiles within one sample
    int spmul;
    // spmul launches whose 256 packed channels are ALL channels of the modulated tensor (the 128 x 256 tile kernel, G's up_1 shortcut): the
    // 1x1 conv_s that consumes h = IN(res)(1 + conv) runs inside the epilogue - xs_out (fp16, xs_cout <= 64 channels) = xs_w h + xs_res; h is
    // not stored (out0 == nullptr).  xs_w: packed [Cout / 32][64][32] (pack_conv of the 1x1 weight).  conv_halo_kernel.h, XSK.
    const half_t* xs_w; TDesc xs_res, xs_out; int xs_cout;
    // conv_halo's 256 x 160 tiles: != 0 asks for the persistent launch (one workgroup per CU walks a list of tiles, the next tile's first
    // chunk staged under the last chunk of the current one; conv_halo_kernel.h); the launcher replaces it by the number of (tile, channel
    // block) entries, or by 0 where the kernel has no such mode.
    int persist_total;
    unsigned in_sample_bytes;   // conv_halo's 256 x 160 tiles (set by the launcher): bytes one sample of the input spans, the range of their buffer-addressed halo DMA
    // conv_halo: up to four convolutions that differ in their weights, their leading padding and an offset into out0 only - the output phases of an
    // up-sampling conv on the source grid (pack.upsampled_conv3d_phases) - as ONE launch, blockIdx.z = phase.  0: off.  Not with split-K.
    int nphase;
    long ph_wofs[4];          // element offset of phase z's packed weights from wgt
    unsigned ph_ooff[4];      // element offset of phase z's outputs inside out0
    int ph_PH[4], ph_PW[4];   // PH / PW of phase z
    int xf_kind;
    // vol32 statistics launches: rows of a partial-statistics block (0: the default 8).  Latency mode passes 2: a one-frame launch then cuts its
    // strips into 2-row segments (256 items instead of 64 on a 64 x 64 volume: every CU gets one); another grouping of the partial sums, i.e.
    // other last bits of (mean, rstd) than the batched path - like split-K, only behind cs_set_latency_mode
    int v32_srows;
    TDesc xf_y, xf_res, xf_out;
    const float* xf_stats; const float* xf_gamma; const float* xf_beta;
    float xf_slope;
#ifdef CS_TIMELINE
    // instrumented builds only (tools/timeline.py): per-wave s_memtime stamps of the kernel's phases, 12 x u64 per wave
    unsigned long long* tl;
    long tl_cap;            // capacity in waves
#endif
};

// GroupNorm apply + residual + LeakyReLU as ONE fixed sequence of operations (explicit fma: no contraction choices), shared by
// norm_act_kernel (kernels.hip) and the transform staging of vol32.hip so that both give the same bits:
//   sc = rstd * gamma, sh = beta - mean * sc, a = v * sc + sh + res, lrelu
__device__ __forceinline__ float gn_scale(float rstd, float gamma) { return rstd * gamma; }
__device__ __forceinline__ float gn_shift(float mean, float sc, float beta) { return fmaf(-mean, sc, beta); }
__device__ __forceinline__ float gn_lrelu(float v, float sc, float sh, float rr, float slope)
{
    float a = fmaf(v, sc, sh);
    a = a + rr;
    return fmaxf(a, a * slope);      // LeakyReLU for 0 <= slope <= 1 (the callers' 0.01): two instructions instead of compare + multiply + select
}

#define CS_CHECK_HIP(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            cs_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -1;                                                                      \
        }                                                                                   \
    } while (0)

void cs_set_error(const char* fmt, ...);

// ---- kernel launchers (conv_halo.hip, kernels.hip); all asynchronous on `st`
int launch_conv(const ConvParams& p, int cfg, int mode, hipStream_t st);   // tests/csrc/conv_igemm.hip: test-only library
int launch_conv_halo(const ConvParams& p, int cfg, int ck, int mode, hipStream_t st);
// vol32.hip: the 3x3x3 32 -> 32 convolutions on [N][H][W][16][32] volumes (weights in registers, persistent workgroups marching along H)
// vol32_fused.hip: one whole ResBlock3d (util.py:80-102) of the feature volume as one launch: out0 = conv2(relu(conv1(a) + b1)) + b2 + x,
// out1 = act1(out0 * s2 + t2); BatchNorms folded into the weights / biases / (s2, t2) at load time.  All four volumes are
// [N][H][W][16][32] with the same element strides (sN, sH, sW; a column of 16 voxels x 32 channels contiguous); a, out1 fp16; x, out0 fp32.
struct ResBlock3dCall {
    const half_t* a; const float* x; float* out0; half_t* out1;
    long sN, sH, sW;
    const half_t* w1; const half_t* w2;          // packed [27][32][32]
    const float* b1; const float* b2; const float* s2; const float* t2;
    int act1; float slope1;
    int N, H, W;
};
bool vol32_fused_supported(const ResBlock3dCall& c);
int launch_vol32_fused(const ResBlock3dCall& c, hipStream_t st);
bool vol32_supported(const ConvParams& p);
int vol32_stat_nblk(const ConvParams& p);        // partial-statistics blocks per sample when ConvParams::stat_out is set
int launch_vol32(const ConvParams& p, hipStream_t st);
// conv_wide.hip: persistent 3x3 kernel for the wide 2-D layers (256 x 256 workgroup tiles, 8 x 8 fragments per wave, one workgroup per CU)
bool conv_wide_supported(const ConvParams& p, int mode);
int launch_conv_wide(const ConvParams& p, int mode, hipStream_t st);
// conv_lat.hip: single-frame kernel for the 512-channel 3x3 layers at 64 x 64 (16 x 8 tiles, the K loop split over the three kernel rows across 12 waves);
// another summation order than conv_halo / conv_wide: latency mode only
bool conv_lat_supported(const ConvParams& p, int mode);
int launch_conv_lat(const ConvParams& p, int mode, hipStream_t st);
const half_t* cs_zero_page();   // per-process device buffer of zeros (lazily allocated on the current device)

int launch_conv_first(const float* img, const float* w, const float* b, half_t* out, int N, int H, int W, hipStream_t st);
int launch_avgpool(const half_t* in, int N, int D, int H, int W, int C, TDesc out, hipStream_t st);
int launch_dm_compress(const float* f, const float* w, const float* b, half_t* comp, int N, int D, int H, int W, hipStream_t st);
// shared_*: one compressed / feature volume, one kp_s set for all N samples (sample stride 0: the v2i body warps ONE swapped canonical volume)
int launch_dm_sparse(const half_t* comp, const float* kp_d, const float* kp_s, half_t* out, int out_stride, int N, int D,
                     int H, int W, hipStream_t st, bool shared_comp = false, bool shared_kps = false);
int launch_dm_softmax(const float* part, const float* bias, const float* kp_d, const float* kp_s, float* deform, float* mask_out,
                      int N, int D, int H, int W, hipStream_t st, int compact = 0, bool shared_kps = false);
int launch_dm_softmax_warp(const float* part, const float* bias, const float* kp_d, const float* kp_s, const float* in, float* out32,
                           half_t* out16, float* deform, int N, int D, int H, int W, hipStream_t st, int compact = 0, bool shared_in = false,
                           bool shared_kps = false);
int launch_occ_finish(const float* part, float bias, float* occ, int N, int H, int W, hipStream_t st);
int launch_occ_finish49(const float* part, float bias, float* occ, int N, int H, int W, hipStream_t st);
int launch_grid_sample(const float* in, const float* grid, float* out32, half_t* out16, int N, int D, int H, int W, hipStream_t st,
                       bool shared_in = false);
int launch_chan_stats(const void* x, int is_f32, int N, long P, int C, float eps, float* partials, float* stats, hipStream_t st);
long chan_stats_partial_floats(int N, long P, int C);
int launch_chan_stats_finish(const float* partials, int nblk, int N, int C, double cnt_inv, float eps, float* stats, hipStream_t st);
int launch_norm_act(const float* y, const float* stats, const float* gamma, const float* beta,
                    const float* res, float slope, float* out32, half_t* out16, const float* s2, const float* t2, int period2,
                    int act2, float slope2, int N, long per_n, hipStream_t st, int split = 0);
int launch_split16(const float* x, half_t* out, long n, hipStream_t st);
int launch_splitk_finish(const ConvParams& p, hipStream_t st);
// re-pack the last chunk of a packed conv weight [chunk * taps][Cout_pad][32] for ConvParams::ragged (in place, via a scratch copy)
int launch_pair_ragged(half_t* w, int Cout_pad, int nchunks, int KD, int KH, int KW, hipStream_t st);
int launch_absmax16(const half_t* x, TDesc t, int N, int D, int H, int W, int C, unsigned* slot, hipStream_t st);
int launch_ncdhw_to_hwdc(const float* in, float* out32, half_t* out16, const float* s2, const float* t2, int act2, float slope2,
                         int N, int C, int D, int H, int W, hipStream_t st);
int launch_hwdc_to_ncdhw(const float* in, float* out, int N, int C, int D, int H, int W, hipStream_t st);
int launch_nchw_to_nhwc16(const float* in, half_t* out, int N, int C, int HW, hipStream_t st);
int launch_nhwc16_to_nchw(const half_t* in, float* out, int N, int C, int HW, hipStream_t st);
int launch_t_mask(const half_t* x, const half_t* wpacked, const float* bias, float* tmask, int N, int H, int W, hipStream_t st);
int launch_t_style(const float* id, const float* fc, float* style, int nlayers, hipStream_t st);
int launch_t_modulate(const float* wraw, const float* style, half_t* packed, int layer, hipStream_t st);
int launch_pack_u8(const float* img, uint8_t* out, int N, int C, int H, int W, hipStream_t st);
int launch_lrelu16(const half_t* in, half_t* out, long n, float slope, hipStream_t st);

int launch_unpack_u8(const uint8_t* in, float* out, int N, int C, int H, int W, hipStream_t st);

// ---- image-space steps around the generator (imgops.hip)
int launch_soft_erosion(const void* mask, int mask_u8, float* tmp_a, float* tmp_b, const float* w, float* part, float* soft, unsigned char* hard,
                        int B, int H, int W, int ksize, float thr, int iters, int per_sample, hipStream_t st);
int launch_paste_batch(const unsigned char* crops, const float* masks, int Hc, int Wc, const double* M, const unsigned char* oris,
                       unsigned char* outs, int B, int Ho, int Wo, hipStream_t st);
int launch_prepare_crops(const unsigned char* in, float* out, int B, int Hc, int Wc, int factor, hipStream_t st);
int launch_paste(const unsigned char* crop, const float* mask_crop, const float* mask_ori, int Hc, int Wc, const double M[6],
                 const unsigned char* ori, unsigned char* out, int Ho, int Wo, hipStream_t st);
int launch_warp_f32(const float* src, int Hs, int Ws, const double M[6], float* dst, int Hd, int Wd, hipStream_t st);

// ---- motion extractor pieces (motion.hip)
int launch_m_keypoints(const float* raw, float* x_t, float* x_can, float* rot, int N, hipStream_t st);
int launch_m_stem(const float* img, const float* w, const float* b, const float* g, const float* be, float* x, int N, int HI, int WI, hipStream_t st);
int launch_m_dwln(const float* x, const float* wt, const float* b, const float* g, const float* be, half_t* y, int N, int H, int W, int C, hipStream_t st);
int launch_m_ln_s2d(const float* x, const float* g, const float* be, half_t* y, int N, int H, int W, int C, hipStream_t st);
int launch_m_grn(const float* h, const float* gamma, const float* beta, float* sumsq, float* scale, half_t* out, int N, int P, int C, hipStream_t st);
int launch_m_head(const float* x, const float* g, const float* be, const float* hw, const float* hb, float* out, int N, int P, hipStream_t st);
