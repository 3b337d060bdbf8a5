/* canonswap_hip.h -- C ABI of the MI355X (gfx950) CanonSwap generator engine.
 *
 * The reference has no FFI: its boundary for this path is the Python class `can_swapper`
 * (src/can_swap_e2e.py:39) whose stage methods/attributes the pipeline calls once per frame
 * (src/can_swap_pipeline_e2e.py:242-263).  Every entry point below replaces one of those calls and keeps
 * its tensor signature: all tensors are caller-owned, contiguous fp32 device buffers in the reference's
 * NCHW / NCDHW layouts, passed as raw pointers (`tensor.data_ptr()`), and all work is enqueued on the
 * caller's HIP stream (`torch.cuda.current_stream().cuda_stream`).  The engine owns its packed weights,
 * per-identity modulated weights and workspace; nothing it allocates is returned.
 *
 * Error convention: every function returns 0 on success, non-zero on failure with a message available
 * from cs_last_error() (thread local).  No C++ exception crosses this boundary.
 * Threading: an engine is bound to one device and is not thread-safe.
 */
#ifndef CANONSWAP_HIP_H
#define CANONSWAP_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cs_engine cs_engine;

/* ---- life cycle (replaces can_swapper.__init__ / load_cpk, src/can_swap_e2e.py:44-100) */
int cs_create(int device_id, int max_batch, cs_engine** out);   /* 1 <= max_batch <= 84 (32-bit element offsets inside one tensor); workspace ~0.35 GB per frame of batch; 64 is the fastest launch size (profiles/r05_b_batch_sweep.txt) */
void cs_destroy(cs_engine* e);
const char* cs_last_error(void);
#define CS_ABI_VERSION 4       /* bumped whenever a struct of this header, an entry point's meaning or the weight blob format changes */
int cs_abi_version(void);
/* Upload one packed weight blob (host pointer).  Names/layouts are produced by canonswap_amd/pack.py from
 * the reference's state-dict keys (BatchNorm / spectral norm folded, channels-last, fp16 MFMA order). */
int cs_upload(cs_engine* e, const char* name, const void* host_ptr, size_t nbytes);
int cs_finalize_weights(cs_engine* e);
/* Per-identity precompute of the 14 modulated+demodulated conv weights of T
 * (AdaptiveSharedWeightConv2d.forward, src/modules/adaptive_modulate.py:148-155).  id: device, 512 fp32. */
#define CS_MAX_IDENTITY_SLOTS 8
/* slot in [0, CS_MAX_IDENTITY_SLOTS): each slot keeps its own 14 modulated weight sets (about 132 MB), so several source
 * identities stay resident (concurrent streams, BASELINE configs[4]) and may be mixed inside one batch (cs_swap_ids). */
int cs_set_identity(cs_engine* e, int slot, const float* id, void* stream);

/* Single-frame latency mode (BASELINE configs[1]): launches that cannot fill the 256 CUs at one or two frames per call take forms that
 * add an output element's products in another (fixed) order than the batched path - the 512-channel 3x3 convs split their K loop over twelve waves
 * of a workgroup (conv_lat.hip), the deep hourglass levels over workgroups (split-K), R's volume convs emit 2-row statistics blocks.  Deterministic,
 * the same tolerance against the reference (>= 50 dB), not bit-identical with the default mode - which is why it is opt-in: the DEFAULT mode never
 * picks a summation order by batch size, so there a frame's bits do not depend on the batch it is part of.  That guarantee does NOT hold inside
 * latency mode: conv_lat and the split-K forms are taken per launch by its workgroup count (N x tiles <= 256), so a frame run at B = 2 in this mode
 * may differ in its last bits from the same frame at B = 1 (both deterministic, both inside the tolerance). */
int cs_set_latency_mode(cs_engine* e, int on);

/* ---- stage calls; B frames per call, B <= max_batch --------------------------------------------------- */
/* can_swapper.extract_feature_3d (can_swap_e2e.py:165-172): img Bx3x256x256 -> f Bx32x16x64x64 */
int cs_extract_feature_3d(cs_engine* e, int B, const float* img, float* f_out, void* stream);
/* WarpingNetwork.warp(feature_3d, kp_source, kp_driving) (warping_network.py:49-62):
 * f Bx32x16x64x64, kp Bx21x3 -> f_out Bx32x16x64x64, occ_out Bx1x64x64 */
int cs_warp(cs_engine* e, int B, const float* f, const float* kp_source, const float* kp_driving, float* f_out,
            float* occ_out, void* stream);
/* WarpingNetwork.warp_out(out, occlusion_map) (warping_network.py:64-71): -> seg Bx256x64x64; occ may be NULL */
int cs_warp_out(cs_engine* e, int B, const float* f, const float* occ, float* seg_out, void* stream);
/* transfer_model2.forward(x, dlatents) == can_swapper.swap (adaptive_modulate.py:522-554), identity from slot */
int cs_swap(cs_engine* e, int slot, int B, const float* f, float* f_out, void* stream);
/* the same with one identity slot per sample (slots: B host ints): the reference's per-sample dlatents, i.e. the groups=N
 * modulated convolution of AdaptiveSharedWeightConv2d.forward (adaptive_modulate.py:157-167) */
int cs_swap_ids(cs_engine* e, const int* slots, int B, const float* f, float* f_out, void* stream);
/* G3d.forward (adaptive_modulate.py:721-733) */
int cs_refine(cs_engine* e, int B, const float* f, float* f_out, void* stream);
/* WarpingNetwork.forward(feature_3d, kp_driving=, kp_source=) (warping_network.py:83-111):
 * any of occ_out (Bx1x64x64), deformation_out (Bx16x64x64x3), seg_out (Bx256x64x64) may be NULL */
int cs_warp_forward(cs_engine* e, int B, const float* f, const float* kp_driving, const float* kp_source,
                    float* occ_out, float* deformation_out, float* seg_out, void* stream);
/* SPADEDecoder.forward (spade_generator.py:41-59): seg Bx256x64x64 -> img Bx3x512x512 in (0,1) */
int cs_spade_decode(cs_engine* e, int B, const float* seg, float* img_out, void* stream);
/* MotionExtractor.forward (motion_extractor.py:33-35 -> convnextv2.py:110-144), as called by can_swapper.get_kp_info
 * (can_swap_e2e.py:174-199): img Bx3x256x256 fp32 in [0,1] -> out Bx328 fp32, the raw head outputs concatenated in the order
 * kp(63) scale(1) pitch(66) yaw(66) roll(66) t(3) exp(63).  Needs the "M.*" blobs. */
int cs_motion_extract(cs_engine* e, int B, const float* img, float* out, void* stream);
/* can_swapper.parse_output on device (can_swap_e2e.py:314-322): Bx3xHxW fp32 -> BxHxWx3 u8 (truncation) */
int cs_pack_u8(cs_engine* e, int B, const float* img, uint8_t* out, int H, int W, void* stream);
/* can_swapper.prepare_source / prepare_videos on device (can_swap_e2e.py:126-163): BxHxWx3 u8 -> Bx3xHxW fp32 = u8 / 255 */
int cs_unpack_u8(cs_engine* e, int B, const uint8_t* img, float* out, int H, int W, void* stream);
/* The whole per-frame loop body (can_swap_pipeline_e2e.py:242-263) for B frames without leaving the device:
 * img Bx3x256x256, x_t / x_can Bx21x3.  out_f32 (Bx3x512x512), out_u8 (Bx512x512x3), rec_can, swap_can
 * (debug decodes of lines 248 / 257, Bx3x512x512) may each be NULL. */
int cs_swap_frames(cs_engine* e, int slot, int B, const float* img, const float* x_t, const float* x_can,
                   float* out_f32, uint8_t* out_u8, float* rec_can, float* swap_can, void* stream);
/* cs_swap_frames with one identity slot per frame (slots: B host ints) */
int cs_swap_frames_ids(cs_engine* e, const int* slots, int B, const float* img, const float* x_t, const float* x_can,
                       float* out_f32, uint8_t* out_u8, float* rec_can, float* swap_can, void* stream);
/* The per-frame body of the video-to-image pipeline (can_swap_pipeline_v2i.py:311-312):
 * warp_decode(f, kp_source, kp_driving) -> 3x512x512 for B driving frames.  f: nf x 32x16x64x64 feature volumes and
 * kp_source: ns x 21x3, with nf / ns = 1 (one swapped canonical volume / key-point set shared by all frames) or B;
 * kp_driving Bx21x3.  out_f32 (Bx3x512x512) and out_u8 (Bx512x512x3) may each be NULL. */
int cs_animate_frames(cs_engine* e, int B, const float* f, int nf, const float* kp_source, int ns, const float* kp_driving,
                      float* out_f32, uint8_t* out_u8, void* stream);

/* ---- image-space steps on either side of the generator; all buffers on the device ---- */
/* SoftErosion.forward (src/utils/crop.py:21-47; the pipeline builds it with kernel_size 21, threshold 0.9, iterations 3 / 2,
 * can_swap_pipeline_e2e.py:42, _v2i.py:43): mask BxHxW fp32 (0/1) -> soft_out BxHxW fp32, hard_out BxHxW u8 (may be NULL).
 * w: the ksize x ksize fp32 kernel (crop.py:29-35), built by the caller exactly as the reference builds it. */
int cs_soft_erosion(cs_engine* e, int B, int H, int W, const float* mask, const float* w, int ksize, float thr, int iters,
                    float* soft_out, uint8_t* hard_out, void* stream);
/* Input staging (src/utils/cropper.py:209 + can_swap_e2e.py:126-163): uint8 crops BxHcxWcx3, 512x512 (cv2.resize to 256x256 with
 * INTER_AREA = 2x2 means, (a+b+c+d+2)>>2) or 256x256 -> Bx3x256x256 fp32 = u8 / 255 */
int cs_prepare_crops(cs_engine* e, int B, const uint8_t* crops, int Hc, int Wc, float* out, void* stream);
/* cv2.warpAffine(src, M, dsize=(Wd, Hd), flags=INTER_LINEAR), BORDER_CONSTANT 0 (src/utils/crop.py:49-63 _transform_img):
 * M: host, 2x3 row major, source -> destination.  8-bit 3-channel and float 1-channel images. */
int cs_warp_affine_u8(cs_engine* e, const uint8_t* src, int Hs, int Ws, const double M[6], uint8_t* dst, int Hd, int Wd, void* stream);
int cs_warp_affine_f32(cs_engine* e, const float* src, int Hs, int Ws, const double M[6], float* dst, int Hd, int Wd, void* stream);
/* paste_back (src/utils/crop.py:523-529) fused with the mask warp of prepare_paste_back (:515-521): crop HcxWcx3 u8, the soft mask
 * either in the crop frame (mask_crop HcxWc, warped here) or already in the frame of the original image (mask_ori HoxWo) - exactly
 * one of the two -, M_c2o 2x3 (crop -> original), img_ori / out HoxWox3 u8:
 * out = clip(mask * warp(crop) + (1 - mask) * img_ori, 0, 255) truncated to 8 bits */
int cs_paste_back(cs_engine* e, const uint8_t* crop, const float* mask_crop, const float* mask_ori, int Hc, int Wc,
                  const double M_c2o[6], const uint8_t* img_ori, uint8_t* out, int Ho, int Wo, void* stream);

/* The same steps for the B frames of one launch of the per-frame loop (can_swap_pipeline_e2e.py:273-283 runs them frame by frame):
 * cs_soft_erosion_frames = B independent SoftErosion calls on (1,1,H,W) masks - the maximum of crop.py:45 is taken per frame, as in the
 * pipeline's loop, not over the batch; masks: BxHxW, fp32 or (masks_u8 != 0) uint8 0/1 labels (`torch.isin(labels, valid).to(int)`,
 * can_swap_pipeline_e2e.py:192).  cs_paste_back_batch = prepare_paste_back + paste_back of B frames in one launch: crops BxHcxWcx3 u8
 * (the generator's frames), masks_crop BxHcxWc fp32 (the soft masks, in the crop frame), M_c2o: HOST, B x 6 doubles (2x3 row major, crop ->
 * original, target_M_c2o_lst[i]), imgs_ori / out BxHoxWox3 u8. */
int cs_soft_erosion_frames(cs_engine* e, int B, int H, int W, const void* masks, int masks_u8, const float* w, int ksize, float thr, int iters,
                           float* soft_out, uint8_t* hard_out, void* stream);
int cs_paste_back_batch(cs_engine* e, int B, const uint8_t* crops, const float* masks_crop, int Hc, int Wc, const double* M_c2o,
                        const uint8_t* imgs_ori, uint8_t* out, int Ho, int Wo, void* stream);
/* Key-points of B frames from cs_motion_extract's raw head outputs, on the device: get_kp_info's refinement (can_swap_e2e.py:192-197,
 * camera.py:14-28), get_rotation_matrix (camera.py:31-73) and transform_keypoint (can_swap_e2e.py:228-256) -> x_t Bx21x3
 * = scale (kp R + exp) + t_xy, and x_can = scale kp Bx21x3 (can_swap_pipeline_e2e.py:243); rot (Bx3x3, may be NULL) = R.
 * Replaces make_motion_template's per-frame D2H of seven tensors (can_swap_pipeline_e2e.py:111-125). */
int cs_motion_keypoints(cs_engine* e, int B, const float* raw, float* x_t, float* x_can, float* rot, void* stream);

/* ---- measurement: per-kernel-family HIP-event timing on the launch stream */
int cs_profile_begin(cs_engine* e);
/* ms[0] = convolution kernels (conv_halo / conv_igemm), ms[1] = all other kernels except ms[2] = the feature warp
 * (grid_sample_kernel); counts likewise; flops = algorithmic conv FLOPs (2*MAC over the reference's logical channel
 * counts) enqueued since cs_profile_begin. CANONSWAP_PROFILE_CSV=<path> additionally dumps one line per launch. */
int cs_profile_end(cs_engine* e, double ms[3], long counts[3], double* flops);
/* MFMA FLOPs actually issued by the convolution launches since cs_profile_begin (padded channel counts, the taps the
 * phase-decomposed up-sampling convs really run): matrix-pipe utilisation, next to the algorithmic figure above */
int cs_profile_exec_flops(cs_engine* e, double* flops);

/* ---- operator level (unit parity tests) ---------------------------------------------------------------- */
typedef struct cs_conv_desc {
    const void* in;           /* fp16 channels-last */
    long in_sN, in_sD, in_sH, in_sW;
    int N, D, H, W, Cin, up_shift;
    int KD, KH, KW;
    const void* wgt;          /* packed fp16 (pack.py pack_conv) */
    int Cout_pad, Cout;
    const float* bias; const float* bias2;
    int act0; float slope0;
    const void* res; int res_f32; int res_shift; long res_sN, res_sD, res_sH, res_sW;
    const float* pixscale; int ps_stride;
    void* out0; int out0_f32; long out0_sN, out0_sD, out0_sH, out0_sW;
    const float* s2; const float* t2; int act1; float slope1;
    void* out1; long out1_sN, out1_sD, out1_sH, out1_sW;
    const float* stats;       /* SPADE: [N][C][2] = (mean, 1/sqrt(var+eps)) from cs_op_chan_stats */
    int mode;                 /* 0 std, 1 T blend, 2 SPADE, 3 pixel-shuffle + sigmoid, 5 std with out0 = act0(IN(res) * (1 + conv + bias)): SPADE's
                                 modulation without the beta half (res fp16, stats as for SPADE; /root/reference/src/modules/util.py:295-302) */
    int cfg;                  /* -2 auto conv_halo, 10..20 conv_halo tile cfg, 30 the vol32 kernel (3x3x3 32 -> 32 on [N][H][W][16][32]), 31 conv_wide, 32 conv_lat
                               * (both: 3x3, 2-D, the engine's tensor combinations; conv_lat: Cin = 512, another summation order); -1, 0..3: test-only library */
    int tile_w, tile_h;       /* 0 = auto */
    int ck;                   /* conv_halo channel chunk: 0 auto, 32 or 64 */
    int xcd_map;              /* conv_halo workgroup -> tile mapping: 0 engine default, k > 0 forces mapping k - 1 (common.h) */
    int ragged;               /* Cin % 32 == 16 and wgt went through cs_op_pair_ragged: paired taps in the last chunk (cfg 19 / 20 only) */
    int hilo;                 /* split-precision conv (util.py:528-544 convs of R): in = [hi | lo] per voxel, wgt = chunks W_hi | W_lo | W_hi, Cin = 96 */
    float* stat_out;          /* optional per-block partial (sum, sum of squares) of the stored fp32 out0 (cfg 30: [N][ceil(H/8) * W/2][32][2]) */
    /* cfg 30 with hilo: transform staging - the conv input is computed from fp32 volumes (strides of out0) while it is staged instead of read
     * from `in`: kind 1: xf_y; kind 2: lrelu((xf_y - mean) * rstd * gamma + beta [+ xf_res]) with (mean, rstd) = xf_stats[n][c][2], written back to
     * xf_out when given (util.py:531-540 fused into the consumer conv) */
    int xf_kind;
    const float* xf_y; const float* xf_res; float* xf_out;
    const float* xf_stats; const float* xf_gamma; const float* xf_beta;
    float xf_slope;
    int ep_general;           /* tests / A/B: 1 forces the general epilogue where a kernel also carries branch-free copies of it (same bits) */
    int pool_hw;              /* 1: out0 = AvgPool(1,2,2) of the activated conv output, computed in the epilogue; out0's strides address the pooled grid
                                 (DownBlock3d, /root/reference/src/modules/util.py:185-190) */
} cs_conv_desc;
int cs_op_conv(const cs_conv_desc* d, void* stream);
/* in place: re-pack the last 32-channel chunk of a packed conv weight [chunks * taps][Cout_pad][32] (Cin % 32 == 16) so that two
 * taps that are neighbours along the row share one 32-deep K-step (what the engine does for the hourglass tail, the mask conv and the
 * first encoder block at load time; dense_motion.py:88, util.py:185-190,261-263) */
int cs_op_pair_ragged(void* w, int Cout_pad, int nchunks, int KD, int KH, int KW, void* stream);
/* T's mask conv + sigmoid (adaptive_modulate.py:118-121,176: Conv2d(512, 1, 3, padding 1)) as a memory-bound VALU kernel: x fp16 [N][H][W][512],
 * w the layer's packed conv weight [16 chunks * 9 taps][16][32] (row 0 is the output channel), tmask[(n H W + h W + w) * 4] = sigmoid(conv + bias[0]) */
int cs_op_t_mask(const void* x, const void* wpacked, const float* bias, float* tmask, int N, int H, int W, void* stream);
/* one ResBlock3d of a feature volume (util.py:80-102, BatchNorms folded): contiguous [N][H][W][16][32] volumes a (fp16), x (fp32) ->
 * out0 (fp32) = conv2(relu(conv1(a) + b1)) + b2 + x, out1 (fp16) = act1(out0 * s2 + t2); w1 / w2 packed like every 3x3x3 32 -> 32 weight */
int cs_op_resblock3d(const void* a, const float* x, float* out0, void* out1, int N, int H, int W, const void* w1, const void* w2,
                     const float* b1, const float* b2, const float* s2, const float* t2, int act1, float slope1, void* stream);
int cs_op_grid_sample3d(const float* in_hwdc, const float* grid, float* out32, void* out16, int N, int D, int H, int W,
                        void* stream);
/* per-(n,c) mean and 1/sqrt(var+eps) of a [N][P][C] tensor; partials: scratch of cs_op_chan_stats_partial_floats floats */
long cs_op_chan_stats_partial_floats(int N, long P, int C);
int cs_op_chan_stats(const void* x, int is_f32, int N, long P, int C, float eps, float* partials, float* stats, void* stream);

#ifdef __cplusplus
}
#endif
#endif
