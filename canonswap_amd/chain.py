"""The whole device-side frame: the per-frame work of CanSwapPipeline.execute (src/can_swap_pipeline_e2e.py) for B frames per call
without leaving the GPU between the cropper's uint8 crop and the pasted-back uint8 frame (SURVEY.md section 8f rows N1-N3 around
the generator).

    reference (per frame, host round trips in brackets)                       here (B frames per launch, all on the device)
    cropper.py:209  cv2.resize(crop 512 -> 256, INTER_AREA) [host]            cs_prepare_crops
    can_swap_e2e.py:147-163  prepare_videos -> fp32 NCHW [upload]               "
    can_swap_pipeline_e2e.py:111-125  get_kp_info + transform_keypoint         cs_motion_extract + cs_motion_keypoints
                              [seven tensors to the host and back per frame]
    :242-263  F -> warp -> T -> R -> warp_decode                               cs_swap_frames_ids
    :267      parse_output [sync + D2H]                                        (pack_u8 inside cs_swap_frames_ids)
    :274      soft_mask(masks[i]) [D2H]                                        cs_soft_erosion_frames
    :279-282  prepare_paste_back + paste_back (two cv2.warpAffine) [host]      cs_paste_back_batch

What stays outside (out of scope): face detection / landmarks / the cropper's geometry (they produce the crops and
M_c2o), SegFormer face parsing (it produces the 0/1 masks), video decode / encode.
"""
from __future__ import annotations

import torch

from . import tail
from .engine import Engine


class FrameChain:
    """chain = FrameChain(swapper);  frames = chain(crops_u8, masks, M_c2o, frames_ori, source_id)["frames"]"""

    def __init__(self, swapper, kernel_size: int = 21, threshold: float = 0.9, iterations: int = 3):
        self.sw = swapper
        self.e: Engine = swapper.engine
        if swapper.motion_extractor is None:
            raise RuntimeError("FrameChain: the loaded weights hold no 'motion_extractor' state-dict")
        self.se = tail.SoftErosion(self.e, kernel_size, threshold, iterations)      # SoftErosion(21, 0.9, 3): can_swap_pipeline_e2e.py:42
        self._buf = {}

    def _get(self, key, shape, dtype):
        t = self._buf.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.e.device)
            self._buf[key] = t
        return t

    def keypoints(self, I):
        """(B,3,256,256) fp32 -> x_t, x_can (B,21,3): make_motion_template's get_kp_info + transform_keypoint and the loop's
        x_can = scale * kp (can_swap_pipeline_e2e.py:111-125, 236-243)."""
        raw = self.e.motion_extract_raw(I, out=self._get("raw", (I.shape[0], 328), torch.float32))
        return self.e.motion_keypoints(raw)

    def __call__(self, crops_u8, masks, M_c2o, frames_ori, source_id=None, slots=None, out=None, keep=False):
        """crops_u8 (B,512,512,3) or (B,256,256,3) u8; masks (B,512,512) u8 0/1 or fp32 (the parser's `torch.isin(labels, valid)`);
        M_c2o (B,2,3)/(B,3,3) host; frames_ori (B,Ho,Wo,3) u8; source_id (1,512)/(B,512) or identity slots.
        -> {"frames": (B,Ho,Wo,3) u8[, "crops_out", "x_t", "x_can", "soft_mask" with keep=True]}"""
        e = self.e
        I = tail.prepare_crops(e, crops_u8)                                                   # cropper.py:209 + can_swap_e2e.py:147-163
        x_t, x_can = self.keypoints(I)                                                        # can_swap_pipeline_e2e.py:111-125, 243
        B = I.shape[0]
        gen = e.swap_frames(I, x_t, x_can, source_id, want_f32=False, want_u8=True, slots=slots,
                            out_u8=self._get("gen", (B, 512, 512, 3), torch.uint8))["out_u8"]   # :242-267
        m = torch.as_tensor(masks)
        soft = tail.soft_erosion_frames(e, m, self.se.weight, self.se.kernel_size, self.se.threshold, self.se.iterations,
                                        out=self._get("soft", (B,) + tuple(m.shape[-2:]), torch.float32))      # :274
        frames = tail.paste_back_batch(e, gen, soft, M_c2o, frames_ori, out=out)                # :279-282
        res = {"frames": frames}
        if keep:
            res.update(crops_out=gen, x_t=x_t, x_can=x_can, soft_mask=soft, I=I)
        return res
