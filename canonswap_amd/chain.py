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
        self._side, self._free, self._pending, self._nslot = None, None, [], 0        # prefetch(): side stream, double buffer

    def _get(self, key, shape, dtype):
        t = self._buf.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.e.device)
            self._buf[key] = t
        return t

    def keypoints(self, I, slot=0):
        """(B,3,256,256) fp32 -> x_t, x_can (B,21,3): make_motion_template's get_kp_info + transform_keypoint and the loop's
        x_can = scale * kp (can_swap_pipeline_e2e.py:111-125, 236-243)."""
        B = I.shape[0]
        raw = self.e.motion_extract_raw(I, out=self._get(("raw", slot), (B, 328), torch.float32))
        return self.e.motion_keypoints(raw, out=(self._get(("x_t", slot), (B, 21, 3), torch.float32), self._get(("x_can", slot), (B, 21, 3), torch.float32)))

    # ---- stage A: everything the generator needs from a batch of crops (input staging + motion extractor).  The reference runs it as a
    # pre-pass over the whole video (prepare_videos + make_motion_template, can_swap_pipeline_e2e.py:196-197) before the swapping loop
    # ... and the soft mask, which depends on the parser's labels only (:274)
    def _stage_a(self, crops_u8, masks, slot):
        t = torch.as_tensor(crops_u8)
        B = t.shape[0] if t.dim() == 4 else 1
        I = tail.prepare_crops(self.e, t, out=self._get(("I", slot), (B, 3, 256, 256), torch.float32))      # cropper.py:209 + can_swap_e2e.py:147-163
        x_t, x_can = self.keypoints(I, slot)                                                              # can_swap_pipeline_e2e.py:111-125, 243
        m = torch.as_tensor(masks)
        soft = tail.soft_erosion_frames(self.e, m, self.se.weight, self.se.kernel_size, self.se.threshold, self.se.iterations,
                                        out=self._get(("soft", slot), (B,) + tuple(m.shape[-2:]), torch.float32))      # :274
        return I, x_t, x_can, soft

    def prefetch(self, crops_u8, masks):
        """Stage A of the NEXT batch on a side stream, so that it runs beside the generator of the current one (M and the staging are
        bandwidth / latency bound, the generator is matrix-pipe bound): call it before __call__ of the current batch; the next __call__ with
        the same crops tensor picks the result up (the soft masks of that batch included: they depend on the parser's labels only).  M's workspace is its own, the batch's inputs land in the other half of a double buffer."""
        e = self.e
        if self._side is None:
            self._side = torch.cuda.Stream(device=e.device)      # (a high-priority side stream measured the same: 0.933 either way)
            self._free = [torch.cuda.Event(), torch.cuda.Event()]
        if len(self._pending) >= 2:
            raise RuntimeError("FrameChain.prefetch: two batches are already staged (double buffer); run one of them first")
        slot = self._nslot = 1 - self._nslot
        main = torch.cuda.current_stream(e.device)
        self._side.wait_stream(main)                      # the crops were produced on the caller's stream
        self._side.wait_event(self._free[slot])           # the generator that read this half of the double buffer is done
        with torch.cuda.stream(self._side):
            res = self._stage_a(crops_u8, masks, slot)
            ready = torch.cuda.Event()
            ready.record(self._side)
        self._pending.append((crops_u8, slot, res, ready))

    def drop_prefetches(self):
        """Forget staged batches that will not be run (their buffers are reused by the next prefetch)."""
        self._pending = []

    def __call__(self, crops_u8, masks, M_c2o, frames_ori, source_id=None, slots=None, out=None, keep=False):
        """crops_u8 (B,512,512,3) or (B,256,256,3) u8; masks (B,512,512) u8 0/1 or fp32 (the parser's `torch.isin(labels, valid)`);
        M_c2o (B,2,3)/(B,3,3) host; frames_ori (B,Ho,Wo,3) u8; source_id (1,512)/(B,512) or identity slots.
        -> {"frames": (B,Ho,Wo,3) u8[, "crops_out", "x_t", "x_can", "soft_mask" with keep=True]}"""
        e = self.e
        main = torch.cuda.current_stream(e.device)
        slot = None
        hit = [k for k, q in enumerate(self._pending) if q[0] is crops_u8]
        if hit:
            _, slot, (I, x_t, x_can, soft), ready = self._pending.pop(hit[0])
            main.wait_event(ready)
        else:
            I, x_t, x_can, soft = self._stage_a(crops_u8, masks, "inline")
        B = I.shape[0]
        gen = e.swap_frames(I, x_t, x_can, source_id, want_f32=False, want_u8=True, slots=slots,
                            out_u8=self._get("gen", (B, 512, 512, 3), torch.uint8))["out_u8"]   # :242-267
        frames = tail.paste_back_batch(e, gen, soft, M_c2o, frames_ori, out=out)                # :279-282
        if slot is not None:
            self._free[slot].record(main)
        res = {"frames": frames}
        if keep:
            res.update(crops_out=gen, x_t=x_t, x_can=x_can, soft_mask=soft, I=I)
        return res
