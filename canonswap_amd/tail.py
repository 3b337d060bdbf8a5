"""Image-space steps on either side of the generator, on the device.

Counterparts of the reference's host-side helpers, same names and argument meaning, operating on device tensors so the
per-frame loop (src/can_swap_pipeline_e2e.py:223-283) no longer leaves the GPU between the generator and the final frame:

* ``SoftErosion``            src/utils/crop.py:21-47            (the reference runs it with .cuda() too: pure torch)
* ``prepare_paste_back``     src/utils/crop.py:515-521          (cv2.warpAffine of the float mask)
* ``paste_back``             src/utils/crop.py:523-529          (cv2.warpAffine of the crop + blend)
* ``paste_back_fused``       both of the above in one kernel launch per frame
* ``prepare_crops``          src/utils/cropper.py:209 + src/can_swap_e2e.py:126-163 (INTER_AREA 512 -> 256, /255, HWC -> CHW)
* ``FrameStreamer``          streamed upload of the uint8 crops instead of the whole-video residency of prepare_videos

Every operation is a HIP kernel of libcanonswap_hip.so (csrc/imgops.hip); there is no torch / CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .engine import Engine, _ptr


def _m6(M):
    m = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(-1)[:6])      # 2x3 or the top rows of a 3x3
    return m, m.ctypes.data_as(C.POINTER(C.c_double))


class SoftErosion:
    """SoftErosion(kernel_size, threshold, iterations) of crop.py:21-47; call -> (soft mask, hard mask), inputs (N,1,H,W)."""

    def __init__(self, engine: Engine, kernel_size=15, threshold=0.6, iterations=1):
        self.e, self.kernel_size, self.threshold, self.iterations = engine, kernel_size, threshold, iterations
        r = kernel_size // 2
        # the kernel exactly as the reference builds it (crop.py:29-35), in fp32 on the host
        y, x = torch.meshgrid(torch.arange(0., kernel_size), torch.arange(0., kernel_size), indexing="ij")
        dist = torch.sqrt((x - r) ** 2 + (y - r) ** 2)
        k = dist.max() - dist
        k /= k.sum()
        self.weight = k.view(1, 1, kernel_size, kernel_size).to(engine.device)

    def __call__(self, x: torch.Tensor):
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError("SoftErosion expects (N, 1, H, W)")
        e = self.e
        m = x.to(e.device).float().contiguous()
        N, _, H, W = m.shape
        soft = torch.empty_like(m)
        hard = torch.empty((N, 1, H, W), dtype=torch.uint8, device=e.device)
        with torch.cuda.device(e.device):
            _lib.check(e.lib.cs_soft_erosion(e.h, N, H, W, _ptr(m), _ptr(self.weight), self.kernel_size, float(self.threshold),
                                             self.iterations, _ptr(soft), _ptr(hard), e._stream()), "cs_soft_erosion")
        return soft, hard.bool()

    forward = __call__


def soft_erosion_frames(e: Engine, masks, weight, kernel_size=21, threshold=0.9, iterations=3, out=None):
    """B independent SoftErosion calls (the pipeline's loop calls the module once per frame, can_swap_pipeline_e2e.py:274: the maximum of
    crop.py:45 is that frame's) in one launch sequence.  masks: (B,H,W) uint8 0/1 labels or fp32, on the device -> soft masks (B,H,W) fp32."""
    m = torch.as_tensor(masks)
    if m.dim() == 4 and m.shape[1] == 1:
        m = m[:, 0]
    if m.dim() != 3:
        raise ValueError("soft_erosion_frames expects (B, H, W) masks")
    if m.dtype not in (torch.uint8, torch.float32):
        m = m.to(torch.uint8) if not m.dtype.is_floating_point else m.float()
    m = m.to(e.device).contiguous()
    B, H, W = m.shape
    soft = torch.empty((B, H, W), dtype=torch.float32, device=e.device) if out is None else out
    with torch.cuda.device(e.device):
        _lib.check(e.lib.cs_soft_erosion_frames(e.h, B, H, W, _ptr(m), int(m.dtype == torch.uint8), _ptr(weight), kernel_size, float(threshold),
                                                iterations, _ptr(soft), None, e._stream()), "cs_soft_erosion_frames")
    return soft


def paste_back_batch(e: Engine, crops, masks_crop, M_c2o, imgs_ori, out=None):
    """prepare_paste_back + paste_back (crop.py:515-529) of B frames in one launch: crops (B,Hc,Wc,3) u8, masks_crop (B,Hc,Wc) fp32 soft masks
    in the crop frame, M_c2o (B,2,3) or (B,3,3) host matrices crop -> original, imgs_ori (B,Ho,Wo,3) u8 -> (B,Ho,Wo,3) u8."""
    crops, ori = torch.as_tensor(crops), torch.as_tensor(imgs_ori)
    if crops.dtype != torch.uint8 or crops.dim() != 4 or crops.shape[3] != 3 or ori.dtype != torch.uint8 or ori.dim() != 4 or ori.shape[3] != 3:
        raise ValueError("expected BxHxWx3 uint8 crops and original frames")
    crops, ori = crops.to(e.device).contiguous(), ori.to(e.device).contiguous()
    B = crops.shape[0]
    mc = torch.as_tensor(masks_crop).to(e.device).float().contiguous()
    if tuple(mc.shape) != tuple(crops.shape[:3]) or ori.shape[0] != B:
        raise ValueError("masks_crop must be (B, Hc, Wc) and imgs_ori must hold B frames")
    M = np.ascontiguousarray(np.asarray(M_c2o, dtype=np.float64).reshape(B, -1)[:, :6])
    if out is None:
        out = torch.empty_like(ori)
    elif out.dtype != torch.uint8 or tuple(out.shape) != tuple(ori.shape) or not out.is_contiguous() or out.device != ori.device:
        raise ValueError("out must be a contiguous uint8 tensor of the shape of imgs_ori on the engine's device")
    with torch.cuda.device(e.device):
        _lib.check(e.lib.cs_paste_back_batch(e.h, B, _ptr(crops), _ptr(mc), crops.shape[1], crops.shape[2], M.ctypes.data_as(C.POINTER(C.c_double)),
                                             _ptr(ori), _ptr(out), ori.shape[1], ori.shape[2], e._stream()), "cs_paste_back_batch")
    return out


def _u8_hwc(e: Engine, img):
    t = torch.as_tensor(img)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("expected an HxWx3 uint8 image")
    return t.to(e.device).contiguous()


def warp_affine_u8(e: Engine, img, M, dsize):
    """cv2.warpAffine(img, M[:2], dsize=(W, H), flags=cv2.INTER_LINEAR) for HxWx3 uint8 (crop.py:49-63 _transform_img)."""
    src = _u8_hwc(e, img)
    Wd, Hd = int(dsize[0]), int(dsize[1])
    dst = torch.empty((Hd, Wd, 3), dtype=torch.uint8, device=e.device)
    m, mp = _m6(M)
    with torch.cuda.device(e.device):
        _lib.check(e.lib.cs_warp_affine_u8(e.h, _ptr(src), src.shape[0], src.shape[1], mp, _ptr(dst), Hd, Wd, e._stream()), "cs_warp_affine_u8")
    return dst


def prepare_paste_back(e: Engine, mask_crop, crop_M_c2o, dsize):
    """crop.py:515-521 with if_float=True (the call of can_swap_pipeline_e2e.py:279): float mask -> frame of the original image.
    mask_crop: HxW or HxWx3 (the reference stacks three identical channels); returns HoxWo float32 on the device."""
    m = torch.as_tensor(mask_crop)
    if m.dim() == 3:
        m = m[..., 0]
    m = m.to(e.device).float().contiguous()
    Wd, Hd = int(dsize[0]), int(dsize[1])
    dst = torch.empty((Hd, Wd), dtype=torch.float32, device=e.device)
    mm, mp = _m6(crop_M_c2o)
    with torch.cuda.device(e.device):
        _lib.check(e.lib.cs_warp_affine_f32(e.h, _ptr(m), m.shape[0], m.shape[1], mp, _ptr(dst), Hd, Wd, e._stream()), "cs_warp_affine_f32")
    return dst


def paste_back(e: Engine, img_crop, M_c2o, img_ori, mask_ori):
    """crop.py:523-529: result = warp(img_crop); clip(mask_ori * result + (1 - mask_ori) * img_ori, 0, 255) as uint8."""
    crop, ori = _u8_hwc(e, img_crop), _u8_hwc(e, img_ori)
    mo = torch.as_tensor(mask_ori)
    if mo.dim() == 3:
        mo = mo[..., 0]
    mo = mo.to(e.device).float().contiguous()
    if tuple(mo.shape) != tuple(ori.shape[:2]):
        raise ValueError("mask_ori must have the size of img_ori")
    out = torch.empty_like(ori)
    mm, mp = _m6(M_c2o)
    with torch.cuda.device(e.device):
        _lib.check(e.lib.cs_paste_back(e.h, _ptr(crop), None, _ptr(mo), crop.shape[0], crop.shape[1], mp, _ptr(ori), _ptr(out),
                                       ori.shape[0], ori.shape[1], e._stream()), "cs_paste_back")
    return out


def paste_back_fused(e: Engine, img_crop, mask_crop, M_c2o, img_ori):
    """prepare_paste_back + paste_back in one launch: the soft mask stays in the crop frame (HcxWc float32)."""
    crop, ori = _u8_hwc(e, img_crop), _u8_hwc(e, img_ori)
    mc = torch.as_tensor(mask_crop)
    if mc.dim() == 3:
        mc = mc[..., 0]
    mc = mc.to(e.device).float().contiguous()
    if tuple(mc.shape) != tuple(crop.shape[:2]):
        raise ValueError("mask_crop must have the size of img_crop")
    out = torch.empty_like(ori)
    mm, mp = _m6(M_c2o)
    with torch.cuda.device(e.device):
        _lib.check(e.lib.cs_paste_back(e.h, _ptr(crop), _ptr(mc), None, crop.shape[0], crop.shape[1], mp, _ptr(ori), _ptr(out),
                                       ori.shape[0], ori.shape[1], e._stream()), "cs_paste_back")
    return out


def prepare_crops(e: Engine, crops_u8, out=None) -> torch.Tensor:
    """uint8 crops (B,512,512,3) or (B,256,256,3), host or device -> (B,3,256,256) fp32 in [0,1] on the device: the cropper's
    cv2.resize(..., (256, 256), INTER_AREA) (cropper.py:209) fused with prepare_source / prepare_videos (can_swap_e2e.py:126-163)."""
    t = torch.as_tensor(crops_u8)
    if t.dim() == 3:
        t = t[None]
    if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3:
        raise ValueError("expected BxHxWx3 uint8 crops")
    t = t.to(e.device).contiguous()
    B, H, W, _ = t.shape
    if out is None:
        out = torch.empty((B, 3, 256, 256), dtype=torch.float32, device=e.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (B, 3, 256, 256) or not out.is_contiguous() or out.device != e.device:
        raise ValueError("out must be a contiguous (B, 3, 256, 256) fp32 tensor on the engine's device")
    with torch.cuda.device(e.device):
        _lib.check(e.lib.cs_prepare_crops(e.h, B, _ptr(t), H, W, _ptr(out), e._stream()), "cs_prepare_crops")
    return out


class FrameStreamer:
    """Streams the uint8 crops of a video to the device in batches (N3): batch k+1 is copied from pinned host memory on a side
    stream while batch k is being processed, instead of prepare_videos' whole-video upload as fp32 (can_swap_e2e.py:147-163,
    786 KB per frame; a uint8 512x512 crop is the same 786 KB but is converted on the device, a 256x256 one is 196 KB).

        for I_batch, (start, stop) in FrameStreamer(engine, crops_u8, batch=32): ...   # I_batch: (n, 3, 256, 256) fp32
    """

    def __init__(self, engine: Engine, crops_u8, batch: int = 32):
        self.e, self.batch = engine, batch
        self.frames = crops_u8
        self.n = len(crops_u8)
        f0 = np.asarray(crops_u8[0])
        if f0.dtype != np.uint8 or f0.ndim != 3 or f0.shape[2] != 3 or f0.shape[0] not in (256, 512) or f0.shape[0] != f0.shape[1]:
            raise ValueError("expected 256x256x3 or 512x512x3 uint8 crops")
        self.shape = f0.shape
        self.copy_stream = torch.cuda.Stream(device=engine.device)
        self.pinned = [torch.empty((batch,) + tuple(self.shape), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.dev = [torch.empty((batch,) + tuple(self.shape), dtype=torch.uint8, device=engine.device) for _ in range(2)]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.consumed = [torch.cuda.Event(), torch.cuda.Event()]

    def _stage(self, k):
        a, b = k * self.batch, min(self.n, (k + 1) * self.batch)
        s = k % 2
        self.consumed[s].synchronize()                       # the pinned slot is free again (first use: event never recorded)
        for j in range(a, b):
            self.pinned[s][j - a].copy_(torch.from_numpy(np.ascontiguousarray(self.frames[j])))
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.consumed[s])
            self.dev[s][: b - a].copy_(self.pinned[s][: b - a], non_blocking=True)
            self.ready[s].record(self.copy_stream)
        return a, b

    def __iter__(self):
        nb = (self.n + self.batch - 1) // self.batch
        if nb == 0:
            return
        span = self._stage(0)
        for k in range(nb):
            nxt = self._stage(k + 1) if k + 1 < nb else None          # upload of the next batch overlaps this batch's work
            s = k % 2
            cur = torch.cuda.current_stream(self.e.device)
            cur.wait_event(self.ready[s])
            out = prepare_crops(self.e, self.dev[s][: span[1] - span[0]])
            self.consumed[s].record(cur)
            yield out, span
            span = nxt
