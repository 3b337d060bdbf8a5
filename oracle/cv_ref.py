"""CPU restatement of the image-space steps around the generator.  TEST INFRASTRUCTURE ONLY:
imported by tests/, never by the product package.

* soft_erosion: src/utils/crop.py:21-47 (pure torch in the reference).  PINNED: tests/golden/soft_erosion.npz was produced by
  executing the reference's own class (tools/make_golden_tail.py extracts it from crop.py, whose module import needs cv2).
* resize_area_2x_u8 / warp_affine / paste_back: the reference calls cv2.resize(..., INTER_AREA) (src/utils/cropper.py:209) and
  cv2.warpAffine(..., INTER_LINEAR) (src/utils/crop.py:49-63, 515-529).  OpenCV (opencv-python, no version pinned by the
  reference's requirements) is not installed in the build container and is not under /root/reference, so these restate OpenCV's
  published algorithm (modules/imgproc/src/resize.cpp ResizeAreaFastVec: (a+b+c+d+2)>>2 for 2x2 areas;
  modules/imgproc/src/imgwarp.cpp warpAffine / remapBilinear: 10-bit fixed-point source coordinates rounded to 1/32 pixel,
  32x32 bilinear table, 15-bit integer weights for 8-bit images, float weights for float images, BORDER_CONSTANT 0).
  PARITY UNPINNED against a real cv2 build: there are no reference-owned vectors for these calls.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
COEF_BITS = 15


# ---------------------------------------------------------------------------------------- crop.py:21-47
def soft_erosion_kernel(kernel_size: int) -> torch.Tensor:
    r = kernel_size // 2
    y, x = torch.meshgrid(torch.arange(0., kernel_size), torch.arange(0., kernel_size), indexing="ij")
    dist = torch.sqrt((x - r) ** 2 + (y - r) ** 2)
    k = dist.max() - dist
    return (k / k.sum()).view(1, 1, kernel_size, kernel_size)


def soft_erosion(x: torch.Tensor, kernel_size=21, threshold=0.9, iterations=3):
    """x: (N,1,H,W). Returns (soft mask, hard mask) like SoftErosion.forward."""
    w = soft_erosion_kernel(kernel_size)
    pad = kernel_size // 2
    x = x.float()
    for _ in range(iterations - 1):
        x = torch.min(x, F.conv2d(x, weight=w, groups=x.shape[1], padding=pad))
    x = F.conv2d(x, weight=w, groups=x.shape[1], padding=pad)
    mask = x >= threshold
    x = x.clone()
    x[mask] = 1.0
    x[~mask] /= x[~mask].max()
    return x, mask


# ---------------------------------------------------------------------------------------- cropper.py:209
def resize_area_2x_u8(img: np.ndarray) -> np.ndarray:
    """cv2.resize(img, (W/2, H/2), interpolation=cv2.INTER_AREA) for uint8 HxWxC with even H, W."""
    a = img.astype(np.int32)
    s = a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2]
    return ((s + 2) >> 2).astype(np.uint8)


# ---------------------------------------------------------------------------------------- crop.py:49-63
def invert_affine(M) -> np.ndarray:
    """What cv2.warpAffine does to M (2x3, source -> destination) without WARP_INVERSE_MAP: the destination -> source map."""
    M = np.array(M, dtype=np.float64).reshape(-1)[:6].copy()
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def _coords(Minv, Hd, Wd):
    """Fixed-point source coordinates of every destination pixel: integer part (sx, sy) and the 5-bit fractions (fx, fy)."""
    x = np.arange(Wd, dtype=np.float64)
    y = np.arange(Hd, dtype=np.float64)
    adelta = np.rint(Minv[0] * x * AB_SCALE).astype(np.int64)          # saturate_cast<int>(double) = round half to even
    bdelta = np.rint(Minv[3] * x * AB_SCALE).astype(np.int64)
    rd = AB_SCALE // INTER_TAB_SIZE // 2
    X0 = np.rint((Minv[1] * y + Minv[2]) * AB_SCALE).astype(np.int64) + rd
    Y0 = np.rint((Minv[4] * y + Minv[5]) * AB_SCALE).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)                        # saturate_cast<short>
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    return sx, sy, X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1)


def _taps(src, sx, sy):
    """The four bilinear taps with BORDER_CONSTANT 0: arrays (Hd, Wd, C) for (y0,x0), (y0,x1), (y1,x0), (y1,x1)."""
    Hs, Ws = src.shape[:2]
    out = []
    for dy in (0, 1):
        for dx in (0, 1):
            yy, xx = sy + dy, sx + dx
            ok = (yy >= 0) & (yy < Hs) & (xx >= 0) & (xx < Ws)
            v = src[np.clip(yy, 0, Hs - 1), np.clip(xx, 0, Ws - 1)]
            out.append(np.where(ok[..., None], v, 0))
    return out


def warp_affine_u8(src: np.ndarray, M, dsize) -> np.ndarray:
    """cv2.warpAffine(src, M[:2], dsize=(W, H), flags=INTER_LINEAR) for uint8 HxWx3."""
    Wd, Hd = dsize
    sx, sy, fx, fy = _coords(invert_affine(M), Hd, Wd)
    t00, t01, t10, t11 = [t.astype(np.int64) for t in _taps(src, sx, sy)]
    fx, fy = fx[..., None], fy[..., None]
    # 15-bit weights: (1-fy)(1-fx) ... with fx, fy in 1/32 steps -> exact multiples of 32, summing to 1 << 15
    w00 = (32 - fy) * (32 - fx) * 32; w01 = (32 - fy) * fx * 32; w10 = fy * (32 - fx) * 32; w11 = fy * fx * 32
    acc = t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11
    return ((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS).astype(np.uint8)


def warp_affine_f32(src: np.ndarray, M, dsize) -> np.ndarray:
    """The same for float32 images (HxW or HxWxC): float table weights, left-to-right float32 sum."""
    Wd, Hd = dsize
    s3 = src[..., None] if src.ndim == 2 else src
    sx, sy, fx, fy = _coords(invert_affine(M), Hd, Wd)
    t00, t01, t10, t11 = [t.astype(np.float32) for t in _taps(s3, sx, sy)]
    scale = np.float32(1.0) / np.float32(INTER_TAB_SIZE)
    ax = (fx.astype(np.float32) * scale)[..., None]
    ay = (fy.astype(np.float32) * scale)[..., None]
    one = np.float32(1.0)
    w00 = (one - ay) * (one - ax); w01 = (one - ay) * ax; w10 = ay * (one - ax); w11 = ay * ax
    out = ((t00 * w00 + t01 * w01) + t10 * w10) + t11 * w11
    return out[..., 0] if src.ndim == 2 else out


# ---------------------------------------------------------------------------------------- crop.py:515-529
def prepare_paste_back(mask_crop: np.ndarray, crop_M_c2o, dsize) -> np.ndarray:
    """if_float=True branch used by the pipeline (can_swap_pipeline_e2e.py:279): float mask warped to the original frame."""
    return warp_affine_f32(mask_crop.astype(np.float32), np.asarray(crop_M_c2o)[:2], dsize)


def paste_back(img_crop: np.ndarray, M_c2o, img_ori: np.ndarray, mask_ori: np.ndarray) -> np.ndarray:
    dsize = (img_ori.shape[1], img_ori.shape[0])
    result = warp_affine_u8(img_crop, np.asarray(M_c2o)[:2], dsize)
    out = mask_ori * result + (1 - mask_ori) * img_ori          # float32 * uint8 -> float32, as in numpy
    return np.clip(out, 0, 255).astype(np.uint8)
