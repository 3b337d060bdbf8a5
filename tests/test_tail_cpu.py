"""Oracle of the image-space steps (oracle/cv_ref.py) against the reference's own SoftErosion vectors and against
self-consistency properties of the OpenCV restatement (no cv2 in the build container: parity with cv2 itself is unpinned)."""
import numpy as np
import torch

from oracle import cv_ref as R


def test_soft_erosion_oracle_equals_reference_class(golden):
    g = golden("soft_erosion.npz")
    for name, (ks, thr, it) in {"e2e": (21, 0.9, 3), "v2i": (21, 0.9, 2)}.items():
        assert np.array_equal(R.soft_erosion_kernel(ks).numpy(), g[name + "_weight"])
        for k in (0, 1):
            x, hard = R.soft_erosion(torch.from_numpy(g[f"{name}_{k}_in"].astype(np.int32))[None, None], ks, thr, it)
            assert np.array_equal(x[0, 0].numpy(), g[f"{name}_{k}_soft"])
            assert np.array_equal(hard[0, 0].numpy(), g[f"{name}_{k}_hard"])


def test_warp_affine_identity_translation_and_border():
    r = np.random.Generator(np.random.PCG64(2))
    img = r.integers(0, 256, size=(40, 56, 3), dtype=np.uint8)
    eye = np.array([[1, 0, 0], [0, 1, 0]], np.float64)
    assert np.array_equal(R.warp_affine_u8(img, eye, (56, 40)), img)
    sh = np.array([[1, 0, 5], [0, 1, -3]], np.float64)                 # integer shift: exact copy, zeros outside
    out = R.warp_affine_u8(img, sh, (56, 40))
    assert np.array_equal(out[:37, 5:], img[3:, :51]) and not out[37:].any() and not out[:, :5].any()
    half = np.array([[1, 0, 0.5], [0, 1, 0]], np.float64)             # half-pixel shift: rounded mean of horizontal neighbours
    out = R.warp_affine_u8(img, half, (56, 40))
    want = (img[:, :-1].astype(np.int32) + img[:, 1:].astype(np.int32) + 1) >> 1
    assert np.array_equal(out[:, 1:], want.astype(np.uint8))
    f = img[..., 0].astype(np.float32)
    assert np.array_equal(R.warp_affine_f32(f, eye, (56, 40)), f)


def test_resize_area_and_paste_back_semantics():
    r = np.random.Generator(np.random.PCG64(3))
    img = r.integers(0, 256, size=(8, 8, 3), dtype=np.uint8)
    small = R.resize_area_2x_u8(img)
    assert small.shape == (4, 4, 3) and int(small[1, 2, 0]) == (int(img[2, 4, 0]) + int(img[2, 5, 0]) + int(img[3, 4, 0]) + int(img[3, 5, 0]) + 2) >> 2
    crop = r.integers(0, 256, size=(32, 32, 3), dtype=np.uint8)
    ori = r.integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
    M = np.array([[0.8, 0.1, 10.2], [-0.1, 0.8, 7.7], [0, 0, 1]])
    ones = R.prepare_paste_back(np.ones((32, 32, 3), np.float32), M, (64, 48))
    out = R.paste_back(crop, M, ori, ones)
    inside = ones[..., 0] == 1.0
    assert inside.sum() > 200 and np.array_equal(out[inside], R.warp_affine_u8(crop, M[:2], (64, 48))[inside])
    assert np.array_equal(out[ones[..., 0] == 0.0], ori[ones[..., 0] == 0.0])


def test_opencv_restatements_against_independent_implementations():
    """cv2 is not in the image, so the OpenCV restatements stay PARITY UNPINNED; what can be checked here is that two independent
    libraries agree with them: PIL's integer box reduction equals resize_area_2x_u8 bit for bit (the (a+b+c+d+2)>>2 rounding of
    cropper.py:209's INTER_AREA at exactly 2x), and scipy's float bilinear resampling agrees with warp_affine (crop.py:49-63, 515-529:
    M maps source to destination, pixel centres at integer coordinates) to within the restated 1/32-pixel coordinate grid +
    rounding wherever all four taps lie inside the source."""
    from PIL import Image
    from scipy import ndimage
    r = np.random.Generator(np.random.PCG64(5))
    img = r.integers(0, 256, size=(512, 512, 3), dtype=np.uint8)
    assert np.array_equal(R.resize_area_2x_u8(img), np.asarray(Image.fromarray(img).reduce(2)))
    yy, xx = np.mgrid[0:96, 0:128]
    smooth = (127.5 + 60 * np.sin(xx / 9.0) + 50 * np.cos(yy / 7.0)).astype(np.float32)       # gradient <= 6.7 / 7.2 per pixel
    src8 = np.clip(smooth, 0, 255).astype(np.uint8)
    for M in (np.array([[0.9, 0.15, 6.3], [-0.12, 0.85, 4.9]]), np.array([[1.3, -0.2, -9.1], [0.25, 1.2, 3.4]])):
        Hd, Wd = 120, 150
        o8 = R.warp_affine_u8(np.repeat(src8[..., None], 3, 2), M, (Wd, Hd))[..., 1]
        of = R.warp_affine_f32(smooth, M, (Wd, Hd))
        Mi = np.linalg.inv(np.vstack([M, [0, 0, 1]]))
        Y, X = np.mgrid[0:Hd, 0:Wd]
        sx = Mi[0, 0] * X + Mi[0, 1] * Y + Mi[0, 2]
        sy = Mi[1, 0] * X + Mi[1, 1] * Y + Mi[1, 2]
        inside = (sx >= 0) & (sx <= 127) & (sy >= 0) & (sy <= 95)
        assert inside.sum() > 5000
        ref = ndimage.map_coordinates(smooth, [sy, sx], order=1, mode="constant", cval=0.0)
        ref8 = ndimage.map_coordinates(src8.astype(np.float64), [sy, sx], order=1, mode="constant", cval=0.0)
        # 1/32-pixel coordinates: at most 1/64 pixel off per axis -> (6.7 + 7.2) / 64 = 0.22 grey levels, + 0.5 rounding for u8
        assert np.abs(of - ref)[inside].max() < 0.3
        assert np.abs(o8 - ref8)[inside].max() < 0.85
        outside = (sx < -1) | (sx > 128) | (sy < -1) | (sy > 96)
        assert not o8[outside].any() and not of[outside].any()
