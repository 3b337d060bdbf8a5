"""Drop-in counterpart of the reference's model wrapper ``can_swapper`` (src/can_swap_e2e.py:39-348).

Same constructor argument, attribute names, method names and tensor signatures as the reference class, so
``CanSwapPipeline`` (src/can_swap_pipeline_e2e.py:40,71-80,98,242-267) can use it unchanged; every tensor
operation is executed by the gfx950 HIP engine (libcanonswap_hip.so) -- there is no PyTorch or CPU fallback.
Stage attributes (``warping_module``, ``swap_module`` ...) are small callables bound to the C ABI.

The motion extractor M runs on the engine too (``get_kp_info``).  The ArcFace identity network
behind ``getid`` (can_swap_e2e.py:80-84,102-107: a pickled third-party module, ``pretrained_weights/arcface_checkpoint.tar``) is
outside the generator path: ``getid`` keeps the reference's arithmetic (nearest resize to 112x112, network, L2 normalisation)
around a network the caller injects (``id_net=`` or ``can_swapper.netArc = ...``) and raises if there is none.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np
import torch

from . import tail
from .engine import Engine


class _WarpingModule:
    """WarpingNetwork surface used by the pipeline (warping_network.py:49-111)."""

    def __init__(self, eng: Engine):
        self._e = eng

    def warp(self, feature_3d, kp_source, kp_driving):          # positional order as in the reference (:49)
        return self._e.warp(feature_3d, kp_source, kp_driving)

    def warp_out(self, out, occlusion_map=None):                # (:64)
        return self._e.warp_out(out, occlusion_map)

    def __call__(self, feature_3d, kp_driving, kp_source):      # forward(feature_3d, kp_driving, kp_source) (:83)
        return self._e.warp_forward(feature_3d, kp_driving=kp_driving, kp_source=kp_source)

    forward = __call__


class _Callable:
    def __init__(self, fn):
        self._fn = fn

    def __call__(self, *a, **k):
        return self._fn(*a, **k)

    forward = __call__


class can_swapper(object):
    """MI355X engine behind the reference's ``can_swapper`` interface."""

    def __init__(self, inference_cfg=None, state_dicts=None, max_batch: int = 8, id_net=None, latency_mode: bool = False,
                 packed_blobs=None):
        """packed_blobs: the result of pack.build_blobs() for the same state-dicts, when another process of this node already ran the
        load-time weight transform (engine.load_blobs)."""
        self.inference_cfg = inference_cfg
        self.device_id = getattr(inference_cfg, "device_id", 0)
        self.compile = False                      # torch.compile switch of the reference (:47,:74-77) has no meaning here
        if getattr(inference_cfg, "flag_force_cpu", False):
            raise RuntimeError("flag_force_cpu=True: this engine runs on an MI355X only (no CPU path)")
        self.device = "cuda:" + str(self.device_id)
        self.engine = Engine(self.device_id, max_batch=max_batch, latency_mode=latency_mode)
        self.appearance_feature_extractor = _Callable(self.engine.extract_feature_3d)
        self.warping_module = _WarpingModule(self.engine)
        self.spade_generator = _Callable(lambda feature: self.engine.spade_decode(feature))
        self.swap_module = _Callable(lambda f, source_id: self.engine.swap(f, source_id))
        self.refine_module = _Callable(self.engine.refine)
        self.motion_extractor = None
        # identity extractor (:80-84): the reference unpickles an ArcFace module; here it is injected, or loaded from the same
        # path when that file exists (torch.load of a pickled nn.Module needs the defining package importable, as in the reference)
        self.netArc = id_net
        arc = "pretrained_weights/arcface_checkpoint.tar"
        if self.netArc is None and os.path.exists(arc):
            self.netArc = torch.load(arc, map_location=torch.device("cpu"), weights_only=False).to(self.device).eval()
        if packed_blobs is not None:
            if any(k.startswith("M.") for k in packed_blobs):
                self.motion_extractor = _Callable(self.engine.motion_extract)
            self.engine.load_blobs(packed_blobs)
        elif state_dicts is not None:
            self.load_state_dicts(state_dicts)
        else:
            self.load_cpk()

    # ---- weights (can_swap_e2e.py:87-100)
    def load_cpk(self, combined_weights_path: str = "pretrained_weights/combined_weights.pth"):
        if os.path.exists(combined_weights_path):
            combined = torch.load(combined_weights_path, map_location=torch.device("cpu"))
            self.load_state_dicts(combined)

    def load_state_dicts(self, combined: dict):
        keys = ["appearance_feature_extractor", "warping_module", "spade_generator", "transfer", "refine"]
        if "motion_extractor" in combined:
            keys.append("motion_extractor")
            self.motion_extractor = _Callable(self.engine.motion_extract)
        self.engine.load_state_dicts({k: combined[k] for k in keys})

    # ---- small helpers kept for interface parity
    def inference_ctx(self):
        return contextlib.nullcontext()           # precision is fixed inside the engine (fp16 operands, fp32 accumulate)

    def update_config(self, user_args):
        for k, v in user_args.items():
            if hasattr(self.inference_cfg, k):
                setattr(self.inference_cfg, k, v)

    def getid(self, img):
        """can_swap_e2e.py:102-107: F.interpolate(img, (112, 112)) [nearest] -> netArc -> L2-normalised (B, 512) identity."""
        if self.netArc is None:
            raise RuntimeError("getid: no identity network (pass id_net= to can_swapper or set .netArc; the reference loads "
                               "pretrained_weights/arcface_checkpoint.tar, which is outside the generator path)")
        with torch.no_grad():
            x = torch.nn.functional.interpolate(img, size=(112, 112))
            out = self.netArc(x)
            idv = out[0] if isinstance(out, (tuple, list)) else out
            return torch.nn.functional.normalize(idv, p=2, dim=1)

    def get_kp_info(self, x, **kwargs):
        """can_swap_e2e.py:174-199: implicit key-point information of Bx3x256x256 images in [0,1]."""
        if self.motion_extractor is None:
            raise RuntimeError("get_kp_info: the loaded weights hold no 'motion_extractor' state-dict")
        kp_info = self.motion_extractor(x)
        if kwargs.get("flag_refine_info", True):
            bs = kp_info["kp"].shape[0]
            for k in ("pitch", "yaw", "roll"):
                kp_info[k] = headpose_pred_to_degree(kp_info[k])[:, None]
            kp_info["kp"] = kp_info["kp"].reshape(bs, -1, 3)
            kp_info["exp"] = kp_info["exp"].reshape(bs, -1, 3)
        return kp_info

    def get_pose_dct(self, kp_info: dict) -> dict:                                       # (:201-207)
        return {k: headpose_pred_to_degree(kp_info[k]).item() for k in ("pitch", "yaw", "roll")}

    def get_fs_and_kp_info(self, source_prepared, driving_first_frame):                  # (:209-226)
        s_info = self.get_kp_info(source_prepared, flag_refine_info=True)
        s_rot = get_rotation_matrix(s_info["pitch"], s_info["yaw"], s_info["roll"])
        d_info = self.get_kp_info(driving_first_frame, flag_refine_info=True)
        d_rot = get_rotation_matrix(d_info["pitch"], d_info["yaw"], d_info["roll"])
        return s_info, s_rot, self.extract_feature_3d(source_prepared), d_info, d_rot

    # ---- landmark ratios of the driving crops (:324-348; host-side numpy on 2-D landmarks, as in the reference)
    def calc_ratio(self, lmk_lst):
        return [eye_close_ratio(l[None]) for l in lmk_lst], [lip_close_ratio(l[None]) for l in lmk_lst]

    def calc_combined_eye_ratio(self, c_d_eyes_i, source_lmk):
        c_s = torch.from_numpy(eye_close_ratio(source_lmk[None])).float().to(self.device)
        c_d = torch.tensor([[float(c_d_eyes_i[0][0])]], device=self.device)
        return torch.cat([c_s, c_d], dim=1)

    def calc_combined_lip_ratio(self, c_d_lip_i, source_lmk):
        c_s = torch.from_numpy(lip_close_ratio(source_lmk[None])).float().to(self.device)
        c_d = torch.tensor([[float(np.asarray(c_d_lip_i[0]).reshape(-1)[0])]], device=self.device)
        return torch.cat([c_s, c_d], dim=1)

    # ---- data preparation (:126-163)
    def prepare_source(self, img: np.ndarray) -> torch.Tensor:
        hw = img.shape[-3:-1]
        if tuple(hw) == (512, 512) and img.dtype == np.uint8:
            # the cropper's 512x512 crop: its cv2.resize(..., (256, 256), INTER_AREA) (cropper.py:209) runs on the device; the
            # reference's own fallback here (cv2.resize, INTER_LINEAR, :131-132) gives the same 2x2 means at exactly half size
            return tail.prepare_crops(self.engine, img)
        if tuple(hw) != (256, 256):
            raise ValueError("prepare_source expects the 256x256 (or 512x512 uint8) crop produced by the cropper")
        if img.dtype == np.uint8:                 # upload 1 byte per sample and convert on the device (same arithmetic)
            return self.engine.unpack_u8(img[np.newaxis] if img.ndim == 3 else img)
        if img.ndim == 3:
            x = img[np.newaxis].astype(np.float32) / 255.
        elif img.ndim == 4:
            x = img.astype(np.float32) / 255.
        else:
            raise ValueError(f'img ndim should be 3 or 4: {img.ndim}')
        x = np.clip(x, 0, 1)
        return torch.from_numpy(x).permute(0, 3, 1, 2).to(self.device)

    def prepare_videos(self, imgs) -> torch.Tensor:
        if isinstance(imgs, list):
            _imgs = np.array(imgs)[..., np.newaxis]
        elif isinstance(imgs, np.ndarray):
            _imgs = imgs
        else:
            raise ValueError(f'imgs type error: {type(imgs)}')
        if _imgs.dtype == np.uint8 and _imgs.ndim == 5 and _imgs.shape[-1] == 1:
            return self.engine.unpack_u8(_imgs[..., 0]).unsqueeze(1)          # T x 1 x 3 x H x W
        y = np.clip(_imgs.astype(np.float32) / 255., 0, 1)
        return torch.from_numpy(y).permute(0, 4, 3, 1, 2).to(self.device)

    # ---- additions: image-space steps around the generator on the device
    def stream_videos(self, crops_u8, batch: int = None):
        """Iterator over (I_batch (n,3,256,256) fp32, (start, stop)): uint8 crops uploaded batch by batch, the next upload
        overlapping the current batch's work, instead of prepare_videos' whole-video residency."""
        return tail.FrameStreamer(self.engine, crops_u8, batch or self.engine.max_batch)

    def soft_mask(self, kernel_size=21, threshold=0.9, iterations=3):
        """SoftErosion as the pipeline builds it (can_swap_pipeline_e2e.py:42), on the engine."""
        return tail.SoftErosion(self.engine, kernel_size, threshold, iterations)

    def paste_back(self, img_crop, M_c2o, img_ori, mask_ori):                           # crop.py:523-529
        return tail.paste_back(self.engine, img_crop, M_c2o, img_ori, mask_ori)

    def prepare_paste_back(self, mask_crop, crop_M_c2o, dsize):                         # crop.py:515-521 (if_float=True)
        return tail.prepare_paste_back(self.engine, mask_crop, crop_M_c2o, dsize)

    def paste_back_fused(self, img_crop, mask_crop, M_c2o, img_ori):
        return tail.paste_back_fused(self.engine, img_crop, mask_crop, M_c2o, img_ori)

    # ---- stages
    def extract_feature_3d(self, x: torch.Tensor) -> torch.Tensor:                      # (:165-172)
        return self.engine.extract_feature_3d(x)

    def swap(self, feature_3d, source_id):                                               # (:109-111)
        return self.engine.swap(feature_3d, source_id)

    def transform_keypoint(self, kp_info: dict):                                         # (:228-256); 21x3 host-size math
        kp = kp_info['kp']
        pitch, yaw, roll = (headpose_pred_to_degree(kp_info[k]) for k in ('pitch', 'yaw', 'roll'))
        bs = kp.shape[0]
        num_kp = kp.shape[1] // 3 if kp.ndim == 2 else kp.shape[1]
        rot_mat = get_rotation_matrix(pitch, yaw, roll)
        kp_t = kp.view(bs, num_kp, 3) @ rot_mat + kp_info['exp'].view(bs, num_kp, 3)
        kp_t = kp_t * kp_info['scale'][..., None]
        kp_t[:, :, 0:2] += kp_info['t'][:, None, 0:2]
        return kp_t

    def warp_decode(self, feature_3d, kp_source, kp_driving) -> dict:                    # (:286-308)
        ret = self.engine.warp_forward(feature_3d, kp_driving=kp_driving, kp_source=kp_source)
        ret['out'] = self.engine.spade_decode(ret['out'])
        return ret

    def conv_decode(self, out, occlusion_map=None) -> torch.Tensor:                      # (:309-312)
        return self.engine.spade_decode(self.engine.warp_out(out, occlusion_map))

    def parse_output(self, out: torch.Tensor) -> np.ndarray:                             # (:314-322), packed on device
        return self.engine.pack_u8(out).cpu().numpy()

    # ---- addition: the whole loop body of can_swap_pipeline_e2e.py:242-263 for a batch of frames
    def swap_frames(self, I_s, x_t, x_can, source_id, debug=False, want_u8=False):
        return self.engine.swap_frames(I_s, x_t, x_can, source_id, want_f32=True, want_u8=want_u8, debug=debug)

    # ---- addition: the per-frame body of can_swap_pipeline_v2i.py:311-312 (warp_decode of one swapped canonical volume
    # under the driving key-points of B frames)
    def animate_frames(self, f_swap_can, x_swap, x_t, want_u8=False):
        return self.engine.animate_frames(f_swap_can, x_swap, x_t, want_f32=True, want_u8=want_u8)


def _distance_ratio(lmk, a, b, c, d, eps=1e-6):
    """|lmk[a]-lmk[b]| / (|lmk[c]-lmk[d]| + eps) per row of lmk (B x n_points x 2) -> (B, 1)  (retargeting_utils.py:9-11)."""
    num = np.linalg.norm(lmk[:, a] - lmk[:, b], axis=1, keepdims=True)
    den = np.linalg.norm(lmk[:, c] - lmk[:, d], axis=1, keepdims=True)
    return num / (den + eps)


def eye_close_ratio(lmk):
    """Left / right eye opening over eye width, (B, 2)  (retargeting_utils.py:14-20 without a target ratio)."""
    return np.concatenate([_distance_ratio(lmk, 6, 18, 0, 12), _distance_ratio(lmk, 30, 42, 24, 36)], axis=1)


def lip_close_ratio(lmk):
    """Lip opening over mouth width, (B, 1)  (retargeting_utils.py:23-24)."""
    return _distance_ratio(lmk, 90, 102, 48, 66)


def headpose_pred_to_degree(pred):
    """src/utils/camera.py:14-28."""
    if pred.ndim > 1 and pred.shape[1] == 66:
        idx = torch.arange(66, dtype=torch.float32, device=pred.device)
        return torch.sum(torch.softmax(pred, dim=1) * idx, dim=1) * 3 - 97.5
    return pred


def get_rotation_matrix(pitch_, yaw_, roll_):
    """src/utils/camera.py:31-73 (degrees in; returns (Rz Ry Rx)^T)."""
    x, y, z = [(a / 180 * np.pi) for a in (pitch_, yaw_, roll_)]
    x, y, z = [a.unsqueeze(1) if a.ndim == 1 else a for a in (x, y, z)]
    bs = x.shape[0]
    one, zero = torch.ones(bs, 1, device=x.device), torch.zeros(bs, 1, device=x.device)
    rx = torch.cat([one, zero, zero, zero, torch.cos(x), -torch.sin(x), zero, torch.sin(x), torch.cos(x)], 1).reshape(bs, 3, 3)
    ry = torch.cat([torch.cos(y), zero, torch.sin(y), zero, one, zero, -torch.sin(y), zero, torch.cos(y)], 1).reshape(bs, 3, 3)
    rz = torch.cat([torch.cos(z), -torch.sin(z), zero, torch.sin(z), torch.cos(z), zero, zero, zero, one], 1).reshape(bs, 3, 3)
    return (rz @ ry @ rx).permute(0, 2, 1)
