"""Thin Python handle over the C ABI (include/canonswap_hip.h).

PyTorch is plumbing here: it owns the caller-visible device tensors and the HIP stream; every operation on
the generator path is a HIP kernel launched by libcanonswap_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, pack


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class Engine:
    """One engine per device; not thread-safe (same contract as the C ABI)."""

    def __init__(self, device_id: int = 0, max_batch: int = 8, latency_mode: bool = False):
        if not torch.cuda.is_available():
            raise RuntimeError("canonswap_amd.Engine needs a ROCm device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(f"cuda:{device_id}")
        self.max_batch = max_batch
        h = C.c_void_p()
        _lib.check(self.lib.cs_create(device_id, max_batch, C.byref(h)), "cs_create")
        self.h = h
        self.latency_mode = bool(latency_mode)
        if latency_mode:         # BASELINE configs[1]: the one-frame forms (conv_lat, split-K, 2-row volume segments; see cs_set_latency_mode)
            _lib.check(self.lib.cs_set_latency_mode(h, 1), "cs_set_latency_mode")
        self._ids = []            # resident identities: [slot, device copy (512,), last tensor seen, its _version, use tick]
        self._tick = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.cs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- weights
    def load_state_dicts(self, state_dicts: dict):
        """Ingest the reference's six state-dicts (src/can_swap_e2e.py:87-100 key layout)."""
        self.load_blobs(pack.build_blobs(state_dicts))

    def load_blobs(self, blobs: dict):
        """Upload packed weight blobs (pack.build_blobs): the load-time transform runs once; N ranks of one node can share its result
        (bench.py packs on rank 0 and hands the blobs over through /dev/shm)."""
        for name, arr in blobs.items():
            arr = np.ascontiguousarray(arr)
            _lib.check(self.lib.cs_upload(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p), arr.nbytes), f"cs_upload({name})")
        _lib.check(self.lib.cs_finalize_weights(self.h), "cs_finalize_weights")
        self._ids = []

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _in(self, t, shape_tail=None):
        if not isinstance(t, torch.Tensor):
            raise TypeError("expected a torch.Tensor")
        if t.device != self.device:
            t = t.to(self.device)
        t = t.contiguous().float()
        if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
            raise ValueError(f"expected shape (B, {', '.join(map(str, shape_tail))}), got {tuple(t.shape)}")
        if t.shape[0] < 1 or t.shape[0] > self.max_batch:
            raise ValueError(f"batch {t.shape[0]} outside [1, {self.max_batch}]")
        return t

    @staticmethod
    def _same_batch(*ts):
        n = {int(t.shape[0]) for t in ts if t is not None}
        if len(n) != 1:
            raise ValueError(f"inputs disagree on the batch size: {sorted(n)}")

    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _out(self, t, shape, dtype):
        """A caller-provided output buffer must be exactly what the C ABI will write: device, dtype, shape, contiguous."""
        if t is None:
            return self._new(*shape, dtype=dtype)
        if not isinstance(t, torch.Tensor) or t.device != self.device or t.dtype != dtype or tuple(t.shape) != tuple(shape) \
                or not t.is_contiguous():
            raise ValueError(f"output buffer must be a contiguous {dtype} tensor of shape {tuple(shape)} on {self.device}")
        return t

    def set_identity(self, source_id: torch.Tensor, slot: int = 0):
        """Per-identity precompute of T's modulated weights (adaptive_modulate.py:148-155) into an identity slot.
        source_id: (1,512) or (512,)."""
        sid = source_id.detach().to(self.device).float().reshape(-1, 512)
        if sid.shape[0] != 1:
            raise ValueError("set_identity takes one identity; pass several rows to swap()/swap_frames() instead")
        sid = sid[0].contiguous().clone()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cs_set_identity(self.h, slot, _ptr(sid), self._stream()), "cs_set_identity")
        self._ids = [r for r in self._ids if r[0] != slot]
        self._multi = None
        self._tick += 1
        self._ids.append([slot, sid, None, None, self._tick])
        return slot

    def _slot_of_row(self, row: torch.Tensor, src=None):
        """Slot that holds identity `row` (512 floats on the device), computing it into the least recently used slot if needed.
        Identities are compared by VALUE (2 KB), never by address: the allocator reuses addresses of freed tensors."""
        self._tick += 1
        for r in self._ids:
            if torch.equal(r[1], row):
                r[4] = self._tick
                return r[0]
        used = {r[0] for r in self._ids}
        free = [k for k in range(_lib.MAX_IDENTITY_SLOTS) if k not in used]
        slot = free[0] if free else min(self._ids, key=lambda r: r[4])[0]
        return self.set_identity(row, slot)

    def identity_slots(self, source_id: torch.Tensor, B: int):
        """Per-sample identity slots for a (1,512), (512,) or (B,512) source_id (the reference's per-sample dlatents)."""
        # fast path of the per-frame loop: the very same tensor object, unmodified since it was resolved (a strong reference
        # is kept, so its address cannot have been recycled)
        for r in self._ids:
            if r[2] is source_id and r[3] == source_id._version and source_id.numel() == 512:
                self._tick += 1
                r[4] = self._tick
                return [r[0]] * B
        mc = getattr(self, "_multi", None)          # the same (B, 512) tensor object as last time, unmodified: slots are known
        if mc is not None and mc[0] is source_id and mc[1] == source_id._version and len(mc[2]) >= B and \
                all(any(r[0] == k for r in self._ids) for k in set(mc[2][:B])):
            return mc[2][:B]
        sid = source_id.detach().to(self.device).float().reshape(-1, 512)
        if sid.shape[0] not in (1, B):
            raise ValueError(f"source_id holds {sid.shape[0]} identities for a batch of {B}")
        uniq, inv = torch.unique(sid, dim=0, return_inverse=True)
        if uniq.shape[0] > _lib.MAX_IDENTITY_SLOTS:
            raise ValueError(f"{uniq.shape[0]} distinct identities in one batch exceed the {_lib.MAX_IDENTITY_SLOTS} identity slots")
        slot_of = [self._slot_of_row(uniq[k].contiguous()) for k in range(uniq.shape[0])]
        if len(set(slot_of)) != len(slot_of):       # an eviction inside this very batch: cannot happen with <= MAX slots
            raise RuntimeError("identity slot assignment collided")
        slots = [slot_of[int(k)] for k in inv.tolist()]
        if sid.shape[0] == B and B > 1:
            self._multi = (source_id, source_id._version, slots)
        if sid.shape[0] == 1:
            slots = slots * B
            for r in self._ids:
                if r[0] == slots[0]:
                    r[2], r[3] = source_id, source_id._version
        return slots

    def ensure_identity(self, source_id: torch.Tensor):
        return self.identity_slots(source_id, 1)[0]

    def _default_slots(self, B):
        if not self._ids:
            raise RuntimeError("no identity has been set (cs_set_identity): pass source_id or call set_identity first")
        return [max(self._ids, key=lambda r: r[4])[0]] * B

    @staticmethod
    def _slot_array(slots):
        return (C.c_int * len(slots))(*slots)

    # ---------------------------------------------------------------- stages
    def extract_feature_3d(self, img):
        img = self._in(img, (3, 256, 256))
        out = self._new(img.shape[0], 32, 16, 64, 64)
        _lib.check(self.lib.cs_extract_feature_3d(self.h, img.shape[0], _ptr(img), _ptr(out), self._stream()), "cs_extract_feature_3d")
        return out

    def warp(self, f, kp_source, kp_driving):
        f = self._in(f, (32, 16, 64, 64)); ks = self._in(kp_source, (21, 3)); kd = self._in(kp_driving, (21, 3))
        self._same_batch(f, ks, kd)
        B = f.shape[0]
        out, occ = self._new(B, 32, 16, 64, 64), self._new(B, 1, 64, 64)
        _lib.check(self.lib.cs_warp(self.h, B, _ptr(f), _ptr(ks), _ptr(kd), _ptr(out), _ptr(occ), self._stream()), "cs_warp")
        return out, occ

    def warp_out(self, f, occ=None):
        f = self._in(f, (32, 16, 64, 64))
        B = f.shape[0]
        occ = None if occ is None else self._in(occ, (1, 64, 64))
        self._same_batch(f, occ)
        seg = self._new(B, 256, 64, 64)
        _lib.check(self.lib.cs_warp_out(self.h, B, _ptr(f), _ptr(occ), _ptr(seg), self._stream()), "cs_warp_out")
        return seg

    def swap(self, f, source_id=None):
        """transfer_model2.forward(x, dlatents): source_id (1,512) or one row per sample (adaptive_modulate.py:157-167)."""
        f = self._in(f, (32, 16, 64, 64))
        B = f.shape[0]
        slots = self.identity_slots(source_id, B) if source_id is not None else self._default_slots(B)
        out = self._new(*f.shape)
        _lib.check(self.lib.cs_swap_ids(self.h, self._slot_array(slots), B, _ptr(f), _ptr(out), self._stream()), "cs_swap_ids")
        return out

    def refine(self, f):
        f = self._in(f, (32, 16, 64, 64))
        out = self._new(*f.shape)
        _lib.check(self.lib.cs_refine(self.h, f.shape[0], _ptr(f), _ptr(out), self._stream()), "cs_refine")
        return out

    def warp_forward(self, f, kp_driving, kp_source):
        f = self._in(f, (32, 16, 64, 64)); kd = self._in(kp_driving, (21, 3)); ks = self._in(kp_source, (21, 3))
        self._same_batch(f, ks, kd)
        B = f.shape[0]
        occ, deform, seg = self._new(B, 1, 64, 64), self._new(B, 16, 64, 64, 3), self._new(B, 256, 64, 64)
        _lib.check(self.lib.cs_warp_forward(self.h, B, _ptr(f), _ptr(kd), _ptr(ks), _ptr(occ), _ptr(deform), _ptr(seg),
                                            self._stream()), "cs_warp_forward")
        return {"occlusion_map": occ, "deformation": deform, "out": seg}

    def spade_decode(self, seg):
        seg = self._in(seg, (256, 64, 64))
        img = self._new(seg.shape[0], 3, 512, 512)
        _lib.check(self.lib.cs_spade_decode(self.h, seg.shape[0], _ptr(seg), _ptr(img), self._stream()), "cs_spade_decode")
        return img

    def motion_extract(self, img):
        """MotionExtractor.forward (motion_extractor.py:33-35): Bx3x256x256 in [0,1] -> dict of raw head outputs."""
        img = self._in(img, (3, 256, 256))
        out = self._new(img.shape[0], 328)
        _lib.check(self.lib.cs_motion_extract(self.h, img.shape[0], _ptr(img), _ptr(out), self._stream()), "cs_motion_extract")
        res, o = {}, 0
        for k, n in pack.M_HEADS:
            res[k] = out[:, o:o + n].contiguous()
            o += n
        return res

    def motion_keypoints(self, raw, want_rot=False, out=None):
        """Key-points from motion_extract's raw head outputs (B, 328), on the device: get_kp_info's refinement + get_rotation_matrix +
        transform_keypoint (can_swap_e2e.py:192-197, 228-256; camera.py:14-73) -> x_t (B,21,3), x_can = scale * kp (B,21,3)
        (can_swap_pipeline_e2e.py:243)[, R (B,3,3)]."""
        raw = self._in(raw, (328,))
        B = raw.shape[0]
        x_t, x_can = (self._out(o, (B, 21, 3), torch.float32) for o in (out if out is not None else (None, None)))
        rot = self._new(B, 3, 3) if want_rot else None
        _lib.check(self.lib.cs_motion_keypoints(self.h, B, _ptr(raw), _ptr(x_t), _ptr(x_can), _ptr(rot), self._stream()), "cs_motion_keypoints")
        return (x_t, x_can, rot) if want_rot else (x_t, x_can)

    def motion_extract_raw(self, img, out=None):
        """cs_motion_extract without the per-head split: (B, 328) fp32."""
        img = self._in(img, (3, 256, 256))
        out = self._out(out, (img.shape[0], 328), torch.float32)
        _lib.check(self.lib.cs_motion_extract(self.h, img.shape[0], _ptr(img), _ptr(out), self._stream()), "cs_motion_extract")
        return out

    def pack_u8(self, img):
        img = self._in(img)
        B, _, H, W = img.shape
        out = self._new(B, H, W, 3, dtype=torch.uint8)
        _lib.check(self.lib.cs_pack_u8(self.h, B, _ptr(img), _ptr(out), H, W, self._stream()), "cs_pack_u8")
        return out

    def unpack_u8(self, img_u8):
        """BxHxWx3 uint8 (host or device) -> Bx3xHxW fp32 in [0,1] on the device (prepare_source / prepare_videos arithmetic)."""
        t = torch.as_tensor(img_u8)
        if t.dtype != torch.uint8 or t.ndim != 4 or t.shape[3] != 3:
            raise ValueError("expected a BxHxWx3 uint8 array")
        t = t.to(self.device).contiguous()
        B, H, W, _ = t.shape
        out = self._new(B, 3, H, W)
        _lib.check(self.lib.cs_unpack_u8(self.h, B, _ptr(t), _ptr(out), H, W, self._stream()), "cs_unpack_u8")
        return out

    def swap_frames(self, img, x_t, x_can, source_id=None, want_f32=True, want_u8=False, debug=False,
                    out_f32=None, out_u8=None, slots=None):
        """Whole loop body of can_swap_pipeline_e2e.py:242-263 for B frames, on device.
        slots: identity slot per frame (ints, set with set_identity) instead of `source_id` rows - no identity lookup at all
        (multi-stream callers that placed their identities themselves)."""
        img = self._in(img, (3, 256, 256)); x_t = self._in(x_t, (21, 3)); x_can = self._in(x_can, (21, 3))
        self._same_batch(img, x_t, x_can)
        B = img.shape[0]
        if slots is not None:
            slots = [int(k) for k in slots]
            if len(slots) != B or source_id is not None:
                raise ValueError("slots: one identity slot per frame, and no source_id")
        else:
            slots = self.identity_slots(source_id, B) if source_id is not None else self._default_slots(B)
        out_f32 = self._out(out_f32, (B, 3, 512, 512), torch.float32) if (want_f32 or out_f32 is not None) else None
        out_u8 = self._out(out_u8, (B, 512, 512, 3), torch.uint8) if (want_u8 or out_u8 is not None) else None
        rec = self._new(B, 3, 512, 512) if debug else None
        swp = self._new(B, 3, 512, 512) if debug else None
        _lib.check(self.lib.cs_swap_frames_ids(self.h, self._slot_array(slots), B, _ptr(img), _ptr(x_t), _ptr(x_can), _ptr(out_f32),
                                               _ptr(out_u8), _ptr(rec), _ptr(swp), self._stream()), "cs_swap_frames_ids")
        res = {"out": out_f32, "out_u8": out_u8}
        if debug:
            res.update(rec_can=rec, swap_can=swp)
        return res

    def animate_frames(self, f, kp_source, kp_driving, want_f32=True, want_u8=False, out_f32=None, out_u8=None):
        """Per-frame body of can_swap_pipeline_v2i.py:311-312 for B driving frames: one (or B) feature volume(s) f and source
        key-point set(s), B driving key-point sets -> Bx3x512x512."""
        kp_driving = self._in(kp_driving, (21, 3))
        B = kp_driving.shape[0]
        f = self._in(f, (32, 16, 64, 64)); kp_source = self._in(kp_source, (21, 3))
        if f.shape[0] not in (1, B) or kp_source.shape[0] not in (1, B):
            raise ValueError("f and kp_source must hold 1 or B entries")
        out_f32 = self._out(out_f32, (B, 3, 512, 512), torch.float32) if (want_f32 or out_f32 is not None) else None
        out_u8 = self._out(out_u8, (B, 512, 512, 3), torch.uint8) if (want_u8 or out_u8 is not None) else None
        _lib.check(self.lib.cs_animate_frames(self.h, B, _ptr(f), f.shape[0], _ptr(kp_source), kp_source.shape[0], _ptr(kp_driving),
                                              _ptr(out_f32), _ptr(out_u8), self._stream()), "cs_animate_frames")
        return {"out": out_f32, "out_u8": out_u8}

    # ---------------------------------------------------------------- measurement
    def profile_begin(self):
        _lib.check(self.lib.cs_profile_begin(self.h), "cs_profile_begin")

    def profile_end(self):
        ms, cnt, fl = (C.c_double * 3)(), (C.c_long * 3)(), C.c_double()
        _lib.check(self.lib.cs_profile_end(self.h, ms, cnt, C.byref(fl)), "cs_profile_end")
        ex = C.c_double()
        _lib.check(self.lib.cs_profile_exec_flops(self.h, C.byref(ex)), "cs_profile_exec_flops")
        return {"exec_flops": ex.value,"conv_ms": ms[0], "other_ms": ms[1], "warp_ms": ms[2], "conv_launches": cnt[0], "other_launches": cnt[1],
                "warp_launches": cnt[2], "conv_flops": fl.value}
