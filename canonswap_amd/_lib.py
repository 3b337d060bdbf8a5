"""ctypes binding of libcanonswap_hip.so (C ABI: include/canonswap_hip.h) and its in-tree build.

The library is the product: if it cannot be loaded, every entry point raises -- there is no CPU or
PyTorch fallback (the CPU restatement under oracle/ is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(HERE, "csrc", f) for f in ("conv_igemm.hip", "conv_halo.hip", "kernels.hip", "motion.hip", "engine.hip")]
HEADERS = [os.path.join(HERE, "csrc", "common.h"), os.path.join(HERE, "csrc", "conv_epilogue.h"), os.path.join(os.path.dirname(HERE), "include", "canonswap_hip.h")]
LIB_PATH = os.path.join(HERE, "libcanonswap_hip.so")
ABI_SYMBOLS = [
    "cs_create", "cs_destroy", "cs_last_error", "cs_abi_version", "cs_upload", "cs_finalize_weights", "cs_set_identity",
    "cs_extract_feature_3d", "cs_warp", "cs_warp_out", "cs_swap", "cs_refine", "cs_warp_forward", "cs_spade_decode",
    "cs_pack_u8", "cs_unpack_u8", "cs_motion_extract", "cs_swap_frames", "cs_animate_frames", "cs_profile_begin", "cs_profile_end", "cs_op_conv", "cs_op_grid_sample3d",
    "cs_op_chan_stats", "cs_op_chan_stats_partial_floats",
]


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 into canonswap_amd/libcanonswap_hip.so (hipcc cross-compiles without a GPU)."""
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(p) for p in SOURCES + HEADERS)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", *SOURCES, "-o", LIB_PATH]
    subprocess.run(cmd, check=True)
    return LIB_PATH


class ConvDesc(C.Structure):
    """Mirror of cs_conv_desc (include/canonswap_hip.h)."""
    _fields_ = [
        ("in_", C.c_void_p), ("in_sN", C.c_long), ("in_sD", C.c_long), ("in_sH", C.c_long), ("in_sW", C.c_long),
        ("N", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("up_shift", C.c_int),
        ("KD", C.c_int), ("KH", C.c_int), ("KW", C.c_int),
        ("wgt", C.c_void_p), ("Cout_pad", C.c_int), ("Cout", C.c_int),
        ("bias", C.c_void_p), ("bias2", C.c_void_p),
        ("act0", C.c_int), ("slope0", C.c_float),
        ("res", C.c_void_p), ("res_f32", C.c_int), ("res_shift", C.c_int),
        ("res_sN", C.c_long), ("res_sD", C.c_long), ("res_sH", C.c_long), ("res_sW", C.c_long),
        ("pixscale", C.c_void_p), ("ps_stride", C.c_int),
        ("out0", C.c_void_p), ("out0_f32", C.c_int),
        ("out0_sN", C.c_long), ("out0_sD", C.c_long), ("out0_sH", C.c_long), ("out0_sW", C.c_long),
        ("s2", C.c_void_p), ("t2", C.c_void_p), ("act1", C.c_int), ("slope1", C.c_float),
        ("out1", C.c_void_p), ("out1_sN", C.c_long), ("out1_sD", C.c_long), ("out1_sH", C.c_long), ("out1_sW", C.c_long),
        ("stats", C.c_void_p),
        ("mode", C.c_int), ("cfg", C.c_int), ("tile_w", C.c_int), ("tile_h", C.c_int), ("ck", C.c_int),
    ]


_lib = None


def load():
    """Load the shared library; raises RuntimeError if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine has not been built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(needs hipcc); canonswap_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.cs_last_error.restype = C.c_char_p
    lib.cs_create.argtypes = [ci, ci, C.POINTER(vp)]
    lib.cs_destroy.argtypes = [vp]
    lib.cs_destroy.restype = None
    lib.cs_upload.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.cs_finalize_weights.argtypes = [vp]
    lib.cs_set_identity.argtypes = [vp, ci, vp, vp]
    lib.cs_extract_feature_3d.argtypes = [vp, ci, vp, vp, vp]
    lib.cs_warp.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
    lib.cs_warp_out.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.cs_swap.argtypes = [vp, ci, ci, vp, vp, vp]
    lib.cs_refine.argtypes = [vp, ci, vp, vp, vp]
    lib.cs_warp_forward.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.cs_spade_decode.argtypes = [vp, ci, vp, vp, vp]
    lib.cs_pack_u8.argtypes = [vp, ci, vp, vp, ci, ci, vp]
    lib.cs_unpack_u8.argtypes = [vp, ci, vp, vp, ci, ci, vp]
    lib.cs_motion_extract.argtypes = [vp, ci, vp, vp, vp]
    lib.cs_animate_frames.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, vp, vp]
    lib.cs_swap_frames.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cs_profile_begin.argtypes = [vp]
    lib.cs_profile_end.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]
    lib.cs_op_conv.argtypes = [C.POINTER(ConvDesc), vp]
    lib.cs_op_grid_sample3d.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.cs_op_chan_stats.argtypes = [vp, ci, ci, C.c_long, ci, cf, vp, vp, vp]
    lib.cs_op_chan_stats_partial_floats.argtypes = [ci, C.c_long, ci]
    lib.cs_op_chan_stats_partial_floats.restype = C.c_long
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {load().cs_last_error().decode(errors='replace')}")
