"""CPU-side checks: weight packer, C-ABI surface, host helpers (no GPU compute)."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from canonswap_amd import pack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pack_conv_roundtrip_and_kstep_order():
    r = np.random.Generator(np.random.PCG64(3))
    w = r.standard_normal((22, 142, 7, 7, 7)).astype(np.float32)
    p = pack.pack_conv(w, 32)
    assert p.shape == (5 * 343, 32, 32) and p.dtype == np.float16
    assert np.array_equal(pack.unpack_conv(p, 22, 142, 7, 7, 7), w.astype(np.float16).astype(np.float32))
    # kstep = ((chunk*KD+kd)*KH+kh)*KW+kw, in-channel = chunk*32 + kk
    chunk, kd, kh, kw, row, kk = 3, 2, 5, 1, 7, 9
    assert p[((chunk * 7 + kd) * 7 + kh) * 7 + kw, row, kk] == np.float16(w[row, chunk * 32 + kk, kd, kh, kw])
    assert np.all(p[:, 22:, :] == 0) and np.all(p[4 * 343:, :, 14:] == 0)      # padded rows / channels are zero


def test_interleave16():
    a = np.arange(32)[:, None] * np.ones((1, 3)); b = -a
    z = pack.interleave16(a, b)
    assert z.shape == (64, 3)
    assert np.array_equal(z[:16], a[:16]) and np.array_equal(z[16:32], b[:16]) and np.array_equal(z[32:48], a[16:])


def test_bn_and_spectral_folding_match_oracle(state_dicts_np, state_dicts):
    from oracle import canonswap_ref as O
    sd, sdt = state_dicts_np["appearance_feature_extractor"], state_dicts["appearance_feature_extractor"]
    s, t = pack.bn_affine(sd, "down_blocks.0.norm")
    w, b = pack.fold_conv_bn(sd["down_blocks.0.conv.weight"], sd["down_blocks.0.conv.bias"], s, t)
    x = torch.randn(1, 64, 8, 8)
    ref = O.bn_eval(O.conv(x, sdt, "down_blocks.0.conv", 1), sdt, "down_blocks.0.norm")
    got = F.conv2d(x, torch.from_numpy(w).float(), torch.from_numpy(b).float(), padding=1)
    assert (ref - got).abs().max() < 1e-4
    g, gt = state_dicts_np["spade_generator"], state_dicts["spade_generator"]
    assert np.abs(pack.spectral_weight(g, "up_1.conv_1") - O.spectral_weight(gt, "up_1.conv_1").numpy()).max() < 1e-5


def test_view_permutation_is_the_reference_view():
    """Memory channel jm = d*32 + c of the [H][W][D][C] volume is reference channel c*16 + d of view(bs, c*d, h, w)."""
    v = torch.arange(32 * 16).view(32, 16)                       # value = reference flat channel index c*16+d
    mem = v.t().reshape(-1)                                      # [d][c] order
    assert np.array_equal(mem.numpy(), pack.MEM2REF)


def test_build_blobs_names_and_sizes(state_dicts_np):
    blobs = pack.build_blobs(state_dicts_np)
    assert blobs["T.b6.c2.w"].shape == (144, 1024, 32) and blobs["T.b0.c1.raw"].shape == (512, 9, 512)
    assert blobs["W.maskp.w"].shape == (5 * 49, 160, 32) and blobs["W.occp.w"].shape == (80 * 7, 32, 32) and blobs["W.occ49.w"].shape == (80, 64, 32)
    assert blobs["G.shared64.w"].shape == (72, 1536, 32) and blobs["G.up1.n1.w"].shape == (36, 128, 32)
    assert all(b.flags["C_CONTIGUOUS"] for b in blobs.values())
    # every blob the engine asks for by literal name exists (names built with snprintf are covered on the GPU)
    src = open(os.path.join(ROOT, "canonswap_amd", "csrc", "engine.hip")).read()
    for name in re.findall(r'"((?:F|W|T|R|G)\.[A-Za-z0-9_.]+\.(?:w|b))"', src):
        assert name in blobs, name


def test_c_abi_library_exports_every_declared_symbol():
    from canonswap_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "canonswap_hip.h")).read()
    declared = set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.cs_abi_version() == _lib.ABI_VERSION == 4


def test_isa_check_of_the_hand_counted_weight_rings():
    """The objects the shipped library was linked from (canonswap_amd/build/): every untracked ring load is covered by a counted wait that
    is deep enough for it before an MFMA reads it (the check marks a register waited only when its load is older than the
    N youngest VMEM ops of `s_waitcnt vmcnt(N)`)."""
    from canonswap_amd import _lib
    if not os.path.isdir(_lib.OBJ_DIR) or not os.path.exists(os.path.join(_lib.OBJ_DIR, "conv_wide.o")):
        pytest.skip("no object files next to the library (prebuilt .so only)")
    seen = _lib.isa_check()
    assert len(seen) >= 70 and any("conv_wide_kernel" in k for k in seen)


def test_isa_check_rejects_a_wait_that_is_too_shallow(tmp_path, monkeypatch):
    """A counted wait whose immediate leaves the consumed load among the outstanding ones must fail the check."""
    from canonswap_amd import _lib
    asm = """0000000000001000 <conv_wide_kernel_fake>:
	global_load_dwordx4 v[10:13], v[0:1], off
	global_load_dwordx4 v[14:17], v[0:1], off
	s_waitcnt vmcnt(%d)
	v_mfma_f32_16x16x32_f16 a[0:3], v[10:13], v[20:23], a[0:3]
	v_mfma_f32_16x16x32_f16 a[0:3], v[14:17], v[20:23], a[0:3]
	s_endpgm
"""
    import subprocess as sp

    def fake_run_factory(n):
        def fake_run(cmd, **kw):
            class R:
                stdout = asm % n
            return R()
        return fake_run

    (tmp_path / "fake.o").write_bytes(b"")
    monkeypatch.setattr(_lib.subprocess, "run", fake_run_factory(0))
    assert _lib._isa_check_ring(str(tmp_path), "objdump", "fake.o", r"conv_wide_kernel_fake", 2) == {"conv_wide_kernel_fake": 2}
    monkeypatch.setattr(_lib.subprocess, "run", fake_run_factory(1))       # vmcnt(1): the second load may still be in flight
    with pytest.raises(RuntimeError, match="no vmcnt wait"):
        _lib._isa_check_ring(str(tmp_path), "objdump", "fake.o", r"conv_wide_kernel_fake", 2)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_fails_loudly_without_gpu():
    from canonswap_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(0)


def test_keypoint_helpers_match_golden(golden):
    from canonswap_amd.can_swap_e2e import get_rotation_matrix, headpose_pred_to_degree
    g = golden("unit_vectors.npz")
    pyr = torch.from_numpy(g["pyr"])
    assert np.abs(get_rotation_matrix(pyr[:, 0], pyr[:, 1], pyr[:, 2]).numpy() - g["rot"]).max() < 1e-6
    assert np.abs(headpose_pred_to_degree(torch.from_numpy(g["bins"])).numpy() - g["deg"]).max() < 1e-4


def test_landmark_ratios_match_reference_vectors(golden):
    """calc_ratio's helpers (can_swap_e2e.py:324-348) against values computed by the reference's retargeting_utils."""
    from canonswap_amd import can_swap_e2e as H
    g = golden("unit_vectors.npz")
    assert np.allclose(H.eye_close_ratio(g["lmk"]), g["eye_ratio"], rtol=0, atol=1e-6)
    assert np.allclose(H.lip_close_ratio(g["lmk"]), g["lip_ratio"], rtol=0, atol=1e-6)
    assert H.eye_close_ratio(g["lmk"][:1]).shape == (1, 2) and H.lip_close_ratio(g["lmk"][:1]).shape == (1, 1)


@pytest.mark.parametrize("s", [2, 4])
def test_upsampled_conv_phase_decomposition(s):
    """pack.upsampled_conv_phases: a 3x3 conv on a nearest-up-sampled map == per-row-phase convs on the source grid."""
    r = np.random.Generator(np.random.PCG64(s))
    w, x = r.standard_normal((5, 3, 3, 3)), r.standard_normal((2, 3, 8, 8))
    ref = F.conv2d(F.interpolate(torch.from_numpy(x), scale_factor=s, mode="nearest"), torch.from_numpy(w), padding=1).numpy()
    out = np.zeros_like(ref)
    phases = pack.upsampled_conv_phases(w, s)
    assert len(phases) == (4 if s == 2 else 12)
    for a, b0, nb, kh, ph, kw, pw, wg in phases:
        assert wg.shape == (nb * 5, 3, kh, kw)
        y = F.conv2d(F.pad(torch.from_numpy(x), (pw, kw - 1 - pw, ph, kh - 1 - ph)), torch.from_numpy(wg)).numpy()
        for k in range(nb):
            out[:, :, a::s, b0 + k::s] = y[:, k * 5:(k + 1) * 5]
    assert np.abs(out - ref).max() < 1e-12
    if s == 4:      # what run_G's de-duplication rests on: rows 4i+1 / 4i+2 and the two middle columns carry the same weights, bit for bit
        by = {(a, b0): wg for a, b0, nb, kh, ph, kw, pw, wg in phases}
        for b0 in (0, 1, 3):
            assert np.array_equal(by[(1, b0)], by[(2, b0)])
        for a in range(4):
            assert np.array_equal(by[(a, 1)][:5], by[(a, 1)][5:])


def test_upsampled_conv3d_phase_decomposition():
    """pack.upsampled_conv3d_phases: 3x3x3 conv after a nearest (1,2,2) up-sampling == four 3x2x2 convs on the source grid."""
    r = np.random.Generator(np.random.PCG64(9))
    w, x = r.standard_normal((5, 3, 3, 3, 3)), r.standard_normal((1, 3, 4, 6, 6))
    ref = F.conv3d(F.interpolate(torch.from_numpy(x), scale_factor=(1, 2, 2), mode="nearest"), torch.from_numpy(w), padding=1).numpy()
    out = np.zeros_like(ref)
    for (a, b), (ph, pw, wab) in pack.upsampled_conv3d_phases(w).items():
        assert (ph, pw) == (int(a == 0), int(b == 0)) and wab.shape == (5, 3, 3, 2, 2)
        xp = F.pad(torch.from_numpy(x), (pw, 1 - pw, ph, 1 - ph, 1, 1))
        out[:, :, :, a::2, b::2] = F.conv3d(xp, torch.from_numpy(wab)).numpy()
    assert np.abs(out - ref).max() < 1e-12


def test_getid_arithmetic_with_injected_network():
    """can_swap_e2e.py:102-107: nearest resize to 112x112 -> identity network -> L2 normalisation; the network is injected."""
    from canonswap_amd.can_swap_e2e import can_swapper

    class Net(torch.nn.Module):
        def forward(self, x):
            assert x.shape[-2:] == (112, 112)
            return x.mean(dim=(2, 3)).repeat(1, 171)[:, :512] + 0.25, None      # (id, aux) like the reference's ArcFace module

    host = type("H", (), {"netArc": Net()})()
    img = torch.rand(2, 3, 160, 160)
    got = can_swapper.getid(host, img)
    want, _ = Net()(F.interpolate(img, size=(112, 112)))
    want = F.normalize(want, p=2, dim=1)
    assert got.shape == (2, 512) and torch.allclose(got, want) and torch.allclose(got.norm(dim=1), torch.ones(2))
    host.netArc = None
    with pytest.raises(RuntimeError, match="identity network"):
        can_swapper.getid(host, img)
