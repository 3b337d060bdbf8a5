"""dm_softmax_warp_kernel: softmax over the 22 mask logits -> deformation -> trilinear feature warp in one kernel (dense_motion.py:88-94 ->
warping_network.py:46-62), the sampling grid never going through HBM.  Same per-voxel operation sequence as dm_softmax_kernel followed by
grid_sample_kernel: identical bits (second engine with CANONSWAP_WARP_FUSED=0 in a subprocess); parity against the oracle is covered by
tests/test_gpu_stages.py (test_warp / test_warp_decode run the fused kernel by default)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fused_softmax_warp_equals_two_kernels(state_dicts):
    from canonswap_amd import synth
    from canonswap_amd.can_swap_e2e import can_swapper
    r = np.random.Generator(np.random.PCG64(17))
    f = torch.from_numpy((0.08 * r.standard_normal((3, 32, 16, 64, 64))).astype(np.float32))
    inp = synth.make_frame_inputs(3, seed=77, size=256)
    x_t, x_can = torch.from_numpy(inp["x_t"]), torch.from_numpy(inp["x_can"])
    torch.save((f, x_t, x_can), "/tmp/wf_in.pt")
    sw = can_swapper(None, state_dicts=state_dicts, max_batch=3)
    out, occ = sw.warping_module.warp(f.cuda(), kp_source=x_t.cuda(), kp_driving=x_can.cuda())
    fwd = sw.warping_module(f.cuda(), kp_source=x_can.cuda(), kp_driving=x_t.cuda())
    got = [out.cpu(), occ.cpu(), fwd["out"].cpu(), fwd["deformation"].cpu()]
    code = ("import torch, sys; sys.path.insert(0, %r); from canonswap_amd import synth; from canonswap_amd.can_swap_e2e import can_swapper;"
            "sd = synth.to_torch(synth.make_state_dicts(0)); sw = can_swapper(None, state_dicts=sd, max_batch=3);"
            "f, x_t, x_can = torch.load('/tmp/wf_in.pt');"
            "out, occ = sw.warping_module.warp(f.cuda(), kp_source=x_t.cuda(), kp_driving=x_can.cuda());"
            "fwd = sw.warping_module(f.cuda(), kp_source=x_can.cuda(), kp_driving=x_t.cuda());"
            "torch.save([out.cpu(), occ.cpu(), fwd['out'].cpu(), fwd['deformation'].cpu()], sys.argv[1])") % ROOT
    env = dict(os.environ, CANONSWAP_WARP_FUSED="0")
    rr = subprocess.run([sys.executable, "-c", code, "/tmp/wf_off.pt"], env=env, capture_output=True, text=True, timeout=600)
    assert rr.returncode == 0, rr.stderr[-2000:]
    want = torch.load("/tmp/wf_off.pt")
    for a, b in zip(got, want):
        assert torch.equal(a, b), float((a - b).abs().max())
