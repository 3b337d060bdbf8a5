"""BASELINE configs[4] numerics: fp16 activations + conv weights quantised to fp8 e4m3 with a per-out-channel scale.

Two statements are tested: (1) the engine computes the QUANTISED network as accurately as it computes the original one
(>= 50 dB against the fp32 oracle running on the same quantised weights, quantiser cross-checked against the independent
restatement in oracle/quant.py); (2) what the quantisation itself costs against the unquantised fp32 reference is REPORTED per
module (printed; DESIGN.md quotes it) - fp8 weights do not meet the 50 dB gate and are not claimed to."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_quantiser_matches_independent_restatement(state_dicts_np):
    from canonswap_amd import pack
    from oracle import quant
    q = pack.quantize_conv_weights_e4m3(state_dicts_np)
    n = 0
    for m in ("spade_generator", "transfer", "warping_module"):
        for k, v in state_dicts_np[m].items():
            if v.ndim >= 4 and (k.endswith(".weight") or k.endswith(".weight_orig")):
                assert np.array_equal(q[m][k], quant.quantize_rows(v)), (m, k)
                n += 1
            else:
                assert np.array_equal(np.asarray(q[m][k]), np.asarray(v))
    assert n > 60


def test_fp8_weight_network_vs_oracle_and_cost_vs_fp32(state_dicts_np, state_dicts):
    from canonswap_amd import pack, synth
    from canonswap_amd.can_swap_e2e import can_swapper
    from oracle import canonswap_ref as O
    qsd = synth.to_torch(pack.quantize_conv_weights_e4m3(state_dicts_np))
    inp = synth.make_frame_inputs(2, seed=1000, size=256)
    args = [torch.from_numpy(inp[k]) for k in ("img", "x_t", "x_can")]
    ids = torch.from_numpy(synth.make_identity(7, n=2))                 # two identities in the batch (configs[4] streams)
    with torch.no_grad():
        ref_q = [O.swap_frame(qsd, *(a[b:b + 1] for a in args), ids[b:b + 1]) for b in range(2)]
        ref_f = [O.swap_frame(state_dicts, *(a[b:b + 1] for a in args), ids[b:b + 1]) for b in range(2)]
    sw = can_swapper(None, state_dicts=state_dicts, max_batch=2, fp8_weights=True)
    try:
        out = sw.swap_frames(*(a.cuda() for a in args), ids.cuda())["out"].cpu()
        f_s = sw.extract_feature_3d(args[0].cuda()).cpu()
    finally:
        sw.engine.close()
    for b in range(2):
        p_q = O.psnr(out[b:b + 1], ref_q[b]["out"])
        p_f = O.psnr(out[b:b + 1], ref_f[b]["out"])
        p_qf = O.psnr(ref_q[b]["out"], ref_f[b]["out"])
        print(f"frame {b}: engine(fp8 w) vs oracle(fp8 w) {p_q:.2f} dB | engine(fp8 w) vs fp32 reference {p_f:.2f} dB | "
              f"oracle(fp8 w) vs fp32 reference {p_qf:.2f} dB")
        assert p_q >= 50.0, (b, p_q)
        assert abs(p_f - p_qf) < 1.5            # the engine adds nothing visible on top of the quantisation error
    for k in ("f_s", "f_can", "f_swap", "f_ref", "seg"):
        a, bq = ref_f[0][k], ref_q[0][k]
        print(f"  quantisation error after {k}: rel L2 {float((a - bq).norm() / a.norm()):.3e}")
    assert float((f_s[:1] - ref_q[0]["f_s"]).norm() / ref_q[0]["f_s"].norm()) < 5e-3
