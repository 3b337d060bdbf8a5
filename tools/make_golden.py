"""Generate tests/golden/*.npz from the REFERENCE modules (build container only).

Imports /root/reference/src/modules/* read-only, loads the repo's deterministic synthetic
weights (canonswap_amd.synth, seed 0) with load_state_dict(strict=True), runs the exact per-frame
sequence of src/can_swap_pipeline_e2e.py:242-263 and stores
  * full fp16 output images,
  * for every stage boundary: values at a fixed pseudo-random index set + mean/std/L2,
  * small unit vectors (rotation matrix / key-point transform from src/utils/camera.py).
The fixtures are data only; the oracle (oracle/canonswap_ref.py) and the HIP engine are both
checked against them.  Re-run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from canonswap_amd import synth  # noqa: E402
from ref_import import load_reference_modules, reference_frame, REF  # noqa: E402

N_SAMPLES = 4096
BOUNDARIES = ("f_s", "f_can", "occ", "f_swap", "f_ref", "seg", "deformation", "occ2")


def sample_idx(name, numel):
    r = np.random.Generator(np.random.PCG64([1234, len(name), numel]))
    return r.integers(0, numel, size=min(N_SAMPLES, numel))


def case(mods, size, n_frames, frame_seed, id_seed, debug):
    inp = synth.make_frame_inputs(n_frames, seed=frame_seed, size=size)
    idv = torch.from_numpy(synth.make_identity(id_seed))
    args = [torch.from_numpy(inp[k]) for k in ("img", "x_t", "x_can")]
    r = reference_frame(mods, *args, idv.expand(n_frames, -1))
    out = {"size": size, "n_frames": n_frames, "frame_seed": frame_seed, "id_seed": id_seed,
           "out_f16": r["out"].numpy().astype(np.float16)}
    for k in BOUNDARIES:
        v = r[k].numpy().reshape(-1)
        idx = sample_idx(k, v.size)
        out[k + "_idx"] = idx
        out[k + "_val"] = v[idx].astype(np.float32)
        out[k + "_stats"] = np.array([v.mean(), v.std(), np.sqrt((v.astype(np.float64) ** 2).sum())], np.float64)
    if debug:   # the two debug decodes of can_swap_pipeline_e2e.py:248,257
        with torch.no_grad():
            w, g = mods["warping_module"], mods["spade_generator"]
            out["rec_can_f16"] = g(w.warp_out(r["f_can"], r["occ"])).numpy().astype(np.float16)
            out["swap_can_f16"] = g(w.warp_out(r["f_swap"], r["occ"])).numpy().astype(np.float16)
    return out


def unit_vectors():
    sys.path.insert(0, REF)
    from src.utils.camera import get_rotation_matrix, headpose_pred_to_degree
    r = np.random.Generator(np.random.PCG64(99))
    pyr = torch.from_numpy(r.uniform(-40, 40, size=(5, 3)).astype(np.float32))
    rot = get_rotation_matrix(pyr[:, 0], pyr[:, 1], pyr[:, 2])
    bins = torch.from_numpy(r.standard_normal((5, 66)).astype(np.float32) * 3)
    deg = headpose_pred_to_degree(bins)
    kp = torch.from_numpy((0.3 * r.standard_normal((5, 21, 3))).astype(np.float32))
    exp = torch.from_numpy((0.02 * r.standard_normal((5, 21, 3))).astype(np.float32))
    t = torch.from_numpy(r.uniform(-0.1, 0.1, size=(5, 3)).astype(np.float32))
    scale = torch.from_numpy(r.uniform(0.9, 1.3, size=(5, 1)).astype(np.float32))
    # can_swap_e2e.py:245-254 evaluated with the reference's own camera helpers
    xt = kp @ rot + exp
    xt = xt * scale[..., None]
    xt[:, :, 0:2] += t[:, None, 0:2]
    from src.utils.retargeting_utils import calc_eye_close_ratio, calc_lip_close_ratio
    lmk = r.uniform(0, 256, size=(4, 106, 2)).astype(np.float32)
    eye, lip = calc_eye_close_ratio(lmk), calc_lip_close_ratio(lmk)
    return dict(lmk=lmk, eye_ratio=eye, lip_ratio=lip,
                pyr=pyr.numpy(), rot=rot.numpy(), bins=bins.numpy(), deg=deg.numpy(), kp=kp.numpy(),
                exp=exp.numpy(), t=t.numpy(), scale=scale.numpy(), x_transformed=xt.numpy())


def motion_case(n=3, img_seed=2000, size=256):
    """Reference MotionExtractor (motion_extractor.py:18-35) + per-stage statistics of its ConvNeXtV2 trunk."""
    sys.path.insert(0, REF)
    import yaml
    from src.modules.motion_extractor import MotionExtractor
    cfg = yaml.safe_load(open(os.path.join(REF, "src/config/models.yaml")))["model_params"]
    m = MotionExtractor(**cfg["motion_extractor_params"]).eval()
    m.load_state_dict(synth.to_torch(synth.make_state_dicts(0, modules=("motion_extractor",)))["motion_extractor"], strict=True)
    img = torch.from_numpy(synth.make_smooth_images(n, seed=img_seed, size=size))
    out = {"n": n, "img_seed": img_seed, "size": size}
    with torch.no_grad():
        for k, v in m(img).items():
            out[k] = v.numpy().astype(np.float32)
        x, d = img, m.detector
        for i in range(4):
            x = d.downsample_layers[i](x)
            x = d.stages[i](x)
            v = x.numpy().reshape(-1)
            idx = sample_idx(f"m_stage{i}", v.size)
            out[f"stage{i}_idx"] = idx
            out[f"stage{i}_val"] = v[idx].astype(np.float32)
            out[f"stage{i}_stats"] = np.array([v.mean(), v.std()], np.float64)
    return out


def main():
    torch.manual_seed(0)
    sds = synth.to_torch(synth.make_state_dicts(0))
    mods = load_reference_modules(sds)
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    np.savez_compressed(os.path.join(gold, "frame_128_b2.npz"), **case(mods, 128, 2, 2000, 7, debug=True))
    np.savez_compressed(os.path.join(gold, "frame_256_b1.npz"), **case(mods, 256, 1, 1000, 7, debug=False))
    np.savez_compressed(os.path.join(gold, "unit_vectors.npz"), **unit_vectors())
    np.savez_compressed(os.path.join(gold, "motion_b3.npz"), **motion_case())
    for f in sorted(os.listdir(gold)):
        print(f, os.path.getsize(os.path.join(gold, f)))


if __name__ == "__main__":
    main()
