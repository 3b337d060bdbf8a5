"""Golden vectors for SoftErosion (src/utils/crop.py:21-47) produced by the reference's OWN class.

crop.py cannot be imported here (its first statement imports cv2, which is not installed), so the class definition is cut out
of the reference source with `ast` at generation time and executed as is, with torch in its namespace - the reference's code
runs, nothing of it is copied into this repository.  Only the resulting arrays are committed (tests/golden/soft_erosion.npz).

    python tools/make_golden_tail.py        # build container only (/root/reference must exist)
"""
import ast
import os

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/utils/crop.py"


def reference_class():
    tree = ast.parse(open(SRC).read())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SoftErosion")
    mod = ast.Module(body=[node], type_ignores=[])
    ns = {"torch": torch, "F": F}
    exec(compile(mod, SRC, "exec"), ns)
    return ns["SoftErosion"]


def face_like_mask(seed, size=512):
    """A blob with holes and ragged edges, like the face-parsing mask the pipeline feeds (int 0/1)."""
    r = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    cx, cy = size * r.uniform(0.4, 0.6), size * r.uniform(0.4, 0.6)
    rad = size * r.uniform(0.25, 0.35)
    ang = np.arctan2(yy - cy, xx - cx)
    wob = 1 + 0.15 * np.sin(3 * ang + r.uniform(0, 6)) + 0.08 * np.sin(7 * ang + r.uniform(0, 6))
    m = (np.hypot(xx - cx, yy - cy) < rad * wob)
    for _ in range(3):      # holes (eyes / mouth regions excluded by the parser)
        hx, hy, hr = cx + rad * r.uniform(-0.5, 0.5), cy + rad * r.uniform(-0.5, 0.5), rad * r.uniform(0.05, 0.15)
        m &= np.hypot(xx - hx, yy - hy) > hr
    return m.astype(np.int32)


def main():
    SE = reference_class()
    out = {}
    for name, (ks, thr, it) in {"e2e": (21, 0.9, 3), "v2i": (21, 0.9, 2)}.items():   # can_swap_pipeline_e2e.py:42, _v2i.py:43
        mod = SE(kernel_size=ks, threshold=thr, iterations=it)
        for k, seed in enumerate((11, 12)):
            m = face_like_mask(seed)
            x, hard = mod(torch.from_numpy(m).unsqueeze(0).unsqueeze(0))
            out[f"{name}_{k}_in"] = m.astype(np.uint8)
            out[f"{name}_{k}_soft"] = x[0, 0].numpy().astype(np.float32)
            out[f"{name}_{k}_hard"] = hard[0, 0].numpy()
        out[f"{name}_weight"] = mod.weight.numpy()
    path = os.path.join(ROOT, "tests", "golden", "soft_erosion.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
