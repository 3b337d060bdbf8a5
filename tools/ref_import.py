"""Import the reference's pure-torch generator modules read-only from /root/reference.

Only usable in the build container (the reference does not travel to the GPU box).  Used by
``tools/make_golden.py`` to validate ``oracle/`` and to emit the fixtures under ``tests/golden/``.
"""
import os
import sys

REF = os.environ.get("CANONSWAP_REFERENCE", "/root/reference")


def load_reference_modules(state_dicts=None):
    import torch
    import yaml
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from src.modules.appearance_feature_extractor import AppearanceFeatureExtractor
    from src.modules.warping_network import WarpingNetwork
    from src.modules.spade_generator import SPADEDecoder
    from src.modules.adaptive_modulate import transfer_model2, G3d
    cfg = yaml.safe_load(open(os.path.join(REF, "src/config/models.yaml")))["model_params"]
    cfg["spade_generator_params"]["upscale"] = 2            # src/can_swap_e2e.py:62
    mods = {
        "appearance_feature_extractor": AppearanceFeatureExtractor(**cfg["appearance_feature_extractor_params"]),
        "warping_module": WarpingNetwork(**cfg["warping_module_params"]),
        "spade_generator": SPADEDecoder(**cfg["spade_generator_params"]),
        "transfer": transfer_model2(),
        "refine": G3d(),
    }
    for k, m in mods.items():
        m.eval()
        if state_dicts is not None:
            m.load_state_dict(state_dicts[k], strict=True)
    return mods


def reference_frame(mods, img, x_t, x_can, source_id, debug=False):
    """The exact per-frame sequence of src/can_swap_pipeline_e2e.py:242-263."""
    import torch
    with torch.no_grad():
        f_s = mods["appearance_feature_extractor"](img).float()
        f_can, occ = mods["warping_module"].warp(f_s, x_t, x_can)
        f_swap = mods["transfer"](f_can, source_id)
        f_ref = mods["refine"](f_swap)
        ret = mods["warping_module"](f_ref, kp_source=x_can, kp_driving=x_t)
        seg = ret["out"]
        out = mods["spade_generator"](feature=seg)
    return dict(f_s=f_s, f_can=f_can, occ=occ, f_swap=f_swap, f_ref=f_ref, seg=seg,
                deformation=ret["deformation"], occ2=ret["occlusion_map"], out=out)
