"""Golden vectors of the reference's WHOLE model forward + loss -- runs ONLY in the build container
(`python tests/golden/make_golden_forward.py`), where /root/reference exists; writes DATA only
(tests/golden/forward_cases.npz: proposal lists, scores, loss values, a few gradients).

What runs here is the reference's own Python: `PointGroup.forward/_loss` (minsu3d/model/pointgroup.py:23-110),
`HAIS.forward/_loss` (hais.py:28-128), `SoftGroup.forward/_loss` (softgroup.py:32-183), `clusters_voxelization` and
`GeneralModel._loss` (general_model.py:36-50,152-193), `Backbone` / `TinyUnet` (model/module/*.py) -- on the stand-ins
of make_golden_model.py (pytorch_lightning / hydra shims; this repository's `MinkowskiEngine` and `COMMON_OPS` module
names with the CPU test double behind the operators).  One more alias is needed: the reference's operator WRAPPERS
(minsu3d/common_ops/functions/*.py) allocate on `device="cuda"` and assert `.is_cuda` (common_ops.py:27-33), so they
cannot run in a container without a GPU; the four wrapper modules are aliased to this repository's wrappers of the
same names and signatures (row B2 of SURVEY section 8 -- pinned on their own by tests/test_dropin_*.py).

Case list, batch and steering: tests/forward_cases.py.  The two `torch.rand(3)` draws of
general_model.py:178-179 are recorded and stored, the tests inject them (`model.voxelization_rand`).
"""
import os
import sys

sys.dont_write_bytecode = True          # nothing is written into /root/reference (no __pycache__ there)
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from make_golden_model import install_standins, reference_cfg      # noqa: E402


def alias_wrappers():
    import importlib
    for name in ("common_ops", "pointgroup_ops", "hais_ops", "softgroup_ops"):
        ours = importlib.import_module("minsu3d_amd.common_ops.functions." + name)
        sys.modules["minsu3d.common_ops.functions." + name] = ours


def main():
    install_standins()
    alias_wrappers()
    import minsu3d.model as RM
    import minsu3d.common_ops.functions as RF
    import importlib
    for name in ("common_ops", "pointgroup_ops", "hais_ops", "softgroup_ops"):
        setattr(RF, name, importlib.import_module("minsu3d_amd.common_ops.functions." + name))
    import minsu3d.model.general_model as RGM
    assert RGM.common_ops.__name__.startswith("minsu3d_amd."), RGM.common_ops
    from forward_cases import CASES, GRAD_KEYS, M, Steered, grouping_batch, steering, summarise, tweak
    from model_cases import seeded_fill

    classes = {"pointgroup": RM.PointGroup, "hais": RM.HAIS, "softgroup": RM.SoftGroup}
    arrays = {}
    real_rand = torch.rand
    for tag, name, training, epoch in CASES:
        model = classes[name](reference_cfg(name, m=M))
        seeded_fill(model, 31)
        tweak(model, name)
        batch = grouping_batch()
        model.backbone = Steered(model.backbone, *steering(batch))
        model.train(training)
        model.current_epoch = epoch
        draws = []

        def recording_rand(*a, **k):
            r = real_rand(*a, **k)
            draws.append(r.clone())
            return r

        torch.manual_seed(5)
        torch.rand = recording_rand
        try:
            with torch.set_grad_enabled(training):
                out = model(batch)
                losses = model._loss(batch, out)
        finally:
            torch.rand = real_rand
        assert len(draws) == 2 and all(d.shape == (3,) for d in draws), [d.shape for d in draws]
        a = summarise(out, losses, name)
        a["rand"] = torch.stack(draws).numpy()
        if training:
            sum(losses.values()).backward()
            params = dict(model.named_parameters())
            for k in GRAD_KEYS[name]:
                a["grad:" + k] = params[k].grad.numpy()
        for k, v in a.items():
            arrays[f"{tag}/{k}"] = v
        # what the case exercises (printed so that a fixture with an idle branch is noticed when it is made)
        P = a["proposals_offset"].shape[0] - 1
        sizes = np.diff(a["proposals_offset"])
        msg = f"{tag}: {batch['point_xyz'].shape[0]} points, {P} proposals ({a['proposals_idx'].shape[0]} rows, " \
              f"sizes {sizes.min()}..{sizes.max()}), losses " + \
              ", ".join(f"{k}={v:.5f}" for k, v in zip(a["loss_names"], a["loss_values"]))
        if "mask_scores" in a and name == "hais":
            sig = 1 / (1 + np.exp(-a["mask_scores"].astype(np.float64)))
            msg += f"; mask filter margin {np.abs(sig - 0.5).min():.2e}, kept {(sig >= 0.5).mean():.2f}"
        print(msg)
    np.savez_compressed(os.path.join(HERE, "forward_cases.npz"), **arrays)
    print("wrote forward_cases.npz (%d arrays, %.0f kB)" % (len(arrays), os.path.getsize(os.path.join(HERE, "forward_cases.npz")) / 1e3))


if __name__ == "__main__":
    main()
