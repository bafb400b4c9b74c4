"""seeded parameters and inputs shared by tests/golden/make_golden_model.py (which runs the REFERENCE's module classes
on them in the build container) and the tests that run OUR modules on them"""
import re

import numpy as np
import torch


def seeded_fill(module, seed):
    """deterministic parameters / buffers from (seed, sorted key order): the same numbers in two module trees exactly
    when their state_dict keys and shapes agree"""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    for k in sorted(sd):
        v = sd[k]
        if not v.dtype.is_floating_point:
            continue
        r = torch.randn(v.shape, generator=g)
        if k.endswith("running_var"):
            r = r.abs() * 0.5 + 0.5
        elif k.endswith("bn.weight") or re.search(r"_branch\.1\.weight$", k):
            r = 1.0 + 0.2 * r
        elif k.endswith("kernel"):
            r = r / np.sqrt(v.shape[-2] * (v.shape[0] if v.dim() == 3 else 1))
        elif k.endswith("weight") and v.dim() == 2:
            r = r / np.sqrt(v.shape[1])
        else:
            r = 0.2 * r
        v.copy_(r)
    module.load_state_dict(sd)


def small_scene(seed, device="cpu"):
    from minsu3d_amd.data import synthetic as S
    return S.to_torch(S.collate([S.make_scene(seed, room=(0.9, 0.8), n_boxes=2, density=330.0, wall_h=0.4),
                                 S.make_scene(seed + 1, room=(0.8, 0.9), n_boxes=1, density=330.0, wall_h=0.3)]), device)


def tiny_input(seed=3):
    """proposal-grid-shaped sparse input of the TinyUnet case: 3 proposals in a 12^3 cube, 4 channels"""
    g = torch.Generator().manual_seed(seed)
    coords = torch.unique(torch.cat([torch.randint(0, 3, (900, 1), generator=g),
                                     torch.randint(0, 12, (900, 3), generator=g)], 1).int(), dim=0)
    return coords, torch.randn(coords.size(0), 4, generator=g)


def loss_inputs(seed=21, n=700):
    g = torch.Generator().manual_seed(seed)
    return dict(sem_scores=torch.randn(n, 20, generator=g), labels=torch.randint(-1, 20, (n,), generator=g).to(torch.int16),
                inst=torch.randint(-1, 6, (n,), generator=g).to(torch.int16), xyz=torch.rand(n, 3, generator=g) * 4,
                centre=torch.rand(n, 3, generator=g) * 4, offs=torch.randn(n, 3, generator=g),
                seg=torch.rand(500, generator=g) * 1.2 - 0.1)
