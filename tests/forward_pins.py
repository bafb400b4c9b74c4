"""OUR PointGroup / HAIS / SoftGroup `forward` + `_loss` against tests/golden/forward_cases.npz -- what the REFERENCE's
own model code (pointgroup.py:23-110, hais.py:28-128, softgroup.py:32-183, general_model.py:36-50,152-193) produced
on the same seeded batch, parameters, steering and uniform draws (tests/golden/make_golden_forward.py).  Shared by the
CPU run (oracle test double behind the operators) and the GPU run (HIP kernels through the C ABI).

Bars: proposal lists (which points, which proposal, in which order) bit-exact; scores, per-point mask scores and
every loss term within `tol` of the largest entry.  Gradients: the heads' within `tol`-grade bars; the proposal network's
first kernel within 2e-2; the backbone's input kernel by direction (cosine >= 0.995) -- a ReLU / max-pool mask that
flips on a 1e-7 difference moves a gradient by a whole term, and how often that happens is a
property of the case, not of the code: with the backbone cut to 3 or 4 levels the reference's composition and ours
agree on that kernel's gradient to 1e-7 / 4e-7 of its largest entry, with 2 levels to 7e-3, with 5 to 5e-4, with all 7
to 4e-3 (measured on the pg_train case in the build container, both sides on the same CPU operators)."""
import os

import numpy as np
import torch

from forward_cases import CASES, GRAD_KEYS, M, Steered, grouping_batch, steering, summarise, tweak
from model_cases import seeded_fill

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "forward_cases.npz")


def _rel(got, want):
    return float(np.abs(got.astype(np.float64) - want).max() / max(float(np.abs(want).max()), 1e-30))


def run_case(tag, device):
    from minsu3d_amd import model as models
    from minsu3d_amd.config import load_config
    _, name, training, epoch = next(c for c in CASES if c[0] == tag)
    want = {k.split("/", 1)[1]: v for k, v in np.load(GOLDEN).items() if k.startswith(tag + "/")}
    cfg = load_config([f"model={name}", f"model.network.m={M}"])
    model = getattr(models, cfg.model.network.module)(cfg)
    seeded_fill(model, 31)
    tweak(model, name)
    batch = grouping_batch(device=device)
    model.backbone = Steered(model.backbone, *steering(batch))
    model = model.to(device).train(training)
    model.current_epoch = epoch
    rand = torch.from_numpy(want["rand"]).to(device)
    model.voxelization_rand = (rand[0], rand[1])
    with torch.set_grad_enabled(training):
        out = model(batch)
        losses = model._loss(batch, out)
    got = summarise(out, losses, name)
    grads = {}
    if training:
        sum(losses.values()).backward()
        params = dict(model.named_parameters())
        grads = {k: params[k].grad.detach().cpu().numpy() for k in GRAD_KEYS[name]}
    return name, got, grads, want


def check_case(tag, device, tol, head_grad_tol):
    name, got, grads, want = run_case(tag, device)
    report = {}
    # ---- which points were grouped into which proposal, in which order: exact
    for k in ("proposals_offset", "proposals_idx") + (("instance_batch_idxs",) if name == "softgroup" else ()):
        assert got[k].shape == want[k].shape, (tag, k, got[k].shape, want[k].shape)
        assert np.array_equal(got[k], want[k]), (tag, k, int((got[k] != want[k]).sum()))
    # ---- per-point / per-proposal scores
    skip = np.zeros(want["proposals_offset"].shape[0] - 1, bool)
    if name == "hais":
        e = _rel(got["mask_scores"], want["mask_scores"])
        report["mask_scores"] = e
        assert e <= tol, (tag, "mask_scores", e)
        # the mask filter (hais.py:81-84) is a step at sigmoid = 0.5: a point within the float noise of the step may
        # fall on the other side and move its proposal's pooled features by a whole term -- those proposals (none in
        # the committed fixture: the closest point is 1.5e-4 away) are compared on everything but their score
        # (a mask score within `tol` of the largest one moves its sigmoid by at most a quarter of that)
        sig = 1 / (1 + np.exp(-want["mask_scores"].astype(np.float64).reshape(-1)))
        risky = np.nonzero(np.abs(sig - 0.5) < 0.25 * tol * float(np.abs(want["mask_scores"]).max()))[0]
        skip[np.searchsorted(want["proposals_offset"], risky, side="right") - 1] = True
        assert skip.sum() <= 1, (tag, "too many proposals at the mask-filter step", int(skip.sum()))
    for k in ("scores", "cls_scores", "iou_scores") + (("mask_scores",) if name == "softgroup" else ()):
        if k in want:
            g, w = got[k], want[k]
            assert g.shape == w.shape, (tag, k, g.shape, w.shape)
            if k == "scores":
                g, w = g[~skip], w[~skip]
            e = _rel(g, w)
            report[k] = e
            assert e <= tol, (tag, k, e)
    # ---- losses: same terms in the same order, same values
    assert list(got["loss_names"]) == list(want["loss_names"]), (tag, got["loss_names"], want["loss_names"])
    for n_, g, w in zip(want["loss_names"], got["loss_values"], want["loss_values"]):
        e = abs(g - w) / max(abs(w), 1e-3)
        report[str(n_)] = e
        assert e <= (tol if not skip.any() or n_ != "score_loss" else 1e-2), (tag, str(n_), g, w)
    for k, g in grads.items():
        w = want["grad:" + k]
        assert g.shape == w.shape, (tag, k)
        e = _rel(g, w)
        report["grad:" + k] = e
        if k.startswith("backbone."):
            cos = float((g.astype(np.float64) * w).sum() / np.linalg.norm(g) / np.linalg.norm(w))
            report["cos:" + k] = 1 - cos
            assert cos >= 0.995 and e <= 0.2, (tag, "grad", k, e, cos)
        else:
            assert e <= (2e-2 if k.endswith(".kernel") else head_grad_tol), (tag, "grad", k, e)
    return report
