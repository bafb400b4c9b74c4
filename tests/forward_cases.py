"""Seeded batches, steering and case list of the whole-model pins: shared by tests/golden/make_golden_forward.py
(which runs the REFERENCE's PointGroup / HAIS / SoftGroup `forward` + `_loss` on them in the build container) and by
tests/test_forward_pins_{cpu,gpu}.py (which run OUR models on them).

A randomly initialised network predicts no foreground, so nothing would be grouped and no branch behind the grouping
would fire.  Both sides therefore wrap their `backbone` in `Steered`: the REAL backbone runs (its `point_features`
feed the proposal networks unchanged), its `semantic_scores` are damped and pushed towards a seeded label field and
its `point_offsets` are replaced by a seeded offset field (exactly -- the ball query's radius test must see identical
coordinates on every backend).  Everything behind the backbone -- foreground mask, scene offsets, the grouping calls
and their merge / renumbering, the proposal cap, voxelisation with the two captured uniform draws, the proposal
network, pools, heads and every loss term -- is the model's own code on both sides."""
import numpy as np
import torch
import torch.nn as nn

# (tag, model, training mode?, current_epoch).  HAIS: epoch 150 takes get_mask_iou_on_cluster and no mask filter
# (hais.py:81-84,103-112: both switch at epoch 200); eval mode takes the set aggregation (hais.py:51).
CASES = [("pg_train", "pointgroup", True, 1000),
         ("hais_train", "hais", True, 1000),
         ("hais_early", "hais", True, 150),
         ("hais_eval", "hais", False, 1000),
         ("sg_train", "softgroup", True, 1000)]
M = 16                       # network width of every case (the reference's YAML has 16 / 32 / 32)
GRAD_KEYS = {"pointgroup": ["score_branch.weight", "score_net.unet.0.blocks.block0.conv_branch.2.kernel",
                            "backbone.inner.unet.0.kernel"],
             "hais": ["score_branch.weight", "mask_branch.2.weight", "tiny_unet.unet.0.blocks.block0.conv_branch.2.kernel",
                      "backbone.inner.unet.0.kernel"],
             "softgroup": ["classification_branch.weight", "iou_score.weight", "mask_scoring_branch.2.weight",
                           "tiny_unet.unet.0.blocks.block0.conv_branch.2.kernel", "backbone.inner.unet.0.kernel"]}


def grouping_batch(seed=41, device="cpu"):
    """two small rooms with boxes of 2 cm-grouping density (SURVEY 8d generator) -> the collate dictionary"""
    from minsu3d_amd.data import synthetic as S
    scenes = [S.make_scene(seed, room=(1.3, 1.0), n_boxes=3, density=1500.0, wall_h=0.3),
              S.make_scene(seed + 1, room=(1.1, 1.2), n_boxes=2, density=1500.0, wall_h=0.25)]
    return S.to_torch(S.collate(scenes), device)


def steering(batch, seed=7, n_classes=20, cell=0.4):
    """-> (sem_push f32 [N, C], offsets f32 [N, 3]).  Per point: the ground-truth label and the offset to its instance
    centre (x 0.97, 1 cm jitter).  On top, whole PATCHES of an object (its points in one 40 cm cell, a few hundred
    points) are treated alike, so that the patches survive the grouping thresholds:
      12 % relabelled to another class    -> wrong-class proposals, IoUs between 0 and 1, the score targets' linear part
      12 % displaced by 12 cm after the shift -> a second, small cluster of the SAME class next to the object: HAIS'
                                                 fragments (absorbed under set aggregation) and "kept" clusters
      15 % pushed towards a second class too  -> SoftGroup groups them under both classes
    and 3 % of all points are relabelled individually.  Offsets are quantised to 1/1024 m so that xyz + offset is the
    same float wherever it is added."""
    g = np.random.default_rng(seed)
    sem = batch["sem_labels"].cpu().numpy().astype(np.int64)
    xyz = batch["point_xyz"].cpu().numpy()
    centre = batch["instance_center_xyz"].cpu().numpy()
    inst = batch["instance_ids"].cpu().numpy().astype(np.int64)
    n = sem.shape[0]
    c = np.floor((xyz - xyz.min(0)) / cell).astype(np.int64)
    key = ((inst + 1) * 64 + c[:, 0]) * 4096 + c[:, 1] * 64 + c[:, 2]
    _, patch = np.unique(key, return_inverse=True)
    n_patch = int(patch.max()) + 1
    kind = g.random(n_patch)                       # one draw per patch
    other = g.integers(2, n_classes, n_patch)
    on_object = sem >= 2
    lab = sem.copy()
    relabel = on_object & (kind[patch] < 0.12)
    lab[relabel] = other[patch][relabel]
    lone = g.random(n) < 0.03
    lab[lone] = g.integers(0, n_classes, int(lone.sum()))
    push = np.zeros((n, n_classes), np.float32)
    push[np.arange(n), lab] = 10.0
    dual = on_object & (kind[patch] >= 0.24) & (kind[patch] < 0.39)
    second = np.where(other[patch] == lab, (other[patch] - 2 + 1) % (n_classes - 2) + 2, other[patch])
    push[np.nonzero(dual)[0], second[dual]] = 9.5
    off = np.where((inst >= 0)[:, None], (centre - xyz) * 0.97, 0.0) + g.normal(0, 0.01, (n, 3))
    displaced = on_object & (kind[patch] >= 0.12) & (kind[patch] < 0.24)
    off[displaced, 0] += 0.12
    off = (np.round(off * 1024) / 1024).astype(np.float32)
    dev = batch["point_xyz"].device
    return torch.from_numpy(push).to(dev), torch.from_numpy(off).to(dev)


def tweak(model, name):
    """after seeded_fill: HAIS' mask head gets a positive bias -- with a symmetric one half of every proposal's points
    fall below the 0.5 mask threshold, no proposal reaches IoU 0.5 on the predicted masks and the mask loss is weighted
    0 everywhere (hais.py:103-119)"""
    if name == "hais":
        with torch.no_grad():
            model.mask_branch[2].bias.fill_(1.0)


class Steered(nn.Module):
    """backbone wrapper, see the module docstring (`inner` keeps the wrapped module's parameters under backbone.inner.*)"""

    def __init__(self, inner, sem_push, offsets):
        super().__init__()
        self.inner = inner
        self.sem_push, self.offsets = sem_push, offsets

    def forward(self, *args):
        out = self.inner(*args)
        out["semantic_scores"] = out["semantic_scores"] * 0.01 + self.sem_push
        out["point_offsets"] = out["point_offsets"] * 0 + self.offsets
        return out


def summarise(out, losses, model_name):
    """the arrays a case is compared on: proposal lists (exact), scores and losses (float)"""
    a = {}
    if model_name == "softgroup":
        a["proposals_idx"] = out["proposals_idx"].detach().cpu().numpy().astype(np.int32)
        a["proposals_offset"] = out["proposals_offset"].detach().cpu().numpy().astype(np.int32)
        for k in ("mask_scores", "cls_scores", "iou_scores"):
            a[k] = out[k].detach().cpu().numpy()
        a["instance_batch_idxs"] = out["instance_batch_idxs"].detach().cpu().numpy().astype(np.int32)
    else:
        ps = out["proposal_scores"]
        a["scores"] = ps[0].detach().cpu().numpy()
        a["proposals_idx"] = ps[1].detach().cpu().numpy().astype(np.int32)
        a["proposals_offset"] = ps[2].detach().cpu().numpy().astype(np.int32)
        if len(ps) > 3:
            a["mask_scores"] = ps[3].detach().cpu().numpy()
    a["loss_names"] = np.array(list(losses.keys()))
    a["loss_values"] = np.array([float(v.detach()) for v in losses.values()], np.float64)
    return a
