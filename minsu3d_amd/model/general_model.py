"""Base task module (reference minsu3d/model/general_model.py:16-213), LightningModule-shaped without
Lightning: `hparams.cfg`, `current_epoch`, forward / _loss / training_step / configure_optimizers /
on_train_epoch_end.  Also the proposal voxelisation and the score-target helper."""
import importlib
import os
import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import MinkowskiEngine as ME
from ..backend import get_backend
from ..common_ops.functions import common_ops
from ..loss import PTOffsetLoss
from .module import Backbone


class _HandOverGradsFn(torch.autograd.Function):
    """The three per-point losses again, as a node whose backward HANDS OVER gradients that were computed earlier
    (GeneralModel._early_point_backward) instead of computing them: d(l0 + l1 + l2) / d(point features, head parameters),
    scaled by the upstream gradient.  The early result is exact whenever the three losses receive the SAME upstream
    gradient tensor (loss = sum(losses.values()), with or without a common factor: autograd hands one tensor to all
    terms of a sum, which is checked by storage address -- no device round trip); otherwise (weighted losses, a loss left
    out) the gradients are computed the ordinary way through the graph that was kept."""

    @staticmethod
    def forward(ctx, pack, pf, *params):
        ctx.pack = pack
        ctx.set_materialize_grads(False)
        return tuple(l.detach() for l in pack["losses"])

    @staticmethod
    def backward(ctx, *gs):
        pack = ctx.pack                      # (kept: a second backward pass over a retained graph hands the same gradients over)
        n = len(pack["grads"])
        if all(g is None for g in gs):
            return (None,) * (1 + n)
        if pack.get("done") is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(pack["done"])            # the early pass ran on the heads' own stream
            for g in pack["grads"]:
                if g is not None:
                    g.record_stream(cur)
        same = all(g is not None and g.numel() == 1 and g.data_ptr() == gs[0].data_ptr() for g in gs)
        if same and pack["grads"] is not None:
            s = gs[0].reshape(())
            return (None,) + tuple(None if g is None else g * s for g in pack["grads"])
        # the general case: a fresh evaluation of the losses (the fused loss node scales its gradients in place and can
        # run once) on the heads' graph, which the early pass kept
        with torch.enable_grad():
            fresh = list(pack["recompute"]().values())
        gl = [(f, g) for f, g in zip(fresh, gs) if g is not None]
        grads = torch.autograd.grad([f for f, _ in gl], pack["inputs"], grad_outputs=[g for _, g in gl], allow_unused=True)
        return (None,) + tuple(grads)


class GeneralModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.hparams = SimpleNamespace(cfg=cfg)
        self.current_epoch = 0
        net = cfg.model.network
        in_ch = 3 + 3 * int(net.use_color) + 3 * int(net.use_normal)
        self.backbone = Backbone(input_channel=in_ch, output_channel=net.m, block_channels=net.blocks,
                                 block_reps=net.block_reps, sem_classes=cfg.data.classes)
        self.offset_criterion = PTOffsetLoss()

    @property
    def device(self):
        return next(self.parameters()).device

    def configure_optimizers(self):
        opt = dict(self.hparams.cfg.model.optimizer)
        mod, _, name = opt.pop("_target_").rpartition(".")
        params = list(self.parameters())
        on_gpu = bool(params) and params[0].is_cuda
        if mod == "torch.optim" and name == "Adam" and on_gpu and os.environ.get("MS3D_ADAM", "1") != "0":
            from ..optim import Adam            # same state and arithmetic, ONE launch for all parameter tensors
            return Adam(params, **opt)
        if name in ("Adam", "AdamW") and "fused" not in opt and on_gpu:
            opt["fused"] = True   # one launch per step instead of ~30 foreach launches with host gaps in between
        return getattr(importlib.import_module(mod), name)(params=params, **opt)

    def __call__(self, *args, **kwargs):
        # every convolution weight of the model (backbone, score / refinement nets) is laid out for the kernels in ONE
        # launch here; the window closes when the forward returns (weights may change afterwards)
        ME.prepare_conv_weights(self)
        try:
            return super().__call__(*args, **kwargs)
        finally:
            ME.release_conv_weights()

    def forward(self, data_dict):
        out = self.backbone(data_dict["voxel_features"], data_dict["voxel_xyz"], data_dict["voxel_point_map"])
        hook = self.__dict__.pop("_after_backbone", None)
        if hook is not None:
            hook()
        return out

    def schedule_after_backbone(self, fn):
        """Scheduling only: run `fn()` once, in the next forward, right after the backbone has been queued.  The training
        loops put the NEXT batch's coordinate prefetch here: behind the backbone comes the grouping window -- two
        latency-bound chains (ball query -> BFS) that leave most of the chip idle for ~3.5 ms -- so the ~2 ms of
        coordinate kernels (hash insert, kernel maps, pair lists, the Morton sort) run there for free instead of beside
        the GPU-bound backward pass.  MS3D_PREFETCH_AT=backward restores the old placement."""
        self.__dict__["_after_backbone"] = fn

    def schedule_after_grouping(self, fn):
        """Scheduling only: run `fn()` once, in the next forward, right behind the grouping's last host round trip (the
        models call `_after_grouping()` in front of the proposal voxelisation) -- the stretch where the GPU has run dry and
        waits for the interpreter.  MS3D_PREFETCH_AT=proposals puts the next batch's coordinate prefetch there."""
        self.__dict__["_after_grouping_fn"] = fn

    def schedule_after_ballquery(self, fn):
        """Scheduling only: run `fn()` once, in the next forward, right behind the (longer) grouping's ball query -- in
        front of its BFS, a chain of small dependent launches that leaves most of the chip idle (MS3D_PREFETCH_AT=bfs)."""
        self.__dict__["_after_ballquery_fn"] = fn

    def _after_ballquery(self):
        fn = self.__dict__.pop("_after_ballquery_fn", None)
        if fn is not None:
            fn()

    def _after_grouping(self):
        fn = self.__dict__.pop("_after_grouping_fn", None)
        if fn is not None:
            fn()

    def _queue_point_losses(self, data_dict, output_dict):
        """Scheduling only: the per-point losses need nothing but the backbone's outputs, so the grouping models put their
        ~30 small launches on the stream right after the backbone -- where the interpreter is milliseconds ahead of the
        GPU -- instead of after the ScoreNet, where the GPU has run dry at the grouping's last host round trip and waits
        for every launch.  `_loss` picks the result up."""
        if torch.is_grad_enabled() and "sem_labels" in data_dict and "instance_center_xyz" in data_dict:
            side = output_dict.get("_heads_stream")
            if side is not None:
                # MS3D_EARLY_HEADS=2: the heads ran on their own stream (Backbone.forward); the losses and the heads'
                # backward follow them there, NOW -- the host is milliseconds ahead of the GPU at this point, and the
                # kernels run beside the grouping window's latency-bound chains instead of inside the GPU-bound backward pass
                with torch.cuda.stream(side):
                    output_dict["_point_losses"] = self._point_losses(data_dict, output_dict)
                    self._early_point_backward(data_dict, output_dict, force=True)
                    done = torch.cuda.Event()
                    done.record(side)
                output_dict["_heads_done"] = done
                pack = output_dict.get("_heads_pack")
                if pack is not None:
                    pack["done"] = done
                return
            output_dict["_point_losses"] = self._point_losses(data_dict, output_dict)

    def _early_point_backward(self, data_dict, output_dict, force=False):
        """Scheduling only (round 5): the backward of the per-point heads and losses -- ~1.2 ms of bandwidth-bound kernels
        over all points that nothing in the grouping / proposal branch feeds -- is queued HERE, right behind the grouping's
        last host round trip, where the GPU has run dry and idles ~1.7 ms while the interpreter issues the proposal
        network's ~160 small launches (profiles/r04_step_timeline.txt), instead of at the start of the GPU-bound backward
        pass.  The gradients are handed to autograd by _HandOverGradsFn when the real backward pass arrives; the losses'
        values, the heads' graph (for anybody who differentiates the scores directly) and every parameter gradient are
        what they would have been.
        OFF by default (MS3D_EARLY_HEADS=1 turns it on): measured, it buys nothing (profiles/r05_experiments.txt): the
        heads' backward is ~0.6 ms of kernels issued through ~80 autograd nodes, i.e. ~0.7 ms of interpreter time that moves
        from the GPU-bound backward pass (where the host is milliseconds ahead) into this host-bound stretch -- the step
        stays at 20.0 ms either way.  Kept as a tested option for a host whose launch path is cheaper."""
        losses = output_dict.get("_point_losses")
        pf = output_dict.get("point_features")
        if (losses is None or pf is None or not pf.requires_grad or not torch.is_grad_enabled()
                or "_heads_pack" in output_dict or not (force or os.environ.get("MS3D_EARLY_HEADS", "0") == "1")):
            return
        keys, vals = list(losses.keys()), list(losses.values())
        if any(v.grad_fn is None for v in vals):
            return
        heads = [getattr(self.backbone, n, None) for n in ("semantic_branch", "offset_branch")]
        if any(h is None for h in heads):           # a wrapped / replaced backbone: leave the step to autograd
            return
        params = [p for m in heads for p in m.parameters() if p.requires_grad]
        inputs = [pf] + params
        total = vals[0]
        for v in vals[1:]:
            total = total + v
        grads = torch.autograd.grad(total, inputs, retain_graph=True, allow_unused=True)
        pack = {"losses": vals, "grads": grads, "inputs": inputs, "done": None,
                "recompute": lambda d=data_dict, o=output_dict: self._point_losses(d, o)}
        output_dict["_heads_pack"] = pack
        output_dict["_point_losses"] = dict(zip(keys, _HandOverGradsFn.apply(pack, *inputs)))

    def _loss(self, data_dict, output_dict):
        queued = output_dict.pop("_point_losses", None)
        done = output_dict.pop("_heads_done", None)
        if done is not None and queued is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(done)                     # the loss values were computed on the heads' stream
            for v in queued.values():
                v.record_stream(cur)
        return queued if queued is not None else self._point_losses(data_dict, output_dict)

    def _point_losses(self, data_dict, output_dict):
        scores = output_dict["semantic_scores"]
        if (scores.is_cuda and scores.dtype == torch.float32 and os.environ.get("MS3D_FUSED_LOSS", "1") != "0"
                and data_dict["sem_labels"].dtype == torch.int16 and data_dict["instance_ids"].dtype == torch.int16
                and hasattr(get_backend(), "point_losses_forward")):
            # one autograd node, three launches (csrc/losses.hip) instead of ~70 torch operators forward + backward
            from ..loss.point_losses import point_losses
            sem, norm_l, dir_l = point_losses(scores, output_dict["point_offsets"], data_dict["sem_labels"],
                                              data_dict["instance_center_xyz"], data_dict["point_xyz"],
                                              data_dict["instance_ids"])
            return {"semantic_loss": sem, "offset_norm_loss": norm_l, "offset_dir_loss": dir_l}
        return GeneralModel._point_losses_torch(self, data_dict, output_dict)

    def _point_losses_torch(self, data_dict, output_dict):
        # cross entropy with ignore_index = -1 (reference general_model.py:39-41), written as log-softmax + gather:
        # torch's fused nll_loss forward reduces 10^5..10^6 rows in a single block
        labels = data_dict["sem_labels"].long()
        valid = labels != -1
        logp = torch.log_softmax(output_dict["semantic_scores"], dim=1)
        picked = logp.gather(1, labels.clamp_min(0).unsqueeze(1)).squeeze(1)
        sem = -(picked * valid).sum() / valid.sum().clamp_min(1)
        gt_offsets = data_dict["instance_center_xyz"] - data_dict["point_xyz"]
        norm_l, dir_l = self.offset_criterion(output_dict["point_offsets"], gt_offsets,
                                              valid_mask=data_dict["instance_ids"] != -1)
        return {"semantic_loss": sem, "offset_norm_loss": norm_l, "offset_dir_loss": dir_l}

    def training_step(self, data_dict, idx=0):
        losses = self._loss(data_dict, self(data_dict))
        total = sum(losses.values())
        self.last_losses = {k: (float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in losses.items()} \
            if getattr(self, "log_losses", False) else None
        return total

    def on_train_epoch_end(self, optimizer):
        """cosine decay from decay_start_epoch (reference util/lr_decay.py:7-12)"""
        cfg = self.hparams.cfg.model
        start, total, base, clip = cfg.lr_decay.decay_start_epoch, cfg.trainer.max_epochs, cfg.optimizer.lr, 1e-6
        if self.current_epoch < start:
            return
        lr = clip + 0.5 * (base - clip) * (1 + math.cos(math.pi * (self.current_epoch - start) / (total - start)))
        for g in optimizer.param_groups:
            g["lr"] = lr


def scene_offsets(batch_idxs, n_scenes):
    """i32 [n_scenes+1] start of every scene in a scene-grouped point list -- the reference's
    cumsum(bincount(batch_idxs + 1)) (pointgroup.py:37) without bincount's device->host max() round trip"""
    counts = torch.zeros(n_scenes + 1, dtype=torch.int64, device=batch_idxs.device)
    counts.scatter_add_(0, batch_idxs.long() + 1, torch.ones_like(batch_idxs, dtype=torch.int64))
    return torch.cumsum(counts, dim=0).int()


def proposal_voxel_coords_torch(clusters_idx, clusters_offset, coords, scale, spatial_shape, u1, u2):
    """the reference's expression chain (general_model.py:152-181) in torch operators -> i32 [S, 4] (proposal, x, y, z).
    The semantics the fused device operator (`HipBackend.proposal_voxel_coords`) is tested against, and what CPU
    tensors take (test double backends)."""
    cluster_of = clusters_idx[:, 0].long()
    xyz = coords[clusters_idx[:, 1].long()]
    xyz = xyz - common_ops.sec_mean(xyz.contiguous(), clusters_offset)[cluster_of]
    lo = common_ops.sec_min(xyz.contiguous(), clusters_offset)
    hi = common_ops.sec_max(xyz.contiguous(), clusters_offset)
    # 0.01 keeps the scaled coordinates strictly inside the cube
    c_scale = torch.clamp(1 / ((hi - lo) / spatial_shape).max(1)[0] - 0.01, min=None, max=scale)
    lo, hi = lo * c_scale[:, None], hi * c_scale[:, None]
    xyz = xyz * c_scale[cluster_of][:, None]
    extent = hi - lo
    shift = -lo + torch.clamp(spatial_shape - extent - 0.001, min=0) * u1
    shift = shift + torch.clamp(spatial_shape - extent + 0.001, max=0) * u2
    vox = (xyz + shift[cluster_of]).int()
    return torch.cat((clusters_idx[:, 0].int().unsqueeze(-1), vox), dim=1).contiguous()


def clusters_voxelization(clusters_idx, clusters_offset, feats, coords, scale, spatial_shape, device, rand=None,
                          max_dup=None):
    """Per-proposal recentre / rescale into a `spatial_shape` cube, random placement, integer cast, dedupe into
    voxels (reference general_model.py:152-193).  `rand` = the two U(0,1)^3 draws shared by all proposals
    (injected by parity tests).  `max_dup`: the caller's bound on how many proposals one point can be a
    member of (PointGroup: its two groupings -> 2; scheduling hint for the member gather's backward only).
    -> (SparseTensor over proposal voxels, point->voxel map)"""
    feats = ME.gather_rows(feats, clusters_idx[:, 1].long(), max_dup)
    if rand is not None:
        u = torch.cat((rand[0].reshape(3), rand[1].reshape(3))).to(device=coords.device, dtype=torch.float32)
    else:
        u = torch.rand(6, device=coords.device)       # the reference's two torch.rand(3) draws
    if coords.is_cuda:
        # one library call (4 launches) instead of ~30 elementwise / gather / segment launches between the grouping's
        # last host round trip and the ScoreNet, where the GPU waits for the interpreter
        batched = get_backend().proposal_voxel_coords(clusters_idx.contiguous(), clusters_offset, coords.contiguous(),
                                                      scale, spatial_shape, u)
    else:
        batched = proposal_voxel_coords_torch(clusters_idx, clusters_offset, coords, scale, spatial_shape, u[:3], u[3:])
    voxel_xyz, voxel_feats, _, p2v = ME.utils.sparse_quantize(batched, feats, return_index=True, return_inverse=True,
                                                              device=device.type)
    return ME.SparseTensor(features=voxel_feats, coordinates=voxel_xyz, device=device), p2v


def get_segmented_scores(scores, fg_thresh=1.0, bg_thresh=0.0):
    """1 above fg_thresh, 0 below bg_thresh, linear in between (reference general_model.py:196-213)"""
    k = 1.0 / (fg_thresh - bg_thresh)
    lin = scores * k + bg_thresh / (bg_thresh - fg_thresh)
    out = (scores > fg_thresh).float()
    mid = ~(scores > fg_thresh) & ~(scores < bg_thresh)
    return torch.where(mid, lin, out)
