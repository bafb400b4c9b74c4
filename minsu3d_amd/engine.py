"""Train / validate / checkpoint loop -- the part of the reference that PyTorch-Lightning + its callbacks provide
(`config/model/base.yaml:11-28`, `train.py:17-41`, `general_model.py:56-110`, per-model `validation_step`s):
epochs over the train loader, cosine decay at every epoch end, validation + `epoch=<n>.ckpt` every
`check_val_every_n_epoch` epochs, resume from a checkpoint.

Checkpoints use Lightning's top-level keys (`epoch`, `global_step`, `state_dict`, `optimizer_states`), so a reference
`.ckpt` (README.md:146-151) loads through `load_checkpoint` as well: parameter names follow the reference modules
(SURVEY Appendix D)."""
import inspect
import os

import numpy as np
import torch

from . import MinkowskiEngine as ME

from .evaluation import (GeneralDatasetEvaluator, evaluate_bbox_acc, evaluate_semantic_accuracy,
                         evaluate_semantic_miou, get_gt_bbox, get_gt_instances)


def save_checkpoint(path, model, optimizer, epoch, global_step):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save({"epoch": epoch, "global_step": global_step, "state_dict": model.state_dict(),
                "optimizer_states": [optimizer.state_dict()] if optimizer is not None else []}, path)


def load_checkpoint(path, model, optimizer=None, strict=True):
    """-> (epoch, global_step) stored in the file; accepts Lightning checkpoints and bare state_dicts"""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    state = ck.get("state_dict", ck)
    own = model.state_dict()
    fixed = {}
    for k, v in state.items():
        if k in own and own[k].shape != v.shape and own[k].numel() == v.numel():
            v = v.reshape(own[k].shape)          # e.g. a kernel_size-1 convolution stored as [Cin, Cout]
        fixed[k] = v
    model.load_state_dict(fixed, strict=strict)
    if optimizer is not None and ck.get("optimizer_states"):
        optimizer.load_state_dict(ck["optimizer_states"][0])
    return int(ck.get("epoch", -1)), int(ck.get("global_step", 0))


def predicted_instances(model, data_dict, output_dict):
    """the per-model argument plumbing of the reference's validation steps (pointgroup.py:136-148, hais.py:152-161,
    softgroup.py:212-219); tensors stay on the device"""
    cfg = model.hparams.cfg
    scan_id, xyz, n_ign = data_dict["scan_ids"][0], data_dict["point_xyz"], len(cfg.data.ignore_classes)
    name = type(model).__name__
    if name == "PointGroup":
        scores, idx, off = output_dict["proposal_scores"]
        return model._get_pred_instances(scan_id, xyz, scores, idx, off.size(0) - 1, output_dict["semantic_scores"], n_ign)
    if name == "HAIS":
        scores, idx, off, mask_scores = output_dict["proposal_scores"]
        return model._get_pred_instances(scan_id, xyz, scores, idx, off.size(0) - 1, mask_scores,
                                         output_dict["semantic_scores"], n_ign)
    if "cls_scores" not in output_dict:          # no proposal survived the grouping
        return []
    return model._get_pred_instances(scan_id, xyz, output_dict["proposals_idx"], output_dict["semantic_scores"].size(0),
                                     output_dict["cls_scores"], output_dict["iou_scores"], output_dict["mask_scores"], n_ign)


def _dist():
    d = torch.distributed
    return d if d.is_available() and d.is_initialized() and d.get_world_size() > 1 else None


class Trainer:
    """One process per GPU (torchrun): the model is wrapped in DistributedDataParallel, every rank trains on its own
    size-matched shard of each step's scenes (data_module.py), validation is sharded too and reduced on rank 0, which
    also writes the checkpoints -- the reference's `strategy: ddp` (config/model/base.yaml:13-16)."""

    def __init__(self, cfg, model, datamodule, out_dir=None, log=print):
        self.cfg, self.model, self.dm, self.log = cfg, model, datamodule, log
        self.out_dir = out_dir or os.path.join(cfg.exp_output_root_path, "training")
        self.optimizer = model.configure_optimizers()
        self.global_step = 0
        self.history = []
        self.rank = _dist().get_rank() if _dist() else 0
        from .parallel import wrap_ddp
        dev = next(model.parameters()).device
        self.ddp = wrap_ddp(model, dev, find_unused_parameters=True)   # the score branch idles until prepare_epochs

    def validate(self):
        model, cfg = self.model, self.cfg
        dist = _dist()
        if dist:
            from .parallel import sync_buffers
            sync_buffers(model)          # every rank evaluates with rank 0's BatchNorm statistics
        model.eval()
        losses, acc, miou, preds, gts, boxes = [], [], [], [], [], []
        loader = self.dm.val_dataloader()
        sampler = getattr(loader, "batch_sampler", None)
        # wrap-around repeats that pad the rank shards to equal length are run (every rank does the same number of
        # steps) but not evaluated: a scan counts once in AP / mIoU
        repeats = sampler.padded_positions() if hasattr(sampler, "padded_positions") else None
        with torch.no_grad():
            for k, batch in enumerate(loader):
                out = model(batch)
                if repeats is not None and repeats[k]:
                    continue
                losses.append(float(sum(model._loss(batch, out).values())))
                sem_pred = out["semantic_scores"].max(1)[1]
                acc.append(evaluate_semantic_accuracy(sem_pred, batch["sem_labels"], ignore_label=-1))
                miou.append(evaluate_semantic_miou(sem_pred, batch["sem_labels"], ignore_label=-1))
                if model.current_epoch > cfg.model.network.prepare_epochs:
                    inst = predicted_instances(model, batch, out)
                    if inst:
                        xyz = batch["point_xyz"].cpu().numpy()
                        ids = batch["instance_ids"].cpu()
                        sem = batch["sem_labels"].cpu()
                        preds.append(inst)
                        boxes.append(get_gt_bbox(xyz, ids.numpy(), sem.numpy(), -1, cfg.data.ignore_classes))
                        gts.append(get_gt_instances(sem.clone(), ids.clone(), cfg.data.ignore_classes))
        if dist:      # the shards of the validation split meet on every rank (small host objects: RLE masks, boxes)
            parts = [None] * dist.get_world_size()
            dist.all_gather_object(parts, (losses, acc, miou, preds, gts, boxes))
            losses, acc, miou, preds, gts, boxes = ([x for p in parts for x in p[i]] for i in range(6))
        res = {"val/total_loss": float(np.mean(losses)), "val_eval/semantic_accuracy": float(np.mean(acc)),
               "val_eval/semantic_mean_iou": float(np.mean(miou))}
        if preds:
            ev = GeneralDatasetEvaluator(cfg.data.class_names, -1, cfg.data.ignore_classes)
            r = ev.evaluate(preds, gts, print_result=False)
            res.update({"val_eval/AP": float(r["all_ap"]), "val_eval/AP 50%": float(r["all_ap_50%"]),
                        "val_eval/AP 25%": float(r["all_ap_25%"])})
            with np.errstate(invalid="ignore"):
                b = evaluate_bbox_acc(preds, boxes, cfg.data.class_names, cfg.data.ignore_classes, print_result=False)
            res.update({"val_eval/BBox AP 25%": float(b["all_bbox_ap_0.25"]["avg"]),
                        "val_eval/BBox AP 50%": float(b["all_bbox_ap_0.5"]["avg"])})
        model.train()
        return res

    def fit(self, max_epochs=None, ckpt_path=None):
        cfg, model, opt = self.cfg, self.model, self.optimizer
        max_epochs = max_epochs or cfg.model.trainer.max_epochs
        every = cfg.model.trainer.check_val_every_n_epoch
        start = 0
        if ckpt_path:
            epoch, self.global_step = load_checkpoint(ckpt_path, model, opt)
            start = epoch + 1
            model.current_epoch = epoch
            model.on_train_epoch_end(opt)          # the decayed rate of the epoch that was just finished
        model.train()
        for epoch in range(start, max_epochs):
            model.current_epoch = epoch
            total, n = 0.0, 0
            if "epoch" in inspect.signature(self.dm.train_dataloader).parameters:
                loader = self.dm.train_dataloader(epoch=epoch)     # rank-sharded samplers reshuffle per epoch
            else:
                loader = self.dm.train_dataloader()
            batches = iter(loader)
            batch = next(batches, None)
            while batch is not None:
                upcoming = next(batches, None)      # collated (on the GPU) one step ahead
                opt.zero_grad(set_to_none=True)
                if upcoming is not None and torch.is_tensor(upcoming.get("voxel_xyz")):
                    # coordinate-only structures of the next batch: queued right behind this step's backbone, i.e. they
                    # are built inside the grouping window (GeneralModel.schedule_after_backbone)
                    model.schedule_after_backbone(
                        lambda nb=upcoming: ME.prefetch_coordinates(nb["voxel_xyz"], model.backbone.n_levels,
                                                                    channels=model.backbone.level_channels,
                                                                    point_map=nb.get("voxel_point_map")))
                if self.ddp is model:
                    loss = model.training_step(batch)
                else:                               # through the DDP wrapper: it arms the gradient all-reduce
                    loss = sum(model._loss(batch, self.ddp(batch)).values())
                loss.backward()
                opt.step()
                batch = upcoming
                self.global_step += 1
                total += float(loss.detach())
                n += 1
            model.on_train_epoch_end(opt)
            rec = {"epoch": epoch, "train/total_loss": total / max(n, 1), "lr": opt.param_groups[0]["lr"]}
            if (epoch + 1) % every == 0:
                rec.update(self.validate())
                if self.rank == 0:
                    save_checkpoint(os.path.join(self.out_dir, f"epoch={epoch}.ckpt"), model, opt, epoch, self.global_step)
                if _dist():
                    _dist().barrier()
            self.history.append(rec)
            if self.rank == 0:
                self.log(rec)
        return self.history
