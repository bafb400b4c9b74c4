"""SoftGroup (reference minsu3d/model/softgroup.py:11-183): per-class soft grouping (softmax score > thr ->
ball query on shifted coordinates -> class-relative BFS threshold), proposals capped at max_proposal_num,
TinyUnet refinement with classification / mask-scoring / IoU heads and a global average pool per proposal."""
import torch
import torch.nn as nn

from .. import MinkowskiEngine as ME
from ..common_ops.functions import common_ops, softgroup_ops
from .general_model import GeneralModel, clusters_voxelization, scene_offsets
from .module import TinyUnet
from .module.networks import PointLinear


class SoftGroup(GeneralModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        m = cfg.model.network.m
        self.instance_classes = cfg.data.classes - len(cfg.data.ignore_classes)
        k = self.instance_classes + 1
        self.tiny_unet = TinyUnet(m)
        self.classification_branch = nn.Linear(m, k)
        # (PointLinear = nn.Linear with the same keys; tall-skinny [rows, m] products on the engine's K = 1 path)
        self.mask_scoring_branch = nn.Sequential(PointLinear(m, m), nn.ReLU(inplace=True), PointLinear(m, k))
        self.iou_score = nn.Linear(m, k)
        self.voxelization_rand = None

    def _soft_grouping_loop(self, data_dict, sem_scores, offsets):
        """the reference's formulation: one ball query + BFS per class (kept as the parity reference of the batched path)"""
        cfg = self.hparams.cfg
        net = cfg.model.network
        idx_parts, off_parts, n_prop, n_rows = [], [], 0, 0
        for class_id in range(cfg.data.classes):
            if class_id + 1 in cfg.data.ignore_classes:
                continue
            object_idxs = (sem_scores[:, class_id] > net.grouping_cfg.score_thr).nonzero().view(-1)
            if object_idxs.size(0) < net.test_cfg.min_npoint:
                continue
            batch_idxs = data_dict["vert_batch_ids"][object_idxs]
            batch_offsets = torch.cumsum(torch.bincount(batch_idxs + 1), dim=0).int()
            shifted = (data_dict["point_xyz"][object_idxs] + offsets[object_idxs]).detach().contiguous()
            idx, start_len = common_ops.ballquery_batch_p(shifted, batch_idxs, batch_offsets, net.grouping_cfg.radius,
                                                          net.grouping_cfg.mean_active)
            p_idx, p_off = softgroup_ops.sg_bfs_cluster(cfg.data.point_num_avg, idx, start_len,
                                                        net.grouping_cfg.npoint_thr, class_id)
            if p_idx.size(0) == 0:
                continue
            p_idx = p_idx.long()
            p_idx[:, 1] = object_idxs[p_idx[:, 1]]
            p_idx[:, 0] += n_prop                       # proposals are numbered across classes, in class order
            idx_parts.append(p_idx)
            off_parts.append(p_off + n_rows if not off_parts else (p_off + n_rows)[1:])
            n_prop += p_off.numel() - 1
            n_rows += p_idx.size(0)
        if not idx_parts:                                # the reference raises on torch.cat([]) here
            dev = sem_scores.device
            return torch.zeros((0, 2), dtype=torch.long, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        return self._cap(torch.cat(idx_parts, 0), torch.cat(off_parts).int())

    def _cap(self, proposals_idx, proposals_offset):
        cap = self.hparams.cfg.model.network.train_cfg.max_proposal_num
        if proposals_offset.numel() > cap:               # reference softgroup.py:80-83 (keeps `cap` proposals)
            proposals_offset = proposals_offset[:cap + 1]
            proposals_idx = proposals_idx[:int(proposals_offset[-1])]
        return proposals_idx, proposals_offset.contiguous()

    def _soft_grouping(self, data_dict, sem_scores, offsets):
        """All classes at once: the (class, point) pairs above the score threshold are laid out class-major, ball-queried
        with group id = class*B + scene and clustered by ONE order-exact BFS with per-group thresholds.  Seeds ascend,
        so the clusters arrive in exactly the order the per-class loop concatenates them."""
        cfg = self.hparams.cfg
        net = cfg.model.network
        dev = sem_scores.device
        C_ = cfg.data.classes
        mask = sem_scores > net.grouping_cfg.score_thr                      # [N, C]
        for cls in cfg.data.ignore_classes:
            mask[:, cls - 1] = False
        mask &= (mask.sum(0) >= net.test_cfg.min_npoint)[None, :]           # classes with too few points are skipped
        # one host round trip for two numbers: the scene count, and in how many classes' candidate sets a point sits at most
        # (= the most proposals one point can be a member of: the member gather's backward then knows whether its float
        # atomics are order-independent -- at most two addends per row -- or a sorted index is needed)
        if data_dict["vert_batch_ids"].numel():
            two = torch.stack((data_dict["vert_batch_ids"].max().long(), mask.sum(1).max())).tolist()
            B, self._max_member_dup = int(two[0]) + 1, int(two[1])
        else:
            B, self._max_member_dup = 1, 1
        if C_ * B > 255:                                  # group ids are uint8 like the reference's batch ids
            self._max_member_dup = None
            return self._soft_grouping_loop(data_dict, sem_scores, offsets)
        cls_id, pt = mask.t().nonzero(as_tuple=True)                        # class-major, points ascending inside
        if pt.numel() == 0:
            return torch.zeros((0, 2), dtype=torch.long, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        group = (cls_id * B + data_dict["vert_batch_ids"][pt].long()).to(torch.uint8)
        group_offsets = scene_offsets(group, C_ * B)
        shifted = (data_dict["point_xyz"][pt] + offsets[pt]).detach().contiguous()
        idx, start_len = common_ops.ballquery_batch_p(shifted, group, group_offsets, net.grouping_cfg.radius,
                                                      net.grouping_cfg.mean_active)
        mean = torch.tensor(cfg.data.point_num_avg, dtype=torch.float32, device=dev)
        thr_cls = torch.where(mean == -1, torch.full_like(mean, net.grouping_cfg.npoint_thr),
                              mean * net.grouping_cfg.npoint_thr)           # bfs_cluster.cpp:113-120
        thr_group = thr_cls.repeat_interleave(B).contiguous()
        p_idx, p_off = softgroup_ops.sg_bfs_cluster_batched(group, thr_group, idx, start_len)
        p_idx = p_idx.long()
        p_idx[:, 1] = pt[p_idx[:, 1]]
        return self._cap(p_idx, p_off)

    def forward(self, data_dict):
        out = super().forward(data_dict)
        self._queue_point_losses(data_dict, out)
        net = self.hparams.cfg.model.network
        if self.current_epoch <= net.prepare_epochs:
            return out
        sem_scores = data_dict.get("grouping_semantic_scores")
        if sem_scores is None:
            sem_scores = out["semantic_scores"].softmax(dim=-1)
        offsets = data_dict.get("grouping_point_offsets")
        if offsets is None:
            offsets = out["point_offsets"]
        proposals_idx, proposals_offset = self._soft_grouping(data_dict, sem_scores, offsets)
        out["proposals_idx"], out["proposals_offset"] = proposals_idx, proposals_offset
        self._after_grouping()                                 # scheduled work (MS3D_PREFETCH_AT=proposals), also when nothing was grouped
        if proposals_offset.numel() <= 1:
            return out
        self._early_point_backward(data_dict, out)     # fills the GPU while the proposal branch is being issued
        vox, p2v = clusters_voxelization(proposals_idx, proposals_offset, out["point_features"], data_dict["point_xyz"],
                                         net.instance_voxel_cfg.scale, net.instance_voxel_cfg.spatial_shape, self.device,
                                         rand=self.voxelization_rand, max_dup=getattr(self, "_max_member_dup", None))
        feats = self.tiny_unet(vox)
        out["mask_scores"] = ME.gather_rows(self.mask_scoring_branch(feats.features), p2v)   # (fixed-order backward, as HAIS)
        out["instance_batch_idxs"] = feats.coordinates[:, 0][p2v]
        pooled = self.global_pool(feats)
        out["cls_scores"] = self.classification_branch(pooled)
        out["iou_scores"] = self.iou_score(pooled)
        return out

    def global_pool(self, x):
        """mean over each proposal's voxels (rows of one proposal are contiguous: quantize keeps first-occurrence order)"""
        owner = x.coordinates[:, 0]
        offsets = torch.cumsum(torch.bincount(owner + 1), dim=0).int()
        return softgroup_ops.global_avg_pool(x.features.contiguous(), offsets)

    def _loss(self, data_dict, output_dict):
        losses = super()._loss(data_dict, output_dict)
        if "cls_scores" not in output_dict:
            return losses
        net = self.hparams.cfg.model.network
        K = self.instance_classes
        pidx = output_dict["proposals_idx"][:, 1].int().contiguous()
        poff = output_dict["proposals_offset"]
        ious_cluster = common_ops.get_mask_iou_on_cluster(pidx, poff, data_dict["instance_ids"],
                                                          data_dict["instance_num_point"])
        fg = data_dict["instance_semantic_cls"] != -1
        fg_cls = data_dict["instance_semantic_cls"][fg]
        fg_ious = ious_cluster[:, fg]
        n = fg_ious.size(0)
        # proposal -> ground-truth assignment: positive when the best foreground IoU reaches pos_iou_thr
        labels = fg_cls.new_full((n,), K).long()
        if fg_ious.size(1) > 0:
            best, arg = fg_ious.max(1)
            pos = best >= net.train_cfg.pos_iou_thr
            labels[pos] = fg_cls[arg[pos]].long()
        losses["classification_loss"] = nn.functional.cross_entropy(output_dict["cls_scores"], labels)

        point_label = labels[output_dict["instance_batch_idxs"].long()]
        rows = torch.arange(point_label.size(0), device=point_label.device)
        sig = output_dict["mask_scores"].sigmoid()[rows, point_label]
        mlabel, mmask = common_ops.get_mask_label(pidx, poff, data_dict["instance_ids"],
                                                  data_dict["instance_semantic_cls"], data_dict["instance_num_point"],
                                                  ious_cluster, -1, net.train_cfg.pos_iou_thr)
        mloss = nn.functional.binary_cross_entropy(sig, mlabel.float(), weight=mmask.float(), reduction="sum")
        losses["mask_scoring_loss"] = mloss / (torch.count_nonzero(mmask) + 1)

        ious_pred = common_ops.get_mask_iou_on_pred(pidx, poff, data_dict["instance_ids"],
                                                    data_dict["instance_num_point"], sig.detach().contiguous())
        prop = torch.arange(n, device=labels.device)
        w = labels < K
        target = ious_pred[:, fg].max(1)[0] if fg_ious.size(1) > 0 else ious_pred.new_zeros(n)
        err = nn.functional.mse_loss(output_dict["iou_scores"][prop, labels], target, reduction="none")
        losses["iou_scoring_loss"] = err[w].sum() / (w.count_nonzero() + 1)
        return losses

    def _get_pred_instances(self, scan_id, gt_xyz, proposals_idx, num_points, cls_scores, iou_scores, mask_scores,
                            num_ignored_classes):
        """same name and arguments as the reference (softgroup.py:269-270); tensors may stay on the device"""
        from .postprocess import softgroup_instances
        t = self.hparams.cfg.model.network.test_cfg
        return softgroup_instances(scan_id, gt_xyz, proposals_idx, num_points, cls_scores, iou_scores, mask_scores,
                                   num_ignored_classes, self.instance_classes, t.cls_score_thr, t.mask_score_thr,
                                   t.min_npoint)
