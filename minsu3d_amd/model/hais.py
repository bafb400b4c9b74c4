"""HAIS (reference minsu3d/model/hais.py:12-128): shifted-coordinate ball query -> hierarchical aggregation
(connected components, class-relative split, optional set aggregation) -> proposal voxelisation -> TinyUnet ->
per-point mask branch + RoI-pooled score branch.  The grouping stays on the device end to end."""
import torch
import torch.nn as nn

from .. import MinkowskiEngine as ME
from ..common_ops.functions import common_ops, hais_ops
from .general_model import GeneralModel, clusters_voxelization, get_segmented_scores, scene_offsets
from .module import TinyUnet
from .module.networks import PointLinear


class HAIS(GeneralModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        m = cfg.model.network.m
        self.tiny_unet = TinyUnet(m)
        self.score_branch = nn.Linear(m, 1)
        # (PointLinear = nn.Linear with the same keys; tall-skinny [rows, m] products on the engine's K = 1 path)
        self.mask_branch = nn.Sequential(PointLinear(m, m), nn.ReLU(inplace=True), PointLinear(m, 1))
        self.voxelization_rand = None

    def forward(self, data_dict):
        out = super().forward(data_dict)
        self._queue_point_losses(data_dict, out)
        cfg = self.hparams.cfg
        net = cfg.model.network
        if self.current_epoch <= net.prepare_epochs:
            return out
        sem_pred = data_dict.get("grouping_semantic_preds")
        if sem_pred is None:
            sem_pred = out["semantic_scores"].argmax(1).to(torch.int16)
        offsets = data_dict.get("grouping_point_offsets")
        if offsets is None:
            offsets = out["point_offsets"]
        fg = torch.ones_like(sem_pred, dtype=torch.bool)
        for cls in cfg.data.ignore_classes:
            fg &= sem_pred != (cls - 1)
        object_idxs = torch.nonzero(fg).view(-1)
        batch_idxs = data_dict["vert_batch_ids"][object_idxs]
        batch_offsets = scene_offsets(batch_idxs, len(data_dict["scan_ids"]))
        shifted = (data_dict["point_xyz"][object_idxs] + offsets[object_idxs]).detach().contiguous()
        idx, start_len = common_ops.ballquery_batch_p(shifted, batch_idxs, batch_offsets, net.point_aggr_radius,
                                                      net.cluster_shift_meanActive)
        set_aggr = net.using_set_aggr_in_training if self.training else net.using_set_aggr_in_testing
        proposals_idx, proposals_offset = hais_ops.hierarchical_aggregation(
            sem_pred[object_idxs].contiguous(), shifted, idx, start_len, batch_idxs, set_aggr,
            cfg.data.point_num_avg, cfg.data.radius_avg, -1)
        proposals_idx = proposals_idx.long()
        proposals_idx[:, 1] = object_idxs[proposals_idx[:, 1]]
        self._after_grouping()                                 # scheduled work (MS3D_PREFETCH_AT=proposals), also when nothing was grouped
        if proposals_offset.numel() <= 1:
            z = out["point_features"].new_zeros((0, 1))
            out["proposal_scores"] = (z, proposals_idx, proposals_offset, z)
            return out
        self._early_point_backward(data_dict, out)     # fills the GPU while the proposal branch is being issued
        vox, p2v = clusters_voxelization(proposals_idx, proposals_offset, out["point_features"], data_dict["point_xyz"],
                                         net.score_scale, net.score_fullscale, self.device, rand=self.voxelization_rand,
                                         max_dup=1)                 # a point belongs to at most one aggregated cluster
        inst = self.tiny_unet(vox)
        # (two gathers over ONE index: the engine's row gather, whose backward sums in a fixed order over a sort of p2v that is
        # queued once, now, off the critical path -- torch's indexing would sort the index again inside each backward)
        score_feats = ME.gather_rows(inst.features, p2v)
        mask_scores = ME.gather_rows(self.mask_branch(inst.features), p2v)   # linear on voxels first, then voxel -> point
        if self.current_epoch > net.use_mask_filter_score_feature_start_epoch:
            keep = (torch.sigmoid(mask_scores) >= net.mask_filter_score_feature_thre).to(score_feats.dtype)
            score_feats = score_feats * keep
        scores = self.score_branch(common_ops.roipool(score_feats.contiguous(), proposals_offset))
        out["proposal_scores"] = (scores, proposals_idx, proposals_offset, mask_scores)
        return out

    def _loss(self, data_dict, output_dict):
        losses = super()._loss(data_dict, output_dict)
        if "proposal_scores" not in output_dict:
            return losses
        net = self.hparams.cfg.model.network
        scores, proposals_idx, proposals_offset, mask_scores = output_dict["proposal_scores"]
        if proposals_offset.numel() <= 1:
            losses["mask_loss"] = losses["score_loss"] = scores.sum() * 0
            return losses
        sig = torch.sigmoid(mask_scores)
        pidx = proposals_idx[:, 1].int().contiguous()
        if self.current_epoch > net.cal_iou_based_on_mask_start_epoch:
            ious = common_ops.get_mask_iou_on_pred(pidx, proposals_offset, data_dict["instance_ids"],
                                                   data_dict["instance_num_point"], sig.detach().view(-1))
        else:
            ious = common_ops.get_mask_iou_on_cluster(pidx, proposals_offset, data_dict["instance_ids"],
                                                      data_dict["instance_num_point"])
        label, label_mask = common_ops.get_mask_label(pidx, proposals_offset, data_dict["instance_ids"],
                                                      data_dict["instance_semantic_cls"],
                                                      data_dict["instance_num_point"], ious, -1, 0.5)
        losses["mask_loss"] = nn.functional.binary_cross_entropy(sig, label.unsqueeze(1).float(),
                                                                 weight=label_mask.unsqueeze(1).float(),
                                                                 reduction="mean")
        target = get_segmented_scores(ious.max(1)[0], net.fg_thresh, net.bg_thresh)
        losses["score_loss"] = nn.functional.binary_cross_entropy_with_logits(scores.view(-1), target)
        return losses

    def _get_pred_instances(self, scan_id, gt_xyz, scores, proposals_idx, num_proposals, mask_scores, semantic_scores,
                            num_ignored_classes):
        """same name and arguments as the reference (hais.py:210); tensors may stay on the device"""
        from .postprocess import hais_instances
        t = self.hparams.cfg.model.network.test
        return hais_instances(scan_id, gt_xyz, scores, proposals_idx, num_proposals, mask_scores, semantic_scores,
                              num_ignored_classes, t.test_mask_score_thre, t.TEST_SCORE_THRESH, t.TEST_NPOINT_THRESH)
