"""PointGroup (reference minsu3d/model/pointgroup.py:12-110): dual-set point grouping (original and
offset-shifted coordinates) -> proposal voxelisation -> ScoreNet -> RoI max-pool -> score.
Everything from the ball query to the clusters stays on the device (the reference round-trips through
host memory for its serial BFS, pointgroup.py:41-66)."""
import os

import torch
import torch.nn as nn

from .. import MinkowskiEngine as ME
from ..backend import get_backend, side_stream as _side_stream, worker as _worker
from ..common_ops.functions import common_ops, pointgroup_ops
from .general_model import GeneralModel, clusters_voxelization, get_segmented_scores, scene_offsets
from .module import TinyUnet


class PointGroup(GeneralModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        m = cfg.model.network.m
        self.score_net = TinyUnet(m)
        self.score_branch = nn.Linear(m, 1)
        self.voxelization_rand = None   # tests inject the two uniform draws here

    def _group(self, xyz, batch_idxs, batch_offsets, sem_fg, object_idxs, mean_active, hook=False):
        net = self.hparams.cfg.model.network.cluster
        idx, start_len = common_ops.ballquery_batch_p(xyz, batch_idxs, batch_offsets, net.cluster_radius, mean_active)
        if hook:
            self._after_ballquery()
        prop_idx, prop_off = pointgroup_ops.pg_bfs_cluster(sem_fg, idx, start_len, net.cluster_npoint_thre)
        prop_idx = prop_idx.long()
        prop_idx[:, 1] = object_idxs[prop_idx[:, 1]]          # foreground index -> global point index
        return prop_idx, prop_off

    def forward(self, data_dict):
        out = super().forward(data_dict)
        self._queue_point_losses(data_dict, out)
        cfg = self.hparams.cfg
        net = cfg.model.network
        if self.current_epoch <= net.prepare_epochs:
            return out
        # Synthetic benchmarks run a randomly initialised network, whose predictions group into nothing; they
        # may supply the grouping INPUTS a trained network would produce (labels / offsets near the ground
        # truth).  Every operator still runs, and the losses still use the network's own outputs.
        sem_pred = data_dict.get("grouping_semantic_preds")
        if sem_pred is None:
            sem_pred = out["semantic_scores"].argmax(1).to(torch.int16)
        grouping_offsets = data_dict.get("grouping_point_offsets")
        if grouping_offsets is None:
            grouping_offsets = out["point_offsets"]
        fg = torch.ones_like(sem_pred, dtype=torch.bool)
        for cls in cfg.data.ignore_classes:                    # floor / wall never form instances
            fg &= sem_pred != (cls - 1)
        object_idxs = torch.nonzero(fg).view(-1)
        batch_idxs = data_dict["vert_batch_ids"][object_idxs]
        batch_offsets = scene_offsets(batch_idxs, len(data_dict["scan_ids"]))
        xyz = data_dict["point_xyz"][object_idxs]
        shifted = (xyz + grouping_offsets[object_idxs]).detach().contiguous()
        sem_fg = sem_pred[object_idxs].contiguous()

        xyz = xyz.contiguous()
        if xyz.is_cuda and os.environ.get("MS3D_GROUP_STREAMS", "1") != "0":
            # The two groupings are independent, and each sizes its outputs on the host twice (ball-query total, cluster
            # count): run the second one from a worker thread on a side stream, so that one grouping's kernels fill the
            # other's host round trips (the library calls release the GIL; scratch buffers are per stream).
            main = torch.cuda.current_stream()
            side = _side_stream(xyz.device)
            timer = getattr(get_backend(), "kernel_timer", None)   # bench.py: wall span of the two concurrent groupings
            span0 = timer.op_begin() if timer is not None else None
            side.wait_stream(main)

            def second():
                with torch.cuda.stream(side), torch.no_grad():
                    return self._group(xyz, batch_idxs, batch_offsets, sem_fg, object_idxs, net.cluster.cluster_meanActive)

            pending = _worker().submit(second)
            try:
                p_shift, o_shift = self._group(shifted, batch_idxs, batch_offsets, sem_fg, object_idxs,
                                               net.cluster.cluster_shift_meanActive, hook=True)
            finally:
                p_orig, o_orig = pending.result()
                main.wait_stream(side)
            if span0 is not None:
                timer.op_end("grouping span (both ball queries + both BFS, two streams, one interval)", span0, 0)
            p_orig.record_stream(main)
            o_orig.record_stream(main)
        else:
            p_shift, o_shift = self._group(shifted, batch_idxs, batch_offsets, sem_fg, object_idxs,
                                           net.cluster.cluster_shift_meanActive, hook=True)
            p_orig, o_orig = self._group(xyz, batch_idxs, batch_offsets, sem_fg, object_idxs,
                                         net.cluster.cluster_meanActive)
        p_shift[:, 0] += o_orig.size(0) - 1                    # renumber the second proposal set after the first
        proposals_idx = torch.cat((p_orig, p_shift), dim=0)
        proposals_offset = torch.cat((o_orig, o_shift[1:] + o_orig[-1]))

        self._after_grouping()                                 # scheduled work (MS3D_PREFETCH_AT=proposals), also when nothing was grouped
        if proposals_offset.numel() <= 1:                      # nothing grouped (the reference would crash here)
            out["proposal_scores"] = (out["point_features"].new_zeros((0, 1)), proposals_idx, proposals_offset)
            return out
        self._early_point_backward(data_dict, out)     # fills the GPU while the proposal branch is being issued
        vox, p2v = clusters_voxelization(proposals_idx, proposals_offset, out["point_features"],
                                         data_dict["point_xyz"], net.score_scale, net.score_fullscale, self.device,
                                         rand=self.voxelization_rand, max_dup=2)   # a point: <= one cluster per grouping
        # (max_dup: the gradient that comes back over this gather is roipool's -- ONE non-zero entry per proposal and channel,
        # a proposal's points only name that proposal's voxels -- so no element of the voxel gradient collects more than one
        # non-zero addend: the one-launch scatter-add is order-independent here, no sorted index is needed)
        score_feats = ME.gather_rows(self.score_net(vox).features, p2v, max_dup=1)      # (sumNPoint, m)
        pooled = common_ops.roipool(score_feats, proposals_offset)            # (nProposal, m)
        out["proposal_scores"] = (self.score_branch(pooled), proposals_idx, proposals_offset)
        return out

    def _loss(self, data_dict, output_dict):
        losses = super()._loss(data_dict, output_dict)
        if "proposal_scores" in output_dict:
            net = self.hparams.cfg.model.network
            scores, proposals_idx, proposals_offset = output_dict["proposal_scores"]
            if proposals_offset.numel() > 1:
                ious = common_ops.get_iou(proposals_idx[:, 1].int().contiguous(), proposals_offset,
                                          data_dict["instance_ids"], data_dict["instance_num_point"])
                target = get_segmented_scores(ious.max(1)[0], net.fg_thresh, net.bg_thresh)
                losses["score_loss"] = nn.functional.binary_cross_entropy_with_logits(scores.view(-1), target)
            else:
                losses["score_loss"] = scores.sum() * 0
        return losses

    def _get_pred_instances(self, scan_id, gt_xyz, proposals_scores, proposals_idx, num_proposals, semantic_scores,
                            num_ignored_classes):
        """same name and arguments as the reference (pointgroup.py:197-199); tensors may stay on the device"""
        from .postprocess import pointgroup_instances
        t = self.hparams.cfg.model.network.test
        return pointgroup_instances(scan_id, gt_xyz, proposals_scores, proposals_idx, num_proposals, semantic_scores,
                                    num_ignored_classes, t.TEST_SCORE_THRESH, t.TEST_NPOINT_THRESH, t.TEST_NMS_THRESH)
