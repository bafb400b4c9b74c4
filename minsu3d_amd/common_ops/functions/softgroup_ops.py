"""`minsu3d/common_ops/functions/softgroup_ops.py` counterpart: sg_bfs_cluster (:7-37) and the
differentiable global_avg_pool (:40-77)."""
import torch

from ...backend import get_backend


def sg_bfs_cluster(class_numpoint_mean, ball_query_idxs, start_len, threshold, class_id):
    with torch.no_grad():
        out = get_backend().sg_bfs_cluster(class_numpoint_mean, ball_query_idxs, start_len, float(threshold),
                                           int(class_id))
    # CPU tensors in (the reference's call, model/softgroup.py:60-63) -> results on the CPU like the reference's
    return out if start_len.is_cuda else tuple(t.cpu() for t in out)


def sg_bfs_cluster_batched(group_of_point, thr_per_group, ball_query_idxs, start_len):
    """every class of SoftGroup's grouping loop in one call (see include/minsu3d_hip.h); same output contract"""
    with torch.no_grad():
        return get_backend().sg_bfs_cluster_batched(group_of_point, thr_per_group, ball_query_idxs, start_len)


class _GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, proposals_offset):
        ctx.save_for_backward(proposals_offset)
        ctx.sum_npoint = feats.size(0)
        return get_backend().global_avg_pool_fp(feats.contiguous(), proposals_offset)

    @staticmethod
    def backward(ctx, d_out):
        (proposals_offset,) = ctx.saved_tensors
        return get_backend().global_avg_pool_bp(d_out.contiguous(), proposals_offset, ctx.sum_npoint), None


def global_avg_pool(feats, proposals_offset):
    return _GlobalAvgPool.apply(feats, proposals_offset)
