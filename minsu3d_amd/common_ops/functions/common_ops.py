"""Operator surface of the reference's `minsu3d/common_ops/functions/common_ops.py`, served by the
HIP backend (minsu3d_amd/backend.py -> libminsu3d_hip.so).  Same names, argument order and return
values; non-differentiable ops simply return tensors that do not require grad.

  ballquery_batch_p   reference :11-47      sec_mean / sec_min / sec_max   :50-133
  roipool             :136-173 (autograd)   get_iou                        :176-209
  get_mask_iou_on_cluster :212-246          get_mask_iou_on_pred           :249-287
  get_mask_label      :290-330
"""
import torch

from ...backend import get_backend


def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, meanActive):
    """coords f32[n,3], batch_idxs u8[n], batch_offsets i32[B+1] -> (idx i32[nActive], start_len i32[n,2]).
    Lists ascending, self included, at most 1000 per point; start = prefix sum of len (canonical form)."""
    assert coords.is_contiguous() and batch_idxs.is_contiguous() and batch_offsets.is_contiguous()
    with torch.no_grad():
        return get_backend().ballquery_batch_p(coords.detach(), batch_idxs, batch_offsets, float(radius),
                                               int(meanActive))


def sec_mean(inp, offsets):
    with torch.no_grad():
        return get_backend().sec_mean(inp.detach(), offsets)


def sec_min(inp, offsets):
    with torch.no_grad():
        return get_backend().sec_min(inp.detach(), offsets)


def sec_max(inp, offsets):
    with torch.no_grad():
        return get_backend().sec_max(inp.detach(), offsets)


class _RoiPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, proposals_offset):
        out, maxidx = get_backend().roipool_fp(feats.contiguous(), proposals_offset)
        ctx.save_for_backward(maxidx, proposals_offset)
        ctx.sum_npoint = feats.size(0)
        return out

    @staticmethod
    def backward(ctx, d_out):
        maxidx, proposals_offset = ctx.saved_tensors
        return get_backend().roipool_bp(d_out.contiguous(), proposals_offset, maxidx, ctx.sum_npoint), None


def roipool(feats, proposals_offset):
    """segment max over proposal rows, differentiable (argmax scatter in backward)"""
    return _RoiPool.apply(feats, proposals_offset)


def get_iou(proposals_idx, proposals_offset, instance_ids, instance_pointnum):
    with torch.no_grad():
        return get_backend().get_iou(proposals_idx, proposals_offset, instance_ids, instance_pointnum)


def get_mask_iou_on_cluster(proposals_idx, proposals_offset, instance_labels, instance_pointnum):
    with torch.no_grad():
        return get_backend().get_mask_iou_on_cluster(proposals_idx, proposals_offset, instance_labels,
                                                     instance_pointnum)


def get_mask_iou_on_pred(proposals_idx, proposals_offset, instance_labels, instance_pointnum, mask_scores_sigmoid):
    with torch.no_grad():
        return get_backend().get_mask_iou_on_pred(proposals_idx, proposals_offset, instance_labels,
                                                  instance_pointnum, mask_scores_sigmoid.detach().contiguous())


def get_mask_label(proposals_idx, proposals_offset, instance_ids, instance_cls, instance_pointnum, proposals_iou,
                   ignored_label, iou_thr):
    with torch.no_grad():
        return get_backend().get_mask_label(proposals_idx, proposals_offset, instance_ids, instance_cls,
                                            proposals_iou, ignored_label, iou_thr)
