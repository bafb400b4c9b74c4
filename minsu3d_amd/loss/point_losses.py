"""The three per-point losses of the backbone heads as ONE autograd node over three library launches
(csrc/losses.hip): semantic cross entropy with ignore_index -1 (reference model/general_model.py:39-41), offset norm and
offset direction losses (loss/pt_offset_loss.py:11-38).  The torch formulation (GeneralModel._point_losses_torch) costs
~30 operators forward and ~40 backward, in the stretch of a training step where the GPU waits for every launch."""
import torch

from ..backend import get_backend


class PointLossesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, pred_offsets, labels, centre, xyz, instance_ids):
        be = get_backend()
        out5, d_scores, d_off = be.point_losses_forward(scores.contiguous(), labels.contiguous(), pred_offsets.contiguous(),
                                                        centre.contiguous(), xyz.contiguous(), instance_ids.contiguous())
        ctx.held = [out5, d_scores, d_off]
        ctx.set_materialize_grads(False)
        return out5[0], out5[1], out5[2]

    @staticmethod
    def backward(ctx, g_sem, g_norm, g_dir):
        if ctx.held is None:
            raise RuntimeError("PointLossesFn: the gradients were scaled in place by the first backward pass")
        out5, d_scores, d_off = ctx.held
        ctx.held = None
        f = lambda g: None if g is None else g.to(torch.float32).contiguous()
        get_backend().point_losses_scale_grads(d_scores, d_off, out5, f(g_sem), f(g_norm), f(g_dir))
        return (d_scores if ctx.needs_input_grad[0] else None), (d_off[0] if ctx.needs_input_grad[1] else None), \
            None, None, None, None


def point_losses(scores, pred_offsets, labels, centre, xyz, instance_ids):
    """-> (semantic_loss, offset_norm_loss, offset_dir_loss), 0-dim tensors"""
    return PointLossesFn.apply(scores, pred_offsets, labels, centre, xyz, instance_ids)
