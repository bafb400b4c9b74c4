"""Point-offset regression loss (reference minsu3d/loss/pt_offset_loss.py:11-38): mean L1 norm of the offset
error and mean negative cosine between predicted and ground-truth offset directions, over valid points."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class PTOffsetLoss(nn.Module):
    def forward(self, pred_offsets, gt_offsets, valid_mask):
        """masked means instead of boolean indexing: no host sync, no data-dependent shapes.  With no valid point both
        losses are 0 (the reference returns the integers 0, 0 there)."""
        w = valid_mask.to(pred_offsets.dtype)
        n = w.sum().clamp_min(1.0)
        gt = torch.where(valid_mask[:, None], gt_offsets, torch.zeros_like(gt_offsets))   # gt is undefined off-mask
        norm_loss = ((pred_offsets - gt).abs().sum(-1) * w).sum() / n
        eps = torch.finfo(gt.dtype).eps
        cos = (F.normalize(gt, p=2, dim=1, eps=eps) * F.normalize(pred_offsets, p=2, dim=1, eps=eps)).sum(-1)
        return norm_loss, -(cos * w).sum() / n
