"""Point-offset regression loss (reference minsu3d/loss/pt_offset_loss.py:11-38): mean L1 norm of the offset
error and mean negative cosine between predicted and ground-truth offset directions, over valid points."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class PTOffsetLoss(nn.Module):
    def forward(self, pred_offsets, gt_offsets, valid_mask):
        if not bool(valid_mask.any()):
            return 0, 0
        pred, gt = pred_offsets[valid_mask], gt_offsets[valid_mask]
        norm_loss = (pred - gt).abs().sum(-1).mean()
        eps = torch.finfo(gt.dtype).eps
        cos = (F.normalize(gt, p=2, dim=1, eps=eps) * F.normalize(pred, p=2, dim=1, eps=eps)).sum(-1)
        return norm_loss, (-cos).mean()
