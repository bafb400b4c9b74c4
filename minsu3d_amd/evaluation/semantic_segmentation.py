"""Per-point semantic accuracy and mean IoU in percent (reference `minsu3d/evaluation/semantic_segmentation.py:4-21`),
one bincount over (gt, prediction) pairs instead of a Python loop over classes; stays on the tensors' device."""
import torch


def evaluate_semantic_accuracy(pred, gt, ignore_label):
    valid = gt != ignore_label
    return torch.count_nonzero(gt[valid] == pred[valid]).item() / int(torch.count_nonzero(valid)) * 100


def evaluate_semantic_miou(pred, gt, ignore_label):
    valid = gt != ignore_label
    p, g = pred[valid].long(), gt[valid].long()
    lo = int(min(p.min().item(), g.min().item()))
    n = int(max(p.max().item(), g.max().item())) - lo + 1
    conf = torch.bincount((g - lo) * n + (p - lo), minlength=n * n).view(n, n)     # [gt, pred] confusion counts
    inter = conf.diagonal()
    gt_n, pred_n = conf.sum(1), conf.sum(0)
    present = gt_n > 0                                                               # classes that occur in gt (:14)
    ious = inter[present] / (gt_n[present] + pred_n[present] - inter[present])       # int / int -> float32, as :18
    return ious.to(torch.float32).mean().item() * 100
