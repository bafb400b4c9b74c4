"""Cosine learning-rate decay from `start_epoch` on (reference `minsu3d/util/lr_decay.py:7-12`)."""
from math import cos, pi


def cosine_lr_decay(optimizer, base_lr, current_epoch, start_epoch, total_epochs, clip):
    if current_epoch < start_epoch:
        return
    phase = (current_epoch - start_epoch) / (total_epochs - start_epoch)
    lr = clip + 0.5 * (base_lr - clip) * (1 + cos(pi * phase))
    for group in optimizer.param_groups:
        group["lr"] = lr
