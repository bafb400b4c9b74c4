"""ScanNet-benchmark prediction files (reference `minsu3d/util/io.py:8-62`):
   <save_path>/instance/<scan_id>.txt                  one line per instance: "predicted_masks/<scan>_<nnn>.txt <class id> <conf>"
   <save_path>/instance/predicted_masks/<scan>_<nnn>.txt   one 0/1 per point
"""
import os

import numpy as np
import torch

from ..evaluation.instance_segmentation import rle_decode, rle_encode


def _kept_ids(mapping_ids, ignored_classes_indices):
    return [v for i, v in enumerate(mapping_ids) if i + 1 not in ignored_classes_indices]


def save_prediction(save_path, all_pred_insts, mapping_ids, ignored_classes_indices):
    inst_dir = os.path.join(save_path, "instance")
    mask_dir = os.path.join(inst_dir, "predicted_masks")
    os.makedirs(mask_dir, exist_ok=True)
    class_of = _kept_ids(mapping_ids, ignored_classes_indices)     # evaluation class 1..C -> dataset label id
    for preds in all_pred_insts:
        scan_id = preds[0]["scan_id"]
        lines = []
        for n, pred in enumerate(preds):
            rel = f"predicted_masks/{scan_id}_{n:03d}.txt"
            lines.append(f"{rel} {class_of[pred['label_id'] - 1]} {pred['conf']:.4f}")
            np.savetxt(os.path.join(inst_dir, rel), rle_decode(pred["pred_mask"]), fmt="%d")
        with open(os.path.join(inst_dir, f"{scan_id}.txt"), "w") as f:
            f.write("\n".join(lines))


def read_gt_files_from_disk(data_path):
    scene = torch.load(data_path)
    scene["xyz"] -= scene["xyz"].mean(axis=0)
    return scene["xyz"], scene["sem_labels"], scene["instance_ids"]


def read_pred_files_from_disk(data_path, gt_xyz, mapping_ids, ignored_classes_indices):
    label_of = {v: i for i, v in enumerate(_kept_ids(mapping_ids, ignored_classes_indices), 1)}
    out = []
    with open(data_path) as f:
        for line in f:
            rel, label, conf = line.strip().split()
            mask = np.loadtxt(os.path.join(os.path.dirname(data_path), rel), dtype=bool)
            pts = gt_xyz[mask]
            out.append({"scan_id": os.path.basename(data_path), "label_id": label_of[int(label)], "conf": float(conf),
                        "pred_mask": rle_encode(mask), "pred_bbox": np.concatenate((pts.min(0), pts.max(0)))})
    return out
