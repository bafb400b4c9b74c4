"""ScanNet-benchmark prediction files (the layout of the reference's `minsu3d/util/io.py:8-62`):

    <save_path>/instance/<scan_id>.txt                      one line per instance:
                                                            "predicted_masks/<scan>_<nnn>.txt <dataset class id> <conf>"
    <save_path>/instance/predicted_masks/<scan>_<nnn>.txt   one 0/1 per point
"""
import os

import numpy as np
import torch

from ..evaluation.instance_segmentation import rle_decode, rle_encode

_MASK_DIR = "predicted_masks"


def _evaluation_classes(mapping_ids, ignored_classes_indices):
    """dataset label ids of the evaluated classes, in evaluation order (class 1, 2, ...)"""
    return [label for pos, label in enumerate(mapping_ids, 1) if pos not in ignored_classes_indices]


def _write_scan(inst_dir, scan_id, instances, dataset_label):
    index_lines = []
    for number, inst in enumerate(instances):
        mask_file = f"{_MASK_DIR}/{scan_id}_{number:03d}.txt"
        np.savetxt(os.path.join(inst_dir, mask_file), rle_decode(inst["pred_mask"]), fmt="%d")
        index_lines.append("%s %s %.4f" % (mask_file, dataset_label[inst["label_id"] - 1], inst["conf"]))
    with open(os.path.join(inst_dir, scan_id + ".txt"), "w") as handle:
        handle.write("\n".join(index_lines))


def save_prediction(save_path, all_pred_insts, mapping_ids, ignored_classes_indices):
    inst_dir = os.path.join(save_path, "instance")
    os.makedirs(os.path.join(inst_dir, _MASK_DIR), exist_ok=True)
    dataset_label = _evaluation_classes(mapping_ids, ignored_classes_indices)
    for instances in all_pred_insts:
        _write_scan(inst_dir, instances[0]["scan_id"], instances, dataset_label)


def read_gt_files_from_disk(data_path):
    scene = torch.load(data_path, weights_only=False)      # pickled numpy arrays (data/scannetv2/preprocess_all_data.py)
    centred = scene["xyz"] - scene["xyz"].mean(axis=0)
    scene["xyz"] = centred
    return centred, scene["sem_labels"], scene["instance_ids"]


def read_pred_files_from_disk(data_path, gt_xyz, mapping_ids, ignored_classes_indices):
    evaluation_class = {label: pos for pos, label in enumerate(_evaluation_classes(mapping_ids, ignored_classes_indices), 1)}
    folder, scan = os.path.split(data_path)
    instances = []
    with open(data_path) as handle:
        entries = [line.split() for line in handle if line.strip()]
    for mask_file, label, conf in entries:
        member = np.loadtxt(os.path.join(folder, mask_file), dtype=bool)
        box_pts = gt_xyz[member]
        instances.append(dict(scan_id=scan, label_id=evaluation_class[int(label)], conf=float(conf),
                              pred_mask=rle_encode(member), pred_bbox=np.concatenate((box_pts.min(0), box_pts.max(0)))))
    return instances
