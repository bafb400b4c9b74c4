"""Offline evaluation of a prediction directory -- the reference's `eval.py:9-56`: read the ground truth `.pth`
scenes of a split and the ScanNet-benchmark prediction files written by `util.io.save_prediction`, then run the
instance-segmentation and box-detection evaluators.

    python -m minsu3d_amd.offline_eval model=pointgroup data=scannetv2 model.inference.split=val
"""
import os
import sys

import numpy as np

from .config import load_config
from .evaluation import GeneralDatasetEvaluator, evaluate_bbox_acc, get_gt_bbox, get_gt_instances
from .util.io import read_gt_files_from_disk, read_pred_files_from_disk


def evaluate_prediction_files(cfg, print_result=True):
    """-> (instance segmentation result dict, box detection result dict), as the two evaluators return them"""
    split = cfg.model.inference.split
    pred_dir = os.path.join(cfg.exp_output_root_path, "inference", split, "predictions", "instance")
    if not os.path.exists(pred_dir):
        raise FileNotFoundError(f"prediction files do not exist: {pred_dir}")
    with open(getattr(cfg.data.metadata, f"{split}_list")) as f:
        scene_names = [line.strip() for line in f if line.strip()]
    all_pred, all_gt, all_gt_bbox = [], [], []
    for scan_id in scene_names:
        gt_xyz, gt_sem, gt_inst = read_gt_files_from_disk(os.path.join(cfg.data.dataset_path, split, f"{scan_id}.pth"))
        all_gt.append(get_gt_instances(gt_sem, gt_inst, cfg.data.ignore_classes))
        all_pred.append(read_pred_files_from_disk(os.path.join(pred_dir, scan_id + ".txt"), gt_xyz,
                                                  cfg.data.mapping_classes_ids, cfg.data.ignore_classes))
        all_gt_bbox.append(get_gt_bbox(gt_xyz, gt_inst, gt_sem, -1, cfg.data.ignore_classes))
    evaluator = GeneralDatasetEvaluator(cfg.data.class_names, -1, cfg.data.ignore_classes)
    inst = evaluator.evaluate(all_pred, all_gt, print_result=print_result)
    with np.errstate(invalid="ignore"):
        bbox = evaluate_bbox_acc(all_pred, all_gt_bbox, cfg.data.class_names, cfg.data.ignore_classes,
                                 print_result=print_result)
    return inst, bbox


def main(argv=None):
    cfg = load_config(list(sys.argv[1:] if argv is None else argv))
    print(f"==> start evaluating {cfg.model.inference.split} set ...")
    evaluate_prediction_files(cfg)


if __name__ == "__main__":
    main()
