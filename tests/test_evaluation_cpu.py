"""Evaluation / scheduling helpers against golden vectors produced by the reference's own
code (tests/golden/make_golden_eval.py imports /root/reference/minsu3d/evaluation and util/lr_decay)."""
import json
import math
import os

import numpy as np
import pytest
import torch

from minsu3d_amd.evaluation import (GeneralDatasetEvaluator, evaluate_bbox_acc, evaluate_semantic_accuracy,
                                    evaluate_semantic_miou, get_gt_bbox, get_gt_instances, rle_decode, rle_encode)
from minsu3d_amd.evaluation.instance_segmentation import rle_indices
from minsu3d_amd.util.lr_decay import cosine_lr_decay

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "eval_cases.json")) as f:
        d = json.load(f)
    d["arrays"] = np.load(os.path.join(HERE, "golden", "eval_arrays.npz"))
    return d


def same(a, b):
    return (isinstance(a, float) and isinstance(b, float) and math.isnan(a) and math.isnan(b)) or a == pytest.approx(b, abs=1e-7)


def load_scan(case, sc, arrays):
    k = sc["key"]
    n = case["n"]
    sem, inst, xyz = arrays[k + "sem"], arrays[k + "inst"], arrays[k + "xyz"]
    off, mem = arrays[k + "pred_offsets"], arrays[k + "pred_members"]
    preds = []
    for i, (label, conf) in enumerate(zip(arrays[k + "pred_label"], arrays[k + "pred_conf"])):
        mask = np.zeros(n, bool)
        mask[mem[off[i]:off[i + 1]]] = True
        pts = xyz[mask]
        preds.append({"scan_id": sc["scan_id"], "label_id": int(label), "conf": np.float32(conf), "pred_mask": rle_encode(mask),
                      "pred_bbox": np.concatenate((pts.min(0), pts.max(0)))})
    return sem, inst, xyz, preds


def test_rle_known_answers(golden):
    for mask, want in golden["kats"]["rle"]:
        got = rle_encode(np.array(mask, dtype=np.int64))
        assert got == want
        assert np.array_equal(rle_decode(got), np.array(mask, dtype=np.uint8))
        assert np.array_equal(rle_indices(got), np.flatnonzero(np.array(mask, dtype=np.int64)))


def test_gt_instances_and_boxes(golden):
    for case in golden["cases"]:
        for sc in case["scans"]:
            k = sc["key"]
            sem, inst, xyz, _ = load_scan(case, sc, golden["arrays"])
            ids = get_gt_instances(torch.from_numpy(sem.astype(np.int64)).clone(), torch.from_numpy(inst.astype(np.int64)).clone(),
                                   case["ignored"]).numpy()
            assert np.array_equal(ids, golden["arrays"][k + "gt_ids"])
            boxes = get_gt_bbox(xyz, inst, sem, -1, case["ignored"])
            assert [int(c) for c, _ in boxes] == golden["arrays"][k + "gt_bbox_cls"].tolist()
            if boxes:
                assert np.array_equal(np.array([b for _, b in boxes], np.float32), golden["arrays"][k + "gt_bbox"])


@pytest.mark.parametrize("use_label", [True, False])
def test_instance_ap_matches_reference(golden, use_label):
    for case in golden["cases"]:
        pred_list, gt_list = [], []
        for sc in case["scans"]:
            _, _, _, preds = load_scan(case, sc, golden["arrays"])
            pred_list.append(preds)
            gt_list.append(golden["arrays"][sc["key"] + "gt_ids"].astype(np.int64))
        ev = GeneralDatasetEvaluator(case["classes"], -1, case["ignored"], use_label=use_label)
        got = ev.evaluate(pred_list, gt_list, print_result=False)
        want = case["expect"][f"inst_use_label_{use_label}"]
        for key, w in want.items():
            if key == "classes":
                for cname, metrics in w.items():
                    for m, v in metrics.items():
                        assert same(float(got["classes"][cname][m]), v), (case["seed"], cname, m)
            else:
                assert same(float(got[key]), w), (case["seed"], key)


def test_bbox_ap_matches_reference(golden):
    checked = 0
    for case in golden["cases"]:
        if case["expect"].get("bbox") is None:
            continue
        pred_list, gts = [], []
        for sc in case["scans"]:
            sem, inst, xyz, preds = load_scan(case, sc, golden["arrays"])
            pred_list.append(preds)
            gts.append(get_gt_bbox(xyz, inst, sem, -1, case["ignored"]))
        with np.errstate(invalid="ignore"):
            got = evaluate_bbox_acc(pred_list, gts, case["classes"], case["ignored"], print_result=False)
        for th, want in case["expect"]["bbox"].items():
            assert set(str(k) for k in got[th]) == set(want)
            for cname, v in want.items():
                g = got[th][cname if cname == "avg" else type(next(k for k in got[th] if str(k) == cname))(cname)]
                assert same(float(g), v), (case["seed"], th, cname)
        checked += 1
    assert checked >= 3


def test_semantic_metrics_and_lr_decay(golden):
    for pred, gt, acc, miou in golden["kats"]["semantic"]:
        p, g = torch.tensor(pred), torch.tensor(gt)
        assert evaluate_semantic_accuracy(p, g, -1) == pytest.approx(acc, abs=1e-9)
        assert evaluate_semantic_miou(p, g, -1) == pytest.approx(miou, rel=1e-6)
    for epoch, lr in golden["kats"]["lr"]:
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.001)
        cosine_lr_decay(opt, 0.001, epoch, 100, 400, 1e-6)
        assert opt.param_groups[0]["lr"] == lr


def test_prediction_files_round_trip(tmp_path, golden):
    from minsu3d_amd.util.io import read_pred_files_from_disk, save_prediction
    case = golden["cases"][0]
    sc = case["scans"][0]
    _, _, xyz, preds = load_scan(case, sc, golden["arrays"])
    preds = [p for p in preds if 1 <= p["label_id"] <= len(case["classes"]) - len(case["ignored"])]
    mapping = list(range(1, len(case["classes"]) + 1))
    save_prediction(str(tmp_path), [preds], mapping, case["ignored"])
    back = read_pred_files_from_disk(str(tmp_path / "instance" / f"{sc['scan_id']}.txt"), xyz, mapping, case["ignored"])
    assert len(back) == len(preds)
    for a, b in zip(preds, back):
        assert a["label_id"] == b["label_id"] and a["pred_mask"] == b["pred_mask"]
        assert b["conf"] == pytest.approx(float(a["conf"]), abs=5e-5) and np.array_equal(a["pred_bbox"], b["pred_bbox"])
