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


# ------------------------------------------------------------------ prediction files
@pytest.fixture(scope="module")
def io_golden():
    with open(os.path.join(HERE, "golden", "io_cases.json")) as f:
        return json.load(f)


def _io_predictions(io_golden):
    out = []
    for sc in io_golden["scans"]:
        preds = []
        for p in sc["preds"]:
            mask = np.zeros(sc["n"], bool)
            mask[p["members"]] = True
            preds.append({"scan_id": sc["scan_id"], "label_id": p["label_id"], "conf": p["conf"], "pred_mask": rle_encode(mask)})
        out.append(preds)
    return out


def test_save_prediction_writes_the_reference_files(tmp_path, io_golden):
    """every file the reference's own save_prediction wrote for the same predictions (tests/golden/make_golden_io.py
    ran minsu3d/util/io.py:8-33), byte for byte, and nothing else"""
    from minsu3d_amd.util.io import save_prediction
    save_prediction(str(tmp_path), _io_predictions(io_golden), io_golden["mapping"], io_golden["ignored"])
    got = {}
    for root, _, names in os.walk(tmp_path):
        for name in names:
            p = os.path.join(root, name)
            got[os.path.relpath(p, tmp_path)] = open(p).read()
    assert sorted(got) == sorted(io_golden["files"])
    for rel, text in io_golden["files"].items():
        assert got[rel] == text, rel


def test_read_pred_files_matches_the_reference_reader(tmp_path, io_golden):
    """the reference-written files, read back by our reader == what the reference's reader returned (util/io.py:42-62)"""
    from minsu3d_amd.util.io import read_gt_files_from_disk, read_pred_files_from_disk
    for rel, text in io_golden["files"].items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    for sc in io_golden["scans"]:
        back = read_pred_files_from_disk(str(tmp_path / "instance" / (sc["scan_id"] + ".txt")),
                                         np.array(sc["xyz"], np.float32), io_golden["mapping"], io_golden["ignored"])
        assert len(back) == len(sc["read_back"])
        for got, want in zip(back, sc["read_back"]):
            assert got["scan_id"] == want["scan_id"] and got["label_id"] == want["label_id"] and got["conf"] == want["conf"]
            assert got["pred_mask"] == want["pred_mask"]
            assert np.array_equal(np.asarray(got["pred_bbox"], np.float32), np.asarray(want["pred_bbox"], np.float32))
    g = io_golden["gt_case"]
    pth = tmp_path / "scene.pth"
    torch.save({"xyz": np.array(g["xyz"], np.float32), "sem_labels": np.array(g["sem_labels"], np.int16),
                "instance_ids": np.array(g["instance_ids"], np.int16)}, pth)
    xyz, sem, inst = read_gt_files_from_disk(str(pth))
    assert np.array_equal(np.asarray(xyz, np.float32), np.array(g["out_xyz"], np.float32))
    assert np.array_equal(np.asarray(sem), np.array(g["out_sem"])) and np.array_equal(np.asarray(inst), np.array(g["out_inst"]))


def test_offline_eval_entry(tmp_path, golden):
    """eval.py:9-56: predictions written to disk, re-read with the ground-truth scenes and evaluated == the evaluators
    fed directly (the evaluators themselves are pinned by the reference's numbers above)"""
    from minsu3d_amd.config import load_config
    from minsu3d_amd.offline_eval import evaluate_prediction_files
    from minsu3d_amd.util.io import save_prediction
    case = golden["cases"][1]
    cfg = load_config(["model=pointgroup", "data=scannetv2", f"exp_output_root_path={tmp_path}/out",
                       f"data.dataset_path={tmp_path}/data", f"data.metadata.val_list={tmp_path}/val.txt"])
    cfg.data.class_names = case["classes"]
    cfg.data.ignore_classes = case["ignored"]
    cfg.data.mapping_classes_ids = list(range(1, len(case["classes"]) + 1))
    os.makedirs(tmp_path / "data" / "val")
    names, all_preds, all_gts, all_boxes = [], [], [], []
    for sc in case["scans"]:
        sem, inst, xyz, preds = load_scan(case, sc, golden["arrays"])
        preds = [p for p in preds if 1 <= p["label_id"] <= len(case["classes"]) - len(case["ignored"])]
        names.append(sc["scan_id"])
        torch.save({"xyz": xyz.copy(), "sem_labels": sem.copy(), "instance_ids": inst.copy()}, tmp_path / "data" / "val" / f"{sc['scan_id']}.pth")
        all_preds.append(preds)
        centred = xyz - xyz.mean(axis=0)
        for p in preds:                     # boxes are taken on the centred scene, as read_gt_files_from_disk returns it
            pts = centred[rle_decode(p["pred_mask"]).astype(bool)]
            p["pred_bbox"] = np.concatenate((pts.min(0), pts.max(0)))
            p["conf"] = float(f"{float(p['conf']):.4f}")          # what survives the text file
        all_gts.append(get_gt_instances(torch.from_numpy(sem.astype(np.int64)).clone(), torch.from_numpy(inst.astype(np.int64)).clone(),
                                        case["ignored"]))
        all_boxes.append(get_gt_bbox(centred, inst, sem, -1, case["ignored"]))
    (tmp_path / "val.txt").write_text("\n".join(names) + "\n")
    save_prediction(str(tmp_path / "out" / "inference" / "val" / "predictions"), all_preds, cfg.data.mapping_classes_ids,
                    case["ignored"])
    inst_res, bbox_res = evaluate_prediction_files(cfg, print_result=False)
    want = GeneralDatasetEvaluator(case["classes"], -1, case["ignored"]).evaluate(all_preds, all_gts, print_result=False)
    for k in ("all_ap", "all_ap_50%", "all_ap_25%"):
        assert same(float(inst_res[k]), float(want[k])), k
    with np.errstate(invalid="ignore"):
        want_b = evaluate_bbox_acc(all_preds, all_boxes, case["classes"], case["ignored"], print_result=False)
    for th in want_b:
        assert same(float(bbox_res[th]["avg"]), float(want_b[th]["avg"])), th
