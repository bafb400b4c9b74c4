"""Evaluator timing on a validation-set-sized workload: the reference's own evaluator (imported from /root/reference,
build container only) next to minsu3d_amd.evaluation on identical inputs; also checks that the results agree.

    python tools/eval_bench.py [n_scans n_points n_instances]"""
import os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
if not hasattr(np, "NINF"):
    np.NINF = -np.inf
sys.path.insert(0, "/root/reference")
import make_golden_eval as G                                      # scan generator + the reference imports
from minsu3d_amd.evaluation import instance_segmentation as ours_is
from minsu3d_amd.evaluation import object_detection as ours_od

n_scans, n, n_inst = (int(a) for a in (sys.argv[1:4] + ["12", "150000", "40"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
scans = [G.make_scan(rng, f"scene{s:04d}_00", n, n_inst, 0.15, 25) for s in range(n_scans)]


def build(rle_encode, get_gt_instances, get_gt_bbox):
    pred_list, gt_list, bbox_gts = [], [], []
    for sc in scans:
        plist = []
        for label, conf, idx in sc["preds"]:
            mask = np.zeros(n, bool); mask[idx] = True
            pts = sc["xyz"][mask]
            plist.append({"scan_id": sc["scan_id"], "label_id": label, "conf": np.float32(conf), "pred_mask": rle_encode(mask),
                          "pred_bbox": np.concatenate((pts.min(0), pts.max(0)))})
        pred_list.append(plist)
        gt_list.append(get_gt_instances(torch.from_numpy(sc["sem"].astype(np.int64)).clone(),
                                        torch.from_numpy(sc["inst"].astype(np.int64)).clone(), G.IGNORED).numpy())
        bbox_gts.append(get_gt_bbox(sc["xyz"], sc["inst"], sc["sem"], -1, G.IGNORED))
    return pred_list, gt_list, bbox_gts


pl_r, gl_r, bb_r = build(G.rle_encode, G.get_gt_instances, G.get_gt_bbox)
pl_o, gl_o, bb_o = build(ours_is.rle_encode, ours_is.get_gt_instances, ours_od.get_gt_bbox)
print(f"{n_scans} scans x {n} points, {n_inst} instances, {np.mean([len(p) for p in pl_r]):.0f} predictions per scan")
t = time.perf_counter(); res_r = G.GeneralDatasetEvaluator(G.CLASSES, -1, G.IGNORED).evaluate(pl_r, gl_r, print_result=False); t_r = time.perf_counter() - t
t = time.perf_counter(); res_o = ours_is.GeneralDatasetEvaluator(G.CLASSES, -1, G.IGNORED).evaluate(pl_o, gl_o, print_result=False); t_o = time.perf_counter() - t
for k in ("all_ap", "all_ap_50%", "all_ap_25%"):
    assert abs(float(res_r[k]) - float(res_o[k])) < 1e-12, (k, res_r[k], res_o[k])
print(f"instance AP: reference {t_r:.2f} s, here {t_o:.2f} s ({t_r / t_o:.1f}x), identical all_ap / ap50 / ap25 = "
      f"{float(res_o['all_ap']):.4f} / {float(res_o['all_ap_50%']):.4f} / {float(res_o['all_ap_25%']):.4f}")
t = time.perf_counter(); b_r = G.evaluate_bbox_acc(pl_r, bb_r, G.CLASSES, G.IGNORED, print_result=False); t_r = time.perf_counter() - t
t = time.perf_counter(); b_o = ours_od.evaluate_bbox_acc(pl_o, bb_o, G.CLASSES, G.IGNORED, print_result=False); t_o = time.perf_counter() - t
for k in b_r:
    for c in b_r[k]:
        x, y = float(b_r[k][c]), float(b_o[k][c])     # classes without ground truth come out NaN in both
        assert (np.isnan(x) and np.isnan(y)) or abs(x - y) < 1e-12, (k, c, x, y)
print(f"box AP: reference {t_r:.2f} s, here {t_o:.2f} s ({t_r / t_o:.1f}x), identical")
