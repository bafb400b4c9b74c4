"""Does the assembled model LEARN?  PointGroup (m = 16, 3-level U-Net) trained from scratch on a fixed set of small
synthetic scenes whose classes are predictable from colour, with the reference's schedule in miniature: `prepare`
steps of backbone-only training (semantic + offset losses), then the grouping branch on the NETWORK'S OWN predictions
(ball query / BFS / proposal voxelisation / ScoreNet / score loss).  Prints a JSON record with the loss curve and the
evaluation (semantic mIoU, instance AP / AP50 / AP25 from the ScanNet-protocol evaluator through the device
post-processing) before training, after the prepare phase and at the end.  The only stand-in this environment allows
for BASELINE config 5 (real ScanNet, val mAP; reference README.md:146).
usage: python tools/convergence.py [--steps 360] [--prepare 160] > profiles/r03_convergence.json"""
import argparse, json, os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minsu3d_amd.config import load_config
from minsu3d_amd.data import synthetic
from minsu3d_amd.engine import predicted_instances
from minsu3d_amd.evaluation import GeneralDatasetEvaluator, evaluate_semantic_miou, get_gt_instances
import minsu3d_amd.model as M

SCENE = dict(room=(2.0, 1.6), n_boxes=3, density=1700.0, wall_h=0.8, n_box_classes=4, class_colours=True)


def evaluate(model, scenes, cfg, with_instances):
    model.eval()
    miou, preds, gts = [], [], []
    with torch.no_grad():
        for b in scenes:
            out = model(b)
            miou.append(evaluate_semantic_miou(out["semantic_scores"].max(1)[1], b["sem_labels"], ignore_label=-1))
            if with_instances:
                preds.append(predicted_instances(model, b, out))
                gts.append(get_gt_instances(b["sem_labels"].cpu().clone(), b["instance_ids"].cpu().clone(), cfg.data.ignore_classes))
    model.train()
    res = {"semantic_mIoU": float(np.mean(miou))}
    if with_instances:
        r = GeneralDatasetEvaluator(cfg.data.class_names, -1, cfg.data.ignore_classes).evaluate(preds, gts, print_result=False)
        res.update({"AP": float(r["all_ap"]), "AP50": float(r["all_ap_50%"]), "AP25": float(r["all_ap_25%"]),
                    "predicted_instances": int(sum(len(p) for p in preds)), "gt_instances": 3 * len(scenes)})
    return res


def run(steps=360, prepare=160, n_scenes=8, batch=4, seed=0, log=None):
    dev = torch.device("cuda", 0)
    cfg = load_config(["model=pointgroup", "model.network.blocks=[1,2,3]", "model.optimizer.lr=0.004"])
    torch.manual_seed(seed)
    model = M.PointGroup(cfg).to(dev).train()
    opt = model.configure_optimizers()
    one = [synthetic.to_torch(synthetic.collate([synthetic.make_scene(100 + i, **SCENE)]), dev) for i in range(n_scenes)]
    batches = [synthetic.to_torch(synthetic.collate([synthetic.make_scene(100 + i, **SCENE) for i in range(j, j + batch)]), dev)
               for j in range(0, n_scenes, batch)]
    prep = cfg.model.network.prepare_epochs
    rec = {"config": {"model": "PointGroup m=16 blocks=[1,2,3]", "scenes": n_scenes, "points_per_scene": int(one[0]["point_xyz"].size(0)),
                      "scene": {k: (list(v) if isinstance(v, tuple) else v) for k, v in SCENE.items()}, "steps": steps,
                      "prepare_steps": prepare, "batch": batch, "lr": cfg.model.optimizer.lr, "seed": seed},
           "loss": [], "eval": []}
    model.current_epoch = prep + 1
    rec["eval"].append(dict(step=0, **evaluate(model, one, cfg, True)))
    t0 = time.perf_counter()
    for step in range(steps):
        model.current_epoch = 0 if step < prepare else prep + 1       # grouping on the network's own predictions
        b = batches[step % len(batches)]
        opt.zero_grad(set_to_none=True)
        losses = model._loss(b, model(b))
        total = sum(losses.values())
        total.backward()
        opt.step()
        rec["loss"].append({k: round(float(v.detach()), 5) for k, v in losses.items()})
        if step + 1 == prepare or step + 1 == steps:
            model.current_epoch = prep + 1
            rec["eval"].append(dict(step=step + 1, **evaluate(model, one, cfg, True)))
            if log:
                log(rec["eval"][-1])
    rec["seconds"] = round(time.perf_counter() - t0, 1)
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=360)
    ap.add_argument("--prepare", type=int, default=160)
    a = ap.parse_args()
    r = run(a.steps, a.prepare, log=lambda e: print(e, file=sys.stderr))
    w = 20
    tot = [sum(l.get(k, 0.0) for k in ("semantic_loss", "offset_norm_loss", "offset_dir_loss")) for l in r["loss"]]
    r["point_loss_window_means"] = [round(float(np.mean(tot[i:i + w])), 4) for i in range(0, len(tot), w)]
    r["loss"] = r["loss"][::10]
    print(json.dumps(r, indent=1))
