"""GPU: the assembled model learns (the stand-in for BASELINE config 5's mAP parity that this environment allows).
PointGroup from scratch on 8 small synthetic scenes with colour-coded classes: backbone-only `prepare` phase, then the
grouping branch driven by the network's OWN semantic predictions and offsets (no ground-truth grouping inputs as in
the benchmark), evaluated by the ScanNet-protocol evaluator through the device post-processing."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_pointgroup_learns_on_synthetic_scenes():
    import convergence
    rec = convergence.run(steps=360, prepare=160)
    before, mid, end = rec["eval"]
    tot = [sum(l.get(k, 0.0) for k in ("semantic_loss", "offset_norm_loss", "offset_dir_loss")) for l in rec["loss"]]
    win = [float(np.mean(tot[i:i + 40])) for i in range(0, len(tot), 40)]
    print("point-loss window means:", [round(w, 3) for w in win])
    print("eval:", before, mid, end)
    assert before["semantic_mIoU"] < 20.0 and before["AP50"] < 0.05         # (mIoU in percent) a random network knows nothing
    assert all(b <= a + 0.02 for a, b in zip(win, win[1:])), win             # the per-point losses fall window by window
    assert win[-1] < win[0] - 2.0                    # (the direction loss is a negative cosine: the sum ends below zero)
    assert mid["semantic_mIoU"] > 85.0 and end["semantic_mIoU"] > 92.0      # classes are read off the colours
    assert end["AP50"] > 0.7 and end["AP25"] > 0.8 and end["AP"] > 0.5, end    # instances found by the learned grouping
    assert end["AP"] > mid["AP"] + 0.2               # ... and the branch trained after `prepare` is what finds them
    assert end["predicted_instances"] <= 2 * end["gt_instances"]
    score = [l["score_loss"] for l in rec["loss"] if "score_loss" in l]
    assert len(score) == 200 and np.mean(score[-40:]) < np.mean(score[:40])     # the ScoreNet learns its IoU targets
