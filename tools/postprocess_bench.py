"""Time the device instance post-processing against the dense-mask host formulation of the reference on one
ScanNet-sized scan (150k points, ~400 proposals).  python tools/postprocess_bench.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from postprocess_cases import make_case
from minsu3d_amd.model import postprocess as PP
from oracle import postprocess_oracle as PO

c = make_case(0, n=150000, n_regions=60, per_region=6, junk=40)
d = lambda a: torch.from_numpy(a).cuda()
args = (d(c["scores"]), d(c["proposals_idx"]), c["P"], d(c["sem"]))
for _ in range(3):
    got = PP.pointgroup_instances("s", c["xyz"], *args, 2, 0.09, 100, 0.3)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10):
    got = PP.pointgroup_instances("s", c["xyz"], *args, 2, 0.09, 100, 0.3)
torch.cuda.synchronize(); gpu = (time.perf_counter() - t) / 10
t = time.perf_counter()
want = PO.pointgroup_instances("s", c["xyz"], c["scores"], c["proposals_idx"], c["P"], c["sem"], 2, 0.09, 100, 0.3)
cpu = time.perf_counter() - t
print(f"proposals {c['P']} pairs {c['proposals_idx'].shape[0]} instances {len(got)} (dense host {len(want)}): "
      f"device path {1e3 * gpu:.2f} ms / scan, dense-mask host path {1e3 * cpu:.1f} ms / scan")
