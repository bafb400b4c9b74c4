"""cProfile of the host side of the BACKBONE forward only (no grouping syncs, no autograd thread): where do the ~45 us of
Python / ctypes / torch dispatch per convolution layer go?   usage: python tools/host_profile.py [--model pointgroup] [--reps 30]"""
import argparse, cProfile, io, os, pstats, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from minsu3d_amd.config import load_config
from minsu3d_amd import backend as ms_backend

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="pointgroup"); ap.add_argument("--reps", type=int, default=30); ap.add_argument("--top", type=int, default=40)
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = load_config([f"model={args.model}", "data=scannetv2"])
ms_backend.get_backend()
model = bench.build(cfg, dev)
batch = bench.make_batch([0, 1, 2, 3], dev)
bb = model.backbone


def fwd():
    return bb(batch["voxel_features"] if "voxel_features" in batch else batch["point_features"], batch["voxel_xyz"], batch["v2p_map"]) \
        if False else model(batch)


for _ in range(3):
    out = fwd()
torch.cuda.synchronize()
prof = cProfile.Profile()
prof.enable()
for _ in range(args.reps):
    out = fwd()
prof.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(args.top)
txt = s.getvalue()
print(txt[:9000].replace(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/", ""))
