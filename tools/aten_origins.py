"""Which Python lines launch the small ATen kernels of a benchmark step (fills, adds, copies ...)?  torch.profiler with
stacks over a few steps, kernels attributed to the innermost frame inside this repository.
usage: python tools/aten_origins.py [--model pointgroup] [--steps 3] [--top 60]"""
import argparse, collections, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="pointgroup"); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--top", type=int, default=60)
args = ap.parse_args()
argv = ["--model", args.model, "--steps", str(args.steps), "--warmup", "4", "--no-cpu-baseline", "--no-roofline"]
ap2 = None
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    bench.main(argv)
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_stack_n=12):
    dev = getattr(ev, "self_device_time_total", 0) or 0
    if not ev.key.startswith("aten::") or dev <= 0:
        continue
    stack = list(ev.stack or [])
    frame = next((f for f in stack if "minsu3d_amd" in f or "bench.py" in f), stack[0] if stack else "?")
    key = (ev.key, frame.replace(root + "/", "")[:120])
    agg[key][0] += ev.count
    agg[key][1] += dev
n = args.steps + 4
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])[:args.top]
print(f"{'calls/step':>10} {'us/step':>9}  operator  <- frame   (over {n} steps incl. warm-up)")
for (name, frame), (calls, us) in rows:
    print(f"{calls / n:10.1f} {us / n:9.1f}  {name}  <- {frame}")

# the same by input shape for the most frequent small operators (which tensors are they?)
print()
shp = collections.defaultdict(int)
for ev in prof.key_averages(group_by_input_shape=True):
    if ev.key in ("aten::fill_", "aten::add_", "aten::copy_", "aten::add", "aten::sum", "aten::zero_", "aten::zeros", "aten::zeros_like"):
        shp[(ev.key, str(ev.input_shapes)[:90])] += ev.count
for (k, sh), c in sorted(shp.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c / n:8.1f}  {k}  {sh}")
