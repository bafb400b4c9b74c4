"""Which Python lines of this repository call ATen operators on device tensors during one training step (forward + loss;
the backward pass runs on autograd's thread and is listed by operator only)?  A TorchDispatchMode around one step of
bench.py's loop, operators attributed to the innermost frame inside the repository.
usage: python tools/aten_dispatch.py [--model pointgroup]"""
import argparse, collections, os, sys, traceback
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from minsu3d_amd.config import load_config

ap = argparse.ArgumentParser(); ap.add_argument("--model", default="pointgroup"); args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = load_config([f"model={args.model}", "data=scannetv2"])
model = bench.build(cfg, dev); opt = model.configure_optimizers()
batches = [bench.make_batch([4 * i + j for j in range(4)], dev) for i in range(3)]
for i in range(4):
    bench.train_step(model, model, opt, batches[i % 3], batches[(i + 1) % 3])
torch.cuda.synchronize()
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
counts = collections.Counter()
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.expand", "aten.slice",
        "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.as_strided", "aten.unbind", "aten.split", "aten.permute",
        "aten.empty", "aten.reshape", "aten.is_pinned", "aten._local_scalar_dense", "aten.lift_fresh", "aten.record_stream",
        "aten.narrow", "aten.unfold", "aten.new_empty", "aten.resize_", "aten.set_")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), k=None):
        name = str(func)
        out = func(*a, **(k or {}))
        if not name.startswith(SKIP):
            cuda = any(isinstance(t, torch.Tensor) and t.is_cuda for t in a) or (isinstance(out, torch.Tensor) and out.is_cuda)
            if cuda:
                fr = "?"
                for f in reversed(traceback.extract_stack(limit=30)):
                    if f.filename.startswith(root) and "tools/aten_dispatch" not in f.filename:
                        fr = f"{f.filename[len(root) + 1:]}:{f.lineno} {f.name}"
                        break
                counts[(name, fr)] += 1
        return out


with Log():
    bench.train_step(model, model, opt, batches[1], batches[2])
torch.cuda.synchronize()
tot = sum(counts.values())
print(f"{tot} device-tensor ATen calls in one step on the calling thread (views / allocations not counted)")
by_frame = collections.Counter()
for (name, fr), c in counts.items():
    by_frame[fr] += c
for fr, c in by_frame.most_common(60):
    ops = ", ".join(f"{n.replace('aten.', '')} x{k}" for (n, f), k in sorted(counts.items(), key=lambda kv: -kv[1]) if f == fr)
    print(f"{c:4d}  {fr:70s} {ops[:150]}")
