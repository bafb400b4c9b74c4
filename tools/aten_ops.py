"""Which lines of the host code put aten operators on the stream: every operator of one training step is caught by a
TorchDispatchMode and attributed to the innermost frame inside this repository (operators issued by the autograd
engine for built-in ops show up under the `loss.backward()` line).
usage: python tools/aten_ops.py [--model pointgroup]"""
import sys, os, argparse, collections, traceback, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd.config import load_config
from torch.utils._python_dispatch import TorchDispatchMode
ap = argparse.ArgumentParser(); ap.add_argument("--model", default="pointgroup"); args = ap.parse_args()
cfg = load_config([f"model={args.model}", "data=scannetv2"]); dev = torch.device("cuda", 0)
model = bench.build(cfg, dev); opt = model.configure_optimizers()
batch = bench.make_batch([0, 1, 2, 3], dev)
for i in range(3): bench.train_step(model, model, opt, batch)
torch.cuda.synchronize()
VIEW = ("view", "reshape", "expand", "permute", "transpose", "t.", "select", "slice", "unsqueeze", "squeeze", "detach",
        "alias", "as_strided", "unbind", "split", "narrow", "_unsafe_view", "empty", "size", "stride", "is_", "numel",
        "_local_scalar_dense", "lift_fresh", "unfold", "resize_", "set_", "record_stream", "item")
by_line = collections.defaultdict(lambda: [0, collections.Counter()])
total = 0


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        global total
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if not any(name.startswith(v) for v in VIEW):
            where = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if ("/minsu3d_amd/" in fr.filename or fr.filename.endswith("bench.py")) and "tools/" not in fr.filename:
                    where = f"{fr.filename.split('minsu3d_amd/')[-1].split('/root/repo/')[-1]}:{fr.lineno} {fr.name}"
                    break
            a = by_line[where]; a[0] += 1; a[1][name.split('.')[0]] += 1; total += 1
        return func(*args, **(kwargs or {}))


with Log():
    bench.train_step(model, model, opt, batch)
torch.cuda.synchronize()
print(f"{total} non-view operators in one step")
print(f"{'issued from':60s} {'ops':>5s}  operators")
for k, (n, ops) in sorted(by_line.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{k[:60]:60s} {n:5d}  {', '.join(f'{o}x{c}' for o, c in ops.most_common(6))[:90]}")
