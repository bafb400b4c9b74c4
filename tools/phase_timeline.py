"""Host clock vs GPU clock per phase of a training step, in bench.py's pipelined loop (no sync between steps, next
batch's coordinate work prefetched): per phase the host time, the GPU time between the phase's boundary events, and the
GPU's LAG behind the host at the phase's end (lag ~ 0: the GPU ran dry and waits for Python there; lag of milliseconds:
the host is ahead and the phase is GPU-bound).
usage: python tools/phase_timeline.py [--model pointgroup|hais|softgroup] [--steps N]"""
import sys, os, time, argparse, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import minsu3d_amd.MinkowskiEngine as ME

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="pointgroup")
ap.add_argument("--steps", type=int, default=12)
args = ap.parse_args()
dev = torch.device("cuda", 0)
from minsu3d_amd.config import load_config
cfg = load_config([f"model={args.model}", "data=scannetv2"])
model = bench.build(cfg, dev); opt = model.configure_optimizers()
batches = [bench.make_batch([4 * i + j for j in range(4)], dev) for i in range(3)]
marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    marks.append((name, time.perf_counter(), ev))


model.backbone.register_forward_pre_hook(lambda m, a: mark("backbone_begin"))
model.backbone.register_forward_hook(lambda m, a, o: mark("backbone_end"))
score = getattr(model, "score_net", None) or getattr(model, "tiny_unet", None)
if score is not None:
    score.register_forward_pre_hook(lambda m, a: mark("grouping_end/scorenet_begin"))
    score.register_forward_hook(lambda m, a, o: mark("scorenet_end"))


# the proposal branch starts at clusters_voxelization (right behind the grouping's last round trip)
import importlib
_mod = importlib.import_module(type(model).__module__)
if hasattr(_mod, "clusters_voxelization"):
    _cv = _mod.clusters_voxelization

    def _cv_marked(*a, **k):
        mark("proposals_begin")
        return _cv(*a, **k)
    _mod.clusters_voxelization = _cv_marked


def step(b, nxt):
    mark("step_begin")
    opt.zero_grad(set_to_none=True)
    pf = lambda: ME.prefetch_coordinates(nxt["voxel_xyz"], model.backbone.n_levels, wait_current_stream=False,
                                         channels=model.backbone.level_channels)
    if bench.PREFETCH_AT == "grouping":
        model.schedule_after_backbone(pf)
    out = model(b)
    mark("forward_end")
    loss = sum(model._loss(b, out).values())
    mark("loss_end")
    if bench.PREFETCH_AT != "grouping":
        pf()
    loss.backward()
    mark("backward_end")
    opt.step()
    mark("opt_end")


for i in range(5): step(batches[i % 3], batches[(i + 1) % 3])
torch.cuda.synchronize()
marks.clear()
ref = torch.cuda.Event(enable_timing=True); ref.record(); torch.cuda.synchronize()
ref_host = time.perf_counter()
N = args.steps
for i in range(N):
    step(batches[(i + 5) % 3], batches[(i + 6) % 3])
mark("end")
torch.cuda.synchronize()
wall = (time.perf_counter() - ref_host) * 1e3
acc = {}
for (n0, h0, e0), (n1, h1, e1) in zip(marks[:-1], marks[1:]):
    a = acc.setdefault(f"{n0} -> {n1}", [0.0, 0.0, 0.0, 0])
    a[0] += (h1 - h0) * 1e3; a[1] += e0.elapsed_time(e1)
    a[2] += ref.elapsed_time(e1) - (h1 - ref_host) * 1e3; a[3] += 1
print(f"{'phase':55s} {'host ms':>8s} {'gpu ms':>8s} {'lag at end':>11s}")
th = tg = 0
for k, (h, g, lag, n) in acc.items():
    print(f"{k:55s} {h / N:8.2f} {g / N:8.2f} {lag / n:11.2f}"); th += h / N; tg += g / N
print(f"{'total':55s} {th:8.2f} {tg:8.2f}    wall/step {wall / N:.2f}")
