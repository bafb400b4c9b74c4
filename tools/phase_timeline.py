"""Host clock vs GPU clock per phase of a training step: where the GPU waits for Python (host-bound phases) and where
Python waits for the GPU.  usage: python tools/phase_timeline.py"""
import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd.config import load_config

cfg = load_config(); dev = torch.device("cuda", 0)
model = bench.build(cfg, dev); opt = model.configure_optimizers()
batches = [bench.make_batch([4 * i + j for j in range(4)], dev) for i in range(3)]
marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    marks.append((name, time.perf_counter(), ev))


model.backbone.register_forward_pre_hook(lambda m, a: mark("backbone_begin"))
model.backbone.register_forward_hook(lambda m, a, o: mark("backbone_end"))
model.score_net.register_forward_pre_hook(lambda m, a: mark("grouping_end/scorenet_begin"))
model.score_net.register_forward_hook(lambda m, a, o: mark("scorenet_end"))


def step(b):
    mark("step_begin")
    opt.zero_grad(set_to_none=True)
    out = model(b)
    mark("forward_end")
    loss = sum(model._loss(b, out).values())
    mark("loss_end")
    loss.backward()
    mark("backward_end")
    opt.step()
    mark("opt_end")


for i in range(5): step(batches[i % 3])
torch.cuda.synchronize()
acc = {}
N = 12
for i in range(N):
    marks.clear()
    step(batches[i % 3])
    torch.cuda.synchronize()
    for (n0, h0, e0), (n1, h1, e1) in zip(marks[:-1], marks[1:]):
        a = acc.setdefault(f"{n0} -> {n1}", [0.0, 0.0])
        a[0] += (h1 - h0) * 1e3; a[1] += e0.elapsed_time(e1)
print(f"{'phase':55s} {'host ms':>8s} {'gpu ms':>8s}")
th = tg = 0
for k, (h, g) in acc.items():
    print(f"{k:55s} {h / N:8.2f} {g / N:8.2f}"); th += h / N; tg += g / N
print(f"{'total':55s} {th:8.2f} {tg:8.2f}")
