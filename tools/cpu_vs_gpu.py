"""How much of a step is host-bound: time to ENQUEUE a step (no sync) vs time with a sync after every step."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd.config import load_config
cfg = load_config(); dev = torch.device("cuda", 0)
model = bench.build(cfg, dev); opt = model.configure_optimizers()
batches = [bench.make_batch([4 * i + j for j in range(4)], dev) for i in range(4)]
for i in range(5): bench.train_step(model, model, opt, batches[i % 4])
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter(); enq = 0.0
for i in range(n):
    t1 = time.perf_counter()
    bench.train_step(model, model, opt, batches[i % 4])
    enq += time.perf_counter() - t1
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"pipelined: {1e3 * tot / n:.2f} ms/step, host enqueue {1e3 * enq / n:.2f} ms/step")
t0 = time.perf_counter()
for i in range(n):
    bench.train_step(model, model, opt, batches[i % 4]); torch.cuda.synchronize()
print(f"sync every step: {1e3 * (time.perf_counter() - t0) / n:.2f} ms/step")
# phases
import collections
ph = collections.defaultdict(float)
for i in range(n):
    b = batches[i % 4]
    torch.cuda.synchronize(); t = time.perf_counter()
    opt.zero_grad(set_to_none=True); out = model(b); torch.cuda.synchronize(); ph["fwd"] += time.perf_counter() - t; t = time.perf_counter()
    loss = sum(model._loss(b, out).values()); torch.cuda.synchronize(); ph["loss"] += time.perf_counter() - t; t = time.perf_counter()
    loss.backward(); torch.cuda.synchronize(); ph["bwd"] += time.perf_counter() - t; t = time.perf_counter()
    opt.step(); torch.cuda.synchronize(); ph["opt"] += time.perf_counter() - t
print({k: round(1e3 * v / n, 2) for k, v in ph.items()})
