"""Count the aten operators (and their device time) of one training step: python tools/aten_ops.py"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd.config import load_config
from torch.profiler import profile, ProfilerActivity
cfg = load_config(); dev = torch.device("cuda", 0)
model = bench.build(cfg, dev); opt = model.configure_optimizers()
batch = bench.make_batch([0, 1, 2, 3], dev)
for i in range(3): bench.train_step(model, model, opt, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.train_step(model, model, opt, batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
print(f"{'op':40s} {'calls':>6s} {'dev us':>9s} {'cpu us':>9s}")
for e in rows[:40]:
    print(f"{e.key:40s} {e.count:6d} {e.device_time_total:9.0f} {e.self_cpu_time_total:9.0f}")
