"""Does a training step reach the driver allocator (hipMalloc / hipFree) once warm?  python tools/alloc_stats.py [pool]"""
import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd.config import load_config
pool = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = load_config(); dev = torch.device("cuda", 0)
model = bench.build(cfg, dev); opt = model.configure_optimizers()
batches = [bench.make_batch([4 * i + j for j in range(4)], dev) for i in range(pool)]
print("points per batch:", [int(b["point_xyz"].shape[0]) for b in batches], "voxels:", [int(b["voxel_xyz"].shape[0]) for b in batches])
for i in range(2 * pool): bench.train_step(model, model, opt, batches[i % pool], batches[(i + 1) % pool])
torch.cuda.synchronize()
keys = ["num_device_alloc", "num_device_free", "num_alloc_retries", "segment.all.allocated", "reserved_bytes.all.current"]
s0 = {k: torch.cuda.memory_stats()[k] for k in keys}
for i in range(2 * pool):
    t = time.perf_counter(); bench.train_step(model, model, opt, batches[i % pool], batches[(i + 1) % pool]); torch.cuda.synchronize()
    s1 = {k: torch.cuda.memory_stats()[k] for k in keys}
    print(f"step {i}: {1e3 * (time.perf_counter() - t):.2f} ms ", {k: s1[k] - s0[k] for k in keys[:4]}, "reserved GB %.2f" % (s1[keys[4]] / 2**30))
    s0 = s1
