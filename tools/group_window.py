"""Host-side anatomy of the lock-step stretch of a PointGroup step (backbone end -> backward start): wall time spent in each
call on the MAIN thread, per step, in bench.py's pipelined loop.  usage: python tools/group_window.py [--steps 30]"""
import argparse, os, sys, time, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from minsu3d_amd.config import load_config
from minsu3d_amd import backend as ms_backend
import minsu3d_amd.MinkowskiEngine as ME
import minsu3d_amd.model.general_model as GM
import minsu3d_amd.model.pointgroup as PG

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = load_config(["model=pointgroup", "data=scannetv2"])
be = ms_backend.get_backend()
model = bench.build(cfg, dev)
opt = model.configure_optimizers()
batches = [bench.make_batch([4 * s + i for i in range(4)], dev) for s in range(3)]
acc = collections.OrderedDict()
import threading
main_thread = threading.main_thread()


def timed(name, fn):
    def w(*a, **k):
        if threading.current_thread() is not main_thread:
            return fn(*a, **k)
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return w


model.backbone.forward = timed("backbone (issue)", model.backbone.forward)
model._queue_point_losses = timed("point losses (issue)", model._queue_point_losses)
PG.common_ops.ballquery_batch_p = timed("  ballquery_batch_p (main thread's)", PG.common_ops.ballquery_batch_p)
PG.pointgroup_ops.pg_bfs_cluster = timed("  pg_bfs_cluster (main thread's)", PG.pointgroup_ops.pg_bfs_cluster)
model._group = timed("_group shifted (main thread)", model._group)
PG.clusters_voxelization = timed("clusters_voxelization", PG.clusters_voxelization)
GM.ME.utils.sparse_quantize = timed("  sparse_quantize", GM.ME.utils.sparse_quantize)
model.score_net.forward = timed("score_net", model.score_net.forward)
PG.common_ops.roipool = timed("roipool", PG.common_ops.roipool)
model._loss = timed("_loss", model._loss)
torch.nonzero_orig = torch.nonzero
PG.torch.nonzero = timed("nonzero (sync: waits for the backbone)", torch.nonzero)
fwd = model.forward
model.forward = timed("MODEL FORWARD total", fwd)

for i in range(5):
    bench.train_step(model, model, opt, batches[i % 3], batches[(i + 1) % 3])
torch.cuda.synchronize()
acc.clear()
t0 = time.perf_counter()
for i in range(args.steps):
    bench.train_step(model, model, opt, batches[(i + 5) % 3], batches[(i + 6) % 3])
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.steps * 1e3
print(f"wall per step {wall:.2f} ms; main-thread time per step inside:")
for k, v in acc.items():
    print(f"  {k:50s} {v / args.steps * 1e3:7.3f} ms")
