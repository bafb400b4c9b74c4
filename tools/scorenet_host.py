"""Host time of the pieces of the proposal branch (the host-bound stretch of a step between the grouping's last round
trip and the backward pass): wall time per call inside bench.py's training loop, and a cProfile of the stretch.
usage: python tools/scorenet_host.py [--model pointgroup]"""
import sys, os, time, argparse, cProfile, pstats, io, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd.config import load_config
from minsu3d_amd.model import general_model as gm, pointgroup as pgm
from minsu3d_amd.common_ops.functions import common_ops
import minsu3d_amd.MinkowskiEngine as ME

ap = argparse.ArgumentParser(); ap.add_argument("--model", default="pointgroup"); args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = load_config([f"model={args.model}", "data=scannetv2"])
model = bench.build(cfg, dev); opt = model.configure_optimizers()
batches = [bench.make_batch([4 * i + j for j in range(4)], dev) for i in range(3)]
acc = {}
prof = cProfile.Profile()
state = {"on": False}


def timed(name, fn, profile=False):
    def w(*a, **k):
        t0 = time.perf_counter()
        if profile and state["on"]:
            prof.enable()
        try:
            return fn(*a, **k)
        finally:
            if profile and state["on"]:
                prof.disable()
            acc.setdefault(name, []).append(time.perf_counter() - t0)
    return w


from minsu3d_amd.MinkowskiEngine.tensor import CoordinateManager
from minsu3d_amd.model.module import common as cm_mod

be = __import__("minsu3d_amd.backend", fromlist=["x"]).get_backend()
for name in ("pairlist", "offsetlist", "downsample", "kmap_k3", "kmap_k2", "sparse_quantize", "proposal_voxel_coords"):
    pass

pgm.clusters_voxelization = timed("clusters_voxelization", pgm.clusters_voxelization, True)
_sn = model.score_net


def _sn_forward(x):
    t0 = time.perf_counter(); x.coordinate_manager.prepare(2); t1 = time.perf_counter()
    be_ = __import__("minsu3d_amd.backend", fromlist=["x"]).get_backend()
    cm = x.coordinate_manager
    for ts, c in ((1, 16), (2, 32)):      # the lists the convolutions will ask for (built lazily inside them otherwise)
        nbr = cm.k3(ts); v = cm.size(ts)
        be_.pairlist(nbr, 27, v, c, c); be_.offsetlist(nbr, 27, v)
    down, up = cm.k2(1); vc = cm.size(2)
    be_.pairlist(down, 8, vc, 16, 32); be_.offsetlist(down, 8, vc); be_.pairlist(up, 8, cm.size(1), 32, 16); be_.offsetlist(up, 8, cm.size(1))
    t2 = time.perf_counter()
    y = _sn.unet(x)
    t3 = time.perf_counter()
    acc.setdefault("  tiny: coordinate sets + kernel maps", []).append(t1 - t0)
    acc.setdefault("  tiny: pair / offset lists", []).append(t2 - t1)
    acc.setdefault("  tiny: unet modules", []).append(t3 - t2)
    return y


model.score_net.forward = timed("score_net.forward", _sn_forward, True)
common_ops.roipool = timed("roipool", common_ops.roipool, True)
real_loss = model._loss
model._loss = timed("_loss", real_loss, True)
ME_gather = ME.gather_rows
pgm.ME.gather_rows = timed("gather_rows(p2v)", ME_gather, True)
for i in range(6):
    bench.train_step(model, model, opt, batches[i % 3], batches[(i + 1) % 3])
torch.cuda.synchronize(); acc.clear(); state["on"] = True
N = 20
for i in range(N):
    bench.train_step(model, model, opt, batches[i % 3], batches[(i + 1) % 3])
torch.cuda.synchronize()
for k, v in acc.items():
    print(f"{k:28s} {1e3 * sum(v) / N:7.3f} ms/step ({len(v) // N} calls)")
s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
