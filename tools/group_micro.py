"""The grouping operators of one benchmark step in isolation, on the benchmark's own inputs (4 synthetic scenes,
foreground points, original and offset-shifted coordinates): ms per call, optional bit-exact check against the CPU
oracle (slow: ~15 s per ball query on 8 cores).   usage: python tools/group_micro.py [--check] [--reps 20]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd.backend import get_backend

ap = argparse.ArgumentParser(); ap.add_argument("--check", action="store_true"); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--noise", type=float, default=0.04)
args = ap.parse_args()
dev = torch.device("cuda", 0); be = get_backend()
b = bench.make_batch([0, 1, 2, 3], dev, offset_noise=args.noise)
sem = b["grouping_semantic_preds"]; fg = sem >= 2
obj = torch.nonzero(fg).view(-1)
bi = b["vert_batch_ids"][obj].contiguous()
bo = torch.cumsum(torch.bincount(bi.long(), minlength=4), 0); bo = torch.cat([bo.new_zeros(1), bo]).int()
xyz = b["point_xyz"][obj].contiguous(); sh = (xyz + b["grouping_point_offsets"][obj]).contiguous(); semfg = sem[obj].contiguous()


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for name, P, ma in (("orig", xyz, 50), ("shift", sh, 300)):
    ms, (idx, sl) = timed(lambda: be.ballquery_batch_p(P, bi, bo, 0.03, ma), args.reps)
    print(f"{name}: n={P.shape[0]} nActive={idx.numel()} ballquery {ms:.3f} ms")
    ms2, (ci, co) = timed(lambda: be.pg_bfs_cluster(semfg, idx, sl, 50), args.reps)
    print(f"{name}: pg_bfs_cluster {ms2:.3f} ms  clusters={co.numel() - 1} rows={ci.shape[0]}")
    if args.check:
        from oracle import oracle as O
        widx, wsl = O.ballquery_batch_p(P.cpu().numpy(), bi.cpu().numpy(), bo.cpu().numpy(), 0.03)
        assert np.array_equal(sl.cpu().numpy(), wsl) and np.array_equal(idx.cpu().numpy(), widx), "ball query differs"
        wi, wo = O.pg_bfs_cluster(semfg.cpu().numpy(), widx, wsl, 50)
        assert np.array_equal(co.cpu().numpy(), wo) and np.array_equal(ci.cpu().numpy().reshape(-1, 2), wi.reshape(-1, 2)), "bfs differs"
        print(f"{name}: bit-exact vs oracle")


# the reference's own data flow (pointgroup.py:43-55): ball query, `.cpu()` of the neighbour lists, clustering on the host
# tensors through the drop-in COMMON_OPS module, clusters back to the device -- with and without the module's reuse of the
# device copies (MS3D_DROPIN_REUSE)
import minsu3d_amd.dropin as dropin
dropin.install()
import COMMON_OPS
for reuse in ("1", "0"):
    os.environ["MS3D_DROPIN_REUSE"] = reuse
    COMMON_OPS._GRAPHS.clear()

    def route():
        n = sh.size(0)
        idx = torch.zeros(n * 300, dtype=torch.int32, device=dev); sl = torch.zeros((n, 2), dtype=torch.int32, device=dev)
        na = COMMON_OPS.ballquery_batch_p(sh, bi, bo, idx, sl, n, 300, 0.03)
        ci, co = torch.empty(0, dtype=torch.int32), torch.empty(0, dtype=torch.int32)
        COMMON_OPS.pg_bfs_cluster(semfg.cpu(), idx[:na].cpu(), sl.cpu(), ci, co, n, 50)
        return ci.long().to(dev), co.to(dev)
    for _ in range(3): route()
    ms, _ = timed(route, max(args.reps // 2, 6))
    print(f"zero-edit route (shifted): ball query + .cpu() + COMMON_OPS.pg_bfs_cluster + .to(device), reuse={reuse}: {ms:.2f} ms")
