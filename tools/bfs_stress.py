"""Randomised stress of the clustering operators against the CPU oracle: many seeds, the three graph regimes (sparse,
dense symmetric, dense directed / capped) and both threshold modes.  Test infrastructure (imports oracle/).
usage: python tools/bfs_stress.py [n_seeds]"""
import sys, os, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from oracle import oracle as O
from minsu3d_amd.backend import get_backend
O.lib(); be = get_backend()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(1000 + seed)
    kind = seed % 3
    n = int(rng.integers(5000, 60000))
    B = int(rng.integers(1, 5))
    if kind == 0:      # sparse: surface-like points, small radius
        xyz = rng.random((n, 3)).astype(np.float32) * np.array([4, 3, 0.2], np.float32); r = 0.04
    elif kind == 1:    # dense symmetric: blobs of a few hundred points
        c = rng.random((n // 300 + 1, 3)) * np.array([6, 5, 2]); xyz = (c[rng.integers(0, len(c), n)] + rng.standard_normal((n, 3)) * 0.02).astype(np.float32); r = 0.03
    else:              # dense directed: blobs above the 1000 cap
        c = rng.random((n // 2500 + 1, 3)) * np.array([6, 5, 2]); xyz = (c[rng.integers(0, len(c), n)] + rng.standard_normal((n, 3)) * 0.012).astype(np.float32); r = 0.04
    b = np.sort(rng.integers(0, B, n)).astype(np.uint8)
    bo = np.concatenate(([0], np.cumsum(np.bincount(b, minlength=B)))).astype(np.int32)
    sem = rng.integers(0, 4, n).astype(np.int16) if seed % 2 else np.full(n, 1, np.int16)
    wi, ws = O.ballquery_batch_p(xyz, b, bo, r)
    gi, gs = be.ballquery_batch_p(dev(xyz), dev(b), dev(bo), r, 50)
    ok = np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gs.cpu().numpy(), ws)
    thr = int(rng.integers(2, 60))
    want = O.pg_bfs_cluster(sem, wi, ws, thr)
    for rep in range(2):   # twice: workspace reuse
        a, o = be.pg_bfs_cluster(dev(sem), gi, gs, thr)
        ok = ok and np.array_equal(o.cpu().numpy(), want[1]) and np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))
    mean = [float(x) for x in rng.integers(20, 400, 6)]
    want = O.sg_bfs_cluster(mean, wi, ws, 0.05, int(seed % 6))
    a, o = be.sg_bfs_cluster(mean, gi, gs, 0.05, int(seed % 6))
    ok = ok and np.array_equal(o.cpu().numpy(), want[1]) and np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))
    print(f"seed {seed} kind {kind} n {n} edges {wi.size} cap {int(ws[:, 1].max())} clusters {want[1].size - 1}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
