"""GPU parity: HIP grouping kernels (through the C ABI) vs the CPU oracle, the golden vectors made by
the reference's own code, and -- when oracle/_ref travelled to this box -- the reference's own GPU
kernels.  Integer/index outputs must be bit-exact; sec_mean / avg-pool float outputs too (same
serial order)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from minsu3d_amd.backend import HipBackend
    return HipBackend()


def dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.cuda()


def scene(rng, n, kind, B):
    if kind == "surface":
        xyz = rng.random((n, 3)) * np.array([2.0, 2.0, 0.04])
    elif kind == "blobs":
        c = rng.random((8, 3)) * 3
        xyz = c[rng.integers(0, 8, n)] + rng.standard_normal((n, 3)) * 0.02
    else:
        xyz = rng.standard_normal((n, 3)) * 0.003
    b = np.sort(rng.integers(0, B, n)).astype(np.uint8)
    bo = np.concatenate([[0], np.cumsum(np.bincount(b, minlength=B))]).astype(np.int32)
    return xyz.astype(np.float32), b, bo


@pytest.mark.parametrize("kind,n,B,radius", [("surface", 20000, 3, 0.03), ("surface", 5000, 1, 0.06),
                                             ("blobs", 6000, 2, 0.03), ("capped", 2500, 1, 0.03),
                                             ("surface", 1, 1, 0.03), ("surface", 70, 2, 0.5)])
def test_ballquery_vs_oracle(be, oracle, kind, n, B, radius):
    rng = np.random.default_rng(n + B)
    xyz, b, bo = scene(rng, n, kind, B)
    want_idx, want_sl = oracle.ballquery_batch_p(xyz, b, bo, radius)
    idx, sl = be.ballquery_batch_p(dev(xyz), dev(b), dev(bo), radius, 20)
    assert np.array_equal(sl.cpu().numpy(), want_sl)
    assert np.array_equal(idx.cpu().numpy(), want_idx)


def test_ballquery_collapsed_scene(be, oracle):
    """every point of a scene inside a few grid cells (a perfectly trained offset branch): neighbourhoods of 20k
    candidates -- the workgroup sort's global-scratch path -- and every list capped at the 1000 lowest indices"""
    rng = np.random.default_rng(77)
    n = 20000
    xyz = (rng.standard_normal((n, 3)) * 0.004).astype(np.float32)
    xyz[15000:] += np.float32(0.5)                       # a second, smaller blob (LDS sort path)
    b = np.zeros(n, np.uint8)
    bo = np.array([0, n], np.int32)
    want_idx, want_sl = oracle.ballquery_batch_p(xyz, b, bo, 0.03)
    idx, sl = be.ballquery_batch_p(dev(xyz), dev(b), dev(bo), 0.03, 300)
    assert want_sl[:, 1].max() == 1000
    assert np.array_equal(sl.cpu().numpy(), want_sl)
    assert np.array_equal(idx.cpu().numpy(), want_idx)


def test_ballquery_empty(be):
    idx, sl = be.ballquery_batch_p(torch.zeros((0, 3), device="cuda"), torch.zeros(0, dtype=torch.uint8, device="cuda"),
                                   torch.zeros(2, dtype=torch.int32, device="cuda"), 0.03, 50)
    assert idx.numel() == 0 and sl.shape == (0, 2)


@pytest.mark.parametrize("ci", range(6))
def test_bfs_vs_golden(be, golden_dir, ci):
    g = np.load(os.path.join(golden_dir, f"bfs_case{ci}.npz"))
    # ball query on the device must rebuild the exact graph the reference BFS consumed
    idx, sl = be.ballquery_batch_p(dev(g["xyz"]), dev(g["batch_idxs"]), dev(g["batch_offsets"]), float(g["radius"]), 50)
    assert np.array_equal(idx.cpu().numpy(), g["ball_idx"]) and np.array_equal(sl.cpu().numpy(), g["start_len"])
    a, b = be.pg_bfs_cluster(dev(g["sem"]), idx, sl, int(g["threshold"]))
    assert np.array_equal(b.cpu().numpy(), g["pg_offsets"])
    assert np.array_equal(a.cpu().numpy().reshape(-1, 2), g["pg_idxs"].reshape(-1, 2))
    for k, cid in enumerate(g["sg_class_ids"]):
        a, b = be.sg_bfs_cluster(g["sg_mean"].tolist(), idx, sl, float(g["sg_threshold"]), int(cid))
        assert np.array_equal(b.cpu().numpy(), g[f"sg{k}_offsets"])
        assert np.array_equal(a.cpu().numpy().reshape(-1, 2), g[f"sg{k}_idxs"].reshape(-1, 2))


def test_bfs_kat(be, golden_dir):
    g = np.load(os.path.join(golden_dir, "bfs_kat.npz"))
    a, b = be.pg_bfs_cluster(dev(g["sem"]), dev(g["ball_idx"]), dev(g["start_len"]), int(g["threshold"]))
    assert a.cpu().numpy().tolist() == g["pg_idxs"].tolist() and b.cpu().numpy().tolist() == g["pg_offsets"].tolist()


@pytest.mark.parametrize("kind,n,B,radius,thr", [("surface", 30000, 2, 0.03, 50), ("surface", 8000, 1, 0.05, 5),
                                                 ("blobs", 9000, 3, 0.03, 50), ("capped", 3000, 1, 0.03, 50),
                                                 ("blobs", 4000, 1, 0.02, 1)])
def test_bfs_vs_oracle(be, oracle, kind, n, B, radius, thr):
    rng = np.random.default_rng(100 + n)
    xyz, b, bo = scene(rng, n, kind, B)
    idx, sl = oracle.ballquery_batch_p(xyz, b, bo, radius)
    sem = rng.integers(2, 4, n).astype(np.int16)
    if kind == "capped":
        sem[:] = 2
    want = oracle.pg_bfs_cluster(sem, idx, sl, thr)
    a, o = be.pg_bfs_cluster(dev(sem), dev(idx), dev(sl), thr)
    assert np.array_equal(o.cpu().numpy(), want[1])
    assert np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))
    mean = [-1.0, 200.0, 1000.0]
    for cid in range(3):
        want = oracle.sg_bfs_cluster(mean, idx, sl, 0.05, cid)
        a, o = be.sg_bfs_cluster(mean, dev(idx), dev(sl), 0.05, cid)
        assert np.array_equal(o.cpu().numpy(), want[1])
        assert np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))


def test_bfs_workspace_reuse(be, oracle):
    """different dense graphs of the same size back to back on one workspace: nothing an earlier call left behind
    (claims, candidate tags, scan state) may leak into the next one"""
    n = 12000
    for seed in (1, 2, 3, 1):
        rng = np.random.default_rng(seed)
        c = rng.random((6, 3)) * 2
        xyz = (c[rng.integers(0, 6, n)] + rng.standard_normal((n, 3)) * 0.025).astype(np.float32)
        b = np.zeros(n, np.uint8); bo = np.array([0, n], np.int32)
        sem = rng.integers(2, 4, n).astype(np.int16)
        widx, wsl = oracle.ballquery_batch_p(xyz, b, bo, 0.03)
        assert wsl[:, 1].max() < 1000 and widx.size >= 24 * n      # the chip-wide level-synchronous path
        idx, sl = be.ballquery_batch_p(dev(xyz), dev(b), dev(bo), 0.03, 300)
        assert np.array_equal(idx.cpu().numpy(), widx)
        want = oracle.pg_bfs_cluster(sem, widx, wsl, 20)
        a, o = be.pg_bfs_cluster(dev(sem), idx, sl, 20)
        assert np.array_equal(o.cpu().numpy(), want[1])
        assert np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))


def test_bfs_deterministic(be, oracle):
    rng = np.random.default_rng(9)
    xyz, b, bo = scene(rng, 20000, "surface", 2)
    idx, sl = oracle.ballquery_batch_p(xyz, b, bo, 0.04)
    sem = rng.integers(0, 2, 20000).astype(np.int16)
    r = [be.pg_bfs_cluster(dev(sem), dev(idx), dev(sl), 10) for _ in range(3)]
    for k in (1, 2):
        assert torch.equal(r[0][0], r[k][0]) and torch.equal(r[0][1], r[k][1])


def _segments(rng, P, maxlen):
    lens = rng.integers(0, maxlen, P)
    lens[0] = maxlen * 3
    if P > 3:
        lens[3] = 0
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


@pytest.mark.parametrize("C_", [3, 16, 32, 19, 100])
def test_segment_ops_vs_oracle(be, oracle, C_):
    rng = np.random.default_rng(C_)
    off = _segments(rng, 300, 400)
    S = int(off[-1])
    x = rng.standard_normal((S, C_)).astype(np.float32)
    x[5:40] = x[5]  # ties: first extremum must win
    xd, od = dev(x), dev(off)
    assert np.array_equal(be.sec_mean(xd, od).cpu().numpy(), oracle.sec_mean(x, off), equal_nan=True)
    assert np.array_equal(be.sec_min(xd, od).cpu().numpy(), oracle.sec_min(x, off))
    assert np.array_equal(be.sec_max(xd, od).cpu().numpy(), oracle.sec_max(x, off))
    out, mi = be.roipool_fp(xd, od)
    wout, wmi = oracle.roipool_fp(x, off)
    assert np.array_equal(out.cpu().numpy(), wout) and np.array_equal(mi.cpu().numpy(), wmi)
    assert np.array_equal(be.global_avg_pool_fp(xd, od).cpu().numpy(), oracle.global_avg_pool_fp(x, off),
                          equal_nan=True)
    # backward passes (drop the empty proposal: its argmax is -1)
    keep = np.diff(off) > 0
    off2 = np.concatenate([[0], np.cumsum(np.diff(off)[keep])]).astype(np.int32)
    out2, mi2 = be.roipool_fp(xd, dev(off2))
    g = rng.standard_normal(out2.shape).astype(np.float32)
    got = be.roipool_bp(dev(g), dev(off2), mi2, S).cpu().numpy()
    assert np.array_equal(got, oracle.roipool_bp(g, off2, mi2.cpu().numpy(), S))
    got = be.global_avg_pool_bp(dev(g), dev(off2), S).cpu().numpy()
    assert np.array_equal(got, oracle.global_avg_pool_bp(g, off2, S))


@pytest.mark.parametrize("I", [1, 37, 300])
def test_iou_family_vs_oracle(be, oracle, I):
    rng = np.random.default_rng(I)
    N = 40000
    inst = rng.integers(-1, I, N).astype(np.int16)
    pn = np.bincount(inst[inst >= 0], minlength=I).astype(np.int32)
    off = _segments(rng, 120, 600)
    S = int(off[-1])
    pidx = rng.integers(0, N, S).astype(np.int32)
    sig = rng.random(S).astype(np.float32)
    cls = rng.integers(-1, 18, I).astype(np.int16)
    args = (dev(pidx), dev(off), dev(inst), dev(pn))
    iou = be.get_iou(*args)
    assert np.array_equal(iou.cpu().numpy(), oracle.get_iou(pidx, off, inst, pn))
    assert np.array_equal(be.get_mask_iou_on_cluster(*args).cpu().numpy(),
                          oracle.get_mask_iou_on_cluster(pidx, off, inst, pn))
    assert np.array_equal(be.get_mask_iou_on_pred(*args, dev(sig)).cpu().numpy(),
                          oracle.get_mask_iou_on_pred(pidx, off, inst, pn, sig))
    for thr in (0.0, 0.05, 0.5):
        ml, mlm = be.get_mask_label(dev(pidx), dev(off), dev(inst), dev(cls), iou, -1, thr)
        wml, wmlm = oracle.get_mask_label(pidx, off, inst, cls, iou.cpu().numpy(), -1, thr)
        assert np.array_equal(ml.cpu().numpy(), wml) and np.array_equal(mlm.cpu().numpy(), wmlm)


def test_full_size_properties(be):
    """BASELINE-size inputs (4 scenes x ~90k foreground points): size-independent properties"""
    rng = np.random.default_rng(1)
    n, B = 360000, 4
    xyz = (rng.random((n, 3)) * np.array([5.0, 4.0, 0.05])).astype(np.float32)
    b = np.sort(rng.integers(0, B, n)).astype(np.uint8)
    bo = np.concatenate([[0], np.cumsum(np.bincount(b, minlength=B))]).astype(np.int32)
    idx, sl = be.ballquery_batch_p(dev(xyz), dev(b), dev(bo), 0.03, 50)
    sl_c = sl.cpu().numpy(); idx_c = idx.cpu().numpy()
    assert np.array_equal(sl_c[:, 0], np.concatenate([[0], np.cumsum(sl_c[:-1, 1])]))   # canonical starts
    assert sl_c[:, 1].min() >= 1 and sl_c[:, 1].max() <= 1000                               # self is a neighbour
    owner = np.repeat(np.arange(n), sl_c[:, 1])
    assert np.array_equal(b[idx_c], b[owner])                                               # never crosses scenes
    d = xyz[idx_c] - xyz[owner]
    assert ((d * d).sum(1) < 0.03 ** 2 * 1.0001).all()
    seg_sorted = np.diff(idx_c) > 0
    boundary = np.zeros(idx_c.size - 1, bool); boundary[np.cumsum(sl_c[:, 1])[:-1] - 1] = True
    assert (seg_sorted | boundary).all()                                                    # ascending lists
    # symmetry (no list is capped here)
    pairs = set(zip(owner[:200000].tolist(), idx_c[:200000].tolist()))
    sub = [(j, i) for (i, j) in list(pairs)[:5000]]
    full = {}
    for (i, j) in sub:
        lst = idx_c[sl_c[i, 0]:sl_c[i, 0] + sl_c[i, 1]]
        assert j in lst
    sem = rng.integers(2, 4, n).astype(np.int16)
    a, o = be.pg_bfs_cluster(dev(sem), idx, sl, 50)
    a = a.cpu().numpy(); o = o.cpu().numpy()
    sizes = np.diff(o)
    assert (sizes >= 50).all() and a.shape[0] == o[-1]
    assert len(np.unique(a[:, 1])) == a.shape[0]                                            # a point is in <= 1 cluster
    assert np.array_equal(a[:, 0], np.repeat(np.arange(sizes.size), sizes))
    seeds = a[o[:-1], 1]
    assert (np.diff(seeds) > 0).all()                                                       # clusters by ascending seed
    mins = np.minimum.reduceat(a[:, 1], o[:-1])
    assert np.array_equal(mins, seeds)                                                      # seed = smallest member
    lab = sem[a[:, 1]]
    assert np.array_equal(lab, np.repeat(sem[seeds], sizes))                                # label-pure clusters


def test_reference_gpu_kernels_agree(be, oracle):
    """the reference's OWN kernels (hipified build in oracle/_ref) on this GPU vs ours and the oracle"""
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref/libminsu3d_ref.so did not travel to this box")
    rng = np.random.default_rng(11)
    off = _segments(rng, 80, 300)
    S = int(off[-1])
    for C_ in (3, 16):
        x = rng.standard_normal((S, C_)).astype(np.float32)
        xd, od = dev(x), dev(off)
        for name, ours in (("ref_sec_mean", be.sec_mean), ("ref_sec_min", be.sec_min), ("ref_sec_max", be.sec_max),
                           ("ref_global_avg_pool_fp", be.global_avg_pool_fp)):
            out = torch.zeros((off.size - 1, C_), device="cuda")
            getattr(R, name)(off.size - 1, C_, C.c_void_p(xd.data_ptr()), C.c_void_p(od.data_ptr()),
                             C.c_void_p(out.data_ptr()))
            mine = ours(xd, od)  # an empty segment gives 0/0 = NaN in the avg pool on both sides
            assert torch.equal(torch.nan_to_num(out, nan=7.0), torch.nan_to_num(mine, nan=7.0)), name
        out = torch.zeros((off.size - 1, C_), device="cuda"); mi = torch.zeros((off.size - 1, C_), dtype=torch.int32, device="cuda")
        R.ref_roipool_fp(off.size - 1, C_, C.c_void_p(xd.data_ptr()), C.c_void_p(od.data_ptr()),
                         C.c_void_p(out.data_ptr()), C.c_void_p(mi.data_ptr()))
        o2, m2 = be.roipool_fp(xd, od)
        assert torch.equal(out, o2) and torch.equal(mi, m2)
    # IoU
    I, N = 23, 5000
    inst = rng.integers(-1, I, N).astype(np.int16); pn = np.bincount(inst[inst >= 0], minlength=I).astype(np.int32)
    pidx = rng.integers(0, N, S).astype(np.int32); sig = rng.random(S).astype(np.float32)
    d = [dev(pidx), dev(off), dev(inst), dev(pn)]
    iou_ref = torch.zeros((off.size - 1, I), device="cuda")
    R.ref_get_iou(I, off.size - 1, *[C.c_void_p(t.data_ptr()) for t in d], C.c_void_p(iou_ref.data_ptr()))
    assert torch.equal(iou_ref, be.get_iou(*d))
    iou_ref.zero_(); sd = dev(sig)
    R.ref_get_mask_iou_on_pred(I, off.size - 1, *[C.c_void_p(t.data_ptr()) for t in d], C.c_void_p(iou_ref.data_ptr()),
                               C.c_void_p(sd.data_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(iou_ref, be.get_mask_iou_on_pred(*d, sd))
    # ball query: the reference's brute-force kernel, canonicalised (its start offsets are atomic-ordered)
    n = 3000
    xyz, b, bo = scene(rng, n, "surface", 2)
    xd, bd, bod = dev(xyz), dev(b), dev(bo)
    idx_r = torch.zeros(n * 60, dtype=torch.int32, device="cuda"); sl_r = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    R.ref_ballquery_batch_p.restype = C.c_int
    tot = R.ref_ballquery_batch_p(n, 60, C.c_float(0.05), C.c_void_p(xd.data_ptr()), C.c_void_p(bd.data_ptr()),
                                  C.c_void_p(bod.data_ptr()), C.c_void_p(idx_r.data_ptr()), C.c_void_p(sl_r.data_ptr()))
    idx_o, sl_o = be.ballquery_batch_p(xd, bd, bod, 0.05, 60)
    assert tot == idx_o.numel()
    sl_r = sl_r.cpu().numpy(); idx_r = idx_r.cpu().numpy(); sl_o = sl_o.cpu().numpy(); idx_o = idx_o.cpu().numpy()
    assert np.array_equal(sl_r[:, 1], sl_o[:, 1])
    for i in range(n):
        assert np.array_equal(idx_r[sl_r[i, 0]:sl_r[i, 0] + sl_r[i, 1]], idx_o[sl_o[i, 0]:sl_o[i, 0] + sl_o[i, 1]])


# ------------------------------------------------------------------ HAIS hierarchical aggregation
@pytest.mark.parametrize("ci", range(6))
def test_hais_vs_golden(be, golden_dir, ci):
    """point aggregation only (using_set_aggr=0): the reference's own CPU output"""
    g = np.load(os.path.join(golden_dir, f"bfs_case{ci}.npz"))
    a, o = be.hierarchical_aggregation(dev(g["sem"]), dev(g["xyz"]), dev(g["ball_idx"]), dev(g["start_len"]),
                                       dev(g["batch_idxs"]), False, g["point_num_avg"].tolist(),
                                       g["radius_avg"].tolist(), -1)
    assert np.array_equal(o.cpu().numpy(), g["hais_offsets"])
    assert np.array_equal(a.cpu().numpy().reshape(-1, 2), g["hais_idxs"].reshape(-1, 2))


def _hais_inputs(rng, n=6000):
    # a few big blobs (primaries) surrounded by small satellites (fragments) of the same class
    centres = rng.random((5, 3)) * 2
    big = centres[rng.integers(0, 5, n)] + rng.standard_normal((n, 3)) * 0.03
    sat_c = centres[rng.integers(0, 5, 40)] + rng.standard_normal((40, 3)) * 0.25
    sat = sat_c[rng.integers(0, 40, n // 4)] + rng.standard_normal((n // 4, 3)) * 0.008
    xyz = np.concatenate([big, sat]).astype(np.float32)
    perm = rng.permutation(len(xyz)); xyz = xyz[perm]
    b = np.sort(rng.integers(0, 2, len(xyz))).astype(np.uint8)
    bo = np.concatenate([[0], np.cumsum(np.bincount(b, minlength=2))]).astype(np.int32)
    sem = np.full(len(xyz), 3, np.int16); sem[rng.random(len(xyz)) < 0.1] = 4
    return xyz, b, bo, sem


@pytest.mark.parametrize("set_aggr", [False, True])
def test_hais_vs_oracle(be, oracle, set_aggr):
    rng = np.random.default_rng(21)
    xyz, b, bo, sem = _hais_inputs(rng)
    idx, sl = oracle.ballquery_batch_p(xyz, b, bo, 0.03)
    pna = [-1, -1, 100.0, 800.0, 200.0]; ra = [-1, -1, 0.2, 0.35, 0.1]
    want = oracle.hierarchical_aggregation(sem, xyz, idx, sl, b, set_aggr, pna, ra)
    a, o = be.hierarchical_aggregation(dev(sem), dev(xyz), dev(idx), dev(sl), dev(b), set_aggr, pna, ra, -1)
    assert o.numel() - 1 > 3
    assert np.array_equal(o.cpu().numpy(), want[1])
    assert np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))


def test_hais_set_aggregation_vs_reference_on_gpu(be, oracle):
    """the reference's own host code + its two GPU kernels (oracle/_ref) with using_set_aggr=1; absorbed fragments
    arrive in atomic order there, so the absorbed tail of every primary is compared as a multiset"""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref/libminsu3d_ref.so did not travel to this box")
    rng = np.random.default_rng(22)
    xyz, b, bo, sem = _hais_inputs(rng, 4000)
    idx, sl = oracle.ballquery_batch_p(xyz, b, bo, 0.03)
    pna = np.array([-1, -1, 100.0, 800.0, 200.0], np.float32); ra = np.array([-1, -1, 0.2, 0.35, 0.1], np.float32)
    ri, ro = oracle.hierarchical_aggregation(sem, xyz, idx, sl, b, True, pna, ra, -1, use_ref=True)
    a, o = be.hierarchical_aggregation(dev(sem), dev(xyz), dev(idx), dev(sl), dev(b), True, pna.tolist(), ra.tolist(), -1)
    a = a.cpu().numpy().reshape(-1, 2); o = o.cpu().numpy()
    assert np.array_equal(o, ro)
    assert np.array_equal(a[:, 0], ri[:, 0])
    for c in range(o.size - 1):
        assert np.array_equal(np.sort(a[o[c]:o[c + 1], 1]), np.sort(ri[o[c]:o[c + 1], 1]))


def test_bfs_dense_graph_with_many_levels(be, oracle):
    """dense (hundreds of neighbours) but ~100 BFS levels deep: the chip-wide expansion gives up after its level budget,
    the assembly has already run speculatively, and the replay kernel must produce the exact result"""
    rng = np.random.default_rng(31)
    n = 24000
    t = rng.random(n) * 3.0
    xyz = np.stack([t, rng.standard_normal(n) * 0.004, rng.standard_normal(n) * 0.004], 1).astype(np.float32)
    xyz[n // 2:, 1] += 1.0                               # two tubes -> two clusters
    b = np.zeros(n, np.uint8); bo = np.array([0, n], np.int32)
    sem = np.full(n, 3, np.int16)
    idx_d, sl_d = be.ballquery_batch_p(dev(xyz), dev(b), dev(bo), 0.03, 300)
    idx, sl = oracle.ballquery_batch_p(xyz, b, bo, 0.03)
    assert np.array_equal(idx_d.cpu().numpy(), idx) and idx.size > 24 * n and sl[:, 1].max() < 1000
    want = oracle.pg_bfs_cluster(sem, idx, sl, 50)
    for graph in ((idx_d, sl_d), (dev(idx), dev(sl))):   # with and without the "not capped" hint of the ball query
        a, o = be.pg_bfs_cluster(dev(sem), graph[0], graph[1], 50)
        assert np.array_equal(o.cpu().numpy(), want[1]) and o.numel() - 1 == 2
        assert np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_bfs_directed_graph_leftover_clusters(be, oracle, seed):
    """a synthetic DIRECTED dense graph (one list at the 1000-entry cap switches the directed path on): in every block
    the low points form the cluster of the block's first point, the high points are in nobody's list from below and
    form further clusters among themselves whose seeds the serial order decides (chains, merges, singletons) -- stage 2
    of the chip-wide expansion (smallest-ancestor labels) must reproduce them, with and without semantic labels"""
    rng = np.random.default_rng(100 + seed)
    blocks, low, high = 24, 220, 90
    per = low + high
    n = blocks * per + 1000
    lists = []
    for bk in range(blocks):
        base = bk * per
        for i in range(per):
            tgt = set((base + rng.choice(low, 40, replace=False)).tolist())
            if i >= low:                                     # a high point: a few edges to other high points
                tgt |= set((base + low + rng.choice(high, rng.integers(0, 3), replace=False)).tolist())
            tgt.add(base + i)
            lists.append(np.array(sorted(tgt), np.int32))
    tail = blocks * per                                      # 1000 points that all list each other: capped lists
    for i in range(1000):
        lists.append(np.arange(tail, tail + 1000, dtype=np.int32))
    sl = np.zeros((n, 2), np.int32)
    sl[:, 1] = [len(x) for x in lists]
    sl[1:, 0] = np.cumsum(sl[:-1, 1])
    idx = np.concatenate(lists)
    assert idx.size >= 24 * n and sl[:, 1].max() == 1000
    sem = np.full(n, 2, np.int16)
    sem[rng.random(n) < 0.05] = 5                            # a second class cuts some of the edges
    want = oracle.pg_bfs_cluster(sem, idx, sl, 2)
    assert want[1].size - 1 > 3 * blocks                     # many leftover clusters of two and more points
    a, o = be.pg_bfs_cluster(dev(sem), dev(idx), dev(sl), 2)
    assert np.array_equal(o.cpu().numpy(), want[1])
    assert np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))
    mean = [10.0, 10.0, 10.0, 10.0, 10.0]
    want = oracle.sg_bfs_cluster(mean, idx, sl, 0.25, 1)     # size > 2.5
    a, o = be.sg_bfs_cluster(mean, dev(idx), dev(sl), 0.25, 1)
    assert np.array_equal(o.cpu().numpy(), want[1])
    assert np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))


@pytest.mark.parametrize("n", [1, 1000, 8192, 8193, 100001, 3000000, 40000000])
def test_single_launch_scan_matches_cumsum(n):
    """the one-launch exclusive scan (decoupled look-back, csrc/scan.hip) against torch.cumsum: out of place, in place,
    the total, repeated on the same stream (the state must come back zeroed) and concurrently on two streams"""
    import ctypes as C
    from minsu3d_amd import _lib
    lib = _lib.lib()
    lib.ms3d_scan_i32_workspace_bytes.restype = C.c_size_t
    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 50, (n,), generator=g, dtype=torch.int32).cuda()
    want = torch.cumsum(x.long(), 0) - x.long()
    assert int(want[-1] + x[-1]) < 2 ** 31
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    streams[1].wait_stream(streams[0])
    outs = []
    for rep in range(3):
        for s_ in streams:
            with torch.cuda.stream(s_):
                ws = torch.empty(lib.ms3d_scan_i32_workspace_bytes() + 256, dtype=torch.uint8, device="cuda")
                out = torch.empty_like(x) if rep < 2 else x.clone()
                total = torch.zeros(1, dtype=torch.int32, device="cuda")
                src = x if rep < 2 else out                      # rep 2: in place
                _lib.check(lib.ms3d_scan_i32(_lib.ptr(src), _lib.ptr(out), n, _lib.ptr(total), _lib.ptr(ws),
                                             C.c_void_p(s_.cuda_stream)), "ms3d_scan_i32")
                outs.append((out, total))
    torch.cuda.synchronize()
    for out, total in outs:
        assert torch.equal(out.long(), want) and int(total) == int(want[-1] + x[-1])


def _random_digraph(rng, n, deg_lo, deg_hi, shuffle_lists, shuffle_starts, local=None, down_frac=0.0):
    """adjacency lists nobody's ball query made: random out-neighbours (no symmetry), optionally unsorted lists and
    start offsets in arbitrary order (the reference's own ball query hands them out in atomic order).
    down_frac: that share of the points only names points BELOW itself -- ascending seeds cannot reach them from below,
    so a dense graph falls into many clusters whose membership and order the serial seed order decides"""
    lists = []
    for i in range(n):
        d = int(rng.integers(deg_lo, deg_hi + 1))
        pool = n if local is None else min(n, local)
        base = 0 if local is None else max(0, min(n - pool, i - pool // 2))
        if rng.random() < down_frac:
            base, pool = max(0, i - pool), min(pool, i) + 1
        tgt = base + rng.choice(pool, min(d, pool), replace=False)
        if rng.random() < 0.8:
            tgt = np.unique(np.append(tgt, i))            # most points list themselves, like a ball query's
        else:
            tgt = np.unique(tgt)
        if shuffle_lists:
            rng.shuffle(tgt)
        lists.append(tgt.astype(np.int32))
    order = rng.permutation(n) if shuffle_starts else np.arange(n)
    sl = np.zeros((n, 2), np.int32)
    pos = 0
    chunks = []
    for i in order:
        sl[i] = (pos, len(lists[i]))
        chunks.append(lists[i])
        pos += len(lists[i])
    return np.concatenate(chunks), sl


@pytest.mark.parametrize("n,deg,shuffle_lists,shuffle_starts,local,down", [
    (4000, (0, 3), False, False, None, 0.0),     # sparse, directed: the per-component replay
    (4000, (1, 4), True, True, None, 0.3),       # ... unsorted lists, lists laid out in arbitrary order
    (3000, (30, 60), False, False, 400, 0.0),    # dense (>= 24 edges per point), directed, NOT capped: was taken for symmetric
    (3000, (30, 60), True, True, 400, 0.97),     # ... unsorted, arbitrary layout, hundreds of clusters decided by the seed order
    (6000, (25, 40), False, True, 300, 0.9),
    (2500, (40, 1100), False, True, 1500, 0.9),  # lists beyond 1024 entries: the replay instead of the masked expansion
])
def test_bfs_on_arbitrary_directed_adjacency(be, oracle, n, deg, shuffle_lists, shuffle_starts, local, down):
    """pg / sg_bfs_cluster through the operator boundary on graphs NO ball query made.  The reference's host
    BFS accepts any adjacency lists (bfs_cluster.cpp:28-54: out-edge reachability from ascending seeds, members in FIFO
    order); rounds 1-4 silently assumed a symmetric graph whenever no list sat at the 1000 cap.  Without a hint from our own
    ball query nothing is assumed any more: bit-exact against the oracle -- and against the reference's OWN compiled
    bfs_cluster.cpp where oracle/_ref is built -- clusters, member order, offsets."""
    rng = np.random.default_rng(n + deg[1] + 7 * shuffle_lists)
    idx, sl = _random_digraph(rng, n, deg[0], deg[1], shuffle_lists, shuffle_starts, local, down)
    sem = rng.integers(2, 4, n).astype(np.int16)
    for thr in (1, 3):
        want = oracle.pg_bfs_cluster(sem, idx, sl, thr)
        if oracle.ref() is not None:
            live = oracle.pg_bfs_cluster(sem, idx, sl, thr, use_ref=True)
            assert np.array_equal(live[0].reshape(-1, 2), want[0].reshape(-1, 2)) and np.array_equal(live[1], want[1])
        a, o = be.pg_bfs_cluster(dev(sem), dev(idx), dev(sl), thr)
        assert np.array_equal(o.cpu().numpy(), want[1])
        assert np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))
    mean = [-1.0, 30.0, 100.0]
    for cid in range(3):
        want = oracle.sg_bfs_cluster(mean, idx, sl, 0.05, cid)
        a, o = be.sg_bfs_cluster(mean, dev(idx), dev(sl), 0.05, cid)
        assert np.array_equal(o.cpu().numpy(), want[1])
        assert np.array_equal(a.cpu().numpy().reshape(-1, 2), want[0].reshape(-1, 2))


def test_bfs_rejects_lists_it_cannot_cluster(be):
    """what the operator refuses instead of answering wrongly (the extension turns the code into an exception): a target
    outside [0, N), a list header outside the edge array, and a list that names one neighbour twice (the serial loop skips
    the second mention; the parallel claims would emit the member twice)"""
    from minsu3d_amd._lib import HipLibraryError
    rng = np.random.default_rng(3)
    n = 2000
    idx, sl = _random_digraph(rng, n, 2, 5, False, False)
    sem = np.full(n, 2, np.int16)
    be.pg_bfs_cluster(dev(sem), dev(idx), dev(sl), 1)                       # fine as it is
    bad = idx.copy(); bad[17] = n
    with pytest.raises(HipLibraryError):
        be.pg_bfs_cluster(dev(sem), dev(bad), dev(sl), 1)
    bad = idx.copy(); bad[5] = -3
    with pytest.raises(HipLibraryError):
        be.pg_bfs_cluster(dev(sem), dev(bad), dev(sl), 1)
    bad_sl = sl.copy(); bad_sl[n - 1, 1] += 10
    with pytest.raises(HipLibraryError):
        be.pg_bfs_cluster(dev(sem), dev(idx), dev(bad_sl), 1)
    # point 0 names point 1 twice, and point 1 is reached from 0 first
    lists = [np.array([0, 1, 1], np.int32), np.array([1, 2], np.int32)] + [np.array([i], np.int32) for i in range(2, 40)]
    sl2 = np.zeros((40, 2), np.int32)
    sl2[:, 1] = [len(x) for x in lists]
    sl2[1:, 0] = np.cumsum(sl2[:-1, 1])
    with pytest.raises(HipLibraryError):
        be.pg_bfs_cluster(dev(np.full(40, 2, np.int16)), dev(np.concatenate(lists)), dev(sl2), 1)
