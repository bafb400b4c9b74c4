"""GPU: the reference's callers run unchanged against the drop-in modules.

Each test REPLAYS the call sequence of a reference wrapper / model method (restated here -- the reference tree does
not exist on the GPU box) with the tensors, devices and ownership rules the reference uses (caller-allocated outputs,
`resize_` by the callee, CPU tensors for the clustering functions), and checks the result against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["ctypes module", "C++ extension"])
def OPS(request):
    """both forms of `import COMMON_OPS`: the Python module over ctypes (minsu3d_amd/dropin/COMMON_OPS.py) and the
    PyTorch-ROCm C++ extension (minsu3d_amd/csrc_host/common_ops_ext.cpp -> minsu3d_amd/dropin_ext/COMMON_OPS.so)"""
    import minsu3d_amd.dropin as dropin
    if request.param == "C++ extension":
        ext = dropin.load_extension()
        assert ext.__file__.endswith(".so")
        return ext
    dropin.install()
    import COMMON_OPS
    return COMMON_OPS


def _scene(seed, n=12000, B=2):
    rng = np.random.default_rng(seed)
    c = rng.random((10, 3)) * np.array([3.0, 3.0, 1.0])
    xyz = (c[rng.integers(0, 10, n)] + rng.standard_normal((n, 3)) * 0.05).astype(np.float32)
    b = np.sort(rng.integers(0, B, n)).astype(np.uint8)
    bo = np.concatenate([[0], np.cumsum(np.bincount(b, minlength=B))]).astype(np.int32)
    sem = rng.integers(2, 5, n).astype(np.int16)
    return xyz, b, bo, sem


def _ballquery(OPS, coords, batch_idxs, batch_offsets, radius, capacity):
    """the caller side of COMMON_OPS.ballquery_batch_p (functions/common_ops.py:11-47): outputs sized by a guessed mean
    list length; the call reports the true total and is repeated with a sufficient size when the guess was too small"""
    n = coords.size(0)
    for attempt in range(3):
        idx = torch.zeros(n * capacity, dtype=torch.int32, device="cuda")
        start_len = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
        total = OPS.ballquery_batch_p(coords, batch_idxs, batch_offsets, idx, start_len, n, capacity, radius)
        if total <= idx.numel():
            return idx[:total], start_len, attempt
        capacity = total // n + 1
    raise AssertionError("ball query did not fit after resizing")


def test_pointgroup_call_sequence(OPS, oracle):
    """model/pointgroup.py:43-55: ball query on the GPU, `.cpu()`, pg_bfs_cluster on CPU tensors with empty_like outputs
    (functions/pointgroup_ops.py:8-29)"""
    xyz, b, bo, sem = _scene(0)
    idx, start_len, retries = _ballquery(OPS, torch.from_numpy(xyz).cuda(), torch.from_numpy(b).cuda(),
                                         torch.from_numpy(bo).cuda(), 0.04, 2)
    assert retries == 1                                                    # meanActive = 2 is too small: one resize
    widx, wsl = oracle.ballquery_batch_p(xyz, b, bo, 0.04)
    assert np.array_equal(idx.cpu().numpy(), widx) and np.array_equal(start_len.cpu().numpy(), wsl)
    semantic_label = torch.from_numpy(sem)                                 # CPU, as semantic_preds_cpu
    ball_query_idxs, start_len = idx.cpu(), start_len.cpu()
    N = start_len.size(0)
    cluster_idxs = torch.empty_like(semantic_label, dtype=torch.int32)
    cluster_offsets = torch.empty_like(semantic_label, dtype=torch.int32)
    OPS.pg_bfs_cluster(semantic_label, ball_query_idxs, start_len, cluster_idxs, cluster_offsets, N, 30)
    want = oracle.pg_bfs_cluster(sem, widx, wsl, 30)
    assert not cluster_idxs.is_cuda and cluster_idxs.dtype == torch.int32
    assert cluster_idxs.shape == (want[0].reshape(-1, 2).shape[0], 2) and cluster_offsets.shape == (len(want[1]),)
    assert np.array_equal(cluster_idxs.numpy(), want[0].reshape(-1, 2)) and np.array_equal(cluster_offsets.numpy(), want[1])
    # the wrapper of this package with the reference's CPU tensors: same result, returned on the CPU
    from minsu3d_amd.common_ops.functions import pointgroup_ops
    a, o = pointgroup_ops.pg_bfs_cluster(semantic_label, ball_query_idxs, start_len, 30)
    assert not a.is_cuda and torch.equal(a, cluster_idxs) and torch.equal(o, cluster_offsets)


def test_host_round_trip_clusters_what_it_is_given_by_default(OPS, oracle, monkeypatch):
    """model/pointgroup.py:43-55 verbatim data flow: ball query on the GPU, `.cpu()` of both results, clustering on the
    HOST tensors.  By default the callee clusters exactly the tensors it is handed: a host copy that was edited (one
    point cut out of the graph) gives the edited graph's clusters, not the device original's."""
    monkeypatch.delenv("MS3D_DROPIN_REUSE", raising=False)
    shim = hasattr(OPS, "_GRAPHS")            # (the C++ extension has no reuse shortcut at all)
    if shim:
        OPS._GRAPHS.clear()
    xyz, b, bo, sem = _scene(4, n=30000, B=2)
    idx, start_len, _ = _ballquery(OPS, torch.from_numpy(xyz).cuda(), torch.from_numpy(b).cuda(),
                                   torch.from_numpy(bo).cuda(), 0.05, 40)
    assert not shim or not OPS._GRAPHS                            # nothing is remembered unless asked for
    idx_cpu, sl_cpu = idx.cpu(), start_len.cpu()
    # the edit: cut the busiest point out of the graph (its own list and every mention of it in its neighbours' lists are
    # overwritten with self references -- sizes unchanged, the graph stays symmetric as a ball query's is)
    edited = idx_cpu.clone()
    victim = int(torch.argmax(sl_cpu[:, 1]))
    s0, l0 = (int(v) for v in sl_cpu[victim])
    for v in idx_cpu[s0:s0 + l0].tolist():
        sv, lv = (int(t) for t in sl_cpu[v])
        seg = edited[sv:sv + lv]
        seg[seg == victim] = v
    edited[s0:s0 + l0] = victim
    hits0 = list(OPS._REUSE_HITS) if shim else None
    out = [torch.empty(0, dtype=torch.int32), torch.empty(0, dtype=torch.int32)]
    OPS.pg_bfs_cluster(torch.from_numpy(sem), edited, sl_cpu, out[0], out[1], len(sem), 30)
    assert not shim or OPS._REUSE_HITS == hits0                   # the shortcut was not even consulted
    want = oracle.pg_bfs_cluster(sem, edited.numpy(), sl_cpu.numpy(), 30)
    assert np.array_equal(out[0].numpy(), want[0].reshape(-1, 2)) and np.array_equal(out[1].numpy(), want[1])
    orig = oracle.pg_bfs_cluster(sem, idx_cpu.numpy(), sl_cpu.numpy(), 30)
    assert victim in orig[0].reshape(-1, 2)[:, 1] and victim not in out[0][:, 1].numpy()     # the edit was honoured


def test_host_round_trip_reuses_the_device_graph_when_asked(OPS, oracle, monkeypatch):
    """MS3D_DROPIN_REUSE=1: the module recognises the host copies of its own last results by a checksum over EVERY entry
    and clusters the device originals (no upload of the neighbour list); a host tensor edited at ANY single position --
    the adversarial case of a sampled fingerprint -- is uploaded and clustered as given."""
    if not hasattr(OPS, "_GRAPHS"):
        pytest.skip("the opt-in reuse shortcut exists in the ctypes module only")
    monkeypatch.setenv("MS3D_DROPIN_REUSE", "1")
    OPS._GRAPHS.clear()
    xyz, b, bo, sem = _scene(4, n=30000, B=2)
    idx, start_len, _ = _ballquery(OPS, torch.from_numpy(xyz).cuda(), torch.from_numpy(b).cuda(),
                                   torch.from_numpy(bo).cuda(), 0.05, 40)
    widx, wsl = oracle.ballquery_batch_p(xyz, b, bo, 0.05)
    idx_cpu, sl_cpu = idx.cpu(), start_len.cpu()
    assert np.array_equal(idx_cpu.numpy(), widx) and len(OPS._GRAPHS) == 1
    # host tensors that are NOT the remembered result: same sizes, ONE entry changed, at seeded random positions of either
    # tensor (first, last, anywhere) -> never taken for the device graph
    hits0 = list(OPS._REUSE_HITS)
    rng = np.random.default_rng(3)
    positions = [0, idx_cpu.numel() - 1] + [int(p) for p in rng.integers(0, idx_cpu.numel(), 6)]
    for n_try, p in enumerate(positions):
        other = idx_cpu.clone()
        other[p] += 1 if other[p] < len(sem) - 1 else -1
        a_, b_ = OPS._device_graph(other, sl_cpu)
        assert a_ is other and b_ is sl_cpu and OPS._REUSE_HITS == [hits0[0], hits0[1] + n_try + 1]
    other_sl = sl_cpu.clone()
    other_sl[int(rng.integers(0, sl_cpu.size(0))), 1] += 1
    a_, b_ = OPS._device_graph(idx_cpu, other_sl)
    assert a_ is idx_cpu and b_ is other_sl
    # the untouched copies ARE recognised: device originals taken, used once, then forgotten
    hits0 = list(OPS._REUSE_HITS)
    out = [torch.empty(0, dtype=torch.int32), torch.empty(0, dtype=torch.int32)]
    OPS.pg_bfs_cluster(torch.from_numpy(sem), idx_cpu, sl_cpu, out[0], out[1], len(sem), 30)
    assert OPS._REUSE_HITS == [hits0[0] + 1, hits0[1]] and not OPS._GRAPHS
    want = oracle.pg_bfs_cluster(sem, widx, wsl, 30)
    assert np.array_equal(out[0].numpy(), want[0].reshape(-1, 2)) and np.array_equal(out[1].numpy(), want[1])
    # ... and a graph of another ball query (the oracle's, smaller radius) is uploaded and clustered as given
    widx2, wsl2 = oracle.ballquery_batch_p(xyz, b, bo, 0.03)
    out2 = [torch.empty(0, dtype=torch.int32), torch.empty(0, dtype=torch.int32)]
    OPS.pg_bfs_cluster(torch.from_numpy(sem), torch.from_numpy(widx2), torch.from_numpy(wsl2), out2[0], out2[1], len(sem), 10)
    want2 = oracle.pg_bfs_cluster(sem, widx2, wsl2, 10)
    assert np.array_equal(out2[0].numpy(), want2[0].reshape(-1, 2)) and np.array_equal(out2[1].numpy(), want2[1])


def test_softgroup_call_sequence(OPS, oracle):
    """functions/softgroup_ops.py:18-30: outputs are `ball_query_idxs.new()` (empty), the class means a CPU float tensor"""
    xyz, b, bo, _ = _scene(1, n=8000, B=1)
    widx, wsl = oracle.ballquery_batch_p(xyz, b, bo, 0.04)
    ball_query_idxs, start_len = torch.from_numpy(widx), torch.from_numpy(wsl)     # CPU, model/softgroup.py:60-63
    mean = [-1.0, 300.0, 2000.0]
    for class_id in range(3):
        cluster_idxs = ball_query_idxs.new()
        cluster_offsets = ball_query_idxs.new()
        OPS.sg_bfs_cluster(torch.tensor(mean, dtype=torch.float32), ball_query_idxs, start_len, cluster_idxs,
                           cluster_offsets, start_len.size(0), 0.05, class_id)
        want = oracle.sg_bfs_cluster(mean, widx, wsl, 0.05, class_id)
        assert np.array_equal(cluster_idxs.numpy().reshape(-1, 2), want[0].reshape(-1, 2))
        assert np.array_equal(cluster_offsets.numpy(), want[1])


@pytest.mark.parametrize("using_set_aggr", [False, True])
def test_hais_call_sequence(OPS, oracle, using_set_aggr):
    """COMMON_OPS.hierarchical_aggregation on CPU tensors with the 11 caller-provided outputs (the wrapper's calling
    convention, functions/hais_ops.py:22-53): every list the reference leaves in them (hierarchical_aggregation.cpp:133-175)
    against the oracle's raw lists"""
    xyz, b, bo, sem = _scene(2)
    widx, wsl = oracle.ballquery_batch_p(xyz, b, bo, 0.04)
    point_num_avg = [100.0, 200.0, 400.0, 800.0, 1600.0]
    radius_avg = [0.1, 0.2, 0.3, 0.5, 0.8]
    out = {}
    for group in ("fragment", "kept", "primary", "post"):           # (idxs, offsets[, centres]) per list, all empty
        out[group] = [torch.empty(0, dtype=torch.int32), torch.empty(0, dtype=torch.int32)]
        if group != "post":
            out[group].append(torch.empty(0, dtype=torch.float32))
    OPS.hierarchical_aggregation(torch.from_numpy(sem), torch.from_numpy(xyz), torch.from_numpy(b), torch.from_numpy(widx),
                                 torch.from_numpy(wsl), *out["fragment"], *out["kept"], *out["primary"], *out["post"],
                                 torch.tensor(point_num_avg), torch.tensor(radius_avg), len(sem), int(using_set_aggr), -1)
    want = oracle.hierarchical_aggregation(sem, xyz, widx, wsl, b, using_set_aggr, point_num_avg, radius_avg, parts=True)
    assert len(want["kept"][1]) > 2 and len(want["primary"][1]) > 3          # the case is not degenerate
    groups = ("kept", "primary") + (("fragment", "post") if using_set_aggr else ())
    for group in groups:
        got = [t.numpy() for t in out[group]]
        wi, wo = want[group][0], want[group][1]
        assert np.array_equal(got[1], wo), group
        if group == "post":     # allocated for every fragment + primary point, zero beyond the last offset (.cpp:166-169)
            assert got[0].shape[0] == want["fragment"][0].shape[0] + want["primary"][0].shape[0]
            assert not got[0][wo[-1]:].any()
            got[0] = got[0][:wo[-1]]
        assert np.array_equal(got[0].reshape(-1, 2), wi), group
        if group != "post":     # centres: serial f32 sums in BFS order / size, class and scene of the seed (.cpp:84-88)
            assert got[2].shape == (len(wo) - 1, 5) and np.array_equal(got[2], want[group][2]), group
    if not using_set_aggr:      # the early return at .cpp:146-148 leaves these untouched
        assert all(t.numel() == 0 for t in out["fragment"] + out["post"])
    else:
        assert want["post"][0].shape[0] > want["primary"][0].shape[0]        # something was absorbed


def test_caller_allocated_outputs(OPS, oracle):
    """functions/common_ops.py:50-173, softgroup_ops.py:40-77: outputs zero-allocated by the wrapper, filled in place"""
    rng = np.random.default_rng(3)
    lens = rng.integers(1, 300, 64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    S, C_ = int(off[-1]), 16
    x = rng.standard_normal((S, C_)).astype(np.float32)
    inp, offsets = torch.from_numpy(x).cuda(), torch.from_numpy(off).cuda()
    nProposal = offsets.size(0) - 1
    for name in ("sec_mean", "sec_min", "sec_max"):
        out = torch.zeros((nProposal, C_), dtype=torch.float32, device="cuda")
        getattr(OPS, name)(inp, offsets, out, nProposal, C_)
        assert np.array_equal(out.cpu().numpy(), getattr(oracle, name)(x, off))
    output_feats = torch.zeros((nProposal, C_), dtype=torch.float32, device="cuda")
    output_maxidx = torch.zeros((nProposal, C_), dtype=torch.int32, device="cuda")
    OPS.roipool_fp(inp, offsets, output_feats, output_maxidx, nProposal, C_)
    wf, wm = oracle.roipool_fp(x, off)
    assert np.array_equal(output_feats.cpu().numpy(), wf) and np.array_equal(output_maxidx.cpu().numpy(), wm)
    g = rng.standard_normal((nProposal, C_)).astype(np.float32)
    d_feats = torch.zeros((S, C_), dtype=torch.float32, device="cuda")
    OPS.roipool_bp(d_feats, offsets, output_maxidx, torch.from_numpy(g).cuda(), nProposal, C_)
    assert np.array_equal(d_feats.cpu().numpy(), oracle.roipool_bp(g, off, wm, S))
    output_feats.zero_()
    OPS.global_avg_pool_fp(inp, offsets, output_feats, nProposal, C_)
    assert np.array_equal(output_feats.cpu().numpy(), oracle.global_avg_pool_fp(x, off))
    d_feats.zero_()
    OPS.global_avg_pool_bp(d_feats, offsets, torch.from_numpy(g).cuda(), nProposal, C_)
    assert np.array_equal(d_feats.cpu().numpy(), oracle.global_avg_pool_bp(g, off, S))
    # IoU family + mask labels
    N, I = 20000, 23
    inst = rng.integers(-1, I, N).astype(np.int16)
    pn = np.bincount(inst[inst >= 0], minlength=I).astype(np.int32)
    pidx = rng.integers(0, N, S).astype(np.int32)
    cls = rng.integers(-1, 18, I).astype(np.int16)
    sig = rng.random(S).astype(np.float32)
    d = lambda a: torch.from_numpy(a).cuda()
    for name, extra, wextra in (("get_iou", (), ()), ("get_mask_iou_on_cluster", (), ()),
                                ("get_mask_iou_on_pred", (d(sig),), (sig,))):
        proposals_iou = torch.zeros((nProposal, I), dtype=torch.float32, device="cuda")
        getattr(OPS, name)(d(pidx), offsets, d(inst), d(pn), proposals_iou, I, nProposal, *extra)
        assert np.array_equal(proposals_iou.cpu().numpy(), getattr(oracle, name)(pidx, off, inst, pn, *wextra))
    iou = torch.zeros((nProposal, I), dtype=torch.float32, device="cuda")
    OPS.get_iou(d(pidx), offsets, d(inst), d(pn), iou, I, nProposal)
    mask_label = torch.zeros(pidx.shape, dtype=torch.bool, device="cuda")
    mask_label_mask = torch.zeros(pidx.shape, dtype=torch.bool, device="cuda")
    OPS.get_mask_label(d(pidx), offsets, d(inst), d(cls), iou, I, nProposal, -1, 0.05, mask_label, mask_label_mask)
    wml, wmlm = oracle.get_mask_label(pidx, off, inst, cls, iou.cpu().numpy(), -1, 0.05)
    assert np.array_equal(mask_label.cpu().numpy(), wml) and np.array_equal(mask_label_mask.cpu().numpy(), wmlm)


def test_minkowski_alias_runs_a_reference_shaped_block():
    """model/module/common.py:21-48 (ResidualBlock) written against `import MinkowskiEngine as ME`"""
    import minsu3d_amd.dropin as dropin
    dropin.install()
    import MinkowskiEngine as ME
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    coords = np.unique(rng.integers(0, 20, (4000, 3)), axis=0).astype(np.int32)
    coords = np.concatenate([np.zeros((len(coords), 1), np.int32), coords], 1)
    feats = torch.randn(len(coords), 16, device="cuda")
    x = ME.SparseTensor(features=feats, coordinates=torch.from_numpy(coords).cuda())
    block = torch.nn.Sequential(ME.MinkowskiBatchNorm(16), ME.MinkowskiReLU(inplace=True),
                                ME.MinkowskiConvolution(16, 16, kernel_size=3, dimension=3)).cuda()
    y = block(x)
    y += x
    z = ME.cat(y, x)
    assert z.F.shape == (len(coords), 32) and torch.equal(z.C, x.C) and torch.isfinite(z.features).all()
