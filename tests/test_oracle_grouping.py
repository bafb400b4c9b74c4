"""CPU: the plain-C oracle against the golden vectors produced by the REFERENCE's own code
(tests/golden/make_golden.py) and, when oracle/_ref is present, against the reference live."""
import glob
import os

import numpy as np
import pytest


def _cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "bfs_case*.npz")))


def test_golden_files_present(golden_dir):
    assert len(_cases(golden_dir)) >= 6 and os.path.exists(os.path.join(golden_dir, "bfs_kat.npz"))


def test_bfs_kat(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "bfs_kat.npz"))
    a, b = oracle.pg_bfs_cluster(g["sem"], g["ball_idx"], g["start_len"], int(g["threshold"]))
    assert np.array_equal(a, g["pg_idxs"]) and np.array_equal(b, g["pg_offsets"])


@pytest.mark.parametrize("ci", range(6))
def test_oracle_vs_golden(oracle, golden_dir, ci):
    g = np.load(os.path.join(golden_dir, f"bfs_case{ci}.npz"))
    # ball query (oracle, canonical form) reproduces the graph the reference BFS consumed
    idx, sl = oracle.ballquery_batch_p(g["xyz"], g["batch_idxs"], g["batch_offsets"], float(g["radius"]))
    assert np.array_equal(idx, g["ball_idx"]) and np.array_equal(sl, g["start_len"])
    a, b = oracle.pg_bfs_cluster(g["sem"], idx, sl, int(g["threshold"]))
    assert np.array_equal(a, g["pg_idxs"]) and np.array_equal(b, g["pg_offsets"])
    for k, cid in enumerate(g["sg_class_ids"]):
        a, b = oracle.sg_bfs_cluster(g["sg_mean"], idx, sl, float(g["sg_threshold"]), int(cid))
        assert np.array_equal(a, g[f"sg{k}_idxs"]) and np.array_equal(b, g[f"sg{k}_offsets"])
    a, b = oracle.hierarchical_aggregation(g["sem"], g["xyz"], idx, sl, g["batch_idxs"], False, g["point_num_avg"],
                                           g["radius_avg"])
    assert np.array_equal(a, g["hais_idxs"]) and np.array_equal(b, g["hais_offsets"])


def test_oracle_vs_reference_live(oracle):
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(7)
    for trial in range(4):
        n = 2000
        xyz = (rng.random((n, 3)) * np.array([0.5, 0.5, 0.05])).astype(np.float32)
        b = np.zeros(n, np.uint8); bo = np.array([0, n], np.int32)
        idx, sl = oracle.ballquery_batch_p(xyz, b, bo, 0.02 + 0.01 * trial)
        sem = rng.integers(0, 3, n).astype(np.int16)
        for thr in (1, 7, 50):
            r0 = oracle.pg_bfs_cluster(sem, idx, sl, thr)
            r1 = oracle.pg_bfs_cluster(sem, idx, sl, thr, use_ref=True)
            assert np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1])
        pna = np.array([30, 100, 300], np.float32); ra = np.array([0.05, 0.1, 0.2], np.float32)
        r0 = oracle.hierarchical_aggregation(sem, xyz, idx, sl, b, False, pna, ra)
        r1 = oracle.hierarchical_aggregation(sem, xyz, idx, sl, b, False, pna, ra, use_ref=True)
        assert np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1])


@pytest.mark.parametrize("set_aggr", [False, True])
def test_hais_raw_lists_are_consistent_with_the_merged_result(oracle, golden_dir, set_aggr):
    """parts=True (the lists hierarchical_aggregation.cpp:133-175 leaves in the caller's tensors): merged the way the
    wrapper merges them they are the oracle's merged result -- which the golden case pins for point aggregation --
    and every primary's post list starts with the primary itself"""
    g = np.load(os.path.join(golden_dir, "bfs_case1.npz"))
    args = (g["sem"], g["xyz"], g["ball_idx"], g["start_len"], g["batch_idxs"], set_aggr, g["point_num_avg"], g["radius_avg"])
    parts = oracle.hierarchical_aggregation(*args, parts=True)
    merged = oracle.hierarchical_aggregation(*args)
    ki, ko = parts["kept"][:2]
    pi, po = parts["post" if set_aggr else "primary"][:2]
    pi = pi.copy(); pi[:, 0] += len(ko) - 1
    assert np.array_equal(np.concatenate([ki, pi]), merged[0].reshape(-1, 2))
    assert np.array_equal(np.concatenate([ko, po[1:] + ko[-1]]), merged[1])
    if not set_aggr:
        assert np.array_equal(merged[0], g["hais_idxs"]) and parts["post"][0].shape[0] == 0
    else:
        raw_i, raw_o = parts["primary"][:2]
        for c in range(len(raw_o) - 1):
            m = raw_o[c + 1] - raw_o[c]
            assert np.array_equal(parts["post"][0][po[c]:po[c] + m, 1], raw_i[raw_o[c]:raw_o[c + 1], 1])
    fi, fo, fc = parts["fragment"]
    assert set(map(tuple, ki[:, 1:])) <= set(map(tuple, fi[:, 1:])) and fc.shape == (len(fo) - 1, 5)


def test_ballquery_brute_force_numpy(oracle):
    """independent check of the oracle's ball query against a numpy f64-free restatement"""
    rng = np.random.default_rng(3)
    n = 700
    xyz = (rng.random((n, 3)) * 0.3).astype(np.float32)
    b = np.sort(rng.integers(0, 3, n)).astype(np.uint8)
    bo = np.concatenate([[0], np.cumsum(np.bincount(b, minlength=3))]).astype(np.int32)
    r = np.float32(0.04)
    idx, sl = oracle.ballquery_batch_p(xyz, b, bo, float(r))
    for i in range(0, n, 37):
        d = xyz[i] - xyz
        d2 = np.float32(d[:, 0] * d[:, 0])
        # fmaf chain emulated in float64 then rounded once per fma (exact for these magnitudes)
        d2 = (d[:, 1].astype(np.float64) * d[:, 1] + d2).astype(np.float32)
        d2 = (d[:, 2].astype(np.float64) * d[:, 2] + d2).astype(np.float32)
        want = np.nonzero((d2 < r * r) & (b == b[i]))[0]
        got = idx[sl[i, 0]:sl[i, 0] + sl[i, 1]]
        assert np.array_equal(got, want[:1000])


def test_segment_ops_small(oracle):
    rng = np.random.default_rng(5)
    S, C = 200, 5
    x = rng.standard_normal((S, C)).astype(np.float32)
    off = np.array([0, 3, 3, 50, 200], np.int32)
    m = oracle.sec_mean(x, off)
    for p in range(4):
        seg = x[off[p]:off[p + 1]]
        if len(seg):
            acc = np.zeros(C, np.float32)
            for row in seg:
                acc = acc + row / np.float32(len(seg))
            assert np.array_equal(m[p], acc)
    mx, am = oracle.roipool_fp(x, off)
    assert np.array_equal(mx[2], x[3:50].max(0)) and np.array_equal(am[2], 3 + x[3:50].argmax(0))
    assert np.all(am[1] == -1) and np.all(np.isneginf(mx[1]))
    assert np.array_equal(oracle.sec_min(x, off)[3], x[50:200].min(0))


def test_iou_small(oracle):
    prop_idx = np.array([0, 1, 2, 3, 4, 5, 6], np.int32)
    prop_off = np.array([0, 4, 7], np.int32)
    inst = np.array([0, 0, 1, -1, 1, 1, 1, 0], np.int16)
    pn = np.array([3, 4], np.int32)
    iou = oracle.get_iou(prop_idx, prop_off, inst, pn)
    want = np.array([[2 / (4 + 3 - 2 + 1e-5), 1 / (4 + 4 - 1 + 1e-5)], [0 / (3 + 3 + 1e-5), 3 / (3 + 4 - 3 + 1e-5)]])
    assert np.allclose(iou, want.astype(np.float32), rtol=0, atol=1e-7)
    ml, mlm = oracle.get_mask_label(prop_idx, prop_off, inst, np.array([5, -1], np.int16), iou, -1, 0.3)
    assert mlm.tolist() == [True] * 4 + [False] * 3 and ml.tolist() == [True, True, False, False, False, False, False]
