"""Regenerate tests/golden/*.npz from the REFERENCE's own code (oracle/_ref, built by
oracle/build_ref.py from /root/reference).  Runs only in the build container; the fixtures it
writes are data (inputs + the reference's outputs), committed so that the GPU box -- which has no
/root/reference -- can still check against the reference.

    python tests/golden/make_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True          # nothing is written into /root/reference (no __pycache__ there)

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def scene(rng, n, kind):
    if kind == "surface":      # thin slab: ScanNet-like neighbour counts (~8-30)
        xyz = rng.random((n, 3)).astype(np.float32) * np.array([0.6, 0.6, 0.03], np.float32)
    elif kind == "blobs":      # shifted-coordinate look: tight blobs, long lists, some capped at 1000
        c = rng.random((6, 3)).astype(np.float32) * 2
        xyz = (c[rng.integers(0, 6, n)] + rng.standard_normal((n, 3)) * 0.012).astype(np.float32)
    elif kind == "capped":     # one blob, every list capped -> directed graph
        xyz = (rng.standard_normal((n, 3)) * 0.002).astype(np.float32)
    else:
        raise ValueError(kind)
    return xyz


def main():
    assert build_ref.build() is not None, "/root/reference is required to regenerate golden vectors"
    rng = np.random.default_rng(20261001)
    cases = {}
    specs = [("surface", 1500, 2, 0.03, 5), ("surface", 1200, 1, 0.05, 50), ("blobs", 2500, 2, 0.03, 50),
             ("capped", 1300, 1, 0.03, 50), ("surface", 64, 1, 0.03, 1), ("surface", 300, 3, 0.02, 1000)]
    for ci, (kind, n, B, radius, thr) in enumerate(specs):
        xyz = scene(rng, n, kind)
        bsz = np.sort(rng.integers(0, B, n)).astype(np.uint8)
        bo = np.concatenate([[0], np.cumsum(np.bincount(bsz, minlength=B))]).astype(np.int32)
        idx, sl = O.ballquery_batch_p(xyz, bsz, bo, radius)
        sem = rng.integers(2, 5, n).astype(np.int16)
        if kind == "capped":
            sem[:] = 3
        pg_i, pg_o = O.pg_bfs_cluster(sem, idx, sl, thr, use_ref=True)
        mean = np.array([-1, -1, 40, 400, 100], np.float32)
        sg = [O.sg_bfs_cluster(mean, idx, sl, 0.05, cid, use_ref=True) for cid in (0, 2, 3)]
        pna = np.array([-1, -1, 60, 300, 120], np.float32)
        ra = np.array([-1, -1, 0.05, 0.2, 0.1], np.float32)
        ha_i, ha_o = O.hierarchical_aggregation(sem, xyz, idx, sl, bsz, False, pna, ra, -1, use_ref=True)
        cases[f"c{ci}"] = dict(kind=kind, radius=radius, thr=thr)
        np.savez_compressed(
            os.path.join(OUT, f"bfs_case{ci}.npz"), xyz=xyz, batch_idxs=bsz, batch_offsets=bo, radius=np.float32(radius),
            threshold=np.int32(thr), sem=sem, ball_idx=idx, start_len=sl, pg_idxs=pg_i, pg_offsets=pg_o,
            sg_mean=mean, sg_threshold=np.float32(0.05), sg_class_ids=np.array([0, 2, 3], np.int32),
            sg0_idxs=sg[0][0], sg0_offsets=sg[0][1], sg1_idxs=sg[1][0], sg1_offsets=sg[1][1],
            sg2_idxs=sg[2][0], sg2_offsets=sg[2][1], point_num_avg=pna, radius_avg=ra, hais_idxs=ha_i,
            hais_offsets=ha_o)
        print(ci, kind, n, "edges", idx.size, "max len", sl[:, 1].max(), "pg clusters", pg_o.size - 1,
              "hais clusters", ha_o.size - 1)
    # hand KAT from SURVEY 8c (verified against the built reference)
    nb = {0: [0, 1], 1: [0, 1, 2], 2: [1, 2], 3: [3, 4], 4: [3, 4], 5: [5]}
    idx, sl = [], []
    for i in range(6):
        sl.append([len(idx), len(nb[i])]); idx += nb[i]
    sem = np.array([3, 3, 3, 3, 3, 4], np.int16)
    a, b = O.pg_bfs_cluster(sem, idx, sl, 2, use_ref=True)
    assert a.tolist() == [[0, 0], [0, 1], [0, 2], [1, 3], [1, 4]] and b.tolist() == [0, 3, 5]
    np.savez_compressed(os.path.join(OUT, "bfs_kat.npz"), sem=sem, ball_idx=np.array(idx, np.int32),
                        start_len=np.array(sl, np.int32), threshold=np.int32(2), pg_idxs=a, pg_offsets=b)


if __name__ == "__main__":
    main()
