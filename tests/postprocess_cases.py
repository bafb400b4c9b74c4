"""seeded proposal sets for the post-processing parity tests (CPU: oracle backend, GPU: HIP backend)"""
import numpy as np


def make_case(seed, n=6000, n_regions=9, per_region=5, junk=6):
    rng = np.random.default_rng(seed)
    bounds = np.sort(rng.choice(np.arange(1, n), n_regions - 1, replace=False))
    region = np.searchsorted(bounds, np.arange(n), side="right")
    rows, scores = [], []
    pid = 0
    for r in range(n_regions):
        members = np.flatnonzero(region == r)
        for _ in range(per_region):
            keep = members[rng.random(members.size) < rng.uniform(0.2, 1.0)]
            extra = rng.integers(0, n, int(members.size * rng.uniform(0, 0.2)))
            idx = np.unique(np.concatenate([keep, extra]))
            rng.shuffle(idx)                                     # BFS order is not sorted
            rows.append(np.stack([np.full(idx.size, pid), idx], 1))
            scores.append(rng.normal(1.0, 2.0))
            pid += 1
    for _ in range(junk):
        idx = np.unique(rng.integers(0, n, int(rng.integers(5, 300))))
        rows.append(np.stack([np.full(idx.size, pid), idx], 1))
        scores.append(rng.normal(0.0, 2.0))
        pid += 1
    proposals_idx = np.concatenate(rows).astype(np.int32)
    sem = rng.standard_normal((n, 20)).astype(np.float32)
    xyz = (rng.random((n, 3)) * 5).astype(np.float32)
    mask_scores = rng.normal(0.0, 1.0, proposals_idx.shape[0]).astype(np.float32)
    return dict(n=n, P=pid, proposals_idx=proposals_idx, scores=np.array(scores, np.float32), sem=sem, xyz=xyz,
                mask_scores=mask_scores)


def assert_same_instances(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["scan_id"] == w["scan_id"] and g["label_id"] == w["label_id"]
        assert g["pred_mask"] == w["pred_mask"]
        assert np.float32(g["conf"]) == np.float32(w["conf"]) or abs(float(g["conf"]) - float(w["conf"])) < 1e-6
        assert np.array_equal(np.asarray(g["pred_bbox"], np.float32), np.asarray(w["pred_bbox"], np.float32))


def make_softgroup_scores(seed, P, S, n_cls):
    """per-class heads of SoftGroup's refinement for the proposals of make_case(seed): classification logits [P, C+1],
    IoU scores [P, C+1], per-point mask scores [S, C+1]"""
    rng = np.random.default_rng(100 + seed)
    cls_scores = rng.normal(0, 2.5, (P, n_cls + 1)).astype(np.float32)
    iou_scores = rng.normal(0.6, 0.5, (P, n_cls + 1)).astype(np.float32)
    mask_scores = rng.normal(0.0, 1.0, (S, n_cls + 1)).astype(np.float32)
    return cls_scores, iou_scores, mask_scores


def mask_digest(rle):
    """{'length', 'counts'} run-length mask -> (length, point count, sha1 of the counts string)"""
    import hashlib
    runs = [int(x) for x in rle["counts"].split()]
    return [int(rle["length"]), int(sum(runs[1::2])), hashlib.sha1(rle["counts"].encode()).hexdigest()]
