"""Synthetic "ScanNet-shaped" scenes: a 5 m x 4 m room (floor + four 1.6 m wall strips) with 12
axis-aligned boxes resting on the floor, points sampled uniformly on every face + 3 mm jitter, voxelised at
2 cm.  Produces exactly the batch dictionary the reference's collate function hands to the model
(reference data/data_module.py:83-98, schema in SURVEY Appendix C).  Harness code: numpy on the host."""
import numpy as np


def _sample_rect(rng, origin, eu, ev, density):
    area = np.linalg.norm(np.cross(eu, ev))
    n = max(int(round(area * density)), 1)
    uv = rng.random((n, 2))
    return origin + uv[:, :1] * eu + uv[:, 1:] * ev


def make_scene(seed, room=(5.0, 4.0), n_boxes=12, density=1700.0, wall_h=1.6, n_box_classes=18, class_colours=False):
    """-> dict(xyz f32[N,3], rgb f32[N,3] in [-1,1], sem_labels i16[N], instance_ids i16[N]).
    class_colours: the colour of a point tells its class (palette + N(0, 0.15)) -- the LEARNABLE variant used by the
    convergence test; the benchmark's scenes keep uniform random colours (labels there are not predictable)."""
    rng = np.random.default_rng(seed)
    L, Wd = room
    pts, sem, inst = [], [], []

    def add(p, s, i):
        pts.append(p); sem.append(np.full(len(p), s, np.int16)); inst.append(np.full(len(p), i, np.int16))

    add(_sample_rect(rng, np.zeros(3), np.array([L, 0, 0.]), np.array([0, Wd, 0.]), density), 0, -1)
    for o, eu in ((np.zeros(3), np.array([L, 0, 0.])), (np.array([0, Wd, 0.]), np.array([L, 0, 0.])),
                  (np.zeros(3), np.array([0, Wd, 0.])), (np.array([L, 0, 0.]), np.array([0, Wd, 0.]))):
        add(_sample_rect(rng, o, eu, np.array([0, 0, wall_h]), density), 1, -1)
    for b in range(n_boxes):
        size = rng.uniform([0.4, 0.4, 0.4], [1.6, 0.9, 1.0])
        size = np.minimum(size, [L * 0.6, Wd * 0.6, 1.2])
        lo = np.array([rng.uniform(0.05, L - size[0] - 0.05), rng.uniform(0.05, Wd - size[1] - 0.05), 0.0])
        hi = lo + size
        sx, sy, sz = np.array([size[0], 0, 0.]), np.array([0, size[1], 0.]), np.array([0, 0, size[2]])
        faces = [(np.array([lo[0], lo[1], hi[2]]), sx, sy),                      # top (bottom face omitted)
                 (lo, sx, sz), (np.array([lo[0], hi[1], 0.]), sx, sz), (lo, sy, sz), (np.array([hi[0], lo[1], 0.]), sy, sz)]
        p = np.concatenate([_sample_rect(rng, o, eu, ev, density) for o, eu, ev in faces], 0)
        add(p, 2 + (b % n_box_classes), b)
    xyz = np.concatenate(pts, 0) + rng.normal(0, 0.003, (sum(len(p) for p in pts), 3))
    perm = rng.permutation(len(xyz))           # scan order is not face order
    xyz = xyz[perm].astype(np.float32)
    sem = np.concatenate(sem)[perm]
    inst = np.concatenate(inst)[perm]
    rgb = rng.uniform(-1, 1, (len(xyz), 3)).astype(np.float32)
    if class_colours:
        palette = np.random.default_rng(12345).uniform(-0.8, 0.8, (20, 3))
        rgb = np.clip(palette[sem] + rng.normal(0, 0.15, (len(xyz), 3)), -1, 1).astype(np.float32)
    xyz -= xyz.mean(0)                          # general_dataset.py:24
    return dict(xyz=xyz, rgb=rgb, sem_labels=sem, instance_ids=inst)


def first_occurrence_unique(keys):
    """(unique_idx in first-occurrence order, inverse) of integer rows -- the canonical sparse_quantize"""
    _, idx, inv = np.unique(keys, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(idx, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    return idx[order], rank[inv.reshape(-1)]


def collate(scenes, voxel_size=0.02, ignore_classes=(1, 2)):
    """batch dictionary with the reference's keys/dtypes (numpy arrays; caller moves them to the device)"""
    out = {k: [] for k in ("point_xyz", "vert_batch_ids", "sem_labels", "instance_ids", "instance_center_xyz",
                           "instance_num_point", "instance_semantic_cls", "voxel_xyz", "voxel_features",
                           "voxel_point_map")}
    inst_off, vox_off, inst_offsets = 0, 0, [0]
    for b, s in enumerate(scenes):
        xyz, rgb, sem, inst = s["xyz"], s["rgb"], s["sem_labels"], s["instance_ids"].astype(np.int64).copy()
        n = len(xyz)
        centers = np.zeros((n, 3), np.float32)
        ids = np.unique(inst[inst >= 0])
        npoint, cls = [], []
        for new, i in enumerate(ids):
            m = inst == i
            centers[m] = xyz[m].mean(0)
            npoint.append(int(m.sum()))
            c = int(sem[m][0])
            cls.append(c - len(ignore_classes) if c >= 0 else -1)
        remap = np.full(int(inst.max()) + 2, -1, np.int64)
        remap[ids] = np.arange(len(ids)) + inst_off
        inst = np.where(inst >= 0, remap[np.maximum(inst, 0)], -1)
        inst_off += len(ids)
        inst_offsets.append(inst_off)
        vox = np.floor((xyz - xyz.min(0)) / voxel_size).astype(np.int32)       # general_dataset.py:109,159-163
        uidx, inv = first_occurrence_unique(vox)
        feats = np.concatenate([rgb, xyz], 1)[uidx].astype(np.float32)          # [rgb, xyz] of the first point per voxel
        out["voxel_xyz"].append(np.concatenate([np.full((len(uidx), 1), b, np.int32), vox[uidx]], 1))
        out["voxel_features"].append(feats)
        out["voxel_point_map"].append(inv.astype(np.int64) + vox_off)
        vox_off += len(uidx)
        out["point_xyz"].append(xyz)
        out["vert_batch_ids"].append(np.full(n, b, np.uint8))
        out["sem_labels"].append(sem.astype(np.int16))
        out["instance_ids"].append(inst.astype(np.int16))
        out["instance_center_xyz"].append(centers)
        out["instance_num_point"].append(np.array(npoint, np.int32))
        out["instance_semantic_cls"].append(np.array(cls, np.int16))
    batch = {k: np.concatenate(v, 0) for k, v in out.items()}
    batch["instance_offsets"] = np.array(inst_offsets, np.int32)
    batch["scan_ids"] = [f"synthetic_{i:04d}" for i in range(len(scenes))]
    return batch


def to_torch(batch, device):
    import torch
    return {k: (torch.from_numpy(v).to(device) if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
