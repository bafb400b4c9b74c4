"""GPU side of the data path: elastic distortion kernel vs the host restatement (itself pinned to the reference's
elastic(); agreement to an ulp of the float64 voxel coordinate, tolerance 1e-9 voxels written below), and the on-device collate vs the numpy restatement of the reference's quantise + collate."""
import numpy as np
import pytest
import torch

from minsu3d_amd.data import synthetic
from minsu3d_amd.data.data_module import sparse_collate_fn
from minsu3d_amd.util import transform as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from minsu3d_amd import backend as B
    return B.get_backend()


def test_elastic_kernel_vs_host(be):
    rng = np.random.default_rng(3)
    pts = ((rng.random((50000, 3)) * [5.0, 4.0, 2.5] - [2.5, 2.0, 1.25]) * 50).astype(np.float32)
    for gran, mag in ((6.0, 40.0), (20.0, 160.0)):
        np.random.seed(17)
        noise = np.stack(T.elastic_noise(pts, gran))
        want = pts + np.hstack([T.trilinear(T.blur_noise(n), gran, pts)[:, None] for n in noise]) * mag
        got = be.elastic(torch.from_numpy(pts).cuda(), torch.from_numpy(noise).cuda(), gran, mag)
        assert got.dtype == torch.float64 and np.abs(got.cpu().numpy() - want).max() < 1e-9   # voxel units
    # points outside the noise grid are left where they are
    far = np.array([[1e5, 0, 0], [0, -1e5, 0]], np.float32)
    np.random.seed(1)
    noise = np.stack(T.elastic_noise(pts, 6.0))
    out = be.elastic(torch.from_numpy(far).cuda(), torch.from_numpy(noise).cuda(), 6.0, 40.0).cpu().numpy()
    assert np.array_equal(out[0], far[0].astype(np.float64)) and np.array_equal(out[1], far[1].astype(np.float64))


def test_collate_on_device(be):
    scenes = [synthetic.make_scene(s, room=(2.0, 1.6), n_boxes=3, density=900.0) for s in (0, 1, 2)]
    samples = []
    for i, sc in enumerate(scenes):
        xyz = sc["xyz"]
        ids = sc["instance_ids"]
        uniq = np.unique(ids[ids >= 0])
        centers = np.zeros((len(xyz), 3), np.float32)
        for u in uniq:
            centers[ids == u] = xyz[ids == u].mean(0)
        samples.append({"scan_id": f"s{i}", "point_xyz": xyz, "sem_labels": sc["sem_labels"], "instance_ids": ids.copy(),
                        "num_instance": np.array(len(uniq), np.int32), "instance_center_xyz": centers,
                        "instance_num_point": np.array([np.count_nonzero(ids == u) for u in uniq], np.int32),
                        "instance_semantic_cls": np.array([sc["sem_labels"][ids == u][0] - 2 for u in uniq], np.int16),
                        "point_xyz_elastic": (xyz - xyz.min(0)).astype(np.float64),
                        "point_features": np.concatenate([sc["rgb"], xyz], 1).astype(np.float32)})
    got = sparse_collate_fn(samples, "cuda", 0.02)
    want = synthetic.collate(scenes, 0.02)
    for k in ("point_xyz", "vert_batch_ids", "sem_labels", "instance_ids", "instance_num_point", "instance_semantic_cls",
              "voxel_xyz", "voxel_features", "voxel_point_map", "instance_offsets"):
        assert got[k].is_cuda and np.array_equal(got[k].cpu().numpy(), want[k]), k
