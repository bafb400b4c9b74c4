"""OUR GeneralDataset + sparse_collate_fn against tests/golden/dataset_cases.npz -- the batches the REFERENCE's
`GeneralDataset.__getitem__` (general_dataset.py:80-165) and `_sparse_collate_fn` (data_module.py:42-98) produced from
the same .pth scenes under the same numpy seed (tests/golden/make_golden_dataset.py).  Ours voxelises in the collate
step on the device (the reference: on the host inside __getitem__), so the comparison is on the collated batch.
Integers (labels, instance ids incl. the re-densing after a crop, voxel coordinates, point -> voxel map) exact;
floats to 1e-6 (the elastic distortion is a restatement of scipy's filters: 1e-9 voxels)."""
import os

import numpy as np

from dataset_cases import BATCH_KEYS, MAX_NUM_POINT, SEED, batch_arrays, write_scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_cases.npz")


def check(root, device, elastic_fn=None):
    from minsu3d_amd.config import load_config
    from minsu3d_amd.data.data_module import sparse_collate_fn
    from minsu3d_amd.data.dataset import GeneralDataset
    write_scenes(str(root))
    cfg = load_config([f"data.dataset_path={root}", f"data.metadata.train_list={root}/train.txt",
                       f"data.metadata.val_list={root}/val.txt", f"data.max_num_point={MAX_NUM_POINT}"])
    want_all = np.load(GOLDEN)
    n_checked = 0
    for split in ("train", "val"):
        ds = GeneralDataset(cfg, split, elastic_fn)
        np.random.seed(SEED)
        items = [ds[i] for i in range(len(ds))]
        got = batch_arrays(sparse_collate_fn(items, device=device, voxel_size=cfg.data.voxel_size))
        want = {k.split("/", 1)[1]: v for k, v in want_all.items() if k.startswith(split + "/")}
        assert list(got["scan_ids"]) == list(want["scan_ids"])
        for k in BATCH_KEYS:
            g, w = got[k], want[k]
            assert g.shape == w.shape and g.dtype == w.dtype, (split, k, g.shape, w.shape, g.dtype, w.dtype)
            if k == "instance_center_xyz":       # rows of unlabelled points are uninitialised memory on both sides (:62)
                m = want["instance_ids"] != -1
                g, w = g[m], w[m]
            if np.issubdtype(w.dtype, np.floating):
                assert np.allclose(g, w, rtol=0, atol=1e-6), (split, k, float(np.abs(g - w).max()))
            else:
                assert np.array_equal(g, w), (split, k, int((g != w).sum()))
            n_checked += 1
        if split == "train":                     # the case exercises what it claims to
            assert got["point_xyz"].shape[0] < 3 * MAX_NUM_POINT and got["instance_num_point"].shape[0] < 15
    return n_checked
