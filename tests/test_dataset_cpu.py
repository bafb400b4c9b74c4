"""Dataset + collate on scenes written in the reference's .pth format; the two operators behind
the collate step (sparse_quantize, elastic) are served by the oracle backend here and by the HIP library on the GPU."""
import os

import numpy as np
import pytest
import torch

from minsu3d_amd import backend as ms_backend
from minsu3d_amd.config import load_config
from minsu3d_amd.data import synthetic
from minsu3d_amd.data.data_module import DataModule, sparse_collate_fn
from minsu3d_amd.data.dataset import GeneralDataset
from oracle.oracle_backend import OracleBackend


@pytest.fixture(autouse=True)
def oracle_backend():
    prev = ms_backend.set_backend(OracleBackend())
    yield
    ms_backend.set_backend(prev)


@pytest.fixture(scope="module")
def dataset_dir(tmp_path_factory):
    root = tmp_path_factory.mktemp("scannetv2")
    for split, seeds in (("train", (0, 1, 2)), ("val", (3,))):
        os.makedirs(root / split)
        names = []
        for s in seeds:
            sc = synthetic.make_scene(s, room=(2.0, 1.6), n_boxes=3, density=900.0)
            rng = np.random.default_rng(100 + s)
            name = f"scene{s:04d}_00"
            torch.save({"xyz": sc["xyz"] + 1.5, "rgb": ((sc["rgb"] + 1) * 127.5).astype(np.uint8),
                        "normal": rng.standard_normal(sc["xyz"].shape).astype(np.float32),
                        "sem_labels": sc["sem_labels"], "instance_ids": sc["instance_ids"]}, root / split / f"{name}.pth")
            names.append(name)
        (root / f"{split}.txt").write_text("\n".join(names) + "\n")
    return root


def make_cfg(root, **extra):
    ov = [f"data.dataset_path={root}", f"data.metadata.train_list={root}/train.txt", f"data.metadata.val_list={root}/val.txt"]
    ov += [f"{k}={v}" for k, v in extra.items()]
    return load_config(ov)


def test_val_sample_is_the_stored_scene(dataset_dir):
    cfg = make_cfg(dataset_dir)
    ds = GeneralDataset(cfg, "val")
    assert len(ds) == 1
    d = ds[0]
    raw = torch.load(dataset_dir / "val" / "scene0003_00.pth", weights_only=False)
    xyz = (raw["xyz"] - raw["xyz"].mean(0)).astype(np.float32)
    assert np.array_equal(d["point_xyz"], xyz)                              # centred, not augmented
    assert np.allclose(d["point_xyz_elastic"], xyz - xyz.min(0), atol=1e-6)  # shifted to the positive octant (metres)
    assert d["point_features"].shape == (len(xyz), 6)                        # rgb in [-1, 1] + xyz
    assert np.array_equal(d["point_features"][:, 3:], xyz) and np.abs(d["point_features"][:, :3]).max() <= 1.0
    ids = np.unique(raw["instance_ids"][raw["instance_ids"] >= 0])
    assert int(d["num_instance"]) == len(ids) and d["instance_num_point"].sum() == np.count_nonzero(raw["instance_ids"] >= 0)
    m = raw["instance_ids"] == ids[0]
    assert np.allclose(d["instance_center_xyz"][m], xyz[m].mean(0), atol=1e-6)
    assert d["instance_semantic_cls"][0] == raw["sem_labels"][m][0] - 2


def test_train_sample_is_seed_deterministic_and_cropped(dataset_dir):
    cfg = make_cfg(dataset_dir, **{"data.max_num_point": 5000})
    ds = GeneralDataset(cfg, "train")
    np.random.seed(5)
    a = ds[1]
    np.random.seed(5)
    b = ds[1]
    for k in ("point_xyz", "point_xyz_elastic", "instance_ids", "sem_labels", "point_features"):
        assert np.array_equal(a[k], b[k])
    n = a["point_xyz"].shape[0]
    assert 2500 <= n <= 5000                                   # cropped to at most max_num_point, at least half of it
    ids = a["instance_ids"]
    assert set(np.unique(ids[ids >= 0])) == set(range(int(a["num_instance"])))      # ids stay dense after the crop
    assert a["point_xyz_elastic"].min() >= 0 and a["point_xyz_elastic"].dtype == np.float64
    # device-style elastic (host-drawn noise handed to the backend) gives the same sample
    be = ms_backend.get_backend()
    ds2 = GeneralDataset(cfg, "train", elastic_fn=lambda x, noise, g, m: be.elastic(torch.from_numpy(np.asarray(x)),
                                                                                    torch.from_numpy(noise), g, m).numpy())
    np.random.seed(5)
    c = ds2[1]
    assert np.array_equal(a["point_xyz_elastic"], c["point_xyz_elastic"]) and np.array_equal(a["instance_ids"], c["instance_ids"])


def test_collate_matches_the_synthetic_reference_batch(dataset_dir):
    """the collate step on two val-style samples against minsu3d_amd.data.synthetic.collate (the numpy restatement of
    general_dataset.py:159-163 + data_module.py:42-98 used by the benchmark)"""
    cfg = make_cfg(dataset_dir)
    ds = GeneralDataset(cfg, "train")
    ds.split = "val"                                            # no augmentation: directly comparable
    samples = [ds[0], ds[2]]
    got = sparse_collate_fn(samples, "cpu", cfg.data.voxel_size)
    scenes = []
    for s in samples:
        scenes.append(dict(xyz=s["point_xyz"], rgb=s["point_features"][:, :3], sem_labels=s["sem_labels"],
                           instance_ids=s["instance_ids"]))
    want = synthetic.collate(scenes, cfg.data.voxel_size)
    for k in ("point_xyz", "vert_batch_ids", "sem_labels", "instance_ids", "instance_num_point", "instance_semantic_cls",
              "voxel_xyz", "voxel_features", "voxel_point_map", "instance_offsets"):
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    labelled = want["instance_ids"] >= 0
    assert np.allclose(got["instance_center_xyz"].numpy()[labelled], want["instance_center_xyz"][labelled], atol=1e-5)
    assert got["scan_ids"] == ["scene0000_00", "scene0002_00"]


def test_data_module_loaders(dataset_dir):
    cfg = make_cfg(dataset_dir, **{"data.batch_size": 2})
    dm = DataModule(cfg, device="cpu")
    dm.setup("fit")
    np.random.seed(0); torch.manual_seed(0)
    batch = next(iter(dm.train_dataloader()))
    assert batch["instance_offsets"].numel() == 3 and batch["voxel_xyz"].shape[1] == 4
    assert int(batch["voxel_point_map"].max()) == batch["voxel_xyz"].shape[0] - 1
    vb = next(iter(dm.val_dataloader()))
    assert vb["vert_batch_ids"].max() == 0
