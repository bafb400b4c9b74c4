"""Golden vectors of the reference's dataset + collate -- runs ONLY in the build container
(`python tests/golden/make_golden_dataset.py`), writes DATA only (tests/golden/dataset_cases.npz).

What runs here is the reference's own Python: `GeneralDataset.__init__/__getitem__` (augmentation matrix, rgb jitter,
two elastic distortions, crop, dense instance ids, instance info, `ME.utils.sparse_quantize` --
minsu3d/data/dataset/general_dataset.py:10-165, util/transform.py) and `_sparse_collate_fn`
(data/data_module.py:42-98) on the seeded .pth scenes of tests/dataset_cases.py, with numpy's global RNG seeded.
`MinkowskiEngine` is this repository's module (CPU test double behind `sparse_quantize`); pytorch_lightning is a stub.
Reference quirks needed to import it: `scipy.ndimage.filters` (removed namespace), `torch.load` without weights_only.
"""
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"
import numpy as np
import scipy.interpolate  # noqa: F401  (the reference's transform.py uses scipy.interpolate without importing it)
import scipy.ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if not hasattr(scipy.ndimage, "filters"):
    scipy.ndimage.filters = types.SimpleNamespace(convolve=scipy.ndimage.convolve)

from make_golden_model import install_standins, reference_cfg      # noqa: E402


def main():
    install_standins()
    sys.modules["pytorch_lightning"].LightningDataModule = object
    from minsu3d.data.dataset.general_dataset import GeneralDataset
    from minsu3d.data.data_module import _sparse_collate_fn
    from dataset_cases import MAX_NUM_POINT, SEED, batch_arrays, write_scenes
    arrays = {}
    with tempfile.TemporaryDirectory() as root:
        write_scenes(root)
        cfg = reference_cfg("pointgroup")
        cfg["data"]["dataset_path"] = root
        cfg["data"]["metadata"]["train_list"] = os.path.join(root, "train.txt")
        cfg["data"]["metadata"]["val_list"] = os.path.join(root, "val.txt")
        cfg["data"]["max_num_point"] = MAX_NUM_POINT
        for split in ("train", "val"):
            ds = GeneralDataset(cfg, split)
            np.random.seed(SEED)
            items = [ds[i] for i in range(len(ds))]
            batch = _sparse_collate_fn(items)
            for k, v in batch_arrays(batch).items():
                arrays[f"{split}/{k}"] = v
            print(split, {k: tuple(v.shape) for k, v in batch_arrays(batch).items() if k != "scan_ids"})
    path = os.path.join(HERE, "dataset_cases.npz")
    np.savez_compressed(path, **arrays)
    print("wrote dataset_cases.npz (%d arrays, %.0f kB)" % (len(arrays), os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
