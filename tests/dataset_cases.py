"""Seeded on-disk scenes of the dataset / collate pin: shared by tests/golden/make_golden_dataset.py (which reads them with
the REFERENCE's GeneralDataset + _sparse_collate_fn in the build container) and tests/test_dataset_pins_{cpu,gpu}.py
(which read them with ours).  Format: the reference's preprocessed .pth dictionaries
(data/scannetv2/preprocess_all_data.py:120-121)."""
import os

import numpy as np
import torch

MAX_NUM_POINT = 3000          # below the scene sizes: the crop (general_dataset.py:111-135) fires for every train sample
SEED = 17                     # numpy seed set right before the samples of a split are drawn


def write_scenes(root):
    """-> root with train/ (3 scenes) and val/ (1 scene) + the split lists"""
    from minsu3d_amd.data import synthetic
    for split, seeds in (("train", (20, 21, 22)), ("val", (23,))):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        names = []
        for s in seeds:
            sc = synthetic.make_scene(s, room=(2.0, 1.6), n_boxes=5, density=420.0, wall_h=0.6)
            rng = np.random.default_rng(100 + s)
            inst = sc["instance_ids"].copy()
            inst[(inst == 1) & (rng.random(len(inst)) < 0.5)] = -1        # unlabelled points inside an object
            sem = sc["sem_labels"].copy()
            sem[rng.random(len(sem)) < 0.02] = -1                          # ignored points
            name = f"scene{s:04d}_00"
            torch.save({"xyz": sc["xyz"] + np.float32(1.5), "rgb": ((sc["rgb"] + 1) * 127.5).astype(np.uint8),
                        "normal": rng.standard_normal(sc["xyz"].shape).astype(np.float32),
                        "sem_labels": sem, "instance_ids": inst}, os.path.join(root, split, f"{name}.pth"))
            names.append(name)
        with open(os.path.join(root, f"{split}.txt"), "w") as f:
            f.write("\n".join(names) + "\n")
    return root


BATCH_KEYS = ("point_xyz", "vert_batch_ids", "sem_labels", "instance_ids", "instance_center_xyz", "instance_num_point",
              "instance_offsets", "instance_semantic_cls", "voxel_xyz", "voxel_features", "voxel_point_map")


def batch_arrays(batch):
    out = {k: batch[k].detach().cpu().numpy() for k in BATCH_KEYS}
    out["scan_ids"] = np.array(batch["scan_ids"])
    return out
