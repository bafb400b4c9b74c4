"""Scene dataset in the reference's on-disk format and with its augmentation pipeline
(`minsu3d/data/dataset/general_dataset.py:10-165`).

Files: <dataset_path>/<split>/<scene>.pth = torch-saved dict {xyz f32[N,3], rgb u8[N,3], normal f32[N,3],
sem_labels i16[N], instance_ids i16[N]} (written by data/scannetv2/preprocess_all_data.py:120-121); scene names
from cfg.data.metadata.<split>_list.

Differences by design: a sample carries the (elastically distorted) metric coordinates `point_xyz_elastic` and the
per-point features instead of voxels -- quantisation and batching happen ON THE GPU in data_module.sparse_collate_fn
(the reference quantises on the host in every DataLoader worker, general_dataset.py:159-163); the elastic distortion
runs on the GPU too when a HIP backend is installed (host-drawn noise, so a numpy seed gives the reference's
augmentation), otherwise through the host restatement in util/transform.py.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from ..util import transform as T


class GeneralDataset(Dataset):
    def __init__(self, cfg, split, elastic_fn=None):
        self.cfg = cfg
        self.split = split
        self.max_num_point = cfg.data.max_num_point
        self.elastic_fn = elastic_fn        # (xyz float [N,3] voxel units, gran, mag) -> float64 [N,3]; None = host
        self._load_from_disk()

    def _load_from_disk(self):
        with open(getattr(self.cfg.data.metadata, f"{self.split}_list")) as f:
            self.scene_names = [line.strip() for line in f if line.strip()]
        self.scenes = []
        for name in self.scene_names:
            scene = torch.load(os.path.join(self.cfg.data.dataset_path, self.split, f"{name}.pth"), weights_only=False)
            scene["xyz"] = scene["xyz"] - scene["xyz"].mean(axis=0)                 # :24
            scene["rgb"] = scene["rgb"].astype(np.float32) / 127.5 - 1                # :25
            self.scenes.append(scene)

    def __len__(self):
        return len(self.scenes)

    def _get_augmentation_matrix(self):
        """jitter -> random x flip -> rotation about z, in the reference's order of random draws (:31-42)"""
        aug = self.cfg.data.augmentation
        m = np.eye(3)
        if aug.jitter_xyz:
            m = np.matmul(m, T.jitter())
        if aug.flip:
            m *= T.flip(0, random=True)      # element-wise, as the reference writes it
        if aug.rotation:
            m = np.matmul(m, T.rotz(np.random.rand() * 2 * np.pi))
        return m.astype(np.float32)

    @staticmethod
    def _get_cropped_inst_ids(instance_ids, valid_idxs):
        """keep ids dense after a crop: an id that lost all its points is taken over by the current largest id (:44-55)"""
        ids = instance_ids[valid_idxs]
        j = 0
        while j < ids.max():
            if not np.any(ids == j):
                ids[ids == ids.max()] = j
            j += 1
        return ids

    def _get_inst_info(self, xyz, instance_ids, sem_labels):
        """per-instance point count / class and per-point instance centre (:57-78), one pass with bincount"""
        ids = np.unique(instance_ids)
        ids = ids[ids != -1]
        centers = np.empty((xyz.shape[0], 3), np.float32)           # rows of unlabelled points stay uninitialised (:62)
        inst_cls = np.full(ids.shape[0], -1, np.int16)
        npoint = []
        for k, i in enumerate(ids):
            rows = np.where(instance_ids == i)[0]
            centers[rows] = xyz[rows].mean(0)
            npoint.append(rows.size)
            c = sem_labels[rows[0]]
            inst_cls[k] = c - len(self.cfg.data.ignore_classes) if c != -1 else c
        return ids.shape[0], centers, npoint, inst_cls

    def _elastic(self, x, gran, mag):
        if self.elastic_fn is None:
            return T.elastic(x, gran, mag)
        noise = np.stack(T.elastic_noise(x, gran))          # same random draws as the host path
        return self.elastic_fn(x, noise, gran, mag)

    def __getitem__(self, idx):
        scene = self.scenes[idx]
        cfg = self.cfg
        train = self.split == "train"
        xyz = scene["xyz"].astype(np.float32)
        colors = scene["rgb"].astype(np.float32)
        normals = scene["normal"].astype(np.float32)
        instance_ids = scene["instance_ids"].astype(np.int16)
        sem_labels = scene["sem_labels"].astype(np.int16)
        if train:
            m = self._get_augmentation_matrix()
            xyz = np.matmul(xyz, m)
            normals = np.matmul(normals, np.transpose(np.linalg.inv(m)))
            if cfg.data.augmentation.jitter_rgb:
                colors += np.random.randn(3) * 0.1
        scale = 1 / cfg.data.voxel_size
        if train and cfg.data.augmentation.elastic:
            e = self._elastic(xyz * scale, 6 * scale // 50, 40 * scale / 50)
            e = self._elastic(e, 20 * scale // 50, 160 * scale / 50)
        else:
            e = xyz * scale
        e = np.asarray(e)
        e = e - e.min(axis=0)
        if train:
            valid = np.ones(xyz.shape[0], dtype=bool)
            if valid.shape[0] > self.max_num_point:                              # :127-140
                count = 0
                for _ in range(20):
                    tmp, valid = T.crop(e, self.max_num_point, cfg.data.full_scale[1])
                    count = np.count_nonzero(valid)
                    if count >= self.max_num_point // 2 and np.any(sem_labels[valid] != -1) \
                            and np.any(instance_ids[valid] != -1):
                        e = tmp
                        break
                if count < self.max_num_point // 2 or np.all(sem_labels[valid] == -1) \
                        and np.all(instance_ids[valid] == -1):
                    raise Exception("Over-cropped!")
            e, xyz, normals, colors, sem_labels = e[valid], xyz[valid], normals[valid], colors[valid], sem_labels[valid]
            instance_ids = self._get_cropped_inst_ids(instance_ids, valid)
        e = e / scale
        num_instance, centers, npoint, inst_cls = self._get_inst_info(xyz, instance_ids, sem_labels)
        feats = [colors] if cfg.model.network.use_color else []
        if cfg.model.network.use_normal:
            feats.append(normals)
        feats.append(xyz)
        return {"scan_id": self.scene_names[idx], "point_xyz": xyz, "sem_labels": sem_labels, "instance_ids": instance_ids,
                "num_instance": np.array(num_instance, dtype=np.int32), "instance_center_xyz": centers,
                "instance_num_point": np.array(npoint, dtype=np.int32), "instance_semantic_cls": inst_cls,
                "point_xyz_elastic": e, "point_features": np.concatenate(feats, axis=1).astype(np.float32)}


ScanNetv2 = GeneralDataset       # the reference selects the class by cfg.data.dataset (data/dataset/__init__.py)
