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

    # ---- the stages of a sample, in the reference's order of random draws (general_dataset.py:96-150)
    def _augment(self, xyz, normals, colors):
        m = self._get_augmentation_matrix()
        xyz = np.matmul(xyz, m)
        normals = np.matmul(normals, np.linalg.inv(m).T)
        if self.cfg.data.augmentation.jitter_rgb:
            colors = colors + np.random.randn(3) * 0.1
        return xyz, normals, colors.astype(np.float32)

    def _voxel_units(self, xyz, train):
        """metric -> voxel units, two elastic distortions when training (coarse 6/40, fine 20/160 at 2 cm), shifted
        into the positive octant"""
        per_metre = 1 / self.cfg.data.voxel_size
        v = xyz * per_metre
        if train and self.cfg.data.augmentation.elastic:
            for gran, mag in ((6 * per_metre // 50, 40 * per_metre / 50), (20 * per_metre // 50, 160 * per_metre / 50)):
                v = self._elastic(v, gran, mag)
        v = np.asarray(v)
        return v - v.min(axis=0)

    def _crop_window(self, v, sem_labels, instance_ids):
        """random window of at most max_num_point points that still contains labelled points (up to 20 tries)"""
        keep = np.ones(v.shape[0], dtype=bool)
        if keep.shape[0] <= self.max_num_point:
            return v, keep
        half, kept = self.max_num_point // 2, 0
        for _ in range(20):
            shifted, keep = T.crop(v, self.max_num_point, self.cfg.data.full_scale[1])
            kept = np.count_nonzero(keep)
            if kept >= half and np.any(sem_labels[keep] != -1) and np.any(instance_ids[keep] != -1):
                return shifted, keep
        if kept < half or np.all(sem_labels[keep] == -1) and np.all(instance_ids[keep] == -1):
            raise Exception("Over-cropped!")
        return v, keep

    def __getitem__(self, idx):
        scene, net = self.scenes[idx], self.cfg.model.network
        train = self.split == "train"
        xyz, colors, normals = (scene[k].astype(np.float32) for k in ("xyz", "rgb", "normal"))
        instance_ids, sem_labels = scene["instance_ids"].astype(np.int16), scene["sem_labels"].astype(np.int16)
        if train:
            xyz, normals, colors = self._augment(xyz, normals, colors)
        v = self._voxel_units(xyz, train)
        if train:
            v, keep = self._crop_window(v, sem_labels, instance_ids)
            v, xyz, normals, colors, sem_labels = v[keep], xyz[keep], normals[keep], colors[keep], sem_labels[keep]
            instance_ids = self._get_cropped_inst_ids(instance_ids, keep)
        num_instance, centers, npoint, inst_cls = self._get_inst_info(xyz, instance_ids, sem_labels)
        channels = ([colors] if net.use_color else []) + ([normals] if net.use_normal else []) + [xyz]
        return dict(scan_id=self.scene_names[idx], point_xyz=xyz, sem_labels=sem_labels, instance_ids=instance_ids,
                    num_instance=np.array(num_instance, dtype=np.int32), instance_center_xyz=centers,
                    instance_num_point=np.array(npoint, dtype=np.int32), instance_semantic_cls=inst_cls,
                    point_xyz_elastic=v / (1 / self.cfg.data.voxel_size),      # the reference's arithmetic (:143)
                    point_features=np.concatenate(channels, axis=1).astype(np.float32))


ScanNetv2 = GeneralDataset       # the reference selects the class by cfg.data.dataset (data/dataset/__init__.py)
