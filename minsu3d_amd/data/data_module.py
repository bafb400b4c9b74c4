"""Loaders + the collate step (reference `minsu3d/data/data_module.py:8-98`, without Lightning).

`sparse_collate_fn` builds the batch dictionary the models consume (same keys / dtypes as the reference's
`_sparse_collate_fn` :42-98) directly on the device: the per-scene voxelisation -- floor(xyz / voxel_size), first
occurrence per voxel, inverse map -- is `ME.utils.sparse_quantize` on the GPU, the batch column is added by
`ME.utils.sparse_collate`."""
from functools import partial

import numpy as np
import torch
from torch.utils.data import DataLoader

import minsu3d_amd.MinkowskiEngine as ME

from .dataset import GeneralDataset


def sparse_collate_fn(batch, device, voxel_size):
    dev = torch.device(device)
    data = {"scan_ids": [b["scan_id"] for b in batch]}
    pts, bids, sem, inst, centers, npoint, cls = [], [], [], [], [], [], []
    vox_xyz, vox_feat, vox_map = [], [], []
    inst_offsets, n_inst, n_vox = [0], 0, 0
    for i, b in enumerate(batch):
        n = b["point_xyz"].shape[0]
        pts.append(torch.from_numpy(np.ascontiguousarray(b["point_xyz"], dtype=np.float32)))
        bids.append(torch.full((n,), i, dtype=torch.uint8))
        ids = b["instance_ids"].copy()
        ids[ids != -1] += n_inst                                   # instance ids unique across the batch (:71-73)
        n_inst += int(b["num_instance"])
        inst.append(torch.from_numpy(ids))
        inst_offsets.append(n_inst)
        sem.append(torch.from_numpy(b["sem_labels"]))
        centers.append(torch.from_numpy(b["instance_center_xyz"]))
        npoint.append(torch.from_numpy(b["instance_num_point"]))
        cls.extend(b["instance_semantic_cls"])
        coords = torch.from_numpy(np.ascontiguousarray(b["point_xyz_elastic"])).to(dev)
        feats = torch.from_numpy(b["point_features"]).to(dev)
        vx, vf, _, inv = ME.utils.sparse_quantize(coords, feats, return_index=True, return_inverse=True,
                                                  quantization_size=voxel_size, device=dev.type)
        vox_xyz.append(vx)
        vox_feat.append(vf)
        vox_map.append(inv + n_vox)
        n_vox += vx.shape[0]
    data["point_xyz"] = torch.cat(pts).to(dev)
    data["vert_batch_ids"] = torch.cat(bids).to(dev)
    data["sem_labels"] = torch.cat(sem).to(dev)
    data["instance_ids"] = torch.cat(inst).to(dev)
    data["instance_center_xyz"] = torch.cat(centers).to(dev)
    data["instance_num_point"] = torch.cat(npoint).to(dev)
    data["instance_offsets"] = torch.tensor(inst_offsets, dtype=torch.int32, device=dev)
    data["instance_semantic_cls"] = torch.tensor(np.array(cls, dtype=np.int16), dtype=torch.int16, device=dev)
    data["voxel_xyz"], data["voxel_features"] = ME.utils.sparse_collate(coords=vox_xyz, feats=vox_feat)
    data["voxel_point_map"] = torch.cat(vox_map)
    return data


class DataModule:
    def __init__(self, data_cfg, device="cuda", elastic_fn=None):
        self.cfg = data_cfg
        self.device = device
        self.elastic_fn = elastic_fn

    def setup(self, stage=None):
        if stage in ("fit", None):
            self.train_set = GeneralDataset(self.cfg, "train", self.elastic_fn)
            self.val_set = GeneralDataset(self.cfg, "val", self.elastic_fn)
        if stage in ("test",):
            self.val_set = GeneralDataset(self.cfg, self.cfg.model.inference.split, self.elastic_fn)
        if stage in ("predict",):
            self.test_set = GeneralDataset(self.cfg, "test", self.elastic_fn)

    def _loader(self, dataset, batch_size, shuffle, epoch=0):
        # samples are host arrays; the device work (quantisation) happens in the collate call of the consuming process
        collate = partial(sparse_collate_fn, device=self.device, voxel_size=self.cfg.data.voxel_size)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # one process per GPU: every rank draws disjoint, size-matched scenes (the reference gets its sharding from
            # Lightning's automatic DistributedSampler, config/model/base.yaml:13-16)
            from ..parallel import BalancedDistributedBatchSampler
            sizes = [len(sc["xyz"]) for sc in dataset.scenes]
            sampler = BalancedDistributedBatchSampler(sizes, batch_size, shuffle=shuffle)
            sampler.set_epoch(epoch)
            return DataLoader(dataset, batch_sampler=sampler, num_workers=0, collate_fn=collate)
        return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=0, collate_fn=collate)

    def train_dataloader(self, epoch=0):
        return self._loader(self.train_set, self.cfg.data.batch_size, True, epoch)

    def val_dataloader(self):
        return self._loader(self.val_set, 1, False)

    test_dataloader = val_dataloader
