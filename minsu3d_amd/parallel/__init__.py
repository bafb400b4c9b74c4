"""Data parallelism over scenes (the only parallelism the reference has: Lightning DDP, one process per GPU,
config/model/base.yaml:13-16).  Scenes are independent units -- ball query, BFS, kernel maps and BatchNorm
statistics never cross a scene/rank -- so ranks own disjoint scenes and the only collective is the gradient
all-reduce (RCCL over xGMI through torch.distributed's "nccl" backend; gloo on CPU for tests)."""
import os

import torch
import torch.distributed as dist


def init_distributed():
    """(rank, local_rank, world_size) from the torchrun environment; initialises RCCL when world_size > 1"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # MS3D_DIST_BACKEND=gloo + MS3D_SHARE_DEVICE=1: several ranks on ONE GPU (how the DDP path is exercised on the
        # single-GPU test box; RCCL refuses two ranks on one device)
        backend = os.environ.get("MS3D_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("MS3D_SHARE_DEVICE") == "1":
            local = 0
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    if os.environ.get("MS3D_SHARE_DEVICE") == "1":
        local = 0
    return rank, local, world


def shard_scene_seeds(step, scenes_per_rank, rank, world_size):
    """scene ids of one step for one rank: global scene g = step*W*S + rank*S + i (disjoint across ranks)"""
    base = (step * world_size + rank) * scenes_per_rank
    return list(range(base, base + scenes_per_rank))


def wrap_ddp(model, device, find_unused_parameters=True):
    """DistributedDataParallel with the reference's setting (unused heads before `prepare_epochs` get no grad ->
    find_unused_parameters; callers that know every parameter is used -- grouping branch on -- pass False and skip
    the per-step graph traversal).  25 MB buckets overlap the all-reduce with the rest of backward; per-rank BatchNorm
    statistics are NOT synchronised, exactly like the reference (no SyncBN)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return model
    ids = None if device is None else [device.index]
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, find_unused_parameters=find_unused_parameters,
                                                     broadcast_buffers=False, bucket_cap_mb=25,
                                                     gradient_as_bucket_view=True)
