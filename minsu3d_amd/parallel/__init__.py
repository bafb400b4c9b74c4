"""Data parallelism over scenes (the only parallelism the reference has: Lightning DDP, one process per GPU,
config/model/base.yaml:13-16).  Scenes are independent units -- ball query, BFS, kernel maps and BatchNorm
statistics never cross a scene/rank -- so ranks own disjoint scenes and the only collective is the gradient
all-reduce (RCCL over xGMI through torch.distributed's "nccl" backend; gloo on CPU for tests)."""
import os

import torch
import torch.distributed as dist


def force_process_group():
    """MS3D_FORCE_PG=1: create the process group and wrap DistributedDataParallel even for ONE rank -- how the RCCL path
    (communicator, its stream beside the product's, bucket hooks, one-rank all-reduce) is exercised on a single-GPU box"""
    return os.environ.get("MS3D_FORCE_PG", "0") == "1"


def stream_plan(world=None):
    """Which streams a rank keeps busy.  A process gets 4 hardware queues by default (GPU_MAX_HW_QUEUES); streams beyond
    that share queues, and a stream that WAITS for another rank (the collective's) on a queue it shares with a stream
    the other rank is waiting for stalls both until the scheduler rotates (round 5: two ranks on one GPU, every second
    step 2-20 s).  So under data parallelism the next batch's coordinate prefetch runs on the second grouping stream
    instead of a stream of its own: main + grouping side + the collective's = 3 (MS3D_PREFETCH_STREAM=own|side overrides).
    -> dict for the bench line"""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else int(os.environ.get("WORLD_SIZE", "1"))
    multi = world > 1 or force_process_group()
    where = os.environ.get("MS3D_PREFETCH_STREAM", "side" if multi else "own")
    on = os.environ.get("MS3D_PREFETCH_COORDS", "1") != "0"
    extra = int(os.environ.get("MS3D_WGRAD_STREAM", "0") != "0") + int(os.environ.get("MS3D_EARLY_HEADS", "0") == "2")
    return {"prefetch_stream": where if on else "off",
            "compute_streams": 2 + int(on and where == "own") + extra,
            "collective_streams": "RCCL's own (1+)" if multi else 0,
            "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)")}


def init_distributed():
    """(rank, local_rank, world_size) from the torchrun environment; initialises RCCL when world_size > 1"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_process_group()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
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
    if world > 1:
        pin_rank_threads(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    return rank, local, world


def pin_rank_threads(local_rank, local_world):
    """Host side of one-process-per-GPU: every rank runs a main thread, the grouping helper thread and the loader's
    collate, and torch's intra-op pool defaults to ALL cores of the node in every process.  Give each rank its own
    contiguous slice of the cores this process may use (threads started later inherit the mask) and size the intra-op
    pool to it, so that 8 ranks do not run 8 x nproc OpenMP threads against each other.  MS3D_PIN=0 leaves both alone.
    -> the list of cores of this rank (None when nothing was changed)"""
    if os.environ.get("MS3D_PIN", "1") == "0" or local_world <= 1 or not hasattr(os, "sched_getaffinity"):
        return None
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // local_world
    if per < 1:
        return None
    mine = cores[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    torch.set_num_threads(max(1, min(per, 16)))      # the host work of a step is a few small torch CPU ops
    return mine


def shard_scene_seeds(step, scenes_per_rank, rank, world_size):
    """scene ids of one step for one rank: global scene g = step*W*S + rank*S + i (disjoint across ranks)"""
    base = (step * world_size + rank) * scenes_per_rank
    return list(range(base, base + scenes_per_rank))


def wrap_ddp(model, device, find_unused_parameters=True):
    """DistributedDataParallel (unused heads before `prepare_epochs` get no grad -> find_unused_parameters; callers
    that know every parameter is used -- grouping branch on -- pass False and skip the per-step graph traversal).
    25 MB buckets overlap the all-reduce with the rest of backward.  BatchNorm statistics are per rank, like the
    reference (no SyncBN).  One deliberate difference: the reference runs Lightning DDP with torch's default
    broadcast_buffers=True, i.e. rank 0's BatchNorm running statistics are broadcast at EVERY forward; here the
    buffers are left alone during training (80 BatchNorm layers x 3 buffers per step for values only evaluation reads)
    and `sync_buffers()` broadcasts rank 0's before validation and before a checkpoint is written, which is where the
    reference's ranks > 0 would see them."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_process_group()):
        return model
    ids = None if device is None or device.type != "cuda" else [device.index]
    extra = {}
    if os.environ.get("MS3D_DDP_STATIC", "0") == "1" and not find_unused_parameters:
        extra["static_graph"] = True      # the autograd graph of a step does not change: skips the per-step graph bookkeeping
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, find_unused_parameters=find_unused_parameters,
                                                    broadcast_buffers=False, bucket_cap_mb=int(os.environ.get("MS3D_DDP_BUCKET_MB", "25")),
                                                    gradient_as_bucket_view=True, **extra)
    if os.environ.get("MS3D_DDP_HOOK", "0") == "1":
        # torch's all-reduce hook divides the FLAT bucket once; without a hook the reducer divides every parameter's
        # bucket view on its own (~250 small launches per step)
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        ddp.register_comm_hook(None, default_hooks.allreduce_hook)
    return ddp


def sync_buffers(model, src=0):
    """rank `src`'s buffers (BatchNorm running_mean / running_var / num_batches_tracked) to every rank"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for m in model.modules():
        flush = getattr(m, "flush_batches_tracked", None)      # host-side counter of MinkowskiBatchNorm
        if flush is not None:
            flush()
    for b in model.buffers():
        dist.broadcast(b, src)


class BalancedDistributedBatchSampler(torch.utils.data.Sampler):
    """Rank-sharded batches (what Lightning's automatic DistributedSampler gives the reference,
    data_module.py:23-39) with the scenes of one step SIZE-MATCHED across ranks: a step lasts as long as its slowest
    rank, and scene sizes (hence ball-query / BFS / convolution work) spread over 2-3x in ScanNet.  Per epoch: one
    seeded permutation (identical on every rank) is cut into windows of world*batch*window_steps scenes; inside a window
    the scenes are sorted by size and cut into steps of world*batch consecutive scenes, so the ranks of a step hold
    neighbours in the size order; inside a step the scenes are dealt to the ranks in SERPENTINE order (0..W-1, W-1..0,
    ...) starting at a rank that rotates with the step, so that no rank systematically receives the larger scenes of
    every step; the steps of a window are then shuffled again.  Like DistributedSampler the tail is padded by
    wrap-around so that every rank runs the same number of steps; `padded_positions()` tells which of THIS rank's
    samples are such repeats (validation drops them before evaluating)."""

    def __init__(self, sizes, batch_size, rank=None, world_size=None, shuffle=True, seed=0, window_steps=8):
        self.sizes = [int(s) for s in sizes]
        self.batch_size = int(batch_size)
        self.rank = dist.get_rank() if rank is None and dist.is_initialized() else int(rank or 0)
        self.world = dist.get_world_size() if world_size is None and dist.is_initialized() else int(world_size or 1)
        self.shuffle, self.seed, self.window_steps, self.epoch = shuffle, seed, window_steps, 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        per_step = self.batch_size * self.world
        return (len(self.sizes) + per_step - 1) // per_step

    def _deal(self, step_scenes, step_index):
        """this rank's share of one step's scenes (ascending size): serpentine over the ranks, rotated per step"""
        W = self.world
        mine = []
        for j, scene in enumerate(step_scenes):
            lap, pos = divmod(j, W)
            r = pos if lap % 2 == 0 else W - 1 - pos
            if (r + step_index) % W == self.rank:
                mine.append(scene)
        return mine

    def _steps(self):
        n = len(self.sizes)
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        order = torch.randperm(n, generator=g).tolist() if self.shuffle else list(range(n))
        per_step = self.batch_size * self.world
        total = len(self) * per_step
        order = (order * (total // max(n, 1) + 1))[:total]               # wrap-around padding
        order = [(i, k >= n) for k, i in enumerate(order)]                 # (scene, is a padding repeat)
        win = per_step * self.window_steps
        steps = []
        for w0 in range(0, total, win):
            chunk = sorted(order[w0:w0 + win], key=lambda e: (self.sizes[e[0]], e[0], e[1]))
            wsteps = [chunk[s:s + per_step] for s in range(0, len(chunk), per_step)]
            if self.shuffle:
                wsteps = [wsteps[i] for i in torch.randperm(len(wsteps), generator=g).tolist()]
            steps.extend(wsteps)
        return [self._deal(st, k) for k, st in enumerate(steps)]

    def __iter__(self):
        for st in self._steps():
            yield [scene for scene, _ in st]

    def padded_positions(self):
        """flags, in iteration order over this rank's samples, of the wrap-around repeats"""
        return [pad for st in self._steps() for _, pad in st]
