"""GPU: the data-parallel training path (reference config/model/base.yaml:13-16: Lightning DDP, one process per GPU) on
the one device of the test box -- two ranks over gloo sharing it (RCCL refuses two ranks on one device).

Round 4's single test compared the all-reduced gradients of a TRAINING-mode step with a second evaluation of the same
step whose BatchNorm statistics differed in the last bits (LDS float atomics), and failed on the driver's box at 2.7e-3
when a ReLU mask flipped between the two evaluations.  Round 5 made the step bit-reproducible
(tests/test_determinism_gpu.py), so the check is now exact arithmetic and split in two:

  * completeness -- what DistributedDataParallel's bucket hooks all-reduced is the average of the ranks' COMPLETE local
    gradients, per parameter tensor, for every combination of the machinery that sits between a backward-weight kernel
    and the hook: fused block nodes on / off, deferred + batched slab reductions on / off, backward-weight on a second
    stream joined per layer (MS3D_WGRAD_STREAM=1) or per layer group inside the deferred-reduction node (= 3).  A hook that fired before its layer group's flush would average unreduced slabs --
    identically on both ranks, so only the comparison with the plain module chain notices;
  * stay in step -- three training steps with the one-launch Adam: bit-identical, finite parameters on both ranks."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

# (fused block nodes, deferred slab reductions, backward-weight stream mode)
CONFIGS = [(True, True, 0), (False, True, 0), (True, False, 0), (False, False, 0), (True, True, 1), (True, True, 3),
           (False, True, 3)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MS3D_SHARE_DEVICE="1", MS3D_DIST_BACKEND="gloo")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    from minsu3d_amd import backend as ms_backend
    from minsu3d_amd.model.module import common
    from minsu3d_amd.parallel import init_distributed, shard_scene_seeds, wrap_ddp
    from test_model_cpu import build_model as bm, small_batch as sb
    init_distributed()
    dev = torch.device("cuda", 0)
    be = ms_backend.get_backend()
    u = (torch.tensor([0.3, 0.6, 0.9], device=dev), torch.tensor([0.1, 0.2, 0.3], device=dev))

    def set_config(fuse, defer, stream_mode):
        common._FUSE_BLOCKS = fuse
        be._wgrad_mode = stream_mode
        be._wgrad_defer = bool(defer and stream_mode in (0, 3))

    def local_grads(model, batch):
        model.zero_grad(set_to_none=True)
        sum(model._loss(batch, model(batch)).values()).backward()
        return [p.grad.detach().clone() for p in model.parameters()]

    seeds = shard_scene_seeds(step=0, scenes_per_rank=2, rank=rank, world_size=world)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sb(tuple(seeds)).items()}
    # the reference: the plain module chain, per-layer slab reductions, one stream, no wrapper; averaged by hand
    model = bm(seed=0).to(dev)
    model.train()
    model.voxelization_rand = u
    set_config(False, False, 0)
    want = []
    for g in local_grads(model, batch):
        parts = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(parts, g)
        want.append(torch.stack(parts).mean(0))
    names = [n for n, _ in model.named_parameters()]
    ddp = wrap_ddp(model, dev, find_unused_parameters=False)
    errs = {}
    for cfg in CONFIGS:
        set_config(*cfg)
        if cfg[1] and cfg[2] in (0, 3):
            q_ = be.wgrad_queue()
            assert q_ is not None and q_.side_mode == (cfg[2] == 3)   # the deferral is what runs below
        model.zero_grad(set_to_none=True)
        sum(model._loss(batch, ddp(batch)).values()).backward()
        torch.cuda.synchronize()
        worst = (0.0, "")
        for n, p, w in zip(names, model.parameters(), want):
            e = float((p.grad - w).abs().max() / (w.abs().max() + 1e-12))
            if e > worst[0]:
                worst = (e, n)
        errs[str(cfg)] = worst
    # stay in step: three optimizer steps on disjoint scenes with the default machinery
    set_config(True, True, 0)
    opt = model.configure_optimizers()
    losses = []
    for step in range(3):
        seeds = shard_scene_seeds(step=step, scenes_per_rank=2, rank=rank, world_size=world)
        b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sb(tuple(seeds)).items()}
        opt.zero_grad(set_to_none=True)
        loss = sum(model._loss(b, ddp(b)).values())
        loss.backward()
        opt.step()
        losses.append(float(loss))
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    q.put((rank, type(opt).__module__, losses, all(torch.equal(gathered[0], t) for t in gathered),
           bool(torch.isfinite(flat).all()), errs))
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "fused%d-defer%d-stream%d" % (int(c[0]), int(c[1]), c[2]))
def test_ddp_all_reduces_complete_gradients(two_ranks, cfg):
    """per parameter tensor: |all-reduced - average of the ranks' local gradients| <= 1e-5 of the tensor's largest entry
    (the step is bit-reproducible and the configurations run the same kernels in the same order, so the expected
    difference is the rounding of gloo's sum against torch's mean: measured 0 .. 1e-7)"""
    for r in two_ranks:
        err, name = r[5][str(cfg)]
        assert err <= 1e-5, (cfg, name, err)


def test_ddp_two_ranks_on_one_gpu_stay_in_step(two_ranks):
    res = two_ranks
    assert res[0][1] == res[1][1] == "minsu3d_amd.optim"      # the library's Adam is what stepped
    assert res[0][2] != res[1][2]                             # different scenes, different losses
    assert res[0][3] and res[1][3] and res[0][4]              # identical, finite parameters on both ranks


# ---- round 6: the N > 1 path of bench.py on the hardware there is -----------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_line(cmd, env):
    import json
    import subprocess
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_through_bench_have_no_stalled_steps():
    """the driver's torchrun line with two gloo ranks sharing the one GPU, at BENCH size.  Round 5 measured every second
    step at 2-20 s: the ranks' streams (main, second grouping, prefetch, gloo's copies, x 2 processes) oversubscribed the
    hardware queues and a rank waiting in the all-reduce held the other up.  Under data parallelism the coordinate
    prefetch now shares the second grouping stream (parallel.stream_plan); a step may not take 3x the median."""
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
           "--no-cpu-baseline", "--no-roofline", "--also", "none"]
    env = dict(os.environ, MS3D_SHARE_DEVICE="1", MS3D_DIST_BACKEND="gloo")
    rec = _bench_line(cmd, env)
    assert rec["n_gpus"] == 2 and rec["config"]["streams"]["prefetch_stream"] == "side"
    st = rec["step_ms"]
    assert st["max"] < 3.0 * st["median"], st


def test_one_rank_rccl_group_costs_little_step_time():
    """RCCL itself on the one GPU: MS3D_FORCE_PG=1 creates the `nccl` process group for world size 1 and wraps
    DistributedDataParallel, so the communicator, its stream beside the product's streams, the bucket hooks and the
    (one-rank) all-reduce all run -- the first time RCCL executed in this project (round 6).  Measured on quiet boxes
    (profiles/r06_rccl_one_rank.txt, 40-step runs): +3 ... +6 % on the median step (19.07 -> 19.64 ... 20.17 ms), i.e. the
    bucket hooks' per-parameter copies and the reducer's host work, not a stall; VERDICT r5 asked for 5 %.  The bound here is
    15 % on the best of two interleaved 20-step runs (a shared host moves single runs by several per cent): what separates that overhead from the round-5 failure mode (steps of
    SECONDS when the streams oversubscribed the hardware queues)."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "6", "--no-cpu-baseline",
            "--no-roofline", "--also", "none"]
    plain, forced = [], []
    for _ in range(2):
        plain.append(_bench_line(base, dict(os.environ))["step_ms"]["median"])
        rec = _bench_line(base, dict(os.environ, MS3D_FORCE_PG="1", MASTER_PORT=str(29800 + os.getpid() % 90)))
        assert rec["config"]["streams"]["collective_streams"] != 0 and rec["n_gpus"] == 1
        forced.append(rec["step_ms"]["median"])
        assert rec["step_ms"]["max"] < 3.0 * rec["step_ms"]["median"], rec["step_ms"]
    assert min(forced) <= 1.15 * min(plain), (plain, forced)
