#!/usr/bin/env python3
"""Headline benchmark: scenes/sec (fwd + loss + bwd + Adam) of PointGroup (m=16) on synthetic ScanNet-shaped
~150k-point scenes voxelised at 2 cm, batch 4 scenes per GPU, grouping branch active
(current_epoch > prepare_epochs), data-parallel over N GPUs of one node (one process per GPU, RCCL).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around the dominant kernel (the
3x3x3 16->16 sparse-conv gather kernel at full resolution, forward and backward-data launches); `cpu_baseline`
times the same training step on the host cores with every operator served by the CPU oracle (a PORT of the
reference algorithms -- the reference's own CPU path needs MinkowskiEngine, which is not available), on a
bounded sample, N=1 / rank 0 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from minsu3d_amd import backend as ms_backend  # noqa: E402
from minsu3d_amd.config import load_config  # noqa: E402
from minsu3d_amd.data import synthetic  # noqa: E402
import minsu3d_amd.model as ms_models  # noqa: E402
import minsu3d_amd.MinkowskiEngine as ME  # noqa: E402
from minsu3d_amd.parallel import init_distributed, shard_scene_seeds, wrap_ddp  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def make_batch(seeds, device, scene_kwargs=None, offset_noise=0.04):
    """scenes -> device batch + the grouping inputs a trained network would produce (GT labels; offsets to the
    instance centre + N(0, 4 cm) so that shifted-coordinate ball queries average a few hundred neighbours, the
    regime the reference's cluster_shift_meanActive=300 is sized for)"""
    scenes = [synthetic.make_scene(s, **(scene_kwargs or {})) for s in seeds]
    b = synthetic.to_torch(synthetic.collate(scenes), device)
    rng = np.random.default_rng(10_000 + seeds[0])
    noise = torch.from_numpy(rng.normal(0, offset_noise, tuple(b["point_xyz"].shape)).astype(np.float32)).to(device)
    b["grouping_semantic_preds"] = torch.where(b["sem_labels"] >= 0, b["sem_labels"], torch.zeros_like(b["sem_labels"]))
    b["grouping_point_offsets"] = torch.where((b["instance_ids"] >= 0)[:, None],
                                              b["instance_center_xyz"] - b["point_xyz"] + noise,
                                              torch.zeros_like(noise))
    n = b["point_xyz"].shape[0]
    sem = torch.full((n, 20), 0.01, device=device)                      # SoftGroup: per-class soft scores
    sem[torch.arange(n, device=device), b["grouping_semantic_preds"].long()] = 0.8
    b["grouping_semantic_scores"] = sem
    return b


def build(cfg, device, seed=0):
    torch.manual_seed(seed)
    model = getattr(ms_models, cfg.model.network.module)(cfg).to(device)
    model.current_epoch = cfg.model.network.prepare_epochs + 1   # grouping + ScoreNet branch on
    model.train()
    return model


def train_step(model, ddp, opt, batch, next_batch=None):
    opt.zero_grad(set_to_none=True)
    out = ddp(batch)
    loss = sum(model._loss(batch, out).values())
    if next_batch is not None:
        # input pipelining, as a data loader would do it: the coordinate-only work of the NEXT batch (row order, kernel
        # maps, pair lists) is built on a side stream while this step's backward pass runs
        ME.prefetch_coordinates(next_batch["voxel_xyz"], model.backbone.n_levels, wait_current_stream=False)
    loss.backward()
    opt.step()
    return loss


def cpu_baseline(cfg, n_points_budget):
    """the identical step with the CPU oracle behind every operator; bounded sample = ONE ~150k-point scene,
    one warm-up-free step (tens of seconds on the host cores)"""
    from oracle.oracle_backend import OracleBackend
    prev = ms_backend.set_backend(OracleBackend())
    try:
        cores = min(os.cpu_count() or 1, 32)   # more threads only thrash on these loop sizes
        torch.set_num_threads(cores)
        from oracle import oracle as O
        O.lib().orc_set_threads(cores)
        dev = torch.device("cpu")
        model = build(cfg, dev)
        opt = model.configure_optimizers()
        batch = make_batch([0], dev)
        t0 = time.perf_counter()
        train_step(model, model, opt, batch)
        dt = time.perf_counter() - t0
        return {"value": round(1.0 / dt, 5), "unit": "scenes/sec", "cores": cores, "kind": "port",
                "sample": f"1 step on 1 synthetic scene ({batch['point_xyz'].shape[0]} points, "
                          f"{batch['voxel_xyz'].shape[0]} voxels), {dt:.1f} s, oracle C (OpenMP) + torch CPU"}
    finally:
        ms_backend.set_backend(prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="scenes per GPU per step (reference batch_size 4)")
    ap.add_argument("--pool", type=int, default=3, help="distinct pre-generated batches per rank (cycled)")
    ap.add_argument("--model", default="pointgroup", choices=["pointgroup", "hais", "softgroup"],
                    help="headline metric = pointgroup; hais / softgroup are the other BASELINE configs")
    ap.add_argument("--density", type=float, default=1700.0,
                    help="points per m^2 of the synthetic scenes (1700 = the ~150k-point headline workload; small "
                         "values expose the host launch floor)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    rank, local, world = init_distributed()
    assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    cfg = load_config([f"model={args.model}", "data=scannetv2"])
    be = ms_backend.get_backend()          # raises if libminsu3d_hip.so is missing: no fallback

    model = build(cfg, device)
    ddp = wrap_ddp(model, device, find_unused_parameters=False)   # grouping branch on: every parameter gets a gradient
    opt = model.configure_optimizers()
    scene_kwargs = None if args.density == 1700.0 else {"density": args.density}
    batches = [make_batch(shard_scene_seeds(s, args.batch, rank, world), device, scene_kwargs) for s in range(args.pool)]
    n_pts = float(np.mean([b["point_xyz"].shape[0] for b in batches])) / args.batch
    n_vox = float(np.mean([b["voxel_xyz"].shape[0] for b in batches])) / args.batch

    timer = None
    if not args.no_roofline:
        timer = ms_backend.KernelTimer(lambda name, K, cin, cout: name == "spconv_fwd" and K == 27 and cin == 16 and cout == 16,
                                       be.lib)
        timer.min_rows = 50_000 * args.batch      # the backbone's full-resolution level, not the proposal grids of the ScoreNet
        be.kernel_timer = timer

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        train_step(model, ddp, opt, batches[i % args.pool], batches[(i + 1) % args.pool])
    if timer is not None:
        timer.enabled = True
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = train_step(model, ddp, opt, batches[i % args.pool], batches[(i + 1) % args.pool])
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(loss).item(), "loss is not finite"

    if rank == 0:
        scenes = world * args.batch * args.steps
        line = {
            "metric": f"scenes/sec (fwd+bwd) {cfg.model.network.module} on ~150k-pt 2cm voxels",
            "value": round(scenes / dt, 3), "unit": "scenes/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg.model.network.module} m={cfg.model.network.m}, synthetic ScanNet-shaped scenes "
                                   f"(~{n_pts / 1000:.0f}k points, ~{n_vox / 1000:.0f}k voxels @2cm each), "
                                   f"{args.batch} scenes/GPU/step, grouping+ScoreNet branch on, fwd+loss+bwd+Adam",
                       "scenes_per_gpu": args.batch, "parallelism": f"dp{world}",
                       "grouping_inputs": "GT labels, GT offsets + N(0,4cm) (random-init net groups nothing)",
                       "input_pipelining": "coordinate-only work of step i+1's batch (row order, kernel maps, pair lists) "
                                           "runs on a side stream during step i's backward; every step builds its own"},
        }
        if timer is not None:
            s = timer.summary()
            if s:
                ach = s["avg_bytes"] / (s["avg_ms"] * 1e-3) / 1e9
                traffic = None   # HBM bytes per launch from the separate rocprofv3 --pmc passes of this command
                tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
                if os.path.exists(tpath):
                    with open(tpath) as f:
                        traffic = int(json.load(f)["traffic_bytes_per_launch"])
                line["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                                    "kernel": "spconv_fwd_pairlist_kernel<1,1> (3x3x3 16->16 pair-list gather/MFMA at full resolution, forward and backward-data launches)",
                                    "launches": s["launches"], "avg_us": round(s["avg_ms"] * 1e3, 2),
                                    "algorithmic_bytes_per_launch": int(s["avg_bytes"])}
        if world == 1 and not args.no_cpu_baseline:
            be.kernel_timer = None
            line["cpu_baseline"] = cpu_baseline(cfg, n_pts)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
