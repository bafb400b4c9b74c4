#!/usr/bin/env python3
"""Headline benchmark: scenes/sec (fwd + loss + bwd + Adam) of PointGroup (m=16) on synthetic ScanNet-shaped
~150k-point scenes voxelised at 2 cm, batch 4 scenes per GPU, grouping branch active
(current_epoch > prepare_epochs), data-parallel over N GPUs of one node (one process per GPU, RCCL).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.

`roofline`: measured live with HIP events on a sample of the timed steps (one in 20: ~520 event records cost a step ~8 ms).
The events are recorded INSIDE the library calls, immediately around each kernel, on the stream it is launched on.
The object describes the dominant kernel family of the step -- ALL sparse-convolution launches (forward,
backward-data, backward-weight: the SURVEY 8d / BASELINE.md figure, sum over the layers of
nM*(Cin+Cout)*4 + nM*8 + K*Cin*Cout*4 with each table's real pair count nM, divided by the summed kernel time) --
and `top_kernels` lists the five largest kernel groups by time (convolutions by variant and shape, and the grouping
operators with their SURVEY 8d byte formulas).  `traffic` (HBM bytes of the same convolution kernels per step, from
PMC counters) needs separate rocprofv3 --pmc passes, which this script cannot run on itself: it is read from
the newest profiles/rNN_traffic_<model>.json (made by tools/scripts/pmc_traffic.sh from FETCH_SIZE / WRITE_SIZE passes of this
script; corrections as MI355X_MICROARCH.md prescribes) and is null when that file is absent or carries the digest of
other kernel sources than the ones running (`kernel_source_digest`).  `step_ms` = median / min / max / mean of the
per-step durations between HIP events at the step boundaries; `ms_per_step` = wall time of the timed region / steps.

`cpu_baseline` times the same training step on the host cores with every operator served by the CPU oracle (a PORT
of the reference algorithms -- the reference's own CPU path needs MinkowskiEngine, which is not available), on a
bounded sample (one ~150k-point scene: 1 warm-up + 3 timed steps, median), N=1 / rank 0 only, with a per-operator
breakdown; `--cpu-config1` runs BASELINE config 1 (4 x ~20k-point scenes) instead.
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
from minsu3d_amd.parallel import init_distributed, shard_scene_seeds, stream_plan, wrap_ddp  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TFLOPS = 157.3  # dense f32 MFMA (v_mfma_f32_16x16x4_f32), same guide / SURVEY 8d
SAMPLE_EVERY = 20          # one timed step in 20 carries the kernel events (a sampled step costs ~8 ms of event records)


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


PREFETCH_AT = os.environ.get("MS3D_PREFETCH_AT", "grouping")


def train_step(model, ddp, opt, batch, next_batch=None):
    opt.zero_grad(set_to_none=True)
    # input pipelining, as a data loader would do it: the coordinate-only work of the NEXT batch (row order, kernel maps,
    # pair lists) is built on its own stream -- issued right behind this step's backbone, so that it runs inside the
    # grouping window where the chip is mostly idle (MS3D_PREFETCH_AT=backward: beside the backward pass, as in round 3)
    prefetch = None
    if next_batch is not None:
        def prefetch():
            # (_ev / bfs: the prefetch stream waits until the GPU -- not just the interpreter -- has reached this point)
            ME.prefetch_coordinates(next_batch["voxel_xyz"], model.backbone.n_levels,
                                    wait_current_stream=PREFETCH_AT in ("grouping_ev", "bfs"),
                                    channels=model.backbone.level_channels, point_map=next_batch["voxel_point_map"])
        if PREFETCH_AT in ("grouping", "grouping_ev"):
            model.schedule_after_backbone(prefetch)
        elif PREFETCH_AT == "bfs" and hasattr(model, "schedule_after_ballquery"):
            model.schedule_after_ballquery(prefetch)
        elif PREFETCH_AT == "proposals":
            model.schedule_after_grouping(prefetch)
    out = ddp(batch)
    loss = sum(model._loss(batch, out).values())
    if prefetch is not None and PREFETCH_AT not in ("grouping", "grouping_ev", "bfs", "proposals"):
        prefetch()
    loss.backward()
    opt.step()
    return loss


# --------------------------------------------------------------------------------------------- CPU baseline
_CPU_FAMILIES = (("sparse convolution (forward, backward-data, backward-weight)", ("conv_", "prep_weights")),
                 ("coordinate maps / quantize", ("sparse_quantize", "kmap", "downsample", "spatial_order", "pairlist",
                                                 "offsetlist", "identity_table")),
                 ("batch norm", ("bn_",)),
                 ("ball query", ("ballquery",)),
                 ("BFS clustering / aggregation", ("bfs", "hierarchical")),
                 ("voxelisation segment ops, pools, IoU", ("sec_", "roipool", "global_avg", "get_", "scatter_add")))


class _TimedBackend:
    """proxy that accumulates the wall time of every backend method (the oracle serves them synchronously)"""

    def __init__(self, inner):
        self._inner, self.seconds = inner, {}

    def __getattr__(self, name):
        attr = getattr(self._inner, name)
        if not callable(attr):
            return attr

        def timed(*a, **k):
            t0 = time.perf_counter()
            try:
                return attr(*a, **k)
            finally:
                self.seconds[name] = self.seconds.get(name, 0.0) + time.perf_counter() - t0
        return timed


def cpu_baseline(cfg, config1=False):
    """the identical step with the CPU oracle behind every operator: 1 warm-up + 3 timed steps, median"""
    from oracle.oracle_backend import OracleBackend
    from oracle import oracle as O
    proxy = _TimedBackend(OracleBackend())
    prev = ms_backend.set_backend(proxy)
    try:
        cores = min(os.cpu_count() or 1, 32)   # more threads only thrash on these loop sizes
        torch.set_num_threads(cores)
        O.lib().orc_set_threads(cores)
        dev = torch.device("cpu")
        model = build(cfg, dev)
        opt = model.configure_optimizers()
        if config1:   # BASELINE config 1: 4 synthetic ~20k-point scenes (room 2 m x 1.6 m, 3 boxes)
            batch = make_batch([0, 1, 2, 3], dev, dict(room=(2.0, 1.6), n_boxes=3))
        else:
            batch = make_batch([0], dev)
        n_scenes = len(batch["scan_ids"])
        times, per_op = [], []
        for i in range(4):
            proxy.seconds = {}
            t0 = time.perf_counter()
            train_step(model, model, opt, batch)
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt); per_op.append(dict(proxy.seconds))
        mid = int(np.argsort(times)[len(times) // 2])
        dt, ops = times[mid], per_op[mid]
        breakdown, used = {}, 0.0
        for fam, prefixes in _CPU_FAMILIES:
            t = sum(v for k, v in ops.items() if any(p in k for p in prefixes))
            breakdown[fam] = round(t, 3); used += t
        breakdown["dense heads, losses, autograd glue, Adam (torch CPU)"] = round(max(dt - used, 0.0), 3)
        return {"value": round(n_scenes / dt, 5), "unit": "scenes/sec", "cores": cores, "kind": "port",
                "sample": f"{n_scenes} synthetic scene(s) ({batch['point_xyz'].shape[0]} points, "
                          f"{batch['voxel_xyz'].shape[0]} voxels), 1 warm-up + 3 timed steps, median {dt:.2f} s "
                          f"(all: {', '.join(f'{t:.2f}' for t in times)}), oracle C (OpenMP) + torch CPU",
                "seconds_per_step_by_operator": breakdown}
    finally:
        ms_backend.set_backend(prev)


# --------------------------------------------------------------------------------------------- roofline
def _conv_variant(kind, K, cin, cout, rows, lib):
    """the kernel a convolution launch is served by (mirrors fwd_geometry / the wgrad dispatch in csrc/spconv.hip)"""
    listed = bool(lib.ms3d_kmap_pairlist_wanted(int(K), int(rows)))
    if kind == "spconv_wgrad":
        ol = listed and cin <= 64 and cout <= 64
        if lib.ms3d_spconv_wgrad_is_bf16x3(int(rows), int(K), int(cin), int(cout), int(ol)):
            return "bf16x3 table-walk"
        return "offset-list" if ol else "table-walk"
    split = "bf16x3 " if int(lib.ms3d_spconv_aux_kind(int(K), int(cin), int(cout))) == 2 else ""
    if -(-rows // 16) <= 1100 and -(-cin // 16) * -(-cout // 16) >= 4:
        return split + "small"
    if listed and max(cin, cout) <= 32:
        return "pair-list"
    if listed and not split and int(lib.ms3d_spconv_wants_stream_image(int(K), int(cin), int(cout))):
        return "weight-stream"
    return split + "table-walk"


def kernel_source_digest():
    """sha256[:16] over the convolution kernels' sources: a committed PMC traffic file carries the digest of the code it
    was measured on, and `roofline.traffic` is reported only when that is the code running now"""
    import hashlib
    h = hashlib.sha256()
    for f in ("spconv.hip", "coords.hip", "common.h"):
        with open(os.path.join(ROOT, "minsu3d_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def roofline_report(groups, n_sampled, lib, model_name="pointgroup"):
    """groups: KernelTimer.summary() over `n_sampled` steps -> (roofline dict, top_kernels list)"""
    if not groups or n_sampled == 0:
        return None, None
    conv = {k: v for k, v in groups.items() if k[0].startswith("spconv_")}
    tot = {f: sum(v[f] for v in conv.values()) for f in ("ms", "bytes", "flops", "launches")}
    ach = tot["bytes"] / (tot["ms"] * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
            "kernel": "spconv_* -- every sparse-convolution launch of the step (forward, backward-data, backward-weight; "
                      "backbone + ScoreNet): sum of algorithmic bytes / sum of kernel time (SURVEY 8d)",
            "launches_per_step": round(tot["launches"] / n_sampled, 1),
            "kernel_ms_per_step": round(tot["ms"] / n_sampled, 3),
            "algorithmic_bytes_per_step": int(tot["bytes"] / n_sampled),
            "flops_per_step": int(tot["flops"] / n_sampled),
            # useful f32-equivalent flops against the f32 MFMA peak (layers on three-piece bf16 operands issue 6 bf16
            # instructions per 32 channels instead of 8 f32 ones: the figure is NOT matrix-pipe occupancy for them)
            "mfma_f32_frac": round(tot["flops"] / (tot["ms"] * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
            "steps_sampled": n_sampled}
    # HBM bytes of the same kernels from the PMC passes committed with this code (separate rocprofv3 --pmc FETCH_SIZE /
    # --pmc WRITE_SIZE runs of this script, tools/scripts/pmc_traffic.sh); null when the file is missing or is for
    # another model
    import glob
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_traffic_{model_name}.json")))
    tfile = tfiles[-1] if tfiles else None          # the newest round's passes
    if tfile is not None:
        with open(tfile) as fh:
            t = json.load(fh)
        if t.get("kernel_source_digest") == kernel_source_digest():
            roof["traffic"] = t.get("spconv_hbm_bytes_per_step")
            roof["traffic_source"] = f"profiles/{os.path.basename(tfile)} ({t.get('commit', '?')}): {t.get('method', '')}"
            if roof["traffic"]:
                # `frac` is an ALGORITHMIC rate (a voxel row fetched from HBM once is re-gathered from L2 for its other
                # pairs); this one is the physical HBM rate of the same kernels: counter bytes / kernel time / peak
                roof["hbm_frac_physical"] = round(roof["traffic"] / (tot["ms"] / n_sampled * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        else:   # counters of other code are not this code's traffic
            roof["traffic_source"] = (f"profiles/{os.path.basename(tfile)} was measured on other kernel sources "
                                      f"(digest {t.get('kernel_source_digest')} != {kernel_source_digest()}): not reported")
    rows = []
    for k, v in groups.items():
        if k[0].startswith("spconv_") and len(k) == 6:
            _, K, cin, cout, nrows, nlayers = k
            name = (f"{k[0]} [table-walk, BATCHED: {nlayers} layers of one kernel shape class in one launch] backward-weight, "
                    f"first layer K={K} {cin}->{cout} rows={nrows}")
        elif k[0].startswith("spconv_"):
            _, K, cin, cout, nrows = k
            kind = "forward/backward-data" if k[0] == "spconv_fwd" else "backward-weight"
            name = f"{k[0]} [{_conv_variant(k[0], K, cin, cout, nrows, lib)}] {kind} K={K} {cin}->{cout} rows={nrows}"
        else:
            name = k[0]
        gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
        rows.append({"name": name, "launches_per_step": round(v["launches"] / n_sampled, 1),
                     "ms_per_step": round(v["ms"] / n_sampled, 3), "avg_us": round(1e3 * v["ms"] / v["launches"], 1),
                     "algorithmic_bytes_per_step": int(v["bytes"] / n_sampled), "GB/s": round(gbs, 1),
                     "frac": round(gbs / HBM_PEAK_GBS, 4)})
    rows.sort(key=lambda r: -r["ms_per_step"])
    spans = [r for r in rows if r["name"].startswith("grouping span")]
    rows = [r for r in rows if not r["name"].startswith("grouping span")]
    for r in rows:
        if r["name"] in ("pg_bfs_cluster", "ballquery_batch_p") and spans:
            r["note"] = ("two calls per step (original and shifted coordinates) run CONCURRENTLY on two streams: "
                         "ms_per_step adds their intervals; the wall time of all four calls together is the span below")
    top = rows[:5]
    for r in spans:
        r.pop("GB/s", None); r.pop("frac", None); r.pop("algorithmic_bytes_per_step", None)
        top.append(r)
    return roof, top


def other_model_line(model_name, args):
    """BASELINE configs 3 / 4 for the driver's one JSON line: the same benchmark for another model family, run in a CHILD
    process after (outside) the timed headline region -- a fresh process, so nothing of it touches the headline numbers"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--model", model_name, "--gpus", "1", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--batch", str(args.batch), "--no-cpu-baseline", "--also", "none"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        roof = d.get("roofline") or {}
        return {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                "step_ms_median": d["step_ms"]["median"], "workload": d["config"]["workload"],
                "roofline_frac": roof.get("frac"), "roofline_achieved_GBs": roof.get("achieved"),
                "conv_kernel_ms_per_step": roof.get("kernel_ms_per_step"), "traffic": roof.get("traffic")}
    except Exception as e:      # the headline line must not be lost to a failure of an extra
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def main(argv=None):
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
    ap.add_argument("--cpu-config1", action="store_true", help="cpu_baseline on BASELINE config 1 (4 x ~20k-point scenes)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--all-kernels", action="store_true", help="print every timed kernel group to stderr")
    ap.add_argument("--override", action="append", default=[], help="Hydra-style key=value on top of the model's config")
    ap.add_argument("--scene", default=None, help="JSON keyword arguments of synthetic.make_scene (smaller scenes)")
    ap.add_argument("--also", default="auto",
                    help="comma list of the other model families (hais,softgroup) to run AFTER the timed headline region, "
                         "each in its own child process, and report under `extra` (BASELINE configs 3 / 4: value, "
                         "ms_per_step, roofline.frac); 'none' = off; 'auto' = hais,softgroup for the plain headline "
                         "invocation (PointGroup, one GPU, roofline and cpu_baseline on), none otherwise")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu = DRY RUN of the launch / sharding / barrier / max-over-ranks / reporting path on host "
                         "tensors over gloo (tests/bench_dryrun.py installs the operator backend); never a measurement")
    args = ap.parse_args(argv)
    dry = args.device == "cpu"

    rank, local, world = init_distributed()
    assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    if rank != 0:
        # stdout belongs to rank 0's JSON line: whatever a library of another rank prints (RCCL's banner, flushed at exit)
        # goes to stderr
        sys.stdout.flush()
        os.dup2(2, 1)
    device = torch.device("cpu") if dry else torch.device("cuda", local)
    if not dry:
        torch.cuda.set_device(device)
    cfg = load_config([f"model={args.model}", "data=scannetv2"] + list(args.override))
    be = ms_backend.get_backend()          # raises if libminsu3d_hip.so is missing: no fallback

    model = build(cfg, device)
    ddp = wrap_ddp(model, device, find_unused_parameters=False)   # grouping branch on: every parameter gets a gradient
    opt = model.configure_optimizers()
    scene_kwargs = None if args.density == 1700.0 else {"density": args.density}
    if args.scene:
        scene_kwargs = {k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(args.scene).items()}
    seeds = [shard_scene_seeds(s, args.batch, rank, world) for s in range(args.pool)]
    batches = [make_batch(sd, device, scene_kwargs) for sd in seeds]
    rank_report = None
    if dry and world > 1:
        # dry run only: which scenes and which host cores every rank got (tests/test_bench_cpu.py checks them disjoint)
        mine = {"rank": rank, "scene_ids": sorted(i for sd in seeds for i in sd),
                "cores": sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []}
        rank_report = [None] * world
        torch.distributed.all_gather_object(rank_report, mine)
    n_pts = float(np.mean([b["point_xyz"].shape[0] for b in batches])) / args.batch
    n_vox = float(np.mean([b["voxel_xyz"].shape[0] for b in batches])) / args.batch

    timer = None
    if not args.no_roofline and rank == 0 and not dry:
        timer = ms_backend.KernelTimer(be.lib)
        timer.reserve(700 * (args.steps // SAMPLE_EVERY + 1))
        be.kernel_timer = timer

    def sync_all():
        if not dry:
            torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            if not dry:
                torch.cuda.synchronize()

    for i in range(args.warmup):
        train_step(model, ddp, opt, batches[i % args.pool], batches[(i + 1) % args.pool])
    sync_all()
    n_sampled = 0
    # per-step durations on the GPU's clock: one event per step boundary on the main stream (recording costs ~2 us of
    # host time and no synchronisation); `ms_per_step` stays wall time / steps
    class _HostMark:        # dry run: host clock
        def record(self): self.t = time.perf_counter()
        def elapsed_time(self, other): return 1000.0 * (other.t - self.t)
    marks = [_HostMark() if dry else torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        if timer is not None:
            timer.sampling = (i % SAMPLE_EVERY == SAMPLE_EVERY // 2) or (args.steps <= SAMPLE_EVERY // 2 and i == args.steps - 1)
            n_sampled += int(timer.sampling)
        loss = train_step(model, ddp, opt, batches[i % args.pool], batches[(i + 1) % args.pool])
        marks[i + 1].record()
    sync_all()
    dt = time.perf_counter() - t0
    step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
    if os.environ.get("MS3D_STEP_DUMP") and rank == 0:
        print("step_ms:", " ".join(f"{v:.2f}" for v in step_ms), file=sys.stderr)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(loss).item(), "loss is not finite"

    if rank == 0:
        scenes = world * args.batch * args.steps
        line = {
            "metric": f"scenes/sec (fwd+bwd) {cfg.model.network.module} on ~150k-pt 2cm voxels",
            "value": round(scenes / dt, 3), "unit": "scenes/sec",
            # the same rate from the MEDIAN step (rank 0's clock): `value` is wall time over all steps incl. the first
            # step's empty pipeline and any host hiccup of the box; the two together show the box noise in one line
            "value_median": round(world * args.batch / (float(np.median(step_ms)) * 1e-3), 3),
            "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3),
            "step_ms": {"median": round(float(np.median(step_ms)), 3), "min": round(float(step_ms.min()), 3),
                        "max": round(float(step_ms.max()), 3), "mean": round(float(step_ms.mean()), 3),
                        "clock": "HIP events at the step boundaries on the main stream (rank 0)"},
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "arithmetic": "f32 storage and accumulation everywhere; layers with both sides >= 48 channels multiply on exact "
                          "3-piece bf16 splits of both operands (6 of the 9 piece products, v_mfma_f32_16x16x32_bf16, f32 "
                          "accumulate: 1.7-2.0e-6 of the largest output vs float64, the f32 MFMA kernel's own 1.3-1.6e-6), "
                          "all other layers on v_mfma_f32_16x16x4_f32",
            "data": "synthetic",
            "config": {"workload": f"{cfg.model.network.module} m={cfg.model.network.m}, synthetic ScanNet-shaped scenes "
                                   f"(~{n_pts / 1000:.0f}k points, ~{n_vox / 1000:.0f}k voxels @2cm each), "
                                   f"{args.batch} scenes/GPU/step, grouping+ScoreNet branch on, fwd+loss+bwd+Adam",
                       "scenes_per_gpu": args.batch, "parallelism": f"dp{world}",
                       "streams": stream_plan(world),
                       "grouping_inputs": "GT labels, GT offsets + N(0,4cm) (random-init net groups nothing)",
                       "input_pipelining": "coordinate-only work of step i+1's batch (row order, kernel maps, pair lists) "
                                           "runs on its own stream during step i's " +
                                           ("grouping window" if PREFETCH_AT == "grouping" else "backward pass") +
                                           "; every step builds its own"},
        }
        if timer is not None:
            timer.sampling = False
            be.kernel_timer = None
            groups = timer.summary()
            roof, top = roofline_report(groups, n_sampled, be.lib, args.model)
            if roof:
                line["roofline"], line["top_kernels"] = roof, top
            if args.all_kernels:
                print("step_ms: " + " ".join(f"{t:.2f}" for t in step_ms), file=sys.stderr)
                for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
                    print(f"{v['ms'] / max(n_sampled, 1):8.3f} ms/step {v['launches'] / max(n_sampled, 1):6.1f} launches "
                          f"{v['bytes'] / (v['ms'] * 1e-3) / 1e9:8.1f} GB/s  {k}", file=sys.stderr)
        if dry:
            line["dry_run"] = "host tensors over gloo: exercises the launch path only, the numbers mean nothing"
            if rank_report is not None:
                line["dry_run_ranks"] = rank_report
        if world == 1 and not args.no_cpu_baseline and not dry:
            line["cpu_baseline"] = cpu_baseline(cfg, config1=args.cpu_config1)
        also = args.also
        if also == "auto":
            plain = (args.model == "pointgroup" and world == 1 and not dry and not args.no_roofline
                     and not args.no_cpu_baseline and not args.override and not args.scene and args.density == 1700.0)
            also = "hais,softgroup" if plain else "none"
        if also != "none" and world == 1 and not dry:
            line["extra"] = {m: other_model_line(m, args) for m in also.split(",") if m and m != args.model}
    # the collective library goes down FIRST and the C stdio buffers are flushed: RCCL prints a version banner through
    # printf (found in round 6, the first time RCCL ran here: behind a pipe it surfaced AFTER the JSON line) -- rank 0's
    # JSON line is the last thing this process writes to stdout
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
