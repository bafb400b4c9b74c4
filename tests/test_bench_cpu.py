"""bench.py's multi-process path without GPUs: the driver's launch line (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`) on N = 1, 2, 4 host
processes over gloo -- environment rendezvous, scene sharding, DDP all-reduce, barrier, max-over-ranks timing, ONE JSON
line from rank 0 with the contract's keys.  (BASELINE config 4 names 8 GPUs; an 8-GPU node is not available to the
build, so the launch path is exercised here and RCCL itself only by the driver.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE = '{"room": [1.0, 0.8], "n_boxes": 2, "density": 700.0, "wall_h": 0.4}'
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "step_ms"}


@pytest.mark.parametrize("world", [1, 2, 4])
def test_driver_launch_line_on_gloo(world):
    tail = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", "1", "--pool", "2", "--scene", SCENE,
            "--override", "model.network.blocks=[1,2]"]
    script = os.path.join(ROOT, "tests", "bench_dryrun.py")
    if world == 1:
        cmd = [sys.executable, script] + tail
    else:
        port = 29600 + (os.getpid() + 7 * world) % 300
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), script] + tail
    env = dict(os.environ, OMP_NUM_THREADS="1", MS3D_DIST_BACKEND="gloo")
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                                  # rank 0 only
    rec = json.loads(lines[0])
    assert KEYS <= set(rec), KEYS - set(rec)
    assert rec["n_gpus"] == world and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["parallelism"] == f"dp{world}" and rec["higher_is_better"] is True and "dry_run" in rec
    # value = scenes of ALL ranks / max-over-ranks time
    assert abs(rec["value"] - world * 1 * 2 / (rec["ms_per_step"] * 2 / 1000.0)) < 1e-2 * rec["value"]
    assert rec["step_ms"]["min"] <= rec["step_ms"]["median"] <= rec["step_ms"]["max"]


def test_eight_ranks_get_disjoint_scenes_and_cores():
    """BASELINE config 4's process layout (8 ranks on one node, LOCAL_WORLD_SIZE = 8) on host processes over gloo: every
    rank trains on its own scenes (no scene id twice) and -- when the container has the cores -- on its own core slice;
    one JSON line, value = scenes of all 8 ranks / the slowest rank's time.  The launch line is the driver's."""
    world = 8
    tail = ["--gpus", str(world), "--steps", "1", "--warmup", "1", "--batch", "1", "--pool", "2", "--scene", SCENE,
            "--override", "model.network.blocks=[1,2]"]
    script = os.path.join(ROOT, "tests", "bench_dryrun.py")
    port = 29600 + (os.getpid() + 61) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), script] + tail
    env = dict(os.environ, OMP_NUM_THREADS="1", MS3D_DIST_BACKEND="gloo")
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["config"]["parallelism"] == "dp8" and rec["scaling"] == "weak"
    ranks = sorted(rec["dry_run_ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == list(range(8))
    ids = [i for r in ranks for i in r["scene_ids"]]
    assert len(ids) == len(set(ids)) == 8 * 2                      # 8 ranks x pool of 2 one-scene batches, all different
    allowed = len(os.sched_getaffinity(0))
    if allowed >= 8:
        cores = [set(r["cores"]) for r in ranks]
        assert all(len(c) == allowed // 8 for c in cores)
        assert all(a.isdisjoint(b) for i, a in enumerate(cores) for b in cores[i + 1:])
    assert abs(rec["value"] - 8 * 1 * 1 / (rec["ms_per_step"] / 1000.0)) < 1e-2 * rec["value"]


def test_ranks_get_disjoint_core_slices():
    """8 ranks x (main + helper + loader threads) + torch's intra-op pools must not fight for the same cores: each rank
    pins itself to its own slice of the cores the process may use (parallel.pin_rank_threads)"""
    code = ("import os, sys, json; sys.path.insert(0, %r)\n"
            "from minsu3d_amd.parallel import pin_rank_threads\n"
            "import torch\n"
            "r = pin_rank_threads(int(sys.argv[1]), int(sys.argv[2]))\n"
            "print(json.dumps([r, sorted(os.sched_getaffinity(0)), torch.get_num_threads()]))\n" % ROOT)
    allowed = sorted(os.sched_getaffinity(0))
    world = 4 if len(allowed) >= 4 else len(allowed)
    if world < 2:
        pytest.skip("one core")
    seen = []
    for rank in range(world):
        out = subprocess.run([sys.executable, "-c", code, str(rank), str(world)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        mine, mask, nthreads = json.loads(out.stdout.strip().splitlines()[-1])
        assert mine == mask and len(mine) == len(allowed) // world and 1 <= nthreads <= len(mine)
        seen.append(set(mine))
    assert all(a.isdisjoint(b) for i, a in enumerate(seen) for b in seen[i + 1:])
    out = subprocess.run([sys.executable, "-c", code, "0", "4"], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, MS3D_PIN="0"))
    assert json.loads(out.stdout.strip().splitlines()[-1])[0] is None
