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
