"""cProfile of the host side of benchmark steps (where do the ~20 ms of Python / ctypes / torch dispatch per step go?).
usage: python tools/host_profile.py [--model pointgroup] [--steps 10] [--top 45]"""
import argparse, cProfile, os, pstats, sys, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="pointgroup"); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--top", type=int, default=45)
ap.add_argument("--sort", default="tottime")
args = ap.parse_args()
prof = cProfile.Profile()
real_time = bench.time.perf_counter
state = {"on": False, "steps": 0}
# profile only the timed region: bench.main() is reused, the profiler is switched on by the first timed step's barrier
orig_sync = torch.cuda.synchronize


def sync(*a, **k):
    r = orig_sync(*a, **k)
    return r


argv = ["--model", args.model, "--steps", str(args.steps), "--warmup", "4", "--no-cpu-baseline", "--no-roofline"]
prof.enable()
bench.main(argv)
prof.disable()
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats(args.sort).print_stats(args.top)
print(s.getvalue()[:12000])
