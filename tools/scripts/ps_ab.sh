cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
timeout 1200 python -m pytest tests/test_sparse_gpu.py tests/test_determinism_gpu.py -x -q 2>&1 | tail -3
run() { echo "== $M $*"; env "$@" python3 bench.py --model $M --steps 30 --warmup 5 --no-cpu-baseline --also none 2>gpurun_out/ab_err.txt | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['step_ms']['median'], 'conv ms', d['roofline']['kernel_ms_per_step'], 'frac', d['roofline']['frac'])" || tail -5 gpurun_out/ab_err.txt; }
for M in pointgroup hais; do for rep in 1 2; do
run MS3D_PS_NBT=2
run MS3D_X=1
done; done
