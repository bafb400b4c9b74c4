cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_determinism_gpu.py tests/test_forward_pins_gpu.py tests/test_reference_pins_gpu.py -x -q 2>&1 | tail -3
run() { echo "== $*"; env "$@" python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --also none 2>gpurun_out/ab_err.txt | python3 tools/bench_line.py || tail -5 gpurun_out/ab_err.txt; }
for rep in 1 2 3; do
run MS3D_RESBLOCK_EXT=0
run MS3D_RESBLOCK_EXT=1
done
MS3D_RESBLOCK_EXT=0 python3 tools/phase_timeline.py 2>&1 | grep -E "scorenet_begin|backbone_begin ->|total"
MS3D_RESBLOCK_EXT=1 python3 tools/phase_timeline.py 2>&1 | grep -E "scorenet_begin|backbone_begin ->|total"
