# the round's evidence run: the whole GPU suite, then the driver's bench line
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/gputest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_head.json 2> gpurun_out/bench_head.err
