# two ranks sharing one GPU over gloo (the test-only DDP configuration) through bench.py: per-step durations under a few switches
cd "${GRAFT_REPO_ROOT:?}" || exit 1
run() { echo "== $*"; env "$@" MS3D_STEP_DUMP=1 MS3D_SHARE_DEVICE=1 MS3D_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 8 --warmup 3 --no-cpu-baseline --also none --no-roofline 2>&1 | grep "step_ms:" | cut -c1-200; }
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=16
run MS3D_PREFETCH_AT=backward
run MS3D_PREFETCH_AT=grouping_ev
