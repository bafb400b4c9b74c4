# one-rank RCCL process group (MS3D_FORCE_PG=1) against the plain single-process step, with the prefetch on its own / the side stream
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
run() { echo "== $*"; env "$@" python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --also none 2>gpurun_out/pg_err.txt | python3 tools/bench_line.py || tail -5 gpurun_out/pg_err.txt; }
for rep in 1 2; do
run MS3D_X=0
run MS3D_FORCE_PG=1 MS3D_PREFETCH_STREAM=own
run MS3D_FORCE_PG=1
run MS3D_FORCE_PG=1 MS3D_DDP_STATIC=1
run MS3D_FORCE_PG=1 MS3D_DDP_BUCKET_MB=200
run MS3D_FORCE_PG=1 MS3D_DDP_HOOK=1
done
