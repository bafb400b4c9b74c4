# usage: bash tools/scripts/prof_model.sh <model> <tag>   -> gpurun_out/<tag>_kernel_stats.csv + summary on stdout
set -e
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1; export TMPDIR=/tmp
M=$1; TAG=$2
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --model $M --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench.log 2>&1 || true
grep '^{' gpurun_out/${TAG}_bench.log | tail -1
find gpurun_out/prof_$TAG -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
python3 tools/prof_summary.py gpurun_out/${TAG}_kernel_stats.csv 13 40
find gpurun_out/prof_$TAG -name '*kernel_trace.csv' -delete
