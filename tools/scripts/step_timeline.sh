# usage: bash tools/scripts/step_timeline.sh <model> <tag>   -> gpurun_out/<tag>_step_timeline.txt
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1; export TMPDIR=/tmp
M=$1; TAG=$2
rm -rf gpurun_out/tl_$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$TAG -o tl -- python3 bench.py --model $M --steps 12 --warmup 8 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 tools/step_timeline.py $(find gpurun_out/tl_$TAG -name '*kernel_trace.csv') --top --window > gpurun_out/${TAG}_step_timeline.txt 2>&1
rm -rf gpurun_out/tl_$TAG
