# usage: bash tools/scripts/pmc_traffic.sh <model> <commit> [tag]  -> gpurun_out/<tag>_traffic_<model>.json
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1; export TMPDIR=/tmp
M=$1; C=$2; TAG=${3:-r06}
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$ctr
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$ctr -o p -- python3 bench.py --model $M --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
python3 tools/pmc_traffic.py $(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv") $(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv") 3 $M $C > gpurun_out/${TAG}_traffic_$M.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
head -12 gpurun_out/${TAG}_traffic_$M.json
