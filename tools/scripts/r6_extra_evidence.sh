cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1; export TMPDIR=/tmp
python3 bench.py --all-kernels --no-cpu-baseline --also none > /dev/null 2> gpurun_out/r06_all_kernels.txt
python3 bench.py --model hais --all-kernels --no-cpu-baseline --also none > /dev/null 2> gpurun_out/r06_all_kernels_hais.txt
python3 tools/group_micro.py > gpurun_out/r06_group_micro.txt 2>&1
bash tools/scripts/step_timeline.sh > gpurun_out/r06_step_timeline_run.log 2>&1
