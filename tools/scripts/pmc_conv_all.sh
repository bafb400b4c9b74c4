# usage: bash tools/scripts/pmc_conv_all.sh "<cin cout K level>" <out file> <kernel substring> [<kernel substring> ...]
# SQ + memory-path counter sets of one convolution shape (tools/conv_micro.py), one rocprofv3 pass per set (program directly
# behind `--`, --pmc with --kernel-trace only), averaged per launch for every kernel whose name contains a substring
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1; export TMPDIR=/tmp
CFG="$1"; OUT="$2"; shift 2
echo "## conv_micro.py $CFG   env: $(env | grep ^MS3D_ | tr '\n' ' ')" >> $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES"; do
  rm -rf gpurun_out/pmc_tmp
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 tools/conv_micro.py $CFG > /dev/null 2>&1
  f=$(find gpurun_out/pmc_tmp -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then for KN in "$@"; do echo "-- $KN"; python3 tools/pmc_kernel.py $f "$KN"; done >> $OUT; else echo "no output for: $set" >> $OUT; fi
done
rm -rf gpurun_out/pmc_tmp
