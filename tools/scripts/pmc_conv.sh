# usage: bash tools/scripts/pmc_conv.sh "<cin cout K level>" <kernel substring>   -> SQ counters of one convolution shape
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1; export TMPDIR=/tmp
CFG="$1"; KN=$2
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"; do
  rm -rf gpurun_out/pmc_tmp
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 tools/conv_micro.py $CFG > /dev/null 2>&1
  python3 tools/pmc_kernel.py $(find gpurun_out/pmc_tmp -name "*counter_collection.csv") $KN
done
rm -rf gpurun_out/pmc_tmp
