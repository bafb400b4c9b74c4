# usage: bash tools/scripts/pmc_conv2.sh "<cin cout K level>" <kernel substring>   -> memory-path counters of one convolution shape
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1; export TMPDIR=/tmp
CFG="$1"; KN=$2
for set in "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_GATE_EN1 TCP_GATE_EN2 TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ" "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_TCC_READ_REQ_LATENCY" "SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD" "TD_TD_BUSY TD_TC_STALL TD_LOAD_WAVEFRONT TCP_TCR_TCP_STALL_CYCLES" "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_TCP_LATENCY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES"; do
  rm -rf gpurun_out/pmc_tmp
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 tools/conv_micro.py $CFG > /dev/null 2>&1
  f=$(find gpurun_out/pmc_tmp -name "*counter_collection.csv"); if [ -n "$f" ]; then python3 tools/pmc_kernel.py $f $KN; else echo "no output for: $set"; fi
done
rm -rf gpurun_out/pmc_tmp
