cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -k "pair_compacted or skewed or pair_lists or adjoint" 2>&1 | tail -6
for cfg in "32 32 27 1" "32 32 27 0"; do
  echo "== $cfg"
  for nar in 0 1; do
    MS3D_PL_NARROW=$nar python3 tools/conv_micro.py $cfg 2>&1 | grep -E "fwd|bwd" | sed "s/^/   narrow=$nar: /" | cut -c1-230
  done
done
