cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -k "pair_compacted or bf16x3 or conv_kernels" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "float32_grade or backbone_forward" 2>&1 | tail -3
for cfg in "64 64 27 1" "48 48 27 2" "96 96 27 2" "128 64 27 1" "64 128 27 1" "128 128 27 2"; do
  echo "== $cfg"
  for v in old new old new; do
    MS3D_LIB=$PWD/build/variants/lib$v.so python3 tools/conv_micro.py $cfg 2>&1 | grep -oE "layer fwd [0-9.]+ us|backward-data side [0-9.]+" | tr '\n' ' ' | sed "s/^/   $v: /"; echo
  done
done
