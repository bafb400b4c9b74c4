cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
for cfg in "32 32 27 1" "32 32 27 0" "16 32 27 0" "32 16 27 0"; do
  echo "== $cfg"
  for mw in 8 5 4 3; do
    r=$(MS3D_PL_MIN_WAVES=$mw python3 tools/conv_micro.py $cfg 2>&1 | grep -oE "fwd [0-9.]+ us .*layer fwd [0-9.]+ us")
    echo "   min_waves=$mw : $r"
  done
done
