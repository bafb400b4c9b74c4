cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
for cfg in "64 64 27 1" "48 48 27 2" "64 64 27 2" "48 48 27 1"; do
  echo "== $cfg"
  for v in "0 64" "1 128" "1 256" "0 64" "1 128"; do set -- $v
    r=$(MS3D_WGRAD_LIST_NCH2=$1 MS3D_WGRAD_LIST_WIDE_CHUNKS=$2 python3 tools/conv_micro.py $cfg 2>&1 | grep -oE "wgrad [0-9.]+ us")
    echo "   nch2=$1 parts=$2: $r"
  done
done
