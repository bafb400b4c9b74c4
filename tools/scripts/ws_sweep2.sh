# weight-stationary route on the rectangular layers of the coarse levels (2c -> c behind a concatenation and its backward-data
# twin c -> 2c): exact-f32 forward / layer forward us, one-tile kernels (off) vs three WS geometries
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
for cfg in "320 160 27 4" "160 320 27 4" "384 192 27 5" "192 384 27 5" "448 224 27 6" "224 448 27 6" "160 80 27 4" "80 160 27 4" "192 96 27 5" "224 112 27 6" "256 128 27 3" "128 256 27 3"; do
  echo "== $cfg"
  MS3D_WS_MAX_TILES=0 python3 tools/conv_micro.py $cfg 2>&1 | grep -oE "fwd [0-9.]+ us .*layer fwd [0-9.]+ us" | sed 's/^/   off            : /'
  for geo in "16 128 1024" "16 128 512" "8 128 1024" "16 96 1024"; do set -- $geo
    r=$(MS3D_WS_MAX_TILES=1100 MS3D_WS_WAVES=$1 MS3D_WS_LDS_KB=$2 MS3D_WS_BLOCKS=$3 python3 tools/conv_micro.py $cfg 2>&1 | grep -oE "fwd [0-9.]+ us .*layer fwd [0-9.]+ us")
    echo "   w=$1 kb=$2 nb=$3 : $r"
  done
done
