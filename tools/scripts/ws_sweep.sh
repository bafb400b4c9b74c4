# weight-stationary route: geometry sweep (waves per block x LDS budget x target blocks), exact-f32 forward us per shape
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
for cfg in "80 80 27 4" "160 160 27 4" "96 96 27 5" "192 192 27 5" "224 224 27 6" "320 160 27 4"; do
  echo "== $cfg"
  MS3D_WS_MAX_TILES=0 python3 tools/conv_micro.py $cfg 2>&1 | grep -oE "fwd [0-9.]+ us .*layer fwd [0-9.]+ us" | sed 's/^/   off            : /'
  for w in 4 8 16; do for kb in 32 64 128; do for nb in 512 1024; do
    r=$(MS3D_WS_WAVES=$w MS3D_WS_LDS_KB=$kb MS3D_WS_BLOCKS=$nb MS3D_WS_MIN_TPP=${MIN_TPP:-4} python3 tools/conv_micro.py $cfg 2>&1 | grep -oE "  fwd [0-9.]+ us")
    echo "   w=$w kb=$kb nb=$nb : $r"
  done; done; done
done
