# usage: bash tools/scripts/ws_micro.sh [extra env assignments for the WS runs, e.g. MS3D_WS_BLOCKS=512]
# coarse-level convolution shapes of PointGroup (m = 16) and HAIS / SoftGroup (m = 32) on the bench's level-4..6 tables:
# the weight-stationary route (default) against the one-tile kernels (MS3D_WS_MAX_TILES=0), one process per run
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
for cfg in "80 80 27 4" "96 96 27 5" "112 112 27 6" "160 160 27 4" "192 192 27 5" "224 224 27 6" "160 80 27 4" "320 160 27 4" \
           "384 192 27 5" "448 224 27 6" "80 96 8 4" "160 192 8 4" "192 160 8u 4"; do
  echo "== $cfg"
  MS3D_WS_MAX_TILES=0 python3 tools/conv_micro.py $cfg 2>&1 | grep -E "fwd|bwd" | sed 's/^/   off: /'
  env "$@" python3 tools/conv_micro.py $cfg 2>&1 | grep -E "fwd|bwd|Error|error" | sed 's/^/   ws : /'
done
