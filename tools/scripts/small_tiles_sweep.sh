# direct-B split-K kernel vs LDS-staged table walk at mid levels (tools/conv_micro.py)
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
export MS3D_PAIRSTREAM=0
for cfg in "48 48 27 2" "96 96 27 2" "80 80 27 2" "128 128 27 2" "64 96 8 1" "96 64 8 1" "64 64 27 1"; do
  for st in 1100 4000 20000; do
    MS3D_SMALL_TILES=$st python3 tools/conv_micro.py $cfg 2>&1 | tail -1 | cut -c1-150
  done
done
