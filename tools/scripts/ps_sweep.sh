cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
for cfg in "32 64 27 1" "64 32 27 1" "48 96 27 2" "96 48 27 2" "32 48 8 1"; do
  echo "== $cfg"
  for v in "2 8" "3 8" "4 8" "3 6" "4 6" "4 5" "1 8"; do set -- $v
    r=$(MS3D_PS_NBT=$1 MS3D_PS_W=$2 python3 tools/conv_micro.py $cfg 2>&1 | grep -oE "layer fwd [0-9.]+ us|backward-data side [0-9.]+" | tr '\n' ' ')
    echo "   nbt<=$1 waves<=$2: $r"
  done
done
