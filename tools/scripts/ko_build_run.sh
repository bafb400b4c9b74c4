# usage: bash tools/scripts/ko_build_run.sh "<-D flags>" "<conv_micro cfg>" ...   (experiment build of the library on the box)
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
FLAGS="$1"; shift
# the variant builds into its own object directory / library file (minsu3d_amd/build.py); the product library is untouched
VARIANT=$(MS3D_EXTRA_HIPCC_FLAGS="$FLAGS" python3 -c "from minsu3d_amd import build; print(build.build())" 2> /dev/null | tail -1)
[ -f "$VARIANT" ] || echo BUILD FAILED
export MS3D_LIB="$VARIANT"
for cfg in "$@"; do python3 tools/conv_micro.py $cfg 2>&1 | tail -1 | grep -o "cin=[0-9]* cout=[0-9]* K=[0-9]* vin=[0-9]*\|layer fwd [0-9.]* us\|wgrad [0-9.]* us" | tr '\n' ' '; echo; done
