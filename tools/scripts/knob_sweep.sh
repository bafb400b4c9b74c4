# usage: bash tools/scripts/knob_sweep.sh [model]   -- a few launch-geometry knobs against the conv kernel time of a step
cd "${GRAFT_REPO_ROOT:?}" || exit 1
M=${1:-pointgroup}
for env in "X=0" "MS3D_PAIRLIST_MIN_ROWS=60000" "MS3D_PAIRLIST_MIN_ROWS=15000" "MS3D_SMALL_TILES=800" "MS3D_SMALL_TILES=1600" "MS3D_SMALL_TILES=3300" \
           "MS3D_WGRAD_MAX_CHUNKS=128" "MS3D_WGRAD_MAX_CHUNKS=512" "MS3D_PL_MIN_WAVES=6" "MS3D_PL_MIN_WAVES=12" "MS3D_WGRAD_F32_ROUNDS=2" "MS3D_SMALL_RT=1"; do
  env $env python3 bench.py --model $M --no-cpu-baseline --also none --steps 20 --warmup 5 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-32s value %.1f  median %.2f  conv %.3f ms  frac %.4f' % ('$env', d['value'], d['step_ms']['median'], r['kernel_ms_per_step'], r['frac']))"
done
