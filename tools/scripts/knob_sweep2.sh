cd "${GRAFT_REPO_ROOT:?}" || exit 1
M=${1:-pointgroup}
for env in "X=0" "MS3D_PAIRSTREAM=2" "MS3D_PAIRSTREAM=3" "MS3D_PAIRSTREAM=0" "MS3D_BF16X3_WGRAD=0" "MS3D_WGRAD_WIDE_CHUNKS=64" "MS3D_WGRAD_LIST_WIDE_CHUNKS=128" "MS3D_STREAM_WAVES=8" "MS3D_PL_W=12"; do
  env $env python3 bench.py --model $M --no-cpu-baseline --also none --steps 20 --warmup 5 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-32s value %.1f  median %.2f  conv %.3f ms  frac %.4f' % ('$env', d['value'], d['step_ms']['median'], r['kernel_ms_per_step'], r['frac']))"
done
