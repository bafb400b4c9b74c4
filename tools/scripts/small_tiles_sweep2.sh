cd "${GRAFT_REPO_ROOT:?}" || exit 1
for M in pointgroup hais; do
for t in 400 700 760 800 900 1020 1100; do
  MS3D_SMALL_TILES=$t python3 bench.py --model $M --no-cpu-baseline --also none --steps 20 --warmup 5 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$M SMALL_TILES=%-5s value %.1f  median %.2f  conv %.3f ms  frac %.4f' % ('$t', d['value'], d['step_ms']['median'], r['kernel_ms_per_step'], r['frac']))"
done; done
