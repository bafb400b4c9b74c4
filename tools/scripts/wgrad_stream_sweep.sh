# backward-weight on a second stream: 0 = off, 1 = joined per layer, 2 = joined at the end of the backward pass
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
for m in pointgroup hais; do
  for v in 0 1 2 0 1 2; do
    MS3D_WGRAD_STREAM=$v python3 bench.py --model $m --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', 'MS3D_WGRAD_STREAM=$v', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"
  done
done
