#!/bin/bash
# A/B of one environment switch on the headline bench: r5_ab.sh VAR "val_a val_b" [model] [steps]
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; VALS=$2; MODEL=${3:-pointgroup}; STEPS=${4:-40}
mkdir -p gpurun_out
for rep in 1 2 3; do
  for v in $VALS; do
    env $VAR=$v python bench.py --model $MODEL --steps $STEPS --warmup 8 --no-cpu-baseline --also none 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); tk = {r['name'].split(' ')[0] + (' span' if r['name'].startswith('grouping span') else ''): r['ms_per_step'] for r in d.get('top_kernels', [])}
print('$VAR=$v', '$MODEL', d['value'], d['value_median'], d['step_ms']['median'], d['step_ms']['min'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], {k: v for k, v in tk.items() if k in ('pg_bfs_cluster', 'ballquery_batch_p', 'grouping span')})"
  done
done
