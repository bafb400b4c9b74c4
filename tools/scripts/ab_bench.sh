# usage: bash tools/scripts/ab_bench.sh "<ENV=a ...>" "<ENV=b ...>" [reps=3] [model=pointgroup]
# A/B of two environments on ONE box, runs interleaved (boxes differ by +-3 %, a box drifts by +-0.5 ms between runs):
# per run the MEDIAN step of 60 (HIP events at the step boundaries), per variant the median and the spread of the runs
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1; export TMPDIR=/tmp
A="$1"; B="$2"; REPS=${3:-3}; M=${4:-pointgroup}
for i in $(seq $REPS); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    env $E python3 bench.py --model $M --no-cpu-baseline --no-roofline --steps 60 --warmup 10 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['step_ms']['median'], d['step_ms']['min'], d['ms_per_step'], d['value'])"
  done
done | tee /tmp/ab.txt
python3 - <<'PY'
import statistics as st
rows = [l.split() for l in open('/tmp/ab.txt')]
for v in 'AB':
    med = [float(r[1]) for r in rows if r[0] == v]; mn = [float(r[2]) for r in rows if r[0] == v]
    print(v, 'median-of-medians %.3f ms (runs %s)  best min %.3f' % (st.median(med), ' '.join('%.2f' % m for m in med), min(mn)))
PY
