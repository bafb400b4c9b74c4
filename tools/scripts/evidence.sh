# usage: bash tools/scripts/evidence.sh <tag> <commit>   -> gpurun_out/<tag>_*  (bench JSON lines, kernel stats, PMC traffic, timelines)
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1; export TMPDIR=/tmp
TAG=$1; C=$2
for M in pointgroup hais softgroup; do
  bash tools/scripts/pmc_traffic.sh $M $C $TAG > /dev/null 2>&1
  cp gpurun_out/${TAG}_traffic_$M.json profiles/ 2>/dev/null   # bench.py reads roofline.traffic from there (this box only)
done
# the headline line three times (a shared host makes single runs jump by +-8 %): all three are kept, the one with the
# median value is THE line
rm -f gpurun_out/${TAG}_bench_pointgroup_runs.jsonl
for i in 1 2 3; do python3 bench.py --also none 2> gpurun_out/${TAG}_bench_pointgroup.err | tail -1 >> gpurun_out/${TAG}_bench_pointgroup_runs.jsonl; done
python3 - <<PY
import json
rows = [json.loads(l) for l in open("gpurun_out/${TAG}_bench_pointgroup_runs.jsonl")]
rows.sort(key=lambda d: d["value"])
open("gpurun_out/${TAG}_bench_pointgroup.json", "w").write(json.dumps(rows[len(rows) // 2]) + "\n")
print("pointgroup runs:", [r["value"] for r in rows])
PY
python3 bench.py --model hais --no-cpu-baseline > gpurun_out/${TAG}_bench_hais.json 2> /dev/null
python3 bench.py --model softgroup --no-cpu-baseline > gpurun_out/${TAG}_bench_softgroup.json 2> /dev/null
for M in pointgroup hais softgroup; do
  bash tools/scripts/prof_model.sh $M ${TAG}_$M > gpurun_out/${TAG}_${M}_prof_summary.txt 2>&1
  rm -rf gpurun_out/prof_${TAG}_$M
done
python3 tools/phase_timeline.py > gpurun_out/${TAG}_phase_timeline.txt 2>&1
python3 tools/phase_timeline.py --model hais >> gpurun_out/${TAG}_phase_timeline.txt 2>&1
python3 tools/phase_timeline.py --model softgroup >> gpurun_out/${TAG}_phase_timeline.txt 2>&1
cut -c1-400 gpurun_out/${TAG}_bench_*.json
