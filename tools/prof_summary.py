"""Group a rocprofv3 kernel_stats.csv by kernel (template arguments folded) -> ms per step.
usage: python tools/prof_summary.py <kernel_stats.csv> <steps incl. warmup> [top]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    key = re.sub(r'[<(].*', '', n)[-48:]
    agg[key][0] += int(r['Calls']); agg[key][1] += float(r['TotalDurationNs'])
print("total ms/step %.2f  launches/step %.0f" % (sum(v[1] for v in agg.values()) / steps / 1e6, sum(v[0] for v in agg.values()) / steps))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k:48s} {v[0] / steps:7.1f}/step {v[1] / steps / 1e6:7.3f} ms  {v[1] / v[0] / 1e3:7.1f} us")
