"""GPU idle-gap analysis of a rocprofv3 kernel_trace.csv: python tools/gap_analysis.py <kernel_trace.csv> [min_gap_us]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'[<(].*', '', n)[-40:]
# take the last 60% of the trace (steady state)
t0 = ev[int(len(ev) * 0.4)][0]
ev = [e for e in ev if e[0] >= t0]
busy = sum(e[1] - e[0] for e in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.defaultdict(lambda: [0, 0.0])
tot_gap = 0.0
# kernels of several streams overlap: a gap is time with NO kernel running (after the latest end seen so far)
last_end, last_name = ev[0][1], ev[0][2]
for b in ev[1:]:
    g = (b[0] - last_end) / 1e3
    if g > 0: tot_gap += g
    if g >= thr:
        k = short(last_name) + " -> " + short(b[2])
        gaps[k][0] += 1; gaps[k][1] += g
    if b[1] > last_end: last_end, last_name = b[1], b[2]
print(f"span {span/1e6:.2f} ms  sum of kernel time {busy/1e6:.2f} ms  no-kernel-running {tot_gap/1e3:.2f} ms ({100*tot_gap*1e3/span:.1f}% of span)")
for k, (n, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{g/1e3:8.3f} ms  n={n:4d}  avg {g/n:8.1f} us  {k}")
