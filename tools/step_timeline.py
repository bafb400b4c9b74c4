"""One steady-state training step of a rocprofv3 kernel_trace.csv as a compact timeline: per phase (delimited by marker
kernels) the span, the kernel time per queue, and the time with no kernel running; and, for the grouping window (first
ball-query kernel .. first ScoreNet kernel), every kernel with queue, start offset, duration and the idle gap before it.
usage: python tools/step_timeline.py <kernel_trace.csv> [--window] [--step K]"""
import csv, sys, re, collections

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'[<(].*', '', n)[-44:]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?'),
             r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows)
# steps are delimited by the one-launch optimizer kernel
opt = [i for i, e in enumerate(ev) if e[2].startswith('adam_step_kernel')]
k = int(sys.argv[sys.argv.index('--step') + 1]) if '--step' in sys.argv else len(opt) - 3
lo, hi = opt[k], opt[k + 1]
step = ev[lo + 1:hi + 1]
t0 = step[0][0]
print(f"step {k}: {len(step)} launches, span {(step[-1][1] - t0) / 1e6:.3f} ms, sum of kernel time {sum(e[1] - e[0] for e in step) / 1e6:.3f} ms")
def first(pred, start=0):
    for i in range(start, len(step)):
        if pred(step[i][2]): return i
    return None
i_bq = first(lambda n: n.startswith('bq_'))
i_emit = max(i for i, e in enumerate(step) if e[2].startswith('bfs_emit') or e[2].startswith('ha_')) if i_bq is not None else None
marks = [("backbone forward", 0, i_bq)]
if i_bq is not None:
    # the ScoreNet starts with the proposal voxelisation (pv_*) after the last emit
    i_pv = first(lambda n: n.startswith('pv_'), i_emit) or i_emit + 1
    i_bwd = first(lambda n: 'wgrad' in n, i_pv)
    marks += [("grouping (first bq_ .. first pv_)", i_bq, i_pv), ("ScoreNet + losses (.. first wgrad)", i_pv, i_bwd),
              ("backward + optimizer", i_bwd, len(step))]
for name, a, b in marks:
    seg = step[a:b]
    if not seg: continue
    span = (step[b][0] if b < len(step) else seg[-1][1]) - seg[0][0]
    per_q = collections.defaultdict(float)
    for e in seg: per_q[e[3]] += (e[1] - e[0]) / 1e3
    idle, last = 0.0, seg[0][0]
    for e in seg:
        if e[0] > last: idle += e[0] - last
        last = max(last, e[1])
    print(f"{name:40s} span {span / 1e6:7.3f} ms  launches {len(seg):4d}  no-kernel {idle / 1e6:6.3f} ms  kernel ms by queue: "
          + ", ".join(f"q{q}={t / 1e3:.2f}" for q, t in sorted(per_q.items())))
if '--top' in sys.argv:
    for name, a, b in marks:
        agg = collections.defaultdict(lambda: [0, 0.0])
        for e in step[a:b]:
            g = agg[e[2]]; g[0] += 1; g[1] += (e[1] - e[0]) / 1e3
        print(f"== {name}: kernel time {sum(v[1] for v in agg.values()) / 1e3:.2f} ms")
        for k_, (n_, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
            print(f"   {k_:46s} {n_:4d}  {us / 1e3:7.3f} ms  {us / n_:7.1f} us")
if '--window' in sys.argv and i_bq is not None:
    a, b = max(i_bq - 3, 0), min(i_pv + 12, len(step))
    last = step[a][0]
    print(f"{'+us':>9s} {'dur us':>8s} {'gap us':>7s}  q  kernel")
    for e in step[a:b]:
        gap = (e[0] - last) / 1e3
        print(f"{(e[0] - step[i_bq][0]) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f} {gap if gap > 0 else 0:7.1f}  {e[3]:>2s} {e[2]}")
        last = max(last, e[1])
