"""Kernel time by phase of the LAST complete training step in a rocprofv3 kernel trace: the step is cut at the Adam
kernel (multi_tensor_apply), the forward / backward boundary is the first softmax backward kernel.
usage: python tools/phase_kernels.py <kernel_trace.csv> [top]"""
import csv, sys, re, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'[<(].*', '', n)[-44:]
adam = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
# Adam = a burst of multi_tensor_apply launches; steps are separated by gaps in the index sequence
bursts = [adam[0]]
for a, b in zip(adam, adam[1:]):
    if b - a > 50: bursts.append(b)
lo, hi = bursts[-2], bursts[-1]
step = rows[lo:hi]
while step and "multi_tensor_apply" in step[0]["Kernel_Name"]: step = step[1:]
cut = next(i for i, r in enumerate(step) if "softmax_warp_backward" in r["Kernel_Name"] or "log_softmax_backward" in r["Kernel_Name"].lower())
for name, part in (("forward (+loss)", step[:cut]), ("backward", step[cut:])):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in part:
        a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(part[-1]["End_Timestamp"]) - int(part[0]["Start_Timestamp"])) / 1e3
    print(f"== {name}: {len(part)} launches, kernel time {sum(v[1] for v in agg.values()) / 1e3:.2f} ms, span {span / 1e3:.2f} ms (profiled)")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"   {k:44s} {n:4d}  {us / 1e3:7.3f} ms  {us / n:7.1f} us")
