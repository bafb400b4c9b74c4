"""Kernels between the end of the grouping (last bfs_emit_kernel) and the start of the backward pass (first softmax
backward kernel) of the LAST step in a rocprofv3 kernel trace, in launch order with the idle gap in front of each:
the host-bound tail of the forward pass.   usage: python tools/tail_kernels.py <kernel_trace.csv>"""
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'[<(].*', '', n)[-46:]
end = max(i for i, r in enumerate(rows) if "softmax_warp_backward" in r["Kernel_Name"] or "log_softmax_backward" in r["Kernel_Name"].lower())
start = max(i for i, r in enumerate(rows[:end]) if "bfs_emit_kernel" in r["Kernel_Name"])
seg = rows[start:end + 1]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
print(f"{len(seg)} launches, kernel time {busy:.0f} us, span {span:.0f} us (profiled run)")
prev = int(seg[0]["End_Timestamp"])
for r in seg[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  gap {max(s - prev, 0) / 1e3:7.1f}  run {(e - s) / 1e3:7.1f}  {short(r['Kernel_Name'])}")
    prev = max(prev, e)
