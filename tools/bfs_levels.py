"""Per-launch durations of the BFS kernels of the LAST step in a rocprofv3 kernel trace, in launch order.
usage: python tools/bfs_levels.py <kernel_trace.csv> [name substrings ...]"""
import csv, sys, re
pats = sys.argv[2:] or ["glob_", "dir_", "bfs_"]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(p in r["Kernel_Name"] for p in pats)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = everything after the second-to-last bfs_init
inits = [i for i, r in enumerate(rows) if "bfs_init" in r["Kernel_Name"]]
if len(inits) >= 2:
    rows = rows[inits[-2]:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = re.sub(r"[<(].*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {n}")
