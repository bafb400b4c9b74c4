"""Per-grid launch statistics of one kernel from a rocprofv3 kernel trace.
usage: python tools/kernel_by_grid.py <kernel_trace.csv> <kernel substring> [min grid threads of the 'large' class]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2]
big = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
by = collections.defaultdict(list)
for r in rows:
    if sub in r["Kernel_Name"]:
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1)
        by[(g, int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
large = [d for (g, _), v in by.items() if g > big for d in v]
if large:
    print(f"{sub} launches with grid > {big} threads: {len(large)} launches, avg {sum(large) / len(large):.2f} us "
          f"(rocprofv3 --kernel-trace, same command as bench)")
for (g, wg), v in sorted(by.items(), key=lambda kv: -len(kv[1]) * 1e6 - kv[0][0]):
    print(f"grid {g} wg {wg} launches {len(v)} avg_us {sum(v) / len(v):.2f}")
