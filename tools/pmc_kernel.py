"""average of every collected counter for the kernels whose name contains a substring
usage: python tools/pmc_kernel.py <counter_collection.csv> <kernel substring>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:28s} {sum(v) / len(v):16.0f}  ({len(v)} launches)")
