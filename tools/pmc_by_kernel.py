"""rocprofv3 --pmc counter_collection.csv -> average of one counter per (kernel, grid).
usage: python tools/pmc_by_kernel.py <counter_collection.csv> <COUNTER> > out.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ctr = sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r["Counter_Name"] == ctr:
        a = agg[(r["Kernel_Name"][:70], int(r["Grid_Size"]))]; a[0] += 1; a[1] += float(r["Counter_Value"])
print(f"kernel,grid,launches,avg_{ctr}")
for (k, g), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'"{k}",{g},{n},{v / n}')
