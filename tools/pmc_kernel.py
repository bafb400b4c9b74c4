"""Average a rocprofv3 counter_collection.csv per kernel: python tools/pmc_kernel.py <csv> <kernel substring>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if sub in r['Kernel_Name']:
        a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in sorted(agg.items()):
    print(f"{k:28s} n={n:4d} avg={v / n:16.1f}")
