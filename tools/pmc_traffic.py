"""HBM traffic per training step from two rocprofv3 counter passes of bench.py (FETCH_SIZE, WRITE_SIZE; separate runs as
the TCC has too few slots for both).  Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
both counters are in KB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide reads at 64 B -> fetch x 2; WRITE_SIZE
is taken as reported.
usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <steps incl. warm-up>
                                   <model> <commit> > profiles/r04_traffic_<model>.json"""
import csv, sys, json, re, collections


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            a = agg[re.sub(r"[<(].*", "", n)]; a[0] += 1; a[1] += float(r["Counter_Value"]) * 1024.0
    return agg


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
steps, model, commit = float(sys.argv[3]), sys.argv[4], sys.argv[5]
rows = []
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
    rows.append({"kernel": k, "launches_per_step": round(max(f[0], w[0]) / steps, 1),
                 "fetch_bytes_raw_per_step": int(f[1] / steps), "write_bytes_per_step": int(w[1] / steps),
                 "hbm_bytes_per_step": int((2.0 * f[1] + w[1]) / steps)})
rows.sort(key=lambda r: -r["hbm_bytes_per_step"])
conv = [r for r in rows if r["kernel"].startswith("spconv_")]
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps({
    "model": model, "commit": commit, "kernel_source_digest": bench.kernel_source_digest(),
    "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py "
              f"--model {model} --steps 2 --warmup 1 --no-cpu-baseline --no-roofline; bytes = (2 x FETCH_SIZE + WRITE_SIZE) "
              "x 1024 (KB units; gfx950 FETCH_SIZE counts 128-B requests of wide reads as 64 B), summed per kernel, "
              "divided by the 3 steps of the run",
    "spconv_hbm_bytes_per_step": sum(r["hbm_bytes_per_step"] for r in conv),
    "spconv_launches_per_step": round(sum(r["launches_per_step"] for r in conv), 1),
    "all_kernels_hbm_bytes_per_step": sum(r["hbm_bytes_per_step"] for r in rows),
    "by_kernel": rows[:40]}, indent=1))
