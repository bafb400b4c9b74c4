"""stdin: the output of bench.py -> value, per-step clock summary and the stream plan of its JSON line (the last line that is JSON)"""
import json, sys
rows = [l for l in sys.stdin.read().splitlines() if l.startswith("{")]
if not rows:
    print("no JSON line"); sys.exit(1)
d = json.loads(rows[-1])
st = d["step_ms"]
print(f"{d['value']:8.2f} scenes/s  median {st['median']:.2f} min {st['min']:.2f} max {st['max']:.2f} ms   streams {d['config'].get('streams')}")
