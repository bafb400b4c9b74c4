#!/bin/bash
# determinism status on the GPU box: the new tests (report what differs), then sort cost and a bench A/B of the tile schedule
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_determinism_gpu.py -q -p no:cacheprovider --timeout 900 2>&1 | tail -40 > gpurun_out/r5_det.log
python - > gpurun_out/r5_sort.log 2>&1 <<'PY'
import torch, time
for n, hi in ((575000, 430000), (800000, 120000), (800000, 575000)):
    idx = torch.randint(0, hi, (n,), device="cuda")
    for dt in (torch.int64, torch.int32):
        k = idx.to(dt)
        for stable in (True, False):
            torch.cuda.synchronize()
            for _ in range(3): torch.sort(k, stable=stable)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): torch.sort(k, stable=stable)
            torch.cuda.synchronize()
            print(n, hi, dt, "stable" if stable else "unstable", "%.1f us" % ((time.perf_counter() - t0) / 20 * 1e6))
PY
for dyn in 0 1 0 1; do
  MS3D_PL_DYNAMIC=$dyn python bench.py --steps 40 --warmup 8 --no-cpu-baseline --also none 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('dyn=$dyn', d['value'], d['step_ms'], d['roofline']['frac'])" >> gpurun_out/r5_ab.log 2>&1
done
cat gpurun_out/r5_det.log gpurun_out/r5_sort.log gpurun_out/r5_ab.log
