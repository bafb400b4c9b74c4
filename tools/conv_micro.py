"""Micro-timing of one conv configuration on the bench's level-0 (or level-1) geometry.
usage: python tools/conv_micro.py [cin cout K level]   (env knobs of the library apply)"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd import backend as B
from minsu3d_amd.MinkowskiEngine.tensor import CoordinateManager

cin, cout, K, level = (int(a) for a in (sys.argv[1:5] + ["16", "16", "27", "0"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
be = B.get_backend()
batch = bench.make_batch([0, 1, 2, 3], dev)
cm = CoordinateManager(batch["voxel_xyz"].int().contiguous(), spatial_sort=True)
ts = 1
for _ in range(level):
    cm.k2(ts); ts *= 2
if K == 27:
    nbr = cm.k3(ts); vin = vout = cm.size(ts)
else:
    nbr, _ = cm.k2(ts); vin, vout = cm.size(ts), cm.size(2 * ts)
pairs = int((nbr >= 0).sum())
x = torch.randn(vin, cin, device=dev)
W = torch.randn(K, cin, cout, device=dev) * 0.05
wf = be.prep_weights(W, K, cin, cout)
for _ in range(3):
    y = be.conv_forward(x, wf, nbr, vout, K, cin, cout)
torch.cuda.synchronize()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    y = be.conv_forward(x, wf, nbr, vout, K, cin, cout)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
g = torch.randn(vout, cout, device=dev)
for _ in range(3):
    dW = be.conv_backward_weight(x, g, nbr, vout, K, cin, cout)
e0.record()
for _ in range(n):
    dW = be.conv_backward_weight(x, g, nbr, vout, K, cin, cout)
e1.record(); torch.cuda.synchronize()
us_w = e0.elapsed_time(e1) / n * 1e3
alg = pairs * (cin + cout) * 4 + pairs * 8 + K * cin * cout * 4
print(f"cin={cin} cout={cout} K={K} vin={vin} vout={vout} pairs/row={pairs / vout:.2f}  fwd {us:.1f} us  {alg / us / 1e3:.0f} GB/s algorithmic | wgrad {us_w:.1f} us"
      f"  env={ {k: v for k, v in os.environ.items() if k.startswith('MS3D_')} }")
# layer-style launch (fused BN+ReLU prologue, residual, output statistics) timed by the in-library HIP events
from minsu3d_amd.backend import KernelTimer
tm = KernelTimer(lambda *a: True, be.lib, max_records=40); be.kernel_timer = tm
scale = torch.rand(cin, device=dev) + 0.5; shift = torch.randn(cin, device=dev) * 0.1
res = torch.randn(vout, cout, device=dev) if cin == cout else None
def layer(pre, r, stats, label):
    tm = KernelTimer(lambda *a: True, be.lib, max_records=40); be.kernel_timer = tm
    for _ in range(3):
        be.conv_layer_forward(x, W, nbr, vout, K, cin, cout, K == 27, pre, True, r, None, stats)
    tm.enabled = True
    for _ in range(20):
        be.conv_layer_forward(x, W, nbr, vout, K, cin, cout, K == 27, pre, True, r, None, stats)
    torch.cuda.synchronize()
    s = tm.summary()
    print(f"layer forward ({label}): events avg {s['avg_ms'] * 1e3:.1f} us over {s['launches']} launches")


layer((scale, shift), res, True, "BN+ReLU prologue, residual, stats")
if os.environ.get("CONV_MICRO_PARTS"):
    layer(None, None, False, "plain")
    layer((scale, shift), None, False, "prologue only")
    layer(None, res, False, "residual only")
    layer(None, None, True, "stats only")
