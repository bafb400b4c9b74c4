"""Micro-timing of one conv configuration on the bench's level-0 (or level-1) geometry.
usage: python tools/conv_micro.py [cin cout K level]   (env knobs of the library apply)"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd import backend as B
from minsu3d_amd.MinkowskiEngine.tensor import CoordinateManager

# K = 8u: the one-hot `up` table of a stride-2 map (transposed convolution forward, strided convolution backward-data):
# output = the FINE rows of `level`, input = the coarse rows of level + 1
UP = len(sys.argv) > 3 and sys.argv[3] == "8u"
if UP:
    sys.argv[3] = "8"
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
elif UP:
    _, nbr = cm.k2(ts); vin, vout = cm.size(2 * ts), cm.size(ts)
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
try:
    for _ in range(3):
        dW = be.conv_backward_weight(x, g, nbr, vout, K, cin, cout)
    e0.record()
    for _ in range(n):
        dW = be.conv_backward_weight(x, g, nbr, vout, K, cin, cout)
    e1.record(); torch.cuda.synchronize()
    us_w = e0.elapsed_time(e1) / n * 1e3
except Exception:        # shapes the backward-weight kernels do not serve (more than 14 output column blocks: a backward-data twin)
    us_w = float("nan")
# the layer entry point (what the modules call): weight images incl. the aux image, fused BatchNorm / ReLU prologue
scale = torch.rand(cin, device=dev) + 0.5; shift = torch.randn(cin, device=dev) * 0.2
yl, _, wfb = be.conv_layer_forward(x, W, nbr, vout, K, cin, cout, K == 27, (scale, shift), True, None, None, False)
for _ in range(3):
    be.conv_layer_forward(x, None, nbr, vout, K, cin, cout, K == 27, (scale, shift), True, None, None, False, wf_ready=wfb)
e0.record()
for _ in range(n):
    yl2, _, _ = be.conv_layer_forward(x, None, nbr, vout, K, cin, cout, K == 27, (scale, shift), True, None, None, False, wf_ready=wfb)
e1.record(); torch.cuda.synchronize()
us_l = e0.elapsed_time(e1) / n * 1e3
yref = be.conv_forward(x, wf, nbr, vout, K, cin, cout, pre=(scale, shift), pre_relu=True)
err = ((yl2.double() - yref.double()).abs().max() / yref.double().abs().max()).item()
alg = pairs * (cin + cout) * 4 + pairs * 8 + K * cin * cout * 4
print(f"cin={cin} cout={cout} K={K} vin={vin} vout={vout} pairs/row={pairs / vout:.2f}  fwd {us:.1f} us  {alg / us / 1e3:.0f} GB/s algorithmic | layer fwd {us_l:.1f} us (aux kind {be.lib.ms3d_spconv_aux_kind(K, cin, cout)}, vs f32 kernel {err:.1e}) | wgrad {us_w:.1f} us"
      f"  env={ {k: v for k, v in os.environ.items() if k.startswith('MS3D_')} }")
# the layer's backward entry point (backward-data with the fused BatchNorm-backward epilogue + its reduction chain +
# backward-weight): what the modules call; minus the stand-alone backward-weight time = the backward-data side
if K == 27:
    mean, invstd = torch.zeros(cin, device=dev), torch.ones(cin, device=dev)
    bn = dict(scale=scale, shift=shift, mean=mean, invstd=invstd, relu=True, training=True)
    for _ in range(3):
        be.conv_layer_backward(x, g, wfb, nbr, nbr, vin, vout, K, cin, cout, bn, True)
    e0.record()
    for _ in range(n):
        be.conv_layer_backward(x, g, wfb, nbr, nbr, vin, vout, K, cin, cout, bn, True)
    e1.record(); torch.cuda.synchronize()
    us_b = e0.elapsed_time(e1) / n * 1e3
    for _ in range(3):
        be.conv_backward_weight(x, g, nbr, vout, K, cin, cout, pre=(scale, shift), pre_relu=True)
    e0.record()
    for _ in range(n):
        be.conv_backward_weight(x, g, nbr, vout, K, cin, cout, pre=(scale, shift), pre_relu=True)
    e1.record(); torch.cuda.synchronize()
    us_w2 = e0.elapsed_time(e1) / n * 1e3
    print(f"layer bwd {us_b:.1f} us = backward-weight (with prologue) {us_w2:.1f} + backward-data side {us_b - us_w2:.1f}")
