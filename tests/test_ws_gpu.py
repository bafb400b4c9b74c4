"""GPU: the weight-stationary split-K convolution route of the coarse levels (csrc/spconv.hip: spconv_fwd_ws_kernel; VERDICT
r5 #1) against the oracle, on every shape class it can serve -- the process runs with MS3D_WS_ALL=1 (the route's default
is the shapes it was measured to win on: K = 27 layers with a side beyond 256 channels at 1-3.5k rows), at the row counts of
the backbone's levels 4-6 (2.6k / 509 / 112 rows), through `_check_conv` (forward, fused BatchNorm / ReLU / residual,
epilogue statistics, backward-data with the fused BatchNorm-backward epilogue, the layer entry points) -- and twice: the
in-launch combine adds the offset groups' partial tiles in GROUP order, whichever workgroup arrives last, so two launches
give the same bytes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
from oracle import oracle as O
O.lib()
import test_sparse_gpu as T
from minsu3d_amd.backend import HipBackend
be = HipBackend()
cin, cout, K, npts, extent = %(shape)s
V = T._check_conv(be, O, cin, cout, K, npts, extent)
# the route ran (its geometry is what the partial-statistics sizing reports) and is bit-reproducible
rng = np.random.default_rng(1)
c = T.surface_coords(rng, 2, npts, extent)
nbr = T.dev(O.kmap_k3(c, 1).T.copy()) if K == 27 else None
if K == 27:
    x = torch.randn(c.shape[0], cin, device="cuda"); W = torch.randn(K, cin, cout, device="cuda") * 0.05
    wf = be.prep_weights(W, K, cin, cout)
    outs = []
    for rep in range(3):
        y, st = be.conv_forward(x, wf, nbr, c.shape[0], K, cin, cout, out_stats=True)
        outs.append((y.clone(), st.clone()))
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:]), "not bit-reproducible"
    # a second stream has its own slab area and counters
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        y2, _ = be.conv_forward(x, wf, nbr, c.shape[0], K, cin, cout, out_stats=True)
    y1, _ = be.conv_forward(x, wf, nbr, c.shape[0], K, cin, cout, out_stats=True)
    torch.cuda.synchronize()
    assert torch.equal(y1, outs[0][0]) and torch.equal(y2, outs[0][0])
print("WS_OK", V)
"""

# (cin, cout, K, points, extent): ~2.6k / ~500 / ~110 rows as levels 4-6 of the bench batch; square layers of both model
# widths, the 2c -> c layers behind the concatenations and their backward-data twins (inside _check_conv), K = 8
SHAPES = [(160, 160, 27, 2700, 60), (192, 192, 27, 520, 26), (224, 224, 27, 115, 12), (80, 80, 27, 2700, 60),
          (96, 96, 27, 520, 26), (112, 112, 27, 115, 12), (320, 160, 27, 2700, 60), (384, 192, 27, 520, 26),
          (160, 80, 27, 2700, 60), (160, 192, 8, 2700, 60), (48, 64, 27, 1500, 40)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dto%d_k%d_%dpts" % s[:4])
def test_weight_stationary_route_vs_oracle(shape):
    env = dict(os.environ, MS3D_WS_ALL="1")
    out = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT, "shape": repr(tuple(shape))}], env=env,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "WS_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_default_route_serves_the_wide_rectangular_layers(oracle):
    """without the switch: 320 -> 160 at ~2.6k rows (no three-piece bf16 image exists beyond 256 channels) takes the route
    in this process -- the statistics partials have one row per (row part, column slice) unit, not one per 16-row tile"""
    import torch
    from minsu3d_amd.backend import HipBackend
    import test_sparse_gpu as T
    be = HipBackend()
    V = T._check_conv(be, oracle, 320, 160, 27, 2700, 60)
    blocks_ws = be.lib.ms3d_spconv_partial_blocks(int(V), 27, 320, 160, 0)
    blocks_tiles = be.lib.ms3d_spconv_partial_blocks(int(V), 27, 160, 160, 0)
    if os.environ.get("MS3D_WS_MAX_TILES", "220") != "0" and os.environ.get("MS3D_WS_ALL", "0") != "1":
        assert blocks_ws < blocks_tiles
    torch.cuda.synchronize()


_STRESS = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
from oracle import oracle as O
O.lib()
import test_sparse_gpu as T
from minsu3d_amd.backend import HipBackend
be = HipBackend()
rng = np.random.default_rng(7)
ref = {}
shapes = [(160, 160, 2700, 60), (320, 160, 2700, 60), (96, 96, 520, 26), (224, 224, 115, 12)]
data = []
for cin, cout, npts, extent in shapes:
    c = T.surface_coords(rng, 2, npts, extent)
    nbr = T.dev(O.kmap_k3(c, 1).T.copy())
    x = torch.randn(c.shape[0], cin, device="cuda"); W = torch.randn(27, cin, cout, device="cuda") * 0.05
    res = torch.randn(c.shape[0], cout, device="cuda")
    wf = be.prep_weights(W, 27, cin, cout)
    y, st = be.conv_forward(x, wf, nbr, c.shape[0], 27, cin, cout, residual=res, out_stats=True)
    torch.cuda.synchronize()
    data.append((x, wf, nbr, c.shape[0], cin, cout, res, y.clone(), st.clone()))
# uneven load: a side stream keeps the chip (and the L2s) busy with streaming copies and matrix products of varying size
# while the split-K launches run back to back on two other streams; every word of every result is compared
side = torch.cuda.Stream(); s2 = torch.cuda.Stream()
big = torch.randn(64 << 20, device="cuda"); a = torch.randn(2048, 2048, device="cuda")
bad = 0
for it in range(40):
    with torch.cuda.stream(side):
        for k in range(1 + it %% 4):
            big2 = big[: (8 << 20) * (1 + (it + k) %% 7)].clone()
            a2 = a @ a
    outs = []
    for j, (x, wf, nbr, V, cin, cout, res, y0, st0) in enumerate(data):
        with torch.cuda.stream(s2 if (it + j) %% 2 else torch.cuda.current_stream()):
            outs.append(be.conv_forward(x, wf, nbr, V, 27, cin, cout, residual=res, out_stats=True))
    torch.cuda.synchronize()
    for (y, st), d in zip(outs, data):
        bad += int(not (torch.equal(y, d[7]) and torch.equal(st, d[8])))
assert bad == 0, bad
print("WS_STRESS_OK")
"""


def test_split_k_hand_off_under_uneven_load():
    """the in-launch combine (release fence -> ticket -> acquire -> plain loads) with the chip busy on other streams and the
    same launches alternating between two streams: 40 rounds x 4 shapes, every result word for word equal to the idle-chip
    result (a stale partial tile would show as a different sum)"""
    env = dict(os.environ, MS3D_WS_ALL="1")
    out = subprocess.run([sys.executable, "-c", _STRESS % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900,
                         cwd=ROOT)
    assert out.returncode == 0 and "WS_STRESS_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
