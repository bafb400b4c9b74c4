"""us per launch of the K = 1 convolutions (1x1 projections, per-point Linear layers) through ms3d_spconv_forward.
usage: MS3D_K1_PATH=0|1|2|3 python tools/k1_micro.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from minsu3d_amd import backend as B
be = B.get_backend()
dev = torch.device("cuda", 0)
shapes = [(64, 128, 200697), (128, 64, 200697), (32, 64, 426882), (64, 32, 426882), (96, 192, 51567), (192, 96, 51567),
          (128, 256, 12063), (256, 128, 12063), (160, 320, 2591), (320, 160, 2591), (384, 192, 509), (16, 16, 575000),
          (16, 20, 575000), (32, 16, 426882), (16, 32, 426882)]
out = []
for cin, cout, v in shapes:
    x = torch.randn(v, cin, device=dev); W = torch.randn(1, cin, cout, device=dev) * 0.05
    wf = be.prep_weights(W, 1, cin, cout); nbr = be.identity_table(v, dev)
    for _ in range(3):
        y = be.conv_forward(x, wf, nbr, v, 1, cin, cout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = be.conv_forward(x, wf, nbr, v, 1, cin, cout)
    e1.record(); torch.cuda.synchronize()
    want = x @ W[0]
    err = float((y - want).abs().max() / want.abs().max())
    out.append("%d->%d@%d %.1f us (err %.0e)" % (cin, cout, v, e0.elapsed_time(e1) / 20 * 1e3, err))
print("MS3D_K1_PATH=" + os.environ.get("MS3D_K1_PATH", "default"), " | ".join(out))
