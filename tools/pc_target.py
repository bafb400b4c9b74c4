"""Many launches of ONE convolution forward (for PC sampling / counters): python tools/pc_target.py [cin cout K level n]"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd import backend as B
from minsu3d_amd.MinkowskiEngine.tensor import CoordinateManager
cin, cout, K, level, n = (int(a) for a in (sys.argv[1:6] + ["16", "16", "27", "0", "300"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0); be = B.get_backend()
batch = bench.make_batch([0, 1, 2, 3], dev)
cm = CoordinateManager(batch["voxel_xyz"].int().contiguous(), spatial_sort=True)
ts = 1
for _ in range(level):
    cm.k2(ts); ts *= 2
nbr = cm.k3(ts); v = cm.size(ts)
x = torch.randn(v, cin, device=dev); W = torch.randn(K, cin, cout, device=dev) * 0.05
wf = be.prep_weights(W, K, cin, cout)
for _ in range(n):
    y = be.conv_forward(x, wf, nbr, v, K, cin, cout)
torch.cuda.synchronize()
print("done", v)
