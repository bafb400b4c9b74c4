"""kernel-map build of the bench batch's levels in isolation: us per ms3d_kmap_k3 call (MS3D_KMAP_SYM=0|1)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd import backend as B
from minsu3d_amd.MinkowskiEngine.tensor import CoordinateManager
dev = torch.device("cuda", 0)
be = B.get_backend()
batch = bench.make_batch([0, 1, 2, 3], dev)
cm = CoordinateManager(batch["voxel_xyz"].int().contiguous(), spatial_sort=True)
cm.prepare(4)
ts = 1
for lvl in range(4):
    c = cm.coords[ts]
    for _ in range(3): be.kmap_k3(c, ts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): nbr = be.kmap_k3(c, ts)
    e1.record(); torch.cuda.synchronize()
    print(f"level {lvl}: rows {c.shape[0]}  kmap_k3 {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  pairs/row {(nbr >= 0).sum().item() / c.shape[0]:.2f}  sym={os.environ.get('MS3D_KMAP_SYM', '1')}")
    ts *= 2
