"""f2 measurement: elastic distortion of one ~150k-point scene (the reference's dataset applies it twice per sample,
general_dataset.py:108-111) -- HIP kernel vs the host restatement (numpy; the reference itself runs 18
scipy.ndimage.convolve calls + 3 RegularGridInterpolators per call on the host) -- and the GPU voxelisation of a
4-scene batch (sparse_quantize, done per sample in the reference's collate on DataLoader workers).

    python tools/augment_bench.py"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from minsu3d_amd import backend as B
from minsu3d_amd.util import transform as T
from minsu3d_amd.data import synthetic
import minsu3d_amd.MinkowskiEngine as ME

be = B.get_backend()
dev = torch.device("cuda", 0)
sc = synthetic.make_scene(0)
x = (sc["xyz"].astype(np.float64) / 0.02)        # voxel units, as the dataset scales before the distortion
for gran, mag in ((6 * (1 / 0.02) // 50, 40 * (1 / 0.02) / 50), (20 * (1 / 0.02) // 50, 160 * (1 / 0.02) / 50)):
    np.random.seed(1)
    noise = T.elastic_noise(x, gran)
    t = time.perf_counter()
    want = x + np.hstack([T.trilinear(T.blur_noise(n), gran, x)[:, None] for n in noise]) * mag
    t_host = time.perf_counter() - t
    xd = torch.from_numpy(x).to(dev); nd = torch.from_numpy(np.stack(noise)).to(dev)
    for _ in range(3):
        got = be.elastic(xd, nd, gran, mag)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        got = be.elastic(xd, nd, gran, mag)
    torch.cuda.synchronize(); t_gpu = (time.perf_counter() - t) / 20
    err = float(np.abs(got.cpu().numpy() - want).max())
    print(f"elastic gran={gran:.0f} mag={mag:.0f}: {x.shape[0]} points, noise grid {noise[0].shape}: host restatement "
          f"{1e3 * t_host:.1f} ms, HIP {1e3 * t_gpu:.3f} ms ({t_host / t_gpu:.0f}x), max |diff| {err:.1e} voxels")
scenes = [synthetic.make_scene(s) for s in range(4)]
coords = [torch.from_numpy(np.ascontiguousarray(s["xyz"])).to(dev) for s in scenes]
feats = [torch.from_numpy(np.ascontiguousarray(s["rgb"])).to(dev) for s in scenes]
def quantize_all():
    return [ME.utils.sparse_quantize(c, f, return_index=True, return_inverse=True, quantization_size=0.02, device="cuda")
            for c, f in zip(coords, feats)]
for _ in range(3): quantize_all()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): out = quantize_all()
torch.cuda.synchronize(); t_q = (time.perf_counter() - t) / 10
print(f"sparse_quantize of 4 scenes ({sum(c.shape[0] for c in coords)} points -> {sum(o[0].shape[0] for o in out)} voxels): "
      f"{1e3 * t_q:.2f} ms on the GPU")
