"""Where does the engine's backward pass leave float64?  The two-level U-Net of tests/test_fullsize_gpu.py
(test_unet_gradients_vs_fp64_on_a_full_scene) with the gradient of the loss w.r.t. EVERY residual block's output
compared between the engine and float64 autograd, walking from the loss backwards.  usage: python tools/grad_bisect.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_fullsize_gpu as T
from minsu3d_amd import backend
from minsu3d_amd.backend import HipBackend
from minsu3d_amd.data import synthetic
from minsu3d_amd.model.module import Backbone
from minsu3d_amd.model.module.common import ResidualBlock
import minsu3d_amd.MinkowskiEngine as ME

backend.set_backend(HipBackend())
dev = torch.device("cuda", 0)
torch.manual_seed(11)
net = Backbone(input_channel=6, output_channel=16, block_channels=[1, 2], block_reps=2, sem_classes=20).to(dev).train()
unet = net.unet
if "--no-skip-fusion" in sys.argv:
    ResidualBlock.fuse_skip_grad = False
with torch.no_grad():
    for n_, p_ in unet.named_parameters():
        if n_.endswith("bn.weight"): p_.uniform_(0.6, 1.4)
        elif n_.endswith("bn.bias"): p_.uniform_(-0.3, 0.3)
b = synthetic.to_torch(synthetic.collate([synthetic.make_scene(6)]), dev)
x = ME.SparseTensor(features=b["voxel_features"], coordinates=b["voxel_xyz"])
cm = x.coordinate_manager; cm.prepare(2)
R = torch.randn(b["voxel_xyz"].size(0), 16, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
got_g, names = {}, []
def fwd_hook(name):
    def h(mod, inp, out):
        names.append(name)
        out._F.register_hook(lambda g, name=name: got_g.__setitem__(name, g.detach().clone()))
    return h
hooks = [m.register_forward_hook(fwd_hook(n)) for n, m in unet.named_modules() if isinstance(m, ResidualBlock)]
ME.prepare_conv_weights(net)
y = unet(x)
ME.release_conv_weights()
(y._raw() * R).sum().backward()
got_p = {n_: p_.grad.detach().clone() for n_, p_ in unet.named_parameters()}
unet.zero_grad(set_to_none=True)
# float64 reference with the block outputs retained
ref_acts = []
orig_block = T._ref_block
def rec_block(h, blk, nbr, cast=None):
    o = orig_block(h, blk, nbr, cast)
    o.retain_grad(); ref_acts.append(o)
    return o
T._ref_block = rec_block
acts = []
h = T._ref_conv(x._raw().detach().double(), unet[0].kernel.double(), cm.k3(1))
h = T._ref_ublock(h, unet[1], cm, 1, acts)
want_y = T._ref_bn_relu(h, unet[2])
(want_y * R.double()).sum().backward()
assert len(ref_acts) == len(names), (len(ref_acts), names)
print("gradient w.r.t. each residual block's OUTPUT (forward order), engine vs float64: max|d| / max|w|")
for name, a in zip(names, ref_acts):
    w = a.grad; g = got_g[name].double()
    d = (g - w).abs()
    rowerr = d.max(1).values / w.abs().max()
    bad = (rowerr > 1e-4).nonzero().view(-1)
    print(f"  {name:28s} {tuple(w.shape)}  {(d.max() / w.abs().max()).item():.2e}   rows off by > 1e-4: {bad.numel()}"
          f"  first {bad[:8].tolist()} last {bad[-4:].tolist()}  median row err {rowerr.median().item():.1e}")
    if bad.numel() and "--rows" in sys.argv:
        r = int(bad[0]); print("     row", r, "engine", g[r, :6].tolist(), "f64", w[r, :6].tolist())
print("parameter gradients:")
for n_, p_ in unet.named_parameters():
    w = p_.grad.double(); g = got_p[n_].double()
    print(f"  {n_:50s} {((g - w).abs().max() / w.abs().max()).item():.2e}")
