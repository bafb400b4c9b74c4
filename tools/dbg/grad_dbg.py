import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from minsu3d_amd.data import synthetic
from minsu3d_amd.model.module import Backbone
import minsu3d_amd.MinkowskiEngine as ME
dev = torch.device("cuda", 0)
torch.manual_seed(11)
net = Backbone(input_channel=6, output_channel=16, block_channels=[1, 2], block_reps=2, sem_classes=20).to(dev).train()
unet = net.unet
b = synthetic.to_torch(synthetic.collate([synthetic.make_scene(6)]), dev)
x = ME.SparseTensor(features=b["voxel_features"], coordinates=b["voxel_xyz"])
x.coordinate_manager.prepare(2)
R = torch.randn(b["voxel_xyz"].size(0), 16, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
ME.prepare_conv_weights(net)
y = unet(x)
ME.release_conv_weights()
yr = y._raw()
(yr * R).sum().backward()
g = unet[2].bn.bias.grad
want = (R.double() * (yr.detach() > 0)).sum(0)
print("last bn.bias grad: engine", g[:4].tolist(), "manual", want[:4].tolist(), "rel", ((g.double() - want).abs().max() / want.abs().max()).item())
gw = unet[2].bn.weight.grad
# second run: identical?
unet.zero_grad(set_to_none=True)
x2 = ME.SparseTensor(features=b["voxel_features"], coordinates=b["voxel_xyz"])
ME.prepare_conv_weights(net); y2 = unet(x2); ME.release_conv_weights()
(y2._raw() * R).sum().backward()
print("repeat: bias grad diff", (unet[2].bn.bias.grad - g).abs().max().item(), "out diff", (y2._raw() - yr).abs().max().item())
