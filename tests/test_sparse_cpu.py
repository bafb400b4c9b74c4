"""CPU: the sparse-engine oracle against dense torch convolutions (the only independent pin available for the
MinkowskiEngine semantics -- ME itself is not in the reference tree, parity UNPINNED), and the custom
autograd of minsu3d_amd.MinkowskiEngine (running on the oracle backend) against torch autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from sparse_ref import densify, random_sparse, ref_bn_relu, ref_conv


@pytest.fixture()
def cpu_backend():
    from minsu3d_amd import backend
    from oracle.oracle_backend import OracleBackend
    prev = backend.set_backend(OracleBackend())
    yield
    backend.set_backend(prev)


def test_sparse_quantize_first_occurrence(oracle):
    c = np.array([[0, 1, 1, 1], [0, 2, 2, 2], [0, 1, 1, 1], [1, 1, 1, 1], [0, 2, 2, 2], [0, 0, 0, 0]], np.int32)
    u, inv = oracle.sparse_quantize(c)
    assert u.tolist() == [0, 1, 3, 5] and inv.tolist() == [0, 1, 0, 2, 1, 3]


def test_k3_conv_vs_dense(oracle):
    rng = np.random.default_rng(0)
    B, grid, Cin, Cout = 2, 9, 5, 7
    coords, feats = random_sparse(rng, B, grid, 300, Cin)
    W = rng.standard_normal((27, Cin, Cout)).astype(np.float32)
    out = oracle.conv_fwd(feats, W, oracle.kmap_k3(coords, 1))
    # dense: weight [out, in, kx, ky, kz] with k = ix + 3*iy + 9*iz
    wd = torch.as_tensor(W).view(3, 3, 3, Cin, Cout).permute(4, 3, 2, 1, 0)  # [out,in,ix,iy,iz]; view is (iz,iy,ix)
    dense = F.conv3d(densify(coords, feats, B, grid), wd, padding=1)
    c = torch.as_tensor(coords).long()
    want = dense[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]].numpy()
    assert np.allclose(out, want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("ts", [1, 2])
def test_k2_down_and_up_vs_dense(oracle, ts):
    rng = np.random.default_rng(1)
    B, grid, Cin, Cout = 2, 8, 4, 6
    coords, feats = random_sparse(rng, B, grid, 200, Cin)
    coords_ts = coords.copy(); coords_ts[:, 1:] *= ts          # the same geometry living at tensor stride ts
    W = rng.standard_normal((8, Cin, Cout)).astype(np.float32)
    oc, parent, koff = oracle.downsample(coords_ts, ts)
    down, up = oracle.kmap_k2(parent, koff, oc.shape[0])
    assert (oc[:, 1:] % (2 * ts) == 0).all() and len(np.unique(oc, axis=0)) == oc.shape[0]
    out = oracle.conv_fwd(feats, W, down)
    wd = torch.as_tensor(W).view(2, 2, 2, Cin, Cout).permute(4, 3, 2, 1, 0)
    dense = F.conv3d(densify(coords, feats, B, grid), wd, stride=2)
    c = torch.as_tensor(oc).long()
    want = dense[c[:, 0], :, c[:, 1] // (2 * ts), c[:, 2] // (2 * ts), c[:, 3] // (2 * ts)].numpy()
    assert np.allclose(out, want, rtol=1e-4, atol=1e-4)
    # transposed conv back onto the cached fine coordinate set
    Wt = rng.standard_normal((8, Cout, Cin)).astype(np.float32)
    back = oracle.conv_fwd(out, Wt, up)
    wtd = torch.as_tensor(Wt).view(2, 2, 2, Cout, Cin).permute(3, 4, 2, 1, 0)  # conv_transpose3d weight [in,out,kx,ky,kz]
    dense_c = torch.zeros(B, Cout, grid // 2, grid // 2, grid // 2)
    dense_c[c[:, 0], :, c[:, 1] // (2 * ts), c[:, 2] // (2 * ts), c[:, 3] // (2 * ts)] = torch.as_tensor(out)
    dense_up = F.conv_transpose3d(dense_c, wtd, stride=2)
    cf = torch.as_tensor(coords).long()
    want = dense_up[cf[:, 0], :, cf[:, 1], cf[:, 2], cf[:, 3]].numpy()
    assert np.allclose(back, want, rtol=1e-4, atol=1e-4)


def test_conv_backward_oracle_vs_autograd(oracle):
    rng = np.random.default_rng(2)
    coords, feats = random_sparse(rng, 2, 8, 150, 5)
    W = rng.standard_normal((27, 5, 4)).astype(np.float32)
    nbr = oracle.kmap_k3(coords, 1)
    x = torch.tensor(feats, requires_grad=True); Wt = torch.tensor(W, requires_grad=True)
    y = ref_conv(x, Wt, torch.as_tensor(nbr.T.copy()))
    g = torch.randn_like(y)
    y.backward(g)
    assert np.allclose(oracle.conv_bwd_data(g.numpy(), W, nbr, 150), x.grad.numpy(), rtol=1e-4, atol=1e-4)
    assert np.allclose(oracle.conv_bwd_weight(feats, g.numpy(), nbr, 27), Wt.grad.numpy(), rtol=1e-4, atol=1e-4)


def _unet_like(ME, c=16):
    import torch.nn as nn
    torch.manual_seed(0)
    net = nn.ModuleDict(dict(
        conv0=ME.MinkowskiConvolution(6, c, kernel_size=3, dimension=3),
        bn1=ME.MinkowskiBatchNorm(c), conv1=ME.MinkowskiConvolution(c, c, kernel_size=3, dimension=3),
        bn2=ME.MinkowskiBatchNorm(c), down=ME.MinkowskiConvolution(c, 2 * c, kernel_size=2, stride=2, dimension=3),
        bn3=ME.MinkowskiBatchNorm(2 * c), mid=ME.MinkowskiConvolution(2 * c, 2 * c, kernel_size=3, dimension=3),
        bn4=ME.MinkowskiBatchNorm(2 * c), up=ME.MinkowskiConvolutionTranspose(2 * c, c, kernel_size=2, stride=2, dimension=3),
        lin=ME.MinkowskiConvolution(2 * c, c, kernel_size=1, dimension=3), bn5=ME.MinkowskiBatchNorm(c)))
    for m in net.values():
        if hasattr(m, "bn"):
            with torch.no_grad():
                m.bn.weight.uniform_(0.5, 1.5); m.bn.bias.uniform_(-0.3, 0.3)
    return net


def run_me_chain(ME, net, feats, coords):
    relu = ME.MinkowskiReLU(inplace=True)
    x = ME.SparseTensor(features=feats, coordinates=coords)
    h = net["conv0"](x)
    ident = h
    h2 = net["conv1"](relu(net["bn1"](h)))
    h2 += ident                                               # residual block shape
    d = net["down"](relu(net["bn2"](h2)))
    d = net["mid"](relu(net["bn3"](d)))
    u = net["up"](relu(net["bn4"](d)))
    cat = ME.cat(h2, u)
    o = net["lin"](cat)
    o = relu(net["bn5"](o))
    return o.features


def run_ref_chain(net, feats, cm):
    g = lambda m: (m.bn.weight, m.bn.bias)
    h = ref_conv(feats, net["conv0"].kernel, cm.k3(1))
    h2 = ref_conv(ref_bn_relu(h, *g(net["bn1"])), net["conv1"].kernel, cm.k3(1)) + h
    down, up = cm.k2(1)
    d = ref_conv(ref_bn_relu(h2, *g(net["bn2"])), net["down"].kernel, down)
    d = ref_conv(ref_bn_relu(d, *g(net["bn3"])), net["mid"].kernel, cm.k3(2))
    u = ref_conv(ref_bn_relu(d, *g(net["bn4"])), net["up"].kernel, up)
    o = torch.cat([h2, u], 1) @ net["lin"].kernel
    return ref_bn_relu(o, *g(net["bn5"]))


def test_me_modules_autograd_vs_torch(cpu_backend):
    """forward + every gradient of a miniature U-Net (submanifold, strided, transposed, 1x1, fused BN/ReLU,
    residual, concat) through the custom autograd == torch autograd of the plain restatement"""
    import minsu3d_amd.MinkowskiEngine as ME
    rng = np.random.default_rng(3)
    coords, feats = random_sparse(rng, 2, 8, 220, 6)
    net = _unet_like(ME)
    ft = torch.tensor(feats, requires_grad=True)
    out = run_me_chain(ME, net, ft, torch.as_tensor(coords))
    gout = torch.randn_like(out)
    out.backward(gout)
    got = {n: p.grad.clone() for n, p in net.named_parameters()}
    gx = ft.grad.clone()
    running = {n: b.clone() for n, b in net.named_buffers()}
    net.zero_grad(); ft.grad = None
    cm = ME.CoordinateManager(torch.as_tensor(coords))
    ref = run_ref_chain(net, ft, cm)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4)
    ref.backward(gout)
    assert torch.allclose(gx, ft.grad, rtol=1e-3, atol=1e-4)
    for n, p in net.named_parameters():
        assert torch.allclose(got[n], p.grad, rtol=1e-3, atol=2e-4), n
    # running statistics follow torch.nn.BatchNorm1d
    bn = torch.nn.BatchNorm1d(16)
    h = ref_conv(ft.detach(), net["conv0"].kernel.detach(), cm.k3(1))
    bn.train(); bn(h)
    assert torch.allclose(running["bn1.bn.running_mean"], bn.running_mean, atol=1e-5)
    assert torch.allclose(running["bn1.bn.running_var"], bn.running_var, rtol=1e-4, atol=1e-5)


def test_me_eval_mode_and_materialize(cpu_backend):
    import minsu3d_amd.MinkowskiEngine as ME
    rng = np.random.default_rng(4)
    coords, feats = random_sparse(rng, 1, 6, 90, 6)
    net = _unet_like(ME)
    run_me_chain(ME, net, torch.tensor(feats), torch.as_tensor(coords))   # one training pass fills running stats
    net.eval()
    out = run_me_chain(ME, net, torch.tensor(feats), torch.as_tensor(coords))
    x = ME.SparseTensor(torch.tensor(feats), torch.as_tensor(coords))
    h = net["conv0"](x)
    y = net["bn1"](h)
    want = F.batch_norm(h.features, net["bn1"].bn.running_mean, net["bn1"].bn.running_var, net["bn1"].bn.weight,
                        net["bn1"].bn.bias, False)
    assert torch.allclose(y.features, want, rtol=1e-5, atol=1e-5) and torch.isfinite(out).all()


def test_sparse_quantize_api(cpu_backend):
    import minsu3d_amd.MinkowskiEngine as ME
    xyz = np.array([[0.011, 0.0, 0.05], [0.012, 0.001, 0.051], [0.5, 0.5, 0.5], [0.019, 0.019, 0.059]], np.float32)
    feats = np.arange(8, dtype=np.float32).reshape(4, 2)
    c, f, idx, inv = ME.utils.sparse_quantize(xyz, feats, return_index=True, return_inverse=True,
                                              quantization_size=0.02)
    assert c.tolist() == [[0, 0, 2], [25, 25, 25]] and idx.tolist() == [0, 2] and inv.tolist() == [0, 0, 1, 0]
    assert f.tolist() == [[0, 1], [4, 5]]
    bc, bf = ME.utils.sparse_collate([c, c], [f, f])
    assert bc.shape == (4, 4) and bc[:, 0].tolist() == [0, 0, 1, 1] and bc.dtype == torch.int32
