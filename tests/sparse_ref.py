"""Differentiable plain-torch restatement of the engine ops (gather - matmul - sum over the same neighbour
tables, F.batch_norm, relu): the fp32 reference the custom backward passes and the HIP kernels are checked
against.  Works on CPU and GPU tensors."""
import numpy as np
import torch
import torch.nn.functional as F


def ref_conv(x, W, nbr):
    """x [Vin,Cin], W [K,Cin,Cout], nbr [K,Vout] (-1 = none)"""
    xp = torch.cat([x, x.new_zeros(1, x.size(1))], 0)
    out = 0
    for k in range(nbr.size(0)):
        idx = nbr[k].long()
        idx = torch.where(idx < 0, torch.full_like(idx, x.size(0)), idx)
        out = out + xp[idx] @ W[k]
    return out


def ref_bn_relu(x, gamma, beta, relu=True, eps=1e-5):
    y = F.batch_norm(x, None, None, gamma, beta, True, 0.1, eps)
    return torch.relu(y) if relu else y


def random_sparse(rng, B=2, grid=10, n=260, C=6, even=True):
    """unique (b,x,y,z) rows in a small grid + random features"""
    pts = set()
    while len(pts) < n:
        pts.add((int(rng.integers(0, B)), int(rng.integers(0, grid)), int(rng.integers(0, grid)),
                 int(rng.integers(0, grid))))
    coords = np.array(sorted(pts), np.int32)
    rng.shuffle(coords)
    feats = rng.standard_normal((n, C)).astype(np.float32)
    return coords, feats


def densify(coords, feats, B, grid):
    d = torch.zeros(B, feats.shape[1], grid, grid, grid)
    c = torch.as_tensor(coords).long()
    d[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = torch.as_tensor(feats)
    return d
