"""ME.utils.sparse_quantize / sparse_collate as the reference calls them
(data/dataset/general_dataset.py:159-163, data/data_module.py:94-96, model/general_model.py:187-189)."""
import numpy as np
import torch

from ..backend import get_backend


def sparse_quantize(coordinates, features=None, return_index=False, return_inverse=False, quantization_size=None,
                    device="cpu"):
    """floor(coordinates / quantization_size) -> unique integer rows, FIRST occurrence wins, rows in
    first-occurrence order (canonical choice on every device).
    coordinates: [N, D] (D = 3, or 4 with a leading batch/cluster column), tensor or ndarray."""
    was_numpy = isinstance(coordinates, np.ndarray)
    c = torch.from_numpy(coordinates) if was_numpy else coordinates
    if quantization_size is not None:
        c = torch.floor(c / quantization_size)
    c = c.to(torch.int32)
    c4 = c if c.size(1) == 4 else torch.cat([torch.zeros((c.size(0), 1), dtype=torch.int32, device=c.device), c], 1)
    uniq, inv = get_backend().sparse_quantize(c4.contiguous())
    uniq_l = uniq.long()
    out = [c[uniq_l]]
    if features is not None:
        f = torch.from_numpy(features) if isinstance(features, np.ndarray) else features
        if torch.is_tensor(f) and f.is_cuda and f.dim() == 2 and f.dtype == torch.float32 and f.requires_grad:
            # the first-occurrence rows: a gather over UNIQUE indices -- the engine's row gather, whose backward is one
            # scatter launch (no two sources per row: order-independent) where torch's indexing backward sorts the index
            from . import functional as Fn
            out.append(Fn.gather_rows(f, uniq_l, max_dup=1))
        else:
            out.append(f[uniq_l.to(f.device)])
    if return_index:
        out.append(uniq_l)
    if return_inverse:
        out.append(inv.long())
    if was_numpy:
        out = [o.cpu() for o in out]
    return out[0] if len(out) == 1 else tuple(out)


def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
    """prepend the batch index column and concatenate: -> (bcoords i32 [sum V, 4], feats f32)"""
    bc, bf = [], []
    for b, (c, f) in enumerate(zip(coords, feats)):
        c = torch.as_tensor(c).to(dtype)
        bc.append(torch.cat([torch.full((c.size(0), 1), b, dtype=dtype, device=c.device), c], 1))
        bf.append(torch.as_tensor(f, dtype=torch.float32))
    bc, bf = torch.cat(bc, 0), torch.cat(bf, 0)
    if device is not None:
        bc, bf = bc.to(device), bf.to(device)
    return bc, bf
