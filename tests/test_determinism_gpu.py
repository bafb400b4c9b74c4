"""GPU: a TRAINING step is bit-reproducible (SURVEY 5.2 "run each kernel twice, byte-compare").

The reference's BatchNorm is torch.nn.BatchNorm1d inside ME.MinkowskiBatchNorm (minsu3d/model/module/common.py:35-39):
the same input gives the same statistics.  Rounds 1-4 summed the training-mode statistics with LDS float atomics and
handed the pair-list tiles to the waves off a counter, so two evaluations of one step differed in the last bits, a ReLU /
max-pool mask flipped now and then, and gradient checks had to tolerate 5e-2.  Since round 5 every float sum of a step has
a fixed order (per-wave statistic slots combined in wave order, a fixed tile schedule in the pair-list kernels, register
sums + ordered LDS combination in the stand-alone BatchNorm passes, the row scatter-adds over a stable sort): the same
step twice gives the same BYTES -- losses, every parameter gradient, the running statistics."""
import copy

import pytest
import torch

from test_model_cpu import _build, small_batch

pytestmark = pytest.mark.gpu


def _cuda(batch):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}


def _one_step(model, batch):
    """forward + losses + backward on a fresh copy of the module state -> everything a step produces"""
    m = copy.deepcopy(model)
    m.voxelization_rand = model.voxelization_rand
    m.train()
    out = m(batch)
    losses = m._loss(batch, out)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    res = {"loss/" + k: v.detach().clone() for k, v in losses.items()}
    for k in ("point_features", "semantic_scores", "point_offsets"):
        res["out/" + k] = out[k].detach().clone()
    res.update({"grad/" + n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    res.update({"buf/" + n: b.detach().clone() for n, b in m.named_buffers()})
    return res


def _diff_report(a, b, limit=12):
    bad = []
    for k in a:
        if not torch.equal(a[k], b[k]):
            x, y = a[k].double(), b[k].double()
            bad.append((float((x - y).abs().max() / y.abs().max().clamp_min(1e-30)), k))
    bad.sort(reverse=True)
    return len(bad), [f"{k}: {e:.1e}" for e, k in bad[:limit]]


def _softgroup_scores(batch):
    n = batch["point_xyz"].size(0)
    sem = torch.full((n, 20), 0.01)
    sem[torch.arange(n), batch["grouping_semantic_preds"].long()] = 0.8
    batch["grouping_semantic_scores"] = sem
    return batch


@pytest.mark.parametrize("name", ["pointgroup", "hais", "softgroup"])
def test_training_step_twice_gives_the_same_bytes(name):
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    be = HipBackend()
    assert be.deterministic()
    backend.set_backend(be)
    model = _build(name, seed=4)
    model.hparams.cfg.data.point_num_avg = [-1, -1] + [400.0] * 18
    model.hparams.cfg.data.radius_avg = [-1.0, -1.0] + [0.3] * 18
    model = model.cuda()
    model.voxelization_rand = (torch.tensor([0.3, 0.6, 0.9]).cuda(), torch.tensor([0.1, 0.2, 0.3]).cuda())
    batch = small_batch((41, 42))
    if name == "softgroup":
        batch = _softgroup_scores(batch)
    batch = _cuda(batch)
    first = _one_step(model, batch)
    assert len([k for k in first if k.startswith("grad/")]) > 100
    for _ in range(2):
        again = _one_step(model, batch)
        assert again.keys() == first.keys()
        n_bad, worst = _diff_report(again, first)
        assert n_bad == 0, (n_bad, worst)


@pytest.mark.parametrize("name", ["pointgroup", "hais", "softgroup"])
def test_training_step_at_bench_size_twice_gives_the_same_bytes(name):
    """the same at BASELINE's sizes (4 x ~144k points, m = 16 / 32): the pair-list / pair-stream / bf16x3 kernels, the
    deferred and batched backward-weight launches, the prefetched coordinate structures"""
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    from minsu3d_amd.config import load_config
    import bench
    backend.set_backend(HipBackend())
    dev = torch.device("cuda", 0)
    cfg = load_config([f"model={name}", "data=scannetv2"])
    model = bench.build(cfg, dev)
    model.voxelization_rand = (torch.tensor([0.3, 0.6, 0.9], device=dev), torch.tensor([0.1, 0.2, 0.3], device=dev))
    batch = bench.make_batch([0, 1, 2, 3], dev)
    first = _one_step(model, batch)
    again = _one_step(model, batch)
    n_bad, worst = _diff_report(again, first)
    assert n_bad == 0, (n_bad, worst)


@pytest.mark.parametrize("n,n_dst,C_", [(200000, 60000, 16), (50000, 300, 32), (1000, 1000, 7), (5, 3, 16)])
def test_scatter_add_rows_fixed_order(n, n_dst, C_):
    """ms3d_scatter_add_rows_sorted: the backward of the many-to-one row gathers as a fixed-order sum -- equal to a serial
    float32 accumulation in ascending source order BIT FOR BIT (checked on a slice), to index_add within rounding, and
    identical when repeated; rows nobody names stay zero"""
    from minsu3d_amd.backend import HipBackend
    be = HipBackend()
    g = torch.Generator().manual_seed(n)
    idx = torch.randint(0, n_dst, (n,), generator=g)
    idx[idx == 1] = 0                                   # row 1 has no source
    src = torch.randn(n, C_, generator=g)
    got = be.scatter_add_rows(src.cuda(), idx.cuda(), n_dst)
    assert torch.equal(got, be.scatter_add_rows(src.cuda(), idx.cuda(), n_dst))
    want = torch.zeros(n_dst, C_, dtype=torch.float64).index_add_(0, idx, src.double())
    assert (got.cpu().double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    if n_dst > 1:
        assert float(got[1].abs().max()) == 0.0
    rows = torch.unique(idx)[:50]
    for r in rows.tolist():
        acc = torch.zeros(C_)
        for i in torch.nonzero(idx == r).view(-1).tolist():     # ascending source row
            acc = acc + src[i]
        assert torch.equal(got[r].cpu(), acc), r
