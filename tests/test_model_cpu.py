"""CPU plumbing (BASELINE config 1 shape): PointGroup fwd + loss + bwd + Adam on small synthetic scenes with every
hot-path operator served by the oracle backend, and the world_size-2 gloo data-parallel path."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_batch(seeds=(0, 1), device="cpu"):
    from minsu3d_amd.data import synthetic as S
    scenes = [S.make_scene(s, room=(1.2, 1.0), n_boxes=2, density=900.0, wall_h=0.5) for s in seeds]
    b = S.to_torch(S.collate(scenes), device)
    rng = np.random.default_rng(seeds[0])
    # grouping inputs a trained network would produce: GT labels, offsets to the instance centre + 4 cm noise
    b["grouping_semantic_preds"] = torch.where(b["sem_labels"] >= 0, b["sem_labels"], torch.zeros_like(b["sem_labels"]))
    noise = torch.from_numpy(rng.normal(0, 0.04, tuple(b["point_xyz"].shape)).astype(np.float32)).to(device)
    b["grouping_point_offsets"] = torch.where((b["instance_ids"] >= 0)[:, None],
                                              b["instance_center_xyz"] - b["point_xyz"] + noise, torch.zeros_like(noise))
    return b


@pytest.fixture()
def cpu_backend():
    from minsu3d_amd import backend
    from oracle.oracle_backend import OracleBackend
    prev = backend.set_backend(OracleBackend())
    yield
    backend.set_backend(prev)


def build_model(seed=0):
    from minsu3d_amd.config import load_config
    from minsu3d_amd.model import PointGroup
    torch.manual_seed(seed)
    cfg = load_config(["model.network.blocks=[1,2,3]"])     # 3 levels keep the CPU test in seconds
    model = PointGroup(cfg)
    model.current_epoch = cfg.model.network.prepare_epochs + 1
    return model


def test_pointgroup_step_cpu(cpu_backend):
    model = build_model()
    batch = small_batch()
    opt = model.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and opt.param_groups[0]["lr"] == 0.002
    model.train()
    out = model(batch)
    scores, pidx, poff = out["proposal_scores"]
    assert poff.numel() - 1 >= 2 and scores.shape == (poff.numel() - 1, 1)
    assert pidx[:, 1].max() < batch["point_xyz"].size(0) and poff[-1] == pidx.size(0)
    losses = model._loss(batch, out)
    assert set(losses) == {"semantic_loss", "offset_norm_loss", "offset_dir_loss", "score_loss"}
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    named = dict(model.named_parameters())
    for n in ("backbone.unet.0.kernel", "backbone.unet.1.u.u.blocks.block0.conv_branch.2.kernel",
              "score_net.unet.0.blocks.block0.conv_branch.0.bn.weight", "score_branch.weight",
              "backbone.offset_branch.3.weight"):
        assert named[n].grad is not None and torch.isfinite(named[n].grad).all() and named[n].grad.abs().sum() > 0, n
    opt.step()
    # before prepare_epochs only the backbone runs
    model.current_epoch = 0
    assert "proposal_scores" not in model(batch)


def test_get_segmented_scores_and_offset_loss():
    from minsu3d_amd.loss import PTOffsetLoss
    from minsu3d_amd.model import get_segmented_scores
    s = torch.tensor([0.0, 0.25, 0.5, 0.75, 0.9, 0.1])
    assert torch.allclose(get_segmented_scores(s, 0.75, 0.25), torch.tensor([0.0, 0.0, 0.5, 1.0, 1.0, 0.0]))
    pred = torch.tensor([[1.0, 0, 0], [0, 2.0, 0], [5.0, 5, 5]]); gt = torch.tensor([[1.0, 0, 0], [0, -1.0, 0], [0.0, 0, 0]])
    n, d = PTOffsetLoss()(pred, gt, torch.tensor([True, True, False]))
    assert torch.isclose(n, torch.tensor(1.5)) and torch.isclose(d, torch.tensor(0.0))
    z = PTOffsetLoss()(pred, gt, torch.tensor([False, False, False]))
    assert float(z[0]) == 0.0 and float(z[1]) == 0.0


def test_clusters_voxelization_semantics(cpu_backend):
    from minsu3d_amd.model import clusters_voxelization
    rng = np.random.default_rng(0)
    N, m = 400, 4
    coords = torch.from_numpy(rng.random((N, 3)).astype(np.float32))
    feats = torch.from_numpy(rng.standard_normal((N, m)).astype(np.float32))
    idx = torch.stack([torch.cat([torch.zeros(150), torch.ones(100)]).long(),
                       torch.from_numpy(rng.permutation(N)[:250])], 1)
    off = torch.tensor([0, 150, 250], dtype=torch.int32)
    u = (torch.tensor([0.3, 0.6, 0.9]), torch.tensor([0.1, 0.2, 0.3]))
    vox, p2v = clusters_voxelization(idx, off, feats, coords, 50, 14, torch.device("cpu"), rand=u)
    C = vox.coordinates
    assert C[:, 1:].min() >= 0 and C[:, 1:].max() < 14 and set(C[:, 0].tolist()) == {0, 1}
    assert p2v.shape == (250,) and p2v.max() == C.size(0) - 1
    # voxel feature = feature of the FIRST point that fell into the voxel
    first = {}
    for i, v in enumerate(p2v.tolist()):
        first.setdefault(v, i)
    for v, i in list(first.items())[:50]:
        assert torch.equal(vox.features[v], feats[idx[i, 1]])


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from minsu3d_amd import backend
    from minsu3d_amd.parallel import shard_scene_seeds, wrap_ddp
    from oracle.oracle_backend import OracleBackend
    backend.set_backend(OracleBackend())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = wrap_ddp(build_model(seed=0), device=None)
    seeds = shard_scene_seeds(step=0, scenes_per_rank=2, rank=rank, world_size=world)
    batch = small_batch(tuple(seeds))
    loss = model.module.training_step(batch) if False else sum(model.module._loss(batch, model(batch)).values())
    loss.backward()
    g = model.module.backbone.unet[0].kernel.grad.clone()
    gathered = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(gathered, g)
    q.put((rank, seeds, float(loss), all(torch.equal(gathered[0], t) for t in gathered)))
    dist.destroy_process_group()


def test_ddp_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] != res[1][1]                      # ranks own different scenes (no data-path collective)
    assert res[0][3] and res[1][3]                     # gradients identical after the all-reduce


def _sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from minsu3d_amd import backend
    from minsu3d_amd.parallel import sync_buffers
    from oracle.oracle_backend import OracleBackend
    backend.set_backend(OracleBackend())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = build_model(seed=0)                      # same parameters on both ranks
    model.train()
    batch = small_batch((10 + 2 * rank, 11 + 2 * rank))          # different scenes -> different BatchNorm statistics
    with torch.no_grad():
        model(batch)
        if rank == 1:
            model(batch)                             # ... and a different number of batches seen
    bn = model.backbone.unet[1].blocks.block0.conv_branch[0].bn
    mine = (bn.running_mean.clone(), bn.running_var.clone())
    sync_buffers(model)
    after = (bn.running_mean.clone(), bn.running_var.clone(), int(model.state_dict()[
        "backbone.unet.1.blocks.block0.conv_branch.0.bn.num_batches_tracked"]))
    # evaluation with the synchronised statistics gives the same numbers on every rank for the same scene
    model.eval()
    with torch.no_grad():
        out = model(small_batch((20, 21)))
    q.put((rank, [t.tolist() for t in mine], [t.tolist() for t in after[:2]], after[2], float(out["semantic_scores"].sum())))
    dist.destroy_process_group()


def test_sync_buffers_gives_every_rank_rank0_statistics():
    """the reference runs Lightning DDP with torch's default broadcast_buffers=True (rank 0's BatchNorm running
    statistics reach the other ranks at every forward); here the buffers are left alone during training and
    sync_buffers() broadcasts rank 0's before validation / checkpoints (parallel/__init__.py): after it, rank 1
    evaluates with rank 0's statistics and batch counter"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 400)
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, mine0, after0, n0, s0), (_, mine1, after1, n1, s1) = res
    assert mine0 != mine1                              # the ranks had diverged
    assert after0 == mine0 and after1 == mine0         # everybody holds rank 0's statistics now
    assert n0 == n1 == 1                               # ... and rank 0's counter (rank 1 had seen 2 batches)
    assert s0 == s1


def _build(name, seed=0):
    from minsu3d_amd.config import load_config
    import minsu3d_amd.model as M
    torch.manual_seed(seed)
    cfg = load_config([f"model={name}", "model.network.blocks=[1,2,3]", "model.network.m=16"])
    model = getattr(M, cfg.model.network.module)(cfg)
    model.current_epoch = cfg.model.network.prepare_epochs + 1
    return model


def test_hais_step_cpu(cpu_backend):
    model = _build("hais")
    # small synthetic boxes are far below ScanNet's class sizes: use matching statistics
    model.hparams.cfg.data.point_num_avg = [-1, -1] + [400.0] * 18
    model.hparams.cfg.data.radius_avg = [-1.0, -1.0] + [0.3] * 18
    batch = small_batch((7, 8))
    for training in (True, False):              # point aggregation / point + set aggregation
        model.train(training)
        out = model(batch)
        scores, pidx, poff, mask_scores = out["proposal_scores"]
        assert poff.numel() - 1 >= 1 and mask_scores.shape == (pidx.size(0), 1) and scores.shape[0] == poff.numel() - 1
    model.train()
    losses = model._loss(batch, model(batch))
    assert {"mask_loss", "score_loss"} <= set(losses)
    sum(losses.values()).backward()
    assert model.mask_branch[0].weight.grad.abs().sum() > 0 and model.backbone.unet[0].kernel.grad.abs().sum() > 0
    model.current_epoch = 300                    # mask-filtered score features + IoU on predicted masks
    losses = model._loss(batch, model(batch))
    assert all(torch.isfinite(v) for v in losses.values())


def test_softgroup_step_cpu(cpu_backend):
    model = _build("softgroup")
    model.hparams.cfg.data.point_num_avg = [-1, -1] + [400.0] * 18
    batch = small_batch((9, 10))
    n, C = batch["point_xyz"].size(0), model.hparams.cfg.data.classes
    sem = torch.full((n, C), 0.01)
    lab = batch["grouping_semantic_preds"].long()
    sem[torch.arange(n), lab] = 0.8
    sem[torch.arange(n), (lab + 1) % C] = 0.25   # soft grouping: a point may enter two classes' proposals
    batch["grouping_semantic_scores"] = sem
    out = model(batch)
    P = out["proposals_offset"].numel() - 1
    assert 1 <= P <= 200 and out["cls_scores"].shape == (P, 19) and out["iou_scores"].shape == (P, 19)
    # the batched grouping (one ball query + one BFS for all classes) == the reference's per-class loop
    li, lo = model._soft_grouping_loop(batch, sem, batch["grouping_point_offsets"])
    assert torch.equal(li, out["proposals_idx"]) and torch.equal(lo, out["proposals_offset"])
    assert out["mask_scores"].shape == (out["proposals_idx"].size(0), 19)
    losses = model._loss(batch, out)
    assert {"classification_loss", "mask_scoring_loss", "iou_scoring_loss"} <= set(losses)
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    assert model.iou_score.weight.grad is not None and model.tiny_unet.unet[0].blocks.block0.conv_branch[2].kernel.grad.abs().sum() > 0


def test_batchnorm_counter_and_cumulative_momentum(cpu_backend):
    """num_batches_tracked is counted on the host and flushed when read; a loaded state replaces (not adds to) it;
    momentum=None is torch's cumulative moving average"""
    import minsu3d_amd.MinkowskiEngine as ME
    torch.manual_seed(0)
    coords = torch.cat([torch.zeros(50, 1, dtype=torch.int32), torch.arange(150, dtype=torch.int32).view(50, 3)], 1)
    x = [ME.SparseTensor(features=torch.randn(50, 4) * (i + 1), coordinates=coords) for i in range(3)]
    bn, ref = ME.MinkowskiBatchNorm(4, momentum=None), torch.nn.BatchNorm1d(4, momentum=None)
    for t in x:
        bn(t).features
        ref(t.features)
    assert torch.allclose(bn.bn.running_mean, ref.running_mean, atol=1e-6) and torch.allclose(bn.bn.running_var, ref.running_var, rtol=1e-5)
    sd = bn.state_dict()
    assert int(sd["bn.num_batches_tracked"]) == 3
    bn2 = ME.MinkowskiBatchNorm(4)
    bn2(x[0]).features                      # one pending batch on the fresh instance ...
    bn2.load_state_dict(sd)                 # ... which the loaded count replaces
    assert int(bn2.state_dict()["bn.num_batches_tracked"]) == 3


def test_skip_gradient_through_the_first_convolution_equals_autograd(cpu_backend):
    """ResidualBlock hands the skip connection's gradient from its last convolution to its first one (SkipLink) instead
    of letting autograd add it with an elementwise kernel: every parameter gradient of the network must come out the same
    (the same two operands are added), with training-mode and with frozen (eval-mode) BatchNorm statistics -- the latter
    takes the un-fused fallback inside the backward function"""
    from minsu3d_amd.model.module.common import ResidualBlock
    import minsu3d_amd.MinkowskiEngine.functional as Fn
    batch = small_batch()

    def grads(fuse, freeze_bn):
        model = build_model(seed=3)
        model.train()
        if freeze_bn:
            for m in model.modules():
                if m.__class__.__name__ == "MinkowskiBatchNorm":
                    m.eval()
        ResidualBlock.fuse_skip_grad = fuse
        try:
            made = []
            orig = Fn.SkipLink.__init__

            def counting(self):
                orig(self)
                made.append(self)
            Fn.SkipLink.__init__ = counting
            try:
                total = sum(model._loss(batch, model(batch)).values())
                total.backward()
            finally:
                Fn.SkipLink.__init__ = orig
        finally:
            ResidualBlock.fuse_skip_grad = True
        assert all(l.grad is None for l in made)               # every kept-back gradient was consumed
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, len(made), \
            sum(l.armed for l in made)

    for freeze_bn in (False, True):
        fused, n_links, n_armed = grads(True, freeze_bn)
        plain, n_off, _ = grads(False, freeze_bn)
        assert n_links >= 8 and n_armed >= 8 and n_off == 0      # the identity blocks of backbone + ScoreNet took the link
        assert fused.keys() == plain.keys()
        for n in fused:
            assert torch.allclose(fused[n], plain[n], rtol=1e-6, atol=1e-9), (n, freeze_bn)


@pytest.mark.parametrize("name", ["pointgroup", "hais", "softgroup"])
def test_early_head_backward_hands_over_the_same_gradients(cpu_backend, monkeypatch, name):
    """GeneralModel._early_point_backward (scheduling: the heads' backward is queued behind the grouping, its gradients
    handed to autograd later) against the ordinary backward pass: identical parameter gradients for loss = sum(losses)
    (also with a common factor), and through the general path -- unequal loss weights, a loss left out -- too"""
    model = _build(name, seed=2)
    model.hparams.cfg.data.point_num_avg = [-1, -1] + [400.0] * 18
    model.hparams.cfg.data.radius_avg = [-1.0, -1.0] + [0.3] * 18
    model.voxelization_rand = (torch.tensor([0.3, 0.6, 0.9]), torch.tensor([0.1, 0.2, 0.3]))
    batch = small_batch((13, 14))
    if name == "softgroup":
        n = batch["point_xyz"].size(0)
        sem = torch.full((n, 20), 0.01)
        sem[torch.arange(n), batch["grouping_semantic_preds"].long()] = 0.8
        batch["grouping_semantic_scores"] = sem
    model.train()

    def grads(early, combine):
        monkeypatch.setenv("MS3D_EARLY_HEADS", "1" if early else "0")
        model.zero_grad(set_to_none=True)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        losses = model._loss(batch, model(batch))
        combine(losses).backward()
        model.load_state_dict(sd)                 # running statistics back: both evaluations see the same module state
        return {n_: p.grad.detach().clone() for n_, p in model.named_parameters() if p.grad is not None}

    seen = []
    from minsu3d_amd.model import general_model as gm
    real = gm._HandOverGradsFn.apply
    monkeypatch.setattr(gm._HandOverGradsFn, "apply", staticmethod(lambda *a: (seen.append(1), real(*a))[1]))
    weights = {"semantic_loss": 0.7, "offset_norm_loss": 1.3, "offset_dir_loss": 0.4}
    for label, combine, exact in (
            ("sum", lambda l: sum(l.values()), True),
            ("scaled sum", lambda l: sum(l.values()) / 4, True),
            ("weighted", lambda l: sum(weights.get(k, 1.0) * v for k, v in l.items()), False),
            ("one left out", lambda l: sum(v for k, v in l.items() if k != "offset_dir_loss"), False)):
        n0 = len(seen)
        want = grads(False, combine)
        assert len(seen) == n0
        got = grads(True, combine)
        assert len(seen) == n0 + 1, label                       # the early pass ran
        assert got.keys() == want.keys()
        for k in want:
            if exact and ("semantic_branch" in k or "offset_branch" in k):
                assert torch.equal(got[k], want[k]), (label, k)       # the heads' own gradients: the same numbers
            else:
                # below the heads the two branches' contributions to d(point features) meet in another order:
                # (score + sem) + off against score + (sem + off)
                assert torch.allclose(got[k], want[k], rtol=1e-4, atol=2e-6 * float(want[k].abs().max())), (label, k)
