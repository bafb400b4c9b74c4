"""GPU: the whole PointGroup training step on the HIP backend vs the same step on the CPU oracle backend
(same weights, same scenes): identical proposals (bit-exact indices), activations / losses / gradients within
float tolerance of a 60-layer fp32 network."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

from test_model_cpu import build_model, small_batch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_pointgroup_step_hip_vs_oracle():
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    from oracle.oracle_backend import OracleBackend
    u = (torch.tensor([0.3, 0.6, 0.9]), torch.tensor([0.1, 0.2, 0.3]))
    ref_model = build_model(seed=1)
    ref_model.voxelization_rand = u
    hip_model = copy.deepcopy(ref_model).cuda()
    hip_model.voxelization_rand = tuple(t.cuda() for t in u)
    batch = small_batch((3, 4))
    prev = backend.set_backend(OracleBackend())
    try:
        out_r = ref_model(batch)
        loss_r = ref_model._loss(batch, out_r)
        sum(loss_r.values()).backward()
    finally:
        backend.set_backend(prev)
    backend.set_backend(HipBackend())
    batch_d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    out_h = hip_model(batch_d)
    loss_h = hip_model._loss(batch_d, out_h)
    sum(loss_h.values()).backward()
    # proposals: same grouping inputs -> bit-exact clusters
    assert torch.equal(out_h["proposal_scores"][1].cpu(), out_r["proposal_scores"][1])
    assert torch.equal(out_h["proposal_scores"][2].cpu(), out_r["proposal_scores"][2])
    assert rel(out_h["point_features"], out_r["point_features"]) < 2e-3
    assert rel(out_h["semantic_scores"], out_r["semantic_scores"]) < 2e-3
    assert rel(out_h["proposal_scores"][0], out_r["proposal_scores"][0]) < 5e-3
    for k in loss_r:
        assert abs(float(loss_h[k].detach()) - float(loss_r[k].detach())) < 2e-3 * max(1.0, abs(float(loss_r[k].detach()))), k
    # Gradients: both sides are float32 evaluations of a piecewise-smooth 60-layer network on tiny scenes (proposal grids
    # of a few hundred voxels behind BatchNorm), so a handful of ReLU masks differ between the two roundings and each
    # flip moves a gradient sum by a whole term.  The smooth comparison -- every backward kernel against float64 autograd
    # at bench size, all tensors within 1e-4 (measured 3e-6) -- is
    # tests/test_fullsize_gpu.py::test_unet_gradients_vs_fp64_on_a_full_scene; here the bar is that no tensor is off by
    # more than a few flipped terms and that the typical tensor agrees to float32 accumulation noise.
    gr = dict(ref_model.named_parameters())
    errs = []
    for n, p in hip_model.named_parameters():
        if gr[n].grad is None:
            assert p.grad is None
            continue
        if gr[n].grad.abs().max() < 1e-6:      # mathematically zero (e.g. a Linear bias in front of BatchNorm1d)
            assert p.grad.abs().max() < 1e-5, n
            continue
        errs.append((rel(p.grad, gr[n].grad), n))
    errs.sort(reverse=True)
    print(f"gradients HIP vs oracle backend: worst {errs[0][0]:.2e} ({errs[0][1]}), median {errs[len(errs) // 2][0]:.2e}")
    assert errs[0][0] < 3e-2, errs[0]
    assert errs[len(errs) // 2][0] < 2e-3, errs[len(errs) // 2]


def test_step_is_reproducible_on_device():
    """same inputs twice (eval mode) -> identical proposals and scores; the TRAINING-mode version of this check, byte for byte
    over losses, gradients and statistics, is tests/test_determinism_gpu.py"""
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    backend.set_backend(HipBackend())
    m = build_model(seed=2).cuda()
    u = (torch.tensor([0.5, 0.5, 0.5]).cuda(), torch.tensor([0.5, 0.5, 0.5]).cuda())
    m.voxelization_rand = u
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in small_batch((5, 6)).items()}
    m.eval()   # freeze running stats so both passes see the same module state
    o1 = m(b); o2 = m(b)
    assert torch.equal(o1["proposal_scores"][1], o2["proposal_scores"][1])
    assert torch.equal(o1["semantic_scores"], o2["semantic_scores"])


@pytest.mark.parametrize("name", ["hais", "softgroup"])
def test_hais_softgroup_forward_hip_vs_oracle(name):
    """configs 3/4 callers: identical proposals (bit-exact grouping) and close head outputs / losses"""
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    from oracle.oracle_backend import OracleBackend
    from test_model_cpu import _build
    u = (torch.tensor([0.3, 0.6, 0.9]), torch.tensor([0.1, 0.2, 0.3]))
    ref_model = _build(name, seed=3)
    ref_model.hparams.cfg.data.point_num_avg = [-1, -1] + [400.0] * 18
    ref_model.hparams.cfg.data.radius_avg = [-1.0, -1.0] + [0.3] * 18
    ref_model.voxelization_rand = u
    hip_model = copy.deepcopy(ref_model).cuda()
    hip_model.voxelization_rand = tuple(t.cuda() for t in u)
    batch = small_batch((11, 12))
    if name == "softgroup":
        n, C = batch["point_xyz"].size(0), 20
        sem = torch.full((n, C), 0.01)
        sem[torch.arange(n), batch["grouping_semantic_preds"].long()] = 0.8
        batch["grouping_semantic_scores"] = sem
    prev = backend.set_backend(OracleBackend())
    try:
        out_r = ref_model(batch)
        loss_r = ref_model._loss(batch, out_r)
    finally:
        backend.set_backend(prev)
    backend.set_backend(HipBackend())
    batch_d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    out_h = hip_model(batch_d)
    loss_h = hip_model._loss(batch_d, out_h)
    sum(loss_h.values()).backward()
    if name == "hais":
        assert torch.equal(out_h["proposal_scores"][1].cpu(), out_r["proposal_scores"][1])
        assert torch.equal(out_h["proposal_scores"][2].cpu(), out_r["proposal_scores"][2])
        assert rel(out_h["proposal_scores"][3], out_r["proposal_scores"][3]) < 5e-3
    else:
        assert torch.equal(out_h["proposals_idx"].cpu(), out_r["proposals_idx"])
        assert torch.equal(out_h["proposals_offset"].cpu(), out_r["proposals_offset"])
        assert rel(out_h["cls_scores"], out_r["cls_scores"]) < 5e-3
    for k in loss_r:
        assert abs(float(loss_h[k].detach()) - float(loss_r[k].detach())) < 3e-3 * max(1.0, abs(float(loss_r[k].detach()))), k


def test_softgroup_batched_grouping_equals_per_class_loop_on_device():
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    from test_model_cpu import _build
    backend.set_backend(HipBackend())
    m = _build("softgroup", seed=4).cuda()
    m.hparams.cfg.data.point_num_avg = [-1, -1] + [400.0] * 18
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in small_batch((13, 14)).items()}
    n = b["point_xyz"].size(0)
    sem = torch.full((n, 20), 0.01, device="cuda")
    lab = b["grouping_semantic_preds"].long()
    sem[torch.arange(n, device="cuda"), lab] = 0.8
    sem[torch.arange(n, device="cuda"), (lab + 3) % 20] = 0.3
    a1, o1 = m._soft_grouping_loop(b, sem, b["grouping_point_offsets"])
    a2, o2 = m._soft_grouping(b, sem, b["grouping_point_offsets"])
    assert o1.numel() > 3 and torch.equal(a1, a2) and torch.equal(o1, o2)


def test_weight_images_one_launch_equals_per_layer_and_tracks_updates(monkeypatch):
    """All convolution weight images of a model are laid out in ONE launch at the start of its forward
    (ME.prepare_conv_weights).  Two training steps must give exactly the per-layer path's parameters -- i.e. step 2
    sees the weights Adam wrote in step 1 -- and a convolution called OUTSIDE the model's forward after a weight
    change must not use a stale image."""
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    import minsu3d_amd.MinkowskiEngine as ME
    backend.set_backend(HipBackend())
    u = tuple(t.cuda() for t in (torch.tensor([0.3, 0.6, 0.9]), torch.tensor([0.1, 0.2, 0.3])))
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in small_batch((7, 8)).items()}

    def two_steps(multi):
        monkeypatch.setenv("MS3D_WEIGHT_MULTI", "1" if multi else "0")
        m = build_model(seed=3).cuda()
        m.voxelization_rand = u
        m.eval()                       # fixed BN statistics
        opt = torch.optim.SGD(m.parameters(), lr=5e-2)   # (Adam would turn last-bit gradient noise into +-lr steps)
        losses = []
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss = sum(m._loss(b, m(b)).values())
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return m, losses

    m1, l1 = two_steps(True)
    assert any(hasattr(p, "_ms3d_wf") for p in m1.parameters())          # the one-launch path really ran
    m0, l0 = two_steps(False)
    assert not any(hasattr(p, "_ms3d_wf") for p in m0.parameters())
    assert l1[0] == l0[0] and abs(l1[1] - l0[1]) < 1e-5 * abs(l0[1]) and abs(l1[0] - l1[1]) > 1e-3 * abs(l1[0])
    for (n, p), (_, q) in zip(m1.named_parameters(), m0.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-6), n
    # outside the model's forward the stamp is void: a changed weight is re-laid by the convolution itself
    monkeypatch.setenv("MS3D_WEIGHT_MULTI", "1")
    conv = next(mod for mod in m1.modules() if isinstance(mod, ME.MinkowskiConvolution) and mod.kernel_size == 3
                and mod.in_channels % 16 == 0)
    coords = torch.cat([torch.zeros(200, 1, dtype=torch.int32), torch.randint(0, 12, (200, 3), dtype=torch.int32)], 1).unique(dim=0).cuda()
    x = ME.SparseTensor(features=torch.randn(coords.size(0), conv.in_channels, device="cuda"), coordinates=coords)
    with torch.no_grad():
        y0 = conv(x).F.clone()
        conv.kernel.mul_(2.0)
        y1 = conv(x).F
    assert torch.allclose(y1, 2 * y0, rtol=1e-5, atol=1e-6)


def test_prefetched_coordinate_structures_give_the_same_forward():
    """ME.prefetch_coordinates builds the row order, kernel maps and pair lists of a batch on the helper thread / side
    stream; the forward that picks them up must equal the one that builds them itself, and a prefetch is used once."""
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    import minsu3d_amd.MinkowskiEngine as ME
    from minsu3d_amd.MinkowskiEngine import tensor as me_tensor
    backend.set_backend(HipBackend())
    m = build_model(seed=4).cuda()
    m.voxelization_rand = (torch.tensor([0.5, 0.5, 0.5]).cuda(), torch.tensor([0.5, 0.5, 0.5]).cuda())
    m.eval()
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in small_batch((9, 10)).items()}
    with torch.no_grad():
        ref = m(b)
        ME.prefetch_coordinates(b["voxel_xyz"], m.backbone.n_levels)
        assert len(me_tensor._PREFETCHED) == 1
        got = m(b)
        assert len(me_tensor._PREFETCHED) == 0            # consumed: the next forward builds its own
        again = m(b)
    for o in (got, again):
        assert torch.equal(o["semantic_scores"], ref["semantic_scores"])
        assert torch.equal(o["point_offsets"], ref["point_offsets"])
        assert torch.equal(o["proposal_scores"][1], ref["proposal_scores"][1])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_proposal_voxel_coords_fused_vs_expression_chain(seed):
    """the fused device operator against the reference's expression chain in torch operators (general_model.py:152-181)
    on the same device: identical integer voxel coordinates -- proposals of 1 point (zero extent: the scale clamps at
    `scale`), of two points, of thousands, with shared points, scaled up and scaled down"""
    from minsu3d_amd.backend import get_backend
    from minsu3d_amd.model.general_model import proposal_voxel_coords_torch
    g = torch.Generator().manual_seed(seed)
    n = 60000
    coords = (torch.rand(n, 3, generator=g) * torch.tensor([8.0, 6.0, 3.0])).cuda()
    sizes = [1, 2, 3, 17, 64, 65, 500, 4097, 20000, 1] + torch.randint(1, 3000, (120,), generator=g).tolist()
    idx, off = [], [0]
    for p, sz in enumerate(sizes):
        centre = torch.rand(3, generator=g) * torch.tensor([8.0, 6.0, 3.0])
        radius = float(torch.rand(1, generator=g)) * (0.02 if p % 7 == 0 else 2.0) + 0.005   # tiny ones scale up to the clamp
        near = torch.nonzero(((coords.cpu() - centre).abs() < radius).all(1)).view(-1)
        pick = near[torch.randperm(near.numel(), generator=g)[:sz]] if near.numel() >= sz else torch.randint(0, n, (sz,), generator=g)
        idx.append(torch.stack((torch.full((pick.numel(),), p, dtype=torch.int64), pick.long()), 1))
        off.append(off[-1] + pick.numel())
    idx = torch.cat(idx).cuda().contiguous()
    off = torch.tensor(off, dtype=torch.int32).cuda()
    u = torch.rand(6, generator=g).cuda()
    for scale, ss in ((50, 14), (15, 20), (3, 128)):
        want = proposal_voxel_coords_torch(idx, off, coords, scale, ss, u[:3], u[3:])
        got = get_backend().proposal_voxel_coords(idx, off, coords, scale, ss, u)
        assert got.dtype == torch.int32 and torch.equal(got, want)
        assert int(got[:, 1:].min()) >= 0 and int(got[:, 1:].max()) < ss


@pytest.mark.gpu
def test_one_launch_adam_matches_torch_adam():
    """minsu3d_amd.optim.Adam (one library launch per step) against torch.optim.Adam on the same parameters and
    gradients: tensors of 1 .. 300k elements (chunk boundaries, odd sizes), a sliced (unaligned) gradient, weight decay,
    20 steps with a changing learning rate; the optimizer states are interchangeable"""
    from minsu3d_amd.optim import Adam
    g = torch.Generator().manual_seed(5)
    shapes = [(1,), (3,), (16,), (17, 3), (4096,), (4097,), (27, 16, 16), (300001,), (8192, 2)]
    for wd in (0.0, 0.01):
        pa = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
        pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
        oa = Adam(pa, lr=1e-3, weight_decay=wd)
        ob = torch.optim.Adam(pb, lr=1e-3, weight_decay=wd)
        for step in range(20):
            for grp in (oa.param_groups[0], ob.param_groups[0]):
                grp["lr"] = 1e-3 * (1.0 - 0.03 * step)
            for a, b in zip(pa, pb):
                grad = torch.randn(a.numel() + 1, generator=g).cuda()
                a.grad = grad[1:].view_as(a) if a.numel() % 2 else grad[:-1].view_as(a).clone()   # odd sizes: 4-byte aligned only
                b.grad = a.grad.clone()
            oa.step(); ob.step()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), float((a - b).abs().max())
            assert torch.allclose(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=2e-5, atol=1e-12)   # 20 steps of f32 rounding (torch: separate mul_ and addcmul_)
            assert float(oa.state[a]["step"]) == float(ob.state[b]["step"]) == 20.0
        # torch's Adam continues from our state, and the other way round (deep copies, as a checkpoint file gives them:
        # torch's load_state_dict keeps the tensor objects it is handed)
        ob.load_state_dict(copy.deepcopy(oa.state_dict()))
        oa.load_state_dict(copy.deepcopy(ob.state_dict()))
        for a, b in zip(pa, pb):
            a.grad = torch.ones_like(a)
            b.grad = torch.ones_like(b)
        oa.step(); ob.step()
        steps = [ob.state[b]["step"] for b in pb]
        assert all(float(t) == 21.0 for t in steps)                  # one increment per parameter, not one per alias
        assert len({id(t) for t in steps}) == len(steps)
        assert all(float(oa.state_dict()["state"][i]["step"]) == 21.0 for i in range(len(pa)))
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
def test_one_launch_adam_starts_each_parameter_at_its_own_first_gradient():
    """torch.optim.Adam keeps one step counter per parameter and creates it with the parameter's first gradient: the
    ScoreNet / refinement branches get gradients only after `prepare_epochs` (pointgroup.py:25, hais.py:30,
    softgroup.py:34) and must then be bias-corrected as step 1, 2, ... while the backbone is thousands of steps in"""
    from minsu3d_amd.optim import Adam
    g = torch.Generator().manual_seed(9)
    shapes = [(27, 16, 16), (16,), (5000,), (33,), (16, 1)]
    late = (2, 4)                                                    # these see their first gradient at step 10
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = Adam(pa, lr=2e-3), torch.optim.Adam(pb, lr=2e-3)
    for step in range(25):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i in late and (step < 10 or step == 17):              # ... and skip one step later on
                a.grad = b.grad = None
                continue
            a.grad = torch.randn(a.shape, generator=g).cuda()
            b.grad = a.grad.clone()
        oa.step(); ob.step()
        if step in (9, 10, 12, 24):
            for i, (a, b) in enumerate(zip(pa, pb)):
                assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (step, i, float((a - b).abs().max()))
    sd = oa.state_dict()["state"]
    assert [float(sd[i]["step"]) for i in range(5)] == [25.0, 25.0, 14.0, 25.0, 14.0]
    assert [float(ob.state[b]["step"]) for b in pb] == [25.0, 25.0, 14.0, 25.0, 14.0]
    # resumed from a state whose counters differ per parameter: continues per parameter
    oc = Adam(pa, lr=2e-3)
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))
    for a, b in zip(pa, pb):
        a.grad = torch.ones_like(a); b.grad = torch.ones_like(b)
    oc.step(); ob.step()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7)
    assert [float(oc.state_dict()["state"][i]["step"]) for i in range(5)] == [26.0, 26.0, 15.0, 26.0, 15.0]


@pytest.mark.gpu
@pytest.mark.parametrize("C_", [16, 32, 20])
def test_point_batchnorm_relu_matches_torch(C_):
    """the heads' BatchNorm1d + ReLU through the engine's kernels against torch.nn.BatchNorm1d + ReLU: outputs, input /
    weight / bias gradients, running statistics; training and evaluation mode"""
    from minsu3d_amd.model.module.networks import PointBatchNormReLU
    torch.manual_seed(C_)
    ours = PointBatchNormReLU(C_).cuda()
    ref = torch.nn.BatchNorm1d(C_).cuda()
    with torch.no_grad():
        ours.weight.uniform_(0.5, 1.5); ours.bias.uniform_(-0.3, 0.3)
    ref.load_state_dict(ours.state_dict())
    for train in (True, False):
        ours.train(train); ref.train(train)
        x1 = (torch.randn(30000, C_, device="cuda") * 2 + 0.5).requires_grad_(True)
        x2 = x1.detach().clone().requires_grad_(True)
        y1, y2 = ours(x1), torch.relu(ref(x2))
        assert torch.allclose(y1, y2, rtol=1e-5, atol=1e-5)
        g = torch.randn_like(y1)
        y1.backward(g); y2.backward(g)
        assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-6)
        assert torch.allclose(ours.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-3)
        assert torch.allclose(ours.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-3)
        ours.zero_grad(); ref.zero_grad()
    assert torch.allclose(ours.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(ours.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked) == 1


def test_backward_weight_on_a_second_stream_gives_the_same_gradients():
    """MS3D_WGRAD_STREAM: the backward-weight kernels of every layer on a second stream beside the backward-data chain
    (joined per layer = 1, once at the end of the backward pass = 2, or per layer group in its deferred-reduction node = 3)
    against the single-stream order: the same
    kernels on the same data, so every gradient agrees (a missing join would show as garbage), also when the step is
    repeated back to back"""
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    be = HipBackend()
    backend.set_backend(be)
    u = tuple(t.cuda() for t in (torch.tensor([0.3, 0.6, 0.9]), torch.tensor([0.1, 0.2, 0.3])))
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in small_batch((21, 22)).items()}
    m = build_model(seed=5).cuda()
    m.voxelization_rand = u
    m.eval()                      # fixed BatchNorm statistics
    grads = {}
    try:
        for mode in (0, 1, 2, 2, 3, 3):
            be._wgrad_mode = mode
            m.zero_grad(set_to_none=True)
            sum(m._loss(b, m(b)).values()).backward()
            g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
            if mode == 0:
                grads = g
            else:
                assert g.keys() == grads.keys()
                for n in g:
                    scale = grads[n].abs().max().item()
                    assert (g[n] - grads[n]).abs().max().item() <= 1e-4 * scale + 1e-12, (mode, n)
    finally:
        be._wgrad_mode = 0


def test_deferred_slab_reduction_gives_the_same_gradients(monkeypatch):
    """The backward-weight slab reductions of a group of layers run as ONE launch when the group's last layer is done
    (modules.prepare_conv_weights -> functional.GroupFlushFn -> backend.WgradQueue) instead of one launch per layer:
    same kernels' slabs, same per-element summation order.  Whole-model gradients with and without the deferral
    (eval-mode statistics), flush count = layer groups, every convolution's
    reduction accounted for; with an existing .grad (accumulation) the sum is right too."""
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend, WgradQueue
    be = HipBackend()
    backend.set_backend(be)
    u = tuple(t.cuda() for t in (torch.tensor([0.3, 0.6, 0.9]), torch.tensor([0.1, 0.2, 0.3])))
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in small_batch((21, 22)).items()}
    m = build_model(seed=5).cuda()
    m.voxelization_rand = u
    m.eval()
    flushed = []
    real_flush = WgradQueue.flush
    monkeypatch.setattr(WgradQueue, "flush", lambda self: (flushed.append(len(self.items)), real_flush(self))[1])

    def grads(defer, accumulate=False):
        be._wgrad_defer = defer
        if not accumulate:
            m.zero_grad(set_to_none=True)
        sum(m._loss(b, m(b)).values()).backward()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

    try:
        want = grads(False)
        assert not flushed
        got = grads(True)
        n_convs = sum(1 for mod in m.modules() if hasattr(mod, "kernel_volume"))
        assert 2 <= len(flushed) <= 4 and 0.8 * n_convs <= sum(flushed) <= n_convs, (flushed, n_convs)
        assert got.keys() == want.keys()
        for n in got:
            scale = want[n].abs().max().item()
            assert (got[n] - want[n]).abs().max().item() <= 1e-4 * scale + 1e-12, n
        twice = grads(True, accumulate=True)             # .grad exists: autograd accumulates behind the flush node
        for n in twice:
            scale = want[n].abs().max().item()
            assert (twice[n] - 2 * want[n]).abs().max().item() <= 2e-4 * scale + 1e-12, n
    finally:
        be._wgrad_defer = None


@pytest.mark.parametrize("n,C_,valid", [(50000, 20, 0.8), (777, 20, 0.5), (3000, 7, 0.0)])
def test_fused_point_losses_match_the_torch_formulation(n, C_, valid):
    """csrc/losses.hip (one autograd node, three launches) against GeneralModel._point_losses_torch -- itself pinned to the
    reference's GeneralModel._loss / PTOffsetLoss by tests/golden/model_cases.npz: the three loss values and the gradients
    w.r.t. the semantic scores and the offsets, with unequal upstream weights, ignored labels, points without an instance,
    a zero prediction (the eps branch of the normalisation) and the no-valid-point case"""
    from types import SimpleNamespace
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    from minsu3d_amd.loss import PTOffsetLoss
    from minsu3d_amd.model.general_model import GeneralModel
    backend.set_backend(HipBackend())
    g = torch.Generator().manual_seed(n)
    labels = torch.randint(0, C_, (n,), generator=g).to(torch.int16)
    labels[torch.rand(n, generator=g) > max(valid, 0.3)] = -1
    inst = torch.randint(0, 9, (n,), generator=g).to(torch.int16)
    inst[torch.rand(n, generator=g) >= valid] = -1
    d = {"sem_labels": labels.cuda(), "instance_ids": inst.cuda(), "point_xyz": (torch.rand(n, 3, generator=g) * 4).cuda(),
         "instance_center_xyz": (torch.rand(n, 3, generator=g) * 4).cuda()}
    scores0 = (torch.randn(n, C_, generator=g) * 3).cuda()
    offs0 = torch.randn(n, 3, generator=g).cuda()
    offs0[5] = 0.0
    offs0[6] = d["instance_center_xyz"][6] - d["point_xyz"][6]          # exact hit: sign(0) = 0
    me = SimpleNamespace(offset_criterion=PTOffsetLoss())
    w = (0.7, 1.3, 0.4)
    res = []
    for fused in (True, False):
        scores, offs = scores0.clone().requires_grad_(True), offs0.clone().requires_grad_(True)
        fn = GeneralModel._point_losses if fused else GeneralModel._point_losses_torch
        losses = fn(me, d, {"semantic_scores": scores, "point_offsets": offs})
        assert list(losses) == ["semantic_loss", "offset_norm_loss", "offset_dir_loss"]
        sum(wi * li for wi, li in zip(w, losses.values())).backward()
        res.append(([float(v.detach()) for v in losses.values()], scores.grad.clone(), offs.grad.clone()))
    (lf, gsf, gof), (lt, gst, got_) = res
    for a, b_ in zip(lf, lt):
        assert abs(a - b_) <= 2e-6 * max(abs(b_), 1e-3), (lf, lt)
    assert (gsf - gst).abs().max().item() <= 2e-6 * max(gst.abs().max().item(), 1e-12) + 1e-12
    assert (gof - got_).abs().max().item() <= 2e-5 * max(got_.abs().max().item(), 1e-12) + 1e-12
    if valid == 0.0:
        assert lf[1] == 0.0 and lf[2] == 0.0 and float(gof.abs().max()) == 0.0


def test_fused_residual_blocks_give_the_same_step(monkeypatch):
    """ResidualBlock as ONE autograd node (ME.functional.ResBlockFn: the same four library calls forward, the same two
    backward, only the interpreter work between them is gone) against the module chain: same proposals, losses and
    gradients of a training step bit for bit, same running statistics and batch counters; inner-module hooks switch a
    block back to the module chain."""
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    from minsu3d_amd.model.module import common
    backend.set_backend(HipBackend())
    u = tuple(t.cuda() for t in (torch.tensor([0.3, 0.6, 0.9]), torch.tensor([0.1, 0.2, 0.3])))
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in small_batch((31, 32)).items()}
    base = build_model(seed=9)
    res = []
    calls = []
    real_apply = common.ME_F.ResBlockFn.apply
    monkeypatch.setattr(common.ME_F.ResBlockFn, "apply", staticmethod(lambda *a: (calls.append(1), real_apply(*a))[1]))
    for fused in (True, False):
        monkeypatch.setattr(common, "_FUSE_BLOCKS", fused)
        m = copy.deepcopy(base).cuda()
        m.voxelization_rand = u
        m.train()
        n0 = len(calls)
        out = m(b)
        losses = m._loss(b, out)
        sum(losses.values()).backward()
        assert (len(calls) > n0) == fused
        sd = m.state_dict()
        res.append((out, {k: float(v.detach()) for k, v in losses.items()},
                    {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None},
                    {k: v.clone() for k, v in sd.items() if "running" in k or "num_batches" in k}))
    n_blocks = len(calls)
    assert n_blocks >= 10                                            # most blocks of the two U-Nets took the fused node
    (o1, l1, g1, s1), (o2, l2, g2, s2) = res
    assert torch.equal(o1["proposal_scores"][1], o2["proposal_scores"][1])
    for k in l1:
        assert l1[k] == l2[k], (k, l1[k], l2[k])
    assert g1.keys() == g2.keys()
    # the same library calls on the same data in the same order, and (round 5) every float sum of a step has a fixed
    # order: the two forms of the block agree BIT FOR BIT -- gradients, running statistics, batch counters.  (Rounds 1-4
    # summed the training-mode statistics with LDS float atomics; this comparison had to allow 5e-2 for flipped masks.)
    bad = [(n, ((g1[n] - g2[n]).abs().max() / (g2[n].abs().max() + 1e-30)).item()) for n in g1 if not torch.equal(g1[n], g2[n])]
    assert not bad, (len(bad), bad[:5])
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
    # a forward hook on an inner module: that block takes the module chain (the hook fires)
    monkeypatch.setattr(common, "_FUSE_BLOCKS", True)
    m = copy.deepcopy(base).cuda(); m.voxelization_rand = u; m.train()
    blk = m.backbone.unet[1].blocks.block0
    seen = []
    h = blk.conv_branch[1].register_forward_hook(lambda mod, i, o: seen.append(1))
    n0 = len(calls)
    m(b)
    h.remove()
    assert seen and len(calls) - n0 == n_blocks - 1


@pytest.mark.parametrize("mode", ["1", "2"])
def test_early_head_backward_modes_give_the_same_step(monkeypatch, mode):
    """MS3D_EARLY_HEADS (scheduling only: 1 = the heads' backward queued behind the grouping, 2 = heads, losses and their
    backward on their own stream beside the grouping window) against the ordinary step on the device: identical proposals
    and losses, the heads' own gradients bit for bit, everything below them to the rounding of one reordered addition
    (d point_features = ScoreNet part + heads part, summed in the other order); repeated, the step gives the same bytes"""
    from minsu3d_amd import backend
    from minsu3d_amd.backend import HipBackend
    backend.set_backend(HipBackend())
    u = tuple(t.cuda() for t in (torch.tensor([0.3, 0.6, 0.9]), torch.tensor([0.1, 0.2, 0.3])))
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in small_batch((51, 52)).items()}
    base = build_model(seed=11)

    def step(env):
        monkeypatch.setenv("MS3D_EARLY_HEADS", env)
        m = copy.deepcopy(base).cuda()
        m.voxelization_rand = u
        m.train()
        out = m(b)
        losses = m._loss(b, out)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        return (out["proposal_scores"][1].clone(), {k: float(v.detach()) for k, v in losses.items()},
                {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})

    p0, l0, g0 = step("0")
    p1, l1, g1 = step(mode)
    p2, l2, g2 = step(mode)
    assert torch.equal(p0, p1) and l0 == l1 and g0.keys() == g1.keys()
    for n in g0:
        if "semantic_branch" in n or "offset_branch" in n:
            assert torch.equal(g0[n], g1[n]), n
        else:
            assert (g0[n] - g1[n]).abs().max().item() <= 1e-5 * g0[n].abs().max().item() + 1e-12, n
        assert torch.equal(g1[n], g2[n]), n
