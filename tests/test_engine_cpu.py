"""Train / validate / checkpoint / resume loop on a tiny dataset, oracle backend on the CPU."""
import os

import numpy as np
import pytest
import torch

from minsu3d_amd import backend as ms_backend
from minsu3d_amd import model as ms_models
from minsu3d_amd.data.data_module import DataModule
from minsu3d_amd.engine import Trainer, load_checkpoint
from oracle.oracle_backend import OracleBackend
from test_dataset_cpu import dataset_dir, make_cfg  # noqa: F401  (fixture + helper)


@pytest.fixture(autouse=True)
def oracle_backend():
    prev = ms_backend.set_backend(OracleBackend())
    yield
    ms_backend.set_backend(prev)


def build(cfg, seed=0):
    torch.manual_seed(seed)
    return getattr(ms_models, cfg.model.network.module)(cfg)


def test_fit_checkpoint_resume(dataset_dir, tmp_path):  # noqa: F811
    over = {"data.batch_size": 2, "model.network.m": 8, "model.network.blocks": "[1,2]",
            "model.network.prepare_epochs": 0, "model.trainer.check_val_every_n_epoch": 1, "model.trainer.max_epochs": 4,
            "model.lr_decay.decay_start_epoch": 1, "data.augmentation.elastic": False}
    cfg = make_cfg(dataset_dir, **over)

    def run(epochs, ckpt=None, out="a"):
        np.random.seed(0); torch.manual_seed(0)
        model = build(cfg)
        dm = DataModule(cfg, device="cpu"); dm.setup("fit")
        tr = Trainer(cfg, model, dm, out_dir=str(tmp_path / out), log=lambda r: None)
        # make every epoch's data order / augmentation a function of the epoch only, so a resumed run sees the same batches
        orig = dm.train_dataloader
        def seeded():
            np.random.seed(100 + model.current_epoch); torch.manual_seed(100 + model.current_epoch)
            return orig()
        dm.train_dataloader = seeded
        hist = tr.fit(max_epochs=epochs, ckpt_path=ckpt)
        return model, tr, hist

    m_full, tr_full, h_full = run(3, out="full")
    assert [r["epoch"] for r in h_full] == [0, 1, 2]
    assert all(np.isfinite(r["train/total_loss"]) for r in h_full)
    assert "val_eval/semantic_mean_iou" in h_full[-1] and "val/total_loss" in h_full[-1]
    assert h_full[0]["lr"] == pytest.approx(cfg.model.optimizer.lr) and h_full[2]["lr"] < h_full[1]["lr"] < h_full[0]["lr"] + 1e-12
    assert sorted(os.listdir(tmp_path / "full")) == ["epoch=0.ckpt", "epoch=1.ckpt", "epoch=2.ckpt"]
    # resume from the epoch-1 checkpoint and train epoch 2 again: same weights as the uninterrupted run
    m_res, tr_res, h_res = run(3, ckpt=str(tmp_path / "full" / "epoch=1.ckpt"), out="resumed")
    assert [r["epoch"] for r in h_res] == [2]
    for (k, a), (_, b) in zip(m_full.state_dict().items(), m_res.state_dict().items()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-7), k
    assert h_res[0]["lr"] == pytest.approx(h_full[2]["lr"])
    # a Lightning-style file with only a state_dict loads too
    torch.save({"state_dict": m_full.state_dict()}, tmp_path / "ref_style.ckpt")
    fresh = build(cfg, seed=5)
    assert load_checkpoint(str(tmp_path / "ref_style.ckpt"), fresh) == (-1, 0)
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), m_full.state_dict().values()))


def test_balanced_sampler_shards_and_matches_sizes():
    """every scene once per epoch (up to the wrap-around padding), disjoint across ranks inside a step, and the scenes of
    one step are neighbours in the size order (a step lasts as long as its largest rank)"""
    from minsu3d_amd.parallel import BalancedDistributedBatchSampler
    rng = np.random.default_rng(0)
    sizes = rng.integers(40_000, 260_000, 203).tolist()
    world, bs = 4, 2
    per_rank = []
    for r in range(world):
        s = BalancedDistributedBatchSampler(sizes, bs, rank=r, world_size=world, seed=3)
        s.set_epoch(5)
        per_rank.append(list(s))
    steps = len(per_rank[0])
    assert steps == -(-203 // (world * bs)) and all(len(p) == steps for p in per_rank)
    seen = [i for p in per_rank for b in p for i in b]
    assert set(seen) == set(range(203)) and len(seen) == steps * world * bs
    spread = []
    for t in range(steps):
        batches = [per_rank[r][t] for r in range(world)]
        flat = [i for b in batches for i in b]
        assert len(set(flat)) == len(flat) or t == steps - 1              # disjoint (the padded last step may repeat)
        load = [sum(sizes[i] for i in b) for b in batches]
        spread.append(max(load) / (sum(load) / world))
    naive = []
    order = np.random.default_rng(1).permutation(203)[:steps * world * bs // 1].tolist()
    for t in range(steps - 1):
        load = [sum(sizes[i] for i in order[(t * world + r) * bs:(t * world + r + 1) * bs]) for r in range(world)]
        naive.append(max(load) / (sum(load) / world))
    assert np.mean(spread) < 1.08 < np.mean(naive)                        # slowest rank within 8 % of the mean (random: ~30 %)
    s2 = BalancedDistributedBatchSampler(sizes, bs, rank=0, world_size=world, seed=3)
    s2.set_epoch(6)
    assert list(s2) != per_rank[0]                                        # reshuffled per epoch
    s2.set_epoch(5)
    assert list(s2) == per_rank[0]                                        # and reproducible


def test_balanced_sampler_has_no_systematic_straggler_and_flags_its_padding():
    """the larger scenes of a step do not always go to the same rank (serpentine dealing, rotated per step), and the
    wrap-around repeats that pad the shards are flagged so that validation can leave them out"""
    from minsu3d_amd.parallel import BalancedDistributedBatchSampler
    rng = np.random.default_rng(1)
    sizes = rng.integers(40_000, 260_000, 403).tolist()
    world = 4
    for bs in (1, 2):
        load = np.zeros(world)
        wins = np.zeros(world)
        samplers = [BalancedDistributedBatchSampler(sizes, bs, rank=r, world_size=world, seed=1) for r in range(world)]
        per_rank = [list(s) for s in samplers]
        for t in range(len(per_rank[0])):
            step = [sum(sizes[i] for i in per_rank[r][t]) for r in range(world)]
            load += step
            wins[int(np.argmax(step))] += 1
        assert load.max() / load.mean() < 1.01                    # epoch totals equal to 1 %
        assert wins.max() / wins.sum() < 0.4                      # nobody is the slowest rank of (almost) every step
    # validation: batch 1, no shuffle; 403 scans on 4 ranks -> 404 slots, exactly one flagged repeat, every scan once
    val = [BalancedDistributedBatchSampler(sizes, 1, rank=r, world_size=world, shuffle=False) for r in range(world)]
    kept = []
    for s in val:
        flags = s.padded_positions()
        batches = list(s)
        assert len(flags) == len(batches) == 101
        kept += [b[0] for b, pad in zip(batches, flags) if not pad]
    assert sorted(kept) == list(range(403))


def _fit_worker(rank, world, port, root, out_dir, q):
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import torch.distributed as dist
    ms_backend.set_backend(OracleBackend())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    over = {"data.batch_size": 1, "model.network.m": 8, "model.network.blocks": "[1,2]", "model.network.prepare_epochs": 0,
            "model.trainer.check_val_every_n_epoch": 1, "model.trainer.max_epochs": 2, "data.augmentation.elastic": False}
    cfg = make_cfg(root, **over)
    np.random.seed(rank); torch.manual_seed(0)                 # same initial weights, different augmentation draws
    model = build(cfg)
    dm = DataModule(cfg, device="cpu"); dm.setup("fit")
    seen = []
    orig = dm.train_dataloader
    def tracked(epoch=0):
        loader = orig(epoch)
        seen.append([list(b) for b in loader.batch_sampler])
        return loader
    dm.train_dataloader = tracked
    tr = Trainer(cfg, model, dm, out_dir=out_dir, log=lambda r: None)
    hist = tr.fit(max_epochs=2)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    q.put((rank, seen, [r["train/total_loss"] for r in hist], hist[-1].get("val_eval/semantic_mean_iou"),
           {k: v.double().sum().item() for k, v in state.items()}))
    dist.destroy_process_group()


def test_trainer_fit_ddp_world_size_2_gloo(dataset_dir, tmp_path):  # noqa: F811
    """config 5's loop on 2 ranks (gloo, CPU oracle backend): rank-sharded scenes, gradients all-reduced by DDP, BatchNorm
    buffers taken from rank 0 for validation / checkpoints, one checkpoint per epoch written by rank 0"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    out_dir = str(tmp_path / "ddp")
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, str(dataset_dir), out_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, seen0, loss0, miou0, st0), (_, seen1, loss1, miou1, st1) = res
    assert len(seen0) == 2 and len(seen0[0]) == len(seen1[0]) == 2           # 3 scenes, 2 ranks, batch 1 -> 2 steps (padded)
    for e in range(2):
        for b0, b1 in zip(seen0[e][:1], seen1[e][:1]):
            assert set(b0).isdisjoint(b1)                                       # the ranks of a step hold different scenes
        assert {i for b in seen0[e] + seen1[e] for i in b} == {0, 1, 2}
    assert seen0[0] != seen0[1] or seen1[0] != seen1[1]                         # reshuffled per epoch
    assert all(np.isfinite(loss0)) and all(np.isfinite(loss1)) and loss0 != loss1   # different data per rank
    assert miou0 == miou1                                                       # validation reduced over the ranks
    assert st0.keys() == st1.keys()
    for k in st0:                                                               # same weights AND same BN buffers
        assert st0[k] == pytest.approx(st1[k], rel=1e-6, abs=1e-9), k
    assert sorted(os.listdir(out_dir)) == ["epoch=0.ckpt", "epoch=1.ckpt"]
