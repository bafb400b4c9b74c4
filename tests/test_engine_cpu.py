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
