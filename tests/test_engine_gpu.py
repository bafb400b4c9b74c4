"""One training epoch + validation (device post-processing -> evaluator) + checkpoint on the GPU through the HIP
library, on scenes stored in the reference's .pth format."""
import os

import numpy as np
import pytest
import torch

from test_dataset_cpu import dataset_dir, make_cfg  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("module", ["pointgroup", "hais", "softgroup"])
def test_epoch_validate_checkpoint_on_gpu(dataset_dir, tmp_path, module):  # noqa: F811
    from minsu3d_amd import backend as B
    from minsu3d_amd import model as ms_models
    from minsu3d_amd.config import load_config
    from minsu3d_amd.data.data_module import DataModule
    from minsu3d_amd.engine import Trainer, load_checkpoint
    be = B.get_backend()
    assert be.name != "oracle-cpu"
    root = dataset_dir
    ov = [f"model={module}", f"data.dataset_path={root}", f"data.metadata.train_list={root}/train.txt",
          f"data.metadata.val_list={root}/val.txt", "data.batch_size=2", "model.network.prepare_epochs=-1",
          "model.trainer.check_val_every_n_epoch=1"]
    cfg = load_config(ov)
    torch.manual_seed(0); np.random.seed(0)
    model = getattr(ms_models, cfg.model.network.module)(cfg).cuda()
    dm = DataModule(cfg, device="cuda", elastic_fn=lambda x, noise, g, m: be.elastic(
        torch.from_numpy(np.asarray(x)).cuda(), torch.from_numpy(noise).cuda(), g, m).cpu().numpy())
    dm.setup("fit")
    tr = Trainer(cfg, model, dm, out_dir=str(tmp_path), log=lambda r: None)
    hist = tr.fit(max_epochs=1)
    assert np.isfinite(hist[0]["train/total_loss"]) and "val_eval/semantic_accuracy" in hist[0]
    assert os.path.exists(tmp_path / "epoch=0.ckpt")
    clone = getattr(ms_models, cfg.model.network.module)(cfg).cuda()
    assert load_checkpoint(str(tmp_path / "epoch=0.ckpt"), clone) == (0, tr.global_step)
    assert all(torch.equal(a, b) for a, b in zip(clone.state_dict().values(), model.state_dict().values()))
