"""Pins to the reference's own Python model code (fixtures: tests/golden/model_tree.json, model_cases.npz, made by
tests/golden/make_golden_model.py from /root/reference in the build container).  CPU: module trees, configuration
values, the forward composition and the instance post-processing host logic with the oracle behind the operators."""
import pytest
import torch

import reference_pins as RP


@pytest.fixture()
def oracle_backend():
    from minsu3d_amd import backend
    from oracle.oracle_backend import OracleBackend
    prev = backend.set_backend(OracleBackend())
    yield
    backend.set_backend(prev)


@pytest.mark.parametrize("name", ["pointgroup", "hais", "softgroup"])
def test_state_dict_is_the_reference_module_tree(name):
    """key names, ORDER, shapes and dtypes of the whole model's state_dict, and which of them
    are trainable, equal the reference's (common.py:21-95, backbone.py:13-34, tiny_unet.py:12-16, pointgroup.py:20-21,
    hais.py:20-26, softgroup.py:20-30) -- a reference checkpoint's `state_dict` loads with strict=True"""
    from minsu3d_amd import model as M
    from minsu3d_amd.config import load_config
    cfg = load_config([f"model={name}"])
    model = getattr(M, cfg.model.network.module)(cfg)
    want = RP.tree()
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()]
    assert got == want[name]
    assert [k for k, _ in model.named_parameters()] == want[name + "_trainable"]
    h = want[name + "_hparams"]
    assert cfg.model.network.m == h["m"] and cfg.model.optimizer.lr == h["lr"]
    assert cfg.model.trainer.max_epochs == h["max_epochs"] and cfg.model.lr_decay.decay_start_epoch == h["decay_start_epoch"]
    assert cfg.model.network.prepare_epochs == h["prepare_epochs"]
    # a checkpoint in Lightning's layout ({'state_dict': ...}) with the reference's keys loads strictly
    ref_sd = {k: torch.zeros(s, dtype=getattr(torch, d)) for k, s, d in want[name]}
    model.load_state_dict(ref_sd, strict=True)


def test_inference_thresholds_are_the_reference_yaml_values():
    from minsu3d_amd.config import load_config
    want = RP.tree()["thresholds"]
    assert dict(load_config(["model=pointgroup"]).model.network.test) == want["pg"]
    assert dict(load_config(["model=hais"]).model.network.test) == want["hais"]
    assert dict(load_config(["model=softgroup"]).model.network.test_cfg) == want["sg"]


def test_forward_composition_vs_reference_modules(oracle_backend):
    """same parameters, same scene, same operators underneath: only the Python composition differs (ours fuses
    BN/ReLU/residual into the convolutions' prologue / epilogue), so the results agree to rounding"""
    worst = RP.check_backbone("cpu", 2e-5)
    print("worst relative error vs the reference's composition: %.2e" % worst)


def test_pred_instances_vs_reference_methods(oracle_backend):
    assert RP.check_instances("cpu") > 500


def test_losses_vs_reference():
    RP.check_losses("cpu")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_postprocess_oracle_is_pinned(seed):
    """oracle/postprocess_oracle.py (the dense-mask restatement the kernels are compared with) against the reference's
    own instance lists -- its header no longer says 'unpinned'"""
    import numpy as np
    from oracle import postprocess_oracle as PO
    from postprocess_cases import make_case, make_softgroup_scores
    c = make_case(seed)
    th = RP.tree()["thresholds"]
    scan = "scene%04d_00" % seed
    got = PO.pointgroup_instances(scan, c["xyz"], c["scores"], c["proposals_idx"], c["P"], c["sem"], 2,
                                  th["pg"]["TEST_SCORE_THRESH"], th["pg"]["TEST_NPOINT_THRESH"], th["pg"]["TEST_NMS_THRESH"])
    RP._check_instances(f"pg_inst{seed}", got)
    got = PO.hais_instances(scan, c["xyz"], c["scores"], c["proposals_idx"], c["P"], c["mask_scores"], c["sem"], 2,
                            th["hais"]["test_mask_score_thre"], th["hais"]["TEST_SCORE_THRESH"],
                            th["hais"]["TEST_NPOINT_THRESH"])
    RP._check_instances(f"hais_inst{seed}", got)
    cls_scores, iou_scores, mask_scores = make_softgroup_scores(seed, c["P"], c["proposals_idx"].shape[0], 18)
    got = PO.softgroup_instances(scan, c["xyz"], c["proposals_idx"], c["n"], cls_scores, iou_scores, mask_scores, 18,
                                 th["sg"]["cls_score_thr"], th["sg"]["mask_score_thr"], th["sg"]["min_npoint"])
    RP._check_instances(f"sg_inst{seed}", got)
