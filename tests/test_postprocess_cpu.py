"""Host logic of the instance post-processing (minsu3d_amd/model/postprocess.py) with the oracle behind the two
operators, against the dense-mask restatement of the reference (oracle/postprocess_oracle.py), plus hand-made cases."""
import numpy as np
import pytest
import torch

from minsu3d_amd import backend as ms_backend
from minsu3d_amd.model import postprocess as PP
from oracle import postprocess_oracle as PO
from oracle.oracle_backend import OracleBackend
from postprocess_cases import assert_same_instances, make_case


@pytest.fixture(autouse=True)
def oracle_backend():
    prev = ms_backend.set_backend(OracleBackend())
    yield
    ms_backend.set_backend(prev)


def test_nms_known_answer():
    # three proposals over 10 points: A = {0..5}, B = {2..7} (IoU 4/8), C = {8, 9}
    pts = np.array([0, 1, 2, 3, 4, 5, 2, 3, 4, 5, 6, 7, 8, 9])
    cl = np.array([0] * 6 + [1] * 6 + [2] * 2)
    o = np.argsort(pts * 3 + cl)
    inter = PO.cross_intersection(pts[o], cl[o], 3)
    assert inter.tolist() == [[6, 4, 0], [4, 6, 0], [0, 0, 2]]
    assert PO.nms_from_counts(inter, [1, 0, 2], 0.3).tolist() == [1, 2]      # A suppressed by B (IoU 0.5 > 0.3)
    assert PO.nms_from_counts(inter, [1, 0, 2], 0.5).tolist() == [1, 0, 2]   # strict '>' keeps IoU == threshold


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_pointgroup_instances_vs_dense_restatement(seed):
    c = make_case(seed)
    want = PO.pointgroup_instances("scene", c["xyz"], c["scores"], c["proposals_idx"], c["P"], c["sem"], 2, 0.09, 100, 0.3)
    got = PP.pointgroup_instances("scene", c["xyz"], torch.from_numpy(c["scores"]), torch.from_numpy(c["proposals_idx"]),
                                  c["P"], torch.from_numpy(c["sem"]), 2, 0.09, 100, 0.3)
    assert len(want) > 3
    assert_same_instances(got, want)


def test_hais_instances_vs_dense_restatement():
    c = make_case(5)
    want = PO.hais_instances("scene", c["xyz"], c["scores"], c["proposals_idx"], c["P"], c["mask_scores"], c["sem"], 2,
                             -0.5, 0.09, 100)
    got = PP.hais_instances("scene", c["xyz"], torch.from_numpy(c["scores"]), torch.from_numpy(c["proposals_idx"]), c["P"],
                            torch.from_numpy(c["mask_scores"]), torch.from_numpy(c["sem"]), 2, -0.5, 0.09, 100)
    assert len(want) > 3
    assert_same_instances(got, want)


def test_no_proposal_survives():
    c = make_case(3)
    got = PP.pointgroup_instances("scene", c["xyz"], torch.from_numpy(c["scores"]) - 100, torch.from_numpy(c["proposals_idx"]),
                                  c["P"], torch.from_numpy(c["sem"]), 2, 0.09, 100, 0.3)
    assert got == []


def test_softgroup_instances_vs_dense_restatement():
    c = make_case(7, n=4000, n_regions=6, per_region=3, junk=3)
    rng = np.random.default_rng(11)
    K = 4
    cls = rng.standard_normal((c["P"], K + 1)).astype(np.float32) * 2
    iou = rng.uniform(-0.2, 1.2, (c["P"], K)).astype(np.float32)
    msk = rng.standard_normal((c["proposals_idx"].shape[0], K)).astype(np.float32)
    want = PO.softgroup_instances("scene", c["xyz"], c["proposals_idx"], c["n"], cls, iou, msk, K, 0.001, -0.5, 100)
    got = PP.softgroup_instances("scene", c["xyz"], torch.from_numpy(c["proposals_idx"]), c["n"], torch.from_numpy(cls),
                                 torch.from_numpy(iou), torch.from_numpy(msk), 0, K, 0.001, -0.5, 100)
    assert len(want) > 5
    assert_same_instances(got, want)
