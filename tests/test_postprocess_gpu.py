"""GPU parity of the instance post-processing (through the C ABI): cross intersections and greedy NMS bit-exact vs
the dense-mask restatement of the reference, and the assembled instance lists."""
import numpy as np
import pytest
import torch

from oracle import postprocess_oracle as PO
from postprocess_cases import assert_same_instances, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from minsu3d_amd.backend import HipBackend
    return HipBackend()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("seed", [0, 1])
def test_kernels_bit_exact(be, seed):
    c = make_case(seed, n=20000, n_regions=30, per_region=8, junk=20)
    key = np.unique(c["proposals_idx"][:, 0].astype(np.int64) * c["n"] + c["proposals_idx"][:, 1])
    cl, pt = key // c["n"], key % c["n"]
    o = np.argsort(pt * c["P"] + cl)
    want = PO.cross_intersection(pt[o], cl[o], c["P"])
    got = be.proposal_cross_intersection(dev(pt[o]), dev(cl[o]), c["P"])
    assert np.array_equal(got.cpu().numpy(), want)
    order = np.argsort(-c["scores"], kind="stable")
    for thr in (0.1, 0.3, 0.7):
        assert np.array_equal(be.nms_greedy(got, dev(order), thr).cpu().numpy(), PO.nms_from_counts(want, order, thr))
    # empty inputs
    assert be.nms_greedy(torch.zeros((0, 0), dtype=torch.int32, device="cuda"), torch.zeros(0, dtype=torch.int32, device="cuda"), 0.3).numel() == 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_instances_hip_vs_dense_restatement(be, seed):
    from minsu3d_amd.model import postprocess as PP
    c = make_case(seed)
    want = PO.pointgroup_instances("scene", c["xyz"], c["scores"], c["proposals_idx"], c["P"], c["sem"], 2, 0.09, 100, 0.3)
    got = PP.pointgroup_instances("scene", c["xyz"], dev(c["scores"]), dev(c["proposals_idx"]), c["P"], dev(c["sem"]), 2,
                                  0.09, 100, 0.3)
    assert len(want) > 3
    assert_same_instances(got, want)
    want = PO.hais_instances("scene", c["xyz"], c["scores"], c["proposals_idx"], c["P"], c["mask_scores"], c["sem"], 2,
                             -0.5, 0.09, 100)
    got = PP.hais_instances("scene", c["xyz"], dev(c["scores"]), dev(c["proposals_idx"]), c["P"], dev(c["mask_scores"]),
                            dev(c["sem"]), 2, -0.5, 0.09, 100)
    assert_same_instances(got, want)
