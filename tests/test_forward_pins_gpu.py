"""Whole-model pins, GPU: our PointGroup / HAIS / SoftGroup `forward` + `_loss` on the HIP kernels (through the C ABI)
against what the reference's own model code produced on the same inputs (tests/golden/forward_cases.npz, generated in
the build container by tests/golden/make_golden_forward.py; nothing of /root/reference is read here).
Proposal lists bit-exact, activations / scores / losses within 1e-4 (north_star's bar)."""
import pytest
import torch

import forward_pins as FP
from forward_cases import CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", [c[0] for c in CASES])
def test_forward_and_loss_vs_reference_model_code_hip(tag):
    from minsu3d_amd import backend
    assert backend.get_backend().name == "hip"
    report = FP.check_case(tag, torch.device("cuda", 0), tol=1e-4, head_grad_tol=1e-3)
    print(tag, {k: "%.1e" % v for k, v in report.items()})
