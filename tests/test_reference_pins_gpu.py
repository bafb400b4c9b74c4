"""GPU side of the pins to the reference's own Python model code (tests/golden/model_*.{json,npz}): the HIP kernels
behind OUR modules reproduce what the reference's module classes computed -- activations within 1e-4, instance lists
(masks, labels, boxes) exactly."""
import pytest

import reference_pins as RP

pytestmark = pytest.mark.gpu


def test_forward_composition_hip_vs_reference_modules():
    worst = RP.check_backbone("cuda", 1e-4)      # north_star: <= 1e-4 relative on sparse-conv activations
    print("worst relative error vs the reference's composition: %.2e" % worst)


def test_pred_instances_hip_vs_reference_methods():
    assert RP.check_instances("cuda") > 500


def test_losses_hip_vs_reference():
    RP.check_losses("cuda")
