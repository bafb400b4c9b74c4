"""Dataset + collate pinned to the reference's own classes, GPU: sparse_quantize and the elastic distortion run in the
HIP library (host-drawn noise, so the numpy seed reproduces the reference's augmentation)."""
import pytest
import torch

import dataset_pins as DP

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("device_elastic", [False, True])
def test_dataset_and_collate_vs_reference_classes_hip(tmp_path, device_elastic):
    from minsu3d_amd import backend
    be = backend.get_backend()
    assert be.name == "hip"
    fn = None
    if device_elastic:
        def fn(x, noise, gran, mag):
            import numpy as np
            return be.elastic(torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(noise), gran, mag).cpu().numpy()
    assert DP.check(tmp_path, torch.device("cuda", 0), fn) == 22
