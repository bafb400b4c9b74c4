"""Augmentation helpers against golden vectors produced by the reference's own
minsu3d/util/transform.py (tests/golden/make_golden_transform.py), same numpy seeds."""
import os

import numpy as np
import pytest

from minsu3d_amd.util import transform as T

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transform_cases.npz"))


def test_matrices_consume_the_same_random_numbers():
    np.random.seed(11)
    assert np.array_equal(T.jitter(), G["jitter"])
    assert np.array_equal(np.stack([T.flip(0, random=True) for _ in range(6)]), G["flip"])
    assert np.array_equal(T.rotz(1.2345), G["rotz"])


def test_elastic_distortion_matches_reference():
    scale = 50.0
    np.random.seed(21)
    e1 = T.elastic(G["elastic_in"] * scale, 6 * scale // 50, 40 * scale / 50)
    e2 = T.elastic(e1, 20 * scale // 50, 160 * scale / 50)
    assert e1.dtype == np.float64
    # voxel units; the result is floor()-quantised afterwards, 1e-9 voxels is far below any boundary effect
    assert np.abs(e1 - G["elastic_out1"]).max() < 1e-9
    assert np.abs(e2 - G["elastic_out2"]).max() < 1e-9


def test_crop_matches_reference():
    np.random.seed(31)
    off, valid = T.crop(G["crop_in"], 6000, 512)
    assert np.array_equal(valid, G["crop_valid"]) and np.array_equal(off, G["crop_out"])
    assert np.count_nonzero(valid) <= 6000
