"""Module-level helpers of the drop-in module (SURVEY.md §2 component 2; SIA:123-188): the float helpers against the oracle's
restatement and against closed forms (the two voxel-work helpers, wall / contact_surface, run on the GPU: test_gpu_module_helpers.py)."""
import numpy as np
import pytest

from oracle.sia_oracle import covariance_and_axes
import tissue_analysis_amd as ta
from tissue_analysis_amd import (compute_covariance_matrix, coordinates_centering3D, distance, eigen_values_vectors)


def test_names_are_exported_by_the_drop_in_module():
    from tissue_analysis_amd import spatial_image_analysis as sia
    for name in ("wall", "contact_surface", "coordinates_centering3D", "compute_covariance_matrix", "eigen_values_vectors",
                 "distance", "dilation", "dilation_by", "real_indices", "return_list_of_vectors"):
        assert callable(getattr(sia, name)) and callable(getattr(ta, name))


def test_centering_both_layouts_and_given_mean():
    rng = np.random.default_rng(3)
    pts = rng.integers(0, 50, size=(3, 40)).astype(float)
    c = coordinates_centering3D(pts)
    assert c.shape == (3, 40) and np.allclose(c.mean(axis=1), 0.0, atol=1e-12)
    assert np.allclose(c, pts - pts.mean(axis=1, keepdims=True))
    assert np.allclose(coordinates_centering3D(pts.T), c)                      # N x 3 goes through the transpose
    assert np.allclose(coordinates_centering3D(pts, mean=[1.0, 2.0, 3.0]), pts - np.array([[1.0], [2.0], [3.0]]))


def test_covariance_and_eigen_pairs_match_the_oracle_restatement():
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 4, 17, 200):
        pts = rng.normal(size=(3, n)) * np.array([[5.0], [2.0], [0.5]])
        c = coordinates_centering3D(pts)
        cov = compute_covariance_matrix(c)
        want_cov, want_val, want_vec = covariance_and_axes(c)
        assert np.allclose(cov, want_cov, rtol=1e-12, atol=1e-14)
        assert np.allclose(cov, np.dot(c, c.T) / max(3, n))                     # the 1 / max(shape) normaliser of SIA:150
        val, vec = eigen_values_vectors(cov)
        assert np.all(np.diff(val) <= 1e-12) and np.allclose(val, want_val, rtol=1e-9, atol=1e-12)
        for k in range(3):                                                      # eigenvectors by ROWS, up to sign
            assert np.allclose(cov @ vec[k], val[k] * vec[k], atol=1e-9)
            if n >= 17:
                assert abs(abs(float(np.dot(vec[k], want_vec[k]))) - 1.0) < 1e-9
    assert np.allclose(compute_covariance_matrix(np.ones((40, 3))), np.full((3, 3), 1.0))      # N x 3 input is transposed


def test_cuboid_closed_form():
    a, b, c = 7, 4, 2
    pts = np.array(np.meshgrid(np.arange(a), np.arange(b), np.arange(c), indexing="ij")).reshape(3, -1).astype(float)
    val, vec = eigen_values_vectors(compute_covariance_matrix(coordinates_centering3D(pts)))
    assert np.allclose(val, [(a * a - 1) / 12.0, (b * b - 1) / 12.0, (c * c - 1) / 12.0])
    assert np.allclose(np.abs(vec), np.eye(3), atol=1e-12)


def test_distance():
    assert distance([0, 0], [3, 4]) == 5.0 and abs(distance((1, 2, 3), (2, 4, 5)) - 3.0) < 1e-12
    with pytest.warns(UserWarning):
        assert distance([0, 0], [1, 2, 3]) is None
