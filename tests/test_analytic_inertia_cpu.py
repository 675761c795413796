"""inertia_axis against CLOSED FORMS (cuboids, 1- and 2-voxel labels, a diagonal line): the oracle's restatement of
SIA:1246-1292 and the product's host code (on integer accumulators injected as the GPU delivers them) must both give
them -- the reference holds no vector for this method, so this is its reference-independent pin."""
import numpy as np
import pytest

import analytic_inertia
from oracle import onepass, sia_oracle
from oracle.sia_oracle import OracleSIA
from tissue_analysis_amd import DICT, Extraction, SpatialImage, SpatialImageAnalysis3D


@pytest.mark.parametrize("real", [False, True])
def test_oracle_gives_the_closed_forms(real):
    vol, cases = analytic_inertia.build()
    ref = OracleSIA(vol, ignoredlabels=0, return_type=sia_oracle.DICT, background=1, voxelsize=analytic_inertia.VOXELSIZE)
    analytic_inertia.check(ref, cases, real)


@pytest.mark.parametrize("real", [False, True])
def test_host_code_on_exact_accumulators_gives_the_closed_forms(real):
    vol, cases = analytic_inertia.build()
    x = Extraction.from_arrays(vol.shape, onepass.extract(vol))
    sia = SpatialImageAnalysis3D(SpatialImage(vol, voxelsize=analytic_inertia.VOXELSIZE), ignoredlabels=0, return_type=DICT,
                                 background=1, extraction=x)
    analytic_inertia.check(sia, cases, real)
    for l, c in cases.items():
        assert int(x.count[l]) == c["count"]


def test_batched_jacobi_solver_against_lapack():
    from tissue_analysis_amd.extraction import sym3_eig
    rng = np.random.default_rng(2)
    m = rng.normal(size=(500, 3, 3)) * rng.uniform(1e-3, 1e3, size=(500, 1, 1))
    m = m + m.transpose(0, 2, 1)
    m[0] = np.diag([3.0, 2.0, 1.0]); m[1] = 0.0; m[2] = np.ones((3, 3)); m[3] = np.diag([1e-30, 0.0, 5e8])
    m[4] = [[2, 1, 0], [1, 2, 0], [0, 0, 5]]
    val, vec = sym3_eig(m)
    scale = np.maximum(np.abs(m).reshape(-1, 9).max(axis=1), 1e-300)
    assert (np.abs(np.sort(val, axis=1) - np.linalg.eigvalsh(m)).max(axis=1) / scale).max() < 1e-14
    assert np.abs(np.einsum('nij,nik->njk', vec, vec) - np.eye(3)).max() < 1e-14
    assert (np.abs(np.einsum('nij,nj,nkj->nik', vec, val, vec) - m).reshape(-1, 9).max(axis=1) / scale).max() < 1e-14
    assert np.array_equal(val[0], [3.0, 2.0, 1.0]) and np.array_equal(vec[0], np.eye(3))      # diagonal input: untouched
    assert sym3_eig(np.zeros((0, 3, 3)))[0].shape == (0, 3)
