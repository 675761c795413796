"""inertia_axis through the HIP path against closed forms (cuboids, labels of 1 and 2 voxels, a diagonal line): second
moments from the sweep, covariance + eigen-decomposition on the host.  The same cases pin the oracle on the CPU
(tests/test_analytic_inertia_cpu.py)."""
import numpy as np
import pytest

import analytic_inertia
from tissue_analysis_amd import DICT, NPLIST, SpatialImage, SpatialImageAnalysis

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.uint16, np.uint32])
@pytest.mark.parametrize("real", [False, True])
def test_closed_form_inertia_through_the_sweep(dtype, real):
    vol, cases = analytic_inertia.build(dtype)
    sia = SpatialImageAnalysis(SpatialImage(vol, voxelsize=analytic_inertia.VOXELSIZE), ignoredlabels=0, return_type=DICT,
                               background=1)
    analytic_inertia.check(sia, cases, real)
    vols = sia.volume(sorted(cases), real=False)
    assert all(vols[l] == cases[l]["count"] for l in cases)


def test_array_mode_gives_the_same_numbers():
    vol, cases = analytic_inertia.build()
    img = SpatialImage(vol, voxelsize=analytic_inertia.VOXELSIZE)
    by_dict = SpatialImageAnalysis(img, ignoredlabels=0, return_type=DICT, background=1)
    arrays = SpatialImageAnalysis(img, ignoredlabels=0, return_type=NPLIST, background=1)
    labels = sorted(cases)
    axes, values = arrays.inertia_axis(list(labels), True)
    d_axes, d_values = by_dict.inertia_axis(list(labels), True)
    assert axes.shape == (len(labels), 3, 3) and values.shape == (len(labels), 3)
    for k, l in enumerate(labels):
        assert np.array_equal(values[k], d_values[l]) and np.array_equal(axes[k], np.asarray(d_axes[l]))
    com = arrays.center_of_mass(list(labels))
    assert com.shape == (len(labels), 3) and all(np.array_equal(com[k], by_dict.center_of_mass(list(labels))[l]) for k, l in enumerate(labels))
    box = arrays.boundingbox(list(labels))
    assert box.shape == (len(labels), 6)
    for k, l in enumerate(labels):
        sl = by_dict.boundingbox(l)
        assert [s.start for s in sl] + [s.stop for s in sl] == box[k].tolist()
    deg = arrays.neighbors_number(list(labels))
    assert deg.tolist() == [len(by_dict.neighbors(l)) for l in labels]
    pairs, area = arrays.wall_areas()
    want = by_dict.wall_areas()
    assert len(want) == len(area) and all(want[(int(a), int(b))] == v for (a, b), v in zip(pairs, area))
    nei = arrays.neighbors()
    assert len(nei) == len(by_dict.neighbors())                       # (keys by position under NPLIST, SIA:642-645)
    assert nei.indptr[-1] == nei.indices.size and nei[2] == by_dict.extraction.neighbors_of(2)


@pytest.mark.parametrize("dtype", [np.uint16, np.uint32])
def test_closed_form_volumes_boxes_barycentres_neighbours_and_walls_through_the_sweep(dtype):
    """The box tiled by 27 staggered cuboids (tests/analytic_shapes.py): every per-label and per-pair answer of the class,
    from the HIP sweep, against closed forms -- and the sweep's integer arrays against the closed-form face counts."""
    import analytic_shapes
    vol, cells = analytic_shapes.build(dtype)
    sia = SpatialImageAnalysis(SpatialImage(vol, voxelsize=analytic_shapes.VOXELSIZE), ignoredlabels=0, return_type=DICT,
                               background=1)
    analytic_shapes.check(sia, cells)
    want = analytic_shapes.expected(cells)
    x = sia.extraction
    got = dict(zip(zip(x.pair_lo.tolist(), x.pair_hi.tolist()), x.pair_faces.tolist()))
    assert sorted(got) == sorted(want["faces"])
    assert all(got[k] == f.tolist() for k, f in want["faces"].items())
