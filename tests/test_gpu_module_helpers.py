"""wall / contact_surface of the drop-in module (SIA:45-60) through the HIP stencil kernel, against the reference's own
formulation (binary dilation of the cell minus the cell, scipy) and the oracle's restatement."""
import numpy as np
import pytest
import scipy.ndimage as nd

from oracle.sia_oracle import wall_labels
from tissue_analysis_amd import contact_surface, wall

from helpers import voronoi

pytestmark = pytest.mark.gpu


def reference_wall(mask_img, label_id):          # the reference's three lines, as the survey's semantics state them
    img = (mask_img == label_id)
    contact = nd.binary_dilation(img) & ~img
    return mask_img[contact]


@pytest.mark.parametrize("dtype", [np.uint16, np.uint32, np.int64])
def test_wall_and_contact_surface_match_the_reference_formulation(dtype):
    vol = voronoi((20, 24, 40), 14, 3, np.uint16, ellipsoid=True).astype(dtype)
    vol[3:6, 3:6, 3:6] = 0                        # a label-0 pocket: shell voxels labelled 0 must come back as 0
    for label in (int(vol[10, 12, 20]), int(vol[4, 6, 4]), 1, 0, 9999):
        got = wall(vol, label)
        want = reference_wall(vol, label)
        assert got.dtype == vol.dtype and np.array_equal(got, want), label
        assert contact_surface(vol, label) == set(np.unique(want).tolist()) == wall_labels(vol, label)


def test_wall_on_the_docstring_image_and_in_2d():
    a = np.array([[1, 2, 7, 7, 1, 1], [1, 6, 5, 7, 3, 3], [2, 2, 1, 7, 3, 3], [1, 1, 1, 4, 1, 1]], dtype=np.uint16)
    assert contact_surface(a, 7) == {1, 2, 3, 4, 5}                   # (SIA:553-574: neighbors(7) of the docstring image)
    assert np.array_equal(wall(a, 7), reference_wall(a, 7))
    f = np.asfortranarray(voronoi((9, 11, 13), 6, 4, np.uint16))
    lab = int(f[4, 5, 6])
    assert np.array_equal(wall(f, lab), reference_wall(f, lab))       # C index order whatever the memory layout
