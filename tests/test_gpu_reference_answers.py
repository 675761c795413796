"""The reference's OWN known answers, through the HIP path.

tissue_analysis holds no tests; the only numbers it pins for this path are the docstring examples on one 4x6
image (spatial_image_analysis.py, lines cited per check).  They are typed in here by hand and asserted on the
drop-in class running on the GPU -- not on the oracle (tests/test_oracle_known_answers.py does that on the CPU).
Fixture: tests/golden/docstring_4x6.json (the same numbers as data, written by tests/golden/gen_golden.py)."""
import json
import os

import numpy as np
import pytest

from tissue_analysis_amd import DICT, LIST, NPLIST, SpatialImageAnalysis

pytestmark = pytest.mark.gpu

A = np.array([[1, 2, 7, 7, 1, 1],
              [1, 6, 5, 7, 3, 3],
              [2, 2, 1, 7, 3, 3],
              [1, 1, 1, 4, 1, 1]], dtype=np.uint16)       # SIA:344-347
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "docstring_4x6.json")


@pytest.fixture()
def sia():
    return SpatialImageAnalysis(A)


def test_labels_and_count(sia):                              # SIA:352-353, 381-382
    assert sia.labels() == [1, 2, 3, 4, 5, 6, 7]
    assert sia.nb_labels() == 7


def test_center_of_mass(sia):                                # SIA:437-450
    np.testing.assert_allclose(sia.center_of_mass(7), [0.75, 2.75, 0.0], rtol=1e-6)
    two = sia.center_of_mass([7, 2])
    np.testing.assert_allclose(two[7], [0.75, 2.75, 0.0], rtol=1e-6)
    np.testing.assert_allclose(two[2], [1.3333333333333333, 0.66666666666666663, 0.0], rtol=1e-6)
    want = {1: [1.8, 2.2999999999999998, 0.0], 2: [1.3333333333333333, 0.66666666666666663, 0.0],
            3: [1.5, 4.5, 0.0], 4: [3.0, 3.0, 0.0], 5: [1.0, 2.0, 0.0], 6: [1.0, 1.0, 0.0],
            7: [0.75, 2.75, 0.0]}
    got = sia.center_of_mass()
    assert sorted(got) == sorted(want)
    for l in want:
        np.testing.assert_allclose(got[l], want[l], rtol=1e-6)


def test_boundingbox(sia):                                   # SIA:498-511
    assert sia.boundingbox(7) == (slice(0, 3), slice(2, 4), slice(0, 1))
    two = sia.boundingbox([7, 2])
    assert two[7] == (slice(0, 3), slice(2, 4), slice(0, 1))
    assert two[2] == (slice(0, 3), slice(0, 2), slice(0, 1))
    want = [(slice(0, 4), slice(0, 6), slice(0, 1)), (slice(0, 3), slice(0, 2), slice(0, 1)),
            (slice(1, 3), slice(4, 6), slice(0, 1)), (slice(3, 4), slice(3, 4), slice(0, 1)),
            (slice(1, 2), slice(2, 3), slice(0, 1)), (slice(1, 2), slice(1, 2), slice(0, 1)),
            (slice(0, 3), slice(2, 4), slice(0, 1))]
    got = sia.boundingbox()
    assert [got[l] for l in range(1, 8)] == want


def test_neighbors(sia):                                     # SIA:561-574
    assert sorted(sia.neighbors(7)) == [1, 2, 3, 4, 5]
    two = sia.neighbors([7, 2])
    assert sorted(two[7]) == [1, 2, 3, 4, 5] and sorted(two[2]) == [1, 6, 7]
    want = {1: [2, 3, 4, 5, 6, 7], 2: [1, 6, 7], 3: [1, 7], 4: [1, 7], 5: [1, 6, 7], 6: [1, 2, 5],
            7: [1, 2, 3, 4, 5]}
    got = sia.neighbors()
    assert dict((k, sorted(v)) for k, v in got.items()) == want


def test_cell_wall_area(sia):                                # SIA:924-927
    assert sia.cell_wall_area(7, 2) == 1.0
    assert sia.cell_wall_area(7, [2, 5]) == {(2, 7): 1.0, (5, 7): 2.0}


def test_wall_areas(sia):                                    # SIA:978-982
    assert sia.wall_areas({1: [2, 3], 2: [6]}) == {(1, 2): 5.0, (1, 3): 4.0, (2, 6): 2.0}
    want = {(1, 2): 5.0, (1, 3): 4.0, (1, 4): 2.0, (1, 5): 1.0, (1, 6): 1.0, (1, 7): 2.0, (2, 6): 2.0,
            (2, 7): 1.0, (3, 7): 2, (4, 7): 1, (5, 6): 1.0, (5, 7): 2.0}
    assert sia.wall_areas() == want


def test_volume(sia):                                        # SIA:1219-1226
    assert list(sia.volume(7).values()) == [4.0] if isinstance(sia.volume(7), dict) else sia.volume(7) == 4.0
    v = sia.volume([7, 2])
    assert v[7] == 4.0 and v[2] == 3.0
    allv = sia.volume()
    assert [allv[l] for l in range(1, 8)] == [10.0, 3.0, 4.0, 1.0, 1.0, 1.0, 4.0]


def test_return_types():                                     # SIA:309-334
    assert SpatialImageAnalysis(A, return_type=LIST).volume() == [10.0, 3.0, 4.0, 1.0, 1.0, 1.0, 4.0]
    np.testing.assert_array_equal(SpatialImageAnalysis(A, return_type=NPLIST).volume(), [10, 3, 4, 1, 1, 1, 4])
    assert isinstance(SpatialImageAnalysis(A, return_type=DICT).volume(), dict)


def test_wall_median_voxel_docstring(sia):                   # SIA:1566-1570: the exact medoid of <= 100 points
    from tissue_analysis_amd.spatial_image_analysis import _find_wall_median_voxel, find_wall_median_voxel
    ar = np.array([[0, 0, 0], [0, 1, 0], [0, 2, 0], [0, 3, 0], [0, 4, 0]])
    assert _find_wall_median_voxel(ar) == 2
    assert find_wall_median_voxel(ar) == 2 and list(find_wall_median_voxel(ar, return_id=False)) == [0, 2, 0]


def test_committed_fixture_is_these_numbers(sia):
    """SURVEY.md §8(c) golden item (1): the docstring case as a data fixture, all methods."""
    gold = json.load(open(GOLD))
    assert gold["image"] == A.tolist()
    assert sia.labels() == gold["labels"]
    vol, com, bb = sia.volume(), sia.center_of_mass(), sia.boundingbox()
    for l in gold["labels"]:
        assert vol[l] == gold["volume"][str(l)]
        np.testing.assert_allclose(com[l], gold["center_of_mass"][str(l)], rtol=1e-6)
        assert [[s.start, s.stop] for s in bb[l]] == gold["boundingbox"][str(l)]
        assert sorted(sia.neighbors(l)) == gold["neighbors"][str(l)]
    walls = sia.wall_areas()
    assert dict(("%d,%d" % k, v) for k, v in walls.items()) == gold["wall_areas"]
