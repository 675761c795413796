"""Pin the oracle against every known answer the reference holds for this path: the docstring
examples on one 4x6 image (spatial_image_analysis.py, lines cited per check).  The reference has no
other tests or fixtures (test/__init__.py:1-11)."""
import numpy as np
import pytest

from oracle import onepass
from oracle.sia_oracle import DICT, LIST, NPLIST, OracleSIA

A = np.array([[1, 2, 7, 7, 1, 1],
              [1, 6, 5, 7, 3, 3],
              [2, 2, 1, 7, 3, 3],
              [1, 1, 1, 4, 1, 1]], dtype=np.uint16)       # SIA:344-347


@pytest.fixture()
def sia():
    return OracleSIA(A)


def test_labels_and_count(sia):                              # SIA:352-353, 381-382
    assert sia.labels() == [1, 2, 3, 4, 5, 6, 7]
    assert sia.nb_labels() == 7


def test_center_of_mass(sia):                                # SIA:437-450
    np.testing.assert_allclose(sia.center_of_mass(7), [0.75, 2.75, 0.0])
    two = sia.center_of_mass([7, 2])
    np.testing.assert_allclose(two[7], [0.75, 2.75, 0.0])
    np.testing.assert_allclose(two[2], [1.3333333333333333, 0.66666666666666663, 0.0])
    want = {1: [1.8, 2.2999999999999998, 0.0], 2: [1.3333333333333333, 0.66666666666666663, 0.0],
            3: [1.5, 4.5, 0.0], 4: [3.0, 3.0, 0.0], 5: [1.0, 2.0, 0.0], 6: [1.0, 1.0, 0.0],
            7: [0.75, 2.75, 0.0]}
    got = sia.center_of_mass()
    assert sorted(got) == sorted(want)
    for l in want:
        np.testing.assert_allclose(got[l], want[l], rtol=1e-15)


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
    assert list(sia.volume(7).values()) == [4.0]
    v = sia.volume([7, 2])
    assert v[7] == 4.0 and v[2] == 3.0
    allv = sia.volume()
    assert [allv[l] for l in range(1, 8)] == [10.0, 3.0, 4.0, 1.0, 1.0, 1.0, 4.0]


def test_return_types():                                     # SIA:309-334
    assert OracleSIA(A, return_type=LIST).volume() == [10.0, 3.0, 4.0, 1.0, 1.0, 1.0, 4.0]
    np.testing.assert_array_equal(OracleSIA(A, return_type=NPLIST).volume(), [10, 3, 4, 1, 1, 1, 4])
    assert isinstance(OracleSIA(A, return_type=DICT).volume(), dict)


def test_onepass_reproduces_the_docstring_numbers():
    """The integer one-pass spec gives the same volumes, boxes, neighbours and wall areas."""
    r = onepass.extract(A[:, :, None])
    assert r["count"].tolist() == [0, 10, 3, 4, 1, 1, 1, 4]
    assert r["bbox"][7].tolist() == [0, 2, 0, 3, 4, 1]
    np.testing.assert_allclose(r["sum1"][7] / r["count"][7], [0.75, 2.75, 0.0])
    walls = dict(((int(a), int(b)), float(f.sum())) for a, b, f in zip(r["pair_lo"], r["pair_hi"], r["pair_faces"]))
    assert walls == {(1, 2): 5.0, (1, 3): 4.0, (1, 4): 2.0, (1, 5): 1.0, (1, 6): 1.0, (1, 7): 2.0,
                     (2, 6): 2.0, (2, 7): 1.0, (3, 7): 2.0, (4, 7): 1.0, (5, 6): 1.0, (5, 7): 2.0}
