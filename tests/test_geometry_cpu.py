"""Host point-set helpers (no GPU): the wall median voxel against the oracle's plain restatement."""
import numpy as np

from oracle import sia_oracle
from tissue_analysis_amd.geometry import _find_wall_median_voxel, find_wall_median_voxel


def test_docstring_example():                                # SIA:1566-1570
    ar = np.array([[0, 0, 0], [0, 1, 0], [0, 2, 0], [0, 3, 0], [0, 4, 0]])
    assert _find_wall_median_voxel(ar) == 2 == sia_oracle.find_wall_median_voxel(ar)
    assert _find_wall_median_voxel(ar.T) == 2                # a 3 x N array is transposed, as in the reference


def test_random_point_sets_and_ties():
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 7, 50, 100, 101, 300):
        pts = rng.integers(0, 12, size=(n, 3))
        if n == 3:
            continue                                          # a 3 x 3 array is read as 3 x N by both: covered below
        assert _find_wall_median_voxel(pts) == sia_oracle.find_wall_median_voxel(pts), n
    sq = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]])      # all four sums equal: the first index wins
    assert _find_wall_median_voxel(sq) == 0 == sia_oracle.find_wall_median_voxel(sq)
    three = np.array([[0, 5, 9], [0, 0, 0], [1, 1, 1]])
    assert _find_wall_median_voxel(three) == sia_oracle.find_wall_median_voxel(three)


def test_dict_and_array_forms():
    a = np.array([[0, 0, 0], [0, 1, 0], [0, 2, 0], [0, 3, 0], [0, 4, 0]])
    d = {(2, 3): a.T, (1, 4): a[:4].T * 2}
    assert find_wall_median_voxel(d, verbose=False) == {(2, 3): 2, (1, 4): 1}
    assert list(find_wall_median_voxel({(2, 3): a.T}, return_id=False, verbose=False)) == [0, 2, 0]   # one entry: bare value
    assert find_wall_median_voxel(d, labels2exclude=1, verbose=False) == {(2, 3): 2}
    assert find_wall_median_voxel("nope") == "Failed to recognise the type of data."


def test_batched_weiszfeld_equals_the_point_by_point_restatement():
    """geometry.weiszfeld_segments (all walls at once over a grouped point table) against the oracle's loop version of
    SIA:1586-1635, wall by wall -- including the +0.1 nudge of the start and the early stopping rule.  The sums run in the
    reference's order, so symmetric walls -- whose median coordinates ARE integers or half-integers and whose terms are exact
    -- come out with the same doubles (a truncation follows: the last bit decides the voxel there).  Generic walls agree to
    the last bits only: the reference squares through libm's pow(), which is not x * x in ~1e-4 of the cases."""
    from oracle.graph_oracle import weiszfeld
    from tissue_analysis_amd.geometry import geometric_median, median_voxels, weiszfeld_segments
    rng = np.random.default_rng(11)
    sets = [rng.integers(0, 9, size=(n, 3)) for n in (2, 3, 4, 5, 9, 17, 40, 120, 6, 2)]
    sets.append(np.array([[0, 0, 0], [2, 0, 0], [1, 1, 0], [1, -1, 0], [1, 0, 5]]))       # centroid coordinates all occur among the samples: nudged
    sets.append(np.array([[4, 4, 4], [4, 4, 4], [4, 4, 4], [5, 4, 4]]))
    first_symmetric = len(sets)
    g = np.stack(np.meshgrid(np.arange(2), np.arange(3, 8), np.arange(20, 31), indexing="ij"), axis=-1).reshape(-1, 3)
    sets.append(g)                                                            # a flat wall two voxels thick: median (0.5, 5, 25)
    sets.append(g[:, [1, 2, 0]] + 7)
    sets.append(np.concatenate([g, g + [0, 0, 11]]))
    pts = np.concatenate(sets)
    sizes = [len(s) for s in sets]
    got = weiszfeld_segments(pts, sizes)
    for k, s in enumerate(sets):
        want = weiszfeld(np.array(s.T, dtype=float))
        np.testing.assert_allclose(got[k], want, rtol=1e-12, atol=1e-12)
        assert np.array_equal(geometric_median(s.T), got[k])                  # one segment or many: the same arithmetic
        if k >= first_symmetric:
            assert np.array_equal(got[k], want), (k, got[k] - want)
    chosen = median_voxels(pts, sizes)
    for k, s in enumerate(sets):
        w = weiszfeld(np.array(s.T, dtype=float))
        if k < first_symmetric and np.abs(w - np.round(w)).min() < 1e-9:
            continue                                                          # truncation of a value at an integer, generic wall: either side
        d = ((s - np.trunc(w)) ** 2).sum(axis=1)
        assert tuple(chosen[k]) == tuple(s[int(np.argmin(d))])
    assert weiszfeld_segments(np.zeros((0, 3)), []).shape == (0, 3)
