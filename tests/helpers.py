"""Shared test helpers: synthetic label volumes and comparison of integer accumulators."""
import numpy as np

from tissue_analysis_amd import synth

INT_KEYS = ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces")


def voronoi(dims, n_cells, seed, dtype=np.uint16, ellipsoid=True):
    return synth.voronoi_labels(dims, n_cells, seed, dtype, ellipsoid=ellipsoid)


def random_blocks(dims, n_labels, seed, dtype=np.uint16, block=(3, 2, 5), with_zero=True):
    """Blocky random labelling: many small regions, labels from a sparse id set (absent ids,
    label 0, ids above 32767), regions touching every border."""
    rng = np.random.default_rng(seed)
    ids = np.unique(rng.integers(1 if not with_zero else 0, np.iinfo(dtype).max // 2 if dtype == np.uint16 else 200000,
                                 size=n_labels)).astype(dtype)
    coarse = [int(np.ceil(d / b)) for d, b in zip(dims, block)]
    c = rng.integers(0, ids.size, size=coarse)
    v = ids[c]
    for ax, b in enumerate(block):
        v = np.repeat(v, b, axis=ax)
    return np.ascontiguousarray(v[:dims[0], :dims[1], :dims[2]])


def assert_same_accumulators(got, want, what=""):
    for k in INT_KEYS:
        g, w = np.asarray(got[k]), np.asarray(want[k])
        assert g.shape == w.shape, "%s %s: shape %s != %s" % (what, k, g.shape, w.shape)
        if not np.array_equal(g, w):
            bad = np.argwhere(g != w)
            raise AssertionError("%s %s differs at %d entries, first %s: got %s want %s" % (
                what, k, len(bad), bad[0], g[tuple(bad[0])], w[tuple(bad[0])]))


def extraction_arrays(x):
    return x.as_arrays()
