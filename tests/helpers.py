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


WALL_OFFSETS = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if 0 < abs(a) + abs(b) + abs(c) < 3]


def brute_wall_records(vol):
    """Wall-voxel records of every label pair by brute force over the 18 offsets of generate_binary_structure(3, 2):
    (lo, hi, coords[n, 3]) sorted by (lo, hi, voxel in memory order) -- what ta_wall_voxels_get_by_pair returns."""
    n0, n1, n2 = vol.shape
    idx = np.arange(vol.size, dtype=np.int64).reshape(vol.shape)
    recs = []
    for a, b, c in WALL_OFFSETS:
        src = (slice(max(0, -a), n0 - max(0, a)), slice(max(0, -b), n1 - max(0, b)), slice(max(0, -c), n2 - max(0, c)))
        dst = (slice(max(0, a), n0 - max(0, -a)), slice(max(0, b), n1 - max(0, -b)), slice(max(0, c), n2 - max(0, -c)))
        v, m, i = vol[src].astype(np.int64), vol[dst].astype(np.int64), idx[src]
        hit = v != m
        recs.append(np.stack([np.minimum(v, m)[hit], np.maximum(v, m)[hit], i[hit]], axis=1))
    r = np.unique(np.concatenate(recs), axis=0)
    coords = np.stack(np.unravel_index(r[:, 2], vol.shape), axis=1).astype(np.int32)
    return r[:, 0].astype(np.uint32), r[:, 1].astype(np.uint32), coords


# ---- the C one-pass oracle on a volume too big for one process to finish in seconds: Z-slabs on forked workers + merge
_PAR_VOLUME = None


def _par_slab(args):
    from oracle import onepass_c
    lo, hi, max_label = args
    halo = 1 if lo > 0 else 0                 # (a face belongs to the slab that owns its higher voxel along axis 0)
    return onepass_c.extract(_PAR_VOLUME[lo - halo:hi], max_label=max_label, origin=(lo - halo, 0, 0), own_first_plane=not halo)


def onepass_c_parallel(vol, max_label, workers=16):
    """oracle/onepass_c over `workers` Z-slabs (one low halo plane each) in forked processes -- the voxels are inherited, not
    pickled -- merged with oracle/onepass.merge: the whole-volume result (tests prove slab + merge == unsharded at small sizes)."""
    import multiprocessing as mp
    from oracle import onepass, onepass_c
    global _PAR_VOLUME
    onepass_c.build()
    n0 = vol.shape[0]
    cuts = [n0 * i // workers for i in range(workers + 1)]
    jobs = [(a, b, max_label) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    _PAR_VOLUME = vol
    try:
        with mp.get_context("fork").Pool(len(jobs)) as pool:
            parts = pool.map(_par_slab, jobs)
    finally:
        _PAR_VOLUME = None
    return onepass.merge(parts)
