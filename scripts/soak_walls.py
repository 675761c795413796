"""Randomised check of the wall-voxel kernels against a numpy brute force (GPU box; not part of the test suite):

    PYTHONPATH=. python scripts/soak_walls.py [examples] [seed]

The brute force: for each of the 18 offsets of scipy's generate_binary_structure(3, 2), every voxel whose neighbour at that
offset carries another label gives a record (lo, hi, voxel); the distinct records, in memory order, are what
ta_wall_voxels_get returns -- and, stably sorted by pair, what ta_wall_voxels_get_by_pair returns."""
import os
import sys
import numpy as np

from tissue_analysis_amd.extraction import ResidentVolume

OFFSETS = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if 0 < abs(a) + abs(b) + abs(c) < 3]


def brute(vol):
    n0, n1, n2 = vol.shape
    idx = np.arange(vol.size, dtype=np.int64).reshape(vol.shape)
    recs = []
    for a, b, c in OFFSETS:
        src = (slice(max(0, -a), n0 - max(0, a)), slice(max(0, -b), n1 - max(0, b)), slice(max(0, -c), n2 - max(0, c)))
        dst = (slice(max(0, a), n0 - max(0, -a)), slice(max(0, b), n1 - max(0, -b)), slice(max(0, c), n2 - max(0, -c)))
        v, m, i = vol[src].astype(np.int64), vol[dst].astype(np.int64), idx[src]
        hit = v != m
        recs.append(np.stack([i[hit], np.minimum(v, m)[hit], np.maximum(v, m)[hit]], axis=1))
    r = np.unique(np.concatenate(recs), axis=0) if recs else np.zeros((0, 3), np.int64)
    return r                                             # sorted by (voxel, lo, hi)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
for it in range(n):
    dtype = [np.uint16, np.uint32][rng.integers(0, 2)]
    shape = (int(rng.integers(1, 20)), int(rng.integers(1, 40)), int(rng.choice([1, 2, 3, 4, 7, 8, 64, 130, 255, 256, 257, 300, 512, 516, 770])))
    nlab = int(rng.integers(1, 12))
    top = 65536 if dtype == np.uint16 else [1 << 22, 1 << 32][rng.integers(0, 2)]          # uint32: half the volumes hold labels >= 2^31
    ids = np.unique(rng.integers(0, top, size=nlab)).astype(dtype)
    staged = [None, 0, 256 * 8, 256 * 64][rng.integers(0, 4)]          # the staging area: default / none / regions of 8 / of 64 records
    if staged is None:
        os.environ.pop("TA_WALL_STAGE_RECORDS", None)
    else:
        os.environ["TA_WALL_STAGE_RECORDS"] = str(staged)
    block = (int(rng.integers(1, 5)), int(rng.integers(1, 7)), int(rng.integers(1, 30)))
    coarse = [int(np.ceil(s / b)) for s, b in zip(shape, block)]
    v = ids[rng.integers(0, ids.size, size=coarse)]
    for ax, b in enumerate(block):
        v = np.repeat(v, b, axis=ax)
    vol = np.ascontiguousarray(v[:shape[0], :shape[1], :shape[2]])
    if rng.random() < 0.3:
        k = int(vol.size * 0.03) + 1
        vol.flat[rng.integers(0, vol.size, size=k)] = ids[rng.integers(0, ids.size, size=k)]
    if it % 25 == 0:
        print("it=%d" % it, flush=True)
    want = brute(vol)
    rv = ResidentVolume(vol)
    try:
        lo, hi, coords, _ = rv.ctx.wall_voxels()
        lin = (coords[:, 0].astype(np.int64) * shape[1] + coords[:, 1]) * shape[2] + coords[:, 2]
        got = np.stack([lin, lo.astype(np.int64), hi.astype(np.int64)], axis=1)
        ok = got.shape == want.shape and np.all(np.diff(lin) >= 0)
        if ok:                                           # same voxel: the order of its pairs is free
            g = got[np.lexsort((got[:, 2], got[:, 1], got[:, 0]))]
            ok = np.array_equal(g, want)
        glo, ghi, gco, _ = rv.ctx.wall_voxels(by_pair=True)
        glin = (gco[:, 0].astype(np.int64) * shape[1] + gco[:, 1]) * shape[2] + gco[:, 2]
        w2 = want[np.lexsort((want[:, 0], want[:, 2], want[:, 1]))]           # by (lo, hi, voxel)
        ok2 = glin.shape[0] == w2.shape[0] and np.array_equal(np.stack([glin, glo.astype(np.int64), ghi.astype(np.int64)], axis=1), w2)
    finally:
        rv.close()
    if not (ok and ok2):
        print("MISMATCH it=%d shape=%s dtype=%s block=%s nlab=%d memory-order=%s by-pair=%s" % (it, shape, vol.dtype.name, block, ids.size, ok, ok2))
        sys.exit(1)
print("wall soak ok: %d random volumes equal to the brute force" % n)
