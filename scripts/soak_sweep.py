"""One long randomised comparison of the sweep with the C oracle (GPU box; not part of the test suite):

    PYTHONPATH=. python scripts/soak_sweep.py [examples] [seed] [progress file] [first case]

Random shapes / dtypes / label sets / layouts / tile heights (0 = the default) / feature masks, bigger and blockier
than tests/test_gpu_property.py draws them.  Prints the first mismatch and exits 1, or a summary.
TA_SOAK_SHAPE=0|1 forces a tile shape of the uint32 sweep (default: the context's own choice); every volume is swept with
dense rows, and once more left to the host, which compacts sparse ids (the rows of the ids present must be the same)."""
import os
import sys
import numpy as np

sys.path.insert(0, "tests")
from oracle import onepass_c
from tissue_analysis_amd import _capi
from tissue_analysis_amd.extraction import extract_volume

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
ctx = _capi.Context(0)
if os.environ.get("TA_SOAK_SHAPE"):
    ctx.set_option(_capi.OPT_SWEEP_SHAPE, int(os.environ["TA_SOAK_SHAPE"]))
# (count, bounding box and first moments come out of every sweep; second moments and adjacency on request)
KEYS = {0: ["count", "bbox", "sum1"], 8: ["sum2"], 16: ["pair_lo", "pair_hi", "pair_faces"]}
done = 0
for it in range(n):
    dtype = [np.uint16, np.uint32][rng.integers(0, 2)]
    shape = (int(rng.integers(1, 40)), int(rng.integers(1, 70)), int(rng.choice([1, 3, 4, 8, 12, 17, 64, 68, 130, 256, 257, 260, 300, 512, 520, 768, 1000, 1024, 1030, 1536])))
    top = 65535 if dtype == np.uint16 else int(rng.choice([70000, 200000, 1 << 20, 1 << 22]))       # (dense per-label rows: 2^28 labels would be 28 GB of them)
    nlab = int(rng.integers(1, 40))
    ids = np.unique(rng.integers(0, top + 1, size=nlab)).astype(dtype)
    block = (int(rng.integers(1, 6)), int(rng.integers(1, 9)), int(rng.integers(1, 60)))
    coarse = [int(np.ceil(s / b)) for s, b in zip(shape, block)]
    v = ids[rng.integers(0, ids.size, size=coarse)]
    for ax, b in enumerate(block):
        v = np.repeat(v, b, axis=ax)
    v = np.ascontiguousarray(v[:shape[0], :shape[1], :shape[2]])
    if rng.random() < 0.3:                               # speckle: single voxels of other labels
        k = int(v.size * 0.02) + 1
        v.flat[rng.integers(0, v.size, size=k)] = ids[rng.integers(0, ids.size, size=k)]
    vol = np.asfortranarray(v) if rng.random() < 0.25 else v
    tp = int(rng.choice([0, 0, 1, 2, 5, 16, 24, 32, 64]))
    mask = int(rng.choice([31, 31, 23, 15, 7, 3, 1, 17, 19, 27]))
    if it < int(sys.argv[4] if len(sys.argv) > 4 else 0):                # (replay: skip to a case, same random stream)
        continue
    if len(sys.argv) > 3:                                # progress file: the case about to run
        with open(sys.argv[3], "a") as fh:
            fh.write("it=%d shape=%s dtype=%s tp=%d mask=0x%x block=%s nlab=%d top=%d\n" % (it, vol.shape, vol.dtype.name, tp, mask, block, ids.size, top))
    if it % 25 == 0:
        print("it=%d" % it, flush=True)
    want = onepass_c.extract(np.ascontiguousarray(vol))
    got = extract_volume(vol, features=mask, context=ctx, tile_planes=tp, sparse=False).as_arrays()
    for bit, keys in KEYS.items():
        if bit and not mask & bit:
            continue
        for k in keys:
            if not (got[k].shape == want[k].shape and np.array_equal(got[k], want[k])):
                print("MISMATCH it=%d key=%s shape=%s dtype=%s tp=%d mask=0x%x order=%s" % (it, k, vol.shape, vol.dtype, tp, mask, "F" if vol.flags.f_contiguous and not vol.flags.c_contiguous else "C"))
                sys.exit(1)
    x = extract_volume(vol, features=mask, context=ctx, tile_planes=tp)
    if x.sparse:
        sel = x.ids
        ok = np.array_equal(sel, np.unique(vol))
        for k in ("count", "bbox", "sum1") + (("sum2",) if mask & 8 else ()):
            ok = ok and np.array_equal(getattr(x, k), np.asarray(want[k]).reshape((-1,) + getattr(x, k).shape[1:])[sel])
        if mask & 16:
            ok = ok and np.array_equal(x.pair_lo, want["pair_lo"]) and np.array_equal(x.pair_hi, want["pair_hi"]) and np.array_equal(x.pair_faces, want["pair_faces"])
        if not ok:
            print("MISMATCH (compacted ids) it=%d shape=%s dtype=%s tp=%d mask=0x%x" % (it, vol.shape, vol.dtype, tp, mask))
            sys.exit(1)
    done += 1
print("soak ok: %d random volumes equal to the oracle" % done)
