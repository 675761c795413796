"""Sparse label ids on a full-size volume (GPU box): C4's cells renamed to random ids in [0, 2^32) on the device, then the
census of the ids, the rank copy, and the sweep over it -- timed, and compared row by row with the sweep of the dense volume.

    python scripts/probe_sparse.py [C4]"""
import sys
import time

import numpy as np
import torch

from tissue_analysis_amd import _capi, device as dev, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
c = synth.CONFIGS[name]
dims, dtype = c["dims"], np.dtype(c["dtype"])
ctx = dev.torch_context(0)
vol, max_label = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"])
torch.cuda.synchronize()
ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)


def timed(f, n=5):
    best, out = 1e9, None
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, out


def sweep(rows):
    ctx.extract(_capi.F_ALL, rows - 1)
    ctx.adjacency_size()
    return ctx.timing()


t_dense, tm = timed(lambda: sweep(max_label + 1))
dense = ctx.labels()
dlo, dhi, dfaces = ctx.adjacency()
print("%s dense ids 0..%d: sweep call %.3f ms (kernel %.3f ms)" % (name, max_label, t_dense, tm["ms_sweep"]))

rng = np.random.default_rng(5)
top = (1 << 32) - 1 if dtype.itemsize == 4 else 65535
lut = np.unique(rng.integers(0, top, size=2 * (max_label + 1), dtype=np.uint64))[:max_label + 1].astype(np.uint32)
rng.shuffle(lut)
ctx.relabel(lut)                                    # in place on the device: v -> lut[v]
t_census, (cmax, ids) = timed(ctx.label_census, 3)
print("census of %d ids (max %d, table %.0f MB): %.3f ms" % (ids.size, cmax, (cmax // 32 + 1) * 8 / 1e6, t_census))
t_compact, table = timed(lambda: ctx.compact_labels(), 3)
print("rank copy (census kept): %.3f ms" % t_compact)
t_sparse, tm = timed(lambda: sweep(ids.size))
print("sweep of the rank copy: call %.3f ms (kernel %.3f ms)" % (t_sparse, tm["ms_sweep"]))
sparse = ctx.labels()
slo, shi, sfaces = ctx.adjacency()
present = np.flatnonzero(dense[0])
rows = np.searchsorted(ids, lut[present])
ok = all(np.array_equal(s[rows], d[present]) for s, d in zip(sparse, dense))
# the pair list: the same walls under the renaming
a, b = lut[dlo].astype(np.uint64), lut[dhi].astype(np.uint64)
dk = (np.minimum(a, b) << np.uint64(32)) | np.maximum(a, b)
order = np.argsort(dk)
sk = (slo.astype(np.uint64) << np.uint64(32)) | shi.astype(np.uint64)
ok = ok and np.array_equal(dk[order], sk) and np.array_equal(dfaces[order], sfaces)
print("rows and walls equal to the dense sweep's under the renaming:", ok)
sys.exit(0 if ok else 1)
