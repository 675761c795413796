"""The wall-voxel kernels on a resident volume with the chip at its sustained clock (what bench.py's secondary.wall_voxels_c2
reports): python scripts/probe_walls_warm.py [C2|C3]"""
import sys
import time
import numpy as np
import torch
from tissue_analysis_amd import device as dev, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
c = synth.CONFIGS[name]
dims, dtype = c["dims"], np.dtype(c["dtype"])
ctx = dev.torch_context(0)
vol, max_label = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"])
ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:                  # settle the clock with the sweep
    ctx.extract(0x0f, max_label)
best = None
for _ in range(4):
    lo, hi, coords, ms = ctx.wall_voxels()
    best = ms if best is None else min(best, ms)
alg = float(np.prod(dims)) * dtype.itemsize + 20.0 * lo.size
print("%s: %d records, count + scan + fetch kernels best %.4f ms = %.1f%% of 8 TB/s" % (name, lo.size, best, alg / best / 1e6 / 80.0), flush=True)
