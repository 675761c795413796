"""Time the wall-voxel passes on a resident volume: python scripts/probe_walls.py [C2|C3]"""
import sys
import numpy as np
import torch
from tissue_analysis_amd import device as dev, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
c = synth.CONFIGS[name]
dims, dtype = c["dims"], np.dtype(c["dtype"])
if len(sys.argv) > 2:
    dims = (int(sys.argv[2]),) * 3
ctx = dev.torch_context(0)
vol, max_label = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"])
ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
import time
for it in range(2):
    t0 = time.perf_counter()
    glo, ghi, gcoords, gms = ctx.wall_voxels(by_pair=True)
    t1 = time.perf_counter()
    print("grouped by pair on the device: kernels %.3f ms, call %.1f ms (records to the host included)" % (gms, (t1 - t0) * 1e3), flush=True)
    lo, hi, coords, ms = ctx.wall_voxels()
    nvox = float(np.prod(dims))
    alg = nvox * dtype.itemsize + 20.0 * lo.size          # the volume once + 20-byte records (lo, hi, 3 coordinates)
    print("%s %s: %d records (%.1f%% of voxels), count+scan+emit kernels %.3f ms -> %.0f Mvoxel/s, %.0f GB/s algorithmic = %.1f%% of 8 TB/s"
          % (name, dims, lo.size, 100.0 * lo.size / nvox, ms, nvox / ms / 1e3, alg / ms / 1e6, alg / ms / 1e6 / 80.0), flush=True)
