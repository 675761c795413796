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
for it in range(2):
    lo, hi, coords, ms = ctx.wall_voxels()
    nvox = float(np.prod(dims))
    print("%s %s: %d records (%.1f%% of voxels), count+emit kernels %.2f ms -> %.0f Mvoxel/s, %.0f GB/s of volume reads (2 passes) + record writes"
          % (name, dims, lo.size, 100.0 * lo.size / nvox, ms, nvox / ms / 1e3,
             (2 * nvox * dtype.itemsize + 16.0 * lo.size) / ms / 1e6), flush=True)
