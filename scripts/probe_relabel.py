"""Time the label lookup-table sweep on a resident volume: python scripts/probe_relabel.py [C4]"""
import sys, time
import numpy as np
import torch
from tissue_analysis_amd import _capi, device as dev, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
c = synth.CONFIGS[name]
dims, dtype = c["dims"], np.dtype(c["dtype"])
ctx = dev.torch_context(0)
vol, max_label = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"])
ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
lut = np.arange(max_label + 1, dtype=np.uint32)
lut[2::3] = 0                                  # erase a third of the cells
nbytes = 2.0 * np.prod(dims) * dtype.itemsize  # one read + one write per voxel
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.relabel(lut)                           # includes the table upload and a stream sync
    dt = time.perf_counter() - t0
    print("%s relabel %.3f ms  %.0f GB/s (read+write) = %.1f%% of 8 TB/s" % (name, dt * 1e3, nbytes / dt / 1e9, nbytes / dt / 8e10), flush=True)
