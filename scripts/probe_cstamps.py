"""Phase breakdown of the consumer wave from a -DTA_CSTAMPS build (GPU box):
    TISSUE_SCAN_LIB=$PWD/scratch/libcstamps.so python scripts/probe_cstamps.py [C4] [--no-ellipsoid]
flags[8..15] (cycles >> 8 summed over the launch): consumer read / probe / payload phases, producer lifetime, consumer lifetime,
producer spin-wait for a free buffer, buffers consumed, probe rounds."""
import sys
import numpy as np
import torch
from tissue_analysis_amd import _capi, device as dev, synth

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "C4"
c = synth.CONFIGS[name]
dims, dtype = c["dims"], np.dtype(c["dtype"])
ctx = dev.torch_context(0)
vol, L = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"], ellipsoid="--no-ellipsoid" not in sys.argv)
torch.cuda.synchronize()
ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
for _ in range(3):
    ctx.extract(0x1f, L)
    ctx.synchronize()
t = ctx.timing()
s = ctx.debug_counters().get("stamps")
rd, pr, pay, plife, clife, spin, nbuf, rounds = [float(x) for x in s]
print("%s sweep %.3f ms; buffers %d, probe rounds %.2f per buffer" % (name, t["ms_sweep"], nbuf, rounds / max(nbuf, 1)))
print("  consumer, cycles per buffer: read %.0f  probe %.0f  payload %.0f  (sum %.0f);  busy %.1f%% of its lifetime"
      % (256 * rd / nbuf, 256 * pr / nbuf, 256 * pay / nbuf, 256 * (rd + pr + pay) / nbuf, 100 * (rd + pr + pay) / clife))
print("  producers: waiting for a free buffer %.1f%% of their lifetime" % (100 * spin / plife))
