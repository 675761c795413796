import sys, time
import numpy as np
import torch
from tissue_analysis_amd import _capi, device as dev, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
c = synth.CONFIGS[cfg]
dims, dtype = c["dims"], np.dtype(c["dtype"])
ctx = dev.torch_context(0)
t0 = time.time()
vol, max_label = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"])
torch.cuda.synchronize()
print("synth %.2fs dims=%s dtype=%s max_label=%d" % (time.time() - t0, dims, dtype, max_label), flush=True)
ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
feats = _capi.feature_mask(c["features"])
for tp in [int(a) for a in sys.argv[2:]] or [32]:
    ctx.set_option(_capi.OPT_TILE_PLANES, tp)
    for it in range(3):
        ctx.extract(feats, max_label)
        ctx.synchronize()
        t = ctx.timing()
        gbs = t["bytes_read"] / t["ms_sweep"] / 1e6
        print("tile_planes=%d it=%d sweep %.3f ms adj %.3f ms total %.3f ms  -> %.1f GB/s (%.1f%% of 8TB/s) %.0f Mvox/s" % (
            tp, it, t["ms_sweep"], t["ms_adjacency"], t["ms_total"], gbs, gbs / 80.0,
            np.prod(dims) / t["ms_total"] / 1e3), flush=True)
count, bbox, s1, s2 = ctx.labels()
lo, hi, f = ctx.adjacency()
print("labels present", int((count > 0).sum()), "pairs", lo.size, "bg frac", count[1] / np.prod(dims))
