"""Quick perf probe: python scripts/probe_perf.py CONFIG [--feat MASK] [--tp N ...] [--iters K]"""
import argparse, time
import numpy as np
import torch
from tissue_analysis_amd import _capi, device as dev, synth

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="C4")
ap.add_argument("--feat", type=lambda s: int(s, 0), nargs="*", default=None)
ap.add_argument("--tp", type=int, nargs="*", default=[64])
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--dims", type=int, nargs=3, default=None)
ap.add_argument("--cells", type=int, default=None)
ap.add_argument("--impl", type=int, default=0)
args = ap.parse_args()
c = synth.CONFIGS[args.config]
dims = tuple(args.dims) if args.dims else c["dims"]
dtype = np.dtype(c["dtype"])
ncell = args.cells or c["n_cells"]
ctx = dev.torch_context(0)
ctx.set_option(_capi.OPT_IMPL, args.impl)
t0 = time.time()
vol, max_label = dev.synth_slab(ctx, dims, dtype, ncell, c["seed"])
torch.cuda.synchronize()
print("synth %.2fs dims=%s dtype=%s max_label=%d" % (time.time() - t0, dims, dtype, max_label), flush=True)
ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
for feats in (args.feat or [_capi.feature_mask(c["features"])]):
    for tp in args.tp:
        ctx.set_option(_capi.OPT_TILE_PLANES, tp)
        best = None
        for it in range(args.iters):
            ctx.extract(feats, max_label)
            ctx.synchronize()
            t = ctx.timing()
            if best is None or t["ms_sweep"] < best["ms_sweep"]:
                best = t
        gbs = best["bytes_read"] / best["ms_sweep"] / 1e6
        print("feat=0x%02x tile_planes=%d sweep %.3f ms adj %.3f ms total %.3f ms -> %.1f GB/s (%.1f%% of 8TB/s) %.0f Mvox/s" % (
            feats, tp, best["ms_sweep"], best["ms_adjacency"], best["ms_total"], gbs, gbs / 80.0,
            np.prod(dims) / best["ms_total"] / 1e3), ctx.debug_counters(), flush=True)
