"""Time several sweep implementations on one resident volume, checking that they agree bit for bit.

    python scripts/probe_impls.py [CONFIG] [--impl 0 1] [--feat 0x1f 0x0f ...] [--iters K] [--no-ellipsoid]
"""
import argparse
import time

import numpy as np
import torch

from tissue_analysis_amd import _capi, device as dev, synth

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="C4")
ap.add_argument("--impl", type=int, nargs="*", default=[0])       # 1 = the per-voxel-atomics cross-check kernel (slow)
ap.add_argument("--feat", type=lambda s: int(s, 0), nargs="*", default=[0x1f])
ap.add_argument("--tp", type=int, nargs="*", default=[0])
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dims", type=int, nargs=3, default=None)
ap.add_argument("--cells", type=int, default=None)
ap.add_argument("--no-ellipsoid", action="store_true")
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--shape", type=int, default=None)      # TA_OPT_SWEEP_SHAPE (uint32 volumes with adjacency): 0 / 1
args = ap.parse_args()
c = synth.CONFIGS[args.config]
dims = tuple(args.dims) if args.dims else c["dims"]
dtype = np.dtype(c["dtype"])
ncell = args.cells or c["n_cells"]
ctx = dev.torch_context(0)
t0 = time.time()
vol, max_label = dev.synth_slab(ctx, dims, dtype, ncell, c["seed"], ellipsoid=not args.no_ellipsoid)
torch.cuda.synchronize()
print("synth %.2fs dims=%s dtype=%s max_label=%d ellipsoid=%s" % (time.time() - t0, dims, dtype, max_label,
                                                                  not args.no_ellipsoid), flush=True)
ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
ctx.set_option(_capi.OPT_TIMING, 2)          # also the step's begin / end events (adj / total columns)
if args.shape is not None:
    ctx.set_option(_capi.OPT_SWEEP_SHAPE, args.shape)
ref = {}
for feats in args.feat:
    for impl in args.impl:
        ctx.set_option(_capi.OPT_IMPL, impl)
        for tp in args.tp:
            ctx.set_option(_capi.OPT_TILE_PLANES, tp)
            ts = []
            for it in range(args.iters):
                ctx.extract(feats, max_label)
                ctx.synchronize()
                ts.append(ctx.timing())
            best = min(ts, key=lambda t: t["ms_sweep"])
            med = sorted(t["ms_sweep"] for t in ts)[len(ts) // 2]
            gbs = best["bytes_read"] / best["ms_sweep"] / 1e6
            msg = ""
            if not args.no_check:
                res = ctx.labels() + (ctx.adjacency() if feats & 16 else ())
                if feats not in ref:
                    ref[feats] = res
                    msg = "(reference for this mask)"
                else:
                    ok = all(np.array_equal(a, b) for a, b in zip(ref[feats], res))
                    msg = "same as first impl" if ok else "*** DIFFERS from first impl ***"
            print("impl=%d feat=0x%02x tp=%d sweep best %.3f med %.3f ms adj %.3f total %.3f -> %.0f GB/s (%.1f%% of 8TB/s) %s %s"
                  % (impl, feats, tp, best["ms_sweep"], med, best["ms_adjacency"], best["ms_total"], gbs, gbs / 80.0,
                     ctx.debug_counters(), msg), flush=True)
