"""Phase breakdown of the sweep from a TA_STAMPS build, and record counts from a TA_RECCOUNT build (GPU box):

    TISSUE_SCAN_LIB=$PWD/scratch/libstamps.so   python scripts/probe_stamps.py [C4] [--no-ellipsoid]
    TISSUE_SCAN_LIB=$PWD/scratch/libreccount.so python scripts/probe_stamps.py [C4] [--no-ellipsoid]
"""
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
for a in sys.argv:                                   # --shape=0|1: TA_OPT_SWEEP_SHAPE
    if a.startswith("--shape="):
        ctx.set_option(_capi.OPT_SWEEP_SHAPE, int(a.split("=")[1]))
for feats in (0x1f,):
    for _ in range(3):
        ctx.extract(feats, L)
        ctx.synchronize()
    t = ctx.timing()
    d = ctx.debug_counters()
    s = d.get("stamps")
    print("%s feat=0x%02x sweep %.3f ms  counters %s" % (name, feats, t["ms_sweep"], d))
    if s and s[4] > 1000 and s[5] > 1000:      # stamps build: cycles >> 8 summed over waves
        cmp_, emit, drain, adv, total, land, evrows, drains = s
        print("  per-phase share of wave time: compares %.1f%%  emit+drain %.1f%% (drain alone %.1f%%)  landing+plane faces %.1f%% "
              "(landing wait alone %.1f%%)  other %.1f%%;  rows with events %d, drains %d"
              % (100.0 * cmp_ / total, 100.0 * emit / total, 100.0 * drain / total, 100.0 * adv / total, 100.0 * land / total,
                 100.0 * (total - cmp_ - emit - adv) / total, evrows, drains))
    elif s:
        print("  records per launch: faces (axes 0, 1) %d, runs (with the axis-2 faces) %d, drains %d" % (s[0], s[1], s[2]))
