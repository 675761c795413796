import sys, numpy as np
sys.path.insert(0, "tests")
from oracle import onepass_c
from tissue_analysis_amd import _capi
from tissue_analysis_amd.extraction import extract_volume
from helpers import voronoi
vol = voronoi((20, 24, 1024), 60, 21, np.uint32)
want = onepass_c.extract(vol)
ctx = _capi.Context(0)
ctx.set_option(_capi.OPT_SWEEP_SHAPE, 0)
for feats in (0x1f, 0x17):
    for tp in (0, 5):
        got = extract_volume(vol, feats, context=ctx, impl=0, tile_planes=tp).as_arrays()
        for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces"):
            a, b = got[k], want[k]
            if k == "sum2" and not feats & 8: continue
            if a.shape != b.shape or not np.array_equal(a, b):
                print("feats %x tp %d: %s differs: shapes %s %s" % (feats, tp, k, a.shape, b.shape))
                if a.shape == b.shape:
                    idx = np.argwhere(a != b)
                    print("   ", len(idx), "entries; first:", idx[:6].tolist(), a[tuple(idx[0])], b[tuple(idx[0])])
        if got["pair_lo"].shape == want["pair_lo"].shape and np.array_equal(got["pair_lo"], want["pair_lo"]):
            d = got["pair_faces"].astype(np.int64) - want["pair_faces"].astype(np.int64)
            print("feats %x tp %d: face delta per axis: %s (abs %s), total faces want %s" % (feats, tp, d.sum(0), np.abs(d).sum(0), want["pair_faces"].sum(0)))
        else:
            gk = set(zip(got["pair_lo"].tolist(), got["pair_hi"].tolist())); wk = set(zip(want["pair_lo"].tolist(), want["pair_hi"].tolist()))
            print("feats %x tp %d: pairs got %d want %d, extra %s missing %s" % (feats, tp, len(gk), len(wk), sorted(gk - wk)[:5], sorted(wk - gk)[:5]))
