import sys, numpy as np
sys.path.insert(0, "tests")
from tissue_analysis_amd import _capi
from helpers import voronoi
vol = voronoi((20, 24, 1024), 60, 21, np.uint32)
ctx = _capi.Context(0)
ctx.set_option(_capi.OPT_SWEEP_SHAPE, 0)
ctx.set_volume(vol)
ctx.extract(0x1f, int(vol.max()))
ctx.synchronize()
print(ctx.debug_counters())
