"""Which test volume makes the automatic tile height adapt (tests/test_gpu_sweep_parity.py)?  Prints spills of five sweeps per candidate."""
import sys
import numpy as np
sys.path.insert(0, "tests")
from helpers import random_blocks
from tissue_analysis_amd import _capi


for seeds, block in ((40000, (5, 4, 9)), (60000, (4, 3, 6)), (80000, (3, 3, 5)), (60000, (4, 3, 9))):
    vol = random_blocks((96, 64, 512), seeds, 18, np.uint32, block=block)
    c = _capi.Context(0)
    c.set_option(_capi.OPT_TILE_PLANES, 0)
    c.set_volume(vol)
    L = int(vol.max())
    out = []
    for _ in range(5):
        c.extract(_capi.F_ALL, L)
        c.labels()
        d = c.debug_counters()
        out.append((d["label_spills"] + d["pair_spills"], c.get_option(_capi.OPT_TILE_PLANES), c.get_option(_capi.OPT_SWEEP_SHAPE_USED)))
    print(seeds, block, "labels", L, out, flush=True)
    c.close()
