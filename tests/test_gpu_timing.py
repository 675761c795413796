"""ta_timing / ta_timing_series: HIP events around the sweep kernel of each extraction (TA_OPT_TIMING, TA_OPT_TIMING_RING)."""
import numpy as np
import pytest

from tissue_analysis_amd import _capi

from helpers import voronoi

pytestmark = pytest.mark.gpu


def test_series_keeps_the_last_launches_and_modes_switch_events():
    vol = voronoi((40, 48, 264), 60, 91, np.uint32)
    ctx = _capi.Context(0)
    try:
        ctx.set_volume(vol)
        L = int(vol.max())
        assert ctx.get_option(_capi.OPT_TIMING) == 1 and ctx.get_option(_capi.OPT_TIMING_RING) == 1
        ctx.extract(_capi.F_ALL, L)
        t = ctx.timing()
        assert t["ms_sweep"] > 0 and t["ms_total"] is None and t["ms_adjacency"] is None   # default: the sweep kernel only
        assert t["bytes_read"] == vol.size * 4
        assert len(ctx.timing_series()) == 1
        ctx.set_option(_capi.OPT_TIMING_RING, 5)
        assert ctx.timing_series() == []                                                # a new series starts
        for _ in range(3):
            ctx.extract(_capi.F_ALL, L)
        assert len(ctx.timing_series()) == 3
        for _ in range(4):
            ctx.extract(_capi.F_ALL, L)
        series = ctx.timing_series()
        assert len(series) == 5 and all(0 < ms < 50 for ms in series)
        assert abs(series[-1] - ctx.timing()["ms_sweep"]) < 1e-9
        assert len(ctx.timing_series(capacity=2)) == 2
        ctx.set_option(_capi.OPT_TIMING, 2)
        ctx.extract(_capi.F_ALL, L)
        t = ctx.timing()
        assert t["ms_total"] >= t["ms_sweep"] > 0 and t["ms_adjacency"] > 0
        ctx.set_option(_capi.OPT_TIMING, 0)
        ctx.extract(_capi.F_ALL, L)
        t = ctx.timing()                                                                # no events recorded: "not measured", not an error
        assert t["ms_sweep"] is None and t["ms_total"] is None and t["ms_adjacency"] is None and t["bytes_read"] == vol.size * 4
        assert ctx.timing_series() == []
        counts = ctx.labels()[0]                                                        # results are unaffected
        assert int(counts.sum()) == vol.size
        with pytest.raises(_capi.TissueScanError):
            ctx.set_option(_capi.OPT_TIMING_RING, 0)
    finally:
        ctx.close()
