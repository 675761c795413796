"""Child of tests/test_sanitized_host.py: runs with the AddressSanitizer runtime preloaded, the host-sanitized
libtissue_scan (TISSUE_SCAN_LIB) and the sanitized C oracle (ONEPASS_ORACLE_LIB).  Walks the part of the C ABI that
needs no GPU -- every entry point's argument checks, the failure paths of ta_ctx_create -- and drives the C oracle
over small, ragged and degenerate volumes.  Any sanitizer report aborts the process (halt_on_error)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tissue_analysis_amd import _capi          # noqa: E402
from oracle import onepass, onepass_c          # noqa: E402  (the checker: this IS a test)


def walk_c_abi():
    lib = _capi.load()
    assert os.path.samefile(_capi.LIB_PATH, os.environ["TISSUE_SCAN_LIB"])
    assert lib.ta_version() == _capi.ABI_VERSION
    n = ctypes.c_int(-1)
    assert lib.ta_device_count(ctypes.byref(n)) in (_capi.TA_OK, _capi.TA_ENODEVICE)
    assert lib.ta_device_count(None) == _capi.TA_EINVAL
    # ta_ctx_create: bad arguments, then (no GPU here) the failure path that must give back what it took
    assert lib.ta_ctx_create(0, None) == _capi.TA_EINVAL
    for dev in (-1, 0, 7, 1 << 20):
        h = ctypes.c_void_p()
        rc = lib.ta_ctx_create(dev, ctypes.byref(h))
        if rc == _capi.TA_OK:
            assert lib.ta_ctx_destroy(h) == _capi.TA_OK
        else:
            assert rc in (_capi.TA_EINVAL, _capi.TA_ENODEVICE, _capi.TA_EHIP) and not h.value
            assert lib.ta_last_error()
    assert lib.ta_ctx_destroy(None) == _capi.TA_OK
    # every entry point that takes a context rejects NULL before touching anything else
    i64, u32, dbl, vp = ctypes.c_int64(0), ctypes.c_uint32(0), ctypes.c_double(0), ctypes.c_void_p()
    dims = (ctypes.c_int64 * 3)(4, 4, 4)
    buf = (ctypes.c_uint32 * 64)()
    null_ctx = {
        "ta_ctx_set_stream": (None, None),
        "ta_ctx_set_option": (None, _capi.OPT_IMPL, 0),
        "ta_ctx_get_option": (None, _capi.OPT_IMPL, ctypes.byref(i64)),
        "ta_ctx_synchronize": (None,),
        "ta_volume_set": (None, buf, 4, dims, None),
        "ta_volume_set_device": (None, buf, 4, dims, 0, 0),
        "ta_volume_max_label": (None, ctypes.byref(u32)),
        "ta_volume_plane_events": (None, buf),
        "ta_volume_label_census": (None, ctypes.byref(u32), ctypes.byref(u32)),
        "ta_label_census_get": (None, buf),
        "ta_volume_compact_labels": (None, None, 0, ctypes.byref(u32)),
        "ta_volume_is_compact": (None, ctypes.byref(ctypes.c_int(0)), None),
        "ta_volume_rerank": (None,),
        "ta_volume_uncompact": (None,),
        "ta_volume_owned_planes": (None, ctypes.byref(i64)),
        "ta_volume_relabel": (None, buf, 4),
        "ta_volume_get": (None, buf),
        "ta_volume_map": (None, buf, 4, buf, 4, buf),
        "ta_volume_first_layer": (None, 1, 1, buf),
        "ta_volume_hollow": (None, 1, 1, 0, buf),
        "ta_volume_layer18": (None, buf),
        "ta_wall_voxels_count": (None, ctypes.byref(i64)),
        "ta_wall_voxels_get": (None, buf, buf, ctypes.byref(dbl)),
        "ta_wall_voxels_get_by_pair": (None, buf, buf, ctypes.byref(dbl)),
        "ta_wall_medians": (None, 10, ctypes.byref(i64), None),
        "ta_wall_medians_get": (None, buf, buf, buf),
        "ta_extract": (None, 31, 10),
        "ta_get_labels": (None, buf, buf, buf, buf),
        "ta_adjacency_size": (None, ctypes.byref(i64)),
        "ta_adjacency_get": (None, buf, buf, buf),
        "ta_timing": (None, ctypes.byref(dbl), ctypes.byref(dbl), ctypes.byref(dbl), None),
        "ta_timing_series": (None, ctypes.byref(dbl), 1, ctypes.byref(ctypes.c_int(0))),
        "ta_read_probe": (None, buf, 256, 1, ctypes.byref(dbl)),
        "ta_debug_counters": (None, buf),
        "ta_bind_accumulators": (None, buf, buf, 3),
        "ta_accumulators_device": (None, ctypes.byref(vp), ctypes.byref(vp), ctypes.byref(u32)),
        "ta_accumulators_reduced": (None,),
        "ta_adjacency_device": (None, ctypes.byref(vp), ctypes.byref(vp), ctypes.byref(i64)),
        "ta_adjacency_scope": (None, ctypes.byref(ctypes.c_int(0))),
        "ta_adjacency_export": (None, buf, buf, 4),
        "ta_adjacency_merge": (None, buf, buf, 4),
        "ta_adjacency_pack": (None, buf, 4),
        "ta_adjacency_pack_shared": (None, buf, 4),
        "ta_adjacency_merge_blocks": (None, buf, 1, 4),
        "ta_synth_voronoi": (None, buf, 4, dims, 0, 4, buf, None, None),
        "ta_device_malloc": (None, 64, ctypes.byref(vp)),
        "ta_device_free": (None, None),
        "ta_memcpy_d2h": (None, buf, buf, 4),
        "ta_memcpy_h2d": (None, buf, buf, 4),
    }
    walked = {"ta_version", "ta_last_error", "ta_device_count", "ta_ctx_create", "ta_ctx_destroy"}
    for name, args in null_ctx.items():
        rc = getattr(lib, name)(*args)
        assert rc == _capi.TA_EINVAL, (name, rc)
        assert b"NULL" in lib.ta_last_error() or b"ctx" in lib.ta_last_error(), (name, lib.ta_last_error())
        walked.add(name)
    assert walked == set(_capi.SYMBOLS), sorted(set(_capi.SYMBOLS) - walked)
    return len(walked)


def drive_c_oracle():
    assert os.path.samefile(onepass_c._LIB, os.environ["ONEPASS_ORACLE_LIB"])
    rng = np.random.default_rng(20261004)
    shapes = [(1, 1, 1), (1, 7, 1), (4, 6, 1), (5, 1, 9), (9, 11, 23), (16, 16, 16), (3, 40, 2), (17, 5, 31)]
    cases = 0
    for shape in shapes:
        for dtype, top in ((np.uint16, 65535), (np.uint32, 70000)):
            for nlab in (1, 3, 40):
                vol = rng.integers(1, nlab + 1, size=shape).astype(dtype)
                if nlab == 40 and vol.size > 8:
                    vol.flat[rng.integers(0, vol.size)] = top          # the largest label the dtype / test carries
                got = onepass_c.extract(vol)
                want = onepass.extract(vol)
                for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces"):
                    assert np.array_equal(got[k], want[k]), (shape, dtype, nlab, k)
                # a slab with an origin whose first plane is a halo
                if shape[0] > 2:
                    g2 = onepass_c.extract(vol, origin=(5, 0, 0), own_first_plane=False)
                    w2 = onepass.extract(vol, origin=(5, 0, 0), own_first_plane=False)
                    for k in ("count", "sum1", "sum2", "pair_faces"):
                        assert np.array_equal(g2[k], w2[k]), (shape, dtype, nlab, k, "slab")
                cases += 1
    try:                                   # a label above max_label is an error, not a write past the rows
        onepass_c.extract(np.full((3, 3, 3), 9, np.uint16), max_label=4)
    except ValueError:
        cases += 1
    else:
        raise AssertionError("label above max_label accepted")
    return cases


if __name__ == "__main__":
    print("c-abi entry points walked: %d" % walk_c_abi())
    print("c-oracle cases: %d" % drive_c_oracle())
    print("SANITIZED-OK")
