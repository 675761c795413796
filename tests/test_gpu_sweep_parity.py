"""GPU parity: the fused HIP sweep (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from oracle import onepass, onepass_c
from tissue_analysis_amd import _capi
from tissue_analysis_amd.extraction import extract_volume

from helpers import assert_same_accumulators, random_blocks, voronoi

pytestmark = pytest.mark.gpu


def run(ctx, vol, impl, features=_capi.F_ALL, tile_planes=None, max_label=None):
    x = extract_volume(vol, features, context=ctx, impl=impl, tile_planes=tile_planes, max_label=max_label)
    return x.as_arrays()


CASES = [
    ("voronoi_u16_small", lambda: voronoi((20, 24, 40), 12, 1, np.uint16)),
    ("voronoi_u16_c64", lambda: voronoi((33, 31, 64), 30, 2, np.uint16)),
    ("voronoi_u32_aligned", lambda: voronoi((40, 32, 256), 60, 3, np.uint32)),
    ("voronoi_u16_aligned", lambda: voronoi((24, 16, 512), 60, 4, np.uint16)),
    ("voronoi_u32_wide", lambda: voronoi((9, 20, 520), 40, 5, np.uint32)),
    ("voronoi_u16_wide", lambda: voronoi((7, 9, 1032), 40, 6, np.uint16)),
    ("blocks_u16", lambda: random_blocks((17, 13, 29), 40, 7, np.uint16)),
    ("blocks_u32", lambda: random_blocks((11, 37, 70), 300, 8, np.uint32)),
    ("single_voxel_labels", lambda: np.arange(2 * 3 * 5, dtype=np.uint16).reshape(2, 3, 5)),
    ("uniform", lambda: np.full((5, 6, 7), 3, dtype=np.uint32)),
    ("flat_2d", lambda: voronoi((30, 40, 1), 10, 9, np.uint16, ellipsoid=False)),
    ("one_plane", lambda: voronoi((1, 40, 300), 10, 10, np.uint32, ellipsoid=False)),
]


@pytest.mark.parametrize("impl", [1, 0, 2, 3, 4, 5, 6], ids=["naive", "default", "split", "rowrun", "fused", "rle", "scan"])
@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
def test_accumulators_match_oracle(gpu_ctx, name, make, impl):
    vol = make()
    want = onepass_c.extract(vol)
    got = run(gpu_ctx, vol, impl)
    assert_same_accumulators(got, want, "%s impl=%d" % (name, impl))


@pytest.mark.parametrize("impl", [0, 2, 3, 4, 5, 6], ids=["default", "split", "rowrun", "fused", "rle", "scan"])
@pytest.mark.parametrize("tile_planes", [1, 2, 5, 64])
def test_tile_planes_do_not_change_results(gpu_ctx, tile_planes, impl):
    vol = voronoi((23, 40, 300), 50, 11, np.uint32)
    want = onepass_c.extract(vol)
    got = run(gpu_ctx, vol, impl, tile_planes=tile_planes)
    assert_same_accumulators(got, want, "tile_planes=%d impl=%d" % (tile_planes, impl))


@pytest.mark.parametrize("features", [_capi.F_VOLUME | _capi.F_BBOX | _capi.F_MOMENT1,
                                      _capi.F_VOLUME | _capi.F_BBOX | _capi.F_MOMENT1 | _capi.F_ADJACENCY,
                                      _capi.F_VOLUME | _capi.F_BBOX | _capi.F_MOMENT1 | _capi.F_MOMENT2])
@pytest.mark.parametrize("impl", [0, 2, 3, 4, 5, 6], ids=["default", "split", "rowrun", "fused", "rle", "scan"])
def test_feature_subsets(gpu_ctx, features, impl):
    vol = voronoi((20, 33, 260), 40, 12, np.uint32)
    want = onepass_c.extract(vol)
    got = run(gpu_ctx, vol, impl, features=features)
    for k in ("count", "bbox", "sum1"):
        assert np.array_equal(got[k], want[k]), k
    if features & _capi.F_MOMENT2:
        assert np.array_equal(got["sum2"], want["sum2"])
    if features & _capi.F_ADJACENCY:
        for k in ("pair_lo", "pair_hi", "pair_faces"):
            assert np.array_equal(got[k], want[k]), k
    else:
        assert got["pair_lo"].size == 0


def test_fortran_and_transposed_layouts(gpu_ctx):
    vol = voronoi((18, 22, 36), 15, 13, np.uint16)
    want = onepass.extract(vol)
    for arr in (np.asfortranarray(vol), np.ascontiguousarray(vol.transpose(1, 2, 0)).transpose(2, 0, 1),
                vol[::1, ::1, ::1], vol[:, ::2, :]):
        w = want if arr.shape == vol.shape else onepass.extract(np.ascontiguousarray(arr))
        got = run(gpu_ctx, arr, 0)
        assert_same_accumulators(got, w, "layout strides=%s" % (arr.strides,))


def test_label_above_max_label_is_reported(gpu_ctx):
    vol = voronoi((8, 8, 64), 6, 14, np.uint16)
    with pytest.raises(_capi.TissueScanError) as e:
        run(gpu_ctx, vol, 0, max_label=int(vol.max()) - 1)
    assert e.value.code == _capi.TA_ERANGE


def test_split_path_falls_back_on_noise(gpu_ctx):
    """Per-voxel noise has far more events than a record region holds: the split path must notice
    and hand the volume to the fused sweep, silently and exactly."""
    rng = np.random.default_rng(15)
    vol = rng.integers(1, 50, size=(40, 16, 256)).astype(np.uint32)
    want = onepass_c.extract(vol)
    got = run(gpu_ctx, vol, 2)
    assert_same_accumulators(got, want, "noise through impl=2")
    got = run(gpu_ctx, voronoi((40, 16, 256), 20, 16, np.uint32), 2)      # and a tissue volume afterwards
    assert_same_accumulators(got, onepass_c.extract(voronoi((40, 16, 256), 20, 16, np.uint32)), "tissue after noise")
