"""Property tests on the GPU (hypothesis): random small labelled volumes of random shape, dtype,
label set (absent ids, label 0, ids above 32767 / 65535), layout and tile size; the fused sweep
through the C ABI must equal the CPU oracle bit for bit.  Also a few API error paths."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import onepass_c
from tissue_analysis_amd import SpatialImageAnalysis, _capi
from tissue_analysis_amd.extraction import extract_volume

from helpers import assert_same_accumulators

pytestmark = pytest.mark.gpu


@st.composite
def volumes(draw):
    dtype = draw(st.sampled_from([np.uint16, np.uint32]))
    shape = (draw(st.integers(1, 12)), draw(st.integers(1, 21)), draw(st.sampled_from([1, 3, 8, 17, 64, 130, 257, 300, 520])))
    top = 65535 if dtype == np.uint16 else draw(st.sampled_from([70000, 200000, 1 << 20]))
    nlab = draw(st.integers(1, 12))
    ids = np.array(sorted(set(draw(st.lists(st.integers(0, top), min_size=nlab, max_size=nlab)))), dtype=dtype)
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    block = (draw(st.integers(1, 4)), draw(st.integers(1, 5)), draw(st.integers(1, 40)))
    coarse = [int(np.ceil(s / b)) for s, b in zip(shape, block)]
    v = ids[rng.integers(0, ids.size, size=coarse)]
    for ax, b in enumerate(block):
        v = np.repeat(v, b, axis=ax)
    v = np.ascontiguousarray(v[:shape[0], :shape[1], :shape[2]])
    order = draw(st.sampled_from(["C", "F"]))
    tile_planes = draw(st.sampled_from([1, 2, 5, 32]))
    impl = draw(st.sampled_from([0, 0, 0, 1]))
    return (np.asfortranarray(v) if order == "F" else v), tile_planes, impl


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(case=volumes())
def test_random_volumes_match_the_oracle(gpu_ctx, case):
    vol, tile_planes, impl = case
    want = onepass_c.extract(np.ascontiguousarray(vol))
    what = "shape=%s dtype=%s tp=%d impl=%d" % (vol.shape, vol.dtype, tile_planes, impl)
    got = extract_volume(vol, context=gpu_ctx, impl=impl, tile_planes=tile_planes, sparse=False).as_arrays()
    assert_same_accumulators(got, want, what)
    # left to itself the host compacts ids that are sparse: the same rows, only those of the ids present
    x = extract_volume(vol, context=gpu_ctx, impl=impl, tile_planes=tile_planes)
    if x.sparse:
        assert np.array_equal(x.ids, np.unique(vol)), what
        for k in ("count", "bbox", "sum1", "sum2"):
            assert np.array_equal(getattr(x, k), np.asarray(want[k]).reshape(getattr(x, k).shape[:0] + (-1,) + getattr(x, k).shape[1:])[x.ids]), (what, k)
        for k in ("pair_lo", "pair_hi", "pair_faces"):
            assert np.array_equal(getattr(x, k).reshape(-1), np.asarray(want[k]).reshape(-1)), (what, k)
    else:
        assert_same_accumulators(x.as_arrays(), want, what)


def test_refresh_after_in_place_edit():
    vol = np.ones((6, 8, 70), dtype=np.uint16)
    vol[2:4, 2:5, 10:30] = 7
    sia = SpatialImageAnalysis(vol, background=1)
    assert sia.volume(7, real=False)[7] == 2 * 3 * 20
    sia.image[2:4, 2:5, 10:20] = 1
    sia.refresh()
    assert sia.volume(7, real=False)[7] == 2 * 3 * 10
    assert sia.neighbors(7) == [1]


def test_bad_inputs_raise():
    with pytest.raises(TypeError):
        SpatialImageAnalysis(np.ones((4, 4, 4), dtype=np.float32), background=1)
    with pytest.raises(ValueError):
        SpatialImageAnalysis(-np.ones((4, 4, 4), dtype=np.int32), background=1)
    ctx = _capi.Context(0)
    with pytest.raises(_capi.TissueScanError) as e:
        ctx.extract(_capi.F_ALL, 10)              # no volume set
    assert e.value.code == _capi.TA_EINVAL
    with pytest.raises(_capi.TissueScanError):
        ctx.set_option(_capi.OPT_TILE_PLANES, 10 ** 6)
    with pytest.raises(TypeError):
        ctx.set_volume(np.zeros((2, 2, 2), dtype=np.int8))
    ctx.close()


def test_two_contexts_are_independent(gpu_ctx):
    a = np.arange(2 * 3 * 64, dtype=np.uint32).reshape(2, 3, 64) % 5 + 1
    b = (np.arange(4 * 2 * 64, dtype=np.uint32).reshape(4, 2, 64) // 7) % 3 + 1
    other = _capi.Context(0)
    xa = extract_volume(a, context=gpu_ctx).as_arrays()
    xb = extract_volume(b, context=other).as_arrays()
    xa2 = extract_volume(a, context=gpu_ctx).as_arrays()
    other.close()
    assert_same_accumulators(xa, onepass_c.extract(a), "ctx A")
    assert_same_accumulators(xb, onepass_c.extract(b), "ctx B")
    assert_same_accumulators(xa2, xa, "ctx A again")
