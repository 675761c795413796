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


def _half_and_half(shape, dtype):
    v = np.ones(shape, dtype=dtype)
    v[:, :, :shape[2] // 2] = 2
    v[1, 3, shape[2] // 2 + 5:shape[2] // 2 + 9] = 3          # one small cell inside the right half, away from the tile edge
    return v


def _flat_wall(shape, dtype):
    v = np.ones(shape, dtype=dtype)
    for r in range(shape[1]):
        v[:, r, (7 * r) % 50 + 3:100 + r] = 2 + (r % 5)
        v[:, r, 100 + r:180] = 9
        v[1::2, r, 180:shape[2] - r] = 11                     # every other plane: 60+ columns x 16 rows change label
    return v


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
    # a tile whose ONLY records are the boundaries with the tile on its left (one-label rows need no closing record): the
    # drain at the end of a tile once kept the last such record for a follower that never comes (found at full C4 size)
    ("left_boundary_only_u32", lambda: _half_and_half((26, 32, 512), np.uint32)),
    ("left_boundary_only_u16", lambda: _half_and_half((9, 16, 1024), np.uint16)),
    # a flat wall between two planes: more axis-0 faces in one wave than its face buffer holds, so the plane's first row is
    # placed a lane range at a time and a boundary record waits in the buffer for its follower
    ("flat_wall_between_planes", lambda: _flat_wall((5, 16, 256), np.uint32)),
]


@pytest.mark.parametrize("impl", [1, 0], ids=["naive", "sweep"])
@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
def test_accumulators_match_oracle(gpu_ctx, name, make, impl):
    vol = make()
    want = onepass_c.extract(vol)
    got = run(gpu_ctx, vol, impl)
    assert_same_accumulators(got, want, "%s impl=%d" % (name, impl))


@pytest.mark.parametrize("tile_planes", [1, 2, 5, 64])
def test_tile_planes_do_not_change_results(gpu_ctx, tile_planes, impl=0):
    vol = voronoi((23, 40, 300), 50, 11, np.uint32)
    want = onepass_c.extract(vol)
    got = run(gpu_ctx, vol, impl, tile_planes=tile_planes)
    assert_same_accumulators(got, want, "tile_planes=%d impl=%d" % (tile_planes, impl))


@pytest.mark.parametrize("features", [_capi.F_VOLUME | _capi.F_BBOX | _capi.F_MOMENT1,
                                      _capi.F_VOLUME | _capi.F_BBOX | _capi.F_MOMENT1 | _capi.F_ADJACENCY,
                                      _capi.F_VOLUME | _capi.F_BBOX | _capi.F_MOMENT1 | _capi.F_MOMENT2])
def test_feature_subsets(gpu_ctx, features, impl=0):
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


def test_noise_goes_through_the_lane_range_path(gpu_ctx):
    """Per-voxel noise has more records per row than a wave's buffers hold: such rows are emitted a lane range at
    a time (drain in between); results stay exact, also for the tissue volume that follows on the same context."""
    rng = np.random.default_rng(15)
    for dtype, width in ((np.uint32, 256), (np.uint16, 512), (np.uint32, 300)):
        vol = rng.integers(1, 50, size=(12, 16, width)).astype(dtype)
        assert_same_accumulators(run(gpu_ctx, vol, 0), onepass_c.extract(vol), "noise %s x%d" % (np.dtype(dtype).name, width))
    tissue = voronoi((40, 16, 256), 20, 16, np.uint32)
    assert_same_accumulators(run(gpu_ctx, tissue, 0), onepass_c.extract(tissue), "tissue after noise")


def test_labels_the_records_cannot_carry_are_reported(gpu_ctx):
    """A voxel at or above 2^28 (and 0xFFFFFFFF, the kernel's outside-the-volume filler) must give TA_ERANGE with
    an explicit max_label, with and without adjacency, in vector-load tiles and in edge tiles."""
    for shape in ((6, 16, 256), (5, 9, 70)):
        for bad in (0x10000000, 0xFFFFFFFF, 0x7FFFFFFF):
            for where in ((2, 3, 17), (0, 0, 0), (shape[0] - 1, shape[1] - 1, shape[2] - 1)):
                vol = voronoi(shape, 6, 21, np.uint32)
                vol[where] = bad
                for feats in (0x0f, 0x1f):
                    with pytest.raises(_capi.TissueScanError) as e:
                        run(gpu_ctx, vol, 0, features=feats, max_label=1000)
                    assert e.value.code == _capi.TA_ERANGE, (shape, hex(bad), where, hex(feats))
    # a uniform row of the filler value inside the volume
    vol = voronoi((4, 16, 256), 4, 22, np.uint32)
    vol[1, 5, :] = 0xFFFFFFFF
    with pytest.raises(_capi.TissueScanError) as e:
        run(gpu_ctx, vol, 0, max_label=1000)
    assert e.value.code == _capi.TA_ERANGE


def test_table_spills_keep_every_mask_exact(gpu_ctx):
    """Thousands of tiny cells per tile overflow the workgroup's label table (128 slots) and pair table (512): the
    contributions that spill to the global rows must give the same results -- and must leave the second-moment columns
    untouched (zero) when TA_F_MOMENT2 was not asked for, as the flush does."""
    vol = random_blocks((48, 32, 512), 6000, 17, np.uint32)
    want = onepass_c.extract(vol)
    for features in (_capi.F_ALL, 0x17, 0x0f, 0x07):
        x = extract_volume(vol, features, context=gpu_ctx, impl=0)
        got = x.as_arrays()
        for k in ("count", "bbox", "sum1"):
            assert np.array_equal(got[k], want[k]), (hex(features), k)
        if features & _capi.F_MOMENT2:
            assert np.array_equal(got["sum2"], want["sum2"]), hex(features)
        else:
            assert not got["sum2"].any(), hex(features)
        if features & _capi.F_ADJACENCY:
            for k in ("pair_lo", "pair_hi", "pair_faces"):
                assert np.array_equal(got[k], want[k]), (hex(features), k)
    spills = gpu_ctx.debug_counters()
    assert spills["label_spills"] > 0, spills          # (the case does what it is meant to)


def test_repeated_sweeps_of_a_spilling_volume_adapt_the_tile_height(gpu_ctx):
    """The automatic tile height is halved (for the following sweeps of the same resident volume) when the workgroup
    tables overflow: same results, fewer spills."""
    # (blocks small enough that a tile of the default height -- 40 planes of the narrow shape since round 5 -- overflows its tables:
    #  180 k spills in the first sweep, 5 k from the second on; scripts/r05_spill_probe.py)
    vol = random_blocks((96, 64, 512), 60000, 18, np.uint32, block=(4, 3, 9))
    want = onepass_c.extract(vol)
    gpu_ctx.set_option(_capi.OPT_TILE_PLANES, 0)
    gpu_ctx.set_option(_capi.OPT_IMPL, 0)
    gpu_ctx.set_volume(vol)
    L = int(vol.max())
    spills, ms = [], []
    for _ in range(5):
        gpu_ctx.extract(_capi.F_ALL, L)
        count, bbox, sum1, sum2 = gpu_ctx.labels()
        lo, hi, faces = gpu_ctx.adjacency()
        assert np.array_equal(count, want["count"]) and np.array_equal(bbox, want["bbox"])
        assert np.array_equal(sum1, want["sum1"]) and np.array_equal(sum2, want["sum2"])
        assert np.array_equal(lo, want["pair_lo"]) and np.array_equal(hi, want["pair_hi"]) and np.array_equal(faces, want["pair_faces"])
        d = gpu_ctx.debug_counters()
        spills.append(d["label_spills"] + d["pair_spills"])
        ms.append(gpu_ctx.timing()["ms_sweep"])
    assert spills[0] > 0 and spills[-1] < spills[0] // 4, spills
    # (time: shorter tiles trade spills for table flushes; with persistent workgroups the two about cancel on this volume)
    assert min(ms[2:]) < 1.5 * ms[0], ms


@pytest.mark.parametrize("shape,dtype", [((1, 1, 68), np.uint32), ((3, 1, 64), np.uint16), ((2, 5, 260), np.uint32), ((4, 17, 1000), np.uint32),
                                         ((2, 3, 8), np.uint16), ((5, 18, 520), np.uint16), ((1, 68, 1), np.uint32)])
def test_partial_tiles_of_volumes_with_aligned_rows(gpu_ctx, shape, dtype):
    """Rows of a multiple of 16 bytes that are not a multiple of the tile: the partial tiles run the PADDED kernel --
    interior-style loads from clamped addresses, filler written over what lies outside (waves entirely below the last
    row, lanes entirely right of the last column; a (1, 68, 1) array is one row of 68 in memory)."""
    rng = np.random.default_rng(sum(shape))
    vol = random_blocks(shape, 9, int(rng.integers(1, 99)), dtype, block=(2, 2, 7))
    want = onepass_c.extract(vol)
    for features in (_capi.F_ALL, 0x0f):
        got = run(gpu_ctx, vol, 0, features=features)
        for k in ("count", "bbox", "sum1", "sum2"):
            assert np.array_equal(got[k], want[k]), (shape, hex(features), k)
        if features & _capi.F_ADJACENCY:
            for k in ("pair_lo", "pair_hi", "pair_faces"):
                assert np.array_equal(got[k], want[k]), (shape, k)


@pytest.mark.parametrize("slack", [False, True], ids=["exact_buffer", "storage_with_slack"])
@pytest.mark.parametrize("shape,dtype", [((9, 11, 23), np.uint16), ((6, 20, 301), np.uint32), ((5, 33, 1030), np.uint16), ((3, 7, 1), np.uint32)])
def test_adopted_buffers_with_unaligned_rows(shape, dtype, slack):
    """A device buffer the library did not allocate itself and that ends with the volume has no slack behind it: with rows
    that are not a multiple of 16 bytes it runs the plain edge kernel (guarded scalar loads).  A tensor that is a view of a
    larger storage (tissue_analysis_amd.device.empty_volume makes them) tells the library so (TA_OPT_VOLUME_SLACK) and gets
    the 16-byte loads.  Same results either way."""
    import torch
    vol = random_blocks(shape, 25, 77, dtype, block=(2, 3, 9))
    want = onepass_c.extract(vol)
    host = torch.from_numpy(vol.view({np.uint16: np.int16, np.uint32: np.int32}[dtype]).copy())
    if slack:
        flat = torch.empty((vol.size + 16,), dtype=host.dtype, device="cuda:0")
        t = flat[:vol.size].view(*vol.shape)
        t.copy_(host)
    else:
        t = host.to("cuda:0")
    ctx = _capi.Context(0)
    try:
        ctx.set_volume_device(t.data_ptr(), vol.dtype.itemsize, vol.shape, keep=t)
        assert (ctx.get_option(_capi.OPT_VOLUME_SLACK) >= 16) == slack
        for features in (_capi.F_ALL, 0x0f):
            ctx.extract(features, int(vol.max()))
            count, bbox, sum1, sum2 = ctx.labels()
            assert np.array_equal(count, want["count"]) and np.array_equal(bbox, want["bbox"])
            assert np.array_equal(sum1, want["sum1"]) and np.array_equal(sum2, want["sum2"])
            if features & _capi.F_ADJACENCY:
                lo, hi, faces = ctx.adjacency()
                assert np.array_equal(lo, want["pair_lo"]) and np.array_equal(hi, want["pair_hi"]) and np.array_equal(faces, want["pair_faces"])
    finally:
        ctx.close()


SHAPE_CASES = [
    ("voronoi_1024_wide", lambda: voronoi((20, 24, 1024), 60, 21, np.uint32)),
    ("left_boundary_512", lambda: _half_and_half((26, 32, 512), np.uint32)),
    ("flat_wall_512", lambda: _flat_wall((5, 16, 512), np.uint32)),
    ("ragged_520", lambda: voronoi((9, 20, 520), 40, 22, np.uint32)),          # forced wide: a partial tile column
    ("short_rows_300", lambda: voronoi((7, 13, 300), 20, 23, np.uint32)),       # forced wide: no whole tile at all
    ("dense_blocks_1024", lambda: random_blocks((6, 16, 1024), 3000, 24, np.uint32)),
]


@pytest.mark.parametrize("shape", [0, 1], ids=["narrow", "wide"])
@pytest.mark.parametrize("name,make", SHAPE_CASES, ids=[c[0] for c in SHAPE_CASES])
def test_both_tile_shapes_of_the_uint32_sweep_match_the_oracle(gpu_ctx, name, make, shape):
    """TA_OPT_SWEEP_SHAPE: two rows of 256 columns a wave (0) or of 512 (1) -- the same integers either way, also where the
    wide tiles are forced onto rows they would not be chosen for."""
    vol = make()
    want = onepass_c.extract(vol)
    gpu_ctx.set_option(_capi.OPT_SWEEP_SHAPE, shape)
    try:
        for tp in (0, 5):
            for feats in (_capi.F_ALL, 0x17):
                got = run(gpu_ctx, vol, 0, feats, tile_planes=tp)
                w = dict(want, sum2=np.zeros_like(want["sum2"])) if not feats & _capi.F_MOMENT2 else want
                assert_same_accumulators(got, w, "%s shape=%d tp=%d feats=0x%x" % (name, shape, tp, feats))
    finally:
        gpu_ctx.set_option(_capi.OPT_SWEEP_SHAPE, -1)
        gpu_ctx.set_option(_capi.OPT_TILE_PLANES, 0)


def test_the_measured_choice_of_the_shape_never_changes_the_result(gpu_ctx):
    """TA_OPT_SWEEP_SHAPE = -2: the first four sweeps of a volume take turns between the shapes and the faster one keeps it:
    eight sweeps of one resident volume, every one equal to the oracle; -1 (the default) decides before the first sweep."""
    vol = voronoi((24, 32, 1024), 80, 25, np.uint32)
    want = onepass_c.extract(vol)
    try:
        for rule in (-2, -1):
            gpu_ctx.set_option(_capi.OPT_SWEEP_SHAPE, rule)
            gpu_ctx.set_volume(vol)
            used = []
            for k in range(8):
                gpu_ctx.extract(_capi.F_ALL, int(vol.max()))
                used.append(gpu_ctx.get_option(_capi.OPT_SWEEP_SHAPE_USED))
                count, bbox, s1, s2 = gpu_ctx.labels()
                lo, hi, f = gpu_ctx.adjacency()
                got = dict(max_label=int(vol.max()), count=count, bbox=bbox, sum1=s1, sum2=s2, pair_lo=lo, pair_hi=hi, pair_faces=f)
                assert_same_accumulators(got, want, "rule %d sweep %d" % (rule, k))
            if rule == -2:
                assert used[:4] == [1, 0, 1, 0] and len(set(used[4:])) == 1
            else:
                assert len(set(used)) == 1                     # one decision, before the first sweep
    finally:
        gpu_ctx.set_option(_capi.OPT_SWEEP_SHAPE, -1)


def test_the_first_sweep_of_a_volume_already_runs_the_faster_shape():
    """a caller that sweeps a volume ONCE (SpatialImageAnalysis(image)) must not get the un-tuned shape.  The
    default rule looks at the label changes per voxel in eight sampled planes before the first sweep: tissue with background
    around it (C4's cells inside the ellipsoid: 0.023 changes a voxel) takes the wide tiles, the same cells everywhere (0.054)
    the narrow ones -- the shapes that are faster on C4 and on the tissue-filled C4 (profiles/NOTES.md, round 5)."""
    import torch
    from tissue_analysis_amd import device as dev, synth
    c = synth.CONFIGS["C4"]
    dims = (64, 256, 1024)
    cells = max(8, c["n_cells"] * dims[0] * dims[1] * dims[2] // (1024 ** 3))
    for ellipsoid, want_shape in ((True, 1), (False, 0)):
        ctx = dev.torch_context(0)
        vol, L = dev.synth_slab(ctx, dims, np.dtype(np.uint32), cells, c["seed"], ellipsoid=ellipsoid)
        torch.cuda.synchronize()
        ctx.set_volume_device(vol.data_ptr(), 4, vol.shape, keep=vol)
        ctx.extract(_capi.F_ALL, L)
        ctx.synchronize()
        assert ctx.get_option(_capi.OPT_SWEEP_SHAPE_USED) == want_shape, "ellipsoid=%s" % ellipsoid
        ctx.close()
    # ... and through the class: one construction, one sweep, the narrow shape on dense tissue
    from tissue_analysis_amd import DICT, SpatialImageAnalysis
    img = voronoi((48, 64, 512), 400, 31, np.uint32, ellipsoid=False)
    sia = SpatialImageAnalysis(img, ignoredlabels=0, return_type=DICT, background=1)
    assert sia._resident().ctx.get_option(_capi.OPT_SWEEP_SHAPE_USED) == 0
