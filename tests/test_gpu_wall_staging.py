"""The wall-voxel pass (kernels_walls.hip) on the inputs that leave its main road: cells whose records are not staged
(regions too small, no staging area at all, voxels with more than three neighbour labels), labels from 2^31 up, strips
that end inside a row, rows that are not aligned for vector loads -- every record against a brute-force restatement of
the 18-neighbourhood (helpers.brute_wall_records; SIA:759-806 semantics)."""
import os

import numpy as np
import pytest

from tissue_analysis_amd.extraction import ResidentVolume

from helpers import brute_wall_records, voronoi

pytestmark = pytest.mark.gpu


def check(vol, ctx=None):
    want_lo, want_hi, want_co = brute_wall_records(np.ascontiguousarray(vol))
    rv = None
    if ctx is None:
        rv = ResidentVolume(vol)
        ctx = rv.ctx
    try:
        glo, ghi, gco, ms = ctx.wall_voxels(by_pair=True)
        assert np.array_equal(glo, want_lo) and np.array_equal(ghi, want_hi) and np.array_equal(gco, want_co)
        lo, hi, co, _ = ctx.wall_voxels()                       # memory order: positions never decrease, same records
        pos = np.ravel_multi_index((co[:, 0], co[:, 1], co[:, 2]), vol.shape) if lo.size else np.zeros(0, np.int64)
        assert np.all(pos[1:] >= pos[:-1])
        order = np.lexsort((pos, hi, lo))
        assert np.array_equal(lo[order], want_lo) and np.array_equal(hi[order], want_hi) and np.array_equal(co[order], want_co)
        return lo.size
    finally:
        if rv is not None:
            rv.close()


@pytest.fixture
def stage_records():
    def set_to(n):
        if n is None:
            os.environ.pop("TA_WALL_STAGE_RECORDS", None)
        else:
            os.environ["TA_WALL_STAGE_RECORDS"] = str(n)
    yield set_to
    os.environ.pop("TA_WALL_STAGE_RECORDS", None)


@pytest.mark.parametrize("records", [None, 0, 256 * 24, 256 * 300])
def test_cells_that_are_not_staged_take_the_second_walk(stage_records, records):
    """No staging area / regions of 24 records (most cells do not fit) / of 300 (some do not) / the default."""
    stage_records(records)
    for dtype in (np.uint16, np.uint32):
        n = check(voronoi((20, 40, 520), 60, 71, dtype))
        assert n > 10000


def test_voxels_with_many_neighbour_labels(stage_records):
    """Noise: up to 18 labels around a voxel -- more than the three a staged cell keeps per voxel."""
    rng = np.random.default_rng(72)
    for dtype in (np.uint16, np.uint32):
        vol = rng.integers(1, 40, size=(6, 20, 300)).astype(dtype)
        assert check(vol) > 10 * vol.size
    vol = voronoi((10, 24, 300), 30, 73, np.uint16)
    vol[3:6, 5:9, 100:140] = rng.integers(1, 30, size=(3, 4, 40))          # a patch of noise inside tissue
    check(vol)


def test_labels_from_two_to_the_31_up():
    vol = voronoi((12, 20, 280), 40, 74, np.uint32)
    wide = vol.copy()
    wide[vol % 3 == 0] += np.uint32(1 << 31)
    wide[vol == 7] = np.uint32(0xFFFFFFFF)
    wide[vol == 8] = 0
    check(wide)
    check(vol + np.uint32(0x7FFFFF00))              # differences stay small, labels straddle 2^31
    rng = np.random.default_rng(77)                 # noise over the whole 32-bit range: voxels ALL of whose neighbours differ in
    for shape in ((4, 6, 40), (3, 5, 300)):         # bit 31 (the first kernel's signed maximum is then none of them: it must
        check(rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32))      # still end, and hand over to WIDE)
    two = rng.integers(0, 2, size=(5, 7, 64)).astype(np.uint32) * np.uint32(0x80000001) + np.uint32(5)
    check(two)


@pytest.mark.parametrize("shape", [(5, 9, 260), (3, 4, 516), (2, 18, 1028), (3, 35, 256), (4, 17, 252), (1, 16, 4), (2, 33, 259)])
def test_strips_that_end_inside_a_row(shape):
    rng = np.random.default_rng(sum(shape))
    for dtype in (np.uint16, np.uint32):
        check(voronoi(shape, 25, 75 + shape[2], dtype))
        vol = rng.integers(1, 4, size=shape).astype(dtype)
        check(vol)


def test_rows_that_are_not_aligned_for_vector_loads():
    import torch
    from tissue_analysis_amd import device as dev
    for dtype, tdtype in ((np.uint16, torch.int16), (np.uint32, torch.int32)):
        vol = voronoi((6, 12, 264), 20, 76, dtype)
        ctx = dev.torch_context(0)
        flat = torch.zeros(vol.size + 3, dtype=tdtype, device="cuda:0")
        for shift in (1, 2, 3):
            view = flat[shift:shift + vol.size].view(vol.shape)
            view.copy_(torch.from_numpy(vol.view(np.int16 if dtype == np.uint16 else np.int32)).to("cuda:0"))
            ctx.set_volume_device(view.data_ptr(), np.dtype(dtype).itemsize, vol.shape, keep=view)
            check(vol, ctx)


@pytest.mark.parametrize("keyed", ["1", "0"])
def test_the_grouped_fetch_from_keys_and_from_records(stage_records, keyed):
    """The grouped fetch sorts KEYS and linear voxel indices the fetch kernels write themselves (volumes below 2^32 voxels;
    coordinates rebuilt from the index in the last pass); TA_WALL_KEYED=0 sorts the records of the plain fetch, as larger
    volumes do.  Both against the brute force: two-label keys of 32 and 64 bits, the second walk."""
    os.environ["TA_WALL_KEYED"] = keyed
    try:
        check(voronoi((20, 33, 300), 50, 81, np.uint16))
        big = voronoi((12, 17, 140), 30, 82, np.uint32)
        big[big > 0] += 70000                                            # 17-bit labels: 64-bit sort keys
        check(big)
        stage_records(0)                                                 # every cell through the second walk
        check(voronoi((9, 20, 260), 30, 85, np.uint16))
    finally:
        os.environ.pop("TA_WALL_KEYED", None)


def test_the_two_grouped_fetches_agree_on_axes_that_are_not_c_ordered():
    """(a pair's voxels come in MEMORY order, which only for C-ordered images is the order of the brute force: here the fetch
    from keys -- coordinates rebuilt from the linear index through the axis permutation -- against the fetch from records)"""
    for vol in (np.asfortranarray(voronoi((14, 18, 40), 25, 83, np.uint16)), np.transpose(voronoi((10, 24, 36), 25, 84, np.uint32), (1, 0, 2))):
        got = {}
        for keyed in ("1", "0"):
            os.environ["TA_WALL_KEYED"] = keyed
            rv = ResidentVolume(vol)
            try:
                got[keyed] = rv.ctx.wall_voxels(by_pair=True)[:3]
            finally:
                rv.close()
                os.environ.pop("TA_WALL_KEYED", None)
        for a, b in zip(got["1"], got["0"]):
            assert a.size and np.array_equal(a, b)
        lo, hi, co = got["1"]
        own = vol[co[:, 0], co[:, 1], co[:, 2]].astype(np.int64)
        assert np.all((own == lo) | (own == hi))
