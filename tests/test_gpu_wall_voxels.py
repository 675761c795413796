"""Wall voxels (SIA:759-880, 1049-1111): the one-pass GPU extraction against the
reference's per-pair crops + 18-connectivity dilations restated in the oracle."""
import numpy as np
import pytest

from oracle import sia_oracle
from oracle.sia_oracle import OracleSIA
from tissue_analysis_amd import DICT, SpatialImage, SpatialImageAnalysis, synth
from tissue_analysis_amd.extraction import wall_voxel_table

from helpers import random_blocks, voronoi

pytestmark = pytest.mark.gpu
VS = synth.PARITY_VOXELSIZE


def both(vol):
    sia = SpatialImageAnalysis(SpatialImage(vol, voxelsize=VS), ignoredlabels=0, return_type=DICT, background=1)
    ref = OracleSIA(vol, ignoredlabels=0, return_type=sia_oracle.DICT, background=1, voxelsize=VS)
    return sia, ref


def same_dict(got, want):
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("make", [
    lambda: voronoi((30, 28, 70), 30, 61, np.uint16),
    lambda: np.asfortranarray(voronoi((22, 26, 40), 20, 62, np.uint32)),
    lambda: random_blocks((9, 11, 23), 12, 63, np.uint16),
], ids=["voronoi_u16", "voronoi_u32_fortran", "blocks"])
def test_every_pair_matches_the_reference_dilations(make):
    vol = make()
    sia, ref = both(vol)
    table = sia.wall_table()
    neigh = ref.neighbors()
    checked = 0
    for a in sorted(neigh):
        for b in neigh[a]:
            if a < b:
                want = ref.wall_voxels_between_two_cells(a, b)
                got = sia.wall_voxels_between_two_cells(a, b)
                assert got.shape == want.shape and np.array_equal(got, want), (a, b)
                checked += 1
    assert checked > 10 and len(table) >= checked          # 18-connectivity also finds edge-only contacts
    far = [l for l in ref.labels() if l not in neigh[ref.labels()[-1]] and l != ref.labels()[-1]]
    if far:                                                 # two labels that do not touch: an empty 3x0 array
        pair = sia.wall_voxels_between_two_cells(ref.labels()[-1], far[0])
        assert pair.shape == (3, 0) or not np.array_equal(pair, pair[:, :0])  # (edge-only contacts are still contacts)
        assert ref.wall_voxels_between_two_cells(ref.labels()[-1], far[0]).shape == pair.shape


def test_per_cell_and_per_pairs_methods():
    vol = voronoi((30, 28, 70), 30, 64, np.uint16)
    sia, ref = both(vol)
    cells = [l for l in ref.labels() if l != 1]
    one = cells[len(cells) // 2]
    same_dict(sia.wall_voxels_per_cell(one), ref.wall_voxels_per_cell(one))
    nb = sorted(ref.neighbors(one))
    same_dict(sia.wall_voxels_per_cell(one, neighbors=list(nb), neighbors2ignore=[nb[0]]),
              ref.wall_voxels_per_cell(one, list(nb), [nb[0]]))
    same_dict(sia.wall_voxels_per_cell(one, neighbors=nb[1]), ref.wall_voxels_per_cell(one, nb[1]))

    some = cells[2:12]
    neigh = ref.neighbors(list(some))
    for ignore_bg in (False, True):
        got = sia.wall_voxels_per_cells_pairs(list(some), dict((k, list(v)) for k, v in neigh.items()),
                                              ignore_background=ignore_bg, verbose=False)
        want = ref.wall_voxels_per_cells_pairs(list(some), dict((k, list(v)) for k, v in neigh.items()),
                                               ignore_background=ignore_bg)
        same_dict(got, want)
        assert any(k[0] == 1 for k in want) != ignore_bg or not any(1 in neigh[l] for l in some)
    same_dict(sia.wall_voxels_per_cells_pairs(verbose=False), ref.wall_voxels_per_cells_pairs())
    same_dict(sia.wall_voxels_per_cells_pairs(only_epidermis=True, verbose=False),
              ref.wall_voxels_per_cells_pairs(only_epidermis=True))


def test_table_properties_on_a_larger_volume():
    """Size-independent checks at 160^3: every record's voxel carries one label of its pair and has an
    18-neighbour with the other; face-adjacent pairs of the sweep are a subset of the wall pairs."""
    vol = voronoi((160, 160, 160), 400, 65, np.uint16)
    t = wall_voxel_table(vol)
    lo = (t.key >> np.uint64(32)).astype(np.int64)
    hi = (t.key & np.uint64(0xffffffff)).astype(np.int64)
    here = vol[t.coords[:, 0], t.coords[:, 1], t.coords[:, 2]].astype(np.int64)
    assert np.all((here == lo) | (here == hi)) and np.all(lo < hi)
    other = np.where(here == lo, hi, lo)
    rng = np.random.default_rng(66)
    for r in rng.integers(0, t.key.size, 300):
        x, y, z = t.coords[r]
        box = vol[max(0, x - 1):x + 2, max(0, y - 1):y + 2, max(0, z - 1):z + 2]
        assert (box == other[r]).any()
    sia = SpatialImageAnalysis(vol, background=1)
    faces = set(zip(sia.extraction.pair_lo.tolist(), sia.extraction.pair_hi.tolist()))
    walls = set(zip(lo.tolist(), hi.tolist()))
    assert faces <= walls
    assert t.ms is not None and t.ms > 0


def test_records_grouped_on_the_device_equal_the_host_sort():
    """ta_wall_voxels_get_by_pair (stable radix sort by pair on the device) against ta_wall_voxels_get + the stable host
    sort WallTable used to do -- same keys, same coordinates, same order."""
    from tissue_analysis_amd.extraction import ResidentVolume, WallTable
    for vol in (voronoi((48, 40, 300), 60, 67, np.uint16), voronoi((31, 33, 70), 40, 68, np.uint32),
                random_blocks((9, 11, 23), 12, 69, np.uint16)):
        rv = ResidentVolume(vol)
        try:
            lo, hi, coords, _ = rv.ctx.wall_voxels()
            want = WallTable(lo, hi, coords)
            glo, ghi, gcoords, ms = rv.ctx.wall_voxels(by_pair=True)
            got = WallTable(glo, ghi, gcoords, ms, grouped=True)
            assert np.array_equal(got.key, want.key) and np.array_equal(got.coords, want.coords)
            assert np.array_equal(got.pairs, want.pairs) and np.array_equal(got.start, want.start)
            assert np.all(got.key[1:] >= got.key[:-1])
            table = rv.wall_table()                  # the product path takes the grouped fetch for C-ordered images
            assert np.array_equal(table.key, want.key) and np.array_equal(table.coords, want.coords)
        finally:
            rv.close()


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 7, 1), (4, 6, 1), (5, 1, 9), (2, 2, 2), (3, 17, 5), (1, 1, 300), (2, 3, 257)])
def test_degenerate_and_ragged_shapes(shape):
    """Planes / rows / columns of extent 1, a strip of one column past 256, odd sizes: every pair's wall voxels against
    the reference's dilations (the kernel clamps out-of-volume neighbours instead of testing for them)."""
    rng = np.random.default_rng(sum(shape))
    for dtype in (np.uint16, np.uint32):
        vol = rng.integers(1, 5, size=shape).astype(dtype)
        sia, ref = both(vol)
        neigh = ref.neighbors()
        labels = ref.labels()
        for a in labels:
            for b in labels:
                if a < b:
                    want = ref.wall_voxels_between_two_cells(a, b)
                    got = sia.wall_voxels_between_two_cells(a, b)
                    assert got.shape == want.shape and np.array_equal(got, want), (shape, dtype, a, b)
