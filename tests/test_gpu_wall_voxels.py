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


def _flat_symmetric_walls():
    """Cuboid cells: flat walls whose geometric median sits ON integer coordinates -- the truncation that follows the Weiszfeld
    iteration is then decided by the last bit of sums that must be taken in the reference's order."""
    v = np.ones((20, 24, 40), dtype=np.uint16)
    v[2:10, 2:11, 3:20] = 2
    v[10:18, 2:11, 3:20] = 3
    v[2:10, 11:21, 3:20] = 4
    v[10:18, 11:21, 3:21] = 5
    v[2:18, 2:21, 21:37] = 6
    return v


@pytest.mark.parametrize("make", [lambda: voronoi((36, 40, 44), 40, 71, np.uint16), lambda: voronoi((30, 28, 260), 50, 72, np.uint32),
                                  _flat_symmetric_walls, lambda: random_blocks((14, 18, 30), 60, 73, np.uint32)],
                         ids=["voronoi_u16", "voronoi_u32", "flat_symmetric", "blocks_u32"])
def test_wall_medians_on_the_device_equal_the_host_arithmetic(gpu_ctx, make):
    """ta_wall_medians (one wave a wall: the reference's Weiszfeld rules in IEEE double, sums in record order, truncation,
    nearest wall voxel) against geometry.median_voxels on the records the device grouped by pair: the same voxel for EVERY wall."""
    from tissue_analysis_amd import geometry
    vol = make()
    gpu_ctx.set_volume(vol)
    keys, sizes, med, ms, moving = gpu_ctx.wall_medians()
    assert not moving.any()
    lo, hi, coords, _ = gpu_ctx.wall_voxels(by_pair=True)
    k = (lo.astype(np.uint64) << np.uint64(32)) | hi.astype(np.uint64)
    uk, first, count = np.unique(k, return_index=True, return_counts=True)
    assert np.array_equal(keys, uk) and np.array_equal(sizes, count.astype(np.uint32)) and np.all(first == np.cumsum(count) - count)
    want = geometry.median_voxels(coords.astype(np.int64), count)
    assert np.array_equal(med.astype(np.int64), want.reshape(-1, 3)), int((med != want.reshape(-1, 3)).any(axis=1).sum())
    assert ms > 0 and keys.size > 5


def test_graph_wall_medians_take_the_device_path():
    """graph_from_image(..., 'wall_median') on an analysis that holds nothing but its resident volume: the medians come from
    ta_wall_medians (no wall table is pulled to the host) and equal the host arithmetic on the wall table."""
    from tissue_analysis_amd import SpatialImageAnalysis3D, graph_from_image
    vol = voronoi((30, 34, 40), 30, 74, np.uint16)
    sia = SpatialImageAnalysis3D(vol, ignoredlabels=0, background=1)
    g = graph_from_image(sia, spatio_temporal_properties=['wall_median'], background=1, ignore_cells_at_stack_margins=False)
    assert sia._walls is None and sia._wall_medians                        # the device answered; no table on the host
    sia2 = SpatialImageAnalysis3D(vol, ignoredlabels=0, background=1)
    sia2.wall_table()                                                       # a table on the host: the host arithmetic answers
    g2 = graph_from_image(sia2, spatio_temporal_properties=['wall_median'], background=1, ignore_cells_at_stack_margins=False)
    a, b = g.edge_property('wall_median'), g2.edge_property('wall_median')
    assert len(a) > 20 and dict(a.items()) == dict(b.items())
    for name in ('epidermis_wall_median', 'unlabelled_wall_median'):
        assert dict(g.vertex_property(name).items()) == dict(g2.vertex_property(name).items())


def test_a_wall_that_does_not_settle_is_marked_and_only_raises_when_asked_for(gpu_ctx):
    """One Weiszfeld pass is never enough (the stopping rule needs five): every wall comes back MARKED instead of the call
    failing, and the class raises -- like the reference, SIA:1630-1633 -- only for a wall that is asked for."""
    from tissue_analysis_amd import SpatialImageAnalysis3D
    vol = voronoi((20, 24, 28), 12, 75, np.uint16)
    gpu_ctx.set_volume(vol)
    keys, sizes, med, ms, moving = gpu_ctx.wall_medians(max_iter=1)
    assert keys.size > 5 and moving.all() and sizes.max() < 2 ** 31
    keys2, sizes2, med2, _, moving2 = gpu_ctx.wall_medians()
    assert not moving2.any() and np.array_equal(keys, keys2) and np.array_equal(sizes, sizes2)
    sia = SpatialImageAnalysis3D(vol, ignoredlabels=0, background=1)
    have, sz, m, mv = sia._resident().wall_medians()
    mv = mv.copy(); mv[3] = True                                           # pretend wall 3 did not settle
    sia._wall_medians = (have, sz, m, mv)
    found, got = sia.wall_medians_of(have[[0, 1, 2]])                       # not asked for: no error
    assert found.all() and np.array_equal(got, m[[0, 1, 2]])
    with pytest.raises(ValueError):
        sia.wall_medians_of(have[[2, 3]])
