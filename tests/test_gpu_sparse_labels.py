"""Sparse label ids (np.unique takes any ids, SIA:358-364): the census of the ids on the device, the sweep over the volume
written in their ranks, rows mapped back -- against the oracle run on the np.unique ranks of the same volume."""
import numpy as np
import pytest

from oracle import onepass
from tissue_analysis_amd import _capi
from tissue_analysis_amd.extraction import Extraction, ResidentVolume, extract_resident, wants_compaction

from helpers import random_blocks, voronoi

pytestmark = pytest.mark.gpu


def scatter_ids(vol, seed, top):
    """The same partition of the voxels with its labels renamed to random ids in [0, top], order kept or not."""
    rng = np.random.default_rng(seed)
    old = np.unique(vol)
    new = np.sort(rng.choice(np.arange(top + 1, dtype=np.uint64) if top < (1 << 22) else
                             np.unique(rng.integers(0, top + 1, size=4 * old.size + 8, dtype=np.uint64)), size=old.size, replace=False))
    rng.shuffle(new)                                            # (an id's rank is NOT the rank of the label it replaces)
    lut = np.zeros(int(old.max()) + 1, dtype=np.uint64)
    lut[old] = new
    return lut[vol].astype(np.uint32)


def oracle_by_rank(vol):
    ids, inv = np.unique(vol, return_inverse=True)
    ranks = inv.reshape(vol.shape).astype(np.uint32)
    want = onepass.extract(ranks, max_label=ids.size - 1)
    return ids.astype(np.int64), want


def assert_sparse_equal(x, ids, want):
    assert x.sparse and np.array_equal(x.ids, ids)
    assert x.max_label == int(ids[-1]) and x.nrows == ids.size
    for k in ("count", "bbox", "sum1", "sum2"):
        assert np.array_equal(np.asarray(getattr(x, k)).reshape(-1), np.asarray(want[k]).reshape(-1)), k
    assert np.array_equal(x.pair_lo.astype(np.int64), ids[np.asarray(want["pair_lo"], dtype=np.int64)])
    assert np.array_equal(x.pair_hi.astype(np.int64), ids[np.asarray(want["pair_hi"], dtype=np.int64)])
    assert np.array_equal(x.pair_faces, np.asarray(want["pair_faces"]).reshape(-1, 3))


CASES = [
    ("voronoi_near_2_31", lambda: scatter_ids(voronoi((24, 32, 256), 50, 1, np.uint32), 1, (1 << 31) + 12345)),
    ("voronoi_full_range", lambda: scatter_ids(voronoi((9, 20, 520), 40, 2, np.uint32), 2, (1 << 32) - 1)),
    ("blocks_above_2_28", lambda: scatter_ids(random_blocks((11, 37, 70), 300, 3, np.uint32), 3, (1 << 29))),
    ("blocks_moderate", lambda: scatter_ids(random_blocks((17, 13, 29), 40, 4, np.uint32), 4, (1 << 20))),
    ("unaligned_rows", lambda: scatter_ids(voronoi((7, 9, 131), 20, 5, np.uint32), 5, (1 << 30))),
    # rows that are not whole 16-byte vectors: the marking pass takes the volume as one run of pseudo-rows plus a remainder
    ("odd_rows_many_ids", lambda: scatter_ids(random_blocks((33, 37, 131), 900, 6, np.uint32), 6, (1 << 31))),
    ("odd_rows_long", lambda: scatter_ids(voronoi((5, 11, 2051), 60, 7, np.uint32), 7, (1 << 32) - 1)),
]


@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
def test_sparse_ids_match_the_oracle_on_their_ranks(gpu_ctx, name, make):
    vol = make()
    ids, want = oracle_by_rank(vol)
    gpu_ctx.set_volume(vol)
    top, present = gpu_ctx.label_census()
    assert top == int(vol.max()) and np.array_equal(present.astype(np.int64), ids)           # np.unique on the device
    assert wants_compaction(top, present.size)
    x = extract_resident(gpu_ctx, vol.shape)
    assert gpu_ctx.is_compact()
    assert_sparse_equal(x, ids, want)
    # a second sweep of the compacted context, and the volume itself is still the one that was uploaded
    assert_sparse_equal(extract_resident(gpu_ctx, vol.shape), ids, want)
    back = np.zeros_like(vol)
    gpu_ctx.get_volume(back)
    assert np.array_equal(back, vol)


def test_ids_with_the_top_bits_set_including_the_largest_uint32(gpu_ctx):
    vol = np.zeros((4, 8, 64), dtype=np.uint32)
    vol[:, :, 10:30] = 0xFFFFFFFF
    vol[:, 2:5, 30:50] = 0xFFFFFFFE
    vol[2:, :, 50:] = 0x80000000
    ids, want = oracle_by_rank(vol)
    gpu_ctx.set_volume(vol)
    x = extract_resident(gpu_ctx, vol.shape)
    assert_sparse_equal(x, ids, want)
    assert x.present().tolist() == [0, 0x80000000, 0xFFFFFFFE, 0xFFFFFFFF]
    assert x.neighbors_of(0xFFFFFFFF) == [0, 0xFFFFFFFE] and x.neighbors_of(0xFFFFFFFE) == [0, 0x80000000, 0xFFFFFFFF]
    assert x.has(0xFFFFFFFE) and not x.has(7) and x.bbox_slices(0x80000000) == (slice(2, 4), slice(0, 8), slice(50, 64))


def test_dense_ids_are_left_dense_and_uint16_volumes_can_be_compacted_when_asked(gpu_ctx):
    vol = voronoi((12, 16, 128), 30, 6, np.uint16)
    gpu_ctx.set_volume(vol)
    x = extract_resident(gpu_ctx, vol.shape)
    assert not x.sparse and not gpu_ctx.is_compact()
    want = onepass.extract(vol.astype(np.uint32), max_label=int(vol.max()))
    for k in ("count", "bbox", "sum1", "sum2"):
        assert np.array_equal(np.asarray(getattr(x, k)).reshape(-1), np.asarray(want[k]).reshape(-1)), k
    wide = (vol.astype(np.uint32) * 997 % 65521).astype(np.uint16)           # ids spread over uint16, some labels merged
    ids, want = oracle_by_rank(wide)
    gpu_ctx.set_volume(wide)
    assert_sparse_equal(extract_resident(gpu_ctx, wide.shape, sparse=True), ids, want)
    odd = np.ascontiguousarray(wide[:, :, :123])                              # uint16 rows of 246 bytes
    ids, want = oracle_by_rank(odd)
    gpu_ctx.set_volume(odd)
    assert_sparse_equal(extract_resident(gpu_ctx, odd.shape, sparse=True), ids, want)


def test_the_host_accessors_answer_in_ids(gpu_ctx):
    dense = voronoi((20, 24, 96), 25, 7, np.uint32)
    vol = scatter_ids(dense, 7, (1 << 31) + 99)
    gpu_ctx.set_volume(dense)
    xd = extract_resident(gpu_ctx, dense.shape)
    gpu_ctx.set_volume(vol)
    xs = extract_resident(gpu_ctx, vol.shape)
    assert xs.sparse and not xd.sparse
    # the id that replaced each dense label
    to_id = {}
    for l in xd.present().tolist():
        z = np.argwhere(dense == l)[0]
        to_id[l] = int(vol[tuple(z)])
    labels = xd.present().tolist()
    sid = [to_id[l] for l in labels]
    assert np.array_equal(xs.volumes(sid), xd.volumes(labels))
    assert np.array_equal(xs.barycenters(sid), xd.barycenters(labels))
    assert np.allclose(xs.covariances(sid), xd.covariances(labels), rtol=0, atol=0)
    assert np.array_equal(xs.surface_faces(sid + [5]), np.concatenate([xd.surface_faces(labels), np.zeros((1, 3), np.uint64)]))
    assert np.array_equal(xs.degrees_of(sid + [5]), np.concatenate([xd.degrees_of(labels), [0]]))
    for l in labels[:8]:
        assert sorted(xs.neighbors_of(to_id[l])) == sorted(to_id[n] for n in xd.neighbors_of(l))
        assert xs.bbox_slices(to_id[l]) == xd.bbox_slices(l)
        nb = xd.neighbors_of(l)
        assert np.array_equal(xs.faces_between(to_id[l], sorted(to_id[n] for n in nb)).sum(axis=0), xd.faces_between(l, nb).sum(axis=0))
    lists = xs.neighbor_lists(sid[:5] + [5])
    assert lists[5] == [] and all(lists[to_id[l]] == xs.neighbors_of(to_id[l]) for l in labels[:5])
    rows = xs.neighbor_rows(sid[:5])
    assert all(rows[to_id[l]] == xs.neighbors_of(to_id[l]) for l in labels[:5])
    boxes = xs.bbox_slices_upto(xs.max_label)
    assert len(boxes) == xs.max_label and boxes[sid[3] - 1] == xs.bbox_slices(sid[3]) and boxes[10] is None
    with pytest.raises(IndexError):
        xs.volumes([5])


def test_a_given_id_list_ranks_every_slab_alike_and_must_cover_the_volume(gpu_ctx):
    vol = scatter_ids(voronoi((16, 16, 128), 20, 8, np.uint32), 8, (1 << 30))
    ids = np.unique(vol)
    union = np.unique(np.concatenate([ids, np.array([3, 77, (1 << 30) + 5], dtype=np.uint32)]))
    gpu_ctx.set_volume(vol[:8])
    table = gpu_ctx.compact_labels(union)
    assert np.array_equal(table, union)
    x = extract_resident(gpu_ctx, vol[:8].shape)
    want = onepass.extract(np.searchsorted(union, vol[:8]).astype(np.uint32), max_label=union.size - 1)
    assert_sparse_equal(x, union.astype(np.int64), want)
    with pytest.raises(_capi.TissueScanError) as e:
        gpu_ctx.compact_labels(ids[:-1])
    assert e.value.code == _capi.TA_ERANGE
    with pytest.raises(_capi.TissueScanError):
        gpu_ctx.compact_labels(np.array([5, 5, 9], dtype=np.uint32))          # not unique


def test_relabel_and_a_new_volume_end_the_compacted_state(gpu_ctx):
    vol = scatter_ids(voronoi((8, 16, 128), 10, 9, np.uint32), 9, (1 << 29))
    gpu_ctx.set_volume(vol)
    gpu_ctx.compact_labels()
    assert gpu_ctx.is_compact()
    with pytest.raises(_capi.TissueScanError):                  # a compacted context relabels through one entry per RANK
        gpu_ctx.relabel(np.arange(4, dtype=np.uint32))
    ids = gpu_ctx.compact_ids()
    table = ids.copy()
    table[3] = ids[2]                                           # the fourth id is fused into the third
    gpu_ctx.relabel(table)
    assert not gpu_ctx.is_compact()
    back = np.zeros_like(vol)
    gpu_ctx.get_volume(back)
    assert np.array_equal(back, np.where(vol == ids[3], ids[2], vol))
    gpu_ctx.compact_labels()
    gpu_ctx.set_volume(vol)
    assert not gpu_ctx.is_compact()


def test_resident_volume_picks_the_sparse_path_by_itself():
    vol = scatter_ids(voronoi((16, 24, 160), 30, 10, np.uint32), 10, (1 << 31))
    ids, want = oracle_by_rank(vol)
    rv = ResidentVolume(vol)
    try:
        x = rv.extract()
        assert_sparse_equal(x, ids, want)
        t = rv.wall_table()                         # the wall voxels work on the ids themselves
        lo = (t.pairs >> np.uint64(32)).astype(np.int64)
        assert np.isin(lo, ids).all() and len(t) > 0
    finally:
        rv.close()


def test_the_analysis_class_end_to_end_on_ids_near_2_31():
    """SpatialImageAnalysis on a uint32 image whose ids are spread to 2^31: every answer equals that of the same image with
    small ids (ids translated), including the passes that rewrite the image through a per-label table (fuse / remove labels,
    property images): in a compacted context the table has one entry per rank."""
    from tissue_analysis_amd import DICT, SpatialImage, SpatialImageAnalysis3D
    dense = voronoi((20, 24, 96), 25, 11, np.uint32)
    rng = np.random.default_rng(11)
    old = np.unique(dense)
    lut = np.arange(int(old.max()) + 1, dtype=np.uint64)
    moved = old[old > 1]
    lut[moved] = np.sort(rng.choice(np.arange(1 << 20, (1 << 31) + 500, 977, dtype=np.uint64), size=moved.size, replace=False))
    vol = lut[dense].astype(np.uint32)                                  # (order-preserving here: min(labels) translates)
    t = dict((int(k), int(lut[k])) for k in old)
    a = SpatialImageAnalysis3D(SpatialImage(vol.copy()), ignoredlabels=0, return_type=DICT, background=1)
    b = SpatialImageAnalysis3D(SpatialImage(dense.copy()), ignoredlabels=0, return_type=DICT, background=1)
    assert a.extraction.sparse and not b.extraction.sparse
    assert a.labels() == [t[l] for l in b.labels()]
    va, vb = a.volume(), b.volume()
    assert all(va[t[l]] == vb[l] for l in b.labels())
    na, nb = a.neighbors(), b.neighbors()
    assert all(na[t[l]] == [t[n] for n in nb[l]] for l in b.labels())
    wa, wb = a.wall_areas(), b.wall_areas()
    assert len(wa) == len(wb) and all(wa[(t[i], t[j])] == w for (i, j), w in wb.items())
    assert a.boundingbox(t[b.labels()[3]]) == b.boundingbox(b.labels()[3])
    assert a.cell_first_layer() == [t[l] for l in b.cell_first_layer()]
    # property image: one value per label
    prop = dict((l, (7 * l) % 251 + 2) for l in b.labels())
    pa = a.property_image(dict((t[l], v) for l, v in prop.items()), dtype=np.uint16)
    pb = b.property_image(prop, dtype=np.uint16)
    assert np.array_equal(np.asarray(pa), np.asarray(pb))
    # fuse three cells, then remove two others: the images stay translations of each other, and so do the new sweeps
    trio = b.labels()[4:7]
    a.fuse_labels_in_image([t[l] for l in trio], verbose=False)
    b.fuse_labels_in_image(list(trio), verbose=False)
    gone = b.labels()[8:10]
    a.remove_labels_from_image([t[l] for l in gone], verbose=False)
    b.remove_labels_from_image(list(gone), verbose=False)
    assert np.array_equal(np.asarray(a.image), lut[np.asarray(b.image)].astype(np.uint32))
    assert a.extraction.sparse and a.labels() == [t[l] for l in b.labels()]
    va, vb = a.volume(), b.volume()
    assert all(va[t[l]] == vb[l] for l in b.labels())
    # the wall voxels and the voxel layers never see ranks
    la, lb = a.voxel_first_layer(), b.voxel_first_layer()
    assert np.array_equal(np.asarray(la), lut[np.asarray(lb)].astype(np.uint32))


def test_compaction_is_a_snapshot_and_rerank_refreshes_it():
    """a compacted context sweeps a rank copy written once.  An adopted device buffer rewritten in place is
    seen again after ta_volume_rerank (SlabJob.refresh); an id outside the list raises at the next getter; ta_volume_uncompact
    (extract_resident(sparse=False)) leaves the compacted state."""
    import torch
    from tissue_analysis_amd.extraction import extract_resident
    a = voronoi((12, 16, 512), 20, 91, np.uint32).astype(np.int64)
    ids = np.unique(a)
    big = (ids * 1000003 + 7) % (2 ** 31)
    order = np.argsort(big)
    table = np.sort(big).astype(np.uint32)
    lut = np.zeros(int(a.max()) + 1, dtype=np.uint32); lut[ids] = big.astype(np.uint32)
    v1 = lut[a]                                               # sparse ids
    t = torch.from_numpy(v1.view(np.int32)).cuda()
    ctx = _capi.Context(0)
    ctx.set_volume_device(t.data_ptr(), 4, t.shape, keep=t)
    ctx.compact_labels(table)
    ctx.extract(_capi.F_ALL, table.size - 1)
    c1 = ctx.labels()[0]
    # rewrite the buffer in place: swap two cells' ids everywhere
    x, y = int(table[1]), int(table[2])
    v2 = v1.copy(); v2[v1 == x] = y; v2[v1 == y] = x
    t.copy_(torch.from_numpy(v2.view(np.int32)).cuda())
    ctx.extract(_capi.F_ALL, table.size - 1)
    assert np.array_equal(ctx.labels()[0], c1)                # stale by construction: the snapshot was swept
    ctx.rerank()
    ctx.extract(_capi.F_ALL, table.size - 1)
    c2 = ctx.labels()[0]
    assert c2[1] == c1[2] and c2[2] == c1[1] and np.array_equal(np.delete(c2, [1, 2]), np.delete(c1, [1, 2]))
    # an id that is not in the list: the next getter raises
    v3 = v2.copy(); v3[0, 0, 0] = int(table[-1]) + 5
    t.copy_(torch.from_numpy(v3.view(np.int32)).cuda())
    ctx.rerank()
    ctx.extract(_capi.F_ALL, table.size - 1)
    with pytest.raises(_capi.TissueScanError):
        ctx.labels()
    # leaving the compacted state
    t.copy_(torch.from_numpy(a.astype(np.uint32).view(np.int32)).cuda())     # small ids again
    assert ctx.is_compact()
    x_dense = extract_resident(ctx, a.shape, _capi.F_ALL, sparse=False)
    assert not ctx.is_compact() and x_dense.ids is None and int(x_dense.count.sum()) == a.size
    ctx.close()


def test_plane_events_length_with_a_size_one_axis():
    """the slowest MEMORY axis is not argmax(strides) when an axis has size 1 (2-D images as (X, Y, 1))."""
    a = voronoi((40, 48, 1), 12, 92, np.uint16, ellipsoid=False)
    ctx = _capi.Context(0)
    ctx.set_volume(a)
    ev = ctx.plane_events()
    assert ev.size == ctx.owned_planes() and int(ev.sum()) == int((a[:, 1:, 0] != a[:, :-1, 0]).sum())
    ctx.close()


def test_the_census_in_one_pass_and_where_it_gives_up(gpu_ctx):
    """The census of a volume whose maximum is not known reads the voxels ONCE (the workgroups' label sets go through a list that
    sizes the table); a volume of noise overflows the list and takes the two passes -- the same answer either way."""
    rng = np.random.default_rng(11)
    # tissue-like, whole 16-byte rows, the id no set slot can hold (0xFFFFFFFF) among them
    vol = scatter_ids(voronoi((40, 64, 512), 400, 8, np.uint32), 8, (1 << 32) - 1)
    vol[3, 5, 100:140] = 0xFFFFFFFF
    gpu_ctx.set_volume(vol)
    top, present = gpu_ctx.label_census()
    assert top == 0xFFFFFFFF and np.array_equal(present, np.unique(vol))
    # noise: two million voxels, nearly as many ids -- more than the list holds
    noise = rng.integers(0, 1 << 32, size=(128, 128, 128), dtype=np.uint64).astype(np.uint32)
    gpu_ctx.set_volume(noise)
    top, present = gpu_ctx.label_census()
    assert top == int(noise.max()) and np.array_equal(present, np.unique(noise))
    # uint16 noise (every id of the type, many evictions from the sets, but few enough for the list)
    n16 = rng.integers(0, 1 << 16, size=(16, 64, 512), dtype=np.uint32).astype(np.uint16)
    gpu_ctx.set_volume(n16)
    top, present = gpu_ctx.label_census()
    assert top == int(n16.max()) and np.array_equal(present, np.unique(n16))
