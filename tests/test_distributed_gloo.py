"""The multi-GPU exchange step on CPU: 2 gloo processes, each holding the accumulators of its
Z-slab (+ one halo plane) in the device layout; after the collectives every rank must hold exactly
the accumulators of the unsharded volume (bit-exact)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, dims, n_cells, seed, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import onepass
        from tissue_analysis_amd import distributed as tad, synth
        vol = synth.voronoi_labels(dims, n_cells, seed, np.uint16)
        whole = onepass.extract(vol)
        L = whole["max_label"]
        lo, hi = tad.slab_range(dims[0], world, rank)
        halo = 1 if lo > 0 else 0
        part = onepass.extract(vol[lo - halo:hi], max_label=L, origin=(lo - halo, 0, 0), own_first_plane=not halo)
        sums, boxes = tad.to_device_layout(part)
        sums_t, boxes_t = torch.from_numpy(sums), torch.from_numpy(boxes)
        tad.allreduce_accumulators(sums_t, boxes_t)
        merged = tad.from_device_layout(sums_t.numpy(), boxes_t.numpy())
        keys = (part["pair_lo"].astype(np.int64) << 32) | part["pair_hi"].astype(np.int64)
        kall, fall, m = tad.allgather_pairs(torch.from_numpy(keys), torch.from_numpy(part["pair_faces"].astype(np.int64)))
        kall, fall = kall.numpy(), fall.numpy()
        keep = kall != tad.EMPTY_KEY
        plo, phi, pf = onepass.compact_pairs(kall[keep] >> 32, kall[keep] & 0xFFFFFFFF, fall[keep])
        ok = all(np.array_equal(merged[k], whole[k]) for k in ("count", "bbox", "sum1", "sum2"))
        ok = ok and np.array_equal(plo, whole["pair_lo"]) and np.array_equal(phi, whole["pair_hi"]) \
            and np.array_equal(pf, whole["pair_faces"])
        out_q.put((rank, bool(ok), int(m)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,n_cells", [((21, 16, 24), 14), ((8, 20, 20), 10)])
def test_two_rank_reduce_equals_unsharded(dims, n_cells):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, dims, n_cells, 31, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    assert all(r[1] for r in results), results


def _balanced_worker(rank, world, port, dims, n_cells, seed, out_q):
    """Cost-balanced cuts + the reduce-scatter of the sums (gloo: an all-reduce of a copy, then the rank's rows): every rank
    ends with the GLOBAL rows of its share of the labels, the gathered table equals the unsharded one, the boxes are global."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import onepass
        from tissue_analysis_amd import distributed as tad, synth
        vol = synth.voronoi_labels(dims, n_cells, seed, np.uint16)
        whole = onepass.extract(vol)
        L = whole["max_label"]
        # the plane weights: label changes along the fast axis, counted by every rank on an equal first split and summed
        lo0, hi0 = tad.slab_range(dims[0], world, rank)
        ev = torch.zeros(dims[0], dtype=torch.int64)
        ev[lo0:hi0] = torch.from_numpy((vol[lo0:hi0, :, 1:] != vol[lo0:hi0, :, :-1]).sum(axis=(1, 2)).astype(np.int64))
        dist.all_reduce(ev)
        cuts = tad.balanced_cuts(tad.plane_costs(ev.numpy(), dims[1] * dims[2]), world)
        lo, hi = tad.slab_range(dims[0], world, rank, cuts)
        halo = 1 if lo > 0 else 0
        part = onepass.extract(vol[lo - halo:hi], max_label=L, origin=(lo - halo, 0, 0), own_first_plane=not halo)
        sums, boxes = tad.to_device_layout(part)
        S = tad.sums_shard_rows(L + 1, world)
        padded = np.zeros((world * S, 10), dtype=np.int64)
        padded[:L + 1] = sums
        sums_t, boxes_t, shard = torch.from_numpy(padded), torch.from_numpy(boxes), torch.zeros((S, 10), dtype=torch.int64)
        tad.reduce_scatter_sums(sums_t, shard)
        dist.all_reduce(boxes_t, op=dist.ReduceOp.MIN)
        gsums, _ = tad.to_device_layout(whole)
        want = np.zeros((world * S, 10), dtype=np.int64)
        want[:L + 1] = gsums
        ok = np.array_equal(shard.numpy(), want[rank * S:(rank + 1) * S])              # this rank's labels: global rows
        ok = ok and np.array_equal(sums_t.numpy()[:L + 1], sums)                          # the slab's own rows are untouched
        merged = tad.from_device_layout(tad.allgather_sums(shard, L + 1).numpy(), boxes_t.numpy())
        ok = ok and all(np.array_equal(merged[k], whole[k]) for k in ("count", "bbox", "sum1", "sum2"))
        out_q.put((rank, bool(ok), [int(c) for c in cuts]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims,n_cells", [(2, (30, 16, 24), 20), (3, (41, 12, 20), 16)])
def test_balanced_cuts_and_reduce_scatter_give_the_unsharded_rows(world, dims, n_cells):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_balanced_worker, args=(r, world, port, dims, n_cells, 33, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in results), results
    cuts = results[0][2]
    assert all(r[2] == cuts for r in results) and cuts[0] == 0 and cuts[-1] == dims[0] and all(b > a for a, b in zip(cuts, cuts[1:]))


def test_balanced_cuts_equalise_the_cost():
    from tissue_analysis_amd import distributed as tad
    n0 = 2048
    a = np.arange(n0)
    events = np.clip(1.0 - ((a - (n0 - 1) / 2.0) / (0.45 * n0)) ** 2, 0.0, None) * 1.2e5      # a tissue inside an ellipsoid
    costs = tad.plane_costs(events, 2048 * 2048)
    for world in (2, 4, 8):
        cuts = tad.balanced_cuts(costs, world)
        slabs = np.array([costs[cuts[r]:cuts[r + 1]].sum() for r in range(world)])
        assert slabs.max() / slabs.mean() < 1.01, (world, cuts)
        equal = np.array([costs[slice(*tad.slab_range(n0, world, r))].sum() for r in range(world)])
        assert world == 2 or equal.max() / equal.mean() > 1.05                                # what equal plane counts leave
    assert tad.balanced_cuts(np.ones(5), 5) == [0, 1, 2, 3, 4, 5]
    assert tad.balanced_cuts(np.array([100.0, 1, 1, 1, 1, 1]), 3) == [0, 1, 2, 6] or tad.balanced_cuts(np.array([100.0, 1, 1, 1, 1, 1]), 3)[1] == 1
    with pytest.raises(ValueError):
        tad.balanced_cuts(np.ones(3), 4)


def test_slab_ranges_partition_the_axis():
    from tissue_analysis_amd import distributed as tad
    for n0 in (1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            spans = [tad.slab_range(n0, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n0
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_device_layout_roundtrip():
    from oracle import onepass
    from tissue_analysis_amd import distributed as tad, synth
    vol = synth.voronoi_labels((10, 12, 14), 6, 3, np.uint16)
    r = onepass.extract(vol, max_label=int(vol.max()) + 3)      # trailing absent labels
    back = tad.from_device_layout(*tad.to_device_layout(r))
    for k in ("count", "bbox", "sum1", "sum2"):
        assert np.array_equal(back[k], r[k]), k


def _halo_worker(rank, world, port, dims, dtype_name, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tissue_analysis_amd import distributed as tad, synth
        vol = synth.voronoi_labels(dims, 12, 5, np.dtype(dtype_name))
        lo, hi = tad.slab_range(dims[0], world, rank)
        signed = {"uint16": np.int16, "uint32": np.int32}[dtype_name]
        owned = torch.from_numpy(vol[lo:hi].view(signed).copy())       # this rank holds ONLY its own planes
        buf, halo = tad.attach_low_halo(owned)
        want = vol[lo - (1 if rank else 0):hi].view(signed)
        ok = bool(halo) == (rank > 0) and tuple(buf.shape) == want.shape and np.array_equal(buf.numpy(), want)
        # a second hand-off into the same buffer (a time series re-using its slab buffers) changes nothing
        tad.exchange_low_halo(buf)
        ok = ok and np.array_equal(buf.numpy(), want)
        out_q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims,dtype_name", [(2, (9, 6, 10), "uint16"), (3, (10, 5, 7), "uint32")])
def test_halo_plane_arrives_from_the_neighbouring_rank(world, dims, dtype_name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, dims, dtype_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world)) and all(r[1] for r in results), results


@pytest.mark.parametrize("world,dims,n_cells,scatter", [(2, (21, 16, 24), 14, 0), (3, (24, 12, 20), 30, 0), (4, (32, 10, 12), 40, 25)])
def test_only_pairs_a_slab_face_can_split_need_to_travel(world, dims, n_cells, scatter):
    """The rule of ta_adjacency_pack_shared, on the CPU: with the GLOBAL boxes, pairs that hold a slab-exclusive label are
    final on their rank (and appear on no other), the rest merged over ranks completes the unsharded list -- also when
    labels come in several pieces (scattered voxels of existing labels)."""
    from oracle import onepass
    from tissue_analysis_amd import distributed as tad, synth
    vol = synth.voronoi_labels(dims, n_cells, 77, np.uint16)
    rng = np.random.default_rng(5)
    for _ in range(scatter):                                     # disconnected pieces of labels, anywhere
        vol[tuple(rng.integers(0, d) for d in dims)] = rng.integers(1, int(vol.max()) + 1)
    whole = onepass.extract(vol)
    L = whole["max_label"]
    _, gboxes = tad.to_device_layout(whole)
    private_all, travelling = {}, {}
    for rank in range(world):
        lo, hi = tad.slab_range(dims[0], world, rank)
        halo = 1 if lo > 0 else 0
        part = onepass.extract(vol[lo - halo:hi], max_label=L, origin=(lo - halo, 0, 0), own_first_plane=not halo)
        excl = tad.slab_exclusive(gboxes, lo, hi)
        for a, b, f in zip(part["pair_lo"].tolist(), part["pair_hi"].tolist(), part["pair_faces"]):
            if excl[a] or excl[b]:
                assert (a, b) not in private_all                 # private lists are disjoint
                private_all[(a, b)] = f.copy()
            else:
                travelling[(a, b)] = travelling.get((a, b), 0) + f
    assert not set(private_all) & set(travelling)
    merged = dict(private_all)
    merged.update(travelling)
    want = dict(((a, b), f) for a, b, f in zip(whole["pair_lo"].tolist(), whole["pair_hi"].tolist(), whole["pair_faces"]))
    assert sorted(merged) == sorted(want)
    assert all(np.array_equal(merged[k], want[k]) for k in want)
    assert len(travelling) < len(want)                           # (something did stay at home)


def _union_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tissue_analysis_amd import distributed as tad
        mine = [np.array([0, 7, 1 << 31, (1 << 32) - 1], dtype=np.uint32), np.array([7, 9], dtype=np.uint32),
                np.zeros(0, dtype=np.uint32)][rank]
        ids = tad.union_of_ids(mine, dist.group.WORLD)
        q.put((rank, ids.dtype == np.uint32 and ids.tolist() == [0, 7, 9, 1 << 31, (1 << 32) - 1]))
    finally:
        dist.destroy_process_group()


def test_union_of_ids_over_three_ranks():
    """The id table the slabs of a volume with sparse ids compact with: the same ascending union on every rank, whatever each
    rank's own list holds (one of them holds nothing)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 300)
    procs = [ctx.Process(target=_union_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results
