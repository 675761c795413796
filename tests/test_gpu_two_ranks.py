"""The multi-rank job on real device memory: two processes share cuda:0 over a gloo group (RCCL cannot
put two ranks on one device; the collectives are the same torch.distributed calls).  Each rank
generates its Z-slab (+ low halo plane) on the device, sweeps it through the C ABI into torch-owned
accumulators, all-reduces them and merges the gathered adjacency; both ranks must end with exactly
the unsharded oracle result."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, dims, n_cells, seed, dtype_name, q, capacity=None, pair_slots=0, depth=1, features=31, balanced=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import onepass_c
        from tissue_analysis_amd import _capi, device as dev, distributed as tad, synth
        dtype = np.dtype(dtype_name)
        ctx = dev.torch_context(0)
        lo, hi = tad.slab_range(dims[0], world, rank)
        halo = 1 if lo > 0 else 0
        vol, max_label = dev.synth_slab(ctx, dims, dtype, n_cells, seed, lo - halo, hi, device=0)
        reduce = "all"
        events_ok = True
        if balanced:
            # cost-balanced slabs + reduce-scattered sums (what bench.py --gpus N runs): the ranks count the label changes
            # of their planes on the device, agree on the cuts and generate their slabs again
            torch.cuda.synchronize()
            ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, a0_origin=lo, has_low_halo=bool(halo), keep=vol)
            mine = torch.zeros(dims[0], dtype=torch.int64)
            ev = ctx.plane_events()
            host = vol.cpu().numpy().view(dtype)[halo:]
            events_ok = np.array_equal(ev, (host[:, :, 1:] != host[:, :, :-1]).sum(axis=(1, 2)).astype(np.uint64))
            mine[lo:hi] = torch.from_numpy(ev.astype(np.int64))
            dist.all_reduce(mine)
            cuts = tad.balanced_cuts(tad.plane_costs(mine.numpy(), dims[1] * dims[2]), world)
            lo, hi = tad.slab_range(dims[0], world, rank, cuts)
            halo = 1 if lo > 0 else 0
            vol, max_label = dev.synth_slab(ctx, dims, dtype, n_cells, seed, lo - halo, hi, device=0)
            reduce = "scatter"
        if pair_slots and rank == 1:    # one rank starts with a table far too small: sizes must be agreed on
            ctx.set_option(_capi.OPT_PAIR_SLOTS, pair_slots)
        if depth > 1:                   # two steps in flight on two streams / contexts
            job = tad.PipelinedSlabJob(vol, dtype.itemsize, a_origin=lo, has_low_halo=bool(halo), max_label=max_label,
                                       features=features, group=dist.group.WORLD, device=0, depth=depth, reduce=reduce)
            for _ in range(3):
                job.step()
        else:
            job = tad.SlabJob(ctx, vol, dtype.itemsize, a_origin=lo, has_low_halo=bool(halo), max_label=max_label,
                              features=features, group=dist.group.WORLD, device=0, exchange_capacity=capacity, reduce=reduce)
            ctx.extract(features, max_label)          # (a plain sweep first: the exchange resets the diagnostic counters)
            spills = ctx.debug_counters()["label_spills"]
            job.step()
            job.step()                  # a second step must give the same answer (tables self-clean)
        got = job.result_arrays()
        if depth > 1:
            spills = 0
        whole = synth.voronoi_labels(dims, n_cells, seed, dtype)
        want = onepass_c.extract(whole, max_label=max_label)
        if not features & _capi.F_MOMENT2:          # not asked for: answered as zero, whatever the spill paths left in the rows
            want = dict(want, sum2=np.zeros_like(want["sum2"]))
        ok = all(np.array_equal(got[k], want[k]) for k in
                 ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces"))
        bad = [k for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces")
               if not (got[k].shape == want[k].shape and np.array_equal(got[k], want[k]))]
        if balanced:
            ok = ok and events_ok and job.result_counts().shape[0] == max_label + 1
            bad.append("plane_events=%s" % events_ok)
        if capacity:                    # a block that small must have forced exactly one collective redo
            ok = ok and job.redo_count == 1
            bad.append("redo_count=%d" % job.redo_count)
        if n_cells > 10000:             # cells of a few voxels: the workgroup tables must have spilled to the global rows
            ok = ok and spills > 0
            bad.append("label_spills=%d" % spills)
        q.put((rank, bool(ok), bad))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,n_cells,dtype_name,capacity,pair_slots,depth,features,balanced", [
    ((37, 40, 264), 50, "uint32", None, 0, 1, 31, False),
    ((20, 24, 520), 30, "uint16", None, 0, 1, 31, False),
    ((37, 40, 264), 50, "uint32", 8, 0, 1, 31, False),        # exchange blocks too small: verdict -> re-size -> redo
    ((37, 40, 264), 50, "uint32", None, 6, 1, 31, False),     # rank 1 starts with a 64-slot table: grown and agreed on
    ((37, 40, 264), 50, "uint32", None, 0, 2, 31, False),     # PipelinedSlabJob: steps alternate between two streams
    ((37, 40, 264), 20000, "uint32", None, 0, 1, 0x17, False),   # tiny cells (table spills) WITHOUT second moments: sum2 must read as zero
    ((37, 40, 264), 50, "uint32", None, 0, 1, 31, True),      # slabs cut by cost (ta_volume_plane_events), sums reduce-scattered
    ((21, 24, 520), 30, "uint16", None, 0, 2, 31, True),      # ... two steps in flight
    ((23, 24, 1024), 60, "uint32", None, 0, 1, 31, False),    # rows of whole 512-column tiles: the slabs' sweeps take turns between the two tile shapes
    ((23, 24, 1024), 60, "uint32", None, 0, 2, 31, True),     # ... cost-balanced, reduce-scattered, two steps in flight
])
def test_two_ranks_on_one_gpu_match_unsharded(dims, n_cells, dtype_name, capacity, pair_slots, depth, features, balanced):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, dims, n_cells, 61, dtype_name, q, capacity, pair_slots, depth, features, balanced))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in results), results


def _handoff_worker(rank, world, port, dims, n_cells, seed, dtype_name, q):
    """Each rank generates ONLY its own planes in device memory; plane lo-1 comes out of the neighbouring rank's
    memory (attach_low_halo) -- no generator, no upload delivers it."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import onepass_c
        from tissue_analysis_amd import _capi, device as dev, distributed as tad, synth
        dtype = np.dtype(dtype_name)
        ctx = dev.torch_context(0)
        lo, hi = tad.slab_range(dims[0], world, rank)
        owned, max_label = dev.synth_slab(ctx, dims, dtype, n_cells, seed, lo, hi, device=0)
        torch.cuda.synchronize()
        vol, halo = tad.attach_low_halo(owned, dist.group.WORLD)
        whole = synth.voronoi_labels(dims, n_cells, seed, dtype)
        signed = {"uint16": np.int16, "uint32": np.int32}[dtype_name]
        bad = []
        if bool(halo) != (rank > 0) or not np.array_equal(vol.cpu().numpy(), whole[lo - int(halo):hi].view(signed)):
            bad.append("slab buffer")
        job = tad.SlabJob(ctx, vol, dtype.itemsize, a_origin=lo, has_low_halo=halo, max_label=max_label,
                          features=_capi.F_ALL, group=dist.group.WORLD, device=0)
        job.step()
        got = job.result_arrays()
        want = onepass_c.extract(whole, max_label=max_label)
        bad += [k for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces")
                if not (got[k].shape == want[k].shape and np.array_equal(got[k], want[k]))]
        q.put((rank, not bad, bad))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,n_cells,dtype_name", [((37, 40, 264), 50, "uint32"), ((21, 24, 520), 30, "uint16")])
def test_halo_handed_over_between_device_resident_slabs(dims, n_cells, dtype_name):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 28800 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_handoff_worker, args=(r, world, port, dims, n_cells, 67, dtype_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in results), results


def _nccl_worker(rank, world, port, dims, n_cells, seed, q):
    """World size 1 over RCCL: the same SlabJob code path as a multi-GPU run (all-reduce, packed all-gather, merge of the
    gathered blocks), with torch.distributed's nccl backend on the one GPU of this box."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from oracle import onepass_c
        from tissue_analysis_amd import _capi, device as dev, distributed as tad, synth
        dtype = np.dtype("uint32")
        ctx = dev.torch_context(0)
        vol, max_label = dev.synth_slab(ctx, dims, dtype, n_cells, seed, 0, dims[0], device=0)
        vol, halo = tad.attach_low_halo(vol, dist.group.WORLD)          # world 1: nothing to fetch
        assert halo is False
        job = tad.SlabJob(ctx, vol, dtype.itemsize, a_origin=0, has_low_halo=False, max_label=max_label,
                          features=_capi.F_ALL, group=dist.group.WORLD, device=0)
        for _ in range(3):
            job.step()
        got = job.result_arrays()
        want = onepass_c.extract(synth.voronoi_labels(dims, n_cells, seed, dtype), max_label=max_label)
        bad = [k for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces")
               if not (got[k].shape == want[k].shape and np.array_equal(got[k], want[k]))]
        pipe = tad.PipelinedSlabJob(vol, dtype.itemsize, a_origin=0, has_low_halo=False, max_label=max_label,
                                    features=_capi.F_ALL, group=dist.group.WORLD, device=0, depth=2)
        for _ in range(4):
            pipe.step()
        got2 = pipe.result_arrays()
        bad += ["pipelined " + k for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces")
                if not np.array_equal(got2[k], want[k])]
        # the exchange step's three collectives at C5's message sizes (100 001 label rows; a 2^17-pair block per rank), world
        # size 1: the launch + local cost RCCL adds per step with no wire -- the floor of SURVEY.md §8(e)'s budget
        rows = 100001
        sums = torch.zeros((rows, 10), dtype=torch.int64, device="cuda:0")
        boxes = torch.zeros((rows, 6), dtype=torch.int32, device="cuda:0")
        words = _capi.exchange_words(1 << 17)
        send = torch.zeros((words,), dtype=torch.int64, device="cuda:0")
        recv = torch.zeros((words,), dtype=torch.int64, device="cuda:0")

        def one():
            tad.allreduce_accumulators(sums, boxes, dist.group.WORLD)
            dist.all_gather_into_tensor(recv, send, group=dist.group.WORLD)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            one()
        b.record(); b.synchronize()
        note = "RCCL at world size 1, C5 message sizes (rows %.1f MB, block %.1f MB): %.3f ms per step for 2 all-reduces + 1 all-gather" % (
            (sums.numel() * 8 + boxes.numel() * 4) / 1e6, words * 8 / 1e6, a.elapsed_time(b) / 20)
        q.put((rank, not bad, bad, note))
        pipe.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_world_size_one_over_rccl(capsys):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    p = ctx.Process(target=_nccl_worker, args=(0, 1, port, (96, 128, 512), 400, 71, q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0 and res[1], res
    with capsys.disabled():
        print("\n[" + res[3] + "]")


def _range_worker(rank, world, port, q):
    """Only rank 1's slab holds a label above max_label: BOTH ranks must leave with TA_ERANGE (no rank may be left
    waiting in a collective) -- on the very first step, where the sizes are agreed on, and on a later one."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tissue_analysis_amd import _capi, device as dev, distributed as tad
        dims, dtype = (24, 32, 256), np.dtype("uint32")
        ctx = dev.torch_context(0)
        lo, hi = tad.slab_range(dims[0], world, rank)
        halo = 1 if lo > 0 else 0
        vol, max_label = dev.synth_slab(ctx, dims, dtype, 30, 72, lo - halo, hi, device=0)
        outcomes = []
        for first in (True, False):
            job = tad.SlabJob(ctx, vol, dtype.itemsize, a_origin=lo, has_low_halo=bool(halo), max_label=max_label,
                              features=_capi.F_ALL, group=dist.group.WORLD, device=0)
            if not first:
                job.step()
                job.finish()
            if rank == 1:
                vol[3, 5, 7] = max_label + 5
            try:
                job.step()
                job.finish()
                outcomes.append("no error")
            except _capi.TissueScanError as e:
                outcomes.append("ERANGE" if e.code == _capi.TA_ERANGE else "code %d" % e.code)
            if rank == 1:
                vol[3, 5, 7] = 1
        q.put((rank, outcomes == ["ERANGE", "ERANGE"], outcomes))
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_a_range_error_on_one_rank_is_raised_on_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 300)
    procs = [ctx.Process(target=_range_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in results), results


def _sparse_worker(rank, world, port, dims, n_cells, seed, depth, reduce, q):
    """SPARSE ids over two slabs: every rank takes the census of its own slab on the device, the ranks agree on the union of
    the ids and compact with it, so that an id has the same row on both; the exchange then runs in rank space."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import onepass_c
        from tissue_analysis_amd import device as dev, distributed as tad, synth
        whole = synth.voronoi_labels(dims, n_cells, seed, np.uint32)
        rng = np.random.default_rng(seed)
        old = np.unique(whole)
        lut = np.zeros(int(old.max()) + 1, dtype=np.uint64)
        lut[old] = rng.choice(np.arange(5, (1 << 32) - 1, 104729, dtype=np.uint64), size=old.size, replace=False)   # (not in order)
        whole = lut[whole].astype(np.uint32)
        ctx = dev.torch_context(0)
        lo, hi = tad.slab_range(dims[0], world, rank)
        halo = 1 if lo > 0 else 0
        vol = torch.from_numpy(whole[lo - halo:hi].copy()).to("cuda:0")
        ctx.set_volume_device(vol.data_ptr(), 4, vol.shape, a0_origin=lo, has_low_halo=bool(halo), keep=vol)
        top, mine = ctx.label_census()
        ids = tad.union_of_ids(mine, dist.group.WORLD)
        census_ok = np.array_equal(mine, np.unique(whole[lo - halo:hi])) and np.array_equal(ids, np.unique(whole))
        if depth > 1:
            job = tad.PipelinedSlabJob(vol, 4, a_origin=lo, has_low_halo=bool(halo), max_label=0, features=31,
                                       group=dist.group.WORLD, device=0, depth=depth, reduce=reduce, ids=ids)
            for _ in range(3):
                job.step()
        else:
            job = tad.SlabJob(ctx, vol, 4, a_origin=lo, has_low_halo=bool(halo), max_label=0, features=31,
                              group=dist.group.WORLD, device=0, reduce=reduce, ids=ids)
            job.step()
            job.step()
        got = job.result_arrays()
        uniq, inv = np.unique(whole, return_inverse=True)
        want = onepass_c.extract(inv.reshape(whole.shape).astype(np.uint32), max_label=uniq.size - 1)
        want["pair_lo"] = uniq[np.asarray(want["pair_lo"], dtype=np.int64)]
        want["pair_hi"] = uniq[np.asarray(want["pair_hi"], dtype=np.int64)]
        bad = [k for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces")
               if not (np.asarray(got[k]).shape == np.asarray(want[k]).shape and np.array_equal(got[k], want[k]))]
        if not np.array_equal(got["ids"], uniq.astype(np.int64)):
            bad.append("ids")
        if not census_ok:
            bad.append("census")
        q.put((rank, not bad, bad))
        if depth > 1:
            job.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth,reduce", [(1, "all"), (2, "scatter")])
def test_two_ranks_with_sparse_ids_agree_on_the_rows(depth, reduce):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 300)
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, (37, 40, 264), 50, 67, depth, reduce, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in results), results
