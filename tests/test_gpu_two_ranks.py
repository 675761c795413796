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


def _worker(rank, world, port, dims, n_cells, seed, dtype_name, q, capacity=None, pair_slots=0, depth=1):
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
        if pair_slots and rank == 1:    # one rank starts with a table far too small: sizes must be agreed on
            ctx.set_option(_capi.OPT_PAIR_SLOTS, pair_slots)
        if depth > 1:                   # two steps in flight on two streams / contexts
            job = tad.PipelinedSlabJob(vol, dtype.itemsize, a_origin=lo, has_low_halo=bool(halo), max_label=max_label,
                                       features=_capi.F_ALL, group=dist.group.WORLD, device=0, depth=depth)
            for _ in range(3):
                job.step()
        else:
            job = tad.SlabJob(ctx, vol, dtype.itemsize, a_origin=lo, has_low_halo=bool(halo), max_label=max_label,
                              features=_capi.F_ALL, group=dist.group.WORLD, device=0, exchange_capacity=capacity)
            job.step()
            job.step()                  # a second step must give the same answer (tables self-clean)
        got = job.result_arrays()
        whole = synth.voronoi_labels(dims, n_cells, seed, dtype)
        want = onepass_c.extract(whole, max_label=max_label)
        ok = all(np.array_equal(got[k], want[k]) for k in
                 ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces"))
        bad = [k for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces")
               if not (got[k].shape == want[k].shape and np.array_equal(got[k], want[k]))]
        if capacity:                    # a block that small must have forced exactly one collective redo
            ok = ok and job.redo_count == 1
            bad.append("redo_count=%d" % job.redo_count)
        q.put((rank, bool(ok), bad))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,n_cells,dtype_name,capacity,pair_slots,depth", [
    ((37, 40, 264), 50, "uint32", None, 0, 1),
    ((20, 24, 520), 30, "uint16", None, 0, 1),
    ((37, 40, 264), 50, "uint32", 8, 0, 1),        # exchange blocks too small: verdict -> re-size -> redo
    ((37, 40, 264), 50, "uint32", None, 6, 1),     # rank 1 starts with a 64-slot table: grown and agreed on
    ((37, 40, 264), 50, "uint32", None, 0, 2),     # PipelinedSlabJob: steps alternate between two streams
])
def test_two_ranks_on_one_gpu_match_unsharded(dims, n_cells, dtype_name, capacity, pair_slots, depth):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, dims, n_cells, 61, dtype_name, q, capacity, pair_slots, depth))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in results), results
