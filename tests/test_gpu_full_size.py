"""BASELINE.json's configurations at FULL size on one GPU (C4 and C5 were exercised at reduced size only).

C3 / C4: 1024^3 uint32, 50k seeds, generated on the device, every integer array compared with the C oracle on the
whole volume (the oracle runs at ~60 Mvoxel/s: ~20 s each way).
C2: 512^3 uint16, 5 000 seeds at its stated size: its own feature subset (volume + bounding box + barycentre sums) and the
full feature set.  The tissue-filled C4 (the same generator WITHOUT the ellipsoid mask: 50 653 labels present, the
densest workload bench.py quotes): full feature set.
C5: 2048^3 uint32, 100k seeds (34 GB) on one GPU: size-independent identities at full scale, three 32-plane windows
(first / middle / last) bit-exact against the C oracle, eight virtual Z-slabs with halo planes merged == unsharded, and
the WHOLE volume against the C oracle run on 16 forked host processes over Z-slabs (skipped, with a printed reason, on a
host without ~46 GB of free memory).
"""
import numpy as np
import pytest

from oracle import onepass, onepass_c
from tissue_analysis_amd import _capi, device as dev, synth

from helpers import assert_same_accumulators, onepass_c_parallel

pytestmark = pytest.mark.gpu
KEYS = ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces")


def fetch(ctx, max_label, adjacency=True):
    count, bbox, s1, s2 = ctx.labels()
    if adjacency:
        lo, hi, f = ctx.adjacency()
    else:
        lo = hi = np.zeros(0, dtype=np.uint32)
        f = np.zeros((0, 3), dtype=np.uint64)
    return dict(max_label=max_label, count=count, bbox=bbox, sum1=s1, sum2=s2, pair_lo=lo, pair_hi=hi, pair_faces=f)


def test_c4_and_c3_full_size_against_the_c_oracle(capsys):
    import torch
    c = synth.CONFIGS["C4"]
    dims, dtype = c["dims"], np.dtype(c["dtype"])
    ctx = dev.torch_context(0)
    vol, L = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"])
    torch.cuda.synchronize()
    ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
    ctx.set_option(_capi.OPT_SWEEP_SHAPE, 1)                        # the wide tiles of the uint32 sweep ...
    ctx.extract(_capi.F_ALL, L)
    got4 = fetch(ctx, L)
    ctx.set_option(_capi.OPT_SWEEP_SHAPE, 0)                        # ... and the narrow ones
    ctx.extract(_capi.F_ALL, L)
    got4n = fetch(ctx, L)
    ctx.set_option(_capi.OPT_SWEEP_SHAPE, -2)                       # four timed sweeps take turns: the second sweep of a volume is a narrow one
    c3 = _capi.feature_mask(synth.CONFIGS["C3"]["features"])
    ctx.extract(c3, L)
    ctx.extract(c3, L)
    got3 = fetch(ctx, L)
    host = vol.cpu().numpy().view(dtype)
    ctx.close()
    del vol
    want = onepass_c.extract(host, max_label=L)
    assert_same_accumulators(got4, want, "C4 1024^3 full feature set, wide tiles")
    assert_same_accumulators(got4n, want, "C4 1024^3 full feature set, narrow tiles")
    for k in ("count", "bbox", "sum1", "pair_lo", "pair_hi", "pair_faces"):
        assert np.array_equal(got3[k], want[k]), "C3 " + k
    assert int((want["count"] > 0).sum()) > 20000 and want["pair_lo"].size > 100000
    # the caller's side on the headline workload (21.8k labels present): graph assembly from the arrays, host time only
    import time
    from tissue_analysis_amd import DICT, Extraction, SpatialImage, SpatialImageAnalysis3D, graph_from_image
    x = Extraction.from_arrays(dims, got4)
    props = ['boundingbox', 'volume', 'barycenter', 'L1', 'border', 'inertia_axis', 'wall_surface', 'epidermis_surface']
    best = 1e9
    for _ in range(3):
        sia = SpatialImageAnalysis3D(SpatialImage(np.zeros((2, 2, 2), dtype), voxelsize=synth.PARITY_VOXELSIZE), ignoredlabels=0,
                                     return_type=DICT, background=1, extraction=x)
        t0 = time.perf_counter()
        g = graph_from_image(sia, spatio_temporal_properties=list(props))
        best = min(best, time.perf_counter() - t0)
    with capsys.disabled():
        print("\n[graph_from_image on C4's arrays: %d vertices, %d edges] %.1f ms host time" % (g.nb_vertices(), g.nb_edges(), best * 1e3))
    assert best < 0.2


def test_c2_full_size_against_the_c_oracle():
    import torch
    c = synth.CONFIGS["C2"]
    dims, dtype = c["dims"], np.dtype(c["dtype"])
    ctx = dev.torch_context(0)
    vol, L = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"])
    torch.cuda.synchronize()
    ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
    ctx.extract(_capi.feature_mask(c["features"]), L)             # 0x07: what BASELINE.json asks of this configuration
    got_own = fetch(ctx, L, adjacency=False)
    ctx.extract(_capi.F_ALL, L)
    got_all = fetch(ctx, L)
    host = vol.cpu().numpy().view(dtype)
    ctx.close()
    del vol
    assert host.shape == (512, 512, 512) and host.dtype == np.uint16
    want = onepass_c.extract(host, max_label=L)
    for k in ("count", "bbox", "sum1"):
        assert np.array_equal(got_own[k], want[k]), "C2 (volume + bbox + barycentre) " + k
    assert not got_own["sum2"].any()                                # not asked for: answered as zero
    assert_same_accumulators(got_all, want, "C2 512^3 full feature set")
    assert 2000 < int((want["count"] > 0).sum()) <= 5001


def test_c2_wall_medians_end_to_end_on_the_device(capsys):
    """`graph_from_image(..., 'wall_median')` at C2's size (512^3 uint16, 15 M wall-voxel records): the medians of all ~16 k walls
    are computed on the device from the records it grouped by pair and E x 3 integers come back (rounds 2-3: 300 MB of
    records to pageable host memory, then Weiszfeld on the host).  Checked against the host arithmetic on a sample of the
    walls; the end-to-end times of both routes are printed."""
    import time
    import torch
    from tissue_analysis_amd import SpatialImageAnalysis3D, geometry, graph_from_image
    c = synth.CONFIGS["C2"]
    dims, dtype = c["dims"], np.dtype(c["dtype"])
    ctx = dev.torch_context(0)
    vol, L = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"])
    torch.cuda.synchronize()
    host = vol.cpu().numpy().view(dtype)
    ctx.close()
    del vol
    sia = SpatialImageAnalysis3D(host, ignoredlabels=0, background=1)
    t0 = time.perf_counter()
    g = graph_from_image(sia, spatio_temporal_properties=['wall_median'], background=1, ignore_cells_at_stack_margins=False)
    t_dev = time.perf_counter() - t0
    assert sia._walls is None and sia._wall_medians
    keys, sizes, med, _moving = sia._wall_medians
    kernels_ms = sia._resident().ms["wall_medians"]
    # a sample of the walls against the host arithmetic on the grouped records
    t0 = time.perf_counter()
    table = sia.wall_table()
    t_fetch = time.perf_counter() - t0
    pick = np.unique(np.linspace(0, keys.size - 1, 400).astype(np.int64))
    at = np.searchsorted(table.pairs, keys[pick])
    assert np.array_equal(table.pairs[at], keys[pick]) and np.array_equal(table.stop[at] - table.start[at], sizes[pick].astype(np.int64))
    rows, seg = geometry.gather_segments(table.start[at], table.stop[at])
    want = geometry.median_voxels(table.coords[rows].astype(np.int64), seg).reshape(-1, 3)
    assert np.array_equal(med[pick], want)
    t0 = time.perf_counter()
    geometry.median_voxels(table.coords.astype(np.int64), (table.stop - table.start))
    t_host = time.perf_counter() - t0
    with capsys.disabled():
        print("\n[C2 wall medians: %d walls, %d wall-voxel records] graph_from_image(..., 'wall_median') end to end %.1f ms with the medians "
              "on the device (group by pair + one wave a wall: %.2f ms of kernels); the route of rounds 2-3: %.1f ms to fetch the grouped "
              "records + %.1f ms of host arithmetic" % (keys.size, int(sizes.sum()), t_dev * 1e3, kernels_ms, t_fetch * 1e3, t_host * 1e3))
    assert len(g.edge_property('wall_median')) > 10000


def test_c4_tissue_filled_full_size_against_the_c_oracle(capsys):
    """The volume bench.py's `secondary.tissue_filled` times: no ellipsoid, every one of the ~50k cells present."""
    import time
    import torch
    c = synth.CONFIGS["C4"]
    dims, dtype = c["dims"], np.dtype(c["dtype"])
    ctx = dev.torch_context(0)
    vol, L = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"], ellipsoid=False)
    torch.cuda.synchronize()
    ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
    ctx.set_option(_capi.OPT_SWEEP_SHAPE, 1)                        # both tile shapes of the uint32 sweep at this size
    ctx.extract(_capi.F_ALL, L)
    got = fetch(ctx, L)
    ctx.set_option(_capi.OPT_SWEEP_SHAPE, 0)
    ctx.extract(_capi.F_ALL, L)
    got_narrow = fetch(ctx, L)
    host = vol.cpu().numpy().view(dtype)
    ctx.close()
    del vol
    want = onepass_c.extract(host, max_label=L)
    assert_same_accumulators(got, want, "tissue-filled C4 1024^3 full feature set, wide tiles")
    assert_same_accumulators(got_narrow, want, "tissue-filled C4 1024^3 full feature set, narrow tiles")
    present = int((want["count"] > 0).sum())
    assert present > 50000 and want["count"][1] == 0                # all cells, no background
    # the caller's side at this size: the tissue graph from these arrays (host time only, the sweep is done)
    from tissue_analysis_amd import DICT, Extraction, SpatialImage, SpatialImageAnalysis3D, graph_from_image
    x = Extraction.from_arrays(dims, got)
    props = ['boundingbox', 'volume', 'barycenter', 'L1', 'border', 'inertia_axis', 'wall_surface', 'epidermis_surface']
    best = 1e9
    for _ in range(3):
        sia = SpatialImageAnalysis3D(SpatialImage(np.zeros((2, 2, 2), dtype), voxelsize=synth.PARITY_VOXELSIZE), ignoredlabels=0,
                                     return_type=DICT, background=1, extraction=x)
        t0 = time.perf_counter()
        g = graph_from_image(sia, spatio_temporal_properties=list(props), ignore_cells_at_stack_margins=False)
        best = min(best, time.perf_counter() - t0)
    assert g.nb_vertices() == present and g.nb_edges() == want["pair_lo"].size
    with capsys.disabled():
        print("\n[graph_from_image on the tissue-filled C4 arrays: %d vertices, %d edges, 9 property columns] %.1f ms host time"
              % (g.nb_vertices(), g.nb_edges(), best * 1e3))


def test_c5_on_one_gpu_identities_windows_and_virtual_slabs():
    import torch
    c = synth.CONFIGS["C5"]
    dims, dtype = c["dims"], np.dtype(c["dtype"])
    n0, n1, n2 = dims
    ctx = dev.torch_context(0)
    vol, L = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"])
    torch.cuda.synchronize()
    ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
    ctx.extract(_capi.F_ALL, L)
    whole = fetch(ctx, L)
    dbg = ctx.debug_counters()
    assert dbg["range_flag"] == 0 and dbg["pair_overflow"] == 0

    # -- size-independent identities at full scale
    nvox = n0 * n1 * n2
    count, s1, s2, bbox = whole["count"], whole["sum1"], whole["sum2"], whole["bbox"]
    assert int(count.sum()) == nvox
    for d, n in enumerate(dims):
        assert int(s1[:, d].sum()) == (nvox // n) * (n * (n - 1) // 2)
        assert int(s2[:, [0, 3, 5][d]].sum()) == (nvox // n) * ((n - 1) * n * (2 * n - 1) // 6)
    lin = (nvox // (n0 * n1)) * (n0 * (n0 - 1) // 2) * (n1 * (n1 - 1) // 2)
    assert int(s2[:, 1].sum()) == lin                      # sum over voxels of a * b
    present = count > 0
    assert np.all(bbox[present, :3] >= 0) and np.all(bbox[present, 3:] <= np.asarray(dims)) and np.all(bbox[~present] == -1)
    lo, hi = whole["pair_lo"], whole["pair_hi"]
    assert np.all(lo < hi) and np.all(np.diff((lo.astype(np.int64) << 32) | hi) > 0)
    assert np.all(whole["pair_faces"].sum(axis=1) > 0) and present[lo].all() and present[hi].all()

    # -- three 32-plane windows, each swept as a volume of its own, bit-exact against the C oracle
    plane_bytes = n1 * n2 * dtype.itemsize
    for a0 in (0, n0 // 2 - 16, n0 - 32):
        ctx.set_volume_device(vol.data_ptr() + a0 * plane_bytes, dtype.itemsize, (32, n1, n2), keep=vol)
        ctx.extract(_capi.F_ALL, L)
        got = fetch(ctx, L)
        sub = vol[a0:a0 + 32].cpu().numpy().view(dtype)
        assert_same_accumulators(got, onepass_c.extract(sub, max_label=L), "C5 planes [%d, %d)" % (a0, a0 + 32))

    # -- eight virtual Z-slabs (the multi-GPU partition on one GPU): merged == unsharded
    parts = []
    for r in range(8):
        a_lo, a_hi = r * n0 // 8, (r + 1) * n0 // 8
        halo = 1 if a_lo > 0 else 0
        ctx.set_volume_device(vol.data_ptr() + (a_lo - halo) * plane_bytes, dtype.itemsize, (a_hi - a_lo + halo, n1, n2),
                              a0_origin=a_lo, has_low_halo=bool(halo), keep=vol)
        ctx.extract(_capi.F_ALL, L)
        parts.append(fetch(ctx, L))
    ctx.close()
    merged = onepass.merge(parts)
    for k in KEYS:
        assert np.array_equal(merged[k], whole[k]), "8 slabs vs unsharded: " + k
    del merged, parts


def test_c5_whole_volume_against_the_c_oracle(capsys):
    """The WHOLE 2048^3 volume against the C oracle: 34 GB to the host, 16 forked workers over Z-slabs (~0.5 Gvoxel/s) +
    merge; every integer array bit-exact.  Needs the volume in host memory next to the workers' results: SKIPPED (visibly,
    with the reason) on a host without ~46 GB available.  Its one-line result is printed past pytest's capture so that the
    tail of the driver's GPU-test record shows whether it ran."""
    import os
    import time
    import psutil
    import torch
    c = synth.CONFIGS["C5"]
    dims, dtype = c["dims"], np.dtype(c["dtype"])
    need = int(np.prod(dims)) * dtype.itemsize + (12 << 30)
    avail = psutil.virtual_memory().available
    if avail < need:
        msg = "C5 in full against the oracle: SKIPPED, %.0f GB of host memory available, %.0f needed" % (avail / 2 ** 30, need / 2 ** 30)
        with capsys.disabled():
            print("\n[" + msg + "]")
        pytest.skip(msg)
    ctx = dev.torch_context(0)
    vol, L = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"])
    torch.cuda.synchronize()
    ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
    ctx.extract(_capi.F_ALL, L)
    whole = fetch(ctx, L)
    dbg = ctx.debug_counters()
    assert dbg["range_flag"] == 0 and dbg["pair_overflow"] == 0
    ctx.close()
    t0 = time.perf_counter()
    host = vol.cpu().numpy().view(dtype)
    del vol
    torch.cuda.empty_cache()
    t1 = time.perf_counter()
    want = onepass_c_parallel(host, L, workers=min(16, os.cpu_count() or 1))
    t2 = time.perf_counter()
    assert_same_accumulators(whole, want, "C5 2048^3 full feature set, whole volume")
    with capsys.disabled():
        print("\n[C5 in full against the oracle: %d labels, %d pairs equal (every integer array); %.1f s to the host, %.1f s oracle]"
              % (int((want["count"] > 0).sum()), want["pair_lo"].size, t1 - t0, t2 - t1))


def test_c5_exchange_budget_eight_contexts_on_one_gpu(capsys):
    """SURVEY.md §8(e) leaves <= 0.25 ms per step for halo + reduce + merge at 8 GPUs.  No 8-GPU node runs this suite, so the
    exchange step is rehearsed at FULL C5 size on one GPU: eight contexts hold the eight Z-slabs (views of one resident
    2048^3 volume, halo plane included), each sweeps its slab, the per-label rows are reduced with torch ops in place of the
    two all-reduces, every context packs the pairs a slab face can split (ta_adjacency_pack_shared), the eight blocks sit in
    one tensor as an all-gather would leave them, every context merges them.  Checked: private lists + travelling pairs ==
    the unsharded adjacency, bit for bit.  Printed: per-slab kernel times of the exchange step (a single-GPU ESTIMATE of
    what each rank would do; the wire time of the collectives is not in it) for DESIGN.md §6."""
    import torch
    from tissue_analysis_amd import distributed as tad
    c = synth.CONFIGS["C5"]
    dims, dtype = c["dims"], np.dtype(c["dtype"])
    n0, n1, n2 = dims
    ctx0 = dev.torch_context(0)
    vol, L = dev.synth_slab(ctx0, dims, dtype, c["n_cells"], c["seed"])
    torch.cuda.synchronize()
    ctx0.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
    ctx0.extract(_capi.F_ALL, L)
    whole = fetch(ctx0, L)
    unsharded = []
    for _ in range(3):
        ctx0.extract(_capi.F_ALL, L)
        unsharded.append(ctx0.timing()["ms_sweep"])
    unsharded_ms = min(unsharded)
    # the slabs are cut by COST (label changes per plane, one streaming pass), not by plane count: the slowest slab sets the step
    events = ctx0.plane_events()
    cuts = tad.balanced_cuts(tad.plane_costs(events, n1 * n2), 8)
    ctx0.close()
    plane_bytes = n1 * n2 * dtype.itemsize
    world = 8

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        return a.elapsed_time(b)

    def slab_sweep_ms(lo, hi):
        halo = 1 if lo > 0 else 0
        cx = dev.torch_context(0)
        cx.set_volume_device(vol.data_ptr() + (lo - halo) * plane_bytes, dtype.itemsize, (hi - lo + halo, n1, n2), a0_origin=lo,
                             has_low_halo=bool(halo), keep=vol)
        cx.extract(_capi.F_ALL, L)
        ms = min(timed(lambda: cx.extract(_capi.F_ALL, L)) for _ in range(3))
        cx.close()
        return ms
    equal_ms = [slab_sweep_ms(*tad.slab_range(n0, world, r)) for r in range(world)]       # what equal plane counts give

    jobs, sweep_ms = [], []
    for r in range(world):
        lo, hi = tad.slab_range(n0, world, r, cuts)
        halo = 1 if lo > 0 else 0
        ctx = dev.torch_context(0)
        sums = torch.zeros((L + 1, 10), dtype=torch.int64, device="cuda:0")
        boxes = torch.zeros((L + 1, 6), dtype=torch.int32, device="cuda:0")
        ctx.set_volume_device(vol.data_ptr() + (lo - halo) * plane_bytes, dtype.itemsize, (hi - lo + halo, n1, n2), a0_origin=lo,
                              has_low_halo=bool(halo), keep=vol)
        ctx.bind_accumulators(sums.data_ptr(), boxes.data_ptr(), L, keep=(sums, boxes))
        ctx.extract(_capi.F_ALL, L)                                   # warm-up: tables sized
        sweep_ms.append(min(timed(lambda: ctx.extract(_capi.F_ALL, L)) for _ in range(3)))
        jobs.append((ctx, sums, boxes, lo, hi))
    npairs = [j[0].adjacency_size() for j in jobs]
    # the two all-reduces, as torch ops on one device (the RCCL wire time is not measured here)
    tot = torch.stack([j[1] for j in jobs]).sum(dim=0)
    mn = torch.stack([j[2] for j in jobs]).amin(dim=0)
    for ctx, s, b, _, _ in jobs:
        s.copy_(tot); b.copy_(mn)
        ctx.accumulators_reduced()
    torch.cuda.synchronize()
    got = tad.from_device_layout(tot.cpu().numpy(), mn.cpu().numpy())
    for k in ("count", "bbox", "sum1", "sum2"):
        assert np.array_equal(got[k], whole[k]), "reduced rows: " + k
    cap = 1 << 17
    words = _capi.exchange_words(cap)
    blocks = torch.empty((world * words,), dtype=torch.int64, device="cuda:0")
    pack_ms = [timed(lambda: ctx.adjacency_pack_shared(blocks[r * words:(r + 1) * words].data_ptr(), cap))
               for r, (ctx, _, _, _, _) in enumerate(jobs)]
    sent = [int(blocks[r * words].item()) for r in range(world)]
    assert max(sent) <= cap
    merge_ms = [timed(lambda: ctx.adjacency_merge_blocks(blocks.data_ptr(), world, cap)) for ctx, _, _, _, _ in jobs]
    # world-size-1 timing of the two in-place reduces' local cost is not a wire time either; the row sizes set the message
    row_mb = (tot.numel() * 8 + mn.numel() * 4) / 1e6
    # union of the private lists + the (identical) merged travelling pairs == the unsharded list
    bx = mn.cpu().numpy()
    keys_all, faces_all, travelling = [], [], None
    for ctx, _, _, lo, hi in jobs:
        plo, phi, pf = ctx.adjacency(allow_partial=True)
        excl = tad.slab_exclusive(bx, lo, hi)
        private = excl[plo] | excl[phi]
        key = (plo.astype(np.int64) << 32) | phi.astype(np.int64)
        keys_all.append(key[private]); faces_all.append(pf[private])
        t = (key[~private], pf[~private])
        if travelling is None:
            travelling = t
        else:
            assert np.array_equal(travelling[0], t[0]) and np.array_equal(travelling[1], t[1])
    keys = np.concatenate(keys_all + [travelling[0]])
    faces = np.concatenate(faces_all + [travelling[1]])
    order = np.argsort(keys, kind="stable")
    assert np.unique(keys).size == keys.size
    want_key = (whole["pair_lo"].astype(np.int64) << 32) | whole["pair_hi"].astype(np.int64)
    assert np.array_equal(keys[order], want_key) and np.array_equal(faces[order], whole["pair_faces"])
    for ctx, _, _, _, _ in jobs:
        ctx.close()
    # (the imbalance term of DESIGN.md §6: with cost-balanced cuts the slowest slab is within a few per cent of the mean)
    assert max(sweep_ms) / float(np.mean(sweep_ms)) < 1.08, (cuts, sweep_ms)
    with capsys.disabled():
        print("\n[C5 as 8 slabs on one GPU] unsharded sweep %.3f ms; equal plane counts: slab sweep mean %.3f max %.3f ms (max / mean %.3f); "
              "cost-balanced cuts %s: mean %.3f max %.3f ms (max / mean %.3f) -> unsharded / slowest slab = %.2fx"
              % (unsharded_ms, float(np.mean(equal_ms)), max(equal_ms), max(equal_ms) / float(np.mean(equal_ms)), cuts,
                 float(np.mean(sweep_ms)), max(sweep_ms), max(sweep_ms) / float(np.mean(sweep_ms)), unsharded_ms / max(sweep_ms)))
        print("\n[C5 as 8 slabs on one GPU, single-GPU estimate] per slab: sweep %.3f ms (max %.3f), local pairs %d..%d, "
              "travelling %d..%d of them (%.1f%%), all travelling pairs merged %d; pack_shared %.3f ms (max %.3f), "
              "merge_blocks of 8 blocks %.3f ms (max %.3f); rows to all-reduce %.1f MB; exchange block %.1f MB per rank"
              % (float(np.mean(sweep_ms)), max(sweep_ms), min(npairs), max(npairs), min(sent), max(sent),
                 100.0 * sum(sent) / sum(npairs), travelling[0].size, float(np.mean(pack_ms)), max(pack_ms),
                 float(np.mean(merge_ms)), max(merge_ms), row_mb, words * 8 / 1e6))
