"""BASELINE.json's configurations at FULL size on one GPU (C4 and C5 were exercised at reduced size only).

C3 / C4: 1024^3 uint32, 50k seeds, generated on the device, every integer array compared with the C oracle on the
whole volume (the oracle runs at ~60 Mvoxel/s: ~20 s each way).
C2: 512^3 uint16, 5 000 seeds at its stated size: its own feature subset (volume + bounding box + barycentre sums) and the
full feature set.  The tissue-filled C4 (the same generator WITHOUT the ellipsoid mask: 50 653 labels present, the
densest workload bench.py quotes): full feature set.
C5: 2048^3 uint32, 100k seeds (34 GB) on one GPU: size-independent identities at full scale, three 32-plane windows
(first / middle / last) bit-exact against the C oracle, and eight virtual Z-slabs with halo planes merged == unsharded.
"""
import numpy as np
import pytest

from oracle import onepass, onepass_c
from tissue_analysis_amd import _capi, device as dev, synth

from helpers import assert_same_accumulators

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
    ctx.extract(_capi.F_ALL, L)
    got4 = fetch(ctx, L)
    c3 = _capi.feature_mask(synth.CONFIGS["C3"]["features"])
    ctx.extract(c3, L)
    got3 = fetch(ctx, L)
    host = vol.cpu().numpy().view(dtype)
    ctx.close()
    del vol
    want = onepass_c.extract(host, max_label=L)
    assert_same_accumulators(got4, want, "C4 1024^3 full feature set")
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
    ctx.extract(_capi.F_ALL, L)
    got = fetch(ctx, L)
    host = vol.cpu().numpy().view(dtype)
    ctx.close()
    del vol
    want = onepass_c.extract(host, max_label=L)
    assert_same_accumulators(got, want, "tissue-filled C4 1024^3 full feature set")
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
    del vol
    merged = onepass.merge(parts)
    for k in KEYS:
        assert np.array_equal(merged[k], whole[k]), "8 slabs vs unsharded: " + k
