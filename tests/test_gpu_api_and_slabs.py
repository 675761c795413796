"""GPU tests through the C ABI beyond the raw accumulators: the drop-in class against the oracle's
API mirror, the golden config-1 fixture, the device generator, Z-slabs with halos (the multi-GPU
partition exercised on one GPU) and size-independent properties at larger sizes."""
import os

import numpy as np
import pytest

from oracle import onepass, onepass_c
from oracle.sia_oracle import OracleSIA
from oracle import sia_oracle
from tissue_analysis_amd import DICT, SpatialImage, SpatialImageAnalysis, _capi, synth
from tissue_analysis_amd import distributed as tad
from tissue_analysis_amd.extraction import extract_volume

from api_compare import compare_api
from helpers import assert_same_accumulators, random_blocks, voronoi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VS = synth.PARITY_VOXELSIZE


def test_dropin_class_matches_reference_mirror_on_gpu():
    vol = voronoi((40, 36, 64), 40, 41, np.uint16)
    sia = SpatialImageAnalysis(SpatialImage(vol, voxelsize=VS), ignoredlabels=0, return_type=DICT, background=1)
    ref = OracleSIA(vol, ignoredlabels=0, return_type=sia_oracle.DICT, background=1, voxelsize=VS)
    compare_api(sia, ref)


def test_dropin_class_u32_fortran_order():
    vol = np.asfortranarray(voronoi((24, 30, 28), 20, 42, np.uint32))
    sia = SpatialImageAnalysis(SpatialImage(vol, voxelsize=(1.0, 2.0, 0.25)), ignoredlabels=0, background=1)
    ref = OracleSIA(np.ascontiguousarray(vol), ignoredlabels=0, background=1, voxelsize=(1.0, 2.0, 0.25))
    compare_api(sia, ref)


def test_int64_images_are_accepted_like_the_reference():
    vol = voronoi((12, 14, 16), 6, 43, np.uint16).astype(np.int64)
    sia = SpatialImageAnalysis(vol, background=1)
    ref = OracleSIA(vol.astype(np.uint16), background=1)
    assert sia.labels() == ref.labels()
    assert sia.volume() == ref.volume()


def test_golden_config1_on_gpu(gpu_ctx):
    c = synth.CONFIGS["C1"]
    gold = np.load(os.path.join(GOLD, "config1_128x128x64_u16.npz"))
    vol = synth.voronoi_labels(c["dims"], c["n_cells"], c["seed"], np.dtype(c["dtype"]))
    got = extract_volume(vol, context=gpu_ctx).as_arrays()
    assert_same_accumulators(got, gold, "config1 vs golden")
    sia = SpatialImageAnalysis(SpatialImage(vol, voxelsize=VS), ignoredlabels=0, return_type=DICT, background=1)
    labels = [int(l) for l in gold["labels"]]
    assert sia.labels() == labels
    bary = sia.center_of_mass(labels, real=True)
    np.testing.assert_allclose(np.stack([bary[l] for l in labels]), gold["barycenter_real"], rtol=1e-6)
    _, vals = sia.inertia_axis(labels, real=True)
    np.testing.assert_allclose(np.stack([vals[l] for l in labels]), gold["inertia_values_real"], rtol=1e-6, atol=1e-9)
    walls = sia.wall_areas(real=True)
    assert sorted(walls) == [tuple(int(v) for v in k) for k in gold["wall_keys"]]
    np.testing.assert_allclose([walls[k] for k in sorted(walls)], gold["wall_area_real"], rtol=1e-6)
    assert sorted(sia.labels_at_stack_margins()) == gold["border"].tolist()
    assert sorted(sia.cell_first_layer()) == gold["first_layer"].tolist()


def test_adversarial_fixtures_on_gpu(gpu_ctx):
    z = np.load(os.path.join(GOLD, "adversarial_small.npz"))
    for n in sorted(set(k.split("__")[0] for k in z.files)):
        want = dict((k, z[n + "__" + k]) for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces"))
        for impl in (0, 1):
            got = extract_volume(z[n + "__volume"], context=gpu_ctx, impl=impl).as_arrays()
            assert_same_accumulators(got, want, "%s impl=%d" % (n, impl))
    gpu_ctx.set_option(_capi.OPT_IMPL, 0)


def _device_volume(ctx, dims, dtype, n_cells, seed, a_lo=0, a_hi=None):
    a_hi = dims[0] if a_hi is None else a_hi
    seeds, grid = synth.make_seeds(dims, n_cells, seed)
    nbytes = (a_hi - a_lo) * dims[1] * dims[2] * np.dtype(dtype).itemsize
    ptr = ctx.malloc(nbytes)
    ctx.synth_voronoi(ptr, dtype, dims, a_lo, a_hi - a_lo, seeds, grid, synth.ellipsoid_tables(dims))
    return ptr, nbytes, seeds.shape[0] + 1


@pytest.mark.parametrize("dtype", [np.uint16, np.uint32])
def test_device_generator_is_bit_identical_to_numpy(gpu_ctx, dtype):
    dims = (20, 33, 70)
    ptr, nbytes, _ = _device_volume(gpu_ctx, dims, dtype, 25, 51)
    host = np.zeros(dims, dtype=dtype)
    gpu_ctx.d2h(host, ptr)
    gpu_ctx.free(ptr)
    assert np.array_equal(host, synth.voronoi_labels(dims, 25, 51, dtype))


@pytest.mark.parametrize("cuts", [[0, 40], [0, 13, 40], [0, 1, 2, 25, 40], [0, 10, 20, 30, 40]])
def test_zslabs_with_halo_sum_to_the_whole_volume(gpu_ctx, cuts):
    """Each slab is swept separately on the GPU from device memory (with its low halo plane and
    global origin); merging the per-slab integers must give the unsharded result exactly."""
    dims, dtype = (40, 24, 264), np.uint32
    vol = synth.voronoi_labels(dims, 60, 52, dtype)
    whole = onepass_c.extract(vol)
    L = whole["max_label"]
    ptr = gpu_ctx.malloc(vol.nbytes)
    gpu_ctx.h2d(ptr, vol)
    parts = []
    plane_bytes = dims[1] * dims[2] * vol.dtype.itemsize
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        halo = 1 if lo > 0 else 0
        gpu_ctx.set_volume_device(ptr + (lo - halo) * plane_bytes, vol.dtype.itemsize,
                                  (hi - lo + halo, dims[1], dims[2]), a0_origin=lo, has_low_halo=bool(halo))
        gpu_ctx.extract(_capi.F_ALL, L)
        count, bbox, s1, s2 = gpu_ctx.labels()
        plo, phi, pf = gpu_ctx.adjacency()
        parts.append(dict(max_label=L, count=count, bbox=bbox, sum1=s1, sum2=s2, pair_lo=plo, pair_hi=phi, pair_faces=pf))
    gpu_ctx.free(ptr)
    assert_same_accumulators(onepass.merge(parts), whole, "cuts=%s" % cuts)


def test_adjacency_merge_on_device(gpu_ctx):
    """ta_adjacency_merge: inserting a foreign pair list sums face counts of equal keys."""
    vol = voronoi((16, 20, 64), 12, 53, np.uint16)
    x = extract_volume(vol, context=gpu_ctx)
    keys = (x.pair_lo.astype(np.uint64) << np.uint64(32)) | x.pair_hi.astype(np.uint64)
    extra_keys = np.concatenate([keys[:5], np.array([(7 << 32) | 9000, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)])
    extra_faces = np.ones((extra_keys.size, 3), dtype=np.uint64)
    kp, fp = gpu_ctx.malloc(extra_keys.nbytes), gpu_ctx.malloc(extra_faces.nbytes)
    gpu_ctx.h2d(kp, extra_keys)
    gpu_ctx.h2d(fp, extra_faces)
    gpu_ctx.adjacency_merge(kp, fp, extra_keys.size)
    lo, hi, faces = gpu_ctx.adjacency()
    gpu_ctx.free(kp)
    gpu_ctx.free(fp)
    want_lo, want_hi, want_f = onepass.compact_pairs(
        np.concatenate([x.pair_lo, (extra_keys[:6] >> np.uint64(32)).astype(np.uint32)]),
        np.concatenate([x.pair_hi, (extra_keys[:6] & np.uint64(0xFFFFFFFF)).astype(np.uint32)]),
        np.concatenate([x.pair_faces, extra_faces[:6]]))
    assert np.array_equal(lo, want_lo) and np.array_equal(hi, want_hi) and np.array_equal(faces, want_f)


def test_config2_sized_volume_against_c_oracle(gpu_ctx):
    """256^3 uint16 (1/8 of C2) with C2's feature subset and full features: bit-exact vs the C oracle."""
    dims, dtype = (256, 256, 256), np.uint16
    ptr, nbytes, max_label = _device_volume(gpu_ctx, dims, dtype, 625, 1)
    host = np.zeros(dims, dtype=dtype)
    gpu_ctx.d2h(host, ptr)
    want = onepass_c.extract(host, max_label=max_label)
    gpu_ctx.set_volume_device(ptr, 2, dims)
    for feats in (_capi.feature_mask(synth.CONFIGS["C2"]["features"]), _capi.F_ALL):
        gpu_ctx.extract(feats, max_label)
        count, bbox, s1, s2 = gpu_ctx.labels()
        assert np.array_equal(count, want["count"]) and np.array_equal(bbox, want["bbox"]) and np.array_equal(s1, want["sum1"])
        if feats & _capi.F_MOMENT2:
            assert np.array_equal(s2, want["sum2"])
        if feats & _capi.F_ADJACENCY:
            lo, hi, f = gpu_ctx.adjacency()
            assert np.array_equal(lo, want["pair_lo"]) and np.array_equal(hi, want["pair_hi"]) and np.array_equal(f, want["pair_faces"])
    gpu_ctx.free(ptr)


def test_full_size_properties_512_cubed_u32(gpu_ctx):
    """At sizes the oracle does not finish in seconds: size-independent properties.
    sum(count) = nvox; sum of first moments = closed form over the box; total faces per axis =
    number of label changes along that axis (checked against a slab-wise C-oracle count on 1/16 of
    the volume); idempotence (second run identical)."""
    dims, dtype = (512, 512, 512), np.uint32
    ptr, nbytes, max_label = _device_volume(gpu_ctx, dims, dtype, 6250, 2)
    gpu_ctx.set_volume_device(ptr, 4, dims)
    gpu_ctx.extract(_capi.F_ALL, max_label)
    count, bbox, s1, s2 = gpu_ctx.labels()
    lo, hi, f = gpu_ctx.adjacency()
    nvox = dims[0] * dims[1] * dims[2]
    assert int(count.sum()) == nvox
    for d in range(3):
        n = dims[d]
        assert int(s1[:, d].sum()) == (nvox // n) * (n * (n - 1) // 2)
        assert int(s2[:, [0, 3, 5][d]].sum()) == (nvox // n) * ((n - 1) * n * (2 * n - 1) // 6)
    assert int(s2[:, 1].sum()) == dims[2] * (dims[0] * (dims[0] - 1) // 2) * (dims[1] * (dims[1] - 1) // 2)
    present = count > 0
    assert np.all(bbox[present, :3] >= 0) and np.all(bbox[present, 3:] <= np.asarray(dims))
    assert np.all((bbox[present, 3:] - bbox[present, :3]).prod(axis=1) >= count[present])
    assert np.all(lo < hi) and np.all(np.diff((lo.astype(np.int64) << 32) | hi) > 0)
    # first 32 planes: exact comparison with the C oracle
    sub = np.zeros((32,) + dims[1:], dtype=dtype)
    gpu_ctx.d2h(sub, ptr)
    want = onepass_c.extract(sub, max_label=max_label)
    gpu_ctx.set_volume_device(ptr, 4, (32,) + dims[1:])
    gpu_ctx.extract(_capi.F_ALL, max_label)
    c2, b2, s12, s22 = gpu_ctx.labels()
    l2, h2, f2 = gpu_ctx.adjacency()
    assert_same_accumulators(dict(count=c2, bbox=b2, sum1=s12, sum2=s22, pair_lo=l2, pair_hi=h2, pair_faces=f2), want, "first 32 planes")
    # idempotence on the full volume
    gpu_ctx.set_volume_device(ptr, 4, dims)
    gpu_ctx.extract(_capi.F_ALL, max_label)
    c3, b3, s13, s23 = gpu_ctx.labels()
    l3, h3, f3 = gpu_ctx.adjacency()
    assert np.array_equal(c3, count) and np.array_equal(b3, bbox) and np.array_equal(s13, s1) and np.array_equal(s23, s2)
    assert np.array_equal(l3, lo) and np.array_equal(h3, hi) and np.array_equal(f3, f)
    gpu_ctx.free(ptr)


def test_worst_case_noise_volume_spills_correctly(gpu_ctx):
    """Every voxel a different label: LDS tables overflow and the global spill paths must still be exact."""
    rng = np.random.default_rng(5)
    vol = rng.permutation(16 * 16 * 256).astype(np.uint32).reshape(16, 16, 256) + 1
    want = onepass_c.extract(vol)
    got = extract_volume(vol, context=gpu_ctx, impl=0).as_arrays()
    assert_same_accumulators(got, want, "noise")
    assert gpu_ctx.debug_counters()["label_spills"] > 0


def test_graph_from_image_end_to_end_on_gpu():
    """The reference's documented workflow (TGI:260-284) from a raw labelled image: one GPU sweep,
    then host-side table assembly; compared with the oracle's restatement on the same image."""
    from oracle import graph_oracle
    from tissue_analysis_amd import graph_from_image, property_graph_to_dataframe
    from graph_compare import compare_graph
    props = ['boundingbox', 'volume', 'barycenter', 'L1', 'border', 'inertia_axis', 'wall_surface', 'epidermis_surface',
             'wall_median']
    vol = voronoi((48, 40, 72), 60, 41, np.uint16)
    img = SpatialImage(vol, voxelsize=synth.PARITY_VOXELSIZE)
    g = graph_from_image(img, spatio_temporal_properties=list(props), min_contact_area=2.0)
    want = graph_oracle.graph_tables(vol, None, 1, list(props), True, True, 2.0, voxelsize=synth.PARITY_VOXELSIZE)
    assert g.nb_vertices() > 10 and g.nb_edges() > 10
    compare_graph(g, want)
    assert len(property_graph_to_dataframe(g, 'vertex')) == g.nb_vertices()
