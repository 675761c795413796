"""Golden fixtures (tests/golden/*.npz, made by tests/golden/gen_golden.py): the generator and the
oracle still reproduce them (CPU), and the host-side float post-processing of the product turns
the golden INTEGER accumulators into the golden float features."""
import hashlib
import os

import numpy as np
import pytest

from oracle import onepass, onepass_c
from tissue_analysis_amd import synth
from tissue_analysis_amd.extraction import Extraction

from helpers import assert_same_accumulators

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def c1():
    return np.load(os.path.join(GOLD, "config1_128x128x64_u16.npz"))


@pytest.fixture(scope="module")
def c1_volume():
    c = synth.CONFIGS["C1"]
    return synth.voronoi_labels(c["dims"], c["n_cells"], c["seed"], np.dtype(c["dtype"]))


def test_generator_is_bit_reproducible(c1, c1_volume):
    digest = np.frombuffer(hashlib.sha256(c1_volume.tobytes()).digest(), dtype=np.uint8)
    assert np.array_equal(digest, c1["volume_sha256"])
    assert c1_volume.shape == tuple(c1["shape"])


def test_c_oracle_reproduces_config1_integers(c1, c1_volume):
    got = onepass_c.extract(c1_volume)
    assert_same_accumulators(got, c1, "config1")


def test_host_floats_from_golden_integers(c1):
    """Extraction (product host code) on the golden integers -> golden barycentres, covariances,
    eigenvalues, areas.  Tolerance 1e-6 relative as stated by the north star; achieved ~1e-12."""
    x = Extraction.from_arrays(tuple(c1["shape"]), dict((k, c1[k]) for k in c1.files))
    labels = c1["labels"]
    vs = c1["voxelsize"]
    np.testing.assert_allclose(x.barycenters(labels) * vs, c1["barycenter_real"], rtol=1e-12)
    np.testing.assert_allclose(x.volumes(labels) * vs.prod(), c1["volume_real"], rtol=1e-15)
    np.testing.assert_allclose(x.covariances(labels), c1["covariance"], rtol=1e-6, atol=1e-9)
    vecs, vals = x.inertia(labels)
    np.testing.assert_allclose(vals, c1["inertia_values_voxel"], rtol=1e-6, atol=1e-9)
    real = vals * np.linalg.norm(vecs * vs, axis=2)
    np.testing.assert_allclose(real, c1["inertia_values_real"], rtol=1e-6, atol=1e-9)
    # eigenvectors: up to sign, where the eigen-gap is healthy
    gv = c1["inertia_vectors"]
    gaps = np.minimum(np.abs(vals[:, 0] - vals[:, 1]), np.abs(vals[:, 1] - vals[:, 2])) / np.maximum(vals[:, 0], 1e-30)
    ok = gaps > 1e-3
    dots = np.abs(np.einsum("lij,lij->li", vecs, gv))
    assert ok.sum() > len(labels) // 2
    np.testing.assert_allclose(dots[ok], 1.0, atol=1e-6)
    face = np.array([vs[1] * vs[2], vs[2] * vs[0], vs[0] * vs[1]])
    for (i, j), area in zip(c1["wall_keys"], c1["wall_area_real"]):
        f = x.faces_between(int(i), [int(j)])[0].astype(float)
        assert abs(f.dot(face) - area) <= 1e-9 * max(1.0, area)
    for k, lo, hi in zip(c1["neighbor_keys"], c1["neighbor_ptr"][:-1], c1["neighbor_ptr"][1:]):
        assert x.neighbors_of(int(k)) == c1["neighbor_idx"][lo:hi].tolist()


def test_adversarial_fixtures_match_oracles():
    z = np.load(os.path.join(GOLD, "adversarial_small.npz"))
    names = sorted(set(k.split("__")[0] for k in z.files))
    assert len(names) == 5
    for n in names:
        vol = z[n + "__volume"]
        want = dict((k, z[n + "__" + k]) for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces"))
        assert_same_accumulators(onepass.extract(vol), want, n + " numpy")
        assert_same_accumulators(onepass_c.extract(vol), want, n + " C")


# ---- round-2 fixture: surface areas, first layer, wall voxels, wall medians of config C1 -------------------------------
@pytest.fixture(scope="module")
def c1r2():
    return np.load(os.path.join(GOLD, "config1_round2.npz"))


def test_round2_fixture_belongs_to_the_same_volume(c1, c1r2):
    assert np.array_equal(c1r2["volume_sha256"], c1["volume_sha256"]) and np.array_equal(c1r2["labels"], c1["labels"])
    # every face-adjacent pair of the integer fixture is a wall pair (18-connectivity finds edge contacts on top)
    faces = set(zip(c1["pair_lo"].tolist(), c1["pair_hi"].tolist()))
    walls = set(map(tuple, c1r2["wall_pairs"].tolist()))
    assert faces <= walls and (c1r2["wall_voxel_count"] > 0).all()


def test_surface_area_from_golden_integers(c1, c1r2):
    """Extraction.surface_faces (product host code) on the golden face counts -> the golden per-label surface areas the
    oracle got by summing the reference's wall areas one wall at a time."""
    x = Extraction.from_arrays(tuple(c1["shape"]), dict((k, c1[k]) for k in c1.files))
    labels = c1r2["labels"]
    faces = x.surface_faces(labels)                     # [n][3] faces per axis
    vs = c1["voxelsize"]
    face_area = np.array([vs[1] * vs[2], vs[0] * vs[2], vs[0] * vs[1]])
    np.testing.assert_allclose(faces.sum(axis=1), c1r2["surface_area_voxel"], rtol=0, atol=0)
    np.testing.assert_allclose(faces @ face_area, c1r2["surface_area_real"], rtol=1e-12)


def test_wall_medians_of_the_product_on_golden_walls(c1_volume, c1r2):
    """geometry._find_wall_median_voxel (exact medoid, the reference's <= 100 point branch) on walls rebuilt by brute force
    must pick the golden indices."""
    from tissue_analysis_amd.geometry import _find_wall_median_voxel
    vol = c1_volume
    checked = 0
    for (a, b), n, want in zip(c1r2["wall_pairs"].tolist(), c1r2["wall_voxel_count"].tolist(), c1r2["wall_median_index"].tolist()):
        if want < 0 or checked >= 25:
            continue
        # the pair's wall voxels by brute force over the 18 offsets, np.where order
        ma, mb = vol == a, vol == b
        pad = np.pad(mb, 1); pada = np.pad(ma, 1)
        near_b = np.zeros_like(ma); near_a = np.zeros_like(ma)
        for da in (-1, 0, 1):
            for db in (-1, 0, 1):
                for dc in (-1, 0, 1):
                    if 0 < abs(da) + abs(db) + abs(dc) < 3:
                        s = (slice(1 + da, 1 + da + vol.shape[0]), slice(1 + db, 1 + db + vol.shape[1]), slice(1 + dc, 1 + dc + vol.shape[2]))
                        near_b |= pad[s]; near_a |= pada[s]
        xyz = np.array(np.where((ma & near_b) | (mb & near_a)))
        assert xyz.shape[1] == n, (a, b)
        assert _find_wall_median_voxel(xyz.T) == want, (a, b)
        checked += 1
    assert checked >= 10
