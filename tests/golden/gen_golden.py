"""Generate the golden fixtures under tests/golden/ (run from the repo root):

    python tests/golden/gen_golden.py

Inputs come from the repo's own deterministic generator; expected outputs come from the oracle
(oracle/sia_oracle.py = the reference's per-label scipy.ndimage algorithm restated for Python 3,
oracle/onepass.py = the exact-integer spec).  Nothing is read from /root/reference.  The reference
cannot be imported here (Python 2 + openalea), so these vectors pin "the restatement on scipy
1.15.3 / numpy 2.2.6", see DESIGN.md.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import onepass                       # noqa: E402
from oracle.sia_oracle import OracleSIA, DICT    # noqa: E402
from tissue_analysis_amd import synth            # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def csr(d, keys):
    ptr, idx = [0], []
    for k in keys:
        idx.extend(sorted(int(v) for v in d[k]))
        ptr.append(len(idx))
    return np.asarray(ptr, dtype=np.int64), np.asarray(idx, dtype=np.int64)


def config1():
    c = synth.CONFIGS["C1"]
    vol = synth.voronoi_labels(c["dims"], c["n_cells"], c["seed"], np.dtype(c["dtype"]))
    vs = synth.PARITY_VOXELSIZE
    ints = onepass.extract(vol)
    sia = OracleSIA(vol, ignoredlabels=0, return_type=DICT, background=1, voxelsize=vs)
    labels = sia.labels()
    nb_all = sia.neighbors()                     # labels + background, like the reference
    keys = sorted(nb_all)
    nptr, nidx = csr(nb_all, keys)
    bary = sia.center_of_mass(labels, real=True)
    vol_real = sia.volume(labels, real=True)
    vecs, vals = sia.inertia_axis(labels, real=True)
    vecs_v, vals_v = sia.inertia_axis(labels, real=False)
    cov = np.stack([sia.covariance(l) for l in labels])
    walls = sia.wall_areas(real=True)
    wkeys = sorted(walls)
    out = dict(
        volume_sha256=np.frombuffer(hashlib.sha256(vol.tobytes()).digest(), dtype=np.uint8),
        shape=np.asarray(vol.shape), voxelsize=np.asarray(vs), max_label=np.asarray(ints["max_label"]),
        count=ints["count"], bbox=ints["bbox"], sum1=ints["sum1"], sum2=ints["sum2"],
        pair_lo=ints["pair_lo"], pair_hi=ints["pair_hi"], pair_faces=ints["pair_faces"],
        labels=np.asarray(labels), barycenter_real=np.stack([bary[l] for l in labels]),
        volume_real=np.asarray([vol_real[l] for l in labels]),
        covariance=cov, inertia_values_real=np.stack([vals[l] for l in labels]),
        inertia_values_voxel=np.stack([vals_v[l] for l in labels]),
        inertia_vectors=np.stack([np.stack(vecs[l]) for l in labels]),
        neighbor_keys=np.asarray(keys), neighbor_ptr=nptr, neighbor_idx=nidx,
        wall_keys=np.asarray(wkeys, dtype=np.int64).reshape(-1, 2),
        wall_area_real=np.asarray([walls[k] for k in wkeys]),
        border=np.asarray(sia.labels_at_stack_margins()),
        first_layer=np.asarray(sia.cell_first_layer()),
    )
    np.savez_compressed(os.path.join(HERE, "config1_128x128x64_u16.npz"), **out)
    print("config1: %d labels, %d pairs, %d walls" % (len(labels), ints["pair_lo"].size, len(wkeys)))


def config1_round2():
    """Config C1 again, the methods added in round 2 (SURVEY.md §8f-3 and the per-label surface area): expected outputs
    from the oracle's restatement of the reference's loops -- the per-pair bounding-box crops + two 18-connectivity
    dilations of wall_voxels_between_two_cells (SIA:759-806), the 6-stencil of voxel_first_layer (SIA:1024-1046), the
    exact medoid of find_wall_median_voxel (SIA:1499-1585).  Wall voxels are stored per pair as count + SHA-256 of the
    int32 (3, N) coordinate array in np.where order (the arrays themselves would be ~2 MB)."""
    c = synth.CONFIGS["C1"]
    vol = synth.voronoi_labels(c["dims"], c["n_cells"], c["seed"], np.dtype(c["dtype"]))
    vs = synth.PARITY_VOXELSIZE
    sia = OracleSIA(vol, ignoredlabels=0, return_type=DICT, background=1, voxelsize=vs)
    labels = sia.labels()
    area_real, area_vox = sia.surface_area(labels, real=True), sia.surface_area(labels, real=False)
    layer = sia.voxel_first_layer(keep_background=True)
    walls = sia.wall_voxels_per_cells_pairs()                 # every neighbouring pair, background included
    wkeys = sorted(walls)
    counts, digests, medians = [], [], []
    from oracle.sia_oracle import find_wall_median_voxel
    for k in wkeys:
        xyz = np.ascontiguousarray(np.asarray(walls[k]).astype(np.int32))
        counts.append(xyz.shape[1])
        digests.append(np.frombuffer(hashlib.sha256(xyz.tobytes()).digest(), dtype=np.uint8))
        medians.append(find_wall_median_voxel(np.asarray(walls[k]).T) if 0 < xyz.shape[1] <= 100 and xyz.shape[1] != 3 else -1)   # (3 points: a 3x3 array, the orientation of which the reference guesses)
    epi = sorted(sia.wall_voxels_per_cells_pairs(only_epidermis=True))
    out = dict(
        volume_sha256=np.frombuffer(hashlib.sha256(vol.tobytes()).digest(), dtype=np.uint8),
        labels=np.asarray(labels),
        surface_area_real=np.asarray([area_real[l] for l in labels]),
        surface_area_voxel=np.asarray([area_vox[l] for l in labels]),
        first_layer_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(layer).tobytes()).digest(), dtype=np.uint8),
        first_layer_dtype=np.asarray(str(layer.dtype)), first_layer_nonzero=np.asarray(int(np.count_nonzero(layer))),
        wall_pairs=np.asarray(wkeys, dtype=np.int64).reshape(-1, 2), wall_voxel_count=np.asarray(counts, dtype=np.int64),
        wall_voxel_sha256=np.stack(digests), wall_median_index=np.asarray(medians, dtype=np.int64),
        epidermis_wall_pairs=np.asarray(epi, dtype=np.int64).reshape(-1, 2),
    )
    np.savez_compressed(os.path.join(HERE, "config1_round2.npz"), **out)
    print("config1_round2: %d labels, %d wall pairs (%d voxels), %d epidermis pairs" % (len(labels), len(wkeys), sum(counts), len(epi)))


def adversarial():
    """Tiny volumes for the edge cases of SURVEY.md §4(3); inputs are stored with the outputs."""
    rng = np.random.default_rng(123)
    cases = {}
    a = np.ones((5, 6, 7), dtype=np.uint16); a[2, 3, 4] = 9; a[0, 0, 0] = 40000; a[4, 5, 6] = 65535
    cases["single_voxels_and_corners_u16"] = a
    b = rng.integers(0, 4, size=(6, 5, 9)).astype(np.uint32) * 70000        # ids 0, 70000, 140000, 210000
    cases["sparse_ids_above_65535_u32"] = b
    c = np.zeros((4, 4, 4), dtype=np.uint16); c[1:3, 1:3, 1:3] = 3
    cases["label_zero_surrounds_cube"] = c
    d = (np.arange(3 * 4 * 5) % 7 + 1).astype(np.uint16).reshape(3, 4, 5)
    cases["striped_small"] = d
    e = rng.integers(1, 30, size=(9, 3, 70)).astype(np.uint32)
    cases["noise_u32"] = e
    out = {}
    for name, vol in cases.items():
        r = onepass.extract(vol)
        out[name + "__volume"] = vol
        for k in ("count", "bbox", "sum1", "sum2", "pair_lo", "pair_hi", "pair_faces"):
            out[name + "__" + k] = r[k]
    np.savez_compressed(os.path.join(HERE, "adversarial_small.npz"), **out)
    print("adversarial: %d cases" % len(cases))


def docstring_case():
    """SURVEY.md §8(c) golden item (1): the 4x6 image of the reference's docstrings with the answers the docstrings
    give (spatial_image_analysis.py:344-353, 437-450, 498-511, 561-574, 970-982, 1219-1226), typed in by hand; the
    oracle must reproduce every one of them before the file is written."""
    import json
    image = [[1, 2, 7, 7, 1, 1], [1, 6, 5, 7, 3, 3], [2, 2, 1, 7, 3, 3], [1, 1, 1, 4, 1, 1]]
    gold = dict(
        image=image, labels=[1, 2, 3, 4, 5, 6, 7],
        volume={"1": 10.0, "2": 3.0, "3": 4.0, "4": 1.0, "5": 1.0, "6": 1.0, "7": 4.0},
        center_of_mass={"1": [1.8, 2.2999999999999998, 0.0], "2": [1.3333333333333333, 0.66666666666666663, 0.0],
                        "3": [1.5, 4.5, 0.0], "4": [3.0, 3.0, 0.0], "5": [1.0, 2.0, 0.0], "6": [1.0, 1.0, 0.0],
                        "7": [0.75, 2.75, 0.0]},
        boundingbox={"1": [[0, 4], [0, 6], [0, 1]], "2": [[0, 3], [0, 2], [0, 1]], "3": [[1, 3], [4, 6], [0, 1]],
                     "4": [[3, 4], [3, 4], [0, 1]], "5": [[1, 2], [2, 3], [0, 1]], "6": [[1, 2], [1, 2], [0, 1]],
                     "7": [[0, 3], [2, 4], [0, 1]]},
        neighbors={"1": [2, 3, 4, 5, 6, 7], "2": [1, 6, 7], "3": [1, 7], "4": [1, 7], "5": [1, 6, 7], "6": [1, 2, 5],
                   "7": [1, 2, 3, 4, 5]},
        wall_areas={"1,2": 5.0, "1,3": 4.0, "1,4": 2.0, "1,5": 1.0, "1,6": 1.0, "1,7": 2.0, "2,6": 2.0, "2,7": 1.0,
                    "3,7": 2.0, "4,7": 1.0, "5,6": 1.0, "5,7": 2.0},
        wall_median_example=dict(points=[[0, 0, 0], [0, 1, 0], [0, 2, 0], [0, 3, 0], [0, 4, 0]], index=2),
    )
    sia = OracleSIA(np.array(image, dtype=np.uint16))
    assert sia.labels() == gold["labels"]
    vol, com, bb, nb = sia.volume(), sia.center_of_mass(), sia.boundingbox(), sia.neighbors()
    for l in gold["labels"]:
        assert vol[l] == gold["volume"][str(l)]
        assert np.allclose(com[l], gold["center_of_mass"][str(l)], rtol=1e-15)
        assert [[s.start, s.stop] for s in bb[l]] == gold["boundingbox"][str(l)]
        assert sorted(int(v) for v in nb[l]) == gold["neighbors"][str(l)]
    assert dict(("%d,%d" % k, float(v)) for k, v in sia.wall_areas().items()) == gold["wall_areas"]
    with open(os.path.join(HERE, "docstring_4x6.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    print("docstring_4x6.json written")


if __name__ == "__main__":
    config1()
    config1_round2()
    adversarial()
    docstring_case()
