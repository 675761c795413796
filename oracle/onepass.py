"""ORACLE -- test infrastructure only.  Never imported by the product package.

One-pass integer formulation of the hot path in plain numpy: per-label voxel counts,
bounding boxes, raw first/second coordinate moments, and per-axis shared-face counts
for every unordered label pair.  It is the *exact-integer* spec the HIP kernels must
match bit for bit (SURVEY.md §8 "Semantics"), and tests prove it equivalent to the
per-label scipy restatement in ``sia_oracle.py`` (which follows SIA line by line):

  count[l]            = #{p : V[p] = l}                          ~ nd.sum           SIA:1231
  bbox[l]             = [min_d, max_d + 1) per axis, -1 if absent ~ nd.find_objects  SIA:517
  sum1[l][d]          = sum of p_d over voxels of l              ~ nd.center_of_mass SIA:466
  sum2[l][(d,e)]      = sum of p_d * p_e, order 00,01,02,11,12,22 ~ cov = P.P^T/N    SIA:137-150
  faces[(lo,hi)][d]   = #{p : {V[p], V[p+e_d]} = {lo,hi}}, lo<hi ~ binary_dilation   SIA:45-52, 947-956

Pinning status: see sia_oracle.py ("parity unpinned" beyond the docstring examples).
"""
from __future__ import annotations

import numpy as np

PAIR_ORDER = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))


def extract(image, max_label=None, origin=(0, 0, 0), own_first_plane=True):
    """Integer accumulators of a 3D label array.

    origin: global coordinate of image[0,0,0] (slab runs).
    own_first_plane=False: plane 0 along axis 0 is a halo -- it contributes faces with
    plane 1 but no voxels (the multi-GPU slab convention: a face belongs to the slab
    that owns its higher voxel along axis 0).
    """
    V = np.asarray(image)
    if V.ndim == 2:
        V = V[:, :, None]
    assert V.ndim == 3
    L = int(V.max()) if max_label is None else int(max_label)
    if V.size and int(V.max()) > L:
        raise ValueError("label %d exceeds max_label %d" % (int(V.max()), L))
    own = V if own_first_plane else V[1:]
    o0 = origin[0] + (0 if own_first_plane else 1)
    lab = own.ravel().astype(np.int64)
    order = np.argsort(lab, kind="stable")
    slab = lab[order]
    present, starts = np.unique(slab, return_index=True)
    count = np.zeros(L + 1, dtype=np.uint64)
    bbox = np.full((L + 1, 6), -1, dtype=np.int32)
    sum1 = np.zeros((L + 1, 3), dtype=np.uint64)
    sum2 = np.zeros((L + 1, 6), dtype=np.uint64)
    if lab.size:
        coords = np.unravel_index(order, own.shape)
        coords = [c.astype(np.int64) + o for c, o in zip(coords, (o0, origin[1], origin[2]))]
        count[present] = np.diff(np.append(starts, slab.size)).astype(np.uint64)
        for d in range(3):
            sum1[present, d] = np.add.reduceat(coords[d], starts).astype(np.uint64)
            bbox[present, d] = np.minimum.reduceat(coords[d], starts)
            bbox[present, 3 + d] = np.maximum.reduceat(coords[d], starts) + 1
        for k, (d, e) in enumerate(PAIR_ORDER):
            sum2[present, k] = np.add.reduceat(coords[d] * coords[e], starts).astype(np.uint64)
    lo, hi, faces = face_pairs(V, L, first_plane_is_halo=not own_first_plane)
    return dict(max_label=L, count=count, bbox=bbox, sum1=sum1, sum2=sum2,
                pair_lo=lo, pair_hi=hi, pair_faces=faces)


def face_pairs(V, L=None, first_plane_is_halo=False):
    """Unique unordered label pairs sharing a voxel face, with per-axis face counts.

    With first_plane_is_halo the faces lying inside plane 0 (axes 1 and 2) are skipped:
    they belong to the slab that owns that plane."""
    V = np.asarray(V)
    L = int(V.max()) if L is None else int(L)
    base = np.int64(L + 1)
    keys, axes = [], []
    for d in range(3):
        W = V[1:] if (first_plane_is_halo and d != 0) else V
        if W.shape[d] < 2:
            continue
        a = np.take(W, np.arange(0, W.shape[d] - 1), axis=d).astype(np.int64)
        b = np.take(W, np.arange(1, W.shape[d]), axis=d).astype(np.int64)
        m = a != b
        a, b = a[m], b[m]
        keys.append(np.minimum(a, b) * base + np.maximum(a, b))
        axes.append(np.full(a.size, d, dtype=np.int64))
    if not keys or sum(k.size for k in keys) == 0:
        z = np.zeros(0, dtype=np.uint32)
        return z, z.copy(), np.zeros((0, 3), dtype=np.uint64)
    keys = np.concatenate(keys)
    axes = np.concatenate(axes)
    uk, inv = np.unique(keys, return_inverse=True)
    faces = np.zeros((uk.size, 3), dtype=np.uint64)
    np.add.at(faces, (inv, axes), 1)
    return (uk // base).astype(np.uint32), (uk % base).astype(np.uint32), faces


def merge(parts):
    """Combine per-slab accumulators (sum / min / max, pairs re-keyed and summed).

    This is the CPU statement of the multi-GPU reduce."""
    L = parts[0]["max_label"]
    count = sum(p["count"].astype(np.uint64) for p in parts)
    sum1 = sum(p["sum1"] for p in parts)
    sum2 = sum(p["sum2"] for p in parts)
    big = np.iinfo(np.int32).max
    mins = np.stack([np.where(p["bbox"][:, :3] < 0, big, p["bbox"][:, :3]) for p in parts]).min(0)
    maxs = np.stack([p["bbox"][:, 3:] for p in parts]).max(0)
    bbox = np.concatenate([np.where(mins == big, -1, mins), maxs], axis=1).astype(np.int32)
    base = np.int64(L + 1)
    keys = np.concatenate([p["pair_lo"].astype(np.int64) * base + p["pair_hi"].astype(np.int64)
                           for p in parts])
    faces = np.concatenate([p["pair_faces"] for p in parts])
    uk, inv = np.unique(keys, return_inverse=True)
    out = np.zeros((uk.size, 3), dtype=np.uint64)
    np.add.at(out, inv, faces)
    return dict(max_label=L, count=count, bbox=bbox, sum1=sum1, sum2=sum2,
                pair_lo=(uk // base).astype(np.uint32), pair_hi=(uk % base).astype(np.uint32),
                pair_faces=out)


def compact_pairs(lo, hi, faces):
    """sort + segmented reduce of a pair COO (what ta_pairs_compact does on the device)."""
    lo = np.asarray(lo, dtype=np.int64)
    hi = np.asarray(hi, dtype=np.int64)
    faces = np.asarray(faces, dtype=np.uint64).reshape(-1, 3)
    if lo.size == 0:
        z = np.zeros(0, dtype=np.uint32)
        return z, z.copy(), np.zeros((0, 3), dtype=np.uint64)
    keys = (lo << 32) | hi
    uk, inv = np.unique(keys, return_inverse=True)
    out = np.zeros((uk.size, 3), dtype=np.uint64)
    np.add.at(out, inv, faces)
    return (uk >> 32).astype(np.uint32), (uk & 0xFFFFFFFF).astype(np.uint32), out
