"""Deterministic synthetic labelled volumes (jittered-grid Voronoi tissue in an ellipsoid).

This is the workload generator named by SURVEY.md §8(d): the reference ships no
data and no benchmark, so every BASELINE.json config is realised by this one
generator.  The definition is pure integer arithmetic so that the numpy version
here and the HIP kernel (`csrc/synth_kernel.hip`, reached through
`ta_synth_voronoi`) produce bit-identical volumes:

* the volume is cut into ``G0 x G1 x G2`` grid cells, ``G_d = max(1, round(D_d / pitch))``
  with ``pitch = (nvox / K) ** (1/3)``; grid cell ``i`` spans
  ``[floor(i*D_d/G_d), floor((i+1)*D_d/G_d))`` along axis ``d``;
* each grid cell owns one seed at ``origin + floor(u * extent)``, ``u`` drawn from
  ``numpy.random.default_rng(seed)`` (PCG64) -- seeds are always computed on the host;
* a voxel ``x`` lies in grid cell ``i_d = (x_d * G_d) // D_d`` and takes the label of the
  nearest seed (integer squared distance) among the grid cells within +-2 of ``i`` on
  every axis; ties go to the lowest label; label = linear grid-cell index + 2;
* voxels outside the centred ellipsoid with semi-axes ``0.45 * D_d`` get the background
  label 1 (test ``E0[x0] + E1[x1] + E2[x2] <= 2**24`` on per-axis integer tables);
* label 0 is never produced.
"""
from __future__ import annotations

import numpy as np

BACKGROUND = 1
ELL_ONE = 1 << 24


def grid_dims(dims, n_cells):
    """Number of grid cells per axis for ~n_cells seeds in a volume of shape dims."""
    dims = [int(d) for d in dims]
    nvox = dims[0] * dims[1] * dims[2]
    pitch = (nvox / float(max(1, int(n_cells)))) ** (1.0 / 3.0)
    return [max(1, int(round(d / pitch))) for d in dims]


def make_seeds(dims, n_cells, seed):
    """Seed coordinates, int32 array [G0*G1*G2, 3], and the grid dims."""
    dims = [int(d) for d in dims]
    G = grid_dims(dims, n_cells)
    rng = np.random.default_rng(int(seed))
    u = rng.random((G[0] * G[1] * G[2], 3))
    idx = np.indices(G).reshape(3, -1).T.astype(np.int64)  # [n,3] grid coordinates, C order
    pos = np.empty_like(idx)
    for d in range(3):
        lo = (idx[:, d] * dims[d]) // G[d]
        hi = ((idx[:, d] + 1) * dims[d]) // G[d]
        ext = np.maximum(hi - lo, 1)
        pos[:, d] = lo + np.floor(u[:, d] * ext).astype(np.int64)
        pos[:, d] = np.minimum(pos[:, d], dims[d] - 1)
    return pos.astype(np.int32), G


def ellipsoid_tables(dims):
    """Per-axis int64 tables E_d; a voxel is inside iff E0[x0]+E1[x1]+E2[x2] <= 2**24."""
    out = []
    for D in dims:
        D = int(D)
        R = max(1, (9 * D) // 10)  # semi-axis 0.45*D in half-voxel units
        q = 2 * np.arange(D, dtype=np.int64) + 1 - D
        out.append(((q * q) << 24) // (R * R))
    return out


def voronoi_labels(dims, n_cells, seed, dtype=np.uint16, a_begin=0, a_end=None,
                   ellipsoid=True):
    """numpy realisation of the generator for planes [a_begin, a_end) of axis 0.

    Meant for small volumes (tests, golden fixtures, the CPU-baseline sample);
    the device generator covers the large configs.
    """
    dims = [int(d) for d in dims]
    a_end = dims[0] if a_end is None else int(a_end)
    seeds, G = make_seeds(dims, n_cells, seed)
    n_seeds = seeds.shape[0]
    if n_seeds + 1 > np.iinfo(dtype).max:
        raise ValueError("dtype %s cannot hold %d labels" % (np.dtype(dtype), n_seeds + 1))
    S = seeds.reshape(G[0], G[1], G[2], 3).astype(np.int64)
    E = ellipsoid_tables(dims)
    out = np.empty((a_end - a_begin, dims[1], dims[2]), dtype=dtype)
    x1 = np.arange(dims[1], dtype=np.int64)[:, None]
    x2 = np.arange(dims[2], dtype=np.int64)[None, :]
    i1 = (x1 * G[1]) // dims[1]
    i2 = (x2 * G[2]) // dims[2]
    i1b = np.broadcast_to(i1, (dims[1], dims[2]))
    i2b = np.broadcast_to(i2, (dims[1], dims[2]))
    big = np.int64(1) << 62
    for a in range(a_begin, a_end):
        i0 = (a * G[0]) // dims[0]
        best = np.full((dims[1], dims[2]), big, dtype=np.int64)
        for o0 in range(-2, 3):
            j0 = i0 + o0
            if j0 < 0 or j0 >= G[0]:
                continue
            for o1 in range(-2, 3):
                j1 = i1b + o1
                ok1 = (j1 >= 0) & (j1 < G[1])
                j1c = np.clip(j1, 0, G[1] - 1)
                for o2 in range(-2, 3):
                    j2 = i2b + o2
                    ok = ok1 & (j2 >= 0) & (j2 < G[2])
                    j2c = np.clip(j2, 0, G[2] - 1)
                    s = S[j0, j1c, j2c]  # [D1, D2, 3]
                    d = (a - s[..., 0]) ** 2 + (x1 - s[..., 1]) ** 2 + (x2 - s[..., 2]) ** 2
                    lab = (j0 * G[1] + j1c) * G[2] + j2c + 2
                    key = np.where(ok, (d << 32) + lab, big)
                    best = np.minimum(best, key)
        lab = (best & 0xFFFFFFFF)
        if ellipsoid:
            inside = (E[0][a] + E[1][x1] + E[2][x2]) <= ELL_ONE
            lab = np.where(inside, lab, BACKGROUND)
        out[a - a_begin] = lab.astype(dtype)
    return out


# The BASELINE.json configs.  "features" are ta feature-mask names.
CONFIGS = {
    "C1": dict(dims=(128, 128, 64), dtype="uint16", n_cells=200, seed=0,
               features=("VOLUME", "BBOX", "MOMENT1", "MOMENT2", "ADJACENCY")),
    "C2": dict(dims=(512, 512, 512), dtype="uint16", n_cells=5000, seed=1,
               features=("VOLUME", "BBOX", "MOMENT1")),
    "C3": dict(dims=(1024, 1024, 1024), dtype="uint32", n_cells=50000, seed=2,
               features=("VOLUME", "BBOX", "MOMENT1", "ADJACENCY")),
    "C4": dict(dims=(1024, 1024, 1024), dtype="uint32", n_cells=50000, seed=2,
               features=("VOLUME", "BBOX", "MOMENT1", "MOMENT2", "ADJACENCY")),
    "C5": dict(dims=(2048, 2048, 2048), dtype="uint32", n_cells=100000, seed=3,
               features=("VOLUME", "BBOX", "MOMENT1", "MOMENT2", "ADJACENCY")),
}
PARITY_VOXELSIZE = (0.5, 0.5, 1.0)
