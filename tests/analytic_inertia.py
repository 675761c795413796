"""Label shapes whose inertia has a closed form -- a pin for SpatialImageAnalysis.inertia_axis (SIA:1246-1292, normaliser
SIA:137-150) that owes nothing to any restatement of the reference:

  * an a x b x c cuboid of N = abc >= 3 voxels: covariance diag((a^2-1)/12, (b^2-1)/12, (c^2-1)/12) (the variance of a
    integers in a row), eigenvectors the coordinate axes, eigenvalues sorted decreasing;
  * one voxel: every eigenvalue 0;   two voxels side by side along axis d: sum of squares 1/2, normaliser max(3, N) = 3
    (SIA:150) -> eigenvalue 1/6 on axis d, 0 twice;
  * n voxels on the diagonal (k, k, k): covariance (n^2-1)/12 * ones(3, 3): eigenvalue 3 (n^2-1)/12 along (1,1,1)/sqrt(3);
  * real=True multiplies eigenvalue i by |v_i * voxelsize| (SIA:1282-1284).
"""
import numpy as np

VOXELSIZE = (0.5, 0.5, 1.0)
BACKGROUND = 1


def build(dtype=np.uint16):
    """(volume, cases): cases[label] = dict(values=[3 eigenvalues, decreasing, voxel units], axes=[3 unit vectors or None
    where the eigenvalue is repeated], real=[3 eigenvalues in real units])."""
    vol = np.full((40, 48, 64), BACKGROUND, dtype=dtype)
    vs = np.asarray(VOXELSIZE)
    cases = {}

    def cuboid(label, origin, size):
        o, n = np.asarray(origin), np.asarray(size)
        vol[o[0]:o[0] + n[0], o[1]:o[1] + n[1], o[2]:o[2] + n[2]] = label
        var = (n.astype(np.float64) ** 2 - 1.0) / 12.0
        if n.prod() < 3:
            var = var * n.prod() / 3.0                                  # SIA:150: 1 / max(3, N)
        order = np.argsort(-var, kind="stable")
        values = var[order]
        axes = [np.eye(3)[d] for d in order]
        real = [values[i] * vs[order[i]] for i in range(3)]
        distinct = [np.sum(np.isclose(values, v)) == 1 for v in values]
        cases[label] = dict(values=values, axes=[a if ok else None for a, ok in zip(axes, distinct)], real=np.asarray(real),
                            count=int(n.prod()), distinct=distinct)

    cuboid(2, (1, 1, 1), (3, 5, 9))
    cuboid(3, (6, 2, 20), (7, 4, 2))
    cuboid(4, (20, 20, 3), (2, 11, 30))
    cuboid(5, (30, 40, 50), (1, 1, 1))                                  # one voxel
    cuboid(6, (30, 44, 50), (1, 1, 2))                                  # two voxels along axis 2
    cuboid(7, (33, 40, 50), (2, 1, 1))                                  # two voxels along axis 0
    cuboid(8, (10, 30, 40), (6, 6, 6))                                  # a cube: one triple eigenvalue
    cuboid(9, (0, 0, 30), (4, 9, 34))                                   # touches three faces of the volume
    n = 12                                                              # the diagonal line
    for k in range(n):
        vol[20 + k, 2 + k, 40 + k] = 10
    lam = 3.0 * (n * n - 1.0) / 12.0
    d = np.ones(3) / np.sqrt(3.0)
    cases[10] = dict(values=np.array([lam, 0.0, 0.0]), axes=[d, None, None], distinct=[True, False, False],
                     real=np.array([lam * np.linalg.norm(d * vs), np.nan, np.nan]), count=n)
    return vol, cases


def check(analysis, cases, real):
    """Compare `analysis.inertia_axis(labels, real)` (this package's class or the oracle's) with the closed forms."""
    labels = sorted(cases)
    axes, values = analysis.inertia_axis(list(labels), real)
    for l in labels:
        c = cases[l]
        got_v, got_a = np.asarray(values[l], dtype=np.float64), np.asarray(axes[l], dtype=np.float64)
        want = c["real"] if real else c["values"]
        for i in range(3):
            if c["distinct"][i]:
                assert abs(got_v[i] - want[i]) <= 1e-9 * max(1.0, abs(want[i])), (l, i, got_v, want)
                assert abs(abs(np.dot(got_a[i], c["axes"][i])) - 1.0) <= 1e-9, (l, i, got_a[i], c["axes"][i])
            elif real and np.isnan(want[i]):
                # a repeated eigenvalue 0 of the line: whatever vector, 0 times its scaled norm is 0
                assert abs(got_v[i]) <= 1e-9, (l, i, got_v)
            elif not real:
                assert abs(got_v[i] - want[i]) <= 1e-9 * max(1.0, abs(want[i])), (l, i, got_v, want)
            else:
                # repeated eigenvalue, real units: lambda * |v * voxelsize| with v anywhere in the eigenspace
                lo, hi = min(VOXELSIZE), max(VOXELSIZE)
                lam = c["values"][i]
                assert lam * lo - 1e-9 <= got_v[i] <= lam * hi + 1e-9, (l, i, got_v, lam)
        assert abs(np.linalg.norm(got_a, axis=1) - 1.0).max() <= 1e-9
