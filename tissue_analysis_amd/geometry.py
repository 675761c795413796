"""Point-set helpers of the wall-median property (host side, float64).

``geometric_median`` follows the reference's own Weiszfeld iteration (spatial_image_analysis.py:1586-1635:
start at the centroid, shifted by 0.1 while it coincides with sample coordinates; stop when the sum of
squared distances changes by < 0.1 over two steps after the 4th iteration; ValueError after `numIter`
iterations), vectorised over the points.  ``closest_from_A`` is imported by the reference from
``openalea.image.algo.analysis`` (temporal_graph_from_image.py:222), which is not in the tree: it is taken
to be what its name and call site say -- the point of `pts` nearest to A (Euclidean), first one on ties.
"""
from __future__ import annotations

import warnings

import numpy as np


def geometric_median(X, numIter=200):
    X = np.asarray(X, dtype=np.float64)
    y = np.mean(X, 1)
    while (y[0] in X[0]) and (y[1] in X[1]) and (y[2] in X[2]):
        y = y + 0.1
    convergence = False
    dist = []
    i = 0
    while (not convergence) and (i < numIter):
        div = np.sqrt((X[0] - y[0]) ** 2 + (X[1] - y[1]) ** 2 + (X[2] - y[2]) ** 2)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / div
        denum = inv.sum()
        dist.append(float((div ** 2).sum()))
        if denum == 0.0:
            warnings.warn("Couldn't compute a geometric median, please check your data!")
            return [0, 0, 0]
        y = np.array([(X[0] * inv).sum() / denum, (X[1] * inv).sum() / denum, (X[2] * inv).sum() / denum])
        if i > 3:
            convergence = abs(dist[i] - dist[i - 2]) < 0.1
        i += 1
    if i == numIter:
        raise ValueError("The Weiszfeld's algoritm did not converged after" + str(numIter) + "iterations !!!!!!!!!")
    return np.array(y)


def _find_wall_median_voxel(array):
    """Index of the median voxel of a point set (SIA:1555-1585).  The reference delegates to PlantGL, which is not
    in its tree: `pointset_median` (<= 100 points) is the exact medoid -- the point with the smallest sum of
    Euclidean distances to all the others, first one on ties -- and is pinned by the reference's docstring example
    (SIA:1566-1570 -> 2).  `approx_pointset_median` (> 100 points) is an unspecified approximation of the same
    thing; here it is the exact medoid too, computed in blocks (parity unpinned for that branch)."""
    a = np.asarray(array, dtype=np.float64)
    if a.ndim != 2:
        raise ValueError("an (N, 3) or (3, N) array of coordinates is required")
    if a.shape[0] == 3:
        a = a.T                                     # like the reference: a 3-row array is read as 3 x N (SIA:1575-1576)
    n = a.shape[0]
    best, best_sum = 0, np.inf
    step = max(1, (1 << 22) // max(n, 1))           # ~32 MB of distances per block
    for i0 in range(0, n, step):
        d = np.sqrt(((a[i0:i0 + step, None, :] - a[None, :, :]) ** 2).sum(axis=2)).sum(axis=1)
        i = int(np.argmin(d))
        if d[i] < best_sum:
            best, best_sum = i0 + i, float(d[i])
    return best


def find_wall_median_voxel_index(array):
    return _find_wall_median_voxel(array)


def find_wall_median_voxel(dict_wall_voxels, labels2exclude=[], return_id=True, verbose=True):
    """The voxel closest to the geometrical median of each wall's voxel set (SIA:1499-1553): a dict keyed by label
    pairs gives a dict (or the bare value when it has one entry), an array gives one answer; `return_id` selects
    the index in the point set or the coordinates."""
    if isinstance(labels2exclude, (int, np.integer)):
        labels2exclude = [labels2exclude]
    if isinstance(dict_wall_voxels, dict):
        wall_median = {}
        for (label_1, label_2) in dict_wall_voxels:
            if label_1 in labels2exclude or label_2 in labels2exclude:
                continue
            xyz = np.array(dict_wall_voxels[(label_1, label_2)])
            if xyz.shape[0] == 3:
                xyz = xyz.T
            median_vox_id = _find_wall_median_voxel(xyz)
            wall_median[(label_1, label_2)] = median_vox_id if return_id else xyz[median_vox_id]
        if len(dict_wall_voxels) == 1:
            return list(wall_median.values())[0]
        return wall_median
    if isinstance(dict_wall_voxels, np.ndarray):
        xyz = dict_wall_voxels
        if xyz.shape[0] == 3:
            xyz = np.array(xyz).T
        median_vox_id = _find_wall_median_voxel(xyz)
        return median_vox_id if return_id else xyz[median_vox_id]
    return "Failed to recognise the type of data."


def closest_from_A(A, pts):
    p = np.asarray(pts, dtype=np.float64)
    d = ((p - np.asarray(A, dtype=np.float64)) ** 2).sum(axis=1)
    return tuple(int(v) for v in np.asarray(pts)[int(np.argmin(d))])
