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


def closest_from_A(A, pts):
    p = np.asarray(pts, dtype=np.float64)
    d = ((p - np.asarray(A, dtype=np.float64)) ** 2).sum(axis=1)
    return tuple(int(v) for v in np.asarray(pts)[int(np.argmin(d))])
