"""Point-set reductions behind the wall-median properties, batched over MANY walls at once (host, float64).

The unit of work is a *grouped point table*: `points[n, 3]` with the points of wall k in rows
`start[k] : start[k] + size[k]` -- exactly what `WallTable` holds after the device grouped the wall-voxel records by
label pair.  Every routine below is a handful of segment reductions (`np.add.reduceat` & co) over that table; nothing
loops over walls or points in Python.

Semantics kept from the reference (spatial_image_analysis.py, cited SIA:<line>):
  * `geometric_median` (SIA:1586-1635) is a Weiszfeld iteration with three particular rules that decide its answer and
    are therefore kept: (1) it starts at the centroid, pushed by +0.1 on all axes for as long as each of its
    coordinates occurs among the samples' coordinates on the same axis; (2) from the fifth pass on it stops as soon as
    the sum of squared distances differs by less than 0.1 from its value two passes earlier -- so it stops well short
    of the true median; (3) a wall that is still moving after `numIter` passes is an error.
  * `find_wall_median_voxel` (SIA:1499-1585) delegates to PlantGL (absent); `pointset_median` is the exact medoid,
    pinned by the docstring answer SIA:1566-1570.
  * `closest_from_A` comes from `openalea.image.algo.analysis` (absent; temporal_graph_from_image.py:222): the sample
    nearest to a position, first one on ties.
"""
from __future__ import annotations

import warnings

import numpy as np


# ----------------------------------------------------------------------------- grouped point tables
def _segments(sizes):
    """(start, owner) of consecutive segments with the given sizes."""
    sizes = np.asarray(sizes, dtype=np.int64)
    start = np.zeros(sizes.size, dtype=np.int64)
    if sizes.size > 1:
        np.cumsum(sizes[:-1], out=start[1:])
    return start, np.repeat(np.arange(sizes.size, dtype=np.int64), sizes)


def gather_segments(start, stop):
    """Row indices that concatenate the ranges [start[k], stop[k]) -- one arange, no Python loop."""
    start = np.asarray(start, dtype=np.int64)
    size = np.asarray(stop, dtype=np.int64) - start
    new_start, owner = _segments(size)
    return np.arange(int(size.sum()), dtype=np.int64) + (start - new_start)[owner], size


def _centroids(P, start, sizes):
    """`np.mean(X, 1)` of every segment's 3 x n coordinate array, bit for bit.  numpy sums a contiguous axis pairwise, in an
    order that depends only on n: segments of equal size are stacked as [count, 3, n] and reduced over the last axis."""
    out = np.empty((sizes.size, 3), dtype=np.float64)
    order = np.argsort(sizes, kind="stable")
    ranked = sizes[order]
    cuts = np.flatnonzero(np.diff(ranked)) + 1
    for group in np.split(order, cuts):
        n = int(sizes[group[0]])
        block = P[start[group][:, None] + np.arange(n, dtype=np.int64)[None, :]]          # [count, n, 3]
        out[group] = np.ascontiguousarray(block.transpose(0, 2, 1)).mean(axis=2)
    return out


class _InOrderSums(object):
    """Per-segment sums taken strictly in row order, ((t0 + t1) + t2) + ..., for all segments at once: step j adds row j of
    every segment that has one.  (`np.add.reduceat` sums pairwise; the reference accumulates point by point, and the
    truncation that follows the iteration makes the last bit count.)"""

    def __init__(self, start, sizes):
        self.order = np.argsort(-sizes, kind="stable")                     # longest first: the active segments are a prefix
        self.first = start[self.order]
        ranked = sizes[self.order]
        longest = int(ranked[0]) if ranked.size else 0
        self.active = ranked.size - np.searchsorted(ranked[::-1], np.arange(longest), side="right")

    def __call__(self, terms):
        acc = np.zeros((self.order.size,) + terms.shape[1:], dtype=np.float64)
        for j, a in enumerate(self.active.tolist()):
            acc[:a] += terms[self.first[:a] + j]
        out = np.empty_like(acc)
        out[self.order] = acc
        return out


def weiszfeld_segments(points, sizes, max_iter=200):
    """Geometric median, by the reference's rules AND in its arithmetic, of every segment of a grouped point table.

    points: [n, 3] (any numeric dtype), segment k = `sizes[k]` consecutive rows, every size >= 1.
    Returns float64 [m, 3]: SIA:1586-1635 run on each segment with the same operations in the same order (the sums point by
    point, see _InOrderSums), except that squares are x * x where the reference ends up in libm's pow(x, 2) -- which is not
    correctly rounded, so the reference itself is only reproducible to the last bits there.  Walls whose terms are exact
    (flat, symmetric ones: the walls whose median sits ON an integer, where the truncation that follows is decided by the
    last bit) come out with identical doubles.  Rows of segments whose weights vanished are (0, 0, 0), as the reference
    answers there.  Raises ValueError when a segment is still moving after `max_iter` passes."""
    P = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    sizes = np.asarray(sizes, dtype=np.int64)
    m = sizes.size
    out = np.zeros((m, 3), dtype=np.float64)
    if m == 0:
        return out
    if (sizes <= 0).any() or int(sizes.sum()) != P.shape[0]:
        raise ValueError("segments must be non-empty and cover the point table")
    start, owner = _segments(sizes)
    y = _centroids(P, start, sizes)
    # rule (1): the centroid is nudged while, on EVERY axis, its coordinate is some sample's coordinate on that axis
    while True:
        on_axis = np.logical_or.reduceat(P == y[owner], start, axis=0)
        nudge = on_axis.all(axis=1)
        if not nudge.any():
            break
        y[nudge] += 0.1
    # The passes run on the walls that have not stopped yet: a wall that stops leaves the working table.
    live = np.arange(m, dtype=np.int64)              # ids of the segments still iterating
    cost_1 = np.zeros(m)                             # sum of squared distances one and two passes ago
    cost_2 = np.zeros(m)
    sums = _InOrderSums(start, sizes)
    terms = np.empty((P.shape[0], 5), dtype=np.float64)
    for it in range(max_iter):
        d = P - y[owner]
        dist = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
        with np.errstate(divide="ignore", invalid="ignore"):
            terms[:, :3] = P / dist[:, None]                                 # x / d, y / d, z / d
            terms[:, 3] = 1.0 / dist
        terms[:, 4] = dist * dist
        total = sums(terms)
        wsum, cost = total[:, 3], total[:, 4]
        with np.errstate(divide="ignore", invalid="ignore"):
            y_new = total[:, :3] / wsum[:, None]
        dead = wsum == 0.0
        if dead.any():
            warnings.warn("no geometric median for %d point set(s): every point is infinitely far" % int(dead.sum()))
            y_new[dead] = 0.0
        y = y_new
        stopped = dead | ((it > 3) & (np.abs(cost - cost_2) < 0.1))          # rule (2): `cost_2` is the value two passes back
        if it == max_iter - 1:
            stopped = dead                           # rule (3) as written: settling on the very last pass still counts as failure
        cost_2, cost_1 = cost_1, cost
        if stopped.any():
            out[live[stopped]] = y[stopped]
            keep = ~stopped
            if not keep.any():
                return out
            rows = keep[owner]
            P, sizes, y = P[rows], sizes[keep], y[keep]
            cost_1, cost_2, live = cost_1[keep], cost_2[keep], live[keep]
            start, owner = _segments(sizes)
            sums = _InOrderSums(start, sizes)
            terms = np.empty((P.shape[0], 5), dtype=np.float64)
    raise ValueError("Weiszfeld iteration: %d point set(s) still moving after %d passes" % (live.size, max_iter))


def nearest_in_segments(points, sizes, targets):
    """Row index (into `points`) of the sample of each segment that is nearest to that segment's target position;
    the first such row on ties.  points [n, 3], targets [m, 3]."""
    P = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    sizes = np.asarray(sizes, dtype=np.int64)
    if sizes.size == 0:
        return np.zeros(0, dtype=np.int64)
    start, owner = _segments(sizes)
    delta = P - np.asarray(targets, dtype=np.float64)[owner]
    sq = np.einsum("ij,ij->i", delta, delta)
    sq[np.isnan(sq)] = np.inf                                      # (a target without a position: the segment's first row)
    best = np.minimum.reduceat(sq, start)
    hits = np.flatnonzero(sq == best[owner])                      # ascending rows: the first hit of a segment is its answer
    return hits[np.searchsorted(hits, start)]


def median_voxels(points, sizes, max_iter=200):
    """The wall-median rule of `_graph_from_image` (temporal_graph_from_image.py:210-224) for every segment at once: the
    Weiszfeld position, truncated to integers, then the sample nearest to it.  Returns the chosen samples [m, 3]."""
    pts = np.asarray(points).reshape(-1, 3)
    origin = np.trunc(weiszfeld_segments(pts, sizes, max_iter))
    return pts[nearest_in_segments(pts, sizes, origin)]


def medoid_index(points):
    """Index of the sample with the smallest sum of Euclidean distances to all the others (first one on ties)."""
    a = np.asarray(points, dtype=np.float64)
    n = a.shape[0]
    block = max(1, (1 << 22) // max(n, 1))                         # ~32 MB of distances at a time
    best, best_total = 0, np.inf
    for lo in range(0, n, block):
        diff = a[lo:lo + block, None, :] - a[None, :, :]
        total = np.sqrt(np.einsum("ijk,ijk->ij", diff, diff)).sum(axis=1)
        k = int(np.argmin(total))
        if total[k] < best_total:
            best, best_total = lo + k, float(total[k])
    return best


# ----------------------------------------------------------------------------- the reference's module-level names
def geometric_median(X, numIter=200):
    """SIA:1586-1635 for one 3 x N coordinate array (one segment of the batched routine)."""
    X = np.asarray(X, dtype=np.float64)
    return weiszfeld_segments(X.T, [X.shape[1]], numIter)[0]


def _as_rows(array):
    a = np.asarray(array)
    if a.ndim != 2:
        raise ValueError("an (N, 3) or (3, N) array of coordinates is required")
    return a.T if a.shape[0] == 3 else a                           # a 3-row array is read as 3 x N (SIA:1575-1576)


def _find_wall_median_voxel(array):
    """SIA:1555-1585: index of the median voxel of a point set.  PlantGL's `pointset_median` (<= 100 points) is the exact
    medoid; `approx_pointset_median` (> 100 points) is an unspecified approximation of it and is the exact medoid here
    too (parity unpinned for that branch)."""
    return medoid_index(_as_rows(array))


find_wall_median_voxel_index = _find_wall_median_voxel


def find_wall_median_voxel(dict_wall_voxels, labels2exclude=[], return_id=True, verbose=True):
    """SIA:1499-1553: a dict keyed by label pairs gives a dict (the bare value when it has ONE entry), an array gives one
    answer; `return_id` selects the index in the point set or the coordinates."""
    if isinstance(labels2exclude, (int, np.integer)):
        labels2exclude = [labels2exclude]
    if isinstance(dict_wall_voxels, np.ndarray):
        rows = _as_rows(dict_wall_voxels)
        k = medoid_index(rows)
        return k if return_id else rows[k]
    if not isinstance(dict_wall_voxels, dict):
        return "Failed to recognise the type of data."
    skip = set(labels2exclude)
    answer = {}
    for pair, voxels in dict_wall_voxels.items():
        if skip.isdisjoint(pair):
            rows = _as_rows(np.array(voxels))
            k = medoid_index(rows)
            answer[pair] = k if return_id else rows[k]
    if len(dict_wall_voxels) == 1:
        return next(iter(answer.values()))
    return answer


def closest_from_A(A, pts):
    pts = np.asarray(pts)
    k = nearest_in_segments(pts, [pts.shape[0]], np.asarray(A, dtype=np.float64)[None, :])[0]
    return tuple(int(v) for v in pts[k])
