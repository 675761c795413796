"""Host side of the hot path: run the fused sweep through the C ABI and turn its exact integer
accumulators into the float quantities the reference computes per label.

Everything here is float64 numpy on *integers produced by the GPU*; there is no CPU path that
scans voxels.  Formulas (SURVEY.md §8 "Semantics"):
  barycentre  com  = sum1 / count                                   SIA:466-467
  covariance  cov  = (sum2 - sum1.sum1^T / N) / max(3, N)           SIA:137-150, 1276-1278
              evaluated on moments first shifted to the label's bounding-box origin in exact
              integer arithmetic, which removes the cancellation of the raw global form
  inertia     eigenpairs of cov, decreasing eigenvalue, vectors as rows          SIA:152-167
  wall area   F0*v1*v2 + F1*v2*v0 + F2*v0*v1 (real) or F0+F1+F2 (voxels)  SIA:751-756, 947-956
"""
from __future__ import annotations

from collections.abc import Mapping

import numpy as np

from . import _capi

PAIR_ORDER = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))


def sym3_eig(matrices, sweeps=12):
    """Eigen-decomposition of MANY symmetric 3x3 matrices at once: cyclic Jacobi, every rotation a handful of array
    operations over all matrices.  Returns (values [n, 3] unordered, vectors [n, 3, 3] with eigenvectors as COLUMNS).
    Jacobi rotations are exactly orthogonal transforms, so eigenvalues are good to a few ulps of the matrix norm and the
    vectors orthonormal to rounding; matrices that are already diagonal are left untouched (identity vectors)."""
    a = np.array(matrices, dtype=np.float64)
    n = a.shape[0]
    d = [a[:, 0, 0].copy(), a[:, 1, 1].copy(), a[:, 2, 2].copy()]
    off = {(0, 1): a[:, 0, 1].copy(), (0, 2): a[:, 0, 2].copy(), (1, 2): a[:, 1, 2].copy()}
    v = [[np.ones(n) if r == c else np.zeros(n) for c in range(3)] for r in range(3)]       # v[r][c]
    scale = np.abs(a).reshape(n, 9).max(axis=1) if n else np.zeros(0)
    tiny = np.finfo(np.float64).tiny
    for _ in range(sweeps):
        if not n or not (np.abs(off[(0, 1)]) + np.abs(off[(0, 2)]) + np.abs(off[(1, 2)]) > 1e-300 + 4e-16 * scale).any():
            break
        for p, q in ((0, 1), (0, 2), (1, 2)):
            r = 3 - p - q
            apq = off[(p, q)]
            live = np.abs(apq) > tiny
            theta = (d[q] - d[p]) / np.where(live, 2.0 * apq, 1.0)
            t = np.where(theta >= 0.0, 1.0, -1.0) / (np.abs(theta) + np.hypot(theta, 1.0))
            t = np.where(live, t, 0.0)
            c = 1.0 / np.sqrt(t * t + 1.0)
            sn = t * c
            d[p] = d[p] - t * apq
            d[q] = d[q] + t * apq
            off[(p, q)] = np.zeros(n)
            kp, kq = (min(r, p), max(r, p)), (min(r, q), max(r, q))
            arp, arq = off[kp], off[kq]
            off[kp], off[kq] = c * arp - sn * arq, sn * arp + c * arq
            for row in range(3):
                vp, vq = v[row][p], v[row][q]
                v[row][p], v[row][q] = c * vp - sn * vq, sn * vp + c * vq
    values = np.stack(d, axis=1)
    vectors = np.stack([np.stack(v[row], axis=1) for row in range(3)], axis=1)
    return values, vectors


class NeighborRows(Mapping):
    """label -> ascending list of its face neighbours, as a read-only mapping over the CSR rows of the adjacency.  Nothing is
    built until it is used: `indptr` / `indices` are the arrays (made on first access), the lists are made per lookup;
    `keys_array` holds the labels it answers for."""

    def __init__(self, keys, extraction):
        self.keys_array = np.asarray(keys, dtype=np.int64)
        self._x = extraction
        self._known = None

    @property
    def indptr(self):
        return self._x._adjacency_csr()[0]

    @property
    def indices(self):
        return self._x._adjacency_csr()[1]

    def _has(self, label):
        if self._known is None:
            self._known = set(self.keys_array.tolist())
        return label in self._known

    def __getitem__(self, label):
        if not self._has(label):
            raise KeyError(label)
        ptr = self.indptr
        row = self._x.row_of(label)
        if 0 <= row < ptr.size - 1:
            return self.indices[ptr[row]:ptr[row + 1]].tolist()
        return []

    def __contains__(self, label):
        return self._has(label)

    def __iter__(self):
        return iter(self.keys_array.tolist())

    def __len__(self):
        return int(self.keys_array.size)


class BoxesByLabel(object):
    """`nd.find_objects(image)` of an image whose ids are sparse: entry k is the box of label k + 1 (None when the image
    does not hold it), made when it is looked up -- the list itself would have max_label entries."""

    def __init__(self, extraction):
        self._x = extraction

    def __len__(self):
        return self._x.max_label

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        k = int(k)
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError("list index out of range")
        return self._x.bbox_slices(k + 1)

    def __iter__(self):
        return (self[k] for k in range(len(self)))


class Extraction(object):
    """Exact integer result of one sweep + adjacency COO sorted by (lo, hi).  Rows 0..max_label, one per label id -- or, for
    a volume whose ids are SPARSE (`ids` given: the ascending ids the volume holds, SIA:358-364 np.unique takes any), one row
    per id in that order; every method takes and answers label IDS either way, `rows_of` is the only place that knows."""

    def __init__(self, shape, max_label, count, bbox, sum1, sum2, pair_lo, pair_hi, pair_faces,
                 timing=None, ids=None):
        self.shape = tuple(int(s) for s in shape)
        self.ids = None if ids is None else np.asarray(ids, dtype=np.int64)
        if self.ids is not None:
            max_label = int(self.ids[-1]) if self.ids.size else 0
        self.max_label = int(max_label)
        self.count = np.asarray(count, dtype=np.uint64)
        self.bbox = np.asarray(bbox, dtype=np.int32).reshape(-1, 6)
        self.sum1 = np.asarray(sum1, dtype=np.uint64).reshape(-1, 3)
        self.sum2 = np.asarray(sum2, dtype=np.uint64).reshape(-1, 6)
        self.pair_lo = np.asarray(pair_lo, dtype=np.uint32)
        self.pair_hi = np.asarray(pair_hi, dtype=np.uint32)
        self.pair_faces = np.asarray(pair_faces, dtype=np.uint64).reshape(-1, 3)
        self.timing = timing
        self.nrows = int(self.count.shape[0])
        if self.ids is not None and self.ids.size != self.nrows:
            raise ValueError("%d ids for %d rows" % (self.ids.size, self.nrows))
        self._csr = None
        self._derived = {}

    @classmethod
    def from_arrays(cls, shape, arrays, timing=None):
        """Build from a dict with the C-ABI array names (what tests / distributed merges hold)."""
        return cls(shape, arrays["max_label"], arrays["count"], arrays["bbox"], arrays["sum1"],
                   arrays["sum2"], arrays["pair_lo"], arrays["pair_hi"], arrays["pair_faces"], timing, arrays.get("ids"))

    def as_arrays(self):
        out = dict(max_label=self.max_label, count=self.count, bbox=self.bbox, sum1=self.sum1,
                   sum2=self.sum2, pair_lo=self.pair_lo, pair_hi=self.pair_hi,
                   pair_faces=self.pair_faces)
        if self.ids is not None:
            out["ids"] = self.ids
        return out

    # ------------------------------------------------------------------ label id -> row
    @property
    def sparse(self):
        return self.ids is not None

    def rows_of(self, labels, missing=None):
        """int64 row of every label id.  An id without a row: IndexError (like indexing the dense rows beyond max_label), or
        `missing` when that is given (callers that answer zeros for unknown labels pass the index of a zero row)."""
        idx = np.asarray(labels, dtype=np.int64)
        if self.ids is None:
            if missing is None:
                return idx
            return np.where((idx >= 0) & (idx < self.nrows), idx, missing)
        if self.ids.size == 0:
            pos = np.zeros(idx.shape, dtype=np.int64)
            ok = np.zeros(idx.shape, dtype=bool)
        else:
            pos = np.minimum(np.searchsorted(self.ids, idx), self.ids.size - 1)
            ok = self.ids[pos] == idx
        if missing is None:
            if not ok.all():
                raise IndexError("label ids %s are not in the image" % np.asarray(idx)[~ok][:5].tolist())
            return pos
        return np.where(ok, pos, missing)

    def row_of(self, label):
        """Row of ONE label id, -1 when it has none."""
        label = int(label)
        if self.ids is None:
            return label if 0 <= label < self.nrows else -1
        k = int(np.searchsorted(self.ids, label))
        return k if k < self.ids.size and self.ids[k] == label else -1

    def labels_of(self, rows):
        """The label ids of rows (the inverse of rows_of)."""
        rows = np.asarray(rows, dtype=np.int64)
        return rows if self.ids is None else self.ids[rows]

    def _cached(self, name, make):
        if name not in self._derived:
            self._derived[name] = make()
        return self._derived[name]

    # ------------------------------------------------------------------ labels / boxes
    def present(self):
        """Ascending ids of the labels that own at least one voxel."""
        return self._cached("present", lambda: self.labels_of(np.nonzero(self.count)[0]))

    @property
    def lo(self):
        """The pair list as int64 index arrays / float64 face counts, converted once."""
        return self._cached("lo", lambda: self.pair_lo.astype(np.int64))

    @property
    def hi(self):
        return self._cached("hi", lambda: self.pair_hi.astype(np.int64))

    @property
    def faces(self):
        return self._cached("faces", lambda: self.pair_faces.astype(np.float64))

    def pair_areas(self, face_surface=None):
        """float64 [P]: F0 s0 + F1 s1 + F2 s2 of every pair (SIA:751-756, 953), or F0 + F1 + F2 without face areas.  One
        expression for every caller, so that a threshold on an area decides the same way everywhere."""
        f = self.faces
        if face_surface is None:
            return self._cached("area_voxels", lambda: f[:, 0] + f[:, 1] + f[:, 2])
        s = tuple(float(v) for v in face_surface)
        return self._cached(("area", s), lambda: f[:, 0] * s[0] + f[:, 1] * s[1] + f[:, 2] * s[2])

    @property
    def lo_rows(self):
        """The rows of the pair list's labels (the ids themselves unless the ids are sparse)."""
        return self.lo if self.ids is None else self._cached("lo_rows", lambda: self.rows_of(self.lo))

    @property
    def hi_rows(self):
        return self.hi if self.ids is None else self._cached("hi_rows", lambda: self.rows_of(self.hi))

    def _row_degrees(self):
        n = max(self.nrows, self.max_label + 1 if self.ids is None else 0) + 1
        return self._cached("degree", lambda: np.bincount(self.lo_rows, minlength=n) + np.bincount(self.hi_rows, minlength=n))

    def degrees(self):
        """int64 [max_label + 2]: how many labels share a face with each label (dense ids only: see degrees_of)."""
        if self.ids is not None:
            raise TypeError("degrees() is indexed by label id: use degrees_of(labels) when the ids are sparse")
        return self._row_degrees()

    def degrees_of(self, labels):
        """int64 [n]: how many labels share a face with each of `labels` (0 for an id the image does not hold)."""
        deg = self._row_degrees()
        return deg[self.rows_of(labels, missing=deg.size - 1)]

    def has(self, label):
        row = self.row_of(label)
        return row >= 0 and self.count[row] > 0

    def bbox_slices(self, label):
        """(slice, slice, slice) like nd.find_objects, or None when the label is absent."""
        row = self.row_of(label)
        if row < 0 or not self.count[row] > 0:
            return None
        b = self.bbox[row]
        return tuple(slice(int(b[d]), int(b[3 + d])) for d in range(3))

    def bbox_slices_upto(self, top):
        """[bbox_slices(1), ..., bbox_slices(top)] in one pass (python ints from two .tolist() calls); sparse ids: the same
        sequence made on demand."""
        if self.ids is not None:
            return BoxesByLabel(self)
        top = min(int(top), self.max_label)
        rows = self.bbox[1:top + 1].tolist()
        have = (self.count[1:top + 1] > 0).tolist()
        return [(slice(b[0], b[3]), slice(b[1], b[4]), slice(b[2], b[5])) if h else None for b, h in zip(rows, have)]

    def neighbor_rows(self, labels):
        """The same answer as `neighbor_lists` without making the lists: a mapping view over the CSR arrays."""
        return NeighborRows(labels, self)

    def neighbor_lists(self, labels):
        """{label: ascending list of its face neighbours} for many labels at once."""
        ptr, dst, _ = self._adjacency_csr()
        flat, p = self._cached("csr_lists", lambda: (dst.tolist(), ptr.tolist()))
        n = len(p) - 1
        if self.ids is None:
            return dict((l, flat[p[l]:p[l + 1]] if 0 <= l < n else []) for l in labels)
        labels = list(labels)
        rows = self.rows_of(labels, missing=-1).tolist() if labels else []
        return dict((l, flat[p[r]:p[r + 1]] if 0 <= r < n else []) for l, r in zip(labels, rows))

    # ------------------------------------------------------------------ moments
    def volumes(self, labels):
        return self.count[self.rows_of(labels)].astype(np.float64)

    def barycenters(self, labels):
        """float64 [n, 3], voxel units."""
        idx = self.rows_of(labels)
        n = self.count[idx].astype(np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            return self.sum1[idx].astype(np.float64) / n[:, None]

    def covariances(self, labels):
        """float64 [n, 3, 3] = sum (p - com)(p - com)^T / max(3, N)."""
        idx = self.rows_of(labels)
        N = self.count[idx].astype(np.int64)
        o = np.where(self.bbox[idx, :3] < 0, 0, self.bbox[idx, :3]).astype(np.int64)   # bbox origin
        s1 = self.sum1[idx].astype(np.int64)
        s2 = self.sum2[idx].astype(np.int64)
        s1s = s1 - N[:, None] * o                                                       # exact
        cov = np.zeros((idx.size, 3, 3), dtype=np.float64)
        Nf = N.astype(np.float64)
        norm = np.maximum(3.0, Nf)
        with np.errstate(invalid="ignore", divide="ignore"):
            for k, (d, e) in enumerate(PAIR_ORDER):
                s2s = s2[:, k] - o[:, d] * s1[:, e] - o[:, e] * s1[:, d] + N * o[:, d] * o[:, e]
                c = (s2s.astype(np.float64) - s1s[:, d].astype(np.float64) * s1s[:, e].astype(np.float64) / Nf) / norm
                cov[:, d, e] = c
                cov[:, e, d] = c
        return cov

    def inertia(self, labels):
        """(vectors [n,3,3] rows = axes, values [n,3]) sorted by decreasing eigenvalue, like eigen_values_vectors
        (SIA:152-167).  The reference calls np.linalg.eig per label (LAPACK geev, one Python call each); here all the
        symmetric 3x3 covariances are diagonalised together by cyclic Jacobi rotations (`sym3_eig`): eigenvalues to
        ~1e-16 of the matrix norm, eigenvectors up to sign -- like any eigen-solver."""
        cov = self.covariances(labels)
        if cov.shape[0] == 0:
            return np.zeros((0, 3, 3)), np.zeros((0, 3))
        val, vec = sym3_eig(np.where(np.isfinite(cov), cov, 0.0))
        order = np.argsort(-val, axis=1, kind="stable")
        val = np.take_along_axis(val, order, axis=1)
        vec = np.take_along_axis(vec, order[:, None, :], axis=2)      # columns re-ordered
        return np.transpose(vec, (0, 2, 1)), val                      # rows = eigenvectors

    # ------------------------------------------------------------------ adjacency
    def _adjacency_csr(self):
        """(ptr [rows + 1], dst, pid): the pair list as CSR over the ROWS; dst holds label ids, ascending inside a row."""
        if self._csr is None:
            lo, hi = self.lo_rows, self.hi_rows
            src = np.concatenate([lo, hi])
            dst = np.concatenate([hi, lo])            # (rows: the order of the rows is the order of the ids)
            pid = np.concatenate([np.arange(lo.size), np.arange(lo.size)])
            nrows = max(self.nrows if self.ids is not None else self.max_label + 1, int(src.max()) + 1 if src.size else 0)
            order = np.argsort(src * np.int64(nrows) + dst, kind="stable")
            ptr = np.zeros(nrows + 1, dtype=np.int64)
            np.cumsum(np.bincount(src, minlength=nrows), out=ptr[1:])
            self._csr = (ptr, self.labels_of(dst[order]), pid[order])
        return self._csr

    def neighbors_of(self, label):
        """Ascending int list of the labels sharing at least one voxel face with `label`."""
        ptr, dst, _ = self._adjacency_csr()
        row = self.row_of(label) if self.ids is not None else int(label)
        if row < 0 or row + 1 >= ptr.size:
            return []
        return [int(v) for v in dst[ptr[row]:ptr[row + 1]]]

    def surface_faces(self, labels):
        """uint64 [n, 3]: per-axis number of voxel faces each label shares with ANY other label (faces on the border
        of the volume belong to no wall and are not counted): the row sums of the adjacency."""
        top = self.nrows if self.ids is not None else self.max_label + 1                   # (index of the zero row)

        def rows():
            n = top + 1
            return np.stack([(np.bincount(self.lo_rows, weights=self.faces[:, d], minlength=n)
                              + np.bincount(self.hi_rows, weights=self.faces[:, d], minlength=n)) for d in range(3)],
                            axis=1).astype(np.uint64)            # (float64 sums of integers far below 2^53: exact)
        out = self._cached("surface_faces", rows)
        if self.ids is not None:
            return out[self.rows_of(labels, missing=top)]
        idx = np.asarray(labels, dtype=np.int64)
        idx = np.where((idx >= 0) & (idx <= self.max_label), idx, top)                    # unknown labels: the zero row
        return out[idx]

    def faces_between(self, label, others):
        """uint64 [len(others), 3]: per-axis shared-face counts of (label, other); 0 when not adjacent."""
        ptr, dst, pid = self._adjacency_csr()
        out = np.zeros((len(others), 3), dtype=np.uint64)
        r = self.row_of(label) if self.ids is not None else int(label)
        if r < 0 or r + 1 >= ptr.size:
            return out
        row = dst[ptr[r]:ptr[r + 1]]
        rid = pid[ptr[r]:ptr[r + 1]]
        pos = np.searchsorted(row, np.asarray(others, dtype=np.int64))
        for i, (p, o) in enumerate(zip(pos, others)):
            if p < row.size and row[p] == o:
                out[i] = self.pair_faces[rid[p]]
        return out


def full_mask():
    return _capi.F_ALL


SPARSE_FROM = 1 << 18          # ids below this cost at most 27 MB of rows: not worth a census
DENSE_ROW_LIMIT = 1 << 28      # ta_extract's own bound on max_label


def wants_compaction(max_label, n_present):
    """Sparse ids: more than 8 ids of room per label present (and enough rows for that to matter), or beyond the dense rows."""
    return max_label >= DENSE_ROW_LIMIT or (max_label >= SPARSE_FROM and max_label + 1 > 8 * max(int(n_present), 1))


def extract_resident(ctx, shape, features=_capi.F_ALL, max_label=None, sparse=None):
    """Run the sweep on the volume already resident in `ctx` and fetch the result.  sparse: None = decide from the volume
    (`wants_compaction`: one max-label pass, and a census of the ids when that is large), True / False = as told; an explicit
    `max_label` means dense rows 0..max_label, as before.  A context left compacted by an earlier call stays compacted when
    nothing else is asked for; `sparse=False` or an explicit `max_label` leave the compacted state first (dense rows it is)."""
    ids = None
    if ctx.is_compact() and (sparse is False or max_label is not None):
        ctx.uncompact()
    if ctx.is_compact():
        ids = ctx.compact_ids()
    elif max_label is None and sparse is not False:
        top = ctx.max_label()
        if sparse or top >= SPARSE_FROM:
            top, present = ctx.label_census()
            if sparse or wants_compaction(top, present.size):
                ids = ctx.compact_labels()
        max_label = top
    elif max_label is None:
        max_label = ctx.max_label()
    if ids is not None:
        max_label = max(int(ids.size) - 1, 0)
    ctx.extract(features, max_label)
    count, bbox, sum1, sum2 = ctx.labels()
    if _capi.feature_mask(features) & _capi.F_ADJACENCY:
        lo, hi, faces = ctx.adjacency()
    else:
        lo = hi = np.zeros(0, dtype=np.uint32)
        faces = np.zeros((0, 3), dtype=np.uint64)
    return Extraction(shape, max_label, count, bbox, sum1, sum2, lo, hi, faces, ctx.timing(), ids=ids)


def _as_label_volume(array):
    """uint16 / uint32 view or copy of an integer label image, 3-D (a 2-D image becomes (X, Y, 1)), dense in some
    axis permutation.  Returns (volume, is_view_of_input)."""
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3:
        raise ValueError("a 2-D or 3-D label image is required")
    same = True
    if a.dtype not in (np.uint16, np.uint32):
        if not np.issubdtype(a.dtype, np.integer):
            raise TypeError("label images must be integer arrays, not %s" % a.dtype)
        if a.size and (a.min() < 0 or a.max() > np.iinfo(np.uint32).max):
            raise ValueError("labels must fit in uint32")
        a = a.astype(np.uint16 if (a.size == 0 or a.max() <= 65535) else np.uint32)
        same = False
    if not _capi._dense_permuted(a):
        a = np.ascontiguousarray(a)
        same = False
    return a, same


class ResidentVolume(object):
    """One label image uploaded ONCE to one GPU context and kept there: the sweep, the wall-voxel extraction, the
    label lookup-table passes and the first voxel layer all run on the resident copy (the reference re-scans the
    host image for each of them; round 1 of this build re-uploaded it for each).  `uploads` counts H2D copies."""

    def __init__(self, array, device=0):
        self.host, self.is_input = _as_label_volume(array)
        self.device = device
        self.ctx = _capi.Context(device)
        self.uploads = 0
        self.ms = {}                  # wall-clock milliseconds of the last upload / sweep / fetch (host side)
        self.upload()

    def close(self):
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, array=None):
        """(Re-)upload: after construction, or after the host image was modified in place."""
        import time
        if array is not None:
            self.host, self.is_input = _as_label_volume(array)
        t0 = time.perf_counter()
        self.ctx.set_volume(self.host)
        self.ms["upload"] = (time.perf_counter() - t0) * 1e3
        self.uploads += 1

    def extract(self, features=_capi.F_ALL, max_label=None, sparse=None):
        """One sweep of the resident volume.  Ids that are sparse (`wants_compaction`) are swept in their ranks: the
        Extraction then has one row per id present (`.ids`) instead of rows 0..max_label; sparse=False insists on dense rows."""
        import time
        t0 = time.perf_counter()
        x = extract_resident(self.ctx, self.host.shape, features, max_label, sparse)
        self.ms["extract"] = (time.perf_counter() - t0) * 1e3
        return x

    def wall_table(self):
        if self.host.flags.c_contiguous:       # memory order IS np.where order: the device groups the records by pair
            lo, hi, coords, ms = self.ctx.wall_voxels(by_pair=True)
            return WallTable(lo, hi, coords, ms, grouped=True)
        lo, hi, coords, ms = self.ctx.wall_voxels()
        if coords.shape[0]:                    # records come in memory order: np.where order wanted
            order = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
            lo, hi, coords = lo[order], hi[order], coords[order]
        return WallTable(lo, hi, coords, ms)

    def wall_medians(self, max_iter=200):
        """(keys uint64[E] ascending = lo << 32 | hi, sizes[E], medians int64[E, 3], moving bool[E]) of every wall, computed on the device
        from the records grouped by pair -- or None when the image is not C-ordered (the order of a wall's voxels decides
        ties: the caller then works from wall_table() on the host)."""
        if not self.host.flags.c_contiguous:
            return None
        keys, sizes, med, ms, moving = self.ctx.wall_medians(max_iter)
        self.ms["wall_medians"] = ms
        return keys, sizes, med.astype(np.int64), moving

    def relabel(self, lut, features=_capi.F_ALL):
        """v -> lut[v] on the resident volume AND on `self.host` (copied back), then sweep the new volume."""
        self.ctx.relabel(lut)
        if not self.host.flags.writeable:
            self.host = self.host.copy()
            self.is_input = False
        self.ctx.get_volume(self.host)
        return self.extract(features)

    def map(self, lut, fill):
        return self.ctx.map_labels(lut, fill, self.host)

    def first_layer(self, background, keep_background=True):
        return self.ctx.first_layer(background, keep_background, self.host)

    def hollow(self, background, remove_background=True, label_bits=0):
        return self.ctx.hollow(background, remove_background, self.host, label_bits)

    def layer18(self):
        return self.ctx.layer18(self.host)


def extract_volume(array, features=_capi.F_ALL, device=0, context=None, max_label=None,
                   impl=None, tile_planes=None, sparse=None):
    """Upload `array` (uint16/uint32, any dense layout) and run the fused sweep on the GPU (`sparse`: see extract_resident)."""
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[:, :, None]
    own = context is None
    ctx = _capi.Context(device) if own else context
    try:
        if impl is not None:
            ctx.set_option(_capi.OPT_IMPL, impl)
        if tile_planes is not None:
            ctx.set_option(_capi.OPT_TILE_PLANES, tile_planes)
        ctx.set_volume(a)
        return extract_resident(ctx, a.shape, features, max_label, sparse)
    finally:
        if own:
            ctx.close()


def relabel_volume(array, lut, features=_capi.F_ALL, device=0):
    """One upload: relabel `array` IN PLACE through `lut` on the GPU (v -> lut[v] for v < len(lut)),
    copy it back, and sweep the relabelled volume.  `array` must be a writeable dense uint16/uint32
    ndarray (any axis permutation).  Returns the Extraction of the new volume."""
    rv = ResidentVolume(array, device)
    try:
        if not rv.is_input:
            raise TypeError("relabel_volume needs a writeable dense uint16 / uint32 array")
        return rv.relabel(lut, features)
    finally:
        rv.close()


def map_volume(array, lut, fill, device=0):
    """out[p] = lut[array[p]] (fill beyond the table) on the GPU; `lut` fixes the output dtype."""
    flat = np.asarray(array).ndim == 2
    rv = ResidentVolume(array, device)
    try:
        out = rv.map(lut, fill)
    finally:
        rv.close()
    return out[:, :, 0] if flat else out


class WallTable(object):
    """Wall voxels of every label pair of one image (SIA:759-880), from one GPU pass: records sorted by
    pair, each pair's voxels in np.where order of the image (lexicographic in the array axes)."""

    def __init__(self, lo, hi, coords, ms=None, grouped=False):
        key = (lo.astype(np.uint64) << np.uint64(32)) | hi.astype(np.uint64)
        if grouped:                            # ta_wall_voxels_get_by_pair: already sorted by pair on the device
            self.key, self.coords = key, coords
        else:
            order = np.argsort(key, kind="stable")
            self.key = key[order]
            self.coords = coords[order]
        self.ms = ms
        if self.key.size:
            cut = np.flatnonzero(self.key[1:] != self.key[:-1]) + 1
            self.start = np.concatenate([[0], cut])
            self.stop = np.concatenate([cut, [self.key.size]])
            self.pairs = self.key[self.start]
        else:
            self.start = self.stop = np.zeros(0, dtype=np.int64)
            self.pairs = np.zeros(0, dtype=np.uint64)

    def __len__(self):
        return int(self.pairs.size)

    def between(self, label_1, label_2):
        """(3, N) int64 coordinates of the wall voxels between the two labels (N = 0 when they do not touch)."""
        a, b = (int(label_1), int(label_2)) if label_1 < label_2 else (int(label_2), int(label_1))
        k = np.uint64((a << 32) | b)
        i = int(np.searchsorted(self.pairs, k))
        if i >= self.pairs.size or self.pairs[i] != k:
            return np.zeros((3, 0), dtype=np.int64)
        return np.ascontiguousarray(self.coords[self.start[i]:self.stop[i]].T).astype(np.int64)


def wall_voxel_table(array, device=0):
    """Upload `array` and extract the wall voxels of all label pairs (18-neighbourhood) on the GPU."""
    rv = ResidentVolume(array, device)
    try:
        return rv.wall_table()
    finally:
        rv.close()
