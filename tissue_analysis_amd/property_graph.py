"""Array-backed property graph: the tables `graph_from_image` fills, behind the part of
``openalea.container.PropertyGraph``'s interface its callers use.

``openalea.container`` is a third-party dependency that is neither in the reference tree nor in this image; the
interface is restated from its call sites (temporal_graph_from_image.py:30-60, 309-407;
tissue_analysis_oalab/property_graph_to_dataframe.py:23-58).  The storage is NOT the container's dict-of-dicts:

  vertex table   `vertex_ids  int64[V]`                                  (the image labels, TGI:44)
  edge table     `edge_sources / edge_targets  int64[E]`  edge id = row   (label pairs lo < hi, sorted, from the sweep)
  adjacency      CSR over vertex rows, built on demand (`csr()`)
  properties     one `Column` per name: `values[n, ...]` + `valid[n]` aligned with the table -- straight from the
                 accumulators of the GPU sweep, no per-label Python objects

`vertex_property(name)` / `edge_property(name)` hand out a mutable mapping VIEW of a column (id -> value), so code written
against the dict interface keeps working; `vertex_column` / `edge_column` hand out the arrays.
"""
from __future__ import annotations

from collections.abc import MutableMapping

import numpy as np


class Column(object):
    """One property over a table: `values[n, ...]`, `valid[n]`, and how a row is shown through the mapping view."""

    def __init__(self, values, valid=None, show=None):
        self.values = values if isinstance(values, np.ndarray) else np.asarray(values)
        n = self.values.shape[0]
        self.valid = np.ones(n, dtype=bool) if valid is None else np.asarray(valid, dtype=bool)
        self.show = show

    @classmethod
    def empty(cls, n):
        return cls(np.empty(n, dtype=object), np.zeros(n, dtype=bool))

    def item(self, row):
        v = self.values[row]
        if self.show is not None:
            return self.show(v)
        if self.values.dtype == np.bool_:
            return bool(v)
        return v

    def assign(self, row, value):
        if self.values.dtype != object:
            try:
                cast = np.asarray(value, dtype=self.values.dtype)
                fits = cast.shape == self.values.shape[1:] and self.show is None and np.array_equal(cast, np.asarray(value))
            except (TypeError, ValueError):
                fits = False
            if fits:
                self.values[row] = cast
                self.valid[row] = True
                return
            shown = [self.item(i) if self.valid[i] else None for i in range(self.values.shape[0])]
            self.values = np.empty(len(shown), dtype=object)
            for i, s in enumerate(shown):
                self.values[i] = s
            self.show = None
        self.values[row] = value
        self.valid[row] = True

    def grown(self, extra):
        pad = np.zeros((extra,) + self.values.shape[1:], dtype=self.values.dtype)
        return Column(np.concatenate([self.values, pad]), np.concatenate([self.valid, np.zeros(extra, dtype=bool)]), self.show)


class _Index(object):
    """id -> row of a table: a dense lookup array when the ids are small non-negative integers (image labels), a dict otherwise."""

    def __init__(self, ids):
        self.n = ids.size
        dense = ids.size and int(ids.min()) >= 0 and int(ids.max()) <= 8 * ids.size + 1024
        if dense:
            self.lut = np.full(int(ids.max()) + 2, -1, dtype=np.int64)
            self.lut[ids] = np.arange(ids.size, dtype=np.int64)
            self.map = None
        else:
            self.lut = None
            self.map = dict(zip(ids.tolist(), range(ids.size)))

    def row(self, key):
        try:
            k = int(key)
        except (TypeError, ValueError):
            return -1
        if k != key:
            return -1
        if self.lut is not None:
            return int(self.lut[k]) if 0 <= k < self.lut.size else -1
        return self.map.get(k, -1)

    def rows(self, keys):
        """Rows of many ids at once (-1 where unknown)."""
        keys = np.asarray(keys, dtype=np.int64)
        if self.lut is not None:
            inside = (keys >= 0) & (keys < self.lut.size)
            return np.where(inside, self.lut[np.where(inside, keys, 0)], -1)
        return np.fromiter((self.map.get(k, -1) for k in keys.tolist()), dtype=np.int64, count=keys.size)


class ColumnView(MutableMapping):
    """Mapping id -> value over the valid rows of one column (what `vertex_property(name)` returns)."""

    def __init__(self, graph, table, name):
        self._graph, self._table, self._name = graph, table, name

    def _column(self):
        return self._graph._columns[self._table][self._name]

    def _ids(self):
        return self._graph.vertex_ids if self._table == "vertex" else np.arange(self._graph.nb_edges(), dtype=np.int64)

    def _row(self, key):
        if self._table == "vertex":
            return self._graph._vertex_index().row(key)
        try:
            k = int(key)
        except (TypeError, ValueError):
            return -1
        return k if (k == key and 0 <= k < self._graph.nb_edges()) else -1

    def __getitem__(self, key):
        row, col = self._row(key), self._column()
        if row < 0 or not col.valid[row]:
            raise KeyError(key)
        return col.item(row)

    def __setitem__(self, key, value):
        row = self._row(key)
        if row < 0:
            raise KeyError("%r is not a %s of the graph" % (key, self._table))
        self._column().assign(row, value)

    def __delitem__(self, key):
        row, col = self._row(key), self._column()
        if row < 0 or not col.valid[row]:
            raise KeyError(key)
        col.valid[row] = False

    def __contains__(self, key):
        row = self._row(key)
        return row >= 0 and bool(self._column().valid[row])

    def __iter__(self):
        return iter(self._ids()[self._column().valid].tolist())

    def __len__(self):
        return int(self._column().valid.sum())

    def __repr__(self):
        return "<%s property %r: %d of %d defined>" % (self._table, self._name, len(self), self._column().valid.size)


class PropertyGraph(object):
    def __init__(self, vertex_ids=None, edge_sources=None, edge_targets=None):
        self.vertex_ids = np.zeros(0, dtype=np.int64) if vertex_ids is None else np.array(vertex_ids, dtype=np.int64)
        self.edge_sources = np.zeros(0, dtype=np.int64) if edge_sources is None else np.array(edge_sources, dtype=np.int64)
        self.edge_targets = np.zeros(0, dtype=np.int64) if edge_targets is None else np.array(edge_targets, dtype=np.int64)
        if self.edge_sources.shape != self.edge_targets.shape:
            raise ValueError("edge sources and targets differ in length")
        self._columns = {"vertex": {}, "edge": {}}
        self._graph_property = {}
        self._vindex = None
        self._csr = None
        if np.unique(self.vertex_ids).size != self.vertex_ids.size:
            raise KeyError("vertex ids must be distinct")
        if self.edge_sources.size and (self._vertex_index().rows(self.edge_sources).min() < 0
                                       or self._vertex_index().rows(self.edge_targets).min() < 0):
            raise KeyError("edge between unknown vertices")

    # -- tables
    def _vertex_index(self):
        if self._vindex is None:
            self._vindex = _Index(self.vertex_ids)
        return self._vindex

    def vertex_rows(self, ids):
        """Rows in the vertex table of many vertex ids (-1 where unknown)."""
        return self._vertex_index().rows(ids)

    def csr(self):
        """(indptr[V+1], neighbour rows, edge ids): symmetric adjacency over vertex ROWS, neighbours of a row ascending."""
        if self._csr is None:
            s = self.vertex_rows(self.edge_sources)
            t = self.vertex_rows(self.edge_targets)
            src, dst = np.concatenate([s, t]), np.concatenate([t, s])
            eid = np.concatenate([np.arange(s.size), np.arange(s.size)])
            order = np.lexsort((dst, src))
            indptr = np.zeros(self.vertex_ids.size + 1, dtype=np.int64)
            np.cumsum(np.bincount(src, minlength=self.vertex_ids.size), out=indptr[1:])
            self._csr = (indptr, dst[order], eid[order])
        return self._csr

    def vertex_column(self, name):
        c = self._columns["vertex"][name]
        return c.values, c.valid

    def edge_column(self, name):
        c = self._columns["edge"][name]
        return c.values, c.valid

    def set_vertex_column(self, name, values, valid=None, show=None):
        """Attach a whole column (values[V, ...]) as a vertex property; replaces one of the same name."""
        col = Column(values, valid, show)
        if col.valid.size != self.vertex_ids.size:
            raise ValueError("column %r has %d rows for %d vertices" % (name, col.valid.size, self.vertex_ids.size))
        self._columns["vertex"][name] = col

    def set_edge_column(self, name, values, valid=None, show=None):
        col = Column(values, valid, show)
        if col.valid.size != self.edge_sources.size:
            raise ValueError("column %r has %d rows for %d edges" % (name, col.valid.size, self.edge_sources.size))
        self._columns["edge"][name] = col

    # -- topology, one element at a time (the container's incremental interface)
    def add_vertex(self, vid=None):
        if vid is None:
            vid = int(self.vertex_ids.max()) + 1 if self.vertex_ids.size else 0
        if self._vertex_index().row(vid) >= 0:
            raise KeyError("vertex %r already in the graph" % (vid,))
        self.vertex_ids = np.append(self.vertex_ids, np.int64(vid))
        cols = self._columns["vertex"]
        for name in cols:
            cols[name] = cols[name].grown(1)
        self._vindex = self._csr = None
        return vid

    def add_edge(self, sid, tid, eid=None):
        if not (self.has_vertex(sid) and self.has_vertex(tid)):
            raise KeyError("edge (%r, %r) between unknown vertices" % (sid, tid))
        if eid is not None and eid != self.edge_sources.size:
            raise ValueError("edge ids are the rows of the edge table: the next one is %d" % self.edge_sources.size)
        self.edge_sources = np.append(self.edge_sources, np.int64(sid))
        self.edge_targets = np.append(self.edge_targets, np.int64(tid))
        cols = self._columns["edge"]
        for name in cols:
            cols[name] = cols[name].grown(1)
        self._csr = None
        return self.edge_sources.size - 1

    def vertices(self):
        return iter(self.vertex_ids.tolist())

    def edges(self):
        return iter(range(self.edge_sources.size))

    def nb_vertices(self):
        return int(self.vertex_ids.size)

    def nb_edges(self):
        return int(self.edge_sources.size)

    def has_vertex(self, vid):
        return self._vertex_index().row(vid) >= 0

    def source(self, eid):
        return int(self.edge_sources[eid])

    def target(self, eid):
        return int(self.edge_targets[eid])

    def edge_vertices(self, eid):
        return int(self.edge_sources[eid]), int(self.edge_targets[eid])

    def neighbors(self, vid):
        row = self._vertex_index().row(vid)
        if row < 0:
            raise KeyError(vid)
        indptr, nbr, _ = self.csr()
        return set(self.vertex_ids[nbr[indptr[row]:indptr[row + 1]]].tolist())

    def nb_neighbors(self, vid):
        return len(self.neighbors(vid))

    # -- properties through the mapping interface
    def _add(self, table, name, values):
        if name in self._columns[table]:
            raise ValueError("Existing %s property '%s'" % (table, name))
        n = self.vertex_ids.size if table == "vertex" else self.edge_sources.size
        self._columns[table][name] = Column.empty(n)
        if values:
            ColumnView(self, table, name).update(values)

    def add_vertex_property(self, name, values=None):
        self._add("vertex", name, values)

    def remove_vertex_property(self, name):
        del self._columns["vertex"][name]

    def vertex_property(self, name):
        if name not in self._columns["vertex"]:
            raise KeyError(name)
        return ColumnView(self, "vertex", name)

    def vertex_properties(self):
        return dict((name, ColumnView(self, "vertex", name)) for name in self._columns["vertex"])

    def vertex_property_names(self):
        return iter(self._columns["vertex"])

    def add_edge_property(self, name, values=None):
        self._add("edge", name, values)

    def remove_edge_property(self, name):
        del self._columns["edge"][name]

    def edge_property(self, name):
        if name not in self._columns["edge"]:
            raise KeyError(name)
        return ColumnView(self, "edge", name)

    def edge_properties(self):
        return dict((name, ColumnView(self, "edge", name)) for name in self._columns["edge"])

    def edge_property_names(self):
        return iter(self._columns["edge"])

    def add_graph_property(self, name, value=None):
        if name in self._graph_property:
            raise ValueError("Existing graph property '%s'" % name)
        self._graph_property[name] = value

    def graph_property(self, name):
        return self._graph_property[name]

    def graph_properties(self):
        return self._graph_property
