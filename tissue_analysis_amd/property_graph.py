"""A small property graph with the part of ``openalea.container.PropertyGraph``'s interface that
``graph_from_image`` and the DataFrame export use (TGI:30-60, 309-407;
tissue_analysis_oalab/property_graph_to_dataframe.py:23-58).

``openalea.container`` is a third-party dependency that is neither in the reference tree nor in
this image, so the interface is restated from its call sites: vertices carry caller-chosen ids
(the image labels), edges get consecutive ids in insertion order, properties are plain dicts keyed
by vertex / edge id, graph properties are a dict.
"""
from __future__ import annotations


class PropertyGraph(object):
    def __init__(self):
        self._vertices = {}            # vid -> set of incident edge ids
        self._edges = {}               # eid -> (source vid, target vid)
        self._vertex_property = {}
        self._edge_property = {}
        self._graph_property = {}

    # -- topology
    def add_vertex(self, vid=None):
        if vid is None:
            vid = max(self._vertices) + 1 if self._vertices else 0
        if vid in self._vertices:
            raise KeyError("vertex %r already in the graph" % (vid,))
        self._vertices[vid] = set()
        return vid

    def add_edge(self, sid, tid, eid=None):
        if sid not in self._vertices or tid not in self._vertices:
            raise KeyError("edge (%r, %r) between unknown vertices" % (sid, tid))
        if eid is None:
            eid = len(self._edges)
        self._edges[eid] = (sid, tid)
        self._vertices[sid].add(eid)
        self._vertices[tid].add(eid)
        return eid

    def vertices(self):
        return iter(self._vertices)

    def edges(self):
        return iter(self._edges)

    def nb_vertices(self):
        return len(self._vertices)

    def nb_edges(self):
        return len(self._edges)

    def has_vertex(self, vid):
        return vid in self._vertices

    def source(self, eid):
        return self._edges[eid][0]

    def target(self, eid):
        return self._edges[eid][1]

    def edge_vertices(self, eid):
        return self._edges[eid]

    def neighbors(self, vid):
        out = set()
        for e in self._vertices[vid]:
            s, t = self._edges[e]
            out.add(t if s == vid else s)
        return out

    # -- properties
    def add_vertex_property(self, name, values=None):
        if name in self._vertex_property:
            raise ValueError("Existing vertex property '%s'" % name)
        self._vertex_property[name] = dict(values) if values else {}

    def remove_vertex_property(self, name):
        del self._vertex_property[name]

    def vertex_property(self, name):
        return self._vertex_property[name]

    def vertex_properties(self):
        return self._vertex_property

    def vertex_property_names(self):
        return iter(self._vertex_property)

    def add_edge_property(self, name, values=None):
        if name in self._edge_property:
            raise ValueError("Existing edge property '%s'" % name)
        self._edge_property[name] = dict(values) if values else {}

    def remove_edge_property(self, name):
        del self._edge_property[name]

    def edge_property(self, name):
        return self._edge_property[name]

    def edge_properties(self):
        return self._edge_property

    def edge_property_names(self):
        return iter(self._edge_property)

    def add_graph_property(self, name, value=None):
        if name in self._graph_property:
            raise ValueError("Existing graph property '%s'" % name)
        self._graph_property[name] = value

    def graph_property(self, name):
        return self._graph_property[name]

    def graph_properties(self):
        return self._graph_property
