"""``graph_from_image``: the immediate caller of the hot path, assembled from the ARRAYS of the GPU
sweep -- one masked reduction over the sweep's label-pair list per property, no per-label Python.

What it keeps of the reference module (TGI = temporal_graph_from_image.py): the public names and signatures
(`graph_from_image` TGI:260-284 and the helpers TGI:30-74, 309-398), the property names, and the rules that decide the
numbers.  How it computes them is unrelated to the reference's dict-and-loop code:

    inputs    Extraction: count[L], bbox[L,6], sum1[L,3], sum2[L,6]; pairs lo[P] < hi[P] sorted, faces[P,3]
    vertices  the requested labels (vertex id = label, TGI:44)
    edges     the pairs with both ends requested whose REAL contact area is not below `min_contact_area`
              (`neighbors(labels, min_contact_area)` always filters on real areas, SIA:538)  -- a mask over the pair list
    columns   volume / barycenter / boundingbox / inertia: row gathers of the accumulators;
              L1, epidermis_surface: the pairs holding the background; border: a test on the bounding boxes;
              wall_surface: faces . face areas of the edge rows; unlabelled_wall_surface: a bincount over the pairs with
              exactly one requested end; wall medians: computed ON THE DEVICE from the wall-voxel records it grouped by pair
              (ta_wall_medians: E x 3 integers come back), or by the same arithmetic on the host (geometry.median_voxels)
              when a wall table is on the host already.

Rules of the reference that are kept because they decide results (each pinned by tests against oracle/graph_oracle.py):
  * `availables_spatial_properties()` advertises 'wall_area' / 'epidermis_area' but the builder tests for 'wall_surface' /
    'epidermis_surface' (TGI:169, 187): only the latter compute anything; 'L2' is advertised and never computed.
  * inertia eigenvalues are in real units iff 'barycenter' was requested: TGI:165 passes the barycentre dict in
    `inertia_axis`'s `real` slot.
  * 'epidermis_surface' calls a method the reference class does not define (TGI:197, `cell_wall_surface`); its evident
    intent, the wall area with the background, is what is computed.
  * wall medians (TGI:210-242): geometric median by the reference's stopping rules, truncated, then the nearest wall
    voxel; walls keyed (0, l) go to 'unlabelled_wall_median', walls keyed (1, l) to 'epidermis_wall_median' whatever the
    background is.
  * when an analysis object is passed with `labels=None`, the labels are taken BEFORE the margin cells are ignored
    (TGI:268-271), so margin cells stay vertices.
"""
from __future__ import annotations

import numpy as np

from .property_graph import PropertyGraph, _Index
from .spatial_image_analysis import AbstractSpatialImageAnalysis, DICT, SpatialImageAnalysis

_INT = (int, np.integer)


def availables_spatial_properties():
    return ['boundingbox', 'volume', 'barycenter', 'L1', 'L2', 'border', 'inertia_axis', 'wall_area', 'epidermis_area',
            'wall_median']


def availables_properties():
    return sorted(availables_spatial_properties())


spatio_temporal_properties2D = ['barycenter', 'boundingbox', 'border', 'L1', 'epidermis_area', 'inertia_axis']
spatio_temporal_properties3D = availables_properties()


# ----------------------------------------------------------------------------- how table rows are shown as Python values
def _show_slices(row):
    return (slice(int(row[0]), int(row[3])), slice(int(row[1]), int(row[4])), slice(int(row[2]), int(row[5])))


def _show_intervals(row):
    return [(row[0, 0], row[0, 1]), (row[1, 0], row[1, 1]), (row[2, 0], row[2, 1])]


def _show_axes(row):
    return [row[0], row[1], row[2]]


def _show_point(row):
    return (int(row[0]), int(row[1]), int(row[2]))


# ----------------------------------------------------------------------------- the assembly
class _PairView(object):
    """The sweep's pair list with the per-call quantities every property needs."""

    def __init__(self, analysis, vertex_ids, min_contact_area):
        x = analysis.extraction
        self.lo = x.pair_lo.astype(np.int64)
        self.hi = x.pair_hi.astype(np.int64)
        faces = x.pair_faces.astype(np.float64)
        surf = np.asarray(analysis.get_voxel_face_surface(), dtype=np.float64)
        self.area_real = faces[:, 0] * surf[0] + faces[:, 1] * surf[1] + faces[:, 2] * surf[2]
        self.area_voxels = faces[:, 0] + faces[:, 1] + faces[:, 2]
        ids = np.asarray(vertex_ids, dtype=np.int64)
        self.row_of = _Index(ids).rows                                   # labels -> vertex rows (-1: not a vertex)
        self.lo_in = self.row_of(self.lo) >= 0
        self.hi_in = self.row_of(self.hi) >= 0
        self.kept = np.ones(self.lo.size, dtype=bool) if min_contact_area is None else ~(self.area_real < min_contact_area)

    def area(self, real):
        return self.area_real if real else self.area_voxels

    def partners_of(self, label):
        """(pair rows holding `label`, the label at their other end)."""
        rows = np.flatnonzero((self.lo == label) | (self.hi == label))
        return rows, np.where(self.lo[rows] == label, self.hi[rows], self.lo[rows])


def _wall_median_columns(graph, analysis, pairs, background):
    """Median voxel of every wall the reference visits (TGI:210-242 through wall_voxels_per_cells_pairs with
    ignore_background=False): kept pairs with both ends requested, or one end requested and the other the background."""
    bg = -1 if background is None else int(background)
    want = pairs.kept & ((pairs.lo_in & pairs.hi_in) | (pairs.lo_in & (pairs.hi == bg)) | (pairs.hi_in & (pairs.lo == bg)))
    lo, hi = pairs.lo[want], pairs.hi[want]
    key = (lo.astype(np.uint64) << np.uint64(32)) | hi.astype(np.uint64)
    found, chosen = analysis.wall_medians_of(key)          # (on the device when the analysis holds a resident volume)
    lo, hi = lo[found], hi[found]

    V, E = graph.nb_vertices(), graph.nb_edges()
    edge_value, edge_valid = np.zeros((E, 3), dtype=np.int64), np.zeros(E, dtype=bool)
    both = (pairs.row_of(lo) >= 0) & (pairs.row_of(hi) >= 0)
    ekey = (graph.edge_sources.astype(np.uint64) << np.uint64(32)) | graph.edge_targets.astype(np.uint64)
    erow = np.searchsorted(ekey, ((lo[both].astype(np.uint64) << np.uint64(32)) | hi[both].astype(np.uint64)))
    edge_value[erow], edge_valid[erow] = chosen[both], True
    graph.set_edge_column('wall_median', edge_value, edge_valid, _show_point)
    for name, first in (('epidermis_wall_median', 1), ('unlabelled_wall_median', 0)):
        value, valid = np.zeros((V, 3), dtype=np.int64), np.zeros(V, dtype=bool)
        sel = (lo == first) & (pairs.row_of(hi) >= 0)
        value[pairs.row_of(hi[sel])], valid[pairs.row_of(hi[sel])] = chosen[sel], True
        graph.set_vertex_column(name, value, valid, _show_point)


def tissue_tables(analysis, labels, background, properties, property_as_real=True, min_contact_area=None):
    """The vertex / edge tables of the tissue graph from the sweep results held by `analysis`.
    `labels`: the vertex ids (any order, distinct).  Returns a PropertyGraph whose columns are plain numpy arrays."""
    x = analysis.extraction
    ids = np.asarray(labels, dtype=np.int64).reshape(-1)
    pairs = _PairView(analysis, ids, min_contact_area)
    is_edge = pairs.kept & pairs.lo_in & pairs.hi_in
    graph = PropertyGraph(ids, pairs.lo[is_edge], pairs.hi[is_edge])
    graph.set_vertex_column('label', ids.copy())
    graph.add_graph_property("units", dict())
    V = ids.size
    rows = x.rows_of(ids, missing=-1)                                      # accumulator rows (sparse ids: not the ids)
    known = rows >= 0
    rows = np.where(known, rows, 0)                                        # (row 0 stands in for unknown ids)
    standin = np.where(known, ids, x.labels_of(0) if x.nrows else 0)       # ... as a label, for the methods that take labels
    present = known & (x.count[rows] > 0)
    # volume / barycenter / inertia go through `label_request`, which drops what the analysis ignores (SIA:387-414)
    ignored = np.fromiter(analysis.ignoredlabels(), dtype=np.int64, count=len(analysis.ignoredlabels()))
    asked = present & ~np.isin(ids, ignored)
    vs = np.asarray(analysis._voxelsize, dtype=np.float64)
    real = bool(property_as_real)

    if 'boundingbox' in properties:
        box = x.bbox[rows].astype(np.int64)
        if real:
            box = np.stack([box[:, :3] * vs, box[:, 3:] * vs], axis=2)      # [V, axis, (start, stop)]
            graph.set_vertex_column('boundingbox', box, present, _show_intervals)
        else:
            graph.set_vertex_column('boundingbox', box, present, _show_slices)
    if 'volume' in properties and analysis.is3D():
        volume = x.count[rows].astype(np.float64)
        graph.set_vertex_column('volume', volume * (vs[0] * vs[1] * vs[2]) if real else volume, asked)
    with_barycenter = 'barycenter' in properties
    if with_barycenter:
        com = x.barycenters(standin)
        graph.set_vertex_column('barycenter', com * vs if real else com, asked)

    bg_rows, bg_partner = pairs.partners_of(-1 if background is None else int(background))
    touches = pairs.row_of(bg_partner) >= 0                                # requested labels sharing a face with the background
    l1 = np.zeros(V, dtype=bool)
    l1[pairs.row_of(bg_partner[touches])] = True
    if 'L1' in properties:
        graph.set_vertex_column('L1', l1)
    if 'border' in properties:
        at_margin = np.asarray(analysis.labels_at_stack_margins(), dtype=np.int64)
        border = np.zeros(V, dtype=bool)
        hit = graph.vertex_rows(at_margin[at_margin != (-1 if background is None else int(background))])
        border[hit[hit >= 0]] = True
        graph.set_vertex_column('border', border)
    if 'inertia_axis' in properties:
        axes, values = x.inertia(standin)
        if with_barycenter and asked.any():                                # TGI:165: the barycentres sit in the `real` slot
            values = values * np.linalg.norm(axes * vs, axis=2)
        graph.set_vertex_column('inertia_axis', axes, asked, _show_axes)
        graph.set_vertex_column('inertia_values', values, asked)

    if 'wall_surface' in properties:
        area = pairs.area(real)
        graph.set_edge_column('wall_surface', area[is_edge])
        # walls with a label that is neither requested nor the background, summed at their requested end -- only those
        # whose outer label is the LARGER one: the reference sums them with `wall_areas`, which skips n <= label (SIA:988)
        one_end = pairs.kept & pairs.lo_in & ~pairs.hi_in & (pairs.hi != (-1 if background is None else int(background)))
        graph.set_vertex_column('unlabelled_wall_surface',
                                np.bincount(pairs.row_of(pairs.lo[one_end]), weights=area[one_end], minlength=V))
    if 'epidermis_surface' in properties:
        value = np.zeros(V, dtype=np.float64)
        value[pairs.row_of(bg_partner[touches])] = pairs.area(real)[bg_rows[touches]]
        graph.set_vertex_column('epidermis_surface', value, l1)
    if 'wall_median' in properties:
        _wall_median_columns(graph, analysis, pairs, background)
    return graph


def _graph_from_image(image, labels, background, default_properties, property_as_real, ignore_cells_at_stack_margins,
                      min_contact_area):
    """TGI:77-244: which analysis object, which labels are ignored, which become vertices -- then the tables."""
    if isinstance(image, AbstractSpatialImageAnalysis):
        analysis = image
    else:
        try:
            analysis = SpatialImageAnalysis(image, ignoredlabels=0, return_type=DICT, background=1)
        except (ValueError, AssertionError):
            analysis = SpatialImageAnalysis(image, ignoredlabels=0, return_type=DICT)
    if ignore_cells_at_stack_margins:
        analysis.add2ignoredlabels(analysis.labels_at_stack_margins())
    if labels is None:
        ids = np.asarray(analysis.labels(), dtype=np.int64)
        ids = ids[ids != background]
    else:
        if isinstance(labels, _INT):
            labels = [labels]
        if background in labels:
            labels.remove(background)                                      # the caller's list, as in the reference
        ids = np.asarray(labels, dtype=np.int64)
        analysis.add2ignoredlabels(np.setdiff1d(np.asarray(analysis.labels(), dtype=np.int64), ids))
    return tissue_tables(analysis, ids, background, default_properties, property_as_real, min_contact_area)


def graph_from_image2D(image, labels, background, spatio_temporal_properties, property_as_real,
                       ignore_cells_at_stack_margins, min_contact_area):
    return _graph_from_image(image, labels, background, spatio_temporal_properties, property_as_real,
                             ignore_cells_at_stack_margins, min_contact_area)


graph_from_image3D = graph_from_image2D        # (the reference's two entry points run the same body, TGI:247-258)


def graph_from_image(image, labels=None, background=1, spatio_temporal_properties=None, property_as_real=True,
                     ignore_cells_at_stack_margins=True, min_contact_area=None):
    """TGI:260-284.  `image`: a label image or an analysis object of this package."""
    if isinstance(image, AbstractSpatialImageAnalysis):
        shape = np.shape(image.image)
        if labels is None:
            labels = image.labels()                                        # before the margin cells are ignored
    else:
        shape = np.shape(image)
    flat = len(shape) == 2 or (len(shape) == 3 and shape[2] == 1)
    if spatio_temporal_properties is None:
        spatio_temporal_properties = spatio_temporal_properties2D if flat else spatio_temporal_properties3D
    build = graph_from_image2D if flat else graph_from_image3D
    return build(image, labels, background, spatio_temporal_properties, property_as_real, ignore_cells_at_stack_margins,
                 min_contact_area)


# ----------------------------------------------------------------------------- helpers of TGI:30-60, 309-398, by name
def generate_graph_topology(labels, neighborhood):
    """TGI:30-60: vertices = `labels`; one edge per pair s < t, both in `labels`, with t listed under s.
    Returns (graph, label -> vertex id, (s, t) -> edge id)."""
    ids = np.asarray(list(labels), dtype=np.int64)
    member = set(ids.tolist())
    src = [s for s, ts in neighborhood.items() if s in member for t in ts if s < t and t in member]
    dst = [t for s, ts in neighborhood.items() if s in member for t in ts if s < t and t in member]
    graph = PropertyGraph(ids, src, dst)
    graph.set_vertex_column('label', ids.copy())
    return graph, dict(zip(ids.tolist(), ids.tolist())), dict(zip(zip(src, dst), range(len(src))))


def label2vertex_map(graph, time_point=None):
    values, valid = graph.vertex_column('label')
    return dict(zip(values[valid].tolist(), graph.vertex_ids[valid].tolist()))


def labelpair2edge_map(graph, time_point=None):
    values, _ = graph.vertex_column('label')
    a = values[graph.vertex_rows(graph.edge_sources)]
    b = values[graph.vertex_rows(graph.edge_targets)]
    return dict(zip(zip(np.minimum(a, b).tolist(), np.maximum(a, b).tolist()), range(graph.nb_edges())))


def _attach(graph, table, name, keys, values, translate, overwrite):
    names = graph.vertex_properties() if table == 'vertex' else graph.edge_properties()
    if name in names:
        if not overwrite:
            raise ValueError("Existing {} property '{}'".format(table, name))
        (graph.remove_vertex_property if table == 'vertex' else graph.remove_edge_property)(name)
    (graph.add_vertex_property if table == 'vertex' else graph.add_edge_property)(name)
    target = graph.vertex_property(name) if table == 'vertex' else graph.edge_property(name)
    for k, v in zip(keys, values):
        target[translate[k]] = v
    return "Done."


def add_vertex_property_from_dictionary(graph, name, dictionary, mlabel2vertex=None, time_point=None, overwrite=False):
    return _attach(graph, 'vertex', name, dictionary.keys(), dictionary.values(),
                   label2vertex_map(graph, time_point) if mlabel2vertex is None else mlabel2vertex, overwrite)


def add_vertex_property_from_label_and_value(graph, name, labels, property_values, mlabel2vertex=None, overwrite=False):
    return _attach(graph, 'vertex', name, labels, property_values,
                   label2vertex_map(graph) if mlabel2vertex is None else mlabel2vertex, overwrite)


def add_edge_property_from_dictionary(graph, name, dictionary, mlabelpair2edge=None, time_point=None, overwrite=False):
    return _attach(graph, 'edge', name, dictionary.keys(), dictionary.values(),
                   labelpair2edge_map(graph, time_point) if mlabelpair2edge is None else mlabelpair2edge, overwrite)


def add_edge_property_from_label_and_value(graph, name, label_pairs, property_values, mlabelpair2edge=None, overwrite=False):
    return _attach(graph, 'edge', name, label_pairs, property_values,
                   labelpair2edge_map(graph) if mlabelpair2edge is None else mlabelpair2edge, overwrite)


add_vertex_property_from_label_property = add_vertex_property_from_dictionary
add_edge_property_from_label_property = add_edge_property_from_dictionary


def retrieve_label_neighbors(SpI_Analysis, label, labelset, min_contact_area, real_area):
    return set(SpI_Analysis.neighbors(label, min_contact_area, real_area)) & labelset


# ----------------------------------------------------------------------------- DataFrame export
def property_graph_to_dataframe(graph, element='vertex', labels=None):
    """tissue_analysis_oalab/property_graph_to_dataframe.py:23-58 on the tables: every scalar property becomes a column,
    'barycenter' becomes barycenter_x / _y / _z, rows are the vertex ids (`labels` restricts them) or the ids of the edges
    between them.  A property not defined on a row gives NaN there (the reference raises KeyError)."""
    import pandas as pd
    vrow = np.arange(graph.nb_vertices()) if labels is None else np.unique(graph.vertex_rows(list(labels)))
    vrow = vrow[vrow >= 0]
    if element == 'vertex':
        rows, ids, table = vrow, graph.vertex_ids[vrow], 'vertex'
    elif element == 'edge':
        chosen = np.zeros(graph.nb_vertices(), dtype=bool)
        chosen[vrow] = True
        rows = np.flatnonzero(chosen[graph.vertex_rows(graph.edge_sources)] & chosen[graph.vertex_rows(graph.edge_targets)])
        ids, table = rows, 'edge'
    else:
        return pd.DataFrame()
    data = {}
    for name in (graph.vertex_property_names() if table == 'vertex' else graph.edge_property_names()):
        values, valid = graph.vertex_column(name) if table == 'vertex' else graph.edge_column(name)
        if not valid.any():
            continue
        if values.dtype == object:
            first = np.asarray(values[np.flatnonzero(valid)[0]])
            if first.ndim or not (np.issubdtype(first.dtype, np.number) or first.dtype == np.bool_):
                continue
            values = np.array([v if ok else np.nan for v, ok in zip(values, valid)])
        if values.ndim == 1:
            data[name] = values[rows] if valid[rows].all() else np.where(valid[rows], values[rows].astype(np.float64), np.nan)
        elif name == 'barycenter':
            for k, axis in enumerate('xyz'[:values.shape[1]]):
                data[name + "_" + axis] = np.where(valid[rows], values[rows, k], np.nan)
    return pd.DataFrame(data, index=ids)
