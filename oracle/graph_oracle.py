"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's graph assembly
(temporal_graph_from_image.py: `_graph_from_image` TGI:77-244, `generate_graph_topology` TGI:30-60)
on top of `OracleSIA`.  Only tests may import this.

PARITY UNPINNED: the reference has no test or fixture for this function, and its graph container
(`openalea.container.PropertyGraph`) is an absent third-party dependency, so the result is returned as
plain dictionaries keyed by LABEL (vertices) and by (lo, hi) LABEL PAIR (edges) -- the quantities the
container would hold, without its ids.  Statement order, label filtering, the property-name spellings
and the argument passed in `inertia_axis`'s `real` slot follow the reference text line by line.
"""
import numpy as np

from .sia_oracle import DICT, OracleSIA


def graph_tables(image, labels=None, background=1, properties=None, property_as_real=True,
                 ignore_cells_at_stack_margins=True, min_contact_area=None, voxelsize=None):
    if isinstance(image, OracleSIA):                                                                   # TGI:268-271, 104-106
        analysis = image
        if labels is None:
            labels = analysis.labels()          # taken BEFORE the margin cells are ignored, as in the reference
    else:
        analysis = OracleSIA(image, ignoredlabels=0, return_type=DICT, background=1, voxelsize=voxelsize)   # TGI:109
    if ignore_cells_at_stack_margins:                                                                  # TGI:112-114
        analysis.add2ignoredlabels(analysis.labels_at_stack_margins())
    if labels is None:                                                                                 # TGI:116-119
        labels = list(analysis.labels())
        if background in labels:
            del labels[labels.index(background)]
    else:                                                                                              # TGI:120-127
        if isinstance(labels, (int, np.integer)):
            labels = [labels]
        if background in labels:
            labels.remove(background)
        analysis.add2ignoredlabels(set(analysis.labels()) - set(labels))
    neighborhood = analysis.neighbors(labels, min_contact_area=min_contact_area)                       # TGI:129
    if not isinstance(neighborhood, dict):
        neighborhood = {labels[0]: neighborhood}
    labelset = set(labels)
    out = {"labels": list(labels), "vertex": {}, "edge": {}}
    out["edges"] = set((s, t) for s, ts in neighborhood.items() if s in labelset                       # TGI:51-55
                       for t in ts if s < t and t in labelset)
    V, E = out["vertex"], out["edge"]
    if "boundingbox" in properties:                                                                    # TGI:140-143
        V["boundingbox"] = dict(analysis.boundingbox(labels, real=property_as_real))
    if "volume" in properties and analysis.is3D():                                                     # TGI:145-148
        V["volume"] = dict(analysis.volume(labels, real=property_as_real))
    barycenters = None
    if "barycenter" in properties:                                                                     # TGI:150-155
        barycenters = analysis.center_of_mass(labels, real=property_as_real)
        V["barycenter"] = dict(barycenters)
    background_neighbors = set(analysis.neighbors(background))                                         # TGI:157-158
    background_neighbors.intersection_update(labelset)
    if "L1" in properties:                                                                             # TGI:159-161
        V["L1"] = dict((l, l in background_neighbors) for l in labels)
    if "border" in properties:                                                                         # TGI:163-170
        border = set(analysis.labels_at_stack_margins()) - set([background])
        V["border"] = dict((l, l in border) for l in labels)
    if "inertia_axis" in properties:                                                                   # TGI:172-176
        axes, values = analysis.inertia_axis(labels, bool(barycenters))
        V["inertia_axis"], V["inertia_values"] = dict(axes), dict(values)
    if "wall_surface" in properties:                                                                   # TGI:178-192
        filtered, unlabelled = {}, {}
        for s, ts in neighborhood.items():
            if s in labelset:
                filtered[s] = [t for t in ts if s < t and t in labelset]
                unlabelled[s] = [t for t in ts if t not in labelset and t != background]
        E["wall_surface"] = dict(analysis.wall_areas(filtered, real=property_as_real))
        V["unlabelled_wall_surface"] = dict(
            (s, sum(analysis.wall_areas({s: unlabelled[s]}, real=property_as_real).values())) for s in unlabelled)
    if "epidermis_surface" in properties:                                                              # TGI:197-208
        areas = analysis.cell_wall_area(background, list(background_neighbors), real=property_as_real)
        V["epidermis_surface"] = dict(((b if a == background else a), v) for (a, b), v in areas.items())
    if "wall_median" in properties:                                                                    # TGI:210-242
        walls = analysis.wall_voxels_per_cells_pairs(list(labels), dict((k, list(v)) for k, v in neighborhood.items()),
                                                     ignore_background=False)
        median = {}
        for pair, (x, y, z) in walls.items():
            origin = np.array([int(v) for v in weiszfeld(np.array([list(x), list(y), list(z)], dtype=float))])
            pts = np.array([x, y, z]).T
            d = ((pts - origin) ** 2).sum(axis=1)
            median[pair] = tuple(int(v) for v in pts[int(np.argmin(d))])    # closest_from_A: third-party, absent (assumed)
        E["wall_median"] = dict((p, m) for p, m in median.items() if p[0] in labelset and p[1] in labelset)
        V["unlabelled_wall_median"] = dict((p[1], m) for p, m in median.items() if p[0] == 0)
        V["epidermis_wall_median"] = dict((p[1], m) for p, m in median.items() if p[0] == 1)
    return out


def weiszfeld(X, numIter=200):
    """geometric_median as written in SIA:1586-1635 (point-by-point loops)."""
    import math
    y = np.mean(X, 1)
    while (y[0] in X[0]) and (y[1] in X[1]) and (y[2] in X[2]):
        y += 0.1
    convergence, dist, i = False, [], 0
    while (not convergence) and (i < numIter):
        num_x = num_y = num_z = denum = 0.0
        d = 0
        for j in range(X.shape[1]):
            div = math.sqrt((X[0, j] - y[0]) ** 2 + (X[1, j] - y[1]) ** 2 + (X[2, j] - y[2]) ** 2)
            num_x += X[0, j] / div
            num_y += X[1, j] / div
            num_z += X[2, j] / div
            denum += 1. / div
            d += div ** 2
        dist.append(d)
        if denum == 0.:
            return [0, 0, 0]
        y = [num_x / denum, num_y / denum, num_z / denum]
        if i > 3:
            convergence = (abs(dist[i] - dist[i - 2]) < 0.1)
        i += 1
    if i == numIter:
        raise ValueError("The Weiszfeld's algoritm did not converged")
    return np.array(y)
