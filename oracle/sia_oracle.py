"""ORACLE -- test infrastructure only.  Never imported by the product package.

CPU restatement (Python 3, numpy + scipy.ndimage) of the per-label feature
extractors of the reference, ``src/vplants/tissue_analysis/spatial_image_analysis.py``
(cited below as SIA:<line>).  It keeps the reference's *algorithm* -- one bounding-box
crop per label and the same scipy.ndimage / numpy calls in the same order -- so it is
both the parity checker for the HIP path and the "reference CPU path" that bench.py
times (cpu_baseline.kind = "port", 1 core: the reference has no parallelism).

Pinning status: the reference cannot be imported here (Python 2 source, and it
needs the un-vendored ``openalea.image``), it has no tests (test/__init__.py:1-11)
and its arithmetic lives in unpinned scipy/numpy.  The only known answers it holds
are the docstring examples on one 4x6 image (SIA:344-353, 429-450, 490-511, 553-574,
916-927, 970-982, 1211-1226); ``tests/test_oracle_known_answers.py`` checks this file
against every one of them.  Beyond those examples: PARITY UNPINNED -- parity means
"equal to this restatement run on scipy 1.15.3 / numpy 2.2.6".

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Deliberate divergences from reference quirks (documented in DESIGN.md):
* ``volume`` casts the label list with ``np.int16`` (SIA:1231), which wraps or raises
  for ids > 32767; the oracle passes the true ids.
* ``labels()`` order is whatever a Python 2 ``set`` yields (SIA:363-364); the oracle
  returns ascending order.
"""
from __future__ import annotations

import copy

import numpy as np
import scipy.ndimage as nd

NPLIST, LIST, DICT = range(3)  # SIA:204


class OracleImage(np.ndarray):
    """Stand-in for openalea.image SpatialImage (SIA:27): ndarray + voxelsize + info."""

    def __new__(cls, arr, voxelsize=None, info=None):
        a = np.asarray(arr)
        if a.ndim == 2:  # SpatialImage presents 2D arrays as (X, Y, 1)
            a = a[:, :, None]
        obj = a.view(cls)
        obj.voxelsize = tuple(float(v) for v in (voxelsize if voxelsize is not None
                                                 else (1.0,) * a.ndim))
        obj.info = dict(info or {})
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.voxelsize = getattr(obj, "voxelsize", (1.0,) * self.ndim)
        self.info = getattr(obj, "info", {})


# ----------------------------------------------------------------------------- helpers
def dilate_slices(slices):
    """Bounding box grown by one voxel, clamped at 0 only (SIA:35-37)."""
    return tuple(slice(max(0, s.start - 1), s.stop + 1) for s in slices)


def wall_labels(crop, label_id):
    """Labels found under the 6-connected one-voxel shell around `label_id` (SIA:45-60)."""
    inside = (crop == label_id)
    grown = nd.binary_dilation(inside)  # default structure = 6-connectivity, border_value 0
    shell = grown & ~inside
    return set(np.unique(np.asarray(crop)[shell]).tolist())


def directional_kernels():
    """Six 3x3x3 structuring elements: centre + ONE face neighbour each (SIA:695-716).

    Order X1, X2, Y1, Y2, Z1, Z2; kernel index a belongs to axis a // 2.
    """
    ks = []
    for axis in range(3):
        for missing in (0, 2):
            k = np.zeros((3, 3, 3), dtype=bool)
            sel = [1, 1, 1]
            sel[axis] = slice(None)
            k[tuple(sel)] = True
            sel[axis] = missing
            k[tuple(sel)] = False
            ks.append(k)
    return tuple(ks)


def covariance_and_axes(coords):
    """cov = P.P^T / max(3, N) (SIA:137-150); eigenpairs sorted by decreasing eigenvalue,
    vectors as rows (SIA:152-167)."""
    coords = np.asarray(coords, dtype=float)
    if coords.shape[0] > 3:
        coords = coords.T
    cov = (1.0 / max(coords.shape)) * np.dot(coords, coords.T)
    val, vec = np.linalg.eig(cov)
    order = val.argsort()[::-1]
    return cov, val[order], np.array(vec[:, order]).T


# ------------------------------------------------------------------------ the API mirror
class OracleSIA(object):
    """Python-3 restatement of AbstractSpatialImageAnalysis + SpatialImageAnalysis3D
    (SIA:206-1448): same public method names, arguments and return-shape rules."""

    def __init__(self, image, ignoredlabels=(), return_type=DICT, background=None,
                 voxelsize=None):
        # SIA:212-270
        if isinstance(image, OracleImage) and voxelsize is None:
            self.image = image
        else:
            vs = voxelsize if voxelsize is not None else getattr(image, "voxelsize", None)
            self.image = OracleImage(image, vs)
        if isinstance(ignoredlabels, (int, np.integer)):
            ignoredlabels = [int(ignoredlabels)]
        self._ignoredlabels = set(int(i) for i in ignoredlabels)
        if background is not None:
            if not isinstance(background, (int, np.integer)):
                raise ValueError("The label you provided as background is not an integer !")
            self._ignoredlabels.add(int(background))
        self._voxelsize = tuple(self.image.voxelsize)
        self._background = background
        self._labels = None
        self._bbox = None
        self._kernels = None
        self._neighbors = None
        self._cell_layer1 = None
        self._center_of_mass = {}
        self.return_type = return_type

    # -- trivial accessors (SIA:273-277, 1195)
    def is3D(self):
        return True

    def background(self):
        return self._background

    def ignoredlabels(self):
        return self._ignoredlabels

    def add2ignoredlabels(self, list2add, verbose=False):  # SIA:279-289
        if isinstance(list2add, (int, np.integer)):
            list2add = [list2add]
        self._ignoredlabels.update(int(i) for i in list2add)
        self._labels = self._compute_labels()

    def consideronlylabels(self, list2consider, verbose=False):  # SIA:291-306
        if isinstance(list2consider, (int, np.integer)):
            list2consider = [list2consider]
        present = set(int(v) for v in np.unique(self.image))
        self._ignoredlabels.update(present - set(int(i) for i in list2consider))
        self._labels = self._compute_labels()

    def convert_return(self, values, labels=None, overide_return_type=None):  # SIA:309-334
        rt = self.return_type if overide_return_type is None else overide_return_type
        if labels is not None and isinstance(labels, (int, np.integer)):
            return values
        if rt == NPLIST:
            return values
        if rt == LIST:
            return values if isinstance(values, list) else values.tolist()
        return dict(zip(labels, values))

    # -- labels (SIA:337-414)
    def _compute_labels(self):
        present = set(int(v) for v in np.unique(self.image))
        return sorted(present - self._ignoredlabels)

    def labels(self):
        if self._labels is None:
            self._labels = self._compute_labels()
        return self._labels

    def nb_labels(self):
        return len(self.labels())

    def label_request(self, labels):
        if isinstance(labels, (int, np.integer)):
            return [int(labels)]
        if isinstance(labels, list):
            return sorted(set(int(l) for l in labels) & set(self.labels()))
        if labels is None:
            return self.labels()
        if isinstance(labels, str):
            key = labels.lower()
            if key == "all":
                return self.labels()
            if key == "l1":
                return self.cell_first_layer()
            if key == "l2":
                return self.cell_second_layer()
            return labels
        raise ValueError("This is not usable as `labels`: {}".format(labels))

    # -- bounding boxes (SIA:483-535, 63-71)
    def boundingbox(self, labels=None, real=False):
        if labels is not None and not isinstance(labels, list) and labels == 0:
            return nd.find_objects(np.asarray(self.image) == 0)[0]
        if self._bbox is None:
            self._bbox = nd.find_objects(np.asarray(self.image))
        if labels is None:
            labels = copy.copy(self.labels())
            if self._background is not None:
                labels.append(self._background)

        def realise(bb):
            return [(s.start * r, s.stop * r) for s, r in zip(bb, self._voxelsize)]

        if isinstance(labels, list):
            boxes = [self._bbox[i - 1] for i in labels]
            if real:
                boxes = [realise(b) for b in boxes]
            return self.convert_return(boxes, labels)
        try:
            bb = self._bbox[labels - 1]
            return realise(bb) if real else bb
        except Exception:
            return None

    # -- volume (SIA:1197-1243); true label ids instead of the int16 cast
    def volume(self, labels=None, real=True):
        labels = self.label_request(labels)
        img = np.asarray(self.image)
        vol = nd.sum(np.ones_like(img), img, index=np.asarray(labels, dtype=np.int64))
        if real:
            vol = np.multiply(vol, self._voxelsize[0] * self._voxelsize[1] * self._voxelsize[2])
        return self.convert_return(vol, labels)

    # -- barycentre (SIA:417-480)
    def center_of_mass(self, labels=None, real=True, verbose=False):
        labels = self.label_request(labels)
        img = np.asarray(self.image)
        center = {}
        for l in labels:
            if l in self._center_of_mass:
                center[l] = self._center_of_mass[l]
                continue
            slices = self.boundingbox(l, real=False)
            if slices is not None:
                crop = img[slices]
                com = np.array(nd.center_of_mass(crop, crop, index=l))
                com = [com[i] + s.start for i, s in enumerate(slices)]
            else:
                com = np.array(nd.center_of_mass(img, img, index=l))
            self._center_of_mass[l] = com
            center[l] = com
        if real:
            center = dict((l, np.multiply(center[l], self._voxelsize)) for l in labels)
        if len(labels) == 1:
            return center[labels[0]]
        return center

    # -- neighbours (SIA:538-693)
    def _shell_neighbors(self, label):
        # crop = bbox grown by one voxel; whole image when no bbox exists (SIA:597-602)
        img = np.asarray(self.image)
        try:
            slices = self.boundingbox(label)
            crop = img[dilate_slices(slices)]
        except Exception:
            crop = img
        return sorted(wall_labels(crop, label))

    def neighbors(self, labels=None, min_contact_area=None, real_area=True, verbose=False):
        if labels is None:
            if self._neighbors is None:
                edges = {}
                boxes = self.boundingbox()
                if self.return_type in (NPLIST, LIST):
                    boxes = dict((i + 1, b) for i, b in enumerate(boxes))
                for label_id in boxes:
                    edges[label_id] = self._shell_neighbors(label_id)
                self._neighbors = edges
            result = self._neighbors
            if min_contact_area is None:
                return result
            return dict((l, self._filter_by_area(l, n, min_contact_area, real_area))
                        for l, n in result.items())
        if not isinstance(labels, list):
            if self._neighbors is not None and labels in self._neighbors:
                neigh = self._neighbors[labels]
            else:
                neigh = self._shell_neighbors(labels)
            if min_contact_area is not None:
                neigh = self._filter_by_area(labels, neigh, min_contact_area, real_area)
            return neigh
        edges = {}
        for label in labels:
            neigh = self._shell_neighbors(label)
            if min_contact_area is not None:
                neigh = self._filter_by_area(label, neigh, min_contact_area, real_area)
            edges[label] = neigh
        return edges

    def _filter_by_area(self, label, neighbors, min_contact_area, real_area):  # SIA:677-693
        areas = self.cell_wall_area(label, list(neighbors), real_area)
        kept = list(neighbors)
        for (i, j), area in areas.items():
            if area < min_contact_area:
                kept.remove(i if j == label else j)
        return kept

    def neighbors_number(self, labels=None, min_contact_area=None, real_area=True, verbose=False):
        nei = self.neighbors(labels, min_contact_area, real_area, verbose)  # SIA:734-742
        if isinstance(nei, dict):
            return dict((k, len(v)) for k, v in nei.items())
        return len(nei)

    # -- wall areas (SIA:695-756, 908-993)
    def neighbor_kernels(self):
        if self._kernels is None:
            self._kernels = directional_kernels()
        return self._kernels

    def get_voxel_face_surface(self):
        a = self._voxelsize
        return np.array([a[1] * a[2], a[2] * a[0], a[0] * a[1]])

    def cell_wall_area(self, label_id, neighbors, real=True):
        face = self.get_voxel_face_surface()
        img = np.asarray(self.image)
        slices = self.boundingbox(label_id)
        crop = img[dilate_slices(slices)] if slices is not None else img
        mask = (crop == label_id)
        single = not isinstance(neighbors, list)
        if single:
            neighbors = [neighbors]
        wall = {}
        for a, kernel in enumerate(self.neighbor_kernels()):
            grown = nd.binary_dilation(mask, structure=kernel)
            frontier = crop[grown & ~mask]
            for n in neighbors:
                nb_pix = int(np.count_nonzero(frontier == n))
                area = float(nb_pix * face[a // 2]) if real else nb_pix
                key = (min(label_id, n), max(label_id, n))
                wall[key] = wall.get(key, 0.0) + area
        if single:
            return next(iter(wall.values()))
        return wall

    def wall_areas(self, neighbors=None, real=True):
        if neighbors is None:
            neighbors = self.neighbors()
        areas = {}
        for label_id, lneigh in neighbors.items():
            higher = [n for n in lneigh if n > label_id]
            if higher:
                for key, val in self.cell_wall_area(label_id, higher, real=real).items():
                    areas[key] = areas.get(key, 0.0) + val
        return areas

    # -- inertia (SIA:123-167, 1246-1341)
    def inertia_axis(self, labels=None, real=True, verbose=False):
        labels = self.label_request(labels)
        img = np.asarray(self.image)
        vecs, vals = [], []
        for label in labels:
            slices = self.boundingbox(label, real=False)
            center = list(copy.copy(self.center_of_mass(label, real=False)))
            if slices is not None:
                for i, s in enumerate(slices):
                    center[i] = center[i] - s.start
                lab_img = (img[slices] == label)
            else:
                lab_img = (img == label)
            xyz = lab_img.nonzero()
            coords = np.array([xyz[0] - center[0], xyz[1] - center[1], xyz[2] - center[2]])
            _, val, vec = covariance_and_axes(coords)
            if real:
                val = np.array(val, dtype=float)
                for i in range(3):
                    val[i] *= np.linalg.norm(np.multiply(vec[i], self._voxelsize))
            vecs.append(vec)
            vals.append(val)
        as_rows = [[v[k] for k in range(len(v))] for v in vecs]  # SIA:191-201
        if len(labels) == 1:
            return as_rows[0], vals[0]
        return self.convert_return(as_rows, labels), self.convert_return(vals, labels)

    reduced_inertia_axis = inertia_axis  # SIA:1295-1341 is computationally identical

    def covariance(self, label):
        """Not a reference method: the 3x3 matrix SIA:1276-1278 builds, for pinning."""
        img = np.asarray(self.image)
        slices = self.boundingbox(label, real=False)
        center = list(self.center_of_mass(label, real=False))
        for i, s in enumerate(slices):
            center[i] = center[i] - s.start
        xyz = (img[slices] == label).nonzero()
        coords = np.array([xyz[0] - center[0], xyz[1] - center[1], xyz[2] - center[2]])
        return covariance_and_axes(coords)[0]

    # -- margins and layers (SIA:996-1022, 1344-1358)
    def labels_at_stack_margins(self, voxel_distance_from_margin=5):
        d = voxel_distance_from_margin
        img = np.asarray(self.image)
        found = set()
        for sl in (np.s_[:d, :, :], np.s_[-d:, :, :], np.s_[:, :d, :], np.s_[:, -d:, :],
                   np.s_[:, :, :d], np.s_[:, :, -d:]):
            found.update(np.unique(img[sl]).tolist())
        return sorted(found - set([self._background]))

    def cell_first_layer(self, filter_by_area=True, minimal_external_area=10, real_area=True):
        if self._cell_layer1 is None:
            self._cell_layer1 = [int(n) for n in self.neighbors(self._background)]
        layer = self._cell_layer1
        if filter_by_area:
            areas = self.cell_wall_area(self._background, list(self._cell_layer1), real_area)
            layer = [l for l in self._cell_layer1
                     if (self._background, l) in areas
                     and areas[(self._background, l)] > minimal_external_area]
        return sorted(set(layer) - self._ignoredlabels)

    def cell_second_layer(self, filter_by_area=True, minimal_L1_area=10, real_area=True):
        l1 = self.cell_first_layer()
        nei = self.neighbors(l1, minimal_L1_area, real_area, False)
        if not isinstance(nei, dict):
            nei = {l1[0]: nei} if l1 else {}
        l2 = set()
        for n in nei.values():
            l2.update(int(v) for v in n)
        return sorted(l2 - set(self._cell_layer1) - self._ignoredlabels)

    def region_boundingbox(self, labels):  # SIA:1361-1396
        if isinstance(labels, list) and len(labels) == 1:
            return self.boundingbox(labels[0])
        if isinstance(labels, (int, np.integer)):
            return self.boundingbox(labels)
        boxes = self.boundingbox(labels)
        if not isinstance(boxes, dict):
            boxes = dict(zip(labels, boxes))
        starts = [min(boxes[c][d].start for c in labels) for d in range(3)]
        stops = [max(boxes[c][d].stop for c in labels) for d in range(3)]
        return tuple(slice(a, b) for a, b in zip(starts, stops))


    def cells_walls_coords(self):  # SIA:883-905, as written: `self.background` is the bound method, not its value
        image = hollow_out_cells(np.asarray(self.image), self.background)
        x, y, z = np.where(image != 0)
        return list(x), list(y), list(z)

    def cells_voxel_layer(self, labels, region_boundingbox=False, single_frame=False):  # SIA:1399-1448, as written
        if isinstance(labels, int):
            labels = [labels]
        if single_frame:
            region_boundingbox = True
        if not isinstance(region_boundingbox, bool):
            if sum([isinstance(s, slice) for s in region_boundingbox]) == 3:
                bbox = region_boundingbox
            else:
                return None
        elif isinstance(region_boundingbox, bool) and region_boundingbox:
            bbox = self.region_boundingbox(labels)
        else:
            bboxes = self.boundingbox(labels, real=False)
        struct = nd.generate_binary_structure(3, 2)
        image = np.asarray(self.image)
        if single_frame:
            vox_layer = np.zeros_like(image[bbox], dtype=int)
        else:
            vox_layer = {}
        for clabel in labels:
            if region_boundingbox:
                bbox_im = image[bbox]
            else:
                bbox_im = image[bboxes[clabel]]
            mask_bbox_im = (bbox_im == clabel)
            eroded_mask_bbox_im = nd.binary_erosion(mask_bbox_im, structure=struct)
            layer = mask_bbox_im ^ eroded_mask_bbox_im      # `mask - eroded` of two boolean arrays (old numpy)
            if single_frame:
                vox_layer += np.array(layer, dtype=int)
            else:
                vox_layer[clabel] = np.array(layer, dtype=int)
        if len(labels) == 1:
            return vox_layer[clabel]
        else:
            return vox_layer

    # -- wall voxels (SIA:759-880, 1049-1111), as written: crops, two masks, 18-connectivity dilations
    def wall_voxels_between_two_cells(self, label_1, label_2):  # SIA:759-806 (bbox given as the dict of boxes)
        b1, b2 = self.boundingbox(label_1), self.boundingbox(label_2)
        if b1 is None or b2 is None:
            return np.zeros((3, 0), dtype=np.int64)
        vol = lambda b: np.prod([s.stop - s.start for s in b])
        box = b1 if vol(b1) < vol(b2) else b2                      # sort_boundingbox, SIA:97-112
        dil = dilate_slices(box)
        crop = np.asarray(self.image)[dil]
        m1, m2 = crop == label_1, crop == label_2
        struct = nd.generate_binary_structure(3, 2)
        d1, d2 = nd.binary_dilation(m1, structure=struct), nd.binary_dilation(m2, structure=struct)
        x, y, z = np.where((d1 & m2) | (d2 & m1))
        return np.array((x + dil[0].start, y + dil[1].start, z + dil[2].start))

    def wall_voxels_per_cell(self, label_1, neighbors=None, neighbors2ignore=()):  # SIA:809-880
        box = self.boundingbox(label_1)
        dil = dilate_slices(dilate_slices(box))
        crop = np.asarray(self.image)[dil]
        m1 = crop == label_1
        struct = nd.generate_binary_structure(3, 2)
        d1 = nd.binary_dilation(m1, structure=struct)
        if neighbors is None:
            neighbors = self.neighbors(label_1)
        if isinstance(neighbors, (int, np.integer)):
            neighbors = [neighbors]
        neighbors = [n for n in neighbors if n not in neighbors2ignore]
        coord = {}
        for label_2 in neighbors:
            m2 = crop == label_2
            d2 = nd.binary_dilation(m2, structure=struct)
            x, y, z = np.where((d1 & m2) | (d2 & m1))
            if len(x):
                coord[min(label_1, label_2), max(label_1, label_2)] = np.array(
                    (x + dil[0].start, y + dil[1].start, z + dil[2].start))
        return coord

    def voxel_first_layer(self, keep_background=True):  # SIA:1024-1046, as written (scipy, 6-neighbourhood)
        if getattr(self, "_voxel_layer1", None) is None:
            img = np.asarray(self.image)
            mask_img_1 = img == self._background
            struct = nd.generate_binary_structure(3, 1)
            dil_1 = nd.binary_dilation(mask_img_1, structure=struct)
            layer = dil_1 ^ mask_img_1                  # `dil_1 - mask_img_1` of two boolean arrays (old numpy)
            out = img * layer + mask_img_1 if keep_background else img * layer
            self._voxel_layer1 = out.astype(img.dtype)
        return self._voxel_layer1

    def surface_area(self, labels=None, real=True):
        """Per-label total surface (SURVEY.md §8 Semantics, last bullet): the sum of the label's wall areas with every
        face neighbour, as the reference computes them one wall at a time (cell_wall_area, SIA:908-959); faces on the
        border of the image belong to no wall and are not counted."""
        single = isinstance(labels, (int, np.integer))
        req = [int(labels)] if single else self.label_request(labels)
        out = []
        for l in req:
            nei = [int(n) for n in self._shell_neighbors(l)]
            areas = self.cell_wall_area(l, nei, real) if nei else {}
            if not isinstance(areas, dict):
                areas = {(min(l, nei[0]), max(l, nei[0])): areas}
            out.append(float(sum(areas.values())))
        return out[0] if single else self.convert_return(out, req)

    def wall_voxels_per_cells_pairs(self, labels=None, neighborhood=None, ignore_background=False,
                                    min_contact_area=None, real_area=True, only_epidermis=False):  # SIA:1049-1111
        compute = neighborhood is None
        if isinstance(labels, list) and isinstance(neighborhood, dict):
            labels = [l for l in labels if l in neighborhood]
        if labels is None and only_epidermis:           # SIA:1062-1065, 1075-1076: only the default label list changes
            labels = [int(v) for v in np.unique(self.voxel_first_layer(True))]
        elif labels is None:
            labels = self.labels()
        elif isinstance(labels, list):
            labels.sort()
        else:
            labels = [labels]
        out = {}
        for label in labels:
            neighbors = list(self.neighbors(label, min_contact_area, real_area)) if compute else list(neighborhood[label])
            keep = labels if ignore_background else labels + [self._background]
            ignore = [n for n in neighbors if n not in keep]
            neighbors = [n for n in neighbors if (min(label, n), max(label, n)) not in out]
            if neighbors:
                out.update(self.wall_voxels_per_cell(label, neighbors, ignore))
        return out

    # -- image mutation (SIA:1114-1176), as written: per-label bounding-box crops, image edited in place;
    # the bounding boxes are the ones cached at construction (the reference never refreshes them)
    def fuse_labels_in_image(self, labels, verbose=False):  # SIA:1114-1136
        assert isinstance(labels, list) and len(labels) >= 2
        assert self._background not in labels
        min_lab = min(labels)
        labels.remove(min_lab)
        for label in labels:
            bbox = self.boundingbox(label)
            if bbox is None:
                continue                                     # "No boundingbox found ..., skipping"
            crop = np.asarray(self.image)[bbox]
            xyz = np.where(crop == label)
            np.asarray(self.image)[tuple(x + b.start for x, b in zip(xyz, bbox))] = min_lab
        return None

    def remove_labels_from_image(self, labels, erase_value=0, verbose=False):  # SIA:1138-1165
        if isinstance(labels, (int, np.integer)):
            labels = [labels]
        try:
            labels.remove(self._background)
        except ValueError:
            pass
        for label in labels:
            bbox = self.boundingbox(label)
            if bbox is None:
                continue
            crop = np.asarray(self.image)[bbox]
            xyz = np.where(crop == label)
            np.asarray(self.image)[tuple(x + b.start for x, b in zip(xyz, bbox))] = erase_value
        self._ignoredlabels.update([erase_value])
        for label in labels:
            self._ignoredlabels.discard(label)

    def remove_stack_margin_labels_from_image(self, erase_value=0, voxel_distance_from_margin=5, verbose=False):
        self.remove_labels_from_image(self.labels_at_stack_margins(voxel_distance_from_margin), erase_value, verbose)


def hollow_out_cells(image, background, remove_background=True):  # SIA:74-95, as written
    b = nd.laplace(image)
    mask = b != 0
    m = image * mask
    if remove_background:
        mask = m != background
        m = m * mask
    return m


def find_wall_median_voxel(array):
    """Index of the exact medoid of a point set: what PlantGL's `pointset_median` computes for the reference's
    `_find_wall_median_voxel` (SIA:1555-1585, <= 100 points; docstring example -> 2).  Plain double loop."""
    a = np.asarray(array, dtype=np.float64)
    if a.shape[0] == 3:
        a = a.T
    best, best_sum = 0, np.inf
    for i in range(a.shape[0]):
        tot = 0.0
        for j in range(a.shape[0]):
            tot += float(np.sqrt(((a[i] - a[j]) ** 2).sum()))
        if tot < best_sum:
            best, best_sum = i, tot
    return best


def property_image(image, property_dict, background, dtype=np.uint16):
    """PropertySpatialImage.create_property_image (PSI:207-221): every label of the image without a
    value gets the background id, so does the background itself; the per-voxel lookup is then cast."""
    values = dict(property_dict)
    for l in np.unique(image):
        if int(l) not in values:
            values[int(l)] = background
    values[background] = background
    img = np.asarray(image)
    out = np.empty(img.shape, dtype=np.float64)
    for l in np.unique(img):
        out[img == l] = values[int(l)]
    with np.errstate(invalid="ignore"):
        return out.astype(dtype)


def full_feature_set(image, voxelsize, background=1, with_inertia=True, with_walls=True,
                     labels=None):
    """The call sequence of temporal_graph_from_image._graph_from_image (TGI:104-191) on one
    image, restricted to the hot-path extractors.  Used by bench.py's cpu_baseline leg and
    by the golden-fixture generator.  Returns a dict of plain numpy / python results."""
    sia = OracleSIA(image, ignoredlabels=0, return_type=DICT, background=background,
                    voxelsize=voxelsize)
    labs = list(sia.labels()) if labels is None else list(labels)
    out = {"labels": labs}
    out["neighbors"] = sia.neighbors(labs)
    out["boundingbox"] = sia.boundingbox(labs, real=False)
    out["volume"] = sia.volume(labs, real=True)
    out["barycenter"] = sia.center_of_mass(labs, real=True)
    out["background_neighbors"] = sia.neighbors(background)
    out["border"] = sia.labels_at_stack_margins()
    if with_inertia:
        out["inertia_axis"], out["inertia_values"] = sia.inertia_axis(labs, real=True)
    if with_walls:
        labelset = set(labs)
        edges = dict((s, [t for t in ts if s < t and t in labelset])
                     for s, ts in out["neighbors"].items())
        out["wall_surface"] = sia.wall_areas(edges, real=True)
    return out
