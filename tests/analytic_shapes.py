"""A labelled volume made of axis-aligned cuboids whose every per-label answer has a closed form -- a pin for the whole
SpatialImageAnalysis surface of the hot path (rows a3-a9 of SURVEY.md §8a) that owes nothing to a restatement of the reference:

  volume        a b c voxels (x the voxel volume when real)                              SIA:1197-1243
  boundingbox   [o_d, o_d + n_d)                                                         SIA:483-535
  barycentre    o + (n - 1) / 2 in voxel units (x voxelsize when real)                   SIA:417-480
  neighbours    two cuboids are face neighbours iff they abut along one axis and their projections on the other two axes
                overlap; the wall between them = the overlap rectangle: its voxel faces all lie on ONE axis, so the area
                is (faces) x (the face area of that axis)                                SIA:538-660, 908-993
  surface       a cuboid's faces with everything else inside the volume (faces on the border of the volume belong to no wall)

The cuboids tile a box completely; the rest of the volume is background (label 1), so every face count is an overlap area.
"""
import itertools

import numpy as np

VOXELSIZE = (0.5, 0.5, 1.0)
BACKGROUND = 1
SHAPE = (20, 26, 300)          # wider than one 256-column tile: walls cross tile edges
BOX = ((2, 3, 10), (18, 23, 290))      # the tiled box [lo, hi): away from the volume faces except where a cut reaches them


def build(dtype=np.uint16):
    """(volume, cells): cells[label] = (origin[3], size[3]).  The box is cut by two planes per axis into 27 cuboids; the
    cuts of different axes are NOT aligned across slabs (staggered), so the walls have T-junctions."""
    vol = np.full(SHAPE, BACKGROUND, dtype=dtype)
    lo, hi = np.asarray(BOX[0]), np.asarray(BOX[1])
    cells = {}
    label = 2
    cuts0 = [lo[0], lo[0] + 5, lo[0] + 11, hi[0]]
    for i in range(3):
        cuts1 = [lo[1], lo[1] + 4 + 2 * i, lo[1] + 13 + i, hi[1]]                  # staggered per slab of axis 0
        for j in range(3):
            cuts2 = [lo[2], lo[2] + 60 + 37 * i + 11 * j, lo[2] + 200 + 20 * j - 9 * i, hi[2]]   # ... and per row of cuboids
            for k in range(3):
                o = np.array([cuts0[i], cuts1[j], cuts2[k]])
                n = np.array([cuts0[i + 1], cuts1[j + 1], cuts2[k + 1]]) - o
                vol[o[0]:o[0] + n[0], o[1]:o[1] + n[1], o[2]:o[2] + n[2]] = label
                cells[label] = (o, n)
                label += 1
    assert (vol[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] != BACKGROUND).all()
    return vol, cells


def overlap_faces(a, b):
    """Per-axis number of voxel faces shared by two cuboids (origin, size): non-zero on at most one axis."""
    (oa, na), (ob, nb) = a, b
    faces = np.zeros(3, dtype=np.int64)
    for d in range(3):
        if oa[d] + na[d] == ob[d] or ob[d] + nb[d] == oa[d]:                       # abut along d
            area = 1
            for e in range(3):
                if e != d:
                    area *= max(0, min(oa[e] + na[e], ob[e] + nb[e]) - max(oa[e], ob[e]))
            faces[d] = area
    return faces


def expected(cells):
    """dict of closed-form answers: volume, bbox, com (voxel units), pair faces {(lo, hi): faces[3]} incl. the background."""
    lo, hi = np.asarray(BOX[0]), np.asarray(BOX[1])
    shape = np.asarray(SHAPE)
    out = dict(volume={}, bbox={}, com={}, faces={})
    for l, (o, n) in cells.items():
        out["volume"][l] = int(n.prod())
        out["bbox"][l] = tuple(slice(int(o[d]), int(o[d] + n[d])) for d in range(3))
        out["com"][l] = o + (n - 1) / 2.0
        # faces with the background: the parts of the cuboid's sides that lie on the side of the tiled box (inside the volume)
        bg = np.zeros(3, dtype=np.int64)
        for d in range(3):
            side = int(np.prod([n[e] for e in range(3) if e != d]))
            if o[d] == lo[d] and lo[d] > 0:
                bg[d] += side
            if o[d] + n[d] == hi[d] and hi[d] < shape[d]:
                bg[d] += side
        if bg.any():
            out["faces"][(BACKGROUND, l)] = bg
    for (la, a), (lb, b) in itertools.combinations(sorted(cells.items()), 2):
        f = overlap_faces(a, b)
        if f.any():
            out["faces"][(la, lb)] = f
    return out


def face_areas(voxelsize):
    vx, vy, vz = voxelsize
    return np.array([vy * vz, vz * vx, vx * vy])


def check(analysis, cells, voxelsize=VOXELSIZE):
    """`analysis`: this package's class or the oracle's, built on `build()`'s volume with background 1, DICT return type."""
    want = expected(cells)
    labels = sorted(cells)
    vol_vox, vol_real = analysis.volume(list(labels), real=False), analysis.volume(list(labels), real=True)
    com = analysis.center_of_mass(list(labels), real=False)
    com_real = analysis.center_of_mass(list(labels), real=True)
    boxes = analysis.boundingbox(list(labels))
    vs = np.asarray(voxelsize, dtype=np.float64)
    for l in labels:
        assert vol_vox[l] == want["volume"][l] and abs(vol_real[l] - want["volume"][l] * vs.prod()) < 1e-9, l
        got_box = boxes[l]
        assert tuple((s.start, s.stop) for s in got_box) == tuple((s.start, s.stop) for s in want["bbox"][l]), l
        assert np.abs(np.asarray(com[l]) - want["com"][l]).max() < 1e-9, l
        assert np.abs(np.asarray(com_real[l]) - want["com"][l] * vs).max() < 1e-9, l
    # neighbours and walls
    nei = analysis.neighbors(list(labels))
    partners = dict((l, set()) for l in labels)
    for (a, b) in want["faces"]:
        if a in partners:
            partners[a].add(b)
        if b in partners:
            partners[b].add(a)
    for l in labels:
        assert set(int(v) for v in nei[l]) == partners[l], (l, sorted(nei[l]), sorted(partners[l]))
    surf = face_areas(voxelsize)
    for (a, b), f in want["faces"].items():
        assert abs(analysis.cell_wall_area(a, b, real=False) - f.sum()) < 1e-9, (a, b)
        assert abs(analysis.cell_wall_area(a, b, real=True) - float(f @ surf)) < 1e-9, (a, b)
    walls = analysis.wall_areas(dict((l, list(nei[l])) for l in labels), real=True)
    cellpairs = dict((k, f) for k, f in want["faces"].items() if k[0] != BACKGROUND)
    got_pairs = dict((k, v) for k, v in walls.items() if k[0] in cells and k[1] in cells)
    assert sorted(got_pairs) == sorted(cellpairs)
    for k, f in cellpairs.items():
        assert abs(got_pairs[k] - float(f @ surf)) < 1e-9, k
    if hasattr(analysis, "surface_area"):
        total = analysis.surface_area(list(labels), real=True)
        for l in labels:
            s = sum(float(f @ surf) for (a, b), f in want["faces"].items() if l in (a, b))
            assert abs(total[l] - s) < 1e-9, l
    first = set(int(v) for v in analysis.neighbors(BACKGROUND))
    assert first & set(labels) == set(b for (a, b) in want["faces"] if a == BACKGROUND)
