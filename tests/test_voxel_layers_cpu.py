"""cells_voxel_layer (SIA:1399-1448) host logic on the CPU: the per-voxel "one of my 18 neighbours carries another label"
image the device delivers (`ta_volume_layer18`) is injected from a numpy brute force; what the class builds from it --
crops, the faces of the crop, the single frame -- against the oracle's erosions.  Also pins the oracle's restatement
of hollow_out_cells to the closed form of scipy's integer Laplacian the kernel implements."""
import numpy as np
import pytest

from oracle import onepass, sia_oracle
from oracle.sia_oracle import OracleSIA
from tissue_analysis_amd import DICT, LIST, NPLIST, Extraction, SpatialImage, SpatialImageAnalysis3D

from helpers import WALL_OFFSETS, random_blocks, voronoi


def brute_layer18(vol):
    out = np.zeros(vol.shape, dtype=np.uint8)
    n0, n1, n2 = vol.shape
    for a, b, c in WALL_OFFSETS:
        src = (slice(max(0, -a), n0 - max(0, a)), slice(max(0, -b), n1 - max(0, b)), slice(max(0, -c), n2 - max(0, c)))
        dst = (slice(max(0, a), n0 - max(0, -a)), slice(max(0, b), n1 - max(0, -b)), slice(max(0, c), n2 - max(0, -c)))
        out[src] |= (vol[src] != vol[dst]).astype(np.uint8)
    return out


def brute_hollow(vol, background, bits=None):
    """Six face neighbours minus 6 v modulo 2^bits (the image's own width), the edge voxel repeated outside."""
    bits = 8 * vol.dtype.itemsize if bits is None else bits
    p = np.pad(vol.astype(object), 1, mode="edge")
    lap = (p[:-2, 1:-1, 1:-1] + p[2:, 1:-1, 1:-1] + p[1:-1, :-2, 1:-1] + p[1:-1, 2:, 1:-1] + p[1:-1, 1:-1, :-2]
           + p[1:-1, 1:-1, 2:] - 6 * p[1:-1, 1:-1, 1:-1])
    keep = np.array([[[int(x) % (1 << bits) != 0 for x in row] for row in plane] for plane in lap])
    out = vol * keep
    if background is not None:
        out = out * (out != background)
    return out.astype(vol.dtype)


def both(vol, **kw):
    x = Extraction.from_arrays(vol.shape, onepass.extract(vol))
    sia = SpatialImageAnalysis3D(SpatialImage(vol, voxelsize=(1., 1., 1.)), return_type=DICT, extraction=x, **kw)
    sia._voxel_layer18 = brute_layer18(vol)
    return sia, OracleSIA(vol, return_type=sia_oracle.DICT, **kw)


def same(got, want):
    if isinstance(want, dict):
        assert sorted(got) == sorted(want)
        for k in want:
            same(got[k], want[k])
    else:
        assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("make", [lambda: voronoi((18, 16, 22), 14, 81, np.uint16),
                                  lambda: random_blocks((9, 11, 13), 10, 82, np.uint32)], ids=["voronoi", "blocks"])
def test_cells_voxel_layer_against_the_erosions(make):
    vol = make()
    sia, ref = both(vol, ignoredlabels=0, background=1)
    labels = [l for l in ref.labels()][:6]
    same(sia.cells_voxel_layer(list(labels)), ref.cells_voxel_layer(list(labels)))                       # own boxes
    same(sia.cells_voxel_layer(labels[2]), ref.cells_voxel_layer(labels[2]))                             # one label: its array
    same(sia.cells_voxel_layer(list(labels), region_boundingbox=True), ref.cells_voxel_layer(list(labels), region_boundingbox=True))
    same(sia.cells_voxel_layer(list(labels), single_frame=True), ref.cells_voxel_layer(list(labels), single_frame=True))
    cut = (slice(2, 8), slice(1, 9), slice(3, 10))                                                      # a crop that cuts cells
    same(sia.cells_voxel_layer(list(labels), region_boundingbox=cut), ref.cells_voxel_layer(list(labels), region_boundingbox=cut))
    assert sia.cells_voxel_layer(list(labels), region_boundingbox=(1, 2, 3)) is None


def test_cells_voxel_layer_and_region_boundingbox_do_not_depend_on_the_return_type():
    """Under NPLIST `boundingbox(list)` answers with an [n, 6] array; the methods that CROP the image with bounding boxes must
    still get slices (an [n, 6] row used as an index fancy-indexes axis 0: wrong masks, no error)."""
    vol = voronoi((18, 16, 22), 14, 81, np.uint16)
    want_sia, _ = both(vol, ignoredlabels=0, background=1)
    labels = want_sia.labels()[:5]
    for rt in (NPLIST, LIST):
        x = Extraction.from_arrays(vol.shape, onepass.extract(vol))
        sia = SpatialImageAnalysis3D(SpatialImage(vol, voxelsize=(1., 1., 1.)), return_type=rt, extraction=x, ignoredlabels=0, background=1)
        sia._voxel_layer18 = brute_layer18(vol)
        assert sia.region_boundingbox(list(labels)) == want_sia.region_boundingbox(list(labels))
        same(sia.cells_voxel_layer(list(labels)), want_sia.cells_voxel_layer(list(labels)))
        same(sia.cells_voxel_layer(list(labels), region_boundingbox=True), want_sia.cells_voxel_layer(list(labels), region_boundingbox=True))
        same(sia.cells_voxel_layer(list(labels), single_frame=True), want_sia.cells_voxel_layer(list(labels), single_frame=True))


def test_the_oracle_hollow_is_the_modular_laplacian():
    rng = np.random.default_rng(83)
    for dtype in (np.uint8, np.uint16, np.uint32):
        top = np.iinfo(dtype).max
        vol = rng.integers(0, 4, size=(5, 6, 7)).astype(dtype)
        vol[rng.random(vol.shape) < 0.3] = top                                # sums that wrap
        vol[2, 2, 2:5] = [9, 10, 11]                                          # neighbours that cancel: 9 + 11 - 2 * 10
        for bg in (None, 0, 3):
            want = brute_hollow(vol, bg)
            got = sia_oracle.hollow_out_cells(vol, bg, remove_background=bg is not None)
            assert got.dtype == vol.dtype and np.array_equal(got, want), (dtype, bg)
