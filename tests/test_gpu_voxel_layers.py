"""hollow_out_cells (SIA:74-95), cells_walls_coords (SIA:883-905) and cells_voxel_layer (SIA:1399-1448) through the HIP
stencils (`ta_volume_hollow`, `ta_volume_layer18`) against the oracle's scipy restatements."""
import numpy as np
import pytest

from oracle import sia_oracle
from oracle.sia_oracle import OracleSIA
from tissue_analysis_amd import DICT, SpatialImage, SpatialImageAnalysis, hollow_out_cells

from helpers import random_blocks, voronoi
from test_voxel_layers_cpu import brute_layer18, same

pytestmark = pytest.mark.gpu

MAKERS = [
    lambda: voronoi((30, 28, 70), 30, 91, np.uint16),
    lambda: np.asfortranarray(voronoi((22, 26, 40), 20, 92, np.uint32)),
    lambda: random_blocks((9, 11, 23), 12, 93, np.uint16),
    lambda: voronoi((5, 9, 260), 14, 94, np.uint32),
    lambda: voronoi((3, 4, 1), 3, 95, np.uint16),
]
IDS = ["voronoi_u16", "voronoi_u32_fortran", "blocks", "long_rows", "one_column"]


@pytest.mark.parametrize("make", MAKERS, ids=IDS)
def test_hollow_out_cells_and_walls_coords(make):
    vol = make()
    for bg, remove in ((1, True), (1, False), (0, True), (70000, True), (None, True)):
        got = hollow_out_cells(SpatialImage(vol, voxelsize=(1., 1., 1.)), bg, remove_background=remove, verbose=False)
        want = sia_oracle.hollow_out_cells(vol, bg, remove_background=remove)
        assert np.asarray(got).dtype == vol.dtype and np.array_equal(np.asarray(got), want), (bg, remove)
    sia = SpatialImageAnalysis(SpatialImage(vol, voxelsize=(1., 1., 1.)), ignoredlabels=0, return_type=DICT, background=1)
    ref = OracleSIA(vol, ignoredlabels=0, return_type=sia_oracle.DICT, background=1)
    got, want = sia.cells_walls_coords(), ref.cells_walls_coords()
    assert len(got) == 3 and all(isinstance(g, list) for g in got)
    for g, w in zip(got, want):
        assert np.array_equal(np.asarray(g), np.asarray(w))


def test_hollow_on_other_integer_types_and_wrapping_sums():
    rng = np.random.default_rng(96)
    for dtype in (np.uint8, np.uint16, np.uint32, np.int32, np.int64):
        # unsigned types wrap in scipy as in the kernel; a SIGNED type whose per-axis sums leave its range is cast by scipy
        # the way the C compiler pleases (x86: 0x80000000) -- labels near 2^31 in an int32 image are not a case to match
        top = {np.uint8: 255, np.uint16: 65535, np.uint32: (1 << 32) - 1, np.int32: (1 << 28), np.int64: (1 << 31) - 1}[dtype]
        vol = rng.integers(0, 5, size=(7, 9, 33)).astype(dtype)
        vol[rng.random(vol.shape) < 0.3] = top
        vol[3, 3, 10:13] = [9, 10, 11]
        got = hollow_out_cells(vol, 2, verbose=False)
        want = sia_oracle.hollow_out_cells(vol, 2)
        assert got.dtype == want.dtype and np.array_equal(got, want), dtype
    flat = rng.integers(1, 6, size=(12, 17)).astype(np.uint16)                 # a 2-D image: no third axis in the sum
    assert np.array_equal(hollow_out_cells(flat, 1, verbose=False), sia_oracle.hollow_out_cells(flat, 1))


@pytest.mark.parametrize("make", MAKERS[:4], ids=IDS[:4])
def test_cells_voxel_layer(make):
    vol = make()
    sia = SpatialImageAnalysis(SpatialImage(vol, voxelsize=(1., 1., 1.)), ignoredlabels=0, return_type=DICT, background=1)
    ref = OracleSIA(vol, ignoredlabels=0, return_type=sia_oracle.DICT, background=1)
    assert np.array_equal(sia._layer18(), brute_layer18(np.ascontiguousarray(vol)))
    labels = ref.labels()[:8]
    same(sia.cells_voxel_layer(list(labels)), ref.cells_voxel_layer(list(labels)))
    same(sia.cells_voxel_layer(labels[1]), ref.cells_voxel_layer(labels[1]))
    same(sia.cells_voxel_layer(list(labels), region_boundingbox=True), ref.cells_voxel_layer(list(labels), region_boundingbox=True))
    same(sia.cells_voxel_layer(list(labels), single_frame=True), ref.cells_voxel_layer(list(labels), single_frame=True))
