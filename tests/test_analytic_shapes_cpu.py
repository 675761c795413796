"""Volumes, bounding boxes, barycentres, neighbours, wall areas and surface areas against CLOSED FORMS on a box tiled by
27 staggered cuboids (tests/analytic_shapes.py): the oracle's restatement of the reference and the product's host code on
exact accumulators must both give them -- together with tests/analytic_inertia.py this pins every row of SURVEY.md §8(a)
independently of any restatement."""
import numpy as np

import analytic_shapes
from oracle import onepass, sia_oracle
from oracle.sia_oracle import OracleSIA
from tissue_analysis_amd import DICT, Extraction, SpatialImage, SpatialImageAnalysis3D


def test_the_cuboids_tile_the_box_and_the_closed_forms_are_consistent():
    vol, cells = analytic_shapes.build()
    want = analytic_shapes.expected(cells)
    assert len(cells) == 27 and sum(want["volume"].values()) == int((vol != 1).sum())
    # every cuboid face inside the volume is shared with exactly one other label: faces add up to the cuboids' surfaces
    for l, (o, n) in cells.items():
        total = sum(f.sum() for (a, b), f in want["faces"].items() if l in (a, b))
        assert total == 2 * (n[0] * n[1] + n[1] * n[2] + n[0] * n[2]), l


def test_oracle_gives_the_closed_forms():
    vol, cells = analytic_shapes.build()
    ref = OracleSIA(vol, ignoredlabels=0, return_type=sia_oracle.DICT, background=1, voxelsize=analytic_shapes.VOXELSIZE)
    analytic_shapes.check(ref, cells)


def test_host_code_on_exact_accumulators_gives_the_closed_forms():
    vol, cells = analytic_shapes.build()
    x = Extraction.from_arrays(vol.shape, onepass.extract(vol))
    sia = SpatialImageAnalysis3D(SpatialImage(vol, voxelsize=analytic_shapes.VOXELSIZE), ignoredlabels=0, return_type=DICT,
                                 background=1, extraction=x)
    analytic_shapes.check(sia, cells)
    # and the integer accumulators themselves, straight from the closed forms
    want = analytic_shapes.expected(cells)
    got = dict(zip(zip(x.pair_lo.tolist(), x.pair_hi.tolist()), x.pair_faces.tolist()))
    assert sorted(got) == sorted(want["faces"])
    for k, f in want["faces"].items():
        assert got[k] == f.tolist(), k
