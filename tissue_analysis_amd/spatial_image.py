"""Minimal stand-in for ``openalea.image.serial.basics.SpatialImage`` (imported by the reference at
spatial_image_analysis.py:27, not vendored): an ndarray that carries ``voxelsize`` and ``info``.
2D arrays are presented as (X, Y, 1) like the original."""
from __future__ import annotations

import numpy as np


class SpatialImage(np.ndarray):
    def __new__(cls, input_array, voxelsize=None, info=None, **kwargs):
        a = np.asarray(input_array)
        if a.ndim == 2:
            a = a[:, :, None]
        if a.ndim != 3:
            raise ValueError("SpatialImage expects a 2D or 3D array")
        obj = a.view(cls)
        if voxelsize is None:
            voxelsize = getattr(input_array, "voxelsize", None)
        if voxelsize is None:
            voxelsize = (1.0,) * a.ndim
        voxelsize = tuple(float(v) for v in voxelsize)
        if len(voxelsize) == 2:
            voxelsize = voxelsize + (1.0,)
        obj.voxelsize = voxelsize
        obj.info = dict(info if info is not None else getattr(input_array, "info", {}) or {})
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.voxelsize = getattr(obj, "voxelsize", (1.0,) * self.ndim)
        self.info = getattr(obj, "info", {})

    @property
    def resolution(self):
        return self.voxelsize
