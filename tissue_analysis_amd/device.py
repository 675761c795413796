"""Device-memory plumbing on top of torch (allocation, streams) for volumes that never touch the
host: synthetic workloads and Z-slabs.  torch is plumbing here; all compute goes through the C ABI."""
from __future__ import annotations

import numpy as np

from . import _capi, synth

_TORCH_DTYPE = {"uint16": "int16", "uint32": "int32"}


def torch_context(device=0):
    """A C-ABI context that launches on torch's current stream of `device` (torch's default stream is the
    device's legacy null stream, handle 0: Context.set_stream maps that to TA_STREAM_LEGACY_DEFAULT)."""
    import torch
    torch.cuda.set_device(device)
    ctx = _capi.Context(device)
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    return ctx


def empty_volume(planes, dims, dtype, device=0):
    import torch
    tdt = getattr(torch, _TORCH_DTYPE[np.dtype(dtype).name])
    # a view of a storage with 16 elements to spare: the sweep may then use 16-byte loads whatever the row length (the
    # strip that straddles the end of the last row reads a few bytes past the volume; _capi.Context.set_volume_device
    # sees the slack in the tensor's storage)
    n = int(planes) * int(dims[1]) * int(dims[2])
    flat = torch.empty((n + 16,), dtype=tdt, device="cuda:%d" % device)
    return flat[:n].view(int(planes), int(dims[1]), int(dims[2]))


def synth_slab(ctx, dims, dtype, n_cells, seed, a_begin=0, a_end=None, device=0, ellipsoid=True):
    """Generate planes [a_begin, a_end) of the synthetic Voronoi volume directly in HBM.
    Returns a torch tensor (int16/int32 storage holding the uint16/uint32 labels)."""
    a_end = int(dims[0]) if a_end is None else int(a_end)
    seeds, grid = synth.make_seeds(dims, n_cells, seed)
    ell = synth.ellipsoid_tables(dims) if ellipsoid else None
    t = empty_volume(a_end - a_begin, dims, dtype, device)
    ctx.synth_voronoi(t.data_ptr(), dtype, dims, a_begin, a_end - a_begin, seeds, grid, ell)
    return t, seeds.shape[0] + 1     # tensor, max label
