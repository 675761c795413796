"""Multi-GPU: Z-slab partition with a one-plane low halo, per-label reduce, adjacency merge
(SURVEY.md §8e).  One process per GPU; collectives go through torch.distributed (backend "nccl"
is RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

Every per-label quantity is a commutative reduction of exact integers, so the exchange step is:
  sums  int64[L+1][10]  all-reduce SUM   (count, 3 first moments, 6 second moments)
  boxes int32[L+1][6]   all-reduce MIN   (min0..2, -max0..2; INT32_MAX when absent)
  pairs                 all-gather of each rank's unique (key, faces[3]) list, then a hash merge
                        (the same pair shows up on two ranks when a wall crosses a slab boundary).
A face is owned by the slab that owns its higher voxel along axis 0, so each slab carries ONE halo
plane on its low side and no face is counted twice.
"""
from __future__ import annotations

import numpy as np

INT32_MAX = np.iinfo(np.int32).max
EMPTY_KEY = -1          # 0xFFFF...F as int64: padding entries of gathered pair lists


def slab_range(n0, world, rank):
    """Planes [lo, hi) of axis 0 owned by `rank` (contiguous, sizes differ by at most one)."""
    base, rem = divmod(int(n0), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ------------------------------------------------------------------ device layout <-> ABI arrays
def to_device_layout(arrays):
    """(sums int64[L+1,10], boxes int32[L+1,6]) from the host-getter arrays."""
    L1 = arrays["count"].shape[0]
    sums = np.zeros((L1, 10), dtype=np.int64)
    sums[:, 0] = arrays["count"].astype(np.int64)
    sums[:, 1:4] = arrays["sum1"].astype(np.int64)
    sums[:, 4:10] = arrays["sum2"].astype(np.int64)
    bb = arrays["bbox"]
    present = bb[:, 0] >= 0
    boxes = np.full((L1, 6), INT32_MAX, dtype=np.int32)
    boxes[present, :3] = bb[present, :3]
    boxes[present, 3:] = -(bb[present, 3:] - 1)
    return sums, boxes


def from_device_layout(sums, boxes):
    sums = np.asarray(sums)
    boxes = np.asarray(boxes)
    present = boxes[:, 0] != INT32_MAX
    bbox = np.full(boxes.shape, -1, dtype=np.int32)
    bbox[present, :3] = boxes[present, :3]
    bbox[present, 3:] = -boxes[present, 3:] + 1
    return dict(count=sums[:, 0].astype(np.uint64), bbox=bbox, sum1=sums[:, 1:4].astype(np.uint64),
                sum2=sums[:, 4:10].astype(np.uint64))


# ------------------------------------------------------------------ collectives (device-agnostic)
def allreduce_accumulators(sums, boxes, group=None):
    """In-place per-label reduce of torch tensors (int64 SUM, int32 MIN)."""
    import torch.distributed as dist
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(boxes, op=dist.ReduceOp.MIN, group=group)


def allgather_pairs(keys, faces, group=None):
    """All-gather variable-length pair lists.  keys int64[n], faces int64[n,3] torch tensors.
    Returns (keys_all int64[world*m], faces_all int64[world*m,3], m) padded with EMPTY_KEY."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([keys.shape[0]], dtype=torch.int64, device=keys.device)
    sizes = torch.zeros(world, dtype=torch.int64, device=keys.device)
    dist.all_gather_into_tensor(sizes, n, group=group)
    m = int(sizes.max().item())                      # one host sync for all ranks' counts
    kpad = torch.full((max(m, 1),), EMPTY_KEY, dtype=torch.int64, device=keys.device)
    fpad = torch.zeros((max(m, 1), 3), dtype=torch.int64, device=keys.device)
    kpad[:keys.shape[0]] = keys
    fpad[:keys.shape[0]] = faces
    kall = torch.empty((world * max(m, 1),), dtype=torch.int64, device=keys.device)
    fall = torch.empty((world * max(m, 1), 3), dtype=torch.int64, device=keys.device)
    dist.all_gather_into_tensor(kall, kpad, group=group)
    dist.all_gather_into_tensor(fall, fpad, group=group)
    return kall, fall, max(m, 1)


class SlabJob(object):
    """One rank's share of a slab-partitioned extraction, device-resident end to end.

    step() = fused sweep of the local slab (+ halo) into torch-owned accumulators bound to the
    C-ABI context, then (world > 1) the RCCL reduce and the adjacency merge.  After step() every
    rank holds the global per-label rows; rank-local `ctx.adjacency()` holds the global pairs.
    """

    def __init__(self, ctx, vol_tensor, itemsize, a_origin, has_low_halo, max_label, features,
                 group=None, device=0):
        import torch
        self.ctx, self.vol, self.group = ctx, vol_tensor, group
        self.has_low_halo = bool(has_low_halo)
        self.max_label, self.features = int(max_label), features
        dev = "cuda:%d" % device
        self.sums = torch.zeros((self.max_label + 1, 10), dtype=torch.int64, device=dev)
        self.boxes = torch.zeros((self.max_label + 1, 6), dtype=torch.int32, device=dev)
        ctx.set_volume_device(vol_tensor.data_ptr(), itemsize, vol_tensor.shape, a0_origin=a_origin,
                              has_low_halo=has_low_halo, keep=vol_tensor)
        ctx.bind_accumulators(self.sums.data_ptr(), self.boxes.data_ptr(), self.max_label,
                              keep=(self.sums, self.boxes))
        self._torch = torch

    def owned_view(self):
        return self.vol[1:] if self.has_low_halo else self.vol

    def step(self):
        from . import _capi
        torch = self._torch
        self.ctx.extract(self.features, self.max_label)
        if self.group is None:
            return
        allreduce_accumulators(self.sums, self.boxes, self.group)
        if _capi.feature_mask(self.features) & _capi.F_ADJACENCY:
            _, _, n = self.ctx.adjacency_device()          # drains the stream, validates the flags
            keys = torch.empty((max(n, 1),), dtype=torch.int64, device=self.sums.device)
            faces = torch.empty((max(n, 1), 3), dtype=torch.int64, device=self.sums.device)
            self.ctx.adjacency_export(keys.data_ptr(), faces.data_ptr(), max(n, 1))
            kall, fall, m = allgather_pairs(keys[:n], faces[:n], self.group)
            import torch.distributed as dist
            rank = dist.get_rank(self.group)
            # drop this rank's own block (already in the local list), merge the others
            kall[rank * m:(rank + 1) * m] = EMPTY_KEY
            # With nccl (RCCL) the current stream -- which the context launches on -- already waits for
            # the collective; gloo moves device tensors through the host on its own streams.
            if dist.get_backend(self.group) != "nccl":
                torch.cuda.synchronize()
            self.ctx.adjacency_merge(kall.data_ptr(), fall.data_ptr(), kall.shape[0])

    def result_counts(self):
        self._torch.cuda.synchronize()
        return self.sums[:, 0].cpu().numpy()

    def result_arrays(self):
        """Global result in the host-getter layout (memory-axis order)."""
        self._torch.cuda.synchronize()
        out = from_device_layout(self.sums.cpu().numpy(), self.boxes.cpu().numpy())
        out["max_label"] = self.max_label
        from . import _capi
        if _capi.feature_mask(self.features) & _capi.F_ADJACENCY:
            lo, hi, faces = self.ctx.adjacency()
        else:
            lo = hi = np.zeros(0, dtype=np.uint32)
            faces = np.zeros((0, 3), dtype=np.uint64)
        out.update(pair_lo=lo, pair_hi=hi, pair_faces=faces)
        return out
