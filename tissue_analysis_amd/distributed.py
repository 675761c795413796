"""Multi-GPU: Z-slab partition with a one-plane low halo, per-label reduce, adjacency merge
(SURVEY.md §8e).  One process per GPU; collectives go through torch.distributed (backend "nccl"
is RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

Every per-label quantity is a commutative reduction of exact integers, so the exchange step is:
  sums  int64[L+1][10]  all-reduce SUM   (count, 3 first moments, 6 second moments)
  boxes int32[L+1][6]   all-reduce MIN   (min0..2, -max0..2; INT32_MAX when absent)
  pairs                 all-gather of each rank's unique (key, faces[3]) list, then a hash merge
                        (the same pair shows up on two ranks when a wall crosses a slab boundary).
A face is owned by the slab that owns its higher voxel along axis 0, so each slab carries ONE halo
plane on its low side and no face is counted twice.  A rank whose slab arrived without that plane gets
it from its neighbour's memory: exchange_low_halo / attach_low_halo (isend/irecv of one plane).
"""
from __future__ import annotations

import numpy as np

INT32_MAX = np.iinfo(np.int32).max
EMPTY_KEY = -1          # 0xFFFF...F as int64: padding entries of gathered pair lists


def slab_range(n0, world, rank, cuts=None):
    """Planes [lo, hi) of axis 0 owned by `rank`: contiguous; equal plane counts (sizes differ by at most one) unless
    `cuts` -- world + 1 plane indices from balanced_cuts -- says otherwise."""
    if cuts is not None:
        return int(cuts[rank]), int(cuts[rank + 1])
    base, rem = divmod(int(n0), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# What a plane costs the sweep, in voxel-equivalents: its voxels stream at the HBM rate, every label change along the fast
# axis is a record the sweep has to consume.  One record ~ EVENT_VOXELS voxels: C4, 1024^3 -- 0.73 ms of stream for 2^30
# voxels, +0.36 ms for its 24.7 M run records and the faces that come with them (profiles/r04_NOTES.md) -- 0.36 / 24.7e6
# against 0.73 / 2^30.
EVENT_VOXELS = 21.0


def plane_costs(events, plane_voxels, event_voxels=EVENT_VOXELS):
    """float64[n0]: the cost model above for every plane of axis 0, from its count of label changes along the fast axis
    (Context.plane_events() of a resident volume, or any estimate of it: a previous volume of a time series, a host-side
    sample)."""
    return float(plane_voxels) + float(event_voxels) * np.asarray(events, dtype=np.float64)


def balanced_cuts(costs, world):
    """world + 1 plane indices 0 = c[0] < c[1] < ... < c[world] = n0 that cut axis 0 into contiguous slabs of about equal
    COST (the slowest slab sets the step: equal plane counts leave the emptier end slabs of a tissue waiting for the
    middle ones).  Slab r ends at the first plane where the running cost reaches (r + 1) / world of the total; every slab
    keeps at least one plane."""
    costs = np.asarray(costs, dtype=np.float64)
    n0, world = int(costs.shape[0]), int(world)
    if world < 1 or n0 < world:
        raise ValueError("cannot cut %d planes into %d slabs" % (n0, world))
    run = np.cumsum(costs)
    cuts = [0]
    for r in range(1, world):
        c = int(np.searchsorted(run, run[-1] * r / world, side="left")) + 1       # planes [0, c) hold >= r / world of the cost
        # (closer of the two neighbouring cut positions to the target)
        if c - 1 > cuts[-1] and abs(run[c - 2] - run[-1] * r / world) < abs(run[c - 1] - run[-1] * r / world):
            c -= 1
        cuts.append(min(max(c, cuts[-1] + 1), n0 - (world - r)))
    cuts.append(n0)
    return cuts


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


def from_device_layout(sums, boxes, second_moments=True):
    """Device rows -> the host getters' arrays.  Without TA_F_MOMENT2 the second-moment columns of the device rows are not
    defined (the rare table-spill path of the sweep leaves cross terms there): they are answered as zero, like
    ta_get_labels does."""
    sums = np.asarray(sums)
    boxes = np.asarray(boxes)
    present = boxes[:, 0] != INT32_MAX
    bbox = np.full(boxes.shape, -1, dtype=np.int32)
    bbox[present, :3] = boxes[present, :3]
    bbox[present, 3:] = -boxes[present, 3:] + 1
    sum2 = sums[:, 4:10].astype(np.uint64) if second_moments else np.zeros((sums.shape[0], 6), dtype=np.uint64)
    return dict(count=sums[:, 0].astype(np.uint64), bbox=bbox, sum1=sums[:, 1:4].astype(np.uint64), sum2=sum2)


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


def union_of_ids(ids, group=None, device=None):
    """COLLECTIVE: the ascending union of every rank's label ids (numpy, e.g. `ctx.label_census()[1]` of the rank's slab): the
    table all ranks of a partitioned volume with SPARSE ids compact with (`SlabJob(ids=...)`), so that an id has the same row
    everywhere.  A few thousand ids per rank: one padded all-gather."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    on = device if (device is not None and dist.get_backend(group) == "nccl") else "cpu"
    mine = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64)).to(on)
    sizes = torch.zeros(world, dtype=torch.int64, device=on)
    dist.all_gather_into_tensor(sizes, torch.tensor([mine.numel()], dtype=torch.int64, device=on), group=group)
    m = max(int(sizes.max().item()), 1)
    pad = torch.full((m,), -1, dtype=torch.int64, device=on)
    pad[:mine.numel()] = mine
    every = torch.empty((world * m,), dtype=torch.int64, device=on)
    dist.all_gather_into_tensor(every, pad, group=group)
    every = every.cpu().numpy()
    return np.unique(every[every >= 0]).astype(np.uint32)


def sums_shard_rows(nrows, world):
    """Rows of the per-label sums every rank keeps after a reduce-scatter: the table is padded to world x this."""
    return -(-int(nrows) // int(world))


def reduce_scatter_sums(sums, shard, group=None):
    """The per-label SUM as a reduce-scatter: `sums` int64[world * S, 10] (this rank's rows, zero padded) -> `shard`
    int64[S, 10], the GLOBAL rows [rank * S, (rank + 1) * S).  Half the bytes of an all-reduce on every link -- nobody needs
    all sums everywhere during a step (only the boxes decide which pairs travel); results gather the shards when asked for
    (allgather_sums).  gloo has no reduce-scatter: there (CPU tests) an all-reduce of a copy, then the rank's rows."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    S = shard.shape[0]
    if dist.get_backend(group) == "nccl":
        dist.reduce_scatter_tensor(shard, sums, op=dist.ReduceOp.SUM, group=group)
    else:
        tmp = sums.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
        shard.copy_(tmp[rank * S:(rank + 1) * S])
    return shard


def allgather_sums(shard, nrows, group=None):
    """COLLECTIVE: the global int64[nrows, 10] table from the ranks' shards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    full = torch.empty((world * shard.shape[0],) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(full, shard.contiguous(), group=group)
    return full[:nrows]


# ------------------------------------------------------------------ halo hand-off between neighbours
def _plane_bytes(t):
    """A contiguous plane (or stack of planes) as a flat uint8 tensor: what travels is bytes, whatever the label dtype."""
    import torch
    return t.contiguous().view(-1).view(torch.uint8)


def exchange_low_halo(buf, group=None):
    """SURVEY.md §8e, "a single P2P copy from the neighbour GPU": every rank but the last SENDS its last owned plane
    (`buf[-1]`) to rank+1, every rank but the first RECEIVES plane lo-1 into `buf[0]` -- the spare low plane of its slab
    buffer.  COLLECTIVE over `group`.  With backend "nccl" (RCCL) the planes go device to device over xGMI on the
    current stream; gloo cannot move device tensors point to point, so there (CPU tests, two ranks sharing one GPU) the
    plane is staged through the host.  `buf` is [planes (+1 on ranks > 0), n1, n2], device or CPU tensor."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1:
        return buf
    if buf.shape[0] < (2 if rank > 0 else 1):
        raise ValueError("every rank must own at least one plane of axis 0 (rank %d holds none)" % rank)
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    staged = buf.is_cuda and dist.get_backend(group) != "nccl"
    send = recv = None
    ops = []
    if rank + 1 < world:
        send = _plane_bytes(buf[-1])
        if staged:
            send = send.cpu()
        ops.append(dist.P2POp(dist.isend, send, peer(rank + 1), group))
    if rank > 0:
        recv = torch.empty_like(_plane_bytes(buf[0]), device="cpu" if staged else buf.device)
        ops.append(dist.P2POp(dist.irecv, recv, peer(rank - 1), group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if recv is not None:
        _plane_bytes(buf[0]).copy_(recv)          # (buf[0] is contiguous: the view aliases the plane)
    return buf


def attach_low_halo(owned, group=None):
    """For a rank that holds ONLY its own planes [lo, hi) in device memory: a slab buffer with one spare low plane
    (ranks > 0), the owned planes copied behind it, and plane lo-1 fetched from the neighbour (exchange_low_halo).
    Returns (buffer, has_low_halo) -- the arguments SlabJob takes.  COLLECTIVE over `group`."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1 or rank == 0:
        buf, halo = owned.contiguous(), False
    else:
        buf = torch.empty((owned.shape[0] + 1,) + tuple(owned.shape[1:]), dtype=owned.dtype, device=owned.device)
        buf[1:].copy_(owned)
        halo = True
    if world > 1:
        exchange_low_halo(buf, group)
    return buf, halo


def slab_exclusive(boxes, lo, hi):
    """bool[L+1]: labels whose GLOBAL axis-0 extent lies inside planes [lo, hi - 2] of this rank's slab [lo, hi) -- no
    other rank's planes, nor its halo, can hold them (the rule of ta_adjacency_pack_shared, on the reduced device-layout
    boxes int32[L+1][6] = min0..2, -max0..2)."""
    boxes = np.asarray(boxes)
    mn, mx = boxes[:, 0].astype(np.int64), -boxes[:, 3].astype(np.int64)
    return (boxes[:, 0] != INT32_MAX) & (mn >= lo) & (mx <= hi - 2)


class SlabJob(object):
    """One rank's share of a slab-partitioned extraction, device-resident end to end.

    step() = fused sweep of the local slab (+ halo) into torch-owned accumulators bound to the
    C-ABI context, then (world > 1) the RCCL reduce and the adjacency exchange.  After step() every
    rank holds the global per-label rows.

    Only the pairs a slab face can split travel (ta_adjacency_pack_shared, on the reduced boxes): after step() the
    context of each rank holds its PRIVATE pairs plus all ranks' travelling pairs merged -- a PARTIAL list
    (`ctx.adjacency_scope() == ADJ_PARTIAL`; `ctx.adjacency()` refuses to hand it out unless asked with
    `allow_partial=True`).  The GLOBAL list is assembled by result_arrays() -- collective -- which gathers the private
    lists once, when the global list is actually asked for; a step therefore does not include that gather (bench.py times
    it separately as `secondary.global_adjacency_gather_ms`).

    In steady state step() only ENQUEUES work (kernels and three collectives on one stream): the
    adjacency travels in fixed-capacity exchange blocks (ta_adjacency_pack_shared / _merge_blocks) whose
    capacity was agreed once, synchronously, on the first step.  Overflow and range flags ride in
    the block headers, so all ranks reach the same verdict; it is read when results are fetched
    (result_*), which re-sizes and redoes the step if a block or table was too small.
    """

    def __init__(self, ctx, vol_tensor, itemsize, a_origin, has_low_halo, max_label, features,
                 group=None, device=0, exchange_capacity=None, stream=None, reduce="all", ids=None):
        """ids: SPARSE label ids -- the ascending table of ALL ids of the partitioned volume (`union_of_ids`, the same on every
        rank): the slab is swept in the ranks of that table, every per-label row of this job (sums, boxes, the exchange) is a
        rank, `max_label` is ignored (rows = len(ids)) and result_arrays() answers pairs in ids and carries the table.
        COMPACTION IS A SNAPSHOT: the context sweeps a rank copy of the slab written once, here.  After rewriting the adopted
        buffer in place (the next frame of a series, a refreshed halo plane) call refresh() before the next step().
        reduce: "all" -- both tables all-reduced, every rank holds the global rows after step(); "scatter" -- the sums
        (8 of the 10.4 MB at 100k labels) are reduce-SCATTERED: a rank keeps the global rows of its share of the labels in
        `sums_shard` (and its own slab's partial rows in `sums`), only the boxes, which decide which pairs travel, are
        all-reduced; result_counts() / result_arrays() gather the shards (collective)."""
        import torch
        self.ctx, self.vol, self.group = ctx, vol_tensor, group
        if reduce not in ("all", "scatter"):
            raise ValueError("reduce must be 'all' or 'scatter'")
        self.reduce = reduce if group is not None else "all"
        # Kernels (C ABI) and collectives (torch.distributed) must be ordered on ONE stream: the torch stream given
        # here, else torch's current stream at construction -- never a private stream of the context, which nothing
        # orders against the stream RCCL enqueues on.
        if stream is None and vol_tensor.is_cuda:
            stream = torch.cuda.current_stream(vol_tensor.device)
        self.stream = stream                    # torch.cuda.Stream all of this job's work is ordered on
        if stream is not None:
            ctx.set_stream(stream.cuda_stream)
        self.has_low_halo = bool(has_low_halo)
        self.a_origin = int(a_origin)           # global index of the first OWNED plane
        self.ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
        if self.ids is not None:
            max_label = max(int(self.ids.size) - 1, 0)
        self.max_label, self.features = int(max_label), features
        dev = "cuda:%d" % device
        rows = self.max_label + 1
        self.sums_shard = None
        if self.reduce == "scatter":
            import torch.distributed as dist
            world = dist.get_world_size(group)
            S = sums_shard_rows(rows, world)
            rows = world * S                            # (the rows past max_label stay zero: the sweep never writes them)
            self.sums_shard = torch.zeros((S, 10), dtype=torch.int64, device=dev)
        self.sums = torch.zeros((rows, 10), dtype=torch.int64, device=dev)
        self.boxes = torch.zeros((self.max_label + 1, 6), dtype=torch.int32, device=dev)
        ctx.set_volume_device(vol_tensor.data_ptr(), itemsize, vol_tensor.shape, a0_origin=a_origin,
                              has_low_halo=has_low_halo, keep=vol_tensor)
        if self.ids is not None:
            ctx.compact_labels(self.ids)
        ctx.bind_accumulators(self.sums.data_ptr(), self.boxes.data_ptr(), self.max_label,
                              keep=(self.sums, self.boxes))
        self._torch = torch
        self._cap = int(exchange_capacity) if exchange_capacity else None
        self._send = self._recv = None
        self._unverified = False
        self.redo_count = 0                     # steps repeated because a block / table was too small

    def owned_view(self):
        return self.vol[1:] if self.has_low_halo else self.vol

    def _pair_rows(self, allow_partial=False):
        """The context's pair list with its labels as ROWS of this job's tables (a compacted context answers in ids)."""
        lo, hi, faces = self.ctx.adjacency(allow_partial=allow_partial)
        if self.ids is not None:
            lo, hi = np.searchsorted(self.ids, lo).astype(np.uint32), np.searchsorted(self.ids, hi).astype(np.uint32)
        return lo, hi, faces

    # -- adjacency exchange ------------------------------------------------------------------
    def _adjacency_wanted(self):
        from . import _capi
        return self.group is not None and bool(_capi.feature_mask(self.features) & _capi.F_ADJACENCY)

    # verdict words shared by all ranks (MAX over ranks decides)
    _OK, _CAPACITY, _RANGE, _FAILED = 0, 1, 2, 3

    def _local_verdict(self):
        """(pair count, status) of this rank's last extraction; never raises: the status is agreed on collectively."""
        from . import _capi
        try:
            return self.ctx.adjacency_size(), self._OK, None     # drains this job's stream, validates the flags
        except _capi.TissueScanError as e:
            code = {_capi.TA_ECAPACITY: self._CAPACITY, _capi.TA_ERANGE: self._RANGE}.get(e.code, self._FAILED)
            return 0, code, e

    def _raise_collectively(self, status, own_error):
        """Every rank leaves with an exception of the same kind."""
        from . import _capi
        if status == self._RANGE:
            raise _capi.TissueScanError(_capi.TA_ERANGE, "a rank's slab holds a label above max_label=%d" % self.max_label)
        if own_error is not None:
            raise own_error
        raise RuntimeError("the extraction failed on another rank of the group")

    def _agree_on_sizes(self):
        """One-off and synchronous: block capacity from the largest local pair list, one table
        size for all ranks.  Returns True when the extraction had to be repeated.  A failure on any
        rank (label above max_label, HIP error ...) is raised on EVERY rank after the collective."""
        import torch.distributed as dist
        from . import _capi
        torch = self._torch
        n, status, err = self._local_verdict()         # a local table overflow re-runs inside
        slots = self.ctx.get_option(_capi.OPT_PAIR_SLOTS)
        t = torch.tensor([n, slots, status], dtype=torch.int64, device=self.sums.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        nmax, smax, status = [int(x) for x in t.tolist()]
        if status not in (self._OK, self._CAPACITY):
            self._unverified = False
            self._raise_collectively(status, err)
        again = smax != slots or status == self._CAPACITY
        if again:
            self.ctx.set_option(_capi.OPT_PAIR_SLOTS, max(smax, self.ctx.get_option(_capi.OPT_PAIR_SLOTS)))
            self.ctx.extract(self.features, self.max_label)
        return again

    def _slab_planes(self):
        lo = self.a_origin
        return lo, lo + int(self.vol.shape[0]) - (1 if self.has_low_halo else 0)

    def _agree_on_capacity(self):
        """One-off and synchronous, after the boxes were reduced: the block capacity from the largest number of pairs
        any rank has to send (a quarter more, in 1024s).  A rank whose pair list cannot be read (table overflow after
        the accumulators were reduced, HIP error ...) does not raise alone: its status travels with the count and every
        rank leaves the collective with the same verdict.  Returns False when the step has to be redone."""
        import torch.distributed as dist
        from . import _capi
        torch = self._torch
        lo, hi = self._slab_planes()
        status, err, ntravel = self._OK, None, 0
        try:
            plo, phi, _ = self._pair_rows()
            excl = slab_exclusive(self.boxes.cpu().numpy(), lo, hi)
            inside = (plo <= self.max_label) & (phi <= self.max_label)
            travels = ~inside | ~(excl[np.minimum(plo, self.max_label)] | excl[np.minimum(phi, self.max_label)])
            ntravel = int(travels.sum())
        except _capi.TissueScanError as e:
            status = {_capi.TA_ECAPACITY: self._CAPACITY, _capi.TA_ERANGE: self._RANGE}.get(e.code, self._FAILED)
            err = e
        t = torch.tensor([ntravel, status], dtype=torch.int64, device=self.sums.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        nmax, status = [int(x) for x in t.tolist()]
        if status == self._CAPACITY:
            return False                            # (the failing rank's table has already been grown by the getter)
        if status != self._OK:
            self._unverified = False
            self._raise_collectively(status, err)
        self._cap = max(1024, -(-(nmax + nmax // 4 + 1) // 1024) * 1024)
        return True

    def _exchange_buffers(self):
        import torch.distributed as dist
        from . import _capi
        torch = self._torch
        words = _capi.exchange_words(self._cap)
        world = dist.get_world_size(self.group)
        if self._send is None or self._send.shape[0] != words:
            self._send = torch.empty((words,), dtype=torch.int64, device=self.sums.device)
            self._recv = torch.empty((world * words,), dtype=torch.int64, device=self.sums.device)
        return world

    def _on_stream(self):
        import contextlib
        return self._torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def refresh(self):
        """The voxels of the adopted buffer were rewritten in place: with `ids` the sweep reads a rank copy, which is written
        again here (asynchronous, on the job's stream; an id outside `ids` makes the next finish() / getter raise TA_ERANGE).
        Without `ids` the sweep reads the buffer itself and there is nothing to do."""
        if self.ids is not None:
            with self._on_stream():
                self.ctx.rerank()

    def step(self):
        with self._on_stream():
            self._step()

    def finish(self):
        """COLLECTIVE (every rank of the group must call it, like step()): read the verdict of the
        last step; re-size and redo it when an exchange block or an adjacency table was too small
        anywhere.  The result getters call it, so they are collective too until it has run."""
        with self._on_stream():
            self._finish()

    def _step(self):
        import torch.distributed as dist
        torch = self._torch
        self.ctx.extract(self.features, self.max_label)
        if self.group is None:
            self._unverified = True          # (finish() reads the range / overflow flags of the last extraction)
            return
        adj = self._adjacency_wanted()
        first = adj and (self._cap is None or self._send is None)
        if first:
            self._agree_on_sizes()
        if self.reduce == "scatter":
            reduce_scatter_sums(self.sums, self.sums_shard, self.group)
            dist.all_reduce(self.boxes, op=dist.ReduceOp.MIN, group=self.group)
        else:
            allreduce_accumulators(self.sums, self.boxes, self.group)
        self.ctx.accumulators_reduced()          # from here on a local re-run would lose the other ranks' rows
        if adj:
            if self._cap is None and not self._agree_on_capacity():
                # a pair table overflowed on some rank after the rows were reduced: every rank redoes the step with the
                # largest table (the sweep starts from clean accumulators: init is part of ta_extract)
                self.redo_count += 1
                if self.redo_count > 8:
                    raise RuntimeError("adjacency table still overflows after %d attempts" % self.redo_count)
                return self._step()
            world = self._exchange_buffers()
            self.ctx.adjacency_pack_shared(self._send.data_ptr(), self._cap)
            dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
            # With nccl (RCCL) the current stream -- which the context launches on -- already waits for
            # the collective; gloo moves device tensors through the host on its own streams.
            if dist.get_backend(self.group) != "nccl":
                torch.cuda.synchronize()
            self.ctx.adjacency_merge_blocks(self._recv.data_ptr(), world, self._cap)
        self._unverified = True

    def _finish(self):
        if not self._unverified:
            return
        if self.group is None:
            # one rank: drain the stream and validate the flags of the last extraction (a label above max_label raises, a
            # pair table that overflowed is grown and the extraction repeated inside the library)
            self._unverified = False
            self.ctx.adjacency_size()
            return
        import torch.distributed as dist
        from . import _capi
        torch = self._torch
        for attempt in range(6):
            _, status, err = self._local_verdict()
            t = torch.tensor([status], dtype=torch.int64, device=self.sums.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            status = int(t.item())
            if status == self._OK:
                self._unverified = False
                return
            if status != self._CAPACITY:                   # same kind of exception on every rank
                self._unverified = False
                self._raise_collectively(status, err)
            self.redo_count += 1
            self._cap = None
            self._send = self._recv = None
            if attempt > 0:                                # re-measured blocks were not enough: grow the tables
                self.ctx.set_option(_capi.OPT_PAIR_SLOTS, self.ctx.get_option(_capi.OPT_PAIR_SLOTS) + 2)
            self._step()
        raise RuntimeError("adjacency exchange still overflows after %d attempts" % 6)

    def global_sums(self):
        """The global int64[max_label + 1, 10] rows on this rank (COLLECTIVE after a reduce-scatter: gathers the shards)."""
        if self.reduce == "scatter":
            with self._on_stream():
                return allgather_sums(self.sums_shard, self.max_label + 1, self.group)
        return self.sums[:self.max_label + 1]

    def result_counts(self):
        self.finish()
        sums = self.global_sums()
        self._torch.cuda.synchronize()
        return sums[:, 0].cpu().numpy()

    def _global_pairs(self, plo, phi, faces):
        """COLLECTIVE: this context holds its private pairs + the merged travelling pairs (the same on every rank);
        gather the other ranks' private lists and return the global list, sorted by (lo, hi)."""
        import torch.distributed as dist
        torch = self._torch
        a0, a1 = self._slab_planes()
        excl = slab_exclusive(self.boxes.cpu().numpy(), a0, a1)
        private = excl[plo] | excl[phi]
        on = self.sums.device if dist.get_backend(self.group) == "nccl" else "cpu"
        keys = (plo.astype(np.int64) << 32) | phi.astype(np.int64)
        kall, fall, _ = allgather_pairs(torch.from_numpy(keys[private]).to(on),
                                        torch.from_numpy(faces[private].astype(np.int64)).to(on), self.group)
        kall, fall = kall.cpu().numpy(), fall.cpu().numpy()
        keep = kall != EMPTY_KEY
        keys = np.concatenate([kall[keep], keys[~private]])
        faces = np.concatenate([fall[keep].astype(np.uint64), faces[~private]])
        order = np.argsort(keys, kind="stable")
        keys = keys[order]
        return (keys >> 32).astype(np.uint32), (keys & 0xFFFFFFFF).astype(np.uint32), faces[order]

    def result_arrays(self):
        """Global result in the host-getter layout (memory-axis order)."""
        self.finish()
        sums = self.global_sums()
        self._torch.cuda.synchronize()
        from . import _capi
        out = from_device_layout(sums.cpu().numpy(), self.boxes.cpu().numpy(),
                                 second_moments=bool(_capi.feature_mask(self.features) & _capi.F_MOMENT2))
        out["max_label"] = self.max_label
        if _capi.feature_mask(self.features) & _capi.F_ADJACENCY:
            lo, hi, faces = self._pair_rows(allow_partial=True)
            if self._adjacency_wanted():
                lo, hi, faces = self._global_pairs(lo, hi, faces)
            if self.ids is not None:
                lo, hi = self.ids[lo], self.ids[hi]
        else:
            lo = hi = np.zeros(0, dtype=np.uint32)
            faces = np.zeros((0, 3), dtype=np.uint64)
        out.update(pair_lo=lo, pair_hi=hi, pair_faces=faces)
        if self.ids is not None:
            out["ids"] = self.ids.astype(np.int64)
        return out


class PipelinedSlabJob(object):
    """`depth` SlabJobs over the SAME resident slab, each with its own C-ABI context, accumulators,
    exchange buffers and HIP stream, used round-robin: step i's reduce / adjacency exchange (ordered
    on stream i % depth) overlaps step i+1's sweep (on the next stream), the way a time series of
    volumes would be processed.  Every step still does all of its work; finish() (collective) waits
    for and validates every step in flight, in issue order."""

    def __init__(self, vol_tensor, itemsize, a_origin, has_low_halo, max_label, features, group=None,
                 device=0, depth=2, tile_planes=0, reduce="all", ids=None):
        import torch
        from . import _capi
        torch.cuda.synchronize(device)            # the slab was written on another stream
        self.jobs = []
        for _ in range(max(1, int(depth))):
            stream = torch.cuda.Stream(device=device)
            ctx = _capi.Context(device)
            if tile_planes:
                ctx.set_option(_capi.OPT_TILE_PLANES, tile_planes)
            with torch.cuda.stream(stream):
                self.jobs.append(SlabJob(ctx, vol_tensor, itemsize, a_origin, has_low_halo, max_label, features,
                                         group=group, device=device, stream=stream, reduce=reduce, ids=ids))
        self._issued = 0

    @property
    def last(self):
        """The SlabJob that ran the most recent step (its ctx holds that step's adjacency)."""
        return self.jobs[(self._issued - 1) % len(self.jobs)]

    def owned_view(self):
        return self.jobs[0].owned_view()

    def refresh(self):
        """The slab's voxels were rewritten in place: every job in flight re-reads them (see SlabJob.refresh)."""
        for j in self.jobs:
            j.refresh()

    def step(self):
        self.jobs[self._issued % len(self.jobs)].step()
        self._issued += 1

    def finish(self):
        n = len(self.jobs)
        for k in range(max(0, self._issued - n), self._issued):     # oldest step in flight first
            self.jobs[k % n].finish()

    def result_counts(self):
        self.finish()
        return self.last.result_counts()

    def result_arrays(self):
        self.finish()
        return self.last.result_arrays()

    def close(self):
        for j in self.jobs:
            j.ctx.close()
