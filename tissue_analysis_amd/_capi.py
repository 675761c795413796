"""ctypes binding of include/tissue_scan.h (libtissue_scan.so).

This is the only door from Python into the HIP kernels.  There is no CPU fallback: if the
shared library is missing, or no gfx950 GPU is visible, the product path raises.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TISSUE_SCAN_LIB") or os.path.join(_HERE, "libtissue_scan.so")

TA_OK, TA_EINVAL, TA_EHIP, TA_ENOMEM, TA_ERANGE, TA_ECAPACITY, TA_ENODEVICE = 0, -1, -2, -3, -4, -5, -6
F_VOLUME, F_BBOX, F_MOMENT1, F_MOMENT2, F_ADJACENCY = 1, 2, 4, 8, 16
F_ALL = 31
ADJ_LOCAL, ADJ_MERGED, ADJ_PARTIAL = 0, 1, 2
ABI_VERSION = 5          # TA_ABI_VERSION of include/tissue_scan.h this binding was written against
FEATURES = dict(VOLUME=F_VOLUME, BBOX=F_BBOX, MOMENT1=F_MOMENT1, MOMENT2=F_MOMENT2,
                ADJACENCY=F_ADJACENCY)
OPT_IMPL, OPT_TILE_PLANES, OPT_PAIR_SLOTS, OPT_TIMING, OPT_TIMING_RING, OPT_VOLUME_SLACK, OPT_SWEEP_SHAPE, OPT_SWEEP_SHAPE_USED = 1, 2, 3, 4, 5, 6, 7, 8
STREAM_LEGACY_DEFAULT = 1          # TA_STREAM_LEGACY_DEFAULT of include/tissue_scan.h

# every symbol include/tissue_scan.h declares
SYMBOLS = (
    "ta_version", "ta_adjacency_scope", "ta_last_error", "ta_device_count", "ta_ctx_create", "ta_ctx_destroy",
    "ta_ctx_set_stream", "ta_ctx_set_option", "ta_ctx_get_option", "ta_ctx_synchronize", "ta_volume_set",
    "ta_volume_set_device", "ta_volume_max_label", "ta_volume_label_census", "ta_label_census_get", "ta_volume_compact_labels",
    "ta_volume_is_compact", "ta_volume_rerank", "ta_volume_uncompact", "ta_volume_owned_planes", "ta_volume_plane_events", "ta_volume_relabel", "ta_volume_get", "ta_volume_map",
    "ta_volume_first_layer", "ta_volume_hollow", "ta_volume_layer18", "ta_wall_voxels_count", "ta_wall_voxels_get", "ta_wall_voxels_get_by_pair",
    "ta_wall_medians", "ta_wall_medians_get",
    "ta_extract", "ta_get_labels",
    "ta_adjacency_size", "ta_adjacency_get", "ta_timing", "ta_timing_series", "ta_read_probe", "ta_debug_counters", "ta_bind_accumulators",
    "ta_accumulators_device", "ta_accumulators_reduced", "ta_adjacency_device", "ta_adjacency_export", "ta_adjacency_merge",
    "ta_adjacency_pack", "ta_adjacency_pack_shared", "ta_adjacency_merge_blocks", "ta_synth_voronoi",
    "ta_device_malloc", "ta_device_free", "ta_memcpy_d2h", "ta_memcpy_h2d",
)


def exchange_words(capacity_pairs):
    """uint64 words of one exchange block (TA_EXCHANGE_WORDS in include/tissue_scan.h)."""
    return 2 + 4 * int(capacity_pairs)


class TissueScanError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libtissue_scan error %d: %s" % (code, message))
        self.code = code


_lib = None


def load():
    """Load libtissue_scan.so (built in-tree by `python -m tissue_analysis_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: the HIP extension has not been built "
            "(run `python -m tissue_analysis_amd.build`); there is no CPU fallback" % LIB_PATH)
    # PyTorch-ROCm ships its own copy of the HIP / HSA runtime; libtissue_scan.so is linked against /opt/rocm's.  Two
    # runtimes in one process cannot both open the GPU ("No HIP GPUs are available" from whichever comes second) unless
    # torch's is the one the loader sees first -- then this library binds to it too.  So when torch is installed, it is
    # imported before the library is opened, whatever order the caller's imports have.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, u32, u64, ci = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64,
                             ctypes.c_int)
    P = ctypes.POINTER
    sig = {
        "ta_version": (ci, []),
        "ta_adjacency_scope": (ci, [vp, P(ci)]),
        "ta_last_error": (ctypes.c_char_p, []),
        "ta_device_count": (ci, [P(ci)]),
        "ta_ctx_create": (ci, [ci, P(vp)]),
        "ta_ctx_destroy": (ci, [vp]),
        "ta_ctx_set_stream": (ci, [vp, vp]),
        "ta_ctx_set_option": (ci, [vp, ci, i64]),
        "ta_ctx_get_option": (ci, [vp, ci, P(i64)]),
        "ta_ctx_synchronize": (ci, [vp]),
        "ta_volume_set": (ci, [vp, vp, ci, P(i64), P(i64)]),
        "ta_volume_set_device": (ci, [vp, vp, ci, P(i64), i64, ci]),
        "ta_volume_max_label": (ci, [vp, P(u32)]),
        "ta_volume_label_census": (ci, [vp, P(u32), P(u32)]),
        "ta_label_census_get": (ci, [vp, vp]),
        "ta_volume_compact_labels": (ci, [vp, vp, u32, P(u32)]),
        "ta_volume_is_compact": (ci, [vp, P(ci), P(u32)]),
        "ta_volume_rerank": (ci, [vp]),
        "ta_volume_uncompact": (ci, [vp]),
        "ta_volume_owned_planes": (ci, [vp, P(i64)]),
        "ta_volume_plane_events": (ci, [vp, vp]),
        "ta_volume_relabel": (ci, [vp, vp, u32]),
        "ta_volume_get": (ci, [vp, vp]),
        "ta_volume_map": (ci, [vp, vp, u32, vp, ci, vp]),
        "ta_volume_first_layer": (ci, [vp, u32, ci, vp]),
        "ta_volume_hollow": (ci, [vp, u32, ci, ci, vp]),
        "ta_volume_layer18": (ci, [vp, vp]),
        "ta_wall_voxels_count": (ci, [vp, P(i64)]),
        "ta_wall_voxels_get": (ci, [vp, vp, vp, P(ctypes.c_double)]),
        "ta_wall_medians": (ci, [vp, ci, P(i64), P(ctypes.c_double)]),
        "ta_wall_medians_get": (ci, [vp, vp, vp, vp]),
        "ta_wall_voxels_get_by_pair": (ci, [vp, vp, vp, P(ctypes.c_double)]),
        "ta_extract": (ci, [vp, u32, u32]),
        "ta_get_labels": (ci, [vp, vp, vp, vp, vp]),
        "ta_adjacency_size": (ci, [vp, P(i64)]),
        "ta_adjacency_get": (ci, [vp, vp, vp, vp]),
        "ta_timing": (ci, [vp, P(ctypes.c_double), P(ctypes.c_double), P(ctypes.c_double), P(u64)]),
        "ta_timing_series": (ci, [vp, P(ctypes.c_double), ci, P(ci)]),
        "ta_read_probe": (ci, [vp, vp, u64, ci, P(ctypes.c_double)]),
        "ta_debug_counters": (ci, [vp, P(u32)]),
        "ta_bind_accumulators": (ci, [vp, vp, vp, u32]),
        "ta_accumulators_device": (ci, [vp, P(vp), P(vp), P(u32)]),
        "ta_accumulators_reduced": (ci, [vp]),
        "ta_adjacency_device": (ci, [vp, P(vp), P(vp), P(i64)]),
        "ta_adjacency_export": (ci, [vp, vp, vp, i64]),
        "ta_adjacency_merge": (ci, [vp, vp, vp, i64]),
        "ta_adjacency_pack": (ci, [vp, vp, i64]),
        "ta_adjacency_pack_shared": (ci, [vp, vp, i64]),
        "ta_adjacency_merge_blocks": (ci, [vp, vp, ci, i64]),
        "ta_synth_voronoi": (ci, [vp, vp, ci, P(i64), i64, i64, vp, P(ctypes.c_int32), vp]),
        "ta_device_malloc": (ci, [vp, u64, P(vp)]),
        "ta_device_free": (ci, [vp, vp]),
        "ta_memcpy_d2h": (ci, [vp, vp, vp, u64]),
        "ta_memcpy_h2d": (ci, [vp, vp, vp, u64]),
    }
    for name in SYMBOLS:
        fn = getattr(lib, name)      # AttributeError here == the .so does not match the header
        fn.restype, fn.argtypes = sig[name]
    if lib.ta_version() != ABI_VERSION:
        raise ImportError("%s speaks ABI version %d, this binding %d: rebuild it (python -m tissue_analysis_amd.build --force)"
                          % (LIB_PATH, lib.ta_version(), ABI_VERSION))
    _lib = lib
    return lib


def _check(rc):
    if rc != TA_OK:
        raise TissueScanError(rc, load().ta_last_error().decode(errors="replace"))


def device_count():
    n = ctypes.c_int(0)
    _check(load().ta_device_count(ctypes.byref(n)))
    return n.value


def feature_mask(names):
    if isinstance(names, int):
        return names
    m = 0
    for n in names:
        m |= FEATURES[n.upper()]
    return m


def _i64x3(v):
    return (ctypes.c_int64 * 3)(*[int(x) for x in v])


def _dense_permuted(a):
    """True when the array is dense in some axis permutation (C, F or transposed layouts)."""
    order = sorted(range(a.ndim), key=lambda d: (a.shape[d] != 1, -a.strides[d]))
    expect = a.dtype.itemsize
    for d in reversed(order):
        if a.shape[d] != 1 and a.strides[d] != expect:
            return False
        expect *= a.shape[d]
    return True


class Context(object):
    """One ta_ctx == one GPU.  Thin, explicit wrapper: every method is one C call."""

    def __init__(self, device=0, stream=None):
        self._lib = load()
        self._h = ctypes.c_void_p()
        _check(self._lib.ta_ctx_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)
        self._keep = []      # objects whose device memory the context currently points at
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ta_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()
        self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- plumbing
    def set_stream(self, stream_handle):
        """stream_handle: a hipStream_t as an integer; 0 = the device's legacy default (null) stream -- what
        torch.cuda.default_stream().cuda_stream is; None = a private non-blocking stream owned by the context."""
        if stream_handle is None:
            h = 0
        elif int(stream_handle) == 0:
            h = STREAM_LEGACY_DEFAULT
        else:
            h = int(stream_handle)
        _check(self._lib.ta_ctx_set_stream(self._h, ctypes.c_void_p(h)))

    def set_option(self, key, value):
        _check(self._lib.ta_ctx_set_option(self._h, int(key), int(value)))

    def get_option(self, key):
        v = ctypes.c_int64(0)
        _check(self._lib.ta_ctx_get_option(self._h, int(key), ctypes.byref(v)))
        return int(v.value)

    def synchronize(self):
        _check(self._lib.ta_ctx_synchronize(self._h))

    # -- volume
    def set_volume(self, array):
        """Upload a uint16/uint32 numpy volume in whatever dense layout it has."""
        a = np.asarray(array)
        if a.ndim != 3:
            raise ValueError("a 3D volume is required")
        if a.dtype not in (np.uint16, np.uint32):
            raise TypeError("label volumes must be uint16 or uint32, not %s" % a.dtype)
        if not _dense_permuted(a):
            a = np.ascontiguousarray(a)
        _check(self._lib.ta_volume_set(self._h, ctypes.c_void_p(a.ctypes.data), a.dtype.itemsize,
                                       _i64x3(a.shape), _i64x3(a.strides)))
        self._owned_planes = int(a.shape[int(np.argmax(a.strides))])         # planes of the slowest MEMORY axis

    def relabel(self, lut):
        """In place on the resident volume: v -> lut[v] for v < len(lut) (host uint32 table)."""
        lut = np.ascontiguousarray(lut, dtype=np.uint32)
        _check(self._lib.ta_volume_relabel(self._h, ctypes.c_void_p(lut.ctypes.data), int(lut.size)))

    def get_volume(self, out):
        """Copy the resident volume into `out`, which must have the dtype, shape and dense layout
        of the array given to set_volume()."""
        if not (isinstance(out, np.ndarray) and _dense_permuted(out) and out.flags.writeable):
            raise ValueError("get_volume needs a writeable dense (possibly axis-permuted) ndarray")
        _check(self._lib.ta_volume_get(self._h, ctypes.c_void_p(out.ctypes.data)))
        return out

    def map_labels(self, lut, fill, like):
        """out[p] = lut[V[p]] (fill beyond the table): `lut` is a 1-D array of a 1/2/4/8-byte dtype;
        the result has that dtype and the shape and layout of `like` (the array given to set_volume)."""
        lut = np.ascontiguousarray(lut)
        if lut.dtype.itemsize not in (1, 2, 4, 8):
            raise TypeError("lookup tables of %s are not supported" % lut.dtype)
        out = np.empty_like(like, dtype=lut.dtype)            # keeps the memory order of `like`
        if not _dense_permuted(out):
            raise ValueError("map_labels needs a dense (possibly axis-permuted) volume")
        fillv = np.array([fill]).astype(lut.dtype)
        _check(self._lib.ta_volume_map(self._h, ctypes.c_void_p(lut.ctypes.data), int(lut.size),
                                       ctypes.c_void_p(fillv.ctypes.data), int(lut.dtype.itemsize),
                                       ctypes.c_void_p(out.ctypes.data)))
        return out

    def first_layer(self, background, keep_background, like):
        """voxel_first_layer of the resident volume (SIA:1024-1046) as a host image shaped and laid out like `like`
        (the array given to set_volume)."""
        out = np.empty_like(like)
        if not _dense_permuted(out):
            raise ValueError("first_layer needs a dense (possibly axis-permuted) volume")
        _check(self._lib.ta_volume_first_layer(self._h, int(background), int(bool(keep_background)),
                                               ctypes.c_void_p(out.ctypes.data)))
        return out

    def hollow(self, background, remove_background, like, label_bits=0):
        """hollow_out_cells of the resident volume (SIA:74-95): the labels where their integer Laplacian (modulo
        2^label_bits, 0 = the volume's own width) is not zero (and, with remove_background, where they are not the
        background), 0 elsewhere; shaped and laid out like `like`."""
        out = np.empty_like(like)
        if not _dense_permuted(out):
            raise ValueError("hollow needs a dense (possibly axis-permuted) volume")
        _check(self._lib.ta_volume_hollow(self._h, int(background) & 0xFFFFFFFF, int(bool(remove_background)), int(label_bits),
                                          ctypes.c_void_p(out.ctypes.data)))
        return out

    def layer18(self, like):
        """uint8 image, 1 where a voxel has one of its 18 neighbours in another label (cells_voxel_layer, SIA:1399-1448,
        for every label at once); shaped and laid out like `like`."""
        out = np.empty_like(like, dtype=np.uint8)
        if not _dense_permuted(out):
            raise ValueError("layer18 needs a dense (possibly axis-permuted) volume")
        _check(self._lib.ta_volume_layer18(self._h, ctypes.c_void_p(out.ctypes.data)))
        return out

    def wall_voxels(self, by_pair=False):
        """All (pair, wall voxel) records of the resident volume: lo u32[n], hi u32[n], coords i32[n,3]
        (array-axis order), ordered by the voxel's position in memory -- or, with by_pair, grouped by (lo, hi) on the
        device with each pair's voxels in memory order; plus the kernels' milliseconds."""
        n = ctypes.c_int64(0)
        _check(self._lib.ta_wall_voxels_count(self._h, ctypes.byref(n)))
        pairs = np.empty((n.value, 2), dtype=np.uint32)
        coords = np.empty((n.value, 3), dtype=np.int32)
        ms = ctypes.c_double(0.0)
        fetch = self._lib.ta_wall_voxels_get_by_pair if by_pair else self._lib.ta_wall_voxels_get
        _check(fetch(self._h, pairs.ctypes.data, coords.ctypes.data, ctypes.byref(ms)))
        return pairs[:, 0], pairs[:, 1], coords, ms.value

    def wall_medians(self, max_iter=200):
        """The median voxel of every wall of the resident volume, computed on the device (C-ordered volumes): keys
        uint64[E] = lo << 32 | hi ascending, sizes uint32[E] (wall voxels), medians int32[E, 3] (array-axis order), and
        the kernels' milliseconds (grouping by pair included) and `moving` bool[E]: walls whose iteration had not settled after
        `max_iter` Weiszfeld passes (the reference raises for such a wall: the caller does, for the walls it asks for)."""
        n = ctypes.c_int64(0)
        _check(self._lib.ta_wall_voxels_count(self._h, ctypes.byref(n)))
        count, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
        _check(self._lib.ta_wall_medians(self._h, int(max_iter), ctypes.byref(count), ctypes.byref(ms)))
        E = count.value
        pairs = np.empty((E, 2), dtype=np.uint32)
        sizes = np.empty(E, dtype=np.uint32)
        med = np.empty((E, 3), dtype=np.int32)
        _check(self._lib.ta_wall_medians_get(self._h, pairs.ctypes.data, sizes.ctypes.data, med.ctypes.data))
        keys = (pairs[:, 0].astype(np.uint64) << np.uint64(32)) | pairs[:, 1].astype(np.uint64)
        moving = (sizes & np.uint32(0x80000000)) != 0
        return keys, sizes & np.uint32(0x7FFFFFFF), med, ms.value, moving

    def set_volume_device(self, dev_ptr, itemsize, buf_dims, a0_origin=0, has_low_halo=False, keep=None):
        _check(self._lib.ta_volume_set_device(self._h, ctypes.c_void_p(int(dev_ptr)), int(itemsize),
                                              _i64x3(buf_dims), int(a0_origin), int(bool(has_low_halo))))
        self._keep = [keep]
        self._owned_planes = int(buf_dims[0]) - (1 if has_low_halo else 0)
        # a torch tensor that is a view of a larger storage: tell the library how many bytes are readable behind it
        try:
            st = keep.untyped_storage()
            end = (keep.storage_offset() + keep.numel()) * keep.element_size()
            if keep.is_contiguous() and int(keep.data_ptr()) == int(dev_ptr) and st.nbytes() > end:
                self.set_option(OPT_VOLUME_SLACK, int(st.nbytes() - end))
        except AttributeError:
            pass

    def max_label(self):
        v = ctypes.c_uint32(0)
        _check(self._lib.ta_volume_max_label(self._h, ctypes.byref(v)))
        return v.value

    # -- sparse label ids
    def label_census(self):
        """(max id, ascending uint32 ids present in the resident volume): np.unique on the device."""
        top, n = ctypes.c_uint32(0), ctypes.c_uint32(0)
        _check(self._lib.ta_volume_label_census(self._h, ctypes.byref(top), ctypes.byref(n)))
        ids = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            _check(self._lib.ta_label_census_get(self._h, ids.ctypes.data_as(ctypes.c_void_p)))
        return top.value, ids

    def compact_labels(self, ids=None):
        """From now on the sweep reads a copy of the volume written in the RANKS of its ids (`ids`: an ascending unique list
        that covers the volume, e.g. the union over the slabs of a partitioned volume; None = this volume's own census).
        Returns the rank -> id table (uint32): rows of `labels()` are ranks, `adjacency()` answers in ids."""
        n = ctypes.c_uint32(0)
        if ids is None:
            _check(self._lib.ta_volume_compact_labels(self._h, None, 0, ctypes.byref(n)))
        else:
            ids = np.ascontiguousarray(ids, dtype=np.uint32)
            _check(self._lib.ta_volume_compact_labels(self._h, ids.ctypes.data_as(ctypes.c_void_p), ids.size, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            _check(self._lib.ta_label_census_get(self._h, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def is_compact(self):
        flag, n = ctypes.c_int(0), ctypes.c_uint32(0)
        _check(self._lib.ta_volume_is_compact(self._h, ctypes.byref(flag), ctypes.byref(n)))
        return bool(flag.value)

    def compact_ids(self):
        """The rank -> id table of a compacted context (uint32, ascending)."""
        flag, n = ctypes.c_int(0), ctypes.c_uint32(0)
        _check(self._lib.ta_volume_is_compact(self._h, ctypes.byref(flag), ctypes.byref(n)))
        if not flag.value:
            raise TissueScanError(TA_EINVAL, "the context is not compacted")
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            _check(self._lib.ta_label_census_get(self._h, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    # -- hot path
    def rerank(self):
        """Compaction is a SNAPSHOT of the voxels: after rewriting an adopted device buffer in place, refresh the rank copy the
        sweep reads (asynchronous; an id outside the census makes the next extraction's getters raise TA_ERANGE)."""
        _check(self._lib.ta_volume_rerank(self._h))

    def uncompact(self):
        """Back to dense rows 0 .. max_label (releases the rank copy)."""
        _check(self._lib.ta_volume_uncompact(self._h))

    def owned_planes(self):
        n = ctypes.c_int64(0)
        _check(self._lib.ta_volume_owned_planes(self._h, ctypes.byref(n)))
        return int(n.value)

    def plane_events(self):
        """uint64[owned planes]: label changes along the fast axis in every owned plane of the resident volume (the weight
        distributed.balanced_cuts balances)."""
        out = np.zeros(self.owned_planes(), dtype=np.uint64)      # (asked of the library: a size-1 axis moves the slowest MEMORY axis)
        _check(self._lib.ta_volume_plane_events(self._h, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def extract(self, features, max_label):
        _check(self._lib.ta_extract(self._h, feature_mask(features), int(max_label)))
        self._max_label = int(max_label)

    def labels(self):
        """(count u64[L+1], bbox i32[L+1,6], sum1 u64[L+1,3], sum2 u64[L+1,6])"""
        n = self._max_label + 1
        count = np.zeros(n, dtype=np.uint64)
        bbox = np.zeros((n, 6), dtype=np.int32)
        sum1 = np.zeros((n, 3), dtype=np.uint64)
        sum2 = np.zeros((n, 6), dtype=np.uint64)
        _check(self._lib.ta_get_labels(self._h, count.ctypes.data, bbox.ctypes.data, sum1.ctypes.data,
                                       sum2.ctypes.data))
        return count, bbox, sum1, sum2

    def adjacency_size(self):
        """Pair count of the last extraction (drains the stream, validates its flags)."""
        n = ctypes.c_int64(0)
        _check(self._lib.ta_adjacency_size(self._h, ctypes.byref(n)))
        return int(n.value)

    def adjacency_scope(self):
        """ADJ_LOCAL / ADJ_MERGED / ADJ_PARTIAL: which pairs the adjacency getters answer with right now."""
        scope = ctypes.c_int(0)
        _check(self._lib.ta_adjacency_scope(self._h, ctypes.byref(scope)))
        return scope.value

    def adjacency(self, allow_partial=False):
        """(lo u32[n], hi u32[n], faces u64[n,3]) sorted by (lo, hi).  After a pack_shared exchange the context holds
        only this rank's private pairs + the travelling pairs of all ranks: that list is handed out only when asked for
        by name (SlabJob.result_arrays assembles the global list from it)."""
        if not allow_partial and self.adjacency_scope() == ADJ_PARTIAL:
            raise TissueScanError(TA_EINVAL, "this context holds a PARTIAL pair list (private + travelling pairs of a "
                                             "slab exchange): use SlabJob.result_arrays() for the global list")
        n = ctypes.c_int64(0)
        _check(self._lib.ta_adjacency_size(self._h, ctypes.byref(n)))
        lo = np.zeros(n.value, dtype=np.uint32)
        hi = np.zeros(n.value, dtype=np.uint32)
        faces = np.zeros((n.value, 3), dtype=np.uint64)
        _check(self._lib.ta_adjacency_get(self._h, lo.ctypes.data, hi.ctypes.data, faces.ctypes.data))
        return lo, hi, faces

    def timing(self):
        a, b, t = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        nbytes = ctypes.c_uint64(0)
        _check(self._lib.ta_timing(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(t),
                                   ctypes.byref(nbytes)))
        # (a duration no event recorded comes back as NaN -- TA_OPT_TIMING 0, or 1 for what only mode 2 brackets -- and is
        #  handed on as None: "not measured" must never read as "took no time" in a bandwidth figure)
        some = lambda v: None if v != v else v
        return dict(ms_sweep=some(a.value), ms_adjacency=some(b.value), ms_total=some(t.value), bytes_read=nbytes.value)

    def timing_series(self, capacity=4096):
        """Sweep-kernel milliseconds of the last extractions (oldest first; OPT_TIMING_RING of them at most)."""
        buf = (ctypes.c_double * int(capacity))()
        n = ctypes.c_int(0)
        _check(self._lib.ta_timing_series(self._h, buf, int(capacity), ctypes.byref(n)))
        return [buf[i] for i in range(n.value)]

    def read_probe(self, dev_ptr, nbytes, repeats=5):
        """Milliseconds of the fastest of `repeats` read-only streaming passes over a device buffer."""
        ms = ctypes.c_double(0.0)
        _check(self._lib.ta_read_probe(self._h, ctypes.c_void_p(int(dev_ptr)), int(nbytes), int(repeats), ctypes.byref(ms)))
        return ms.value

    def debug_counters(self):
        out = (ctypes.c_uint32 * 16)()
        _check(self._lib.ta_debug_counters(self._h, out))
        d = dict(range_flag=out[0], pair_overflow=out[1], label_spills=out[2], pair_spills=out[3])
        if any(out[8:16]):
            d["stamps"] = [int(v) for v in out[8:16]]
        return d

    # -- multi-GPU views
    def bind_accumulators(self, sums_ptr, boxes_ptr, max_label, keep=None):
        _check(self._lib.ta_bind_accumulators(self._h, ctypes.c_void_p(int(sums_ptr) if sums_ptr else 0),
                                              ctypes.c_void_p(int(boxes_ptr) if boxes_ptr else 0),
                                              int(max_label)))
        self._keep_acc = keep

    def accumulators_reduced(self):
        """The bound accumulators now hold other ranks' contributions (see include/tissue_scan.h)."""
        _check(self._lib.ta_accumulators_reduced(self._h))

    def adjacency_device(self):
        k, f, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64(0)
        _check(self._lib.ta_adjacency_device(self._h, ctypes.byref(k), ctypes.byref(f), ctypes.byref(n)))
        return k.value, f.value, n.value

    def adjacency_export(self, keys_ptr, faces_ptr, capacity):
        _check(self._lib.ta_adjacency_export(self._h, ctypes.c_void_p(int(keys_ptr) if keys_ptr else 0),
                                             ctypes.c_void_p(int(faces_ptr) if faces_ptr else 0), int(capacity)))

    def adjacency_merge(self, keys_ptr, faces_ptr, npairs):
        _check(self._lib.ta_adjacency_merge(self._h, ctypes.c_void_p(int(keys_ptr) if keys_ptr else 0),
                                            ctypes.c_void_p(int(faces_ptr) if faces_ptr else 0), int(npairs)))

    def adjacency_pack(self, block_ptr, capacity):
        """Enqueue: write this rank's exchange block (exchange_words(capacity) uint64 words)."""
        _check(self._lib.ta_adjacency_pack(self._h, ctypes.c_void_p(int(block_ptr)), int(capacity)))

    def adjacency_pack_shared(self, block_ptr, capacity):
        """Enqueue: keep the pairs no other rank can hold (needs the REDUCED boxes), pack the rest."""
        _check(self._lib.ta_adjacency_pack_shared(self._h, ctypes.c_void_p(int(block_ptr)), int(capacity)))

    def adjacency_merge_blocks(self, blocks_ptr, nblocks, capacity):
        """Enqueue: rebuild the adjacency from all ranks' gathered exchange blocks."""
        _check(self._lib.ta_adjacency_merge_blocks(self._h, ctypes.c_void_p(int(blocks_ptr)), int(nblocks),
                                                   int(capacity)))

    # -- synthetic workload + raw memory
    def synth_voronoi(self, dev_ptr, dtype, dims, a_begin, a_count, seeds, grid, ell=None):
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        g = (ctypes.c_int32 * 3)(*[int(x) for x in grid])
        ell_arr = None if ell is None else np.ascontiguousarray(np.concatenate(ell), dtype=np.int64)
        _check(self._lib.ta_synth_voronoi(self._h, ctypes.c_void_p(int(dev_ptr)), np.dtype(dtype).itemsize,
                                          _i64x3(dims), int(a_begin), int(a_count), seeds.ctypes.data, g,
                                          None if ell_arr is None else ell_arr.ctypes.data))

    def malloc(self, nbytes):
        p = ctypes.c_void_p()
        _check(self._lib.ta_device_malloc(self._h, int(nbytes), ctypes.byref(p)))
        return p.value

    def free(self, ptr):
        _check(self._lib.ta_device_free(self._h, ctypes.c_void_p(int(ptr))))

    def d2h(self, host_array, dev_ptr):
        _check(self._lib.ta_memcpy_d2h(self._h, host_array.ctypes.data, ctypes.c_void_p(int(dev_ptr)),
                                       host_array.nbytes))

    def h2d(self, dev_ptr, host_array):
        a = np.ascontiguousarray(host_array)
        _check(self._lib.ta_memcpy_h2d(self._h, ctypes.c_void_p(int(dev_ptr)), a.ctypes.data, a.nbytes))
