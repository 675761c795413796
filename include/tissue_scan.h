/* tissue_scan.h -- C ABI of the MI355X-native per-label voxel-scan library (libtissue_scan.so).
 *
 * The reference (VirtualPlants/tissue_analysis, pure Python 2) has NO plugin / operator / FFI
 * interface for this path: its boundary is the Python class API of
 * src/vplants/tissue_analysis/spatial_image_analysis.py (SIA).  This header is therefore the
 * binding a maintainer of the reference would add underneath that class; every entry point
 * cites the reference code whose arithmetic it replaces.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *  - every function returns TA_OK (0) or a negative TA_E* code; ta_last_error() gives the text
 *    (thread-local, valid until the next call on that thread);
 *  - no C++ exceptions, Python callbacks, or torch types cross this boundary -- plain pointers
 *    and sizes only; all host output buffers are caller-allocated, the library never returns
 *    owned host memory;
 *  - a ta_ctx is bound to ONE GPU (one process per GPU; multi-GPU = one context per rank plus
 *    RCCL collectives issued by the host on the device buffers exposed below); a context is
 *    not thread-safe, distinct contexts may be used from different threads;
 *  - axes: "memory axis 0" is the slowest-varying axis of the volume, axis 2 the fastest.
 *    ta_volume_set() accepts any dense permuted layout (C- or F-ordered numpy arrays) and the
 *    host getters report per-axis outputs in ARRAY axis order; the device-resident entry points
 *    take dense C-ordered buffers (array order == memory order);
 *  - all per-label outputs are exact integers.  Float work (barycentre = sum1/count, covariance,
 *    eigen-decomposition, real-unit scaling) is host-side float64 on these integers.
 */
#ifndef TISSUE_SCAN_H
#define TISSUE_SCAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ta_wall_voxels_get takes (pairs, coords, ms); ta_ctx_set_stream(NULL) = the device's legacy default stream;
 *    TA_OPT_IMPL is 0 or 1; ta_adjacency_scope added; ta_timing answers zeros when no events were recorded.
 * 3: ta_volume_plane_events, ta_wall_medians(_get); ta_timing answers NaN for what was not measured.
 * 4: sparse label ids: ta_volume_label_census, ta_label_census_get, ta_volume_compact_labels, ta_volume_is_compact.
 * 5: ta_volume_rerank, ta_volume_uncompact, ta_volume_owned_planes; ta_wall_medians marks a wall that did not settle (bit 31 of its
 *    size word) instead of failing the call; TA_OPT_SWEEP_SHAPE -1 decides by label-change density, -2 by four timed sweeps.
 * A caller checks ta_version() == TA_ABI_VERSION of the header it was built against (the ctypes binding does). */
#define TA_ABI_VERSION 5

#if defined(TA_BUILD)
#define TA_API __attribute__((visibility("default")))
#else
#define TA_API
#endif

/* status codes */
#define TA_OK          0
#define TA_EINVAL     -1  /* bad argument / call order                                   */
#define TA_EHIP       -2  /* HIP runtime error (text in ta_last_error)                   */
#define TA_ENOMEM     -3  /* host or device allocation failed                            */
#define TA_ERANGE     -4  /* the volume holds a label above max_label                    */
#define TA_ECAPACITY  -5  /* adjacency table / exchange block too small (see the call's doc)  */
#define TA_ENODEVICE  -6  /* no usable GPU                                               */

/* feature mask of ta_extract() */
#define TA_F_VOLUME     1u  /* count[l]                 ~ nd.sum(ones, image, index)  SIA:1231          */
#define TA_F_BBOX       2u  /* bbox[l]                  ~ nd.find_objects(image)      SIA:517           */
#define TA_F_MOMENT1    4u  /* sum1[l][3]               ~ nd.center_of_mass per label SIA:464-467       */
#define TA_F_MOMENT2    8u  /* sum2[l][6]               ~ cov = P.P^T/N               SIA:137-150,1276  */
#define TA_F_ADJACENCY 16u  /* faces[(lo,hi)][3]        ~ binary_dilation shells      SIA:45-60,947-956 */
#define TA_F_ALL       31u

/* ta_ctx_set_option keys */
#define TA_OPT_IMPL        1  /* 0 = default: the sweep kernel (runs along the contiguous axis, scan-allocated record
                                 buffers, hand-issued plane loads); 1 = per-voxel global atomics (slow; shares no logic
                                 with the sweep: the two cross-check each other on the GPU) */
#define TA_OPT_TILE_PLANES 2  /* planes of memory axis 0 walked by one workgroup (tuning)          */
#define TA_OPT_PAIR_SLOTS  3  /* log2 of the device adjacency hash capacity (0 = automatic)       */
#define TA_OPT_TIMING      4  /* HIP events per extraction: 0 none, 1 = begin / end of the sweep kernel (default; they ride on
                               * the kernel's own launch, hipExtLaunchKernelGGL: no packet of their own on the queue),
                               * 2 = also the step's begin / end (two event records, ~4 us of queue time each)      */
#define TA_OPT_SWEEP_SHAPE 7  /* uint32 volumes with adjacency: the tile shape of the sweep.  1: a wave walks two rows of 512 columns
                               * (eight voxels a lane, four waves per SIMD); 0: two rows of 256 (four voxels a lane, five waves).
                               * Same results; 1 is ~5 % faster on tissue with background around it, 0 on a volume that is cells
                               * everywhere, and on rows that are not whole 512-column tiles.  -1 (default): rows that are not take
                               * 0; otherwise decided BEFORE the volume's first sweep from the label changes per voxel in eight
                               * sampled planes (above 0.032: shape 0) -- one small read-back, one stream synchronisation, once per
                               * resident volume.  -2: the first four sweeps of a volume take turns between two events each and the
                               * faster shape keeps the volume (round 4's rule). */
#define TA_OPT_SWEEP_SHAPE_USED 8 /* read only: the shape the last sweep of this context ran with (0 where the shape does not apply) */
#define TA_OPT_VOLUME_SLACK 6 /* bytes that are readable behind the volume adopted by ta_volume_set_device (reset to 0 by that
                               * call): with >= 16 the sweep uses 16-byte loads whatever the row length -- the strip that
                               * straddles the end of the last row reads up to 16 - itemsize bytes past the volume        */
#define TA_OPT_TIMING_RING 5  /* sweep durations kept for ta_timing_series: the last N extractions, N in [1,4096]
                               * (default 1); setting it drains the stream and starts a new series             */

typedef struct ta_ctx ta_ctx;

TA_API int         ta_version(void);
TA_API const char* ta_last_error(void);
TA_API int         ta_device_count(int* count);

/* One context per GPU.  Owns a HIP stream (unless ta_ctx_set_stream replaces it), the resident
 * volume (when uploaded with ta_volume_set), accumulators, the adjacency hash and timing events. */
TA_API int ta_ctx_create(int device_id, ta_ctx** out);
TA_API int ta_ctx_destroy(ta_ctx* ctx);
/* hip_stream: a caller-owned hipStream_t; NULL = a private non-blocking stream owned by the context (the default);
 * TA_STREAM_LEGACY_DEFAULT = the device's legacy default ("null") stream, e.g. torch's default stream. */
#define TA_STREAM_LEGACY_DEFAULT ((void*)1)
TA_API int ta_ctx_set_stream(ta_ctx* ctx, void* hip_stream);
TA_API int ta_ctx_set_option(ta_ctx* ctx, int key, int64_t value);
/* Current effective value of an option (TA_OPT_PAIR_SLOTS: log2 of the table in use, which may
 * have grown past the requested size). */
TA_API int ta_ctx_get_option(ta_ctx* ctx, int key, int64_t* value);
TA_API int ta_ctx_synchronize(ta_ctx* ctx);

/* Upload a host volume (SIA:225-227 "self.image").  itemsize 2 (uint16) or 4 (uint32).
 * strides_bytes == NULL means dense C order; otherwise the strides must describe a dense layout
 * in SOME axis permutation (C/F-ordered or transposed arrays).  The host buffer stays caller-owned
 * and may be freed after return. */
TA_API int ta_volume_set(ta_ctx* ctx, const void* host_ptr, int itemsize,
                  const int64_t dims[3], const int64_t strides_bytes[3]);

/* Adopt a volume already resident in this GPU's HBM (dense C order, not copied, not owned).
 * buf_dims[0] counts every plane in the buffer.  For a Z-slab of a larger volume
 * a0_origin is the global axis-0 coordinate of the first OWNED plane and has_low_halo != 0 says
 * that plane 0 of the buffer is the neighbour slab's last plane: it only contributes the faces
 * it shares with plane 1 (a face belongs to the slab owning its higher voxel). */
TA_API int ta_volume_set_device(ta_ctx* ctx, const void* dev_ptr, int itemsize,
                         const int64_t buf_dims[3], int64_t a0_origin, int has_low_halo);

/* Largest label in the resident volume (device max-reduction; ~ np.unique(image) SIA:363). */
TA_API int ta_volume_max_label(ta_ctx* ctx, uint32_t* max_label);

/* ---- sparse label ids (np.unique takes any ids, SIA:358-364; the sweep keeps one 104-byte row per id 0..max_label) --------
 * ta_volume_label_census: which ids does the resident volume (halo plane included) hold -- np.unique on the device: one
 *   presence-bitmap pass over the volume, a prefix count over (max_label + 1) / 32 words.  ta_label_census_get: the ids, ascending.
 * ta_volume_compact_labels: from then on the SWEEP reads a context-owned copy of the volume rewritten in the RANKS of its ids
 *   (0 .. n_rows - 1, order-preserving); the resident volume itself is not touched, so the wall voxels, the label maps and the
 *   voxel layers keep working on ids.  In a compacted context every per-label ROW (ta_get_labels, ta_accumulators_device,
 *   ta_bind_accumulators: pass max_label = n_rows - 1 to ta_extract) is indexed by rank, and every label VALUE a getter hands
 *   out (ta_adjacency_get) is an id; ta_label_census_get is the rank -> id table.  ids == NULL: the census of this volume;
 *   otherwise a HOST list, ascending and unique, that must cover the volume (TA_ERANGE if it does not): the union over the ranks
 *   of a partitioned volume, so that every slab ranks alike and the device-side adjacency exchange works in rank space.
 *   The label TABLES a caller hands in are per-label rows too: in a compacted context ta_volume_relabel takes one entry per
 *   rank (lut_len == n_rows, the entries are ids: v -> lut[rank(v)]) and ta_volume_map answers out[p] = lut[rank(V[p])].
 *   More than 2^28 - 1 ids present => TA_ERANGE.  A new volume or ta_volume_relabel ends the compacted state.
 * New in TA_ABI_VERSION 4. */
TA_API int ta_volume_label_census(ta_ctx* ctx, uint32_t* max_label, uint32_t* n_present);
TA_API int ta_label_census_get(ta_ctx* ctx, uint32_t* ids /* [n_present] */);
TA_API int ta_volume_compact_labels(ta_ctx* ctx, const uint32_t* ids, uint32_t n_ids, uint32_t* n_rows);
/* COMPACTION IS A SNAPSHOT OF THE VOXELS: the rank copy is written once, by ta_volume_compact_labels, and the sweep reads it from
 * then on.  A caller that rewrites an adopted device buffer in place (the next frame of a time series, a refreshed halo plane)
 * calls ta_volume_rerank before the next ta_extract: the same copy pass again, with the census the context already holds,
 * asynchronous on the context's stream (~copy bandwidth: 1.7 ms for 1024^3 uint32).  An id the list does not hold makes the
 * getters of the next extraction answer TA_ERANGE.  ta_volume_uncompact leaves the compacted state (dense rows 0 .. max_label
 * again) and releases the rank copy.  New in TA_ABI_VERSION 5. */
TA_API int ta_volume_rerank(ta_ctx* ctx);
TA_API int ta_volume_uncompact(ta_ctx* ctx);
/* planes along the slowest memory axis that this context owns (the halo plane excluded): the length of ta_volume_plane_events' output */
TA_API int ta_volume_owned_planes(ta_ctx* ctx, int64_t* planes);
TA_API int ta_volume_is_compact(ta_ctx* ctx, int* compact, uint32_t* n_rows);

/* events[p] = label changes along memory axis 2 in OWNED plane p of the resident volume (one streaming pass; the halo
 * plane of a slab is not counted): what a record-producing plane costs the sweep on top of its voxels -- the weight a
 * Z-slab partition balances (tissue_analysis_amd/distributed.py: plane_costs, balanced_cuts).  New in
 * TA_ABI_VERSION 3; nothing in the reference to mirror. */
TA_API int ta_volume_plane_events(ta_ctx* ctx, uint64_t* events /* [owned planes] */);

/* The hot path: one fused sweep of the resident volume + adjacency compaction.
 * Replaces the four per-label Python loops of the reference (SIA:417-480, 632-660, 908-993,
 * 1246-1292) and its whole-volume scans (SIA:517, 1231).  Asynchronous on the context stream;
 * the getters synchronise.  Labels above max_label => TA_ERANGE (reported by the getters too). */
TA_API int ta_extract(ta_ctx* ctx, uint32_t feature_mask, uint32_t max_label);

/* Per-label results, rows 0..max_label, any pointer may be NULL:
 *   count [L+1]     voxels per label
 *   bbox  [L+1][6]  min0,min1,min2,max0+1,max1+1,max2+1 ; -1 when the label is absent
 *   sum1  [L+1][3]  sum of coordinates
 *   sum2  [L+1][6]  sum of coordinate products, order 00,01,02,11,12,22; zero when the extraction did not ask for
 *                   TA_F_MOMENT2 (the device columns behind ta_accumulators_device are then undefined) */
TA_API int ta_get_labels(ta_ctx* ctx, uint64_t* count, int32_t* bbox, uint64_t* sum1, uint64_t* sum2);

/* Face-neighbour adjacency (size-then-fill): unique pairs lo<hi sorted by (lo,hi);
 * faces[i][d] = number of voxel faces normal to ARRAY axis d shared by the pair. */
TA_API int ta_adjacency_size(ta_ctx* ctx, int64_t* npairs);
TA_API int ta_adjacency_get(ta_ctx* ctx, uint32_t* lo, uint32_t* hi, uint64_t* faces);

/* Timing of the last ta_extract (HIP events on the context stream): the sweep kernel alone; with TA_OPT_TIMING = 2
 * also what follows it (fold of the per-workgroup hot-label rows + adjacency collection) and the whole call from
 * the accumulator init on.  A duration that no event recorded (TA_OPT_TIMING 0; the last two under TA_OPT_TIMING 1)
 * is answered as NaN -- "not measured", never 0; bytes_read = nvox * itemsize (algorithmic bytes).
 * ta_timing_series: the sweep kernel's duration of each of the last extractions (oldest first, at most capacity and
 * at most TA_OPT_TIMING_RING of them) -- drains the stream; how a host times every launch of a pipelined loop. */
TA_API int ta_timing_series(ta_ctx* ctx, double* ms_sweep, int capacity, int* count);
TA_API int ta_timing(ta_ctx* ctx, double* ms_sweep, double* ms_adjacency, double* ms_total,
              uint64_t* bytes_read);

/* Read-bandwidth probe: the best of `repeats` timed launches (HIP events on the context stream) of a
 * trivial 16-bytes-per-lane read + XOR-reduce kernel over `bytes` of device memory: the streaming ceiling the box
 * actually reaches, next to the 8 TB/s of the data sheet.  Reads only. */
TA_API int ta_read_probe(ta_ctx* ctx, const void* dev_ptr, uint64_t bytes, int repeats, double* ms_best);

/* Diagnostics of the last ta_extract: out[0] = label-range flag, out[1] = adjacency-table overflow
 * flag, out[2] = run records that missed the workgroup LDS label table (went to global atomics),
 * out[3] = face records that missed the LDS pair table, out[4..15] reserved (cycle stamps of
 * diagnostic builds). */
TA_API int ta_debug_counters(ta_ctx* ctx, uint32_t out[16]);

/* ---- device-side views for the multi-GPU reduce (RCCL runs on these in place) -------------
 * sums  : uint64 [L+1][10] = count, s0, s1, s2, s00, s01, s02, s11, s12, s22   (reduce: SUM)
 * boxes : int32  [L+1][6]  = min0, min1, min2, -max0, -max1, -max2 (inclusive max, INT32_MAX
 *                            when absent)                                      (reduce: MIN)
 * By default the context owns them; ta_bind_accumulators lets the caller supply device memory
 * (e.g. torch tensors) sized for max_label so collectives need no copy; ta_get_labels reads
 * whatever the buffers hold at call time, i.e. the reduced values after an all-reduce. */
TA_API int ta_bind_accumulators(ta_ctx* ctx, void* sums_dev, void* boxes_dev, uint32_t max_label);
TA_API int ta_accumulators_device(ta_ctx* ctx, void** sums_dev, void** boxes_dev, uint32_t* max_label);
/* Tell the context that the bound accumulators have been reduced across ranks since the last ta_extract: a later
 * adjacency-table overflow then returns TA_ECAPACITY (with the table already grown) instead of silently re-running
 * the sweep on this rank alone, which would replace the global rows by local ones. */
TA_API int ta_accumulators_reduced(ta_ctx* ctx);

/* Unsorted unique pairs of the last extraction, on the device:
 * keys uint64[n] = lo<<32|hi, faces uint64[n][3] (memory-axis order). */
TA_API int ta_adjacency_device(ta_ctx* ctx, void** keys_dev, void** faces_dev, int64_t* npairs);

/* Copy the unsorted unique pairs into caller-owned device buffers (e.g. torch tensors that an
 * RCCL all-gather will send); capacity_pairs must be >= the current pair count. */
TA_API int ta_adjacency_export(ta_ctx* ctx, void* keys_dst_dev, void* faces_dst_dev, int64_t capacity_pairs);

/* Merge foreign pair lists (other ranks' ta_adjacency_device output, gathered by the host with
 * RCCL) into this context's adjacency: sums face counts of equal keys. */
TA_API int ta_adjacency_merge(ta_ctx* ctx, const void* keys_dev, const void* faces_dev, int64_t npairs);

/* Which pairs the adjacency getters (ta_adjacency_size / _get / _device / _export) answer with right now:
 *   TA_ADJ_LOCAL    every pair of this context's own volume (after ta_extract);
 *   TA_ADJ_MERGED   the pairs of ALL ranks' blocks (after ta_adjacency_pack on every rank + ta_adjacency_merge_blocks);
 *   TA_ADJ_PARTIAL  this rank's PRIVATE pairs plus all ranks' travelling pairs (after ta_adjacency_pack_shared +
 *                   ta_adjacency_merge_blocks): NOT the list of any volume -- the global list is the union over ranks. */
#define TA_ADJ_LOCAL   0
#define TA_ADJ_MERGED  1
#define TA_ADJ_PARTIAL 2
TA_API int ta_adjacency_scope(ta_ctx* ctx, int* scope);

/* ---- label lookup-table sweeps over the resident volume -------------------
 * ta_volume_relabel: in place, v -> lut[v] for v < lut_len, other voxels unchanged.  Replaces the
 * per-label bounding-box loops of fuse_labels_in_image / remove_labels_from_image (SIA:1114-1165).
 * lut is a HOST array; with a uint16 volume every entry must be <= 65535.  Invalidates the last
 * extraction.  Not allowed on a slab with a halo plane (the neighbour owns that plane).
 * ta_volume_get: copy the resident volume back to the host, same dense layout as it was set with.
 * ta_volume_map: out[p] = lut[V[p]] (fill for V[p] >= lut_len) into a HOST image of out_itemsize
 * (1, 2, 4 or 8) byte words, same layout as the volume; lut and fill are words of that size
 * (PropertySpatialImage.create_property_image, PSI:207-221). */
TA_API int ta_volume_relabel(ta_ctx* ctx, const uint32_t* lut, uint32_t lut_len);
TA_API int ta_volume_get(ta_ctx* ctx, void* host_dst);
TA_API int ta_volume_map(ta_ctx* ctx, const void* lut, uint32_t lut_len, const void* fill, int out_itemsize,
                         void* host_dst);

/* ---- first voxel layer (voxel_first_layer, SIA:1024-1046) -------------------------
 * out[p] = V[p] when V[p] != background and one of the six face neighbours of p is background; 1 (keep_background)
 * or 0 where V[p] == background; 0 elsewhere -- `image * (dilate6(mask) - mask) + mask` in one stencil pass.  The host
 * image has the dtype and dense layout of the volume given to ta_volume_set.  Not available on a slab with a halo. */
TA_API int ta_volume_first_layer(ta_ctx* ctx, uint32_t background, int keep_background, void* host_dst);

/* ---- hollowed-out cells and voxel layers (SURVEY.md §2 component 3: hollow_out_cells SIA:74-95, cells_walls_coords
 * SIA:883-905, cells_voxel_layer SIA:1399-1448) ------------------------------------------------------------------------
 * ta_volume_hollow: out[p] = V[p] where the Laplacian of the labels at p -- the sum of the six face neighbours minus
 * 6 V[p], modulo 2^label_bits (the integer type of the image the caller holds: 8, 16, 32 or 64; 0 = the volume's own
 * 16 / 32), the edge voxel repeated outside the image: what scipy.ndimage.laplace gives on an integer image -- is not
 * zero and, with remove_background, V[p] != background; 0 elsewhere.  The host image has the dtype and dense layout of the
 * volume given to ta_volume_set.
 * ta_volume_layer18: out[p] = 1 where one of the 18 neighbours of p (faces + edges, inside the image) carries another
 * label, else 0: `mask - binary_erosion(mask, generate_binary_structure(3, 2))` of EVERY label at once, up to the voxels
 * on the faces of the crop the erosion is taken in, which the caller adds.  One byte per voxel, same layout.
 * Neither is available on a slab with a halo plane. */
TA_API int ta_volume_hollow(ta_ctx* ctx, uint32_t background, int remove_background, int label_bits, void* host_dst);
TA_API int ta_volume_layer18(ta_ctx* ctx, uint8_t* host_dst);

/* ---- wall voxels ------------------------------------------------------------
 * A voxel p of label l is a wall voxel of the pair (l, m) when one of its 18 neighbours (faces and
 * edges: scipy generate_binary_structure(3, 2), SIA:796-799) carries a label m != l; this is
 * (dil(mask_l) & mask_m) | (dil(mask_m) & mask_l) of wall_voxels_between_two_cells (SIA:759-806)
 * for every pair at once.  ta_wall_voxels_count runs the one compute pass -- it finds every record and keeps it in a
 * staging area on the device (about 6 bytes per voxel of the volume, held until the volume changes), counts per row
 * strip, scanned on the device: the host reads back one line -- and returns the number of (pair, voxel) records;
 * ta_wall_voxels_get moves them into place and fills caller-allocated arrays of that length: the labels
 * pairs[r] = (lo, hi), lo < hi, and the voxel's coordinates in ARRAY-axis order, records ordered by the
 * voxel's position in memory (for a C-ordered array: np.where order; the records of one voxel in no particular
 * order).  ms (optional) receives the duration of the kernels of both calls.  Not available on a slab with a halo
 * plane.  Environment: TA_WALL_STAGE_RECORDS=<n> sizes the staging area (0: none, every strip is recomputed by the
 * fetch -- the fallback noise volumes take anyway), TA_WALL_VERBOSE=1 prints one line per count. */
TA_API int ta_wall_voxels_count(ta_ctx* ctx, int64_t* nrecords);
TA_API int ta_wall_voxels_get(ta_ctx* ctx, uint32_t* pairs /* [n][2] */, int32_t* coords /* [n][3] */, double* ms);
/* The same records GROUPED BY PAIR: sorted by (lo, hi) on the device (stable radix sort), the voxels of one pair still
 * in memory order -- what wall_voxels_between_two_cells / _per_cell / _per_cells_pairs look up (SIA:759-880). */
TA_API int ta_wall_voxels_get_by_pair(ta_ctx* ctx, uint32_t* pairs /* [n][2] */, int32_t* coords /* [n][3] */, double* ms);

/* The median voxel of every wall, computed on the device (TGI:210-242 through SIA:1586-1635; new in TA_ABI_VERSION 3, the
 * reference has no native interface to mirror): after ta_wall_voxels_count, ta_wall_medians groups the records by pair on
 * the device, runs the reference's Weiszfeld iteration (its start, its stopping rule, IEEE double sums in record order: the
 * arithmetic of tissue_analysis_amd/geometry.py::weiszfeld_segments) on every wall -- one wave a wall --, truncates the
 * position and picks the wall voxel nearest to it (the first one on ties).  *nwalls = number of walls E; the results stay
 * on the context until the volume changes.  ta_wall_medians_get copies them out, sorted by (lo, hi): pairs u32[E][2], the
 * walls' voxel counts u32[E], the median voxels i32[E][3] in array-axis order.  A wall that is still moving after max_iter
 * passes (the reference raises there) is MARKED, not refused: bit 31 of its count word is set and its median is not to be
 * used -- the caller raises for the walls it asked for (TA_ABI_VERSION 5; before, one such wall failed the whole call with
 * TA_EINVAL).  Volumes in memory (C) order only (TA_EINVAL otherwise): the order of a wall's voxels decides ties. */
TA_API int ta_wall_medians(ta_ctx* ctx, int max_iter, int64_t* nwalls, double* ms_kernels);
TA_API int ta_wall_medians_get(ta_ctx* ctx, uint32_t* pairs, uint32_t* sizes, int32_t* medians);

/* ---- stream-ordered adjacency exchange (no host round trip) ------------------
 * One exchange block per rank, uint64 words, TA_EXCHANGE_WORDS(capacity) long:
 *   [0] pair count (may exceed capacity)  [1] status bits  [2..2+cap) keys, ~0 padded
 *   [2+cap .. 2+4*cap) faces[cap][3]
 * ta_adjacency_pack writes this rank's block from the last extraction; the host all-gathers the
 * blocks (RCCL, same stream); ta_adjacency_merge_blocks rebuilds the context's adjacency from ALL
 * nblocks blocks (this rank's included).  Both only enqueue work.  Range / table-overflow flags
 * travel in the status word, so every rank reaches the same verdict: the next result getter
 * (ta_adjacency_size / _get / _device, ta_get_labels) returns TA_ECAPACITY when some block or
 * table was too small -- raise the capacity (or TA_OPT_PAIR_SLOTS) on every rank and redo the
 * step -- and TA_ERANGE when any rank saw a label above max_label.
 *
 * ta_adjacency_pack_shared is the cheaper pack for slabs: call it AFTER the bound per-label boxes were reduced over
 * all ranks.  A label whose global axis-0 extent lies inside this slab's planes [lo, hi - 2] exists on no other rank
 * (nor in its halo), so a pair with such a label is final here and stays (it is put back into this context's table);
 * only the remaining pairs -- those a slab face can split -- are written to the block.  After
 * ta_adjacency_merge_blocks the context then holds its PRIVATE pairs plus ALL ranks' travelling pairs merged; the
 * global list is the union over ranks (private lists are disjoint, the merged part is the same everywhere).  Without
 * TA_F_BBOX in the feature mask no label is exclusive and the call packs everything, like ta_adjacency_pack. */
#define TA_EXCHANGE_WORDS(capacity_pairs) (2 + 4 * (int64_t)(capacity_pairs))
TA_API int ta_adjacency_pack(ta_ctx* ctx, void* block_dev, int64_t capacity_pairs);
TA_API int ta_adjacency_pack_shared(ta_ctx* ctx, void* block_dev, int64_t capacity_pairs);
TA_API int ta_adjacency_merge_blocks(ta_ctx* ctx, const void* blocks_dev, int nblocks, int64_t capacity_pairs);

/* Synthetic workload generator, bit-identical to tissue_analysis_amd/synth.py:
 * writes planes [a_begin, a_begin+a_count) of the dims[] Voronoi volume to dev_out (dense C).
 * seeds: host int32 [G0*G1*G2][3]; ell: host int64 tables E0|E1|E2 concatenated, or NULL. */
TA_API int ta_synth_voronoi(ta_ctx* ctx, void* dev_out, int itemsize, const int64_t dims[3],
                     int64_t a_begin, int64_t a_count, const int32_t* seeds,
                     const int32_t grid[3], const int64_t* ell);

/* Raw device memory helpers so a host without torch can still stage buffers. */
TA_API int ta_device_malloc(ta_ctx* ctx, uint64_t bytes, void** dev_ptr);
TA_API int ta_device_free(ta_ctx* ctx, void* dev_ptr);
TA_API int ta_memcpy_d2h(ta_ctx* ctx, void* host_dst, const void* dev_src, uint64_t bytes);
TA_API int ta_memcpy_h2d(ta_ctx* ctx, void* dev_dst, const void* host_src, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* TISSUE_SCAN_H */
