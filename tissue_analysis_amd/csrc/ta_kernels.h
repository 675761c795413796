// ta_kernels.h -- host-callable launchers implemented in the kernels_*.hip files.
#pragma once
#include "ta_device.h"

namespace ta {

// kernels_basic.hip
void launch_init_accumulators(hipStream_t s, uint64_t* sums, int32_t* boxes, uint64_t nlabels,
                              uint32_t* flags, uint32_t* pair_cursor, uint64_t* hot_rows);
void launch_naive(hipStream_t s, const SweepArgs& a, int itemsize, uint32_t feature_mask);
void launch_max_label(hipStream_t s, const void* vol, int itemsize, uint64_t nvox, uint32_t* out_dev);
void launch_plane_events(hipStream_t s, const void* vol, int itemsize, int64_t planes, int64_t n1, int64_t n2, uint64_t* out_dev);
void launch_pairs_collect(hipStream_t s, const PairTable& pt, uint64_t* out_keys, uint64_t* out_faces,
                          uint32_t* cursor);
void launch_pairs_insert(hipStream_t s, const PairTable& pt, const uint64_t* keys, const uint64_t* faces,
                         uint64_t n, uint32_t* flags);
void launch_pairs_clear(hipStream_t s, const PairTable& pt);
void launch_pairs_pack(hipStream_t s, const uint64_t* keys, const uint64_t* faces, const uint32_t* cursor,
                       const uint32_t* flags, uint64_t* block, uint64_t cap);
void launch_pairs_pack_shared(hipStream_t s, const PairTable& pt, const uint64_t* keys, const uint64_t* faces,
                              const uint32_t* cursor, uint32_t* flags, const int32_t* boxes, uint32_t max_label,
                              int64_t lo, int64_t hi, uint64_t* block, uint64_t cap, uint64_t npairs_bound);
void launch_pairs_insert_blocks(hipStream_t s, const PairTable& pt, const uint64_t* blocks, int nblocks,
                                uint64_t cap, uint32_t* flags);
void launch_pairs_collect_hot(hipStream_t s, const PairTable& pt, uint64_t* out_keys, uint64_t* out_faces, uint32_t* cursor,
                              const SweepArgs& a, int itemsize, const uint64_t* hot_rows, uint32_t nrows);
void launch_hot_reduce(hipStream_t s, const SweepArgs& a, int itemsize, const uint64_t* hot_rows, uint32_t nrows,
                       uint32_t* publish = nullptr, int nwords = 0);
void launch_relabel(hipStream_t s, const void* src, void* vol, int itemsize, uint64_t n, const uint32_t* lut, uint32_t lut_len);
void launch_map(hipStream_t s, const void* vol, int itemsize, void* out, int out_itemsize, uint64_t n,
                const void* lut, uint32_t lut_len, uint64_t fill);
void launch_read_probe(hipStream_t s, const void* p, uint64_t bytes, uint32_t* sink);
void launch_first_layer(hipStream_t s, const void* vol, int itemsize, void* out, int64_t n0, int64_t n1, int64_t n2,
                        uint32_t background, int keep_background);

void launch_hollow(hipStream_t s, const void* vol, int itemsize, void* out, int64_t n0, int64_t n1, int64_t n2, uint32_t background,
                   int remove_background, int label_bits);
void launch_layer18(hipStream_t s, const void* vol, int itemsize, uint8_t* out, int64_t n0, int64_t n1, int64_t n2);

// kernels_walls.hip -- wall voxels: count + stage per (row, strip) cell, device scan, copy in memory order (+ a second walk for
// the cells that were not staged)
struct WallPlan { int32_t nstrips, rows_per_wave; uint64_t cells, scan_blocks, waves; };
WallPlan wall_plan(int64_t n0, int64_t n1, int64_t n2);
struct WallBuffers {
    uint32_t* counts;        // [cells]
    uint32_t* cell_base;     // [cells]
    uint8_t* lane_counts;    // [cells][64]
    uint64_t* offsets;       // [cells]
    uint64_t* block_sums;    // [scan_blocks]
    uint64_t* total;         // total u64 | status u32[6] (cells not staged, wide label seen, OR of all labels, -): the 32-byte line the host reads back
    uint32_t* status;        // = (uint32_t*)(total + 1)
    void* stage;             // wall_stage_bytes(region) or NULL: nothing is staged, every cell takes the second walk
    uint32_t* cursors;       // wall_cursor_bytes()
    uint32_t* todo;          // [cells] the cells left to the second walk
    uint32_t region;         // records per staging region
};
uint64_t wall_stage_bytes(uint64_t records_per_region, int itemsize);      // 8-byte records for uint16 volumes, 12-byte ones for uint32
uint64_t wall_cursor_bytes();
uint32_t wall_stage_regions();
void launch_wall_count(hipStream_t s, const void* vol, int itemsize, int64_t n0, int64_t n1, int64_t n2, const WallBuffers& b,
                       bool wide);
void launch_wall_fetch(hipStream_t s, const void* vol, int itemsize, int64_t n0, int64_t n1, int64_t n2, const WallBuffers& b,
                       bool wide, uint32_t not_staged, uint32_t* out_pairs, int32_t* out_coords, const int perm[3], int key_bits = 0);

// kernels_wallsort.hip -- the records grouped by pair (stable radix sort by lo << 32 | hi, then a gather)
uint64_t wall_sort_temp_bytes(uint64_t n);
hipError_t launch_wall_group_by_pair(hipStream_t s, const uint32_t* pairs, const int32_t* coords, uint64_t n, uint64_t* keys0,
                                     uint64_t* keys1, uint32_t* index0, uint32_t* index1, void* temp, uint64_t temp_bytes,
                                     int label_bits, uint32_t* pairs_out, int32_t* coords_out);
// ... from sort keys and linear voxel indices the fetch has written itself (launch_wall_fetch with key_bits = label_bits, into
// keys0 / index0): no key pass, and the last pass makes the coordinates out of the index instead of gathering them.
// mdims: the volume in memory order; perm[k] = array axis of memory axis k
hipError_t launch_wall_group_keyed(hipStream_t s, uint64_t n, uint64_t* keys0, uint64_t* keys1, uint32_t* index0, uint32_t* index1,
                                   void* temp, int label_bits, const int64_t mdims[3], const int perm[3], uint32_t* pairs_out,
                                   int32_t* coords_out);

// kernels_walls.hip (continued): exclusive scan of uint32 counts into uint64 offsets (three small kernels)
uint64_t scan_u32_scratch_bytes(uint64_t n);
void launch_scan_u32_exclusive(hipStream_t s, const uint32_t* counts, uint64_t n, void* scratch, uint64_t* offsets);
uint64_t* scan_u32_total(void* scratch, uint64_t n);        // where that scan leaves the sum of all counts (device)

// kernels_wallmedian.hip -- the median voxel of every wall, from the records grouped by pair (one wave per wall)
uint64_t wall_median_scratch_bytes(uint64_t n);
void launch_wall_starts(hipStream_t s, const uint32_t* pairs, uint64_t n, void* scratch, uint32_t* starts, uint64_t** nwalls_dev);
void launch_wall_medians(hipStream_t s, const uint32_t* pairs, const int32_t* coords, const uint32_t* starts, uint32_t nwalls, uint64_t n,
                         int max_iter, uint32_t* out_pairs, uint32_t* out_sizes, int32_t* out_medians, uint32_t* status);

// kernels_pairsort.hip -- the unique pairs sorted by (lo, hi): counting sort over the label rows + rank inside a bucket
uint64_t pairs_sort_scratch_bytes(uint64_t n, uint32_t max_label);
hipError_t launch_pairs_sort(hipStream_t s, const uint64_t* keys, const uint64_t* faces, uint64_t n, uint32_t max_label, void* scratch,
                             uint64_t* keys_out, uint64_t* faces_out, const void* vol, int itemsize, int64_t corner);

// kernels_census.hip -- the label ids a volume holds (device np.unique) and the volume rewritten in their ranks
uint64_t census_words(uint32_t max_label);
uint64_t census_bytes(uint32_t max_label);              // { bits, ids below } per 32 ids
uint64_t census_scratch_bytes(uint32_t max_label);
void launch_census_mark(hipStream_t s, const void* vol, int itemsize, uint64_t n, int64_t row_len, void* census, void* scratch,
                        uint32_t max_label);
uint64_t census_list_head_bytes();
uint32_t census_list_parts();
void launch_census_from_list(hipStream_t s, const void* list, uint32_t cap, uint32_t most, void* census, void* scratch, uint32_t max_label);
uint64_t census_list_capacity(uint64_t n);
bool launch_census_list(hipStream_t s, const void* vol, int itemsize, uint64_t n, int64_t row_len, void* list, uint32_t cap);
void launch_census_from_ids(hipStream_t s, const uint32_t* ids_dev, uint64_t n, void* census, void* scratch, uint32_t max_label);
void launch_census_scan(hipStream_t s, void* census, uint32_t max_label, void* scratch, uint32_t* ids_out, uint32_t** total_dev);
void launch_census_rank(hipStream_t s, const void* vol, void* out, int itemsize, uint64_t n, const void* census, uint32_t max_label,
                        uint32_t* status);

// kernels_basic.hip (continued)
void launch_synth(hipStream_t s, void* out, int itemsize, const int64_t dims[3], int64_t a_begin,
                  int64_t a_count, const int32_t* seeds_dev, const int32_t grid[3],
                  const int64_t* ell_dev);

// kernels_scan.hip -- the sweep
// ev_start / ev_stop: optional HIP events that receive the begin of the first and the end of the last sweep launch
void launch_scan(hipStream_t s, const SweepArgs& a, int itemsize, uint32_t feature_mask, hipEvent_t ev_start = nullptr,
                 hipEvent_t ev_stop = nullptr);
uint64_t sweep_grid_size(const SweepArgs& a, int itemsize, bool adjacency);   // workgroups of the sweep kernels of that feature class
int sweep_default_tile_planes(bool adjacency, int itemsize, int shape);
int sweep_max_tile_planes(bool adjacency, int itemsize, int shape);   // what the kernel for these volumes packs its tile sums for
int sweep_tile_planes_limit();                                        // what TA_OPT_TILE_PLANES accepts (taller than a kernel's cap: clamped)

}  // namespace ta
