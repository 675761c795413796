// ta_api.hip -- the C ABI of include/tissue_scan.h on top of the gfx950 kernels.
#include "../../include/tissue_scan.h"
#include "ta_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define TA_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? TA_ENOMEM : TA_EHIP, "%s: %s (%s:%d)",  \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                   \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    uint64_t bytes = 0;
    int reserve(uint64_t need) {
        if (need <= bytes) return TA_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (hipMalloc(&p, need ? need : 16) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(TA_ENOMEM, "hipMalloc of %llu bytes failed", (unsigned long long)need);
        }
        bytes = need;
        return TA_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// page-locked host memory (grow-only): device-to-host copies into it run at PCIe speed, into a std::vector they are staged
struct PinnedBuf {
    void* p = nullptr;
    uint64_t bytes = 0;
    int reserve(uint64_t need) {
        if (need <= bytes) return TA_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        if (hipHostMalloc(&p, need ? need : 16, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(TA_ENOMEM, "hipHostMalloc of %llu bytes failed", (unsigned long long)need);
        }
        bytes = need;
        return TA_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};

}  // namespace

struct ta_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // [0] step begin, [3] step end (TA_OPT_TIMING = 2 only)
    std::vector<hipEvent_t> ring;                       // 2 events per slot around the sweep kernel of the last `ring.size()/2` extractions
    int timing = 1;                                     // TA_OPT_TIMING
    uint64_t extract_seq = 0, ring_since = 0;           // extractions run; the one the ring's oldest valid slot belongs to

    // resident volume
    const void* vol = nullptr;       // device pointer (owned_vol.p or adopted)
    DevBuf owned_vol;
    int itemsize = 0;
    int64_t mdims[3] = {0, 0, 0};    // buffer dims in memory-axis order
    int perm[3] = {0, 1, 2};         // perm[k] = array axis of memory axis k
    int64_t a_origin = 0;
    int first_owned = 0;

    // sparse label ids: the census of the volume's ids and the copy of the volume in their ranks (what the sweep then reads)
    DevBuf census, census_ids, compact_vol, census_list;      // (census_list: the label list of the one-pass census, ~n / 256 entries)
    uint32_t census_max = 0;            // ids 0 .. census_max have a bit
    int64_t census_n = -1;              // ids present, -1 = no census
    int64_t vol_max = -1;               // largest label of the resident buffer (halo included), -1 = not known
    bool compact = false;               // per-label ROWS are ranks 0 .. census_n - 1; every label VALUE handed out is an id
    bool census_of_volume = false;      // the census on the context was taken from THIS volume (not a caller's id list)
    bool rerank_check = false;          // ta_volume_rerank's "id not in the list" word has not been looked at yet
    std::vector<uint32_t> h_ids;        // rank -> id (host copy, compact mode)

    // accumulators
    DevBuf own_sums, own_boxes;
    uint64_t* sums = nullptr;
    int32_t* boxes = nullptr;
    bool bound = false;
    uint32_t bound_max_label = 0;
    uint32_t max_label = 0;

    // adjacency
    DevBuf pkeys, pfaces, out_keys, out_faces, small;   // small: flags[NFLAGS] | cursor | maxlabel
    DevBuf hot_rows;                                    // [workgroups][16] private rows of the hot label
    DevBuf sort_buf;                                    // scratch of ta_adjacency_get's device sort (kept between calls)
    DevBuf wall_counts;                                 // wall voxels: per-chunk record counts, then offsets
    DevBuf wall_stage;                                  // wall voxels: the records the count pass staged (kept until the volume changes)
    int64_t wall_records = -1;                          // result of the last ta_wall_voxels_count, -1 = none
    uint32_t wall_region = 0, wall_not_staged = 0;      // records per staging region of that call (0 = nothing staged); cells left to the second walk
    DevBuf wall_medians;                                // ta_wall_medians: pairs u32[E][2] | sizes u32[E] | medians i32[E][3]
    int64_t wall_median_count = -1;                     // E of the last ta_wall_medians, -1 = none
    bool wall_wide = false;                             // that call met a label >= 2^31
    uint32_t wall_label_or = 0;                         // OR of all labels of the volume (that call): the bits a label takes
    double wall_ms = 0.0;
    int pair_log2 = 0;                                  // current table log2 capacity
    int opt_pair_log2 = 0;
    bool table_clean = false;
    uint32_t* h_small = nullptr;                        // pinned, device-mapped mirror of `small`
    uint32_t* h_small_dev = nullptr;                    // its device address: the last kernel of a step writes it

    // options / state
    int impl = 0;
    int tile_planes = 0;
    // the tile shape of the sweep of a uint32 volume with adjacency (kernels_scan.hip): both give the same results; which one
    // is faster depends on the tissue (background around it: the wide one; cells everywhere: the narrow one), so the first four
    // sweeps of a volume take turns (wide, narrow, wide, narrow) between two events each, and the faster shape keeps the volume
    int opt_shape = -1;                                 // TA_OPT_SWEEP_SHAPE: -1 = measure, 0 / 1 = as told
    int shape_pick = -1;                                // choice for this volume, -1 = not yet
    double shape_density = -1.0;                        // label changes per voxel in the sampled planes (what decided it), -1 = not measured
    int last_shape = 0;                                 // TA_OPT_SWEEP_SHAPE_USED: the shape of the last sweep
    int tune_launched = 0;                              // measuring sweeps launched (0 .. 4)
    hipEvent_t tune_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool tune_done[4] = {false, false, false, false};
    float tune_ms[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t volume_slack = 0;                           // TA_OPT_VOLUME_SLACK: bytes readable behind an adopted volume
    int auto_tile_shift = 0;                            // automatic tile height halved this many times (table spills seen)
    uint64_t last_grid = 0;                             // workgroups of the last sweep
    uint32_t feature_mask = 0;
    bool extracted = false, checked = false;
    bool exchanged = false;                             // adjacency rebuilt by ta_adjacency_merge_blocks
    bool shared_packed = false;                         // ... from ta_adjacency_pack_shared blocks: the list is PARTIAL
    bool reduced = false;                               // the bound accumulators hold other ranks' contributions too
    int64_t npairs = 0;
    PinnedBuf h_pairs;                                  // sorted host copy for ta_adjacency_get: keys u64[n], then faces u64[n][3]
    bool host_pairs_ready = false;
};

namespace {

constexpr int SMALL_WORDS = ta::SMALL_WORDS_DEV;   // flags, cursor, max label, parked hot-row pointer (2 words), tile queues (8)

uint32_t* flags_dev(ta_ctx* c) { return (uint32_t*)c->small.p; }
uint32_t* cursor_dev(ta_ctx* c) { return (uint32_t*)c->small.p + ta::NFLAGS; }
uint32_t* maxlab_dev(ta_ctx* c) { return (uint32_t*)c->small.p + ta::NFLAGS + 1; }

// the volume the sweep reads: the rank copy in compact mode
const void* sweep_vol(const ta_ctx* c) { return c->compact ? c->compact_vol.p : c->vol; }

void drop_census(ta_ctx* c) {          // (whenever the voxels change)
    c->census_n = -1;
    c->census_of_volume = false;
    c->rerank_check = false;
    c->vol_max = -1;
    c->shape_pick = -1;
    c->shape_density = -1.0;
    c->tune_launched = 0;
    for (bool& d : c->tune_done) d = false;
    if (c->compact) { c->compact = false; c->extracted = c->checked = false; }
}

int use_device(ta_ctx* c) {
    TA_HIP(hipSetDevice(c->device));
    return TA_OK;
}

int ensure_pair_table(ta_ctx* c, int log2cap) {
    if (c->pair_log2 == log2cap && c->pkeys.p) return TA_OK;
    const uint64_t cap = 1ull << log2cap;
    int rc;
    if ((rc = c->pkeys.reserve(cap * 8)) != TA_OK) return rc;
    if ((rc = c->pfaces.reserve(cap * 24)) != TA_OK) return rc;
    if ((rc = c->out_keys.reserve(cap * 8)) != TA_OK) return rc;
    if ((rc = c->out_faces.reserve(cap * 24)) != TA_OK) return rc;
    c->pair_log2 = log2cap;
    c->table_clean = false;
    return TA_OK;
}

ta::PairTable pair_table(ta_ctx* c) {
    ta::PairTable pt;
    pt.keys = (uint64_t*)c->pkeys.p;
    pt.faces = (uint64_t*)c->pfaces.p;
    pt.mask = (uint32_t)((1ull << c->pair_log2) - 1);
    return pt;
}

int auto_pair_log2(uint32_t max_label) {
    uint64_t want = 16ull * ((uint64_t)max_label + 1);
    int l = 16;
    while ((1ull << l) < want && l < 28) ++l;
    return l;
}

// Label changes per voxel along the fast axis, from a SAMPLE of the owned planes (eight planes spread over the slab, one small
// kernel each, one 64-byte read-back): what predicts which tile shape of the uint32 adjacency sweep is faster.  One stream
// synchronisation, once per resident volume.  < 0: could not be measured.
double sampled_event_density(ta_ctx* c) {
    const int64_t owned = c->mdims[0] - c->first_owned;
    if (owned <= 0 || c->mdims[1] <= 0 || c->mdims[2] <= 0 || !c->vol) return -1.0;
    const int nsample = (int)std::min<int64_t>(8, owned);
    DevBuf d;
    if (d.reserve((uint64_t)nsample * sizeof(uint64_t)) != TA_OK) return -1.0;
    const size_t plane_bytes = (size_t)c->mdims[1] * c->mdims[2] * c->itemsize;
    for (int k = 0; k < nsample; ++k) {
        const int64_t p = c->first_owned + ((2 * k + 1) * owned) / (2 * nsample);
        ta::launch_plane_events(c->stream, (const char*)c->vol + (size_t)p * plane_bytes, c->itemsize, 1, c->mdims[1], c->mdims[2], (uint64_t*)d.p + k);
    }
    uint64_t ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(ev, d.p, (size_t)nsample * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    d.release();
    if (e != hipSuccess) { (void)hipGetLastError(); return -1.0; }
    uint64_t tot = 0;
    for (int k = 0; k < nsample; ++k) tot += ev[k];
    return (double)tot / ((double)nsample * (double)c->mdims[1] * (double)c->mdims[2]);
}

// Above this many label changes per voxel the narrow tiles win (five waves a SIMD where every plane step is full of records),
// below it the wide ones (fewer plane steps where most steps are background).  Same box, same call, round 5: C4 (0.023 changes a
// voxel) wide 0.98 vs narrow 1.01 ms; the same cells without the ellipsoid (0.054) 1.38 vs 1.28 ms.
constexpr double SHAPE_DENSITY_NARROW = 0.032;
constexpr double WIDE_SHORTER_TILES_DENSITY = 0.02;       // (see run_extract: the default tile height of the wide tiles)

// The sweep shape of this launch; *tune = the measuring slot (0 .. 3) whose events bracket it, or -1.
int sweep_shape(ta_ctx* c, bool adjacency, int* tune) {
    *tune = -1;
    if (c->itemsize != 4 || !adjacency) return 0;
    if (c->opt_shape >= 0) return c->opt_shape;
    // the wide tiles want whole 512-column tiles: the partial ones run a kernel with three waves per SIMD (1000^3: 1.24
    // against 1.05 ms), and a volume narrower than a tile has nothing else
    if (c->mdims[2] % 512 != 0) return 0;
    if (c->shape_pick >= 0) return c->shape_pick;
    if (c->opt_shape == -1) {
        // decided BEFORE the first sweep, from the density of label changes in a sample of planes (a caller that sweeps a volume
        // once -- SpatialImageAnalysis(image) -- gets the faster shape on that sweep)
        const double density = sampled_event_density(c);
        c->shape_density = density;
        c->shape_pick = (density >= 0.0 && density > SHAPE_DENSITY_NARROW) ? 0 : 1;
        return c->shape_pick;
    }
    // TA_OPT_SWEEP_SHAPE = -2: the first four sweeps of the volume take turns (wide, narrow, wide, narrow), each between two
    // events of its own, and the faster shape keeps the volume
    bool all = c->tune_launched == 4;
    for (int k = 0; k < c->tune_launched; ++k) {
        if (!c->tune_done[k]) {
            if (hipEventQuery(c->tune_ev[2 * k + 1]) == hipSuccess &&
                hipEventElapsedTime(&c->tune_ms[k], c->tune_ev[2 * k], c->tune_ev[2 * k + 1]) == hipSuccess) c->tune_done[k] = true;
            else (void)hipGetLastError();             // (not ready: asked again by the next sweep)
        }
        all = all && c->tune_done[k];
    }
    if (all) {
        c->shape_pick = std::min(c->tune_ms[0], c->tune_ms[2]) <= std::min(c->tune_ms[1], c->tune_ms[3]) ? 1 : 0;
        return c->shape_pick;
    }
    if (c->tune_launched < 4 && c->tune_ev[7]) {
        *tune = c->tune_launched;                     // (counted as launched by run_extract once BOTH its events are on the stream)
        return (*tune & 1) ^ 1;                       // wide, narrow, wide, narrow
    }
    return 1;                                         // (measured sweeps still in flight)
}

// One full extraction pass on the stream (no host sync).
int run_extract(ta_ctx* c) {
    const uint64_t nlabels = (uint64_t)c->max_label + 1;
    ta::SweepArgs a;
    a.vol = sweep_vol(c);
    a.n0 = c->mdims[0]; a.n1 = c->mdims[1]; a.n2 = c->mdims[2];
    a.a_origin = c->a_origin;
    a.first_owned = c->first_owned;
    int tune = -1;
    a.shape = sweep_shape(c, c->feature_mask & TA_F_ADJACENCY, &tune);
    c->last_shape = a.shape;
    a.tile_planes = c->tile_planes > 0 ? c->tile_planes : ta::sweep_default_tile_planes(c->feature_mask & TA_F_ADJACENCY, c->itemsize, a.shape);
    if (c->tile_planes <= 0) {
        // the wide tiles of a volume whose sampled planes change label often hold more labels and pairs a plane: a little shorter
        // (C4, 0.023 changes a voxel: 28 planes 0.948 against 0.954 ms at 32; C5, 0.014: 32 planes 6.546 against 6.562 at 28 --
        // profiles/r05_tile_planes.txt; without a measured density -- a forced shape -- the default stays)
        if (a.shape == 1 && c->shape_density >= WIDE_SHORTER_TILES_DENSITY && a.tile_planes > 28) a.tile_planes = 28;
        // automatic: small volumes get shorter tiles until the launch has >= 2048 workgroups (8 per CU)
        while (a.tile_planes > 8 && ta::sweep_grid_size(a, c->itemsize, c->feature_mask & TA_F_ADJACENCY) < 2048) a.tile_planes /= 2;
        // volumes whose cells are so small that a tile holds more labels than the workgroup tables (the contributions
        // then spill to global atomics, ~100x dearer) get shorter tiles still: see finish_extract
        for (int k = 0; k < c->auto_tile_shift && a.tile_planes > 1; ++k) a.tile_planes /= 2;
    }
    {   // packed LDS moment words: each kernel is built for tiles up to this height
        const int cap = ta::sweep_max_tile_planes(c->feature_mask & TA_F_ADJACENCY, c->itemsize, a.shape);
        if (a.tile_planes > cap) a.tile_planes = cap;
    }
    // 16-byte loads: rows that are 16-byte aligned, or ANY rows of a volume the library uploaded itself (unaligned 16-byte
    // global loads are legal on gfx950 -- 6.2 TB/s from dword-aligned, 4.8 TB/s from odd addresses, measured -- and the
    // strip that straddles the end of the very last row reads into the slack ta_volume_set leaves behind the buffer)
    a.vec_ok = ((((uintptr_t)a.vol & 15) == 0) && ((a.n2 * c->itemsize) % 16 == 0)) || (a.vol == c->owned_vol.p && c->owned_vol.p) ||
               c->compact || c->volume_slack >= 16;       // (an adopted buffer whose owner promises readable bytes behind it: TA_OPT_VOLUME_SLACK)
    a.max_label = c->max_label;
    a.sums = c->sums;
    a.boxes = c->boxes;
    a.pairs = pair_table(c);
    a.flags = flags_dev(c);
    uint64_t* hot_rows = nullptr;
    uint64_t nwg = 0;
    if (c->impl == 0) {           // the sweep keeps a private row per workgroup for the hot label
        nwg = ta::sweep_grid_size(a, c->itemsize, c->feature_mask & TA_F_ADJACENCY);
        int rc0 = c->hot_rows.reserve(nwg * ta::HOTW * 8);
        if (rc0 != TA_OK) return rc0;
        hot_rows = (uint64_t*)c->hot_rows.p;
    }

    const bool adj = c->feature_mask & TA_F_ADJACENCY;
    c->last_grid = ta::sweep_grid_size(a, c->itemsize, c->feature_mask & TA_F_ADJACENCY);
    // A hipEventRecord costs ~4 us of queue time: by default only the sweep kernel is bracketed (TA_OPT_TIMING)
    const size_t nslots = c->ring.size() / 2;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    if (c->timing >= 1 && nslots) {
        ev_a = c->ring[2 * (c->extract_seq % nslots)];
        ev_b = c->ring[2 * (c->extract_seq % nslots) + 1];
        if (c->extract_seq - c->ring_since >= nslots) c->ring_since = c->extract_seq - nslots + 1;
    } else {
        c->ring_since = c->extract_seq + 1;
    }
    ++c->extract_seq;
    if (c->timing >= 2) TA_HIP(hipEventRecord(c->ev[0], c->stream));
    if (adj && !c->table_clean) {
        ta::launch_pairs_clear(c->stream, a.pairs);
        c->table_clean = true;
    }
    ta::launch_init_accumulators(c->stream, c->sums, c->boxes, nlabels, flags_dev(c), cursor_dev(c), hot_rows);
    // the sweep kernel alone (what the roofline is quoted on): the two events ride on the sweep's own launches
    // (begin / end timestamps of the dispatch, no event-record packets on the queue); the naive kernel gets plain records
    const bool own_dims = c->mdims[0] - c->first_owned > 0 && c->mdims[1] > 0 && c->mdims[2] > 0;
    if (tune >= 0) TA_HIP(hipEventRecord(c->tune_ev[2 * tune], c->stream));
    if (c->impl == 1 || !own_dims) {
        if (ev_a) TA_HIP(hipEventRecord(ev_a, c->stream));
        if (c->impl == 1) ta::launch_naive(c->stream, a, c->itemsize, c->feature_mask);
        if (ev_b) TA_HIP(hipEventRecord(ev_b, c->stream));
    } else {
        ta::launch_scan(c->stream, a, c->itemsize, c->feature_mask, ev_a, ev_b);
    }
    if (tune >= 0) {
        TA_HIP(hipEventRecord(c->tune_ev[2 * tune + 1], c->stream));
        c->tune_launched = tune + 1;                  // (a slot counts only with both of its events recorded: an early return above leaves it to be measured again)
    }
    // Without adjacency the LAST kernel of the step (the hot-row fold) mirrors the flag words into host-mapped memory
    // itself: no device-to-host copy (a blit kernel and a queue barrier) at the end of the step.  With adjacency the pair
    // count is final only when the collect kernel has ended; letting its last block publish it was measured and costs
    // more (every block then waits for its own stores before it can count itself done: 29 -> 44 us) than the copy.
    uint32_t* mirror = c->h_small_dev;
    bool published = false;
    if (hot_rows && !adj) {
        ta::launch_hot_reduce(c->stream, a, c->itemsize, hot_rows, (uint32_t)nwg, mirror, SMALL_WORDS);
        published = mirror != nullptr;
    }
    if (adj && hot_rows)             // the fold of the hot-label rows rides along with the collect: one launch
        ta::launch_pairs_collect_hot(c->stream, a.pairs, (uint64_t*)c->out_keys.p, (uint64_t*)c->out_faces.p, cursor_dev(c),
                                     a, c->itemsize, hot_rows, (uint32_t)nwg);
    else if (adj)
        ta::launch_pairs_collect(c->stream, a.pairs, (uint64_t*)c->out_keys.p, (uint64_t*)c->out_faces.p, cursor_dev(c));
    if (c->timing >= 2) TA_HIP(hipEventRecord(c->ev[3], c->stream));
    if (!published)
        TA_HIP(hipMemcpyAsync(c->h_small, c->small.p, SMALL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost,
                              c->stream));
    TA_HIP(hipGetLastError());
    return TA_OK;
}

// The word ta_volume_rerank writes shares its place with the max-label passes: whoever is about to reuse it looks at it first.
int settle_rerank(ta_ctx* c) {
    if (!c->rerank_check) return TA_OK;
    uint32_t status = 0;
    TA_HIP(hipMemcpyAsync(&status, maxlab_dev(c), sizeof(status), hipMemcpyDeviceToHost, c->stream));
    TA_HIP(hipStreamSynchronize(c->stream));
    c->rerank_check = false;
    if (status) { c->extracted = false; return fail(TA_ERANGE, "the refreshed volume holds a label id that is not in the list the context was compacted with"); }
    return TA_OK;
}

// Drain the stream and validate the flags of the last pass; grows the adjacency table and
// re-runs when it overflowed.
int finish_extract(ta_ctx* c) {
    if (!c->extracted) return fail(TA_EINVAL, "no extraction has been run on this context");
    if (c->checked) return TA_OK;
    if (c->exchanged) {       // the list came from other ranks too: a re-run is the host's call
        TA_HIP(hipStreamSynchronize(c->stream));
        if (c->rerank_check) {
            c->rerank_check = false;
            if (c->h_small[ta::NFLAGS + 1]) { c->extracted = false; return fail(TA_ERANGE, "the refreshed volume holds a label id that is not in the list the context was compacted with"); }
        }
        if (c->h_small[ta::FLAG_RANGE])
            return fail(TA_ERANGE, "a rank saw a label above max_label=%u", c->max_label);
        if (c->h_small[ta::FLAG_EXCHANGE_OVERFLOW])
            return fail(TA_ECAPACITY, "an exchange block was too small for a rank's pair list");
        if (c->h_small[ta::FLAG_PAIR_OVERFLOW])
            return fail(TA_ECAPACITY, "adjacency table overflow on some rank (2^%d slots here)", c->pair_log2);
        c->npairs = (int64_t)c->h_small[ta::NFLAGS];
        c->checked = true;
        return TA_OK;
    }
    for (int attempt = 0; attempt < 8; ++attempt) {
        TA_HIP(hipStreamSynchronize(c->stream));
        if (c->rerank_check) {          // (the word ta_volume_rerank left behind came back with this extraction's flags)
            c->rerank_check = false;
            if (c->h_small[ta::NFLAGS + 1]) {
                c->extracted = false;
                return fail(TA_ERANGE, "the refreshed volume holds a label id that is not in the list the context was compacted with");
            }
        }
        if (c->h_small[ta::FLAG_RANGE])
            return fail(TA_ERANGE, "the volume holds a label above max_label=%u", c->max_label);
        if (!c->h_small[ta::FLAG_PAIR_OVERFLOW]) {
            c->npairs = (c->feature_mask & TA_F_ADJACENCY) ? (int64_t)c->h_small[ta::NFLAGS] : 0;
            c->checked = true;
            // more than a handful of table spills per workgroup: the next sweeps of this context use shorter tiles
            // (results do not depend on the tile height; only the automatic height adapts, an explicit one is kept)
            const uint64_t spills = (uint64_t)c->h_small[ta::FLAG_LDS_LABEL_SPILL] + c->h_small[ta::FLAG_LDS_PAIR_SPILL];
            if (c->impl == 0 && c->tile_planes <= 0 && c->auto_tile_shift < 4 && spills > 8 * c->last_grid) ++c->auto_tile_shift;
            return TA_OK;
        }
        if (c->pair_log2 >= 30) break;
        int rc = ensure_pair_table(c, c->pair_log2 + 2);
        if (rc != TA_OK) return rc;
        if (c->reduced)       // a local re-run would replace the reduced (global) rows by this rank's: the host redoes the step
            return fail(TA_ECAPACITY, "adjacency table overflow (grown to 2^%d slots): repeat the extraction on every rank", c->pair_log2);
        if ((rc = run_extract(c)) != TA_OK) return rc;
    }
    return fail(TA_ECAPACITY, "adjacency table overflow at 2^%d slots", c->pair_log2);
}

}  // namespace

extern "C" {

TA_API int ta_version(void) { return TA_ABI_VERSION; }

TA_API const char* ta_last_error(void) { return g_err.c_str(); }

TA_API int ta_device_count(int* count) {
    if (!count) return fail(TA_EINVAL, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return TA_OK;
}

TA_API int ta_ctx_destroy(ta_ctx* c);

TA_API int ta_ctx_create(int device_id, ta_ctx** out) {
    if (!out) return fail(TA_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(TA_ENODEVICE, "no HIP device visible (libtissue_scan needs an MI355X / gfx950 GPU)");
    }
    if (device_id < 0 || device_id >= n) return fail(TA_EINVAL, "device_id %d out of range [0,%d)", device_id, n);
    ta_ctx* c = new (std::nothrow) ta_ctx();
    if (!c) return fail(TA_ENOMEM, "out of host memory");
    c->device = device_id;
    // every failure below goes through ta_ctx_destroy: it releases whatever was created so far
    int rc = TA_OK;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) c->own_stream = true;
    for (auto& ev : c->ev) if (e == hipSuccess) e = hipEventCreate(&ev);
    try { c->ring.assign(2, nullptr); } catch (...) { rc = fail(TA_ENOMEM, "out of host memory"); }
    for (auto& ev : c->ring) if (e == hipSuccess && rc == TA_OK) e = hipEventCreate(&ev);
    for (auto& ev : c->tune_ev) if (e == hipSuccess && rc == TA_OK) e = hipEventCreate(&ev);
    if (e == hipSuccess) rc = c->small.reserve(SMALL_WORDS * sizeof(uint32_t));
    if (e == hipSuccess && rc == TA_OK) e = hipHostMalloc((void**)&c->h_small, SMALL_WORDS * sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess && rc == TA_OK) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, c->h_small, 0) == hipSuccess) c->h_small_dev = (uint32_t*)dp;
        else (void)hipGetLastError();          // no mapping: the steps fall back to the device-to-host copy
    }
    if (e != hipSuccess || rc != TA_OK) {
        if (e != hipSuccess) rc = fail(e == hipErrorOutOfMemory ? TA_ENOMEM : TA_EHIP, "ta_ctx_create: %s", hipGetErrorString(e));
        const std::string keep = g_err;
        (void)ta_ctx_destroy(c);
        g_err = keep;
        return rc;
    }
    memset(c->h_small, 0, SMALL_WORDS * sizeof(uint32_t));
    *out = c;
    return TA_OK;
}

TA_API int ta_ctx_destroy(ta_ctx* c) {
    if (!c) return TA_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->owned_vol.release(); c->own_sums.release(); c->own_boxes.release();
    c->pkeys.release(); c->pfaces.release(); c->out_keys.release(); c->out_faces.release();
    c->small.release();
    c->hot_rows.release(); c->sort_buf.release(); c->h_pairs.release();
    c->wall_counts.release();
    c->wall_stage.release();
    c->wall_medians.release();
    c->census.release(); c->census_ids.release(); c->compact_vol.release(); c->census_list.release();
    if (c->h_small) (void)hipHostFree(c->h_small);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ring) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->tune_ev) if (e) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TA_OK;
}

TA_API int ta_ctx_set_stream(ta_ctx* c, void* hip_stream) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if (c->stream) TA_HIP(hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = nullptr;
    if (hip_stream == TA_STREAM_LEGACY_DEFAULT) {
        c->stream = hipStreamLegacy;              // the null stream: ordered with every blocking stream of the device
        c->own_stream = false;
    } else if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
        c->own_stream = false;
    } else {
        TA_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return TA_OK;
}

TA_API int ta_ctx_set_option(ta_ctx* c, int key, int64_t value) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    switch (key) {
        case TA_OPT_IMPL:
            if (value < 0 || value > 1) return fail(TA_EINVAL, "TA_OPT_IMPL must be 0 (sweep) or 1 (per-voxel atomics)");
            c->impl = (int)value; return TA_OK;
        case TA_OPT_TILE_PLANES:
            if (value < 0 || value > ta::sweep_tile_planes_limit()) return fail(TA_EINVAL, "TA_OPT_TILE_PLANES must be in [0,%d]", ta::sweep_tile_planes_limit());
            c->tile_planes = (int)value; return TA_OK;
        case TA_OPT_PAIR_SLOTS:
            if (value != 0 && (value < 4 || value > 30)) return fail(TA_EINVAL, "TA_OPT_PAIR_SLOTS must be 0 or in [4,30]");
            c->opt_pair_log2 = (int)value; return TA_OK;
        case TA_OPT_SWEEP_SHAPE:
            if (value < -2 || value > 1) return fail(TA_EINVAL, "TA_OPT_SWEEP_SHAPE is -1 (by label-change density), -2 (by four timed sweeps), 0 or 1");
            c->opt_shape = (int)value;
            c->auto_tile_shift = 0;
            c->shape_pick = -1;                    // (decided again, by the new rule, at the next sweep)
            c->shape_density = -1.0;
            c->tune_launched = 0;
            for (bool& d : c->tune_done) d = false;
            return TA_OK;
        case TA_OPT_VOLUME_SLACK:
            if (value < 0) return fail(TA_EINVAL, "TA_OPT_VOLUME_SLACK must be >= 0");
            c->volume_slack = value; return TA_OK;
        case TA_OPT_TIMING:
            if (value < 0 || value > 2) return fail(TA_EINVAL, "TA_OPT_TIMING must be 0, 1 or 2");
            c->timing = (int)value; return TA_OK;
        case TA_OPT_TIMING_RING: {
            if (value < 1 || value > 4096) return fail(TA_EINVAL, "TA_OPT_TIMING_RING must be in [1,4096]");
            int rc = use_device(c);
            if (rc != TA_OK) return rc;
            if (c->stream) TA_HIP(hipStreamSynchronize(c->stream));
            const size_t want = 2 * (size_t)value;
            while (c->ring.size() > want) { (void)hipEventDestroy(c->ring.back()); c->ring.pop_back(); }
            while (c->ring.size() < want) {
                hipEvent_t ev = nullptr;
                TA_HIP(hipEventCreate(&ev));
                try { c->ring.push_back(ev); } catch (...) { (void)hipEventDestroy(ev); return fail(TA_ENOMEM, "out of host memory"); }
            }
            c->ring_since = c->extract_seq;          // the kept durations start with the next extraction
            return TA_OK;
        }
        default:
            return fail(TA_EINVAL, "unknown option key %d", key);
    }
}

TA_API int ta_ctx_get_option(ta_ctx* c, int key, int64_t* value) {
    if (!c || !value) return fail(TA_EINVAL, "NULL argument");
    switch (key) {
        case TA_OPT_IMPL: *value = c->impl; return TA_OK;
        case TA_OPT_VOLUME_SLACK: *value = c->volume_slack; return TA_OK;
        case TA_OPT_SWEEP_SHAPE: *value = c->opt_shape; return TA_OK;
        case TA_OPT_SWEEP_SHAPE_USED: *value = c->last_shape; return TA_OK;
        case TA_OPT_TIMING: *value = c->timing; return TA_OK;
        case TA_OPT_TIMING_RING: *value = (int64_t)(c->ring.size() / 2); return TA_OK;
        case TA_OPT_TILE_PLANES: {
            const int shape = c->opt_shape >= 0 ? c->opt_shape : (c->shape_pick >= 0 ? c->shape_pick : 1);
            int planes = c->tile_planes > 0 ? c->tile_planes : ta::sweep_default_tile_planes(c->feature_mask & TA_F_ADJACENCY, c->itemsize, shape);
            if (c->tile_planes <= 0 && c->itemsize == 4 && (c->feature_mask & TA_F_ADJACENCY) && shape == 1 &&
                c->shape_density >= WIDE_SHORTER_TILES_DENSITY && planes > 28) planes = 28;      // (the rule of run_extract)
            *value = planes; return TA_OK;
        }
        case TA_OPT_PAIR_SLOTS: *value = c->pkeys.p ? c->pair_log2 : c->opt_pair_log2; return TA_OK;
        default: return fail(TA_EINVAL, "unknown option key %d", key);
    }
}

TA_API int ta_ctx_synchronize(ta_ctx* c) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    TA_HIP(hipStreamSynchronize(c->stream));
    return TA_OK;
}

TA_API int ta_volume_set(ta_ctx* c, const void* host_ptr, int itemsize, const int64_t dims[3],
                  const int64_t strides_bytes[3]) {
    if (!c || !host_ptr || !dims) return fail(TA_EINVAL, "NULL argument");
    if (itemsize != 2 && itemsize != 4) return fail(TA_EINVAL, "itemsize must be 2 (uint16) or 4 (uint32)");
    for (int d = 0; d < 3; ++d)
        if (dims[d] <= 0 || dims[d] > (1ll << 30)) return fail(TA_EINVAL, "dims[%d]=%lld out of range", d, (long long)dims[d]);
    int perm[3] = {0, 1, 2};
    if (strides_bytes) {
        // memory order = axes by decreasing stride (size-1 axes are layout-neutral: keep them first)
        std::stable_sort(perm, perm + 3, [&](int x, int y) {
            const int64_t sx = dims[x] == 1 ? INT64_MAX : strides_bytes[x];
            const int64_t sy = dims[y] == 1 ? INT64_MAX : strides_bytes[y];
            return sx > sy;
        });
        int64_t expect = itemsize;
        for (int k = 2; k >= 0; --k) {
            const int ax = perm[k];
            if (dims[ax] != 1 && strides_bytes[ax] != expect)
                return fail(TA_EINVAL, "strides do not describe a dense permuted layout (axis %d: stride %lld, expected %lld)",
                            ax, (long long)strides_bytes[ax], (long long)expect);
            expect *= dims[ax];
        }
    }
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint64_t bytes = (uint64_t)dims[0] * dims[1] * dims[2] * itemsize;
    TA_HIP(hipStreamSynchronize(c->stream));
    if ((rc = c->owned_vol.reserve(bytes + 64)) != TA_OK) return rc;      // (+ slack: see run_extract, vec_ok)
    TA_HIP(hipMemcpyAsync(c->owned_vol.p, host_ptr, bytes, hipMemcpyHostToDevice, c->stream));
    TA_HIP(hipStreamSynchronize(c->stream));   // the host buffer may be freed after return
    c->vol = c->owned_vol.p;
    drop_census(c);
    c->auto_tile_shift = 0;
    c->wall_records = -1;
    c->wall_median_count = -1;
    c->wall_stage.release();
    c->itemsize = itemsize;
    for (int k = 0; k < 3; ++k) { c->perm[k] = perm[k]; c->mdims[k] = dims[perm[k]]; }
    c->a_origin = 0;
    c->first_owned = 0;
    c->extracted = c->checked = false;
    return TA_OK;
}

TA_API int ta_volume_set_device(ta_ctx* c, const void* dev_ptr, int itemsize, const int64_t buf_dims[3],
                         int64_t a0_origin, int has_low_halo) {
    if (!c || !dev_ptr || !buf_dims) return fail(TA_EINVAL, "NULL argument");
    if (itemsize != 2 && itemsize != 4) return fail(TA_EINVAL, "itemsize must be 2 (uint16) or 4 (uint32)");
    for (int d = 0; d < 3; ++d)
        if (buf_dims[d] <= 0 || buf_dims[d] > (1ll << 30)) return fail(TA_EINVAL, "buf_dims[%d]=%lld out of range", d, (long long)buf_dims[d]);
    if (has_low_halo && buf_dims[0] < 2) return fail(TA_EINVAL, "a slab with a halo needs at least 2 planes");
    if (a0_origin < 0) return fail(TA_EINVAL, "a0_origin must be >= 0");
    if (((uintptr_t)dev_ptr % itemsize) != 0) return fail(TA_EINVAL, "device pointer is not aligned to the label type");
    c->vol = dev_ptr;
    drop_census(c);
    c->volume_slack = 0;
    c->auto_tile_shift = 0;
    c->wall_records = -1;
    c->wall_median_count = -1;
    c->wall_stage.release();
    c->itemsize = itemsize;
    for (int k = 0; k < 3; ++k) { c->perm[k] = k; c->mdims[k] = buf_dims[k]; }
    c->a_origin = a0_origin;
    c->first_owned = has_low_halo ? 1 : 0;
    c->extracted = c->checked = false;
    return TA_OK;
}

TA_API int ta_volume_relabel(ta_ctx* c, const uint32_t* lut, uint32_t lut_len) {
    if (!c || (!lut && lut_len)) return fail(TA_EINVAL, "NULL argument");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    if (c->first_owned) return fail(TA_EINVAL, "cannot relabel a slab that carries a halo plane");
    if (c->compact && lut_len != (uint32_t)c->census_n)
        return fail(TA_EINVAL, "a compacted context relabels through one entry per rank: %u entries for %lld ranks", lut_len, (long long)c->census_n);
    if (c->itemsize == 2)
        for (uint32_t i = 0; i < lut_len; ++i)
            if (lut[i] > 0xFFFFu) return fail(TA_ERANGE, "lut[%u]=%u does not fit the uint16 volume", i, lut[i]);
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if (lut_len == 0) return TA_OK;
    DevBuf d;
    if ((rc = d.reserve((uint64_t)lut_len * 4)) != TA_OK) return rc;
    hipError_t e = hipMemcpyAsync(d.p, lut, (uint64_t)lut_len * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        ta::launch_relabel(c->stream, sweep_vol(c), const_cast<void*>(c->vol), c->itemsize,
                           (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2], (const uint32_t*)d.p, lut_len);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    d.release();
    if (e != hipSuccess) return fail(TA_EHIP, "relabel: %s", hipGetErrorString(e));
    drop_census(c);                 // (the ids changed: a compacted context goes back to dense rows until it is compacted again)
    c->extracted = c->checked = false;
    c->wall_median_count = -1;
    c->wall_records = -1;           // the staged wall records carry the OLD labels: a fetch must ask for a fresh count
    c->wall_region = 0; c->wall_not_staged = 0;
    return TA_OK;
}

TA_API int ta_volume_get(ta_ctx* c, void* host_dst) {
    if (!c || !host_dst) return fail(TA_EINVAL, "NULL argument");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint64_t bytes = (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2] * c->itemsize;
    TA_HIP(hipMemcpyAsync(host_dst, c->vol, bytes, hipMemcpyDeviceToHost, c->stream));
    TA_HIP(hipStreamSynchronize(c->stream));
    return TA_OK;
}

TA_API int ta_volume_map(ta_ctx* c, const void* lut, uint32_t lut_len, const void* fill, int out_itemsize,
                         void* host_dst) {
    if (!c || !fill || !host_dst || (!lut && lut_len)) return fail(TA_EINVAL, "NULL argument");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    if (out_itemsize != 1 && out_itemsize != 2 && out_itemsize != 4 && out_itemsize != 8)
        return fail(TA_EINVAL, "out_itemsize must be 1, 2, 4 or 8");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint64_t n = (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2];
    uint64_t fillw = 0;
    memcpy(&fillw, fill, (size_t)out_itemsize);
    DevBuf dl, dout;
    if ((rc = dl.reserve((uint64_t)lut_len * out_itemsize + 8)) != TA_OK) return rc;
    if ((rc = dout.reserve(n * out_itemsize)) != TA_OK) { dl.release(); return rc; }
    hipError_t e = hipSuccess;
    if (lut_len) e = hipMemcpyAsync(dl.p, lut, (uint64_t)lut_len * out_itemsize, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        ta::launch_map(c->stream, sweep_vol(c), c->itemsize, dout.p, out_itemsize, n, dl.p, lut_len, fillw);     // (compacted: lut[rank])
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host_dst, dout.p, n * out_itemsize, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dl.release(); dout.release();
    if (e != hipSuccess) return fail(TA_EHIP, "map: %s", hipGetErrorString(e));
    return TA_OK;
}

TA_API int ta_volume_first_layer(ta_ctx* c, uint32_t background, int keep_background, void* host_dst) {
    if (!c || !host_dst) return fail(TA_EINVAL, "NULL argument");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    if (c->first_owned) return fail(TA_EINVAL, "the first voxel layer is not available on a slab that carries a halo plane");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint64_t bytes = (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2] * c->itemsize;
    DevBuf dout;
    if ((rc = dout.reserve(bytes)) != TA_OK) return rc;
    ta::launch_first_layer(c->stream, c->vol, c->itemsize, dout.p, c->mdims[0], c->mdims[1], c->mdims[2], background,
                           keep_background ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host_dst, dout.p, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dout.release();
    if (e != hipSuccess) return fail(TA_EHIP, "first voxel layer: %s", hipGetErrorString(e));
    return TA_OK;
}

TA_API int ta_volume_hollow(ta_ctx* c, uint32_t background, int remove_background, int label_bits, void* host_dst) {
    if (!c || !host_dst) return fail(TA_EINVAL, "NULL argument");
    if (label_bits == 0) label_bits = 8 * c->itemsize;
    if (label_bits != 8 && label_bits != 16 && label_bits != 32 && label_bits != 64) return fail(TA_EINVAL, "label_bits must be 0, 8, 16, 32 or 64");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    if (c->first_owned) return fail(TA_EINVAL, "hollowed-out cells are not available on a slab that carries a halo plane");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint64_t bytes = (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2] * c->itemsize;
    DevBuf dout;
    if ((rc = dout.reserve(bytes)) != TA_OK) return rc;
    ta::launch_hollow(c->stream, c->vol, c->itemsize, dout.p, c->mdims[0], c->mdims[1], c->mdims[2], background, remove_background ? 1 : 0, label_bits);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host_dst, dout.p, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dout.release();
    if (e != hipSuccess) return fail(TA_EHIP, "hollowed-out cells: %s", hipGetErrorString(e));
    return TA_OK;
}

TA_API int ta_volume_layer18(ta_ctx* c, uint8_t* host_dst) {
    if (!c || !host_dst) return fail(TA_EINVAL, "NULL argument");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    if (c->first_owned) return fail(TA_EINVAL, "the voxel layers are not available on a slab that carries a halo plane");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint64_t bytes = (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2];
    DevBuf dout;
    if ((rc = dout.reserve(bytes)) != TA_OK) return rc;
    ta::launch_layer18(c->stream, c->vol, c->itemsize, (uint8_t*)dout.p, c->mdims[0], c->mdims[1], c->mdims[2]);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host_dst, dout.p, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dout.release();
    if (e != hipSuccess) return fail(TA_EHIP, "voxel layers: %s", hipGetErrorString(e));
    return TA_OK;
}

namespace {
// layout of ta_ctx::wall_counts: counts u32[cells] | cell_base u32[cells] (each padded to 8 bytes) | offsets u64[cells] |
// block sums u64[scan_blocks] | total u64 + status u32[6] (the 32 bytes the host reads back) | cursors | todo u32[cells] |
// lane counts u8[cells][64]
uint64_t wall_bufs(void* base, const ta::WallPlan& p, ta::WallBuffers& b) {
    const uint64_t counts_bytes = (p.cells * 4 + 7) & ~7ull;
    char* at = (char*)base;
    b.counts = (uint32_t*)at; at += counts_bytes;
    b.cell_base = (uint32_t*)at; at += counts_bytes;
    b.offsets = (uint64_t*)at; at += p.cells * 8;
    b.block_sums = (uint64_t*)at; at += p.scan_blocks * 8;
    b.total = (uint64_t*)at; b.status = (uint32_t*)(b.total + 1); at += 32;
    b.cursors = (uint32_t*)at; at += ta::wall_cursor_bytes();
    b.todo = (uint32_t*)at; at += counts_bytes;
    b.lane_counts = (uint8_t*)at; at += p.cells * 64;
    return (uint64_t)(at - (char*)base);
}

// Room for the records the count pass stages: half a record per voxel (tissue: 0.1 - 0.25), split into regions; a
// volume with more takes the second walk for the cells that did not fit.  No memory for it: nothing is staged.
void wall_stage(ta_ctx* c, const ta::WallPlan& p, ta::WallBuffers& b) {
    const uint64_t nvox = (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2];
    uint64_t records = std::min<uint64_t>(std::max<uint64_t>(nvox / 2, 1u << 16), 1ull << 31);
    // (tests: force the second walk / small regions.  Clamped: region_of_wave * region + got must stay below 2^32 in the kernel)
    if (const char* env = getenv("TA_WALL_STAGE_RECORDS")) records = std::min<uint64_t>(std::strtoull(env, nullptr, 10), 1ull << 31);
    const uint32_t regions = ta::wall_stage_regions();
    b.region = (uint32_t)(records / regions);
    b.stage = nullptr;
    (void)p;
    if (b.region == 0) return;
    const uint64_t need = ta::wall_stage_bytes(b.region, c->itemsize);
    if (c->wall_stage.bytes < need) {
        c->wall_stage.release();
        if (hipMalloc(&c->wall_stage.p, need) != hipSuccess) {
            (void)hipGetLastError();
            c->wall_stage.p = nullptr;
            b.region = 0;
            return;
        }
        c->wall_stage.bytes = need;
    }
    b.stage = c->wall_stage.p;
}
}  // namespace

TA_API int ta_wall_voxels_count(ta_ctx* c, int64_t* nrecords) {
    if (!c || !nrecords) return fail(TA_EINVAL, "NULL argument");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    if (c->first_owned) return fail(TA_EINVAL, "wall voxels are not available on a slab that carries a halo plane");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const ta::WallPlan plan = ta::wall_plan(c->mdims[0], c->mdims[1], c->mdims[2]);
    if (plan.cells >= (1ull << 32)) return fail(TA_EINVAL, "volume too large for the wall voxel pass (%llu row strips)", (unsigned long long)plan.cells);
    ta::WallBuffers wb;
    if ((rc = c->wall_counts.reserve(wall_bufs(nullptr, plan, wb))) != TA_OK) return rc;
    (void)wall_bufs(c->wall_counts.p, plan, wb);
    wall_stage(c, plan, wb);
    c->wall_region = wb.region;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct { uint64_t total; uint32_t not_staged, wide_seen, label_or, unused[3]; } line = {0, 0, 0, 0, {0, 0, 0}};
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms_all = 0.f;
    bool wide = false;
    for (int attempt = 0; attempt < 2 && e == hipSuccess; ++attempt) {
        e = hipEventRecord(e0, c->stream);
        if (e == hipSuccess) {
            // count + stage per (row, strip), scan on the device: the only thing the host needs before the fetch is one line
            ta::launch_wall_count(c->stream, c->vol, c->itemsize, c->mdims[0], c->mdims[1], c->mdims[2], wb, wide);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&line, wb.total, 32, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        float ms = 0.f;
        if (e == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
        ms_all += ms;
        if (!line.wide_seen || wide) break;
        wide = true;                    // a label from 2^31 up: once more with the kernel that takes them
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(TA_EHIP, "wall voxel count: %s", hipGetErrorString(e));
    c->wall_records = (int64_t)line.total;
    c->wall_median_count = -1;
    c->wall_not_staged = line.not_staged;
    c->wall_wide = wide;
    c->wall_label_or = line.label_or;
    c->wall_ms = ms_all;
    if (getenv("TA_WALL_VERBOSE"))
        fprintf(stderr, "[tissue_scan] wall voxels: %llu records in %llu cells of 256 voxels, %u cells not staged (regions of %u records), wide=%d, %.3f ms\n",
                (unsigned long long)line.total, (unsigned long long)plan.cells, line.not_staged, wb.region, (int)wide, ms_all);
    *nrecords = c->wall_records;
    return TA_OK;
}

namespace {
// The records of the last ta_wall_voxels_count on the DEVICE, in memory order or grouped by pair: `buf` owns them, *pairs_dev /
// *coords_dev point into it; the launches are bracketed by e0 / e1 when given.  Only enqueues work (and allocates).
int wall_records_device(ta_ctx* c, bool by_pair, DevBuf& buf, uint32_t** pairs_dev, int32_t** coords_dev, hipEvent_t e0, hipEvent_t e1) {
    const uint64_t n = (uint64_t)c->wall_records;
    int rc;
    const ta::WallPlan plan = ta::wall_plan(c->mdims[0], c->mdims[1], c->mdims[2]);
    ta::WallBuffers wb;
    (void)wall_bufs(c->wall_counts.p, plan, wb);
    wb.region = c->wall_region;
    wb.stage = c->wall_region ? c->wall_stage.p : nullptr;
    // one allocation: records in memory order | (grouped fetch) the same again grouped, sort keys / indices x 2, sort temp.
    // A volume of fewer than 2^32 voxels is grouped from KEYS: the fetch writes sort keys and linear voxel indices straight into
    // the sort's buffers (no records in memory order, no key pass, no gather of coordinates behind the last pass)
    const uint64_t nvox = (uint64_t)c->mdims[0] * (uint64_t)c->mdims[1] * (uint64_t)c->mdims[2];
    // (tests and same-call comparisons: TA_WALL_KEYED=0 sorts the records of the plain fetch, as volumes of 2^32 voxels and more do)
    const char* env_keyed = getenv("TA_WALL_KEYED");
    const bool keyed = by_pair && nvox < (1ull << 32) && !(env_keyed && env_keyed[0] == '0');
    const uint64_t temp_bytes = by_pair ? ta::wall_sort_temp_bytes(n) : 0;
    const uint64_t rec = n * 8, co = (n * 12 + 15) & ~15ull, ix = (n * 4 + 15) & ~15ull;
    if ((rc = buf.reserve(by_pair ? (keyed ? 0 : rec + co) + rec + co + 2 * rec + 2 * ix + temp_bytes + 64 : rec + co)) != TA_OK) return rc;
    char* p = (char*)buf.p;
    int label_bits = 1;                                             // bits a label of this volume takes
    while (label_bits < 32 && (c->wall_label_or >> label_bits) != 0u) ++label_bits;
    uint32_t* dpa = nullptr; int32_t* dco = nullptr;
    if (!keyed) { dpa = (uint32_t*)p; p += rec; dco = (int32_t*)p; p += co; }
    uint32_t* gpa = dpa; int32_t* gco = dco;
    uint64_t *k0 = nullptr, *k1 = nullptr; uint32_t *i0 = nullptr, *i1 = nullptr;
    if (by_pair) {
        gpa = (uint32_t*)p; p += rec;
        gco = (int32_t*)p; p += co;
        k0 = (uint64_t*)p; p += rec;
        k1 = (uint64_t*)p; p += rec;
        i0 = (uint32_t*)p; p += ix;
        i1 = (uint32_t*)p; p += ix;
    }
    hipError_t e = e0 ? hipEventRecord(e0, c->stream) : hipSuccess;
    if (e == hipSuccess) {
        // records leave the kernels as (lo, hi) / coordinates in ARRAY-axis order -- or as keys / linear indices for the sort
        ta::launch_wall_fetch(c->stream, c->vol, c->itemsize, c->mdims[0], c->mdims[1], c->mdims[2], wb, c->wall_wide,
                              c->wall_not_staged, keyed ? (uint32_t*)k0 : dpa, keyed ? (int32_t*)i0 : dco, c->perm, keyed ? label_bits : 0);
        e = hipGetLastError();
    }
    if (e == hipSuccess && keyed)
        e = ta::launch_wall_group_keyed(c->stream, n, k0, k1, i0, i1, p, label_bits, c->mdims, c->perm, gpa, gco);
    else if (e == hipSuccess && by_pair)
        e = ta::launch_wall_group_by_pair(c->stream, dpa, dco, n, k0, k1, i0, i1, p, temp_bytes, label_bits, gpa, gco);
    if (e == hipSuccess && e1) e = hipEventRecord(e1, c->stream);
    if (e != hipSuccess) return fail(TA_EHIP, "wall voxels: %s", hipGetErrorString(e));
    *pairs_dev = gpa; *coords_dev = gco;
    return TA_OK;
}

int wall_voxels_fetch(ta_ctx* c, uint32_t* pairs, int32_t* coords, double* ms_out, bool by_pair) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if (c->wall_records < 0) return fail(TA_EINVAL, "call ta_wall_voxels_count first");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint64_t n = (uint64_t)c->wall_records;
    if (ms_out) *ms_out = c->wall_ms;
    if (n == 0) return TA_OK;
    if (!pairs || !coords) return fail(TA_EINVAL, "NULL output array");
    if (by_pair && n >= (1ull << 32)) return fail(TA_EINVAL, "too many records (%llu) for the grouped fetch", (unsigned long long)n);
    DevBuf buf;
    uint32_t* gpa = nullptr; int32_t* gco = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess && (rc = wall_records_device(c, by_pair, buf, &gpa, &gco, e0, e1)) != TA_OK) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        buf.release();
        return rc;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(pairs, gpa, n * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(coords, gco, n * 12, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    float ms = 0.f;
    if (e == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    buf.release();
    if (e != hipSuccess) return fail(TA_EHIP, "wall voxels: %s", hipGetErrorString(e));
    if (ms_out) *ms_out = c->wall_ms + ms;
    return TA_OK;
}
}  // namespace

TA_API int ta_wall_medians(ta_ctx* c, int max_iter, int64_t* nwalls, double* ms_out) {
    if (!c || !nwalls) return fail(TA_EINVAL, "NULL argument");
    if (c->wall_records < 0) return fail(TA_EINVAL, "call ta_wall_voxels_count first");
    if (max_iter < 1) return fail(TA_EINVAL, "max_iter must be positive");
    if (c->perm[0] != 0 || c->perm[1] != 1 || c->perm[2] != 2)
        return fail(TA_EINVAL, "wall medians need a C-ordered volume (the order of a wall's voxels decides ties)");
    if ((uint64_t)c->wall_records >= (1ull << 31)) return fail(TA_EINVAL, "too many records for the wall medians");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint64_t n = (uint64_t)c->wall_records;
    c->wall_median_count = -1;
    *nwalls = 0;
    if (ms_out) *ms_out = 0.0;
    if (n == 0) { c->wall_median_count = 0; return TA_OK; }
    if (n >= (1ull << 32)) return fail(TA_EINVAL, "too many records (%llu) for the grouped fetch", (unsigned long long)n);
    DevBuf buf, scratch, starts;
    uint32_t* gpa = nullptr; int32_t* gco = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
    rc = e == hipSuccess ? wall_records_device(c, true, buf, &gpa, &gco, nullptr, nullptr) : TA_EHIP;
    if (rc == TA_OK) rc = scratch.reserve(ta::wall_median_scratch_bytes(n));
    if (rc == TA_OK) rc = starts.reserve(n * 4 + 16);
    uint64_t E = 0;
    uint32_t status = 0;
    if (rc == TA_OK) {
        uint64_t* total_dev = nullptr;
        ta::launch_wall_starts(c->stream, gpa, n, scratch.p, (uint32_t*)starts.p, &total_dev);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&E, total_dev, sizeof(E), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) rc = c->wall_medians.reserve(E * 24 + 16);
        if (e == hipSuccess && rc == TA_OK) {
            uint32_t* op = (uint32_t*)c->wall_medians.p;
            uint32_t* os = op + 2 * E;
            int32_t* om = (int32_t*)(os + E);
            uint32_t* st = (uint32_t*)scratch.p;                          // (the flags are dead: their first word takes the status)
            e = hipMemsetAsync(st, 0, 4, c->stream);
            if (e == hipSuccess) {
                ta::launch_wall_medians(c->stream, gpa, gco, (const uint32_t*)starts.p, (uint32_t)E, n, max_iter, op, os, om, st);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&status, st, 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
    }
    float ms = 0.f;
    if (rc == TA_OK && e == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    buf.release(); scratch.release(); starts.release();
    if (rc != TA_OK) return rc;
    if (e != hipSuccess) return fail(TA_EHIP, "wall medians: %s", hipGetErrorString(e));
    // (walls still moving after max_iter passes are MARKED -- bit 31 of their size word -- not refused: a caller asks for
    //  some walls, and one that nobody asks for -- the background's, say -- must not fail the rest)
    c->wall_median_count = (int64_t)E;
    *nwalls = (int64_t)E;
    if (ms_out) *ms_out = ms;
    return TA_OK;
}

TA_API int ta_wall_medians_get(ta_ctx* c, uint32_t* pairs, uint32_t* sizes, int32_t* medians) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if (c->wall_median_count < 0) return fail(TA_EINVAL, "call ta_wall_medians first");
    const uint64_t E = (uint64_t)c->wall_median_count;
    if (E == 0) return TA_OK;
    if (!pairs || !sizes || !medians) return fail(TA_EINVAL, "NULL output array");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint32_t* op = (const uint32_t*)c->wall_medians.p;
    TA_HIP(hipMemcpyAsync(pairs, op, E * 8, hipMemcpyDeviceToHost, c->stream));
    TA_HIP(hipMemcpyAsync(sizes, op + 2 * E, E * 4, hipMemcpyDeviceToHost, c->stream));
    TA_HIP(hipMemcpyAsync(medians, op + 3 * E, E * 12, hipMemcpyDeviceToHost, c->stream));
    TA_HIP(hipStreamSynchronize(c->stream));
    return TA_OK;
}


TA_API int ta_wall_voxels_get(ta_ctx* c, uint32_t* pairs, int32_t* coords, double* ms_out) {
    return wall_voxels_fetch(c, pairs, coords, ms_out, false);
}

TA_API int ta_wall_voxels_get_by_pair(ta_ctx* c, uint32_t* pairs, int32_t* coords, double* ms_out) {
    return wall_voxels_fetch(c, pairs, coords, ms_out, true);
}

TA_API int ta_volume_max_label(ta_ctx* c, uint32_t* max_label) {
    if (!c || !max_label) return fail(TA_EINVAL, "NULL argument");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint64_t nvox = (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2];
    if ((rc = settle_rerank(c)) != TA_OK) return rc;
    ta::launch_max_label(c->stream, c->vol, c->itemsize, nvox, maxlab_dev(c));
    uint32_t v = 0;
    TA_HIP(hipMemcpyAsync(&v, maxlab_dev(c), sizeof(v), hipMemcpyDeviceToHost, c->stream));
    TA_HIP(hipStreamSynchronize(c->stream));
    TA_HIP(hipGetLastError());
    *max_label = v;
    c->vol_max = v;
    return TA_OK;
}

// ---- sparse label ids -------------------------------------------------------------------------------------------
namespace {
// census of `ids` (host, ascending, unique; NULL: of the resident volume itself) on the context; leaves census_n / census_ids
int build_census(ta_ctx* c, const uint32_t* ids, uint32_t n_ids) {
    int rc;
    const uint64_t nvox = (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2];
    uint32_t top = 0, listed = 0;
    uint64_t cap = 0;
    bool have_list = false;
    DevBuf& list = c->census_list;                      // (kept: allocating and freeing it costs more than the pass it saves)
    if (ids) {
        for (uint32_t i = 1; i < n_ids; ++i)
            if (ids[i] <= ids[i - 1]) return fail(TA_EINVAL, "ids must be ascending and unique (ids[%u]=%u after %u)", i, ids[i], ids[i - 1]);
        top = n_ids ? ids[n_ids - 1] : 0u;
    } else if (c->vol_max >= 0 && c->vol == c->owned_vol.p) {
        top = (uint32_t)c->vol_max;                     // (ta_volume_max_label has been here, and only this library writes
                                                        //  a volume it uploaded itself: no second pass)
    } else {
        // ONE pass over the voxels where the maximum is not known: the workgroups' label sets go to a list, the list gives the
        // maximum (the table's size) and is marked afterwards -- a few hundred thousand entries against a second read of the volume
        cap = ta::census_list_capacity(nvox);
        const uint32_t parts = ta::census_list_parts();
        if (list.reserve(ta::census_list_head_bytes() + (uint64_t)parts * cap * 4) == TA_OK) {
            TA_HIP(hipMemsetAsync(list.p, 0, ta::census_list_head_bytes(), c->stream));
            if (ta::launch_census_list(c->stream, c->vol, c->itemsize, nvox, c->mdims[2], list.p, (uint32_t)cap)) {
                std::vector<uint32_t> head;
                try { head.resize(2 * (size_t)parts); } catch (...) { return fail(TA_ENOMEM, "out of host memory"); }
                TA_HIP(hipMemcpyAsync(head.data(), list.p, ta::census_list_head_bytes(), hipMemcpyDeviceToHost, c->stream));
                TA_HIP(hipStreamSynchronize(c->stream));
                have_list = true;
                for (uint32_t p = 0; p < parts; ++p) {
                    if (head[2 * p] > cap) have_list = false;
                    if (head[2 * p] > listed) listed = head[2 * p];
                    if (head[2 * p + 1] > top) top = head[2 * p + 1];
                }
                if (have_list) c->vol_max = top; else top = 0;
            }
        }
        if (!have_list) {                               // (rows that are not whole vectors, or a volume of noise: the two passes)
            ta::launch_max_label(c->stream, c->vol, c->itemsize, nvox, maxlab_dev(c));
            TA_HIP(hipMemcpyAsync(&top, maxlab_dev(c), sizeof(top), hipMemcpyDeviceToHost, c->stream));
            TA_HIP(hipStreamSynchronize(c->stream));
            c->vol_max = top;
        }
    }
    c->census_n = -1;
    if ((rc = c->census.reserve(ta::census_bytes(top))) != TA_OK) return rc;
    DevBuf scratch, staged;
    if ((rc = scratch.reserve(ta::census_scratch_bytes(top))) != TA_OK) return rc;
    hipError_t e = hipMemsetAsync(c->census.p, 0, ta::census_bytes(top), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(scratch.p, 0, ta::census_scratch_bytes(top), c->stream);
    if (e == hipSuccess && ids && n_ids) {
        if ((rc = staged.reserve((uint64_t)n_ids * 4)) != TA_OK) { scratch.release(); return rc; }
        e = hipMemcpyAsync(staged.p, ids, (uint64_t)n_ids * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) ta::launch_census_from_ids(c->stream, (const uint32_t*)staged.p, n_ids, c->census.p, scratch.p, top);
    } else if (e == hipSuccess && !ids && have_list) {
        ta::launch_census_from_list(c->stream, list.p, (uint32_t)cap, listed, c->census.p, scratch.p, top);
    } else if (e == hipSuccess && !ids) {
        ta::launch_census_mark(c->stream, c->vol, c->itemsize, nvox, c->mdims[2], c->census.p, scratch.p, top);
    }
    uint32_t* total_dev = nullptr;
    uint32_t total = 0;
    if (e == hipSuccess) {
        ta::launch_census_scan(c->stream, c->census.p, top, scratch.p, nullptr, &total_dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&total, total_dev, sizeof(total), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && total) {
        rc = c->census_ids.reserve((uint64_t)total * 4);
        if (rc != TA_OK) { scratch.release(); staged.release(); return rc; }
        ta::launch_census_scan(c->stream, c->census.p, top, scratch.p, (uint32_t*)c->census_ids.p, nullptr);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    scratch.release();
    staged.release();
    if (e != hipSuccess) return fail(TA_EHIP, "label census: %s", hipGetErrorString(e));
    c->census_max = top;
    c->census_n = (int64_t)total;
    c->census_of_volume = ids == nullptr;
    return TA_OK;
}
}  // namespace

TA_API int ta_volume_label_census(ta_ctx* c, uint32_t* max_label, uint32_t* n_present) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if (c->compact) return fail(TA_EINVAL, "the context is compacted: its census is the one it was compacted with");
    if ((rc = build_census(c, nullptr, 0)) != TA_OK) return rc;
    if (max_label) *max_label = c->census_max;
    if (n_present) *n_present = (uint32_t)c->census_n;
    return TA_OK;
}

TA_API int ta_label_census_get(ta_ctx* c, uint32_t* ids) {
    if (!c || !ids) return fail(TA_EINVAL, "NULL argument");
    if (c->census_n < 0) return fail(TA_EINVAL, "no label census on this context");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if (c->census_n == 0) return TA_OK;
    TA_HIP(hipMemcpyAsync(ids, c->census_ids.p, (uint64_t)c->census_n * 4, hipMemcpyDeviceToHost, c->stream));
    TA_HIP(hipStreamSynchronize(c->stream));
    return TA_OK;
}

TA_API int ta_volume_compact_labels(ta_ctx* c, const uint32_t* ids, uint32_t n_ids, uint32_t* n_rows) {
    if (!c || (!ids && n_ids)) return fail(TA_EINVAL, "NULL argument");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    c->compact = false;
    c->rerank_check = false;
    c->extracted = c->checked = false;
    if (ids || c->census_n < 0 || !c->census_of_volume)          // (ids == NULL means THIS volume's census: never a caller's list left behind)
        if ((rc = build_census(c, ids, n_ids)) != TA_OK) return rc;
    if (c->census_n >= (1ll << 28)) return fail(TA_ERANGE, "%lld label ids are present: too many for per-label rows", (long long)c->census_n);
    const uint64_t nvox = (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2];
    if ((rc = c->compact_vol.reserve(nvox * c->itemsize + 64)) != TA_OK) return rc;
    uint32_t status = 0;
    hipError_t e = hipMemsetAsync(maxlab_dev(c), 0, sizeof(uint32_t), c->stream);       // (the word is free between max-label passes)
    if (e == hipSuccess) {
        ta::launch_census_rank(c->stream, c->vol, c->compact_vol.p, c->itemsize, nvox, c->census.p, c->census_max, maxlab_dev(c));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&status, maxlab_dev(c), sizeof(status), hipMemcpyDeviceToHost, c->stream);
    try { c->h_ids.resize((size_t)c->census_n); } catch (...) { return fail(TA_ENOMEM, "out of host memory"); }
    if (e == hipSuccess && c->census_n)
        e = hipMemcpyAsync(c->h_ids.data(), c->census_ids.p, (uint64_t)c->census_n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(TA_EHIP, "compact labels: %s", hipGetErrorString(e));
    if (status) return fail(TA_ERANGE, "the volume holds a label id that is not in the list it was to be compacted with");
    c->compact = true;
    c->auto_tile_shift = 0;
    if (n_rows) *n_rows = (uint32_t)c->census_n;
    return TA_OK;
}

TA_API int ta_volume_is_compact(ta_ctx* c, int* compact, uint32_t* n_rows) {
    if (!c || !compact) return fail(TA_EINVAL, "NULL argument");
    *compact = c->compact ? 1 : 0;
    if (n_rows) *n_rows = c->compact ? (uint32_t)c->census_n : 0u;
    return TA_OK;
}

TA_API int ta_volume_rerank(ta_ctx* c) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    if (!c->compact) return fail(TA_EINVAL, "the context is not compacted");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const uint64_t nvox = (uint64_t)c->mdims[0] * c->mdims[1] * c->mdims[2];
    c->extracted = c->checked = false;
    // asynchronous on the context's stream: the "id not in the census" word travels to the host with the flags of the next
    // extraction, whose getters then answer TA_ERANGE
    hipError_t e = hipMemsetAsync(maxlab_dev(c), 0, sizeof(uint32_t), c->stream);
    if (e == hipSuccess) {
        ta::launch_census_rank(c->stream, c->vol, c->compact_vol.p, c->itemsize, nvox, c->census.p, c->census_max, maxlab_dev(c));
        e = hipGetLastError();
    }
    if (e != hipSuccess) return fail(TA_EHIP, "re-rank: %s", hipGetErrorString(e));
    c->rerank_check = true;
    c->vol_max = -1;
    return TA_OK;
}

TA_API int ta_volume_uncompact(ta_ctx* c) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if (c->compact) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        c->compact = false;
        c->rerank_check = false;
        c->extracted = c->checked = false;
        c->compact_vol.release();
        c->auto_tile_shift = 0;
    }
    return TA_OK;
}

TA_API int ta_volume_owned_planes(ta_ctx* c, int64_t* planes) {
    if (!c || !planes) return fail(TA_EINVAL, "NULL argument");
    *planes = c->vol ? c->mdims[0] - c->first_owned : 0;
    return TA_OK;
}

TA_API int ta_volume_plane_events(ta_ctx* c, uint64_t* events) {
    if (!c || !events) return fail(TA_EINVAL, "NULL argument");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const int64_t owned = c->mdims[0] - c->first_owned;
    if (owned <= 0) return TA_OK;
    DevBuf d;
    if ((rc = d.reserve((uint64_t)owned * sizeof(uint64_t))) != TA_OK) return rc;
    const char* first = (const char*)c->vol + (size_t)c->first_owned * c->mdims[1] * c->mdims[2] * c->itemsize;
    ta::launch_plane_events(c->stream, first, c->itemsize, owned, c->mdims[1], c->mdims[2], (uint64_t*)d.p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(events, d.p, (size_t)owned * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    d.release();
    if (e != hipSuccess) return fail(TA_EHIP, "plane events: %s", hipGetErrorString(e));
    return TA_OK;
}

TA_API int ta_bind_accumulators(ta_ctx* c, void* sums_dev, void* boxes_dev, uint32_t max_label) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if ((sums_dev == nullptr) != (boxes_dev == nullptr)) return fail(TA_EINVAL, "bind both buffers or neither");
    if (sums_dev && (((uintptr_t)sums_dev & 15) || ((uintptr_t)boxes_dev & 7)))
        return fail(TA_EINVAL, "bound accumulators must be 16-byte (sums) / 8-byte (boxes) aligned");
    c->bound = sums_dev != nullptr;
    c->bound_max_label = max_label;
    if (c->bound) { c->sums = (uint64_t*)sums_dev; c->boxes = (int32_t*)boxes_dev; }
    else { c->sums = nullptr; c->boxes = nullptr; }
    c->extracted = c->checked = false;
    return TA_OK;
}

TA_API int ta_extract(ta_ctx* c, uint32_t feature_mask, uint32_t max_label) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if (!c->vol) return fail(TA_EINVAL, "no volume set");
    if (feature_mask == 0 || (feature_mask & ~TA_F_ALL)) return fail(TA_EINVAL, "bad feature mask 0x%x", feature_mask);
    if (max_label >= (1u << 28)) return fail(TA_EINVAL, "max_label %u too large for dense per-label rows", max_label);
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    {   // exactness guard: every u64 sum must stay below 2^64
        const long double g0 = (long double)(c->a_origin + c->mdims[0]), nv = (long double)c->mdims[0] * c->mdims[1] * c->mdims[2];
        const long double gm = std::max(g0, std::max((long double)c->mdims[1], (long double)c->mdims[2]));
        if (nv * gm * gm >= 1.8e19L) return fail(TA_EINVAL, "volume too large for exact 64-bit second moments");
    }
    const uint64_t nlabels = (uint64_t)max_label + 1;
    if (c->bound) {
        if (c->bound_max_label != max_label)
            return fail(TA_EINVAL, "bound accumulators are sized for max_label=%u, not %u", c->bound_max_label, max_label);
    } else {
        if ((rc = c->own_sums.reserve(nlabels * ta::NSUM * 8)) != TA_OK) return rc;
        if ((rc = c->own_boxes.reserve(nlabels * ta::NBOX * 4)) != TA_OK) return rc;
        c->sums = (uint64_t*)c->own_sums.p;
        c->boxes = (int32_t*)c->own_boxes.p;
    }
    c->max_label = max_label;
    c->feature_mask = feature_mask;
    {
        const bool adj = feature_mask & TA_F_ADJACENCY;
        // never below a size the table has already grown to (a fixed TA_OPT_PAIR_SLOTS is a starting size)
        int want = c->opt_pair_log2 ? std::max(c->opt_pair_log2, c->pkeys.p ? c->pair_log2 : 0)
                                    : std::max(c->pkeys.p ? c->pair_log2 : 4, adj ? auto_pair_log2(max_label) : 4);
        if ((rc = ensure_pair_table(c, want)) != TA_OK) return rc;
    }
    c->extracted = true;
    c->checked = false;
    c->exchanged = false;
    c->shared_packed = false;
    c->reduced = false;
    c->host_pairs_ready = false;
    return run_extract(c);
}

TA_API int ta_get_labels(ta_ctx* c, uint64_t* count, int32_t* bbox, uint64_t* sum1, uint64_t* sum2) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if ((rc = finish_extract(c)) != TA_OK) return rc;
    const uint64_t n = (uint64_t)c->max_label + 1;
    std::vector<uint64_t> hs;
    std::vector<int32_t> hb;
    try {
        if (count || sum1 || sum2) hs.resize(n * ta::NSUM);
        if (bbox) hb.resize(n * ta::NBOX);
    } catch (...) { return fail(TA_ENOMEM, "out of host memory"); }
    if (!hs.empty()) TA_HIP(hipMemcpyAsync(hs.data(), c->sums, hs.size() * 8, hipMemcpyDeviceToHost, c->stream));
    if (!hb.empty()) TA_HIP(hipMemcpyAsync(hb.data(), c->boxes, hb.size() * 4, hipMemcpyDeviceToHost, c->stream));
    TA_HIP(hipStreamSynchronize(c->stream));
    // second-moment slot of an (array axis, array axis) pair
    auto pair_slot = [](int x, int y) { if (x > y) std::swap(x, y); return x == 0 ? y : (x == 1 ? 2 + y : 5); };
    static const int mem_pair[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
    const bool mom2 = c->feature_mask & TA_F_MOMENT2;
    if (c->perm[0] == 0 && c->perm[1] == 1 && c->perm[2] == 2) {
        // C-ordered input (array axes = memory axes): no permutation, one tight loop per output
        if (count) for (uint64_t l = 0; l < n; ++l) count[l] = hs[l * ta::NSUM];
        if (sum1) for (uint64_t l = 0; l < n; ++l) { sum1[3 * l] = hs[l * ta::NSUM + 1]; sum1[3 * l + 1] = hs[l * ta::NSUM + 2]; sum1[3 * l + 2] = hs[l * ta::NSUM + 3]; }
        if (sum2) {
            if (mom2) for (uint64_t l = 0; l < n; ++l) memcpy(sum2 + 6 * l, &hs[l * ta::NSUM + 4], 6 * sizeof(uint64_t));
            else memset(sum2, 0, n * 6 * sizeof(uint64_t));
        }
        if (bbox) for (uint64_t l = 0; l < n; ++l) {
            const bool present = hb[l * 6] != INT32_MAX;
            for (int k = 0; k < 3; ++k) { bbox[l * 6 + k] = present ? hb[l * 6 + k] : -1; bbox[l * 6 + 3 + k] = present ? (-hb[l * 6 + 3 + k] + 1) : -1; }
        }
        return TA_OK;
    }
    for (uint64_t l = 0; l < n; ++l) {
        if (count) count[l] = hs[l * ta::NSUM];
        if (sum1) for (int k = 0; k < 3; ++k) sum1[l * 3 + c->perm[k]] = hs[l * ta::NSUM + 1 + k];
        // (without TA_F_MOMENT2 the device columns are not defined -- the rare table-spill path writes cross terms there --
        //  and the getter answers zero)
        if (sum2) for (int q = 0; q < 6; ++q)
            sum2[l * 6 + pair_slot(c->perm[mem_pair[q][0]], c->perm[mem_pair[q][1]])] = mom2 ? hs[l * ta::NSUM + 4 + q] : 0ull;
        if (bbox) {
            const bool present = hb[l * 6] != INT32_MAX;
            for (int k = 0; k < 3; ++k) {
                bbox[l * 6 + c->perm[k]] = present ? hb[l * 6 + k] : -1;
                bbox[l * 6 + 3 + c->perm[k]] = present ? (-hb[l * 6 + 3 + k] + 1) : -1;
            }
        }
    }
    return TA_OK;
}

TA_API int ta_adjacency_scope(ta_ctx* c, int* scope) {
    if (!c || !scope) return fail(TA_EINVAL, "NULL argument");
    if (!c->extracted || !(c->feature_mask & TA_F_ADJACENCY))
        return fail(TA_EINVAL, "no extraction with adjacency has been run on this context");
    *scope = !c->exchanged ? TA_ADJ_LOCAL : (c->shared_packed ? TA_ADJ_PARTIAL : TA_ADJ_MERGED);
    return TA_OK;
}

TA_API int ta_adjacency_size(ta_ctx* c, int64_t* npairs) {
    if (!c || !npairs) return fail(TA_EINVAL, "NULL argument");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if ((rc = finish_extract(c)) != TA_OK) return rc;
    *npairs = c->npairs;
    return TA_OK;
}

TA_API int ta_adjacency_get(ta_ctx* c, uint32_t* lo, uint32_t* hi, uint64_t* faces) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if ((rc = finish_extract(c)) != TA_OK) return rc;
    const uint64_t n = (uint64_t)c->npairs;
    if (!c->host_pairs_ready) {
        if ((rc = c->h_pairs.reserve(n * 32 + 16)) != TA_OK) return rc;
        if (n) {
            // sorted by (lo, hi) on the device (kernels_pairsort.hip: counting sort over the label rows, rank inside a bucket;
            // a std::sort of ~10^5 records used to cost more than the sweep, a 64-bit library radix sort 0.45 ms)
            if (n >= (1ull << 32)) return fail(TA_EINVAL, "too many pairs (%llu)", (unsigned long long)n);
            const uint64_t kb = n * 8;
            DevBuf& buf = c->sort_buf;
            if ((rc = buf.reserve(kb + n * 24 + ta::pairs_sort_scratch_bytes(n, c->max_label) + 64)) != TA_OK) return rc;
            char* p = (char*)buf.p;
            uint64_t* ks = (uint64_t*)p; p += kb;
            uint64_t* fo = (uint64_t*)p; p += n * 24;
            const bool has_voxel = sweep_vol(c) && c->mdims[0] - c->first_owned > 0 && c->mdims[1] > 0 && c->mdims[2] > 0;
            hipError_t e = ta::launch_pairs_sort(c->stream, (const uint64_t*)c->out_keys.p, (const uint64_t*)c->out_faces.p, n, c->max_label,
                                                 p, ks, fo, has_voxel ? sweep_vol(c) : nullptr, c->itemsize,
                                                 (int64_t)c->first_owned * c->mdims[1] * c->mdims[2]);
            // (keys and faces sit back to back in the sort's output: one copy)
            if (e == hipSuccess) e = hipMemcpyAsync(c->h_pairs.p, ks, n * 32, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) return fail(TA_EHIP, "adjacency sort: %s", hipGetErrorString(e));
        }
        c->host_pairs_ready = true;
    }
    const bool identity = c->perm[0] == 0 && c->perm[1] == 1 && c->perm[2] == 2;
    const uint64_t* h_keys = (const uint64_t*)c->h_pairs.p;
    const uint64_t* h_faces = h_keys + n;
    if (c->compact) {               // rows are ranks, label values are ids (order-preserving: the list stays sorted)
        const uint64_t nid = c->h_ids.size();
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t a = h_keys[i] >> 32, b = h_keys[i] & 0xffffffffu;
            if (a >= nid || b >= nid) return fail(TA_ERANGE, "adjacency holds rank %llu, the census has %llu ids", (unsigned long long)std::max(a, b), (unsigned long long)nid);
            if (lo) lo[i] = c->h_ids[a];
            if (hi) hi[i] = c->h_ids[b];
        }
    } else {
        if (lo) for (uint64_t i = 0; i < n; ++i) lo[i] = (uint32_t)(h_keys[i] >> 32);
        if (hi) for (uint64_t i = 0; i < n; ++i) hi[i] = (uint32_t)(h_keys[i] & 0xffffffffu);
    }
    if (faces && identity && n) memcpy(faces, h_faces, n * 3 * sizeof(uint64_t));      // C-ordered input: a plain copy
    else if (faces) for (uint64_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) faces[3 * i + c->perm[k]] = h_faces[3 * i + k];
    return TA_OK;
}

TA_API int ta_timing(ta_ctx* c, double* ms_sweep, double* ms_adjacency, double* ms_total, uint64_t* bytes_read) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if (!c->extracted) return fail(TA_EINVAL, "no extraction has been run on this context");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const size_t nslots = c->ring.size() / 2;
    // (durations no event recorded are NaN, not 0: "not measured" must not read as "took no time" in a bandwidth figure)
    float a = NAN, b = NAN, t = NAN;
    if (c->extract_seq > c->ring_since && nslots) {          // the last extraction recorded its sweep events
        hipEvent_t ev_a = c->ring[2 * ((c->extract_seq - 1) % nslots)], ev_b = c->ring[2 * ((c->extract_seq - 1) % nslots) + 1];
        TA_HIP(hipEventSynchronize(ev_b));
        TA_HIP(hipEventElapsedTime(&a, ev_a, ev_b));
        if (c->timing >= 2) {
            TA_HIP(hipEventSynchronize(c->ev[3]));
            TA_HIP(hipEventElapsedTime(&b, ev_b, c->ev[3]));
            TA_HIP(hipEventElapsedTime(&t, c->ev[0], c->ev[3]));
        }
    }      // (else: the last extraction recorded no events -- TA_OPT_TIMING is 0)
    if (ms_sweep) *ms_sweep = a;
    if (ms_adjacency) *ms_adjacency = b;
    if (ms_total) *ms_total = t;
    if (bytes_read)
        *bytes_read = (uint64_t)(c->mdims[0] - c->first_owned) * c->mdims[1] * c->mdims[2] * c->itemsize;
    return TA_OK;
}

TA_API int ta_timing_series(ta_ctx* c, double* ms_sweep, int capacity, int* count) {
    if (!c || !count || (capacity > 0 && !ms_sweep) || capacity < 0) return fail(TA_EINVAL, "NULL ctx / output or negative capacity");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const size_t nslots = c->ring.size() / 2;
    uint64_t first = c->ring_since, last = c->extract_seq;          // extractions [first, last) have valid events
    if (last - first > (uint64_t)capacity) first = last - (uint64_t)capacity;
    *count = 0;
    if (c->stream) TA_HIP(hipStreamSynchronize(c->stream));
    for (uint64_t q = first; q < last && nslots; ++q) {
        float ms = 0;
        TA_HIP(hipEventElapsedTime(&ms, c->ring[2 * (q % nslots)], c->ring[2 * (q % nslots) + 1]));
        ms_sweep[(*count)++] = ms;
    }
    return TA_OK;
}

TA_API int ta_read_probe(ta_ctx* c, const void* dev_ptr, uint64_t bytes, int repeats, double* ms_best) {
    if (!c || !dev_ptr || !ms_best) return fail(TA_EINVAL, "NULL argument");
    if (((uintptr_t)dev_ptr & 15) || bytes < 16) return fail(TA_EINVAL, "the probe needs a 16-byte aligned buffer of at least 16 bytes");
    if (repeats < 1) repeats = 1;
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if ((rc = settle_rerank(c)) != TA_OK) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best = -1.0;
    for (int r = 0; r < repeats + 1 && e == hipSuccess; ++r) {            // the first launch is a warm-up
        e = hipEventRecord(e0, c->stream);
        if (e == hipSuccess) { ta::launch_read_probe(c->stream, dev_ptr, bytes, maxlab_dev(c)); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && r > 0 && (best < 0 || ms < best)) best = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(TA_EHIP, "read probe: %s", hipGetErrorString(e));
    *ms_best = best;
    return TA_OK;
}

TA_API int ta_debug_counters(ta_ctx* c, uint32_t out[16]) {
    if (!c || !out) return fail(TA_EINVAL, "NULL argument");
    if (!c->extracted) return fail(TA_EINVAL, "no extraction has been run on this context");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    TA_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 16; ++i) out[i] = i < ta::NFLAGS ? c->h_small[i] : 0u;
    return TA_OK;
}

TA_API int ta_accumulators_reduced(ta_ctx* c) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if (!c->extracted) return fail(TA_EINVAL, "no extraction has been run on this context");
    c->reduced = true;
    return TA_OK;
}

TA_API int ta_accumulators_device(ta_ctx* c, void** sums_dev, void** boxes_dev, uint32_t* max_label) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if (!c->extracted) return fail(TA_EINVAL, "no extraction has been run on this context");
    if (sums_dev) *sums_dev = c->sums;
    if (boxes_dev) *boxes_dev = c->boxes;
    if (max_label) *max_label = c->max_label;
    return TA_OK;
}

TA_API int ta_adjacency_device(ta_ctx* c, void** keys_dev, void** faces_dev, int64_t* npairs) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if ((rc = finish_extract(c)) != TA_OK) return rc;
    if (keys_dev) *keys_dev = c->out_keys.p;
    if (faces_dev) *faces_dev = c->out_faces.p;
    if (npairs) *npairs = c->npairs;
    return TA_OK;
}

TA_API int ta_adjacency_export(ta_ctx* c, void* keys_dst_dev, void* faces_dst_dev, int64_t capacity_pairs) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if ((rc = finish_extract(c)) != TA_OK) return rc;
    if (capacity_pairs < c->npairs) return fail(TA_EINVAL, "export buffers hold %lld pairs, %lld needed", (long long)capacity_pairs, (long long)c->npairs);
    if (c->npairs > 0) {
        if (!keys_dst_dev || !faces_dst_dev) return fail(TA_EINVAL, "NULL export buffer");
        TA_HIP(hipMemcpyAsync(keys_dst_dev, c->out_keys.p, (uint64_t)c->npairs * 8, hipMemcpyDeviceToDevice, c->stream));
        TA_HIP(hipMemcpyAsync(faces_dst_dev, c->out_faces.p, (uint64_t)c->npairs * 24, hipMemcpyDeviceToDevice, c->stream));
    }
    return TA_OK;
}

TA_API int ta_adjacency_merge(ta_ctx* c, const void* keys_dev, const void* faces_dev, int64_t npairs) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    if (npairs < 0 || (npairs > 0 && (!keys_dev || !faces_dev))) return fail(TA_EINVAL, "bad pair list");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    if ((rc = finish_extract(c)) != TA_OK) return rc;
    if (!(c->feature_mask & TA_F_ADJACENCY)) return fail(TA_EINVAL, "the last extraction did not request adjacency");
    // local list (already collected, so the table is clean) + foreign list -> table -> collect again
    ta::PairTable pt = pair_table(c);
    DevBuf local_k, local_f;
    const uint64_t nl = (uint64_t)c->npairs;
    if ((rc = local_k.reserve(nl * 8 + 8)) != TA_OK) return rc;
    if ((rc = local_f.reserve(nl * 24 + 8)) != TA_OK) { local_k.release(); return rc; }
    hipError_t e = hipSuccess;
    if (nl) {
        e = hipMemcpyAsync(local_k.p, c->out_keys.p, nl * 8, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(local_f.p, c->out_faces.p, nl * 24, hipMemcpyDeviceToDevice, c->stream);
    }
    if (e == hipSuccess) e = hipMemsetAsync(c->small.p, 0, SMALL_WORDS * sizeof(uint32_t), c->stream);
    if (e != hipSuccess) { local_k.release(); local_f.release(); return fail(TA_EHIP, "merge staging: %s", hipGetErrorString(e)); }
    ta::launch_pairs_insert(c->stream, pt, (const uint64_t*)local_k.p, (const uint64_t*)local_f.p, nl, flags_dev(c));
    ta::launch_pairs_insert(c->stream, pt, (const uint64_t*)keys_dev, (const uint64_t*)faces_dev, (uint64_t)npairs, flags_dev(c));
    ta::launch_pairs_collect(c->stream, pt, (uint64_t*)c->out_keys.p, (uint64_t*)c->out_faces.p, cursor_dev(c));
    e = hipMemcpyAsync(c->h_small, c->small.p, SMALL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    local_k.release(); local_f.release();
    if (e != hipSuccess) return fail(TA_EHIP, "merge: %s", hipGetErrorString(e));
    if (c->h_small[ta::FLAG_PAIR_OVERFLOW])
        return fail(TA_ECAPACITY, "adjacency table overflow while merging (2^%d slots); raise TA_OPT_PAIR_SLOTS", c->pair_log2);
    c->npairs = (int64_t)c->h_small[ta::NFLAGS];
    c->host_pairs_ready = false;
    return TA_OK;
}

TA_API int ta_adjacency_pack(ta_ctx* c, void* block_dev, int64_t capacity_pairs) {
    if (!c || !block_dev) return fail(TA_EINVAL, "NULL argument");
    if (capacity_pairs < 1) return fail(TA_EINVAL, "capacity_pairs must be >= 1");
    if (!c->extracted || !(c->feature_mask & TA_F_ADJACENCY))
        return fail(TA_EINVAL, "no extraction with adjacency has been run on this context");
    if (c->exchanged) return fail(TA_EINVAL, "the adjacency of this extraction was already exchanged");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    ta::launch_pairs_pack(c->stream, (const uint64_t*)c->out_keys.p, (const uint64_t*)c->out_faces.p, cursor_dev(c),
                          flags_dev(c), (uint64_t*)block_dev, (uint64_t)capacity_pairs);
    TA_HIP(hipGetLastError());
    return TA_OK;
}

TA_API int ta_adjacency_pack_shared(ta_ctx* c, void* block_dev, int64_t capacity_pairs) {
    if (!c || !block_dev) return fail(TA_EINVAL, "NULL argument");
    if (capacity_pairs < 1) return fail(TA_EINVAL, "capacity_pairs must be >= 1");
    if (!c->extracted || !(c->feature_mask & TA_F_ADJACENCY))
        return fail(TA_EINVAL, "no extraction with adjacency has been run on this context");
    if (c->exchanged) return fail(TA_EINVAL, "the adjacency of this extraction was already exchanged");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    const int64_t lo = c->a_origin, hi = c->a_origin + (c->mdims[0] - c->first_owned);
    TA_HIP(hipMemsetAsync(block_dev, 0, 8, c->stream));          // the block's pair count: the kernel's append cursor
    // (the local list can hold no more pairs than the table has slots)
    ta::launch_pairs_pack_shared(c->stream, pair_table(c), (const uint64_t*)c->out_keys.p, (const uint64_t*)c->out_faces.p,
                                 cursor_dev(c), flags_dev(c), c->boxes, c->max_label, lo, hi, (uint64_t*)block_dev,
                                 (uint64_t)capacity_pairs, 1ull << c->pair_log2);
    TA_HIP(hipGetLastError());
    c->table_clean = false;          // holds this rank's private pairs until ta_adjacency_merge_blocks collects
    c->shared_packed = true;
    return TA_OK;
}

TA_API int ta_adjacency_merge_blocks(ta_ctx* c, const void* blocks_dev, int nblocks, int64_t capacity_pairs) {
    if (!c || !blocks_dev) return fail(TA_EINVAL, "NULL argument");
    if (nblocks < 1 || capacity_pairs < 1) return fail(TA_EINVAL, "nblocks and capacity_pairs must be >= 1");
    if (!c->extracted || !(c->feature_mask & TA_F_ADJACENCY))
        return fail(TA_EINVAL, "no extraction with adjacency has been run on this context");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    // the collect of the extraction left the table clean: rebuild it from every rank's block
    ta::PairTable pt = pair_table(c);
    TA_HIP(hipMemsetAsync(c->small.p, 0, SMALL_WORDS * sizeof(uint32_t), c->stream));
    ta::launch_pairs_insert_blocks(c->stream, pt, (const uint64_t*)blocks_dev, nblocks, (uint64_t)capacity_pairs,
                                   flags_dev(c));
    ta::launch_pairs_collect(c->stream, pt, (uint64_t*)c->out_keys.p, (uint64_t*)c->out_faces.p, cursor_dev(c));
    TA_HIP(hipMemcpyAsync(c->h_small, c->small.p, SMALL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    TA_HIP(hipGetLastError());
    c->table_clean = true;
    c->exchanged = true;
    c->checked = false;
    c->host_pairs_ready = false;
    return TA_OK;
}

TA_API int ta_synth_voronoi(ta_ctx* c, void* dev_out, int itemsize, const int64_t dims[3], int64_t a_begin,
                     int64_t a_count, const int32_t* seeds, const int32_t grid[3], const int64_t* ell) {
    if (!c || !dev_out || !dims || !seeds || !grid) return fail(TA_EINVAL, "NULL argument");
    if (itemsize != 2 && itemsize != 4) return fail(TA_EINVAL, "itemsize must be 2 or 4");
    if (a_begin < 0 || a_count < 0 || a_begin + a_count > dims[0]) return fail(TA_EINVAL, "plane range out of bounds");
    const int64_t ncell = (int64_t)grid[0] * grid[1] * grid[2];
    if (ncell <= 0) return fail(TA_EINVAL, "empty seed grid");
    if (itemsize == 2 && ncell + 1 > 65535) return fail(TA_EINVAL, "uint16 cannot hold %lld labels", (long long)ncell + 1);
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    DevBuf dseeds, dell;
    if ((rc = dseeds.reserve((uint64_t)ncell * 12)) != TA_OK) return rc;
    hipError_t e = hipMemcpyAsync(dseeds.p, seeds, (uint64_t)ncell * 12, hipMemcpyHostToDevice, c->stream);
    const uint64_t nell = (uint64_t)(dims[0] + dims[1] + dims[2]);
    if (e == hipSuccess && ell) {
        if ((rc = dell.reserve(nell * 8)) != TA_OK) { dseeds.release(); return rc; }
        e = hipMemcpyAsync(dell.p, ell, nell * 8, hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) {
        ta::launch_synth(c->stream, dev_out, itemsize, dims, a_begin, a_count, (const int32_t*)dseeds.p, grid,
                         ell ? (const int64_t*)dell.p : nullptr);
        e = hipStreamSynchronize(c->stream);
    }
    dseeds.release(); dell.release();
    if (e != hipSuccess) return fail(TA_EHIP, "ta_synth_voronoi: %s", hipGetErrorString(e));
    TA_HIP(hipGetLastError());
    return TA_OK;
}

TA_API int ta_device_malloc(ta_ctx* c, uint64_t bytes, void** dev_ptr) {
    if (!c || !dev_ptr) return fail(TA_EINVAL, "NULL argument");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    *dev_ptr = nullptr;
    if (hipMalloc(dev_ptr, bytes ? bytes : 16) != hipSuccess) {
        (void)hipGetLastError();
        return fail(TA_ENOMEM, "hipMalloc of %llu bytes failed", (unsigned long long)bytes);
    }
    return TA_OK;
}

TA_API int ta_device_free(ta_ctx* c, void* dev_ptr) {
    if (!c) return fail(TA_EINVAL, "ctx is NULL");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    TA_HIP(hipStreamSynchronize(c->stream));
    if (dev_ptr) TA_HIP(hipFree(dev_ptr));
    return TA_OK;
}

TA_API int ta_memcpy_d2h(ta_ctx* c, void* host_dst, const void* dev_src, uint64_t bytes) {
    if (!c || (bytes && (!host_dst || !dev_src))) return fail(TA_EINVAL, "NULL argument");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    TA_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c->stream));
    TA_HIP(hipStreamSynchronize(c->stream));
    return TA_OK;
}

TA_API int ta_memcpy_h2d(ta_ctx* c, void* dev_dst, const void* host_src, uint64_t bytes) {
    if (!c || (bytes && (!dev_dst || !host_src))) return fail(TA_EINVAL, "NULL argument");
    int rc = use_device(c);
    if (rc != TA_OK) return rc;
    TA_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, c->stream));
    TA_HIP(hipStreamSynchronize(c->stream));
    return TA_OK;
}

}  // extern "C"
