// ta_device.h -- shared device-side definitions for the gfx950 kernels of libtissue_scan.
// Everything here is exact integer arithmetic: per-label accumulators are commutative
// u64 sums / i32 minima, so results are bit-identical under any atomic ordering, any tiling
// and any slab partition (the race-safety argument of SURVEY.md §5).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ta {

constexpr uint32_t INVALID_LABEL = 0xFFFFFFFFu;     // "outside the volume" sentinel (never a label)
constexpr uint64_t EMPTY_KEY = ~0ull;               // empty adjacency slot (lo<hi, so never a key)
constexpr int NSUM = 10;                            // count, s0,s1,s2, s00,s01,s02,s11,s12,s22
constexpr int NBOX = 6;                             // min0,min1,min2,-max0,-max1,-max2

// flag words written by kernels, read back by the host after the stream drains
enum { FLAG_RANGE = 0, FLAG_PAIR_OVERFLOW = 1, FLAG_LDS_LABEL_SPILL = 2, FLAG_LDS_PAIR_SPILL = 3,
       FLAG_EXCHANGE_OVERFLOW = 4, NFLAGS = 16 };

// Exchange block of one rank (multi-GPU adjacency merge), u64 words:
//   [0] pair count n (may exceed the capacity: receivers flag the overflow)   [1] status bits
//   [2 .. 2+cap) keys, EMPTY_KEY padded        [2+cap .. 2+4*cap) faces[cap][3]
constexpr int XHDR = 2;
constexpr uint64_t XSTATUS_RANGE = 1, XSTATUS_PAIR_OVERFLOW = 2;

struct PairTable {           // device-global open-addressing hash: key = lo<<32|hi
    uint64_t* keys;          // [cap], EMPTY_KEY when free
    uint64_t* faces;         // [cap][3] per memory axis
    uint32_t mask;           // cap - 1 (cap is a power of two)
};

struct SweepArgs {
    const void* vol;         // dense C-ordered [n0][n1][n2] labels (u16 or u32)
    int64_t n0, n1, n2;      // buffer dims; n0 counts the halo plane when first_owned == 1
    int64_t a_origin;        // global axis-0 coordinate of buffer plane `first_owned`
    int32_t first_owned;     // 0, or 1 when plane 0 is the low halo of a slab
    int32_t tile_planes;     // owned planes walked by one workgroup
    int32_t shape;           // uint32 volumes with adjacency: 0 = two rows of 256 columns a wave, 1 = two rows of 512 (kernels_scan.hip)
    int32_t vec_ok;          // 16-byte loads allowed: rows are 16-byte aligned, or the buffer is the library's own (gfx950 loads
                             // 16 bytes from any address; a strip that straddles the last row's end reads into the buffer's slack)
    uint32_t max_label;
    uint64_t* sums;          // [max_label+1][NSUM]
    int32_t* boxes;          // [max_label+1][NBOX]
    PairTable pairs;
    uint32_t* flags;         // [NFLAGS] flag words, then cursor, max label, and the hot-row pointer (HOT_PTR_WORD)
};
// Private rows for the "hot" label (the one at the slab's first voxel -- the background), see flush_tables /
// hot_reduce_kernel.  The pointer to them ([workgroups][HOTW] u64, or NULL) is NOT a kernel argument: the
// sweep kernels sit on the edge of their register budget, so it is parked behind the flag words by
// init_kernel and fetched with a scalar load where it is needed.
constexpr int HOTW = 16;             // u64 words of a hot row: sums u64[NSUM] | boxes i32[NBOX] | padding
constexpr int HOT_PTR_WORD = NFLAGS + 2;   // uint32 index into the flags buffer, 8-byte aligned
// Tile queues of the persistent sweep kernel (one counter per XCD, zeroed by init_kernel before every sweep): a workgroup
// takes the next tile of its own XCD's list -- consecutive tiles stay on one XCD's L2 -- and helps the others when it is empty.
constexpr int QUEUE_WORD = NFLAGS + 4, NQUEUES = 8;
constexpr int SMALL_WORDS_DEV = NFLAGS + 4 + NQUEUES;     // flags | pair cursor | max label | hot-row pointer (2) | tile queues
constexpr uint32_t NO_TILE = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t hash_pair(uint32_t lo, uint32_t hi) {
    return hash_u32(lo * 0x9E3779B1u ^ hash_u32(hi));
}

// the adds / minima of a tile flush into the device-global tables.  TA_FLUSH_SCOPE (ta_sweep_switches.h) is the memory scope they are
// issued with: agent = performed where every XCD sees them; workgroup (an ABLATION: results wrong across XCDs) = in the issuing XCD's L2
#ifndef TA_FLUSH_SCOPE
#define TA_FLUSH_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
__device__ __forceinline__ void flush_add(unsigned long long* p, unsigned long long v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, TA_FLUSH_SCOPE); }
__device__ __forceinline__ void flush_min(int32_t* p, int32_t v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, TA_FLUSH_SCOPE); }

// Insert/accumulate one pair into the device-global table.  Slots only ever go EMPTY -> key
// inside a launch, so a stale EMPTY read is repaired by the CAS and a non-EMPTY read is final.
// (`h`, `k`: the pair's home slot and the key a caller has read from it already -- it issues that read early, with other
//  work in between, so that the first global round trip of the probe is not waited for; see flush_tables.)
__device__ __forceinline__ void pair_add_global_from(const PairTable& pt, uint32_t lo, uint32_t hi, uint64_t f0, uint64_t f1, uint64_t f2,
                                                     uint32_t* flags, uint32_t h, uint64_t k) {
    const uint64_t key = ((uint64_t)lo << 32) | hi;
    for (uint32_t probe = 0; probe < 512u; ++probe) {
        if (k == EMPTY_KEY) {
            k = atomicCAS((unsigned long long*)&pt.keys[h], (unsigned long long)EMPTY_KEY,
                          (unsigned long long)key);
            if (k == EMPTY_KEY) k = key;
        }
        if (k == key) {
            unsigned long long* f = (unsigned long long*)&pt.faces[3ull * h];
            if (f0) flush_add(f + 0, (unsigned long long)f0);
            if (f1) flush_add(f + 1, (unsigned long long)f1);
            if (f2) flush_add(f + 2, (unsigned long long)f2);
            return;
        }
        h = (h + 1) & pt.mask;
        k = __hip_atomic_load(&pt.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    atomicOr(&flags[FLAG_PAIR_OVERFLOW], 1u);
}

__device__ inline void pair_add_global(const PairTable& pt, uint32_t lo, uint32_t hi,
                                       uint64_t f0, uint64_t f1, uint64_t f2, uint32_t* flags) {
    const uint64_t key = ((uint64_t)lo << 32) | hi;
    uint32_t h = hash_pair(lo, hi) & pt.mask;
    for (uint32_t probe = 0; probe < 512u; ++probe) {
        uint64_t k = __hip_atomic_load(&pt.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == EMPTY_KEY) {
            k = atomicCAS((unsigned long long*)&pt.keys[h], (unsigned long long)EMPTY_KEY,
                          (unsigned long long)key);
            if (k == EMPTY_KEY) k = key;
        }
        if (k == key) {
            unsigned long long* f = (unsigned long long*)&pt.faces[3ull * h];
            if (f0) atomicAdd(f + 0, (unsigned long long)f0);
            if (f1) atomicAdd(f + 1, (unsigned long long)f1);
            if (f2) atomicAdd(f + 2, (unsigned long long)f2);
            return;
        }
        h = (h + 1) & pt.mask;
    }
    atomicOr(&flags[FLAG_PAIR_OVERFLOW], 1u);
}

// Closed-form moments of one run of `n` voxels of a label along memory axis 0:
// voxels (a, b, c) for a = a0 .. a0+n-1.  s[] gets the NSUM contributions.
template <bool MOM2>
__device__ __forceinline__ void run_moments(uint64_t a0, uint32_t n, uint64_t b, uint64_t c,
                                            uint64_t s[NSUM]) {
    const uint64_t n64 = n;
    const uint64_t t1 = n64 * (n64 - 1) / 2;                   // sum_{i<n} i
    const uint64_t sa = n64 * a0 + t1;                         // sum a
    s[0] = n64; s[1] = sa; s[2] = n64 * b; s[3] = n64 * c;
    if (MOM2) {
        const uint64_t t2 = (n64 - 1) * n64 * (2 * n64 - 1) / 6;   // sum_{i<n} i^2
        s[4] = n64 * a0 * a0 + 2 * a0 * t1 + t2;               // sum a^2
        s[5] = b * sa; s[6] = c * sa;
        s[7] = n64 * b * b; s[8] = n64 * b * c; s[9] = n64 * c * c;
    } else {
        s[4] = s[5] = s[6] = s[7] = s[8] = s[9] = 0;
    }
}

// Accumulate a run straight into the global per-label rows (slow path / spill path).
template <bool MOM2>
__device__ inline void run_add_global(const SweepArgs& A, uint32_t label, uint64_t a0, uint32_t n,
                                      uint64_t b, uint64_t c) {
    if (label > A.max_label) { atomicOr(&A.flags[FLAG_RANGE], 1u); return; }
    uint64_t s[NSUM];
    run_moments<MOM2>(a0, n, b, c, s);
    unsigned long long* row = (unsigned long long*)&A.sums[(uint64_t)label * NSUM];
#pragma unroll
    for (int k = 0; k < (MOM2 ? NSUM : 4); ++k) atomicAdd(row + k, (unsigned long long)s[k]);
    int32_t* box = &A.boxes[(uint64_t)label * NBOX];
    atomicMin(box + 0, (int32_t)a0);  atomicMin(box + 3, -(int32_t)(a0 + n - 1));
    atomicMin(box + 1, (int32_t)b);   atomicMin(box + 4, -(int32_t)b);
    atomicMin(box + 2, (int32_t)c);   atomicMin(box + 5, -(int32_t)c);
}

template <typename T>
__device__ __forceinline__ uint32_t load_label(const void* vol, int64_t idx) {
    return (uint32_t)((const T*)vol)[idx];
}

}  // namespace ta
