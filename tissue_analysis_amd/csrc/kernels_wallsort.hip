// kernels_wallsort.hip -- the wall-voxel records grouped by label pair ON THE DEVICE (what every caller of the table wants).
//
// A STABLE least-significant-digit radix sort, hand-written (rounds 2-3 called hipCUB's): key = lo << bits | hi -- bits = what a
// label of this volume takes, found by the count pass, so the sort runs over 2 x bits instead of 64 -- with the record index
// as value; stable, so each pair's voxels stay in memory order.  The key splits into as few digits of at most 10 bits as it
// takes, all of one width (28 or 30 bits: three passes of 10; 32: four of 8).  Three kernels a pass:
//   histogram  a workgroup counts the digits of its tile of 4096 keys in LDS and writes its ROW hist[workgroup][digit] (one coalesced
//              store; a digit-major table cost every workgroup 512 scattered 4-byte writes, and the scatter as many scattered reads)
//              and adds the row to the totals of its SEGMENT of 64 workgroups;
//   offsets    where each workgroup's keys of each digit start: a thread per (segment, digit) adds the keys of smaller digits and of
//              the digit in earlier segments (totals the histogram pass has added up), then walks the 64 rows of its segment;
//   scatter    a wave owns 1024 consecutive keys of the tile and walks them 64 at a time, IN ORDER: a lane's rank among the
//              lanes of its chunk that hold the same digit comes from one ballot per digit bit, its position from the wave's
//              running cursor of that digit in LDS -- no key ever overtakes an equal one.  The LAST pass writes the records
//              themselves (pair from the key, coordinates through the index): no gather afterwards.
// HBM-bound in principle (two reads + one write of 8 bytes a record per pass); the scatter's writes are runs of equal pairs.
#include "ta_kernels.h"

namespace ta {

constexpr int RS_WAVES = 4, RS_PER_WAVE = 1024, RS_TILE = RS_WAVES * RS_PER_WAVE;      // keys per workgroup
constexpr int RS_MAX_DIGIT_BITS = 10;
constexpr uint32_t RS_COPIES = 8;                    // copies of the digit totals the histogram workgroups add to (blockIdx mod copies)
constexpr uint32_t RS_SEG = 64;                       // workgroups (rows of the digit table) a segment of the offset scan
#ifndef TA_RS_RUNS
#define TA_RS_RUNS 1       // digits are counted a RUN of equal neighbours at a time (0: a key at a time, round 4)
#endif

// the passes a key of `key_bits` takes: as few as digits of at most 10 bits allow, all of the same width
static int rs_passes(int key_bits) { return (key_bits + RS_MAX_DIGIT_BITS - 1) / RS_MAX_DIGIT_BITS; }
static int rs_digit_bits(int key_bits) { const int p = rs_passes(key_bits); return (key_bits + p - 1) / p; }

template <typename K>
__global__ void __launch_bounds__(256) wall_sort_keys_kernel(const uint2* pairs, uint64_t n, K* keys, uint32_t* index, int bits) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 p = pairs[i];
        keys[i] = (K)(((uint64_t)p.x << bits) | p.y);
        index[i] = (uint32_t)i;
    }
}

// The lanes of a chunk that hold the same digit as this lane (one ballot per digit bit).
template <int DB>
__device__ __forceinline__ uint64_t digit_peers(const uint32_t d, const bool valid) {
    uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int b = 0; b < DB; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __builtin_amdgcn_ballot_w64(valid && bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// Counting digits: the records of a wall are neighbours in memory and stay neighbours through every (stable) pass, so a wave's 64
// keys hold RUNS of equal digits -- and an LDS atomic costs the CU ~8 cycles times the lanes that share its address
// (profiles/r05_pipes.txt).  Only the first lane of a run adds, the run's length: one DPP shift, one ballot, a count of trailing zeros.
// `d` of a lane without a key must differ from every digit.  Returns the length for the first lane of a run, 0 for the others.
__device__ __forceinline__ uint32_t run_length_at_head(const uint32_t d, const int lane) {
    const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp((int)~d, (int)d, 0x138, 0xf, 0xf, false);    // wave_shr:1 (lane 0: no lane before it)
    const bool head = d != before;
    const uint64_t heads = __builtin_amdgcn_ballot_w64(head);
    const uint64_t later = (heads >> 1) >> lane;                                  // the heads behind this lane
    const uint32_t len = later ? (uint32_t)__builtin_ctzll(later) + 1u : 64u - (uint32_t)lane;
    return head ? len : 0u;
}

// (Counting the digits with one fire-and-forget LDS atomic per key puts many lanes on one address -- a wall's records are
//  neighbours in memory -- which the CU serialises; counting by one lane per group of equal digits, through digit_peers, was
//  built and is SLOWER -- C2 1.98 against 1.44 ms for the grouped fetch: three rounds of eight ballots a chunk and a
//  read-modify-write the wave has to wait for cost more than the serialised atomics nobody waits for.)
template <typename K, int DB>
__global__ void __launch_bounds__(256) radix_hist_kernel(const K* keys, uint64_t n, int shift, uint32_t* hist, uint32_t* seg_total, uint32_t* digit_total) {
    constexpr int ND = 1 << DB;
    __shared__ uint32_t h[ND];
    for (int d = threadIdx.x; d < ND; d += 256) h[d] = 0u;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int k = 0; k < RS_TILE / 256; ++k) {
        const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        const uint32_t d = i < n ? ((uint32_t)(keys[i] >> shift) & (ND - 1)) : 0xffffffffu;
#if TA_RS_RUNS
        const uint32_t len = run_length_at_head(d, (int)(threadIdx.x & 63));
        if (len && i < n) atomicAdd(&h[d], len);
#else
        if (i < n) atomicAdd(&h[d], 1u);
#endif
    }
    __syncthreads();
    // the row also goes to the totals of its segment and to one of RS_COPIES copies of the digit totals of the whole array (one copy:
    // 3655 atomics on each of 512 words, +9 us a pass on C2; the workgroup that finishes a segment last adding the segment's totals
    // needs a device-scope fence in every workgroup -- an L2 write-back each on this chip: 0.33 ms a pass)
    const uint32_t seg = blockIdx.x / RS_SEG, copy = blockIdx.x % RS_COPIES;
    for (int d = threadIdx.x; d < ND; d += 256) {
        hist[(uint64_t)blockIdx.x * ND + d] = h[d];
        if (h[d]) { atomicAdd(&seg_total[(uint64_t)seg * ND + d], h[d]); atomicAdd(&digit_total[copy * ND + d], h[d]); }      // (zeroed by the launcher)
    }
}

// offs[b][d] for the rows b of one segment: a thread per digit.  Where the digit starts in its segment = keys of smaller digits
// (an exclusive scan of the digit totals, redone by every workgroup: 2^DB words) + keys of the digit in the segments before
// (their totals, one coalesced row each), then the thread walks the rows of its segment.
// (a separate one-workgroup kernel for the segment bases took 20 us a pass on C2: one workgroup cannot keep enough loads in flight)
template <int DB>
__global__ void __launch_bounds__(256) radix_offsets_kernel(const uint32_t* hist, uint32_t nblocks, const uint32_t* seg_total,
                                                            const uint32_t* digit_total, uint32_t* offs) {
    constexpr int ND = 1 << DB;
    __shared__ uint32_t part[4], low[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t d = blockIdx.y * 256u + (uint32_t)tid, seg = blockIdx.x;
    uint32_t lower = 0u;                                           // keys of the digits below this workgroup's 256
    for (uint32_t c = 0; c < RS_COPIES; ++c)
        for (uint32_t k = (uint32_t)tid; k < blockIdx.y * 256u; k += 256u) lower += digit_total[c * ND + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lower += (uint32_t)__shfl_down((int)lower, o, 64);
    uint32_t mine = 0u;
#pragma unroll
    for (uint32_t c = 0; c < RS_COPIES; ++c) mine += d < (uint32_t)ND ? digit_total[c * ND + d] : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 0) low[w] = lower;
    if (lane == 63) part[w] = incl;
    __syncthreads();
    uint32_t run = low[0] + low[1] + low[2] + low[3] + incl - mine;
    for (int k = 0; k < w; ++k) run += part[k];
    if (d >= (uint32_t)ND) return;
#pragma unroll 8
    for (uint32_t s = 0; s < seg; ++s) run += seg_total[(uint64_t)s * ND + d];
    const uint32_t b0 = seg * RS_SEG, b1 = b0 + RS_SEG < nblocks ? b0 + RS_SEG : nblocks;
#pragma unroll 8
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t v = hist[(uint64_t)b * ND + d];
        offs[(uint64_t)b * ND + d] = run;
        run += v;
    }
}

struct __attribute__((packed, aligned(4))) WallInt3 { int32_t x, y, z; };
// the values are linear voxel indices (memory order) when n2 != 0: the last pass makes the coordinates out of them
struct WallLin { uint32_t n1, n2; int32_t inv[3]; };      // inv[i] = memory axis of array axis i

// LAST = the pass of the most significant digit: the records themselves go to their places (pair unpacked from the key, the
// coordinates fetched through the index) instead of keys and indices that a gather would have to read again.
// (Measured and dropped: the tile put in digit order in LDS first and written out a thread per local position, so that
//  neighbouring lanes write neighbouring addresses -- C2 1.27 against 1.23 ms, C3 9.5 against 9.4: the scattered 4-byte
//  stores are not what bounds a pass.  11-bit digits: 32 KB of cursors a workgroup and 2048 runs a tile, C3 12.9 against
//  9.5 ms; tiles of 8192 keys: no change.)
template <typename K, int DB, bool LAST>
__global__ void __launch_bounds__(256) radix_scatter_kernel(const K* keys_in, const uint32_t* vals_in, uint64_t n, int shift,
                                                            const uint32_t* offs, K* keys_out, uint32_t* vals_out,
                                                            const WallInt3* coords, uint2* pairs_out, WallInt3* coords_out, int bits,
                                                            const WallLin lin) {
    constexpr int ND = 1 << DB;
    __shared__ uint32_t cursor[RS_WAVES][ND];           // first the digit counts of each wave, then its running cursors
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int d = tid; d < ND; d += 256)
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) cursor[k][d] = 0u;
    __syncthreads();
    const uint64_t wbase = (uint64_t)blockIdx.x * RS_TILE + (uint64_t)w * RS_PER_WAVE;
    const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;                  // the lanes before this one
    K key[RS_PER_WAVE / 64];
    uint32_t val[RS_PER_WAVE / 64];
#pragma unroll
    for (int c = 0; c < RS_PER_WAVE / 64; ++c) {
        const uint64_t i = wbase + (uint64_t)c * 64 + lane;
        key[c] = i < n ? keys_in[i] : (K)0;
        val[c] = i < n ? vals_in[i] : 0u;
#if TA_RS_RUNS
        const uint32_t d = i < n ? ((uint32_t)(key[c] >> shift) & (ND - 1)) : 0xffffffffu;
        const uint32_t len = run_length_at_head(d, lane);
        if (len && i < n) atomicAdd(&cursor[w][d], len);
#else
        if (i < n) atomicAdd(&cursor[w][(uint32_t)(key[c] >> shift) & (ND - 1)], 1u);
#endif
    }
    __syncthreads();
    for (int d = tid; d < ND; d += 256) {   // a thread owns its digits: where the workgroup's keys of that digit start, then wave by wave
        uint32_t at = offs[(uint64_t)blockIdx.x * ND + d];
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) { const uint32_t cnt = cursor[k][d]; cursor[k][d] = at; at += cnt; }
    }
    __syncthreads();
    const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
#pragma unroll
    for (int c = 0; c < RS_PER_WAVE / 64; ++c) {
        const bool valid = wbase + (uint64_t)c * 64 + lane < n;
        const uint32_t d = (uint32_t)(key[c] >> shift) & (ND - 1);
        const uint64_t peers = digit_peers<DB>(d, valid);
        const uint32_t rank = (uint32_t)__builtin_popcountll(peers & lt);
        uint32_t pos = 0u;
        if (valid) pos = cursor[w][d] + rank;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid && rank == 0u) cursor[w][d] += (uint32_t)__builtin_popcountll(peers);      // one lane per digit of the chunk
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid) {
            if (LAST) {
                const uint64_t k = (uint64_t)key[c];
                pairs_out[pos] = make_uint2((uint32_t)(k >> bits), (uint32_t)(k & mask));
                if (lin.n2) {
                    const uint32_t row = val[c] / lin.n2;
                    const int32_t mc = (int32_t)(val[c] - row * lin.n2), ma = (int32_t)(row / lin.n1), mb = (int32_t)(row - (uint32_t)ma * lin.n1);
                    WallInt3 xyz;
                    xyz.x = lin.inv[0] == 0 ? ma : (lin.inv[0] == 1 ? mb : mc);
                    xyz.y = lin.inv[1] == 0 ? ma : (lin.inv[1] == 1 ? mb : mc);
                    xyz.z = lin.inv[2] == 0 ? ma : (lin.inv[2] == 1 ? mb : mc);
                    coords_out[pos] = xyz;
                } else {
                    coords_out[pos] = coords[val[c]];
                }
            } else {
                keys_out[pos] = key[c];
                vals_out[pos] = val[c];
            }
        }
    }
}


static uint64_t rs_blocks(uint64_t n) { return (n + RS_TILE - 1) / RS_TILE; }
static uint64_t rs_segments(uint64_t n) { return (rs_blocks(n) + RS_SEG - 1) / RS_SEG; }
// hist u32[blocks][digits] | offsets u32[blocks][digits] | per pass (<= 8): segment totals [segments][digits], digit totals [copies][digits]
uint64_t wall_sort_temp_bytes(uint64_t n) {
    const uint64_t nd = 1u << RS_MAX_DIGIT_BITS, cells = nd * rs_blocks(n);
    return 2 * ((cells * 4 + 15) & ~15ull) + 8 * (rs_segments(n) + RS_COPIES) * nd * sizeof(uint32_t) + 64;
}

template <typename K, int DB>
static hipError_t wall_group_db(hipStream_t s, const uint32_t* pairs, const int32_t* coords, uint64_t n, K* keys0, K* keys1,
                                uint32_t* index0, uint32_t* index1, void* temp, int bits, uint32_t* pairs_out, int32_t* coords_out,
                                const WallLin lin) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (pairs) hipLaunchKernelGGL(wall_sort_keys_kernel<K>, dim3(blocks), dim3(256), 0, s, (const uint2*)pairs, n, keys0, index0, bits);
    const uint32_t nb = (uint32_t)rs_blocks(n);
    const uint64_t cells = (uint64_t)(1 << DB) * nb;
    char* p = (char*)temp;
    uint32_t* hist = (uint32_t*)p; p += (cells * 4 + 15) & ~15ull;
    uint32_t* offs = (uint32_t*)p; p += (cells * 4 + 15) & ~15ull;
    const uint32_t nseg = (uint32_t)rs_segments(n);
    const size_t seg_words = (size_t)nseg * (1 << DB);
    K* kin = keys0; K* kout = keys1;
    uint32_t* vin = index0; uint32_t* vout = index1;
    const int passes = rs_passes(2 * bits);
    // per pass: segment totals [segments][digits] | digit totals [copies][digits]
    uint32_t* totals = (uint32_t*)p;
    const size_t pass_words = seg_words + (size_t)RS_COPIES * (1 << DB);
    const hipError_t ez = hipMemsetAsync(totals, 0, (size_t)passes * pass_words * sizeof(uint32_t), s);
    if (ez != hipSuccess) return ez;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = pass * DB;
        uint32_t* seg_total = totals + (size_t)pass * pass_words;
        hipLaunchKernelGGL((radix_hist_kernel<K, DB>), dim3(nb), dim3(256), 0, s, kin, n, shift, hist, seg_total, seg_total + seg_words);
        hipLaunchKernelGGL((radix_offsets_kernel<DB>), dim3(nseg, ((1 << DB) + 255) / 256), dim3(256), 0, s, hist, nb, seg_total, seg_total + seg_words, offs);
        if (pass + 1 < passes)
            hipLaunchKernelGGL((radix_scatter_kernel<K, DB, false>), dim3(nb), dim3(256), 0, s, kin, vin, n, shift, offs, kout, vout,
                               (const WallInt3*)nullptr, (uint2*)nullptr, (WallInt3*)nullptr, bits, lin);
        else
            hipLaunchKernelGGL((radix_scatter_kernel<K, DB, true>), dim3(nb), dim3(256), 0, s, kin, vin, n, shift, offs, kout, vout,
                               (const WallInt3*)coords, (uint2*)pairs_out, (WallInt3*)coords_out, bits, lin);
        K* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    return hipGetLastError();
}

template <typename K>
static hipError_t wall_group(hipStream_t s, const uint32_t* pairs, const int32_t* coords, uint64_t n, K* keys0, K* keys1,
                             uint32_t* index0, uint32_t* index1, void* temp, int bits, uint32_t* pairs_out, int32_t* coords_out,
                             const WallLin lin) {
    switch (rs_digit_bits(2 * bits)) {      // (2 .. 10: the widths a key of 2 .. 64 bits splits into; below 8: as 8)
#define TA_RS_CASE(DB) case DB: return wall_group_db<K, DB>(s, pairs, coords, n, keys0, keys1, index0, index1, temp, bits, pairs_out, coords_out, lin);
        TA_RS_CASE(10) TA_RS_CASE(9) TA_RS_CASE(8)
#undef TA_RS_CASE
        default: return wall_group_db<K, 8>(s, pairs, coords, n, keys0, keys1, index0, index1, temp, bits, pairs_out, coords_out, lin);
    }
}

// pairs / coords: the records in memory order (n < 2^32).  keys[2], index[2]: double buffers of n entries (keys: 8 bytes an
// entry, used as 4-byte ones where two labels fit 32 bits); temp: wall_sort_temp_bytes; label_bits: the bits a label takes
// (1 .. 32).  Writes the grouped records to pairs_out / coords_out.  Only enqueues work.
hipError_t launch_wall_group_by_pair(hipStream_t s, const uint32_t* pairs, const int32_t* coords, uint64_t n, uint64_t* keys0,
                                     uint64_t* keys1, uint32_t* index0, uint32_t* index1, void* temp, uint64_t temp_bytes,
                                     int label_bits, uint32_t* pairs_out, int32_t* coords_out) {
    (void)temp_bytes;
    if (n == 0) return hipSuccess;
    if (label_bits < 1) label_bits = 1;
    if (label_bits > 32) label_bits = 32;
    const WallLin none = {0u, 0u, {0, 1, 2}};
    if (2 * label_bits <= 32)
        return wall_group<uint32_t>(s, pairs, coords, n, (uint32_t*)keys0, (uint32_t*)keys1, index0, index1, temp, label_bits,
                                    pairs_out, coords_out, none);
    return wall_group<uint64_t>(s, pairs, coords, n, keys0, keys1, index0, index1, temp, label_bits, pairs_out, coords_out, none);
}

hipError_t launch_wall_group_keyed(hipStream_t s, uint64_t n, uint64_t* keys0, uint64_t* keys1, uint32_t* index0, uint32_t* index1,
                                   void* temp, int label_bits, const int64_t mdims[3], const int perm[3], uint32_t* pairs_out,
                                   int32_t* coords_out) {
    if (n == 0) return hipSuccess;
    if (label_bits < 1) label_bits = 1;
    if (label_bits > 32) label_bits = 32;
    WallLin lin;
    lin.n1 = (uint32_t)mdims[1]; lin.n2 = (uint32_t)mdims[2];
    for (int k = 0; k < 3; ++k) lin.inv[perm[k]] = k;
    if (2 * label_bits <= 32)
        return wall_group<uint32_t>(s, nullptr, nullptr, n, (uint32_t*)keys0, (uint32_t*)keys1, index0, index1, temp, label_bits,
                                    pairs_out, coords_out, lin);
    return wall_group<uint64_t>(s, nullptr, nullptr, n, keys0, keys1, index0, index1, temp, label_bits, pairs_out, coords_out, lin);
}

}  // namespace ta
