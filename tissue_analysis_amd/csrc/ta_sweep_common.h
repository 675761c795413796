// ta_sweep_common.h -- pieces of the sweep kernel (kernels_scan.hip) that are not its hot loop:
// tuning constants, the DPP lane shift, tile-local sum records, the global spill paths, the private
// hot-label rows and the flush of the workgroup LDS tables (label -> packed tile-local moments + bbox,
// pair -> per-axis face counts) with global atomics.
#pragma once
#include "ta_kernels.h"
#include "ta_sweep_switches.h"

namespace ta {

constexpr int WAVES = TA_WAVES;   // PRODUCER waves per workgroup, stacked along axis 1 (a kernel with a consumer wave has one more)
constexpr int LSLOTS = TA_LSLOTS; // label table slots per workgroup
constexpr int PSLOTS = TA_PSLOTS; // pair table slots per workgroup
constexpr int ilog2_c(int v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }
constexpr int LSLOTS_LOG2 = ilog2_c(LSLOTS), PSLOTS_LOG2 = ilog2_c(PSLOTS);
static_assert((1 << LSLOTS_LOG2) == LSLOTS && (1 << PSLOTS_LOG2) == PSLOTS, "table sizes must be powers of two");
constexpr int LPROBE = 16;        // max probes before spilling to global atomics
constexpr int PPROBE = 32;
constexpr int MAX_TILE_PLANES = 64;
constexpr uint32_t LABEL_LIMIT = 1u << 28;    // max_label < 2^28: the two top bits of a record word are free
// TA_PCNT64 (measured, not adopted): a pair's three per-axis face counts of ONE tile in a u64 LDS word, 21 bits each (a tile
// holds at most 16 rows x 512 columns x 64 planes = 2^19 voxels, so a field cannot carry into the next): 8 bytes per slot
// instead of 12, but every LDS atomic on it costs twice the cycles of a u32 one
constexpr int PCNT_BITS = 21;
constexpr uint64_t PCNT_MASK = (1ull << PCNT_BITS) - 1ull;

__device__ __forceinline__ uint32_t lane_shr1(uint32_t src, uint32_t lane0_value) {
    // lane i <- src of lane i-1 ; lane 0 keeps lane0_value   (DPP wave_shr:1)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0_value, (int)src, 0x138, 0xf, 0xf, false);
}

// one voxel at a wave-uniform address through the scalar cache (SMEM): no VGPR, no vector-memory slot
template <typename T>
__device__ __forceinline__ uint32_t load_uniform_voxel(const T* p) {
    typedef const __attribute__((address_space(4))) uint32_t* cptr;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t word = *reinterpret_cast<cptr>(a & ~(uintptr_t)3);
    if (sizeof(T) == 4) return word;
    return (a & 2) ? (word >> 16) : (word & 0xffffu);
}

// local sums of one label contribution (tile-local coordinates)
struct LocalSums { uint64_t n, sa, sb, sc, saa, sab, sac, sbb, sbc, scc; };
// one run's contribution: every term fits 32 bits (n <= 64, a < 64, b < 16, c < 512)
struct RunSums { uint32_t n, sa, sb, sc, saa, sab, sac, sbb, sbc, scc; };

// ---- the ten tile-local sums of a label slot, PACKED into four u64 words (two without second moments) --------------------
// An LDS atomic costs the CU's one LDS pipe ~15 cycles (u64; ~7.7 for u32) times the number of lanes of the instruction that
// share its address (scripts/pipes_bench.hip, profiles/r05_pipes.txt), and a group of 64 run records holds each label about
// 3.5 times: the six u64 adds of a run were the largest single item of LDS time in tissue.  The bounds of a tile are known at
// compile time -- P planes, B rows, C columns, every coordinate tile-local -- so every sum gets exactly the bits its largest
// possible value takes, and the ten fit four words:   w0 = cc | bc,   w1 = ac | aa,   w2 = c | a | n,   w3 = ab | bb | b.
// (All contributions are non-negative and the totals of a tile stay inside their fields: no field ever carries into the next.)
template <int P, int B, int C>
struct SumPack {
    static constexpr uint64_t tri(uint64_t n) { return n * (n - 1) / 2; }                      // sum of i, i < n
    static constexpr uint64_t sq(uint64_t n) { return (n - 1) * n * (2 * n - 1) / 6; }         // sum of i^2, i < n
    static constexpr int bits(uint64_t x) { int b = 0; while (x) { ++b; x >>= 1; } return b; }
    static constexpr int bN = bits((uint64_t)P * B * C);
    static constexpr int bA = bits((uint64_t)B * C * tri(P)), bB = bits((uint64_t)P * C * tri(B)), bC = bits((uint64_t)P * B * tri(C));
    static constexpr int bAA = bits((uint64_t)B * C * sq(P)), bAB = bits((uint64_t)C * tri(P) * tri(B)), bAC = bits((uint64_t)B * tri(P) * tri(C));
    static constexpr int bBB = bits((uint64_t)P * C * sq(B)), bBC = bits((uint64_t)P * tri(B) * tri(C)), bCC = bits((uint64_t)P * B * sq(C));
    static_assert(bCC + bBC <= 64 && bAC + bAA <= 64 && bC + bA + bN <= 64 && bAB + bBB + bB <= 64,
                  "the tile is too large for four packed words: lower the kernel's cap on the tile height (tile_planes_cap)");
    static constexpr int planes = P;
    static constexpr uint64_t mask(int b) { return b >= 64 ? ~0ull : ((1ull << b) - 1ull); }
    // MOM2: four words; else two: w0 = n | b << 32, w1 = a | c << 32 (every sum of a tile is below 2^32 there)
    template <bool MOM2, typename SUMS>
    static __device__ __forceinline__ void pack(const SUMS& L, uint64_t (&w)[4]) {
        if (MOM2) {
            w[0] = (uint64_t)L.scc + ((uint64_t)L.sbc << bCC);
            w[1] = (uint64_t)L.sac + ((uint64_t)L.saa << bAC);
            w[2] = (uint64_t)L.sc + ((uint64_t)L.sa << bC) + ((uint64_t)L.n << (bC + bA));
            w[3] = (uint64_t)L.sab + ((uint64_t)L.sbb << bAB) + ((uint64_t)L.sb << (bAB + bBB));
        } else {
            w[0] = (uint64_t)L.n | ((uint64_t)L.sb << 32);
            w[1] = (uint64_t)L.sa | ((uint64_t)L.sc << 32);
            w[2] = w[3] = 0ull;
        }
    }
    // adds the words of one slot (one replica) to L
    template <bool MOM2>
    static __device__ __forceinline__ void unpack_add(const uint64_t* w, LocalSums& L) {
        if (MOM2) {
            L.scc += w[0] & mask(bCC); L.sbc += w[0] >> bCC;
            L.sac += w[1] & mask(bAC); L.saa += w[1] >> bAC;
            L.sc += w[2] & mask(bC); L.sa += (w[2] >> bC) & mask(bA); L.n += w[2] >> (bC + bA);
            L.sab += w[3] & mask(bAB); L.sbb += (w[3] >> bAB) & mask(bBB); L.sb += w[3] >> (bAB + bBB);
        } else {
            L.n += w[0] & 0xffffffffull; L.sb += w[0] >> 32; L.sa += w[1] & 0xffffffffull; L.sc += w[1] >> 32;
        }
    }
};
// the tallest tile each kind of sweep kernel packs its sums for (launches clamp the tile height to it; results never depend on it)
// (moments only: uint16 tiles are 8 rows x 512 columns like the wide adjacency tiles; uint32 tiles 16 rows x 256, whose four words end at 16 planes)
constexpr int tile_planes_cap(bool adjacency, int itemsize, int vpl) { return adjacency ? (vpl == 4 ? 48 : TA_PLANES_CAP_ADJ8) : (itemsize == 2 ? 32 : 16); }

// shift tile-local sums to global coordinates (origin A0,B0,C0): exact u64
__device__ __forceinline__ void local_to_global(const LocalSums& L, uint64_t A0, uint64_t B0, uint64_t C0,
                                                uint64_t (&g)[NSUM]) {
    g[0] = L.n;
    g[1] = L.sa + L.n * A0; g[2] = L.sb + L.n * B0; g[3] = L.sc + L.n * C0;
    g[4] = L.saa + 2 * A0 * L.sa + L.n * A0 * A0;
    g[5] = L.sab + A0 * L.sb + B0 * L.sa + L.n * A0 * B0;
    g[6] = L.sac + A0 * L.sc + C0 * L.sa + L.n * A0 * C0;
    g[7] = L.sbb + 2 * B0 * L.sb + L.n * B0 * B0;
    g[8] = L.sbc + B0 * L.sc + C0 * L.sb + L.n * B0 * C0;
    g[9] = L.scc + 2 * C0 * L.sc + L.n * C0 * C0;
}

// ---- rare spill paths, kept out of line so they do not bloat the hot loops ------------------
static __device__ __noinline__ void label_spill_global(uint64_t* sums, int32_t* boxes, uint32_t* flags,
                                                uint32_t max_label, uint32_t label, const LocalSums* L,
                                                uint64_t A0, uint64_t B0, uint64_t C0, const uint32_t* box) {
    atomicAdd(&flags[FLAG_LDS_LABEL_SPILL], 1u);
    if (label > max_label) { atomicOr(&flags[FLAG_RANGE], 1u); return; }
    uint64_t g[NSUM];
    local_to_global(*L, A0, B0, C0, g);
    unsigned long long* row = (unsigned long long*)&sums[(uint64_t)label * NSUM];
    for (int k = 0; k < NSUM; ++k) if (g[k]) atomicAdd(row + k, (unsigned long long)g[k]);
    int32_t* gb = &boxes[(uint64_t)label * NBOX];
    atomicMin(gb + 0, (int32_t)(A0 + box[0])); atomicMin(gb + 3, -(int32_t)(A0 + box[3]));
    atomicMin(gb + 1, (int32_t)(B0 + box[1])); atomicMin(gb + 4, -(int32_t)(B0 + box[4]));
    atomicMin(gb + 2, (int32_t)(C0 + box[2])); atomicMin(gb + 5, -(int32_t)(C0 + box[5]));
}

static __device__ __noinline__ void pair_spill_global(PairTable pt, uint32_t* flags, uint32_t lo, uint32_t hi,
                                               uint32_t axis, uint32_t count) {
    atomicAdd(&flags[FLAG_LDS_PAIR_SPILL], 1u);
    pair_add_global(pt, lo, hi, axis == 0 ? count : 0, axis == 1 ? count : 0, axis == 2 ? count : 0, flags);
}

// sum_{x=x0}^{x0+n-1} x  and  x^2  (exact, u64)
__device__ __forceinline__ uint64_t range_sum1(uint64_t x0, uint64_t n) { return n * x0 + n * (n - 1) / 2; }
__device__ __forceinline__ uint64_t range_sum2(uint64_t x0, uint64_t n) {
    return n * x0 * x0 + x0 * n * (n - 1) + (n - 1) * n * (2 * n - 1) / 6;
}

// ---- flush the workgroup tables with global atomics (local -> global coordinates here) -------
// One label -- the background -- is in almost every tile of a tissue-in-a-box volume: thousands of
// workgroups flushing it means tens of thousands of atomics on ONE row of the global accumulators, which
// serialise in a single L2 channel (it is what made small tiles slow: C2 at 8-plane tiles 0.30 -> 0.12 ms).
// The label of the slab's first voxel is taken as that hot label; every workgroup owns a private row for
// it (plain stores, identity written in the prologue) and hot_reduce_kernel folds the rows afterwards.
template <typename T>
__device__ __forceinline__ uint32_t hot_label_of(const SweepArgs& A) {
    return load_uniform_voxel<T>(reinterpret_cast<const T*>(A.vol) + (int64_t)A.first_owned * A.n1 * A.n2);
}
// row layout (HOTW u64 words = 128 bytes): sums u64[NSUM] | boxes i32[NBOX] | padding
__device__ __forceinline__ uint64_t* hot_rows_of(const SweepArgs& A) {
    typedef uint64_t* const __attribute__((address_space(4)))* cpp;          // scalar (SMEM) load of the parked pointer
    return *reinterpret_cast<cpp>(reinterpret_cast<uintptr_t>(A.flags + HOT_PTR_WORD));
}
__device__ __forceinline__ void hot_row_init(const SweepArgs& A, const int tid, const uint32_t wg) {
    uint64_t* rows = hot_rows_of(A);
    if (rows && tid < HOTW) {
        const uint64_t imax2 = ((uint64_t)(uint32_t)INT32_MAX << 32) | (uint32_t)INT32_MAX;
        rows[(uint64_t)wg * HOTW + tid] = tid < NSUM ? 0ull : imax2;
    }
}
__device__ __forceinline__ void hot_row_init(const SweepArgs& A, const int tid) { hot_row_init(A, tid, blockIdx.x); }

// HOT = false compiles the hot-row path out (the naive cross-check kernel's rows; the sweep uses it for every mask).
// RESET: every slot that held something is emptied again as it is read (the persistent kernel goes on with the next tile).
//
// DENSE flush.  A CU issues a global atomic wave-instruction about every 50 ns whatever the number of lanes that take part
// (MI355X_MICROARCH.md), and a tile's tables are mostly empty -- ~38 of 128 label slots, ~120 of 512 pair slots on C4: walked
// slot by slot, a flush issued ~60 such instructions a tile, most of them with a quarter of the lanes alive.  So the occupied
// slots are first GATHERED into two lists (a ballot and an LDS counter per wave; the lists live in the record buffers of
// waves 0 and 1, which nobody needs any more), and then the first K threads flush the K labels -- and the LAST K' threads
// the K' pairs, so that the two kinds of global traffic start on different waves -- with every lane of an instruction alive.
template <int NW, bool ADJ, bool MOM2, bool HOT, typename LDS, int NT = WAVES * 64, bool RESET = false>
__device__ __forceinline__ void flush_tables(const SweepArgs& A, LDS& S, const int tid, const uint64_t A0,
                                             const uint64_t B0, const uint64_t C0, const uint32_t hot,
                                             const uint32_t wg) {
    // (the record buffers are dead by now and lie back to back: the label list starts in wave 0's run arrays and must end
    //  before wave 1's, the pair list starts in wave 1's and may run on into the buffers of the waves behind it)
    static_assert(LSLOTS * 2 <= (int)sizeof(S.wave[0].cql) + (int)sizeof(S.wave[0].cqc) &&
                  PSLOTS * 2 <= (int)sizeof(S.wave[1].cql) + (int)sizeof(S.wave[1].cqc) + (WAVES - 2) * (int)sizeof(S.wave[0]),
                  "the slot lists of the flush live in the record buffers of the waves");
    uint16_t* const llist = reinterpret_cast<uint16_t*>(&S.wave[0].cql[0]);
    uint16_t* const plist = reinterpret_cast<uint16_t*>(&S.wave[1].cql[0]);
    const int lane = tid & 63;
    if (tid < 2) S.fcnt[tid] = 0u;
    __syncthreads();
    for (int i0 = 0; i0 < LSLOTS; i0 += NT) {
        const int i = i0 + tid;
        const bool live = i < LSLOTS && S.lkeys[i] != INVALID_LABEL;
        const uint64_t m = __builtin_amdgcn_ballot_w64(live);
        uint32_t base = 0u;
        if (lane == 0 && m) base = atomicAdd(&S.fcnt[0], (uint32_t)__builtin_popcountll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (live) llist[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)i;
    }
    if (ADJ) {
        for (int i0 = 0; i0 < PSLOTS; i0 += NT) {
            const int i = i0 + tid;
            const bool live = i < PSLOTS && S.pkeys[i] != EMPTY_KEY;
            const uint64_t m = __builtin_amdgcn_ballot_w64(live);
            uint32_t base = 0u;
            if (lane == 0 && m) base = atomicAdd(&S.fcnt[1], (uint32_t)__builtin_popcountll(m));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (live) plist[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)i;
        }
    }
    __syncthreads();
    const int nl = (int)S.fcnt[0], np = ADJ ? (int)S.fcnt[1] : 0;

    // TRANSPOSED label flush (TA_FLUSH_TRANSPOSE): a label's NSUM sums are consecutive u64 words of one global row, and a global atomic
    // is a request to the memory side per cache line a wave instruction touches.  Lane = label sends every add of an instruction to a
    // different row (64 requests, ten instructions a label); lane = (label, word) sends 64 consecutive words of ~6 rows.  The sums go
    // through LDS for that: the pair table's keys and counts are read into registers first (two pairs a thread at most; their home
    // slots' global reads are in flight from then on), the sums are staged where the pair table was, and the pairs' adds go last.
    constexpr int NS = MOM2 ? NSUM : 4;
    constexpr bool TR = ADJ && !RESET && TA_FLUSH_TRANSPOSE != 0 && !TA_PCNT64 && PSLOTS <= 2 * NT &&
                        (size_t)LSLOTS * NSUM * 8 <= sizeof(S.pkeys) + sizeof(S.pcnt);
    if constexpr (TR) {
        static_assert(!TR || offsetof(LDS, pcnt) == offsetof(LDS, pkeys) + sizeof(S.pkeys), "the staged sums run from the pair keys on into the pair counts");
        uint64_t pkey[2], gkey[2]; uint32_t pf[2][3], ghome[2]; bool plive[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = NT - 1 - tid + q * NT;
            plive[q] = j < np;
            pkey[q] = gkey[q] = 0ull; ghome[q] = 0u; pf[q][0] = pf[q][1] = pf[q][2] = 0u;
#ifndef TA_ABL_NOFLUSH_PAIRS
            if (plive[q]) {
                const int i = plist[j];
                pkey[q] = S.pkeys[i];
                ghome[q] = hash_pair((uint32_t)(pkey[q] >> 32), (uint32_t)pkey[q]) & A.pairs.mask;
                gkey[q] = __hip_atomic_load(&A.pairs.keys[ghome[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pf[q][0] = S.pcnt[i * 3 + 0]; pf[q][1] = S.pcnt[i * 3 + 1]; pf[q][2] = S.pcnt[i * 3 + 2];
            }
#endif
        }
        __syncthreads();                                   // (every pair is in registers: the table's space is free)
        uint64_t* const stage = reinterpret_cast<uint64_t*>(&S.pkeys[0]);
        uint64_t* const hot_rows = HOT ? hot_rows_of(A) : nullptr;
        uint64_t* const hr = hot_rows + (uint64_t)wg * HOTW;
#ifndef TA_ABL_NOFLUSH_LABELS
        for (int j = tid; j < nl; j += NT) {
            const int i = llist[j];
            const uint32_t label = S.lkeys[i];
            LocalSums L;
            L.n = L.sa = L.sb = L.sc = L.saa = L.sab = L.sac = L.sbb = L.sbc = L.scc = 0;
#pragma unroll
            for (int rp = 0; rp < LDS::REP; ++rp) {
                uint64_t w[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
                for (int k = 0; k < NW; ++k) w[k] = S.lsum[(i * LDS::REP + rp) * NW + k];
                LDS::Pack::template unpack_add<MOM2>(w, L);
            }
            uint64_t g[NSUM];
            local_to_global(L, A0, B0, C0, g);
            const bool ok = label <= A.max_label;
            if (!ok) atomicOr(&A.flags[FLAG_RANGE], 1u);
#pragma unroll
            for (int k = 0; k < NS; ++k) stage[j * NS + k] = ok ? g[k] : 0ull;
            if (ok) {
                const bool priv = HOT && hot_rows && label == hot;
                int32_t* box = priv ? reinterpret_cast<int32_t*>(hr + NSUM) : &A.boxes[(uint64_t)label * NBOX];
                const int32_t m0 = (int32_t)(A0 + S.lbox[i * 8 + 0]), m1 = (int32_t)(B0 + S.lbox[i * 8 + 1]), m2 = (int32_t)(C0 + S.lbox[i * 8 + 2]);
                const int32_t m3 = -(int32_t)(A0 + S.lbox[i * 8 + 3]), m4 = -(int32_t)(B0 + S.lbox[i * 8 + 4]), m5 = -(int32_t)(C0 + S.lbox[i * 8 + 5]);
                const int2 g01 = *reinterpret_cast<const int2*>(box), g23 = *reinterpret_cast<const int2*>(box + 2), g45 = *reinterpret_cast<const int2*>(box + 4);
                if (m0 < g01.x) flush_min(box + 0, m0);
                if (m1 < g01.y) flush_min(box + 1, m1);
                if (m2 < g23.x) flush_min(box + 2, m2);
                if (m3 < g23.y) flush_min(box + 3, m3);
                if (m4 < g45.x) flush_min(box + 4, m4);
                if (m5 < g45.y) flush_min(box + 5, m5);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < nl * NS; idx += NT) {
            const int j = idx / NS, k = idx - j * NS;
            const uint32_t label = S.lkeys[llist[j]];
            const uint64_t v = stage[idx];
            if (v) {
                const bool priv = HOT && hot_rows && label == hot;
                unsigned long long* row = (unsigned long long*)(priv ? hr : &A.sums[(uint64_t)label * NSUM]);
                flush_add(row + k, (unsigned long long)v);
            }
        }
#endif
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (plive[q])
                pair_add_global_from(A.pairs, (uint32_t)(pkey[q] >> 32), (uint32_t)pkey[q], pf[q][0], pf[q][1], pf[q][2], A.flags, ghome[q], gkey[q]);
    } else {
    // -- the pairs, from the last thread down: the home slot of the pair in the device-global table is READ first and looked
    //    at afterwards (the first global round trip of the probe is in flight while the rest is prepared)
#ifdef TA_ABL_NOFLUSH_PAIRS
    if (false) {
#else
    if (ADJ) {
#endif
        for (int j = NT - 1 - tid; j < np; j += NT) {
            const int i = plist[j];
            const uint64_t key = S.pkeys[i];
            const uint32_t lo = (uint32_t)(key >> 32), hi = (uint32_t)key;
            const uint32_t gh = hash_pair(lo, hi) & A.pairs.mask;
            const uint64_t gk = __hip_atomic_load(&A.pairs.keys[gh], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if TA_PCNT64
            const uint64_t c = S.pcnt[i];
            const uint32_t f0 = (uint32_t)(c & PCNT_MASK), f1 = (uint32_t)((c >> PCNT_BITS) & PCNT_MASK), f2 = (uint32_t)(c >> (2 * PCNT_BITS));
            if (RESET) S.pcnt[i] = 0ull;
#else
            const uint32_t f0 = S.pcnt[i * 3 + 0], f1 = S.pcnt[i * 3 + 1], f2 = S.pcnt[i * 3 + 2];
            if (RESET) { S.pcnt[i * 3 + 0] = 0u; S.pcnt[i * 3 + 1] = 0u; S.pcnt[i * 3 + 2] = 0u; }
#endif
            if (RESET) S.pkeys[i] = EMPTY_KEY;
            pair_add_global_from(A.pairs, lo, hi, f0, f1, f2, A.flags, gh, gk);
        }
    }
    // -- the labels, from the first thread up
    uint64_t* const hot_rows = HOT ? hot_rows_of(A) : nullptr;
#ifdef TA_ABL_NOFLUSH_LABELS
    for (int j = tid; j < 0; j += NT) {
#else
    for (int j = tid; j < nl; j += NT) {
#endif
        const int i = llist[j];
        const uint32_t label = S.lkeys[i];
        LocalSums L;
        L.n = L.sa = L.sb = L.sc = L.saa = L.sab = L.sac = L.sbb = L.sbc = L.scc = 0;
#pragma unroll
        for (int rp = 0; rp < LDS::REP; ++rp) {            // (the replicas a label's runs were spread over: see drain_run_group)
            uint64_t w[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
            for (int k = 0; k < NW; ++k) w[k] = S.lsum[(i * LDS::REP + rp) * NW + k];
            LDS::Pack::template unpack_add<MOM2>(w, L);
        }
        const uint32_t bx0 = S.lbox[i * 8 + 0], bx1 = S.lbox[i * 8 + 1], bx2 = S.lbox[i * 8 + 2];
        const uint32_t bx3 = S.lbox[i * 8 + 3], bx4 = S.lbox[i * 8 + 4], bx5 = S.lbox[i * 8 + 5];
        if (RESET) {
            S.lkeys[i] = INVALID_LABEL;
#pragma unroll
            for (int k = 0; k < NW * LDS::REP; ++k) S.lsum[i * NW * LDS::REP + k] = 0ull;
            S.lbox[i * 8 + 0] = 0xFFFFFFFFu; S.lbox[i * 8 + 1] = 0xFFFFFFFFu; S.lbox[i * 8 + 2] = 0xFFFFFFFFu;
            S.lbox[i * 8 + 3] = 0u; S.lbox[i * 8 + 4] = 0u; S.lbox[i * 8 + 5] = 0u;
        }
        if (label > A.max_label) { atomicOr(&A.flags[FLAG_RANGE], 1u); continue; }
        uint64_t g[NSUM];
        local_to_global(L, A0, B0, C0, g);
        // the hot label goes to this workgroup's private row: same atomics, nobody to contend with
        const bool priv = HOT && hot_rows && label == hot;
        uint64_t* hr = hot_rows + (uint64_t)wg * HOTW;
        unsigned long long* row = (unsigned long long*)(priv ? hr : &A.sums[(uint64_t)label * NSUM]);
#pragma unroll
        for (int k = 0; k < (MOM2 ? NSUM : 4); ++k) flush_add(row + k, (unsigned long long)g[k]);
        int32_t* box = priv ? reinterpret_cast<int32_t*>(hr + NSUM) : &A.boxes[(uint64_t)label * NBOX];
        // The global box is READ first (plain loads) and only the bounds this tile extends are sent: a cell meets ~10 tiles and its
        // box stops moving after the outermost ones.  A stale value out of a cache is an OLDER one -- boxes only ever shrink towards
        // their minima -- so a skip decided on it is always safe; the price is one more round trip in a flush that waits for several.
        const int32_t m0 = (int32_t)(A0 + bx0), m1 = (int32_t)(B0 + bx1), m2 = (int32_t)(C0 + bx2);
        const int32_t m3 = -(int32_t)(A0 + bx3), m4 = -(int32_t)(B0 + bx4), m5 = -(int32_t)(C0 + bx5);
#if TA_FLUSH_BOX_READ
        const int2 g01 = *reinterpret_cast<const int2*>(box), g23 = *reinterpret_cast<const int2*>(box + 2), g45 = *reinterpret_cast<const int2*>(box + 4);
        if (m0 < g01.x) flush_min(box + 0, m0);
        if (m1 < g01.y) flush_min(box + 1, m1);
        if (m2 < g23.x) flush_min(box + 2, m2);
        if (m3 < g23.y) flush_min(box + 3, m3);
        if (m4 < g45.x) flush_min(box + 4, m4);
        if (m5 < g45.y) flush_min(box + 5, m5);
#else
        flush_min(box + 0, m0); flush_min(box + 3, m3);
        flush_min(box + 1, m1); flush_min(box + 4, m4);
        flush_min(box + 2, m2); flush_min(box + 5, m5);
#endif
    }
    }
}

template <int NW, bool ADJ, bool MOM2, bool HOT, typename LDS>
__device__ __forceinline__ void flush_tables(const SweepArgs& A, LDS& S, const int tid, const uint64_t A0,
                                             const uint64_t B0, const uint64_t C0, const uint32_t hot) {
    flush_tables<NW, ADJ, MOM2, HOT, LDS>(A, S, tid, A0, B0, C0, hot, blockIdx.x);
}

}  // namespace ta
