// kernels_census.hip -- which label ids does the volume hold, and the volume rewritten in their RANKS (sparse label ids).
//
// The reference takes any ids (np.unique(image), SIA:358-364); the sweep keeps one 104-byte row per id 0..max_label, which a
// volume with a few thousand cells numbered near 2^31 cannot afford.  The census is the device form of np.unique:
//   census[w] = { bits, below }   bits  = bit b set <=> id 32 w + b occurs in the volume (one atomicOr per FIRST sighting:
//                                        the word is read first, and a thread skips a voxel equal to the one before it)
//                                 below = number of ids present in words 0 .. w-1 (three-kernel exclusive scan of popcounts)
// so rank(v) = census[v >> 5].below + popc(census[v >> 5].bits & ((1 << (v & 31)) - 1)) is ONE 8-byte gather per voxel, the
// ranks are dense 0 .. n-1 and ORDER-PRESERVING (everything the sweep orders by label -- lo < hi, the sorted pair list --
// carries over), and ids[rank] (expanded from the same words) maps the rows back.  All of it is HBM-bound streaming over the
// volume plus a table of (max_label + 1) / 4 bytes.
#include "ta_kernels.h"

namespace ta {

namespace {

constexpr int CENSUS_WORDS_PER_BLOCK = 4096;     // 256 threads x 16 words

// `touched[b]` = some id of block b (4096 words = 2^17 ids) is present: the scan skips the blocks nobody touched -- a handful of
// cells numbered near 2^32 leave 16384 blocks of table, of which the scan then reads a few
__device__ __forceinline__ void census_mark(uint2* census, uint8_t* touched, uint32_t v) {
    uint32_t* w = &census[v >> 5].x;
    const uint32_t bit = 1u << (v & 31u);
    if (!(*w & bit)) {               // (a plain, cached read: a stale line only costs a redundant atomic)
        atomicOr(w, bit);
        touched[v >> 17] = 1;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) census_mark_kernel(const T* __restrict__ vol, uint64_t n, uint64_t nvec, uint2* census, uint8_t* touched) {
    constexpr int PER = 16 / (int)sizeof(T);
    const uint4* v4 = reinterpret_cast<const uint4*>(vol);
    uint32_t prev = 0xffffffffu;
    bool have = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 x = v4[i];
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (sizeof(T) == 4) {
                if (!have || w[k] != prev) { census_mark(census, touched, w[k]); prev = w[k]; have = true; }
            } else {
                const uint32_t a = w[k] & 0xffffu, b = w[k] >> 16;
                if (!have || a != prev) { census_mark(census, touched, a); prev = a; have = true; }
                if (b != prev) { census_mark(census, touched, b); prev = b; }
            }
        }
    }
    for (uint64_t t = nvec * PER + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        census_mark(census, touched, (uint32_t)vol[t]);
}

// The same over a volume whose rows are whole 16-byte vectors: a workgroup walks DOWN the rows of a 256-vector strip, so a
// thread's next vector is the same columns one row further -- the same cells nine times out of ten.  What the flat kernel above
// spends its time on is not bytes but the table: a wave meets SOME new label in nearly every row, and its read of the table
// and, worse, its atomic on it keep the wave from the rows it has in flight (3.5 ms on 1024^3 against 0.83 ms for the pass that
// only takes the maximum).  Here a lane compares its labels with the SAME positions of the row above (and with the voxel to its
// left inside the vector) and puts a new one into an LDS set of the workgroup; the global table is written once per label of
// the workgroup, after its last row: 0.77 ms.
constexpr int CENSUS_ROWS_PER_BLOCK = 64;
constexpr int CENSUS_SEEN_LOG2 = 11, CENSUS_SEEN = 1 << CENSUS_SEEN_LOG2;
constexpr int CENSUS_LIST_PARTS = 256;           // a label list is kept in parts, each with its own cursor (one word that 16 k workgroups
                                                 // add to is a queue: returning atomics on one address take ~0.4 us each)
constexpr int CENSUS_LIST_HEAD = 2 * CENSUS_LIST_PARTS;      // words before the entries: { asked for, maximum } per part

//
// LIST (one pass over a volume whose maximum is not known yet, so the table cannot be sized): the labels leave the workgroup for a
// LIST in memory instead of the table, kept in CENSUS_LIST_PARTS parts of `cap` entries (a workgroup writes to part blockIdx mod
// parts): `list[2 p]` = entries asked for in part p (more than `cap`: the list was too short and the caller takes the two passes),
// `list[2 p + 1]` = the largest label listed there (the largest of all = the volume's maximum: every id is listed at least
// once), part p's entries from `list + CENSUS_LIST_HEAD + p * cap` on; a workgroup asks for the places of its whole set with
// ONE atomic, a label pushed out of its slot on the way with one of its own (tissue: a few hundred a volume; a volume of noise
// fills the list and is told to go away).
template <typename T, bool LIST>
__global__ void __launch_bounds__(256) census_mark_rows_kernel(const T* __restrict__ vol, uint32_t rowvec, uint64_t nrows, uint32_t strips,
                                                               uint32_t fold, uint2* census, uint8_t* touched, uint32_t* list, uint32_t cap) {
    constexpr int PER = 16 / (int)sizeof(T);
    const uint4* v4 = reinterpret_cast<const uint4*>(vol);
    const uint32_t strip = blockIdx.x % strips;
    const uint64_t chunk = blockIdx.x / strips;
    const uint32_t sub = fold > 1 ? threadIdx.x / rowvec : 0u;
    const uint32_t col = fold > 1 ? threadIdx.x % rowvec : strip * 256u + threadIdx.x;
    // the labels this workgroup has met (direct-mapped, exchanged in: a label pushed out of its slot is marked at once, the
    // others when the workgroup is through): a workgroup meets a few dozen labels in its 64 rows, and a lookup in the global
    // table -- whose lines the L2s of the eight XCDs do not keep coherent inside a kernel -- is a trip to memory or a redundant
    // device-scope atomic, which the wave then waits for before it can use the rows it has in flight
    __shared__ uint32_t seen[CENSUS_SEEN];
    constexpr uint32_t NONE = 0xffffffffu;
    for (int i = threadIdx.x; i < CENSUS_SEEN; i += 256) seen[i] = NONE;
    __syncthreads();
    uint64_t row = chunk * CENSUS_ROWS_PER_BLOCK * fold + sub;
    const bool active = col < rowvec && sub < fold && row < nrows;
    auto out = [&](uint32_t v) {
        if (!LIST) { census_mark(census, touched, v); return; }
        uint32_t* const head = list + 2u * (blockIdx.x % CENSUS_LIST_PARTS);
        if (__hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > cap) return;     // (too short already: nobody will read it)
        const uint32_t at = atomicAdd(head, 1u);
        if (at < cap) list[CENSUS_LIST_HEAD + (uint64_t)(blockIdx.x % CENSUS_LIST_PARTS) * cap + at] = v;
        atomicMax(head + 1, v);
    };
    auto mark = [&](uint32_t v) {
        if (v == NONE) { out(v); return; }                                  // (the one id that cannot sit in the set)
        const uint32_t slot = (v * 2654435761u) >> (32 - CENSUS_SEEN_LOG2);
        if (seen[slot] == v) return;
        const uint32_t old = atomicExch(&seen[slot], v);
        if (old != v && old != NONE) out(old);
    };
    if (active) {
    uint32_t above[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) above[q] = 0xffffffffu;
    bool first = true;
    auto process = [&](const uint4& x) {
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
        uint32_t lab[PER];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (sizeof(T) == 4) lab[q] = w[q];
            else { lab[(2 * q) % PER] = w[q] & 0xffffu; lab[(2 * q + 1) % PER] = w[q] >> 16; }
        }
        uint32_t fresh = 0u;                               // bit q: label q is not the one above it nor the one to its left
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const bool f = (first || lab[q] != above[q]) && (q == 0 || lab[q] != lab[q - 1]);
            fresh |= f ? (1u << q) : 0u;
            above[q] = lab[q];
        }
        first = false;
        if (fresh) {
            const int q0 = __ffs((int)fresh) - 1;
            uint32_t cand = lab[0];
#pragma unroll
            for (int q = 1; q < PER; ++q) cand = q == q0 ? lab[q] : cand;
            mark(cand);
            fresh &= fresh - 1u;
            if (fresh) {                                   // two or more new labels in one vector: rare
#pragma unroll
                for (int q = 1; q < PER; ++q)
                    if (fresh & (1u << q)) mark(lab[q]);
            }
        }
    };
    // four rows per step, the next four in flight while these are looked at (one load per wave in flight is a quarter of the
    // bytes the memory system needs outstanding)
    constexpr int B = 4;
    const uint64_t last = nrows - 1;
    uint4 cur[B], nxt[B];
#pragma unroll
    for (int j = 0; j < B; ++j) { const uint64_t r = row + (uint64_t)j * fold; cur[j] = v4[(r < nrows ? r : last) * rowvec + col]; }
    for (int k = 0; k < CENSUS_ROWS_PER_BLOCK; k += B) {
        const uint64_t ahead = row + (uint64_t)B * fold;
        if (k + B < CENSUS_ROWS_PER_BLOCK) {
#pragma unroll
            for (int j = 0; j < B; ++j) { const uint64_t r = ahead + (uint64_t)j * fold; nxt[j] = v4[(r < nrows ? r : last) * rowvec + col]; }
        }
#pragma unroll
        for (int j = 0; j < B; ++j)
            if (row + (uint64_t)j * fold < nrows) process(cur[j]);
#pragma unroll
        for (int j = 0; j < B; ++j) cur[j] = nxt[j];
        row = ahead;
        if (row >= nrows) break;
    }
    }
    __syncthreads();
    if (!LIST) {
        for (int i = threadIdx.x; i < CENSUS_SEEN; i += 256)
            if (seen[i] != NONE) census_mark(census, touched, seen[i]);
    } else {
        __shared__ uint32_t placed[3];                      // entries of the workgroup, where they start in the list, their maximum
        if (threadIdx.x < 3) placed[threadIdx.x] = 0u;
        __syncthreads();
        uint32_t mine = 0u, top = 0u;
        for (int i = threadIdx.x; i < CENSUS_SEEN; i += 256)
            if (seen[i] != NONE) { ++mine; top = max(top, seen[i]); }
        const uint32_t off = mine ? atomicAdd(&placed[0], mine) : 0u;
        if (mine) atomicMax(&placed[2], top);
        __syncthreads();
        uint32_t* const head = list + 2u * (blockIdx.x % CENSUS_LIST_PARTS);
        if (threadIdx.x == 0 && placed[0]) { placed[1] = atomicAdd(head, placed[0]); atomicMax(head + 1, placed[2]); }
        __syncthreads();
        uint32_t* const part = list + CENSUS_LIST_HEAD + (uint64_t)(blockIdx.x % CENSUS_LIST_PARTS) * cap;
        uint32_t at = placed[1] + off;
        for (int i = threadIdx.x; i < CENSUS_SEEN; i += 256)
            if (seen[i] != NONE) { if (at < cap) part[at] = seen[i]; ++at; }
    }
}

// a census from a LIST of ids (the union over the ranks of a partitioned volume: every rank then ranks alike)
__global__ void __launch_bounds__(256) census_from_ids_kernel(const uint32_t* __restrict__ ids, uint64_t n, uint2* census, uint8_t* touched) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { atomicOr(&census[ids[i] >> 5].x, 1u << (ids[i] & 31u)); touched[ids[i] >> 17] = 1; }
}

// ... and from the parts of a label list (census_mark_rows_kernel, LIST): blockIdx.y = the part
__global__ void __launch_bounds__(256) census_from_list_kernel(const uint32_t* __restrict__ list, uint32_t cap, uint2* census, uint8_t* touched) {
    const uint32_t n = list[2u * blockIdx.y];
    const uint32_t* ids = list + CENSUS_LIST_HEAD + (uint64_t)blockIdx.y * cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t v = ids[i];
        atomicOr(&census[v >> 5].x, 1u << (v & 31u)); touched[v >> 17] = 1;
    }
}

__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* lds) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ void __launch_bounds__(256) census_block_sums_kernel(const uint2* __restrict__ census, uint64_t words, const uint8_t* __restrict__ touched,
                                                                uint32_t* block_sums) {
    __shared__ uint32_t lds[4];
    if (!touched[blockIdx.x]) {                        // (the whole workgroup: nobody is left at a barrier)
        if (threadIdx.x == 0) block_sums[blockIdx.x] = 0u;
        return;
    }
    const uint64_t base = (uint64_t)blockIdx.x * CENSUS_WORDS_PER_BLOCK;
    uint32_t s = 0;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const uint64_t w = base + (uint64_t)k * 256 + threadIdx.x;
        if (w < words) s += (uint32_t)__popc(census[w].x);
    }
    s = block_sum_256(s, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s;
}

// one workgroup: block_sums -> exclusive offsets in place, the total behind them
__global__ void __launch_bounds__(1024) census_top_kernel(uint32_t* block_sums, uint32_t nblocks) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (nblocks + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, nblocks);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += block_sums[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 1024; ++i) { const uint32_t t = part[i]; part[i] = run; run += t; }
        block_sums[nblocks] = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t t = block_sums[i]; block_sums[i] = run; run += t; }
}

// census[w].below, and the ids themselves (ids == NULL: only the prefix)
__global__ void __launch_bounds__(256) census_apply_kernel(uint2* census, uint64_t words, const uint8_t* __restrict__ touched,
                                                           const uint32_t* __restrict__ block_sums, uint32_t* ids) {
    __shared__ uint32_t wave_tot[4];
    if (!touched[blockIdx.x]) return;                  // no id of this block is present: nothing will ask for its prefix
    const uint64_t base = (uint64_t)blockIdx.x * CENSUS_WORDS_PER_BLOCK;
    uint32_t run = block_sums[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = 0; k < 16; ++k) {                      // 256 consecutive words per round
        const uint64_t w = base + (uint64_t)k * 256 + threadIdx.x;
        const uint32_t bits = w < words ? census[w].x : 0u;
        const uint32_t c = (uint32_t)__popc(bits);
        uint32_t inc = c;                               // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        uint32_t before = run;
        for (int q = 0; q < wv; ++q) before += wave_tot[q];
        const uint32_t round_total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        const uint32_t below = before + inc - c;
        if (w < words) {
            census[w].y = below;
            if (ids) {
                uint32_t b = bits, at = below;
                while (b) { const int bit = __ffs((int)b) - 1; ids[at++] = (uint32_t)(w << 5) + (uint32_t)bit; b &= b - 1u; }
            }
        }
        run += round_total;
        __syncthreads();
    }
}

__device__ __forceinline__ uint32_t census_rank(const uint2* __restrict__ census, uint32_t v, uint32_t limit, uint32_t& missing) {
    if (v > limit) { missing = 1u; return 0u; }
    const uint2 e = census[v >> 5];
    const uint32_t bit = 1u << (v & 31u);
    if (!(e.x & bit)) missing = 1u;
    return e.y + (uint32_t)__popc(e.x & (bit - 1u));
}

// out[p] = rank(vol[p]); *status |= 1 when a voxel holds an id the census does not know (a census from a list)
template <typename T>
__global__ void __launch_bounds__(256) census_rank_kernel(const T* __restrict__ vol, T* __restrict__ out, uint64_t n, uint64_t nvec,
                                                          const uint2* __restrict__ census, uint32_t limit, uint32_t* status) {
    constexpr int PER = 16 / (int)sizeof(T);
    const uint4* v4 = reinterpret_cast<const uint4*>(vol);
    uint4* o4 = reinterpret_cast<uint4*>(out);
    uint32_t missing = 0u, prev = 0u, prev_rank = 0u;
    bool have = false;
    auto rank_of = [&](uint32_t v) {
        if (!have || v != prev) { prev_rank = census_rank(census, v, limit, missing); prev = v; have = true; }
        return prev_rank;
    };
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 x = v4[i];
        uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (sizeof(T) == 4) w[k] = rank_of(w[k]);
            else { const uint32_t a = rank_of(w[k] & 0xffffu); const uint32_t b = rank_of(w[k] >> 16); w[k] = a | (b << 16); }
        }
        o4[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    for (uint64_t t = nvec * PER + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        out[t] = (T)rank_of((uint32_t)vol[t]);
    if (missing) atomicOr(status, 1u);
}

unsigned stream_blocks(uint64_t items) {
    uint64_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

uint64_t census_words(uint32_t max_label) { return ((uint64_t)max_label >> 5) + 1; }
uint64_t census_bytes(uint32_t max_label) { return census_words(max_label) * sizeof(uint2); }
static uint64_t census_blocks(uint32_t max_label) { return (census_words(max_label) + CENSUS_WORDS_PER_BLOCK - 1) / CENSUS_WORDS_PER_BLOCK; }
// scratch: block sums u32[blocks + 1] (the last one: the number of ids), then touched u8[blocks] (zero before the marking pass)
uint64_t census_scratch_bytes(uint32_t max_label) { return (census_blocks(max_label) + 1) * sizeof(uint32_t) + census_blocks(max_label) + 16; }
static uint8_t* census_touched(void* scratch, uint32_t max_label) { return (uint8_t*)((uint32_t*)scratch + census_blocks(max_label) + 1); }

namespace {
struct CensusRows { uint32_t rowvec, strips, fold; uint64_t nrows, chunks, done; };
// rows of whole 16-byte vectors: the strips walk down the real rows.  Otherwise the volume as one long run of 256-vector
// pseudo-rows: what is above a vector is then no neighbour of it, but the workgroup's LDS set -- the part that keeps the table
// out of the streaming loop -- works the same.  `done`: the voxels the row kernel takes (0: it does not apply)
CensusRows census_rows(const void* vol, int itemsize, uint64_t n, int64_t row_len) {
    CensusRows r{};
    if (((uintptr_t)vol & 15) != 0 || n == 0) return r;
    const int per = 16 / itemsize;
    const bool real_rows = row_len > 0 && (row_len * itemsize) % 16 == 0 && n % (uint64_t)row_len == 0;
    r.rowvec = real_rows ? (uint32_t)(row_len * itemsize / 16) : 256u;
    r.nrows = real_rows ? n / (uint64_t)row_len : (n / per) / 256u;
    r.strips = r.rowvec >= 256 ? (r.rowvec + 255) / 256 : 1;
    r.fold = r.rowvec >= 256 ? 1 : 256 / r.rowvec;
    r.chunks = (r.nrows + (uint64_t)CENSUS_ROWS_PER_BLOCK * r.fold - 1) / ((uint64_t)CENSUS_ROWS_PER_BLOCK * r.fold);
    if (r.nrows && r.chunks * r.strips < (1ull << 31)) r.done = r.nrows * r.rowvec * (uint64_t)per;
    return r;
}
}  // namespace

void launch_census_mark(hipStream_t s, const void* vol, int itemsize, uint64_t n, int64_t row_len, void* census, void* scratch,
                        uint32_t max_label) {
    uint8_t* touched = census_touched(scratch, max_label);
    if (n == 0) return;
    const int per = 16 / itemsize;
    const CensusRows r = census_rows(vol, itemsize, n, row_len);
    const uint64_t done = r.done;
    if (done) {
        const dim3 grid((unsigned)(r.chunks * r.strips));
        if (itemsize == 2) hipLaunchKernelGGL((census_mark_rows_kernel<uint16_t, false>), grid, dim3(256), 0, s, (const uint16_t*)vol, r.rowvec, r.nrows, r.strips, r.fold, (uint2*)census, touched, (uint32_t*)nullptr, 0u);
        else               hipLaunchKernelGGL((census_mark_rows_kernel<uint32_t, false>), grid, dim3(256), 0, s, (const uint32_t*)vol, r.rowvec, r.nrows, r.strips, r.fold, (uint2*)census, touched, (uint32_t*)nullptr, 0u);
    }
    if (done == n) return;
    // what is left (less than a pseudo-row), or a volume that does not start on a 16-byte boundary: the flat kernel
    const char* rest = (const char*)vol + done * itemsize;
    const uint64_t m = n - done;
    const uint64_t nvec = ((uintptr_t)rest & 15) ? 0 : m / per;
    const unsigned blocks = stream_blocks(nvec ? nvec : m);
    if (itemsize == 2) hipLaunchKernelGGL(census_mark_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)rest, m, nvec, (uint2*)census, touched);
    else               hipLaunchKernelGGL(census_mark_kernel<uint32_t>, dim3(blocks), dim3(256), 0, s, (const uint32_t*)rest, m, nvec, (uint2*)census, touched);
}

// ONE pass over a volume of unknown maximum (see census_mark_rows_kernel, LIST): its labels, with repeats, into `list` (a zeroed
// head of census_list_head_bytes(), then census_list_parts() parts of `cap` entries).  Returns false -- nothing enqueued -- where
// the row kernel does not take the whole volume.  Afterwards the head holds { entries asked for, maximum } per part (asked for
// > cap anywhere: too short); launch_census_from_list marks them.
uint64_t census_list_head_bytes() { return CENSUS_LIST_HEAD * sizeof(uint32_t); }
uint32_t census_list_parts() { return CENSUS_LIST_PARTS; }
uint64_t census_list_capacity(uint64_t n) {             // per part.  Tissue lists ~40 labels per 64 rows of a strip; a volume of noise every voxel
    const uint64_t cap = n / 256 / CENSUS_LIST_PARTS;
    return cap < 4096 ? 4096 : (cap > (1u << 22) ? (1u << 22) : cap);
}
void launch_census_from_list(hipStream_t s, const void* list, uint32_t cap, uint32_t most, void* census, void* scratch, uint32_t max_label) {
    if (most == 0) return;
    unsigned bx = (most + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(census_from_list_kernel, dim3(bx, CENSUS_LIST_PARTS), dim3(256), 0, s, (const uint32_t*)list, cap, (uint2*)census,
                       census_touched(scratch, max_label));
}
bool launch_census_list(hipStream_t s, const void* vol, int itemsize, uint64_t n, int64_t row_len, void* list, uint32_t cap) {
    const CensusRows r = census_rows(vol, itemsize, n, row_len);
    if (r.done != n || n == 0) return false;
    const dim3 grid((unsigned)(r.chunks * r.strips));
    if (itemsize == 2) hipLaunchKernelGGL((census_mark_rows_kernel<uint16_t, true>), grid, dim3(256), 0, s, (const uint16_t*)vol, r.rowvec, r.nrows, r.strips, r.fold, (uint2*)nullptr, (uint8_t*)nullptr, (uint32_t*)list, cap);
    else               hipLaunchKernelGGL((census_mark_rows_kernel<uint32_t, true>), grid, dim3(256), 0, s, (const uint32_t*)vol, r.rowvec, r.nrows, r.strips, r.fold, (uint2*)nullptr, (uint8_t*)nullptr, (uint32_t*)list, cap);
    return true;
}

void launch_census_from_ids(hipStream_t s, const uint32_t* ids_dev, uint64_t n, void* census, void* scratch, uint32_t max_label) {
    if (n == 0) return;
    hipLaunchKernelGGL(census_from_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ids_dev, n, (uint2*)census,
                       census_touched(scratch, max_label));
}

// the prefix counts of a marked census; `*total_dev` (inside scratch) receives the number of ids.  Call once with ids_out == NULL
// to learn the total, again with the array to expand the ids (or once with an array known to be large enough).
void launch_census_scan(hipStream_t s, void* census, uint32_t max_label, void* scratch, uint32_t* ids_out, uint32_t** total_dev) {
    const uint64_t words = census_words(max_label);
    const uint32_t nblocks = (uint32_t)((words + CENSUS_WORDS_PER_BLOCK - 1) / CENSUS_WORDS_PER_BLOCK);
    uint32_t* block_sums = (uint32_t*)scratch;
    const uint8_t* touched = census_touched(scratch, max_label);
    if (ids_out == nullptr || total_dev != nullptr) {      // (the second call, for the ids alone, finds the offsets in place)
        hipLaunchKernelGGL(census_block_sums_kernel, dim3(nblocks), dim3(256), 0, s, (const uint2*)census, words, touched, block_sums);
        hipLaunchKernelGGL(census_top_kernel, dim3(1), dim3(1024), 0, s, block_sums, nblocks);
    }
    hipLaunchKernelGGL(census_apply_kernel, dim3(nblocks), dim3(256), 0, s, (uint2*)census, words, touched, block_sums, ids_out);
    if (total_dev) *total_dev = block_sums + nblocks;
}

void launch_census_rank(hipStream_t s, const void* vol, void* out, int itemsize, uint64_t n, const void* census, uint32_t max_label,
                        uint32_t* status) {
    if (n == 0) return;
    const uint64_t nvec = (((uintptr_t)vol | (uintptr_t)out) & 15) ? 0 : n / (16 / itemsize);
    const unsigned blocks = stream_blocks(nvec ? nvec : n);
    if (itemsize == 2) hipLaunchKernelGGL(census_rank_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)vol, (uint16_t*)out, n, nvec, (const uint2*)census, max_label, status);
    else               hipLaunchKernelGGL(census_rank_kernel<uint32_t>, dim3(blocks), dim3(256), 0, s, (const uint32_t*)vol, (uint32_t*)out, n, nvec, (const uint2*)census, max_label, status);
}

}  // namespace ta
