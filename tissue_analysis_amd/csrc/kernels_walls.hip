// kernels_walls.hip -- wall voxels of every label pair (SIA:759-880, 1049-1111) on gfx950.
//
// A voxel p of label l is a wall voxel of the pair (l, m) when one of its 18 neighbours (faces + edges: scipy's
// generate_binary_structure(3, 2), the structure the reference dilates with) carries m != l.  One record
// {lo, hi, coordinates of p} per distinct m, records ordered by the position of p in memory.
//
// Layout of the work.  A wave walks DOWN axis 1 over `rows_per_wave` rows of one plane, for one strip of 256 columns:
// lane i holds the four columns c0 + 4 i .. + 3 of the 3 x 3 rows around the current one in registers (one 8- or
// 16-byte load per row and lane), so one step loads only the three new rows (one per plane) -- every row is read three
// times in all, by the walkers of its own plane and of the two planes next to it, from L2 -- three of the four column
// neighbours are the lane's own registers and the fourth comes from the next lane through DPP.
// Out-of-volume neighbours are handled by CLAMPING the plane / row / column index: the clamped position is itself one
// of the 18 neighbours (or the voxel), so it adds no label.
// Distinct labels per voxel without an 18 x 18 compare: with x_q = neighbour_q XOR l, the distinct non-zero x are
// pulled out in increasing order, one per round, by "smallest x above the last one" (18 subtracts + a min3 tree); a
// round runs only while some lane of the wave still has one left (most steps: one or two rounds, none at all where
// the 3 x 3 x 3 block is one label).
// Two passes over the same decomposition: COUNT writes one record count per (row, strip) and one byte per lane of it;
// an exclusive scan ON THE DEVICE turns the former into offsets; EMIT starts each lane at the strip's offset plus the
// lanes before it, so the records come out in memory order with no atomics and without a second walk over the columns
// (a column's labels are stored as soon as its rounds are done), already as (lo, hi) / coordinates in array-axis order.  The host reads back ONE number (the total)
// between the passes, to size the output.
#include "ta_kernels.h"
#include "ta_sweep_common.h"

namespace ta {

namespace {

constexpr int WNJ = 4;                 // columns per lane
constexpr int WSC = 64 * WNJ;          // columns per strip

struct WallRow {                       // one row of the 3 x 3 window: 4 columns per lane + the two columns beside the strip
    uint32_t v[WNJ];
    uint32_t hl, hr;
};

__device__ __forceinline__ uint32_t lane_shl1(uint32_t src, uint32_t lane63_value) {
    // lane i <- src of lane i+1 ; lane 63 keeps lane63_value   (DPP wave_shl:1)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane63_value, (int)src, 0x130, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }

// inclusive add-scan over the 64 lanes (row_shr 1,2,4,8 then the two row broadcasts)
__device__ __forceinline__ uint32_t wall_scan_add(uint32_t x) {
#define TA_DPP_ADD(ctrl, rmask) \
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xf, false);
    TA_DPP_ADD(0x111, 0xf) TA_DPP_ADD(0x112, 0xf) TA_DPP_ADD(0x114, 0xf) TA_DPP_ADD(0x118, 0xf)
    TA_DPP_ADD(0x142, 0xa) TA_DPP_ADD(0x143, 0xc)
#undef TA_DPP_ADD
    return x;
}

// One row of the window: lane i takes the 4 consecutive columns c0 + 4 i .. + 3 -- ONE 8- or 16-byte load where the rows
// are aligned for it -- clamped to the last column of the row; hl / hr: the columns beside the strip (scalar loads).
template <typename T>
__device__ __forceinline__ void wall_load_row(WallRow& r, const T* row, uint32_t colq, uint32_t last, bool quad_ok,
                                              uint32_t col_left, uint32_t col_right) {
    if (quad_ok) {
        if (sizeof(T) == 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(row + colq);
            r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
        } else {
            const uint2 q = *reinterpret_cast<const uint2*>(row + colq);
            r.v[0] = q.x & 0xffffu; r.v[1] = q.x >> 16; r.v[2] = q.y & 0xffffu; r.v[3] = q.y >> 16;
        }
    } else {
#pragma unroll
        for (int j = 0; j < WNJ; ++j) {
            const uint32_t c = colq + (uint32_t)j;
            r.v[j] = (uint32_t)row[c <= last ? c : last];
        }
    }
    r.hl = load_uniform_voxel<T>(row + col_left);
    r.hr = load_uniform_voxel<T>(row + col_right);
}

// x_q = neighbour_q XOR v for the 18 neighbours of column j of the lane: the row itself and its four face rows
// (side[0..4]) with their column neighbours, the four edge rows at the same column
__device__ __forceinline__ void wall_neighbours(uint32_t (&x)[18], const WallRow* const (&side)[5], const uint32_t (&Lc)[5],
                                                const uint32_t (&Rc)[5], const WallRow& e0, const WallRow& e1,
                                                const WallRow& e2, const WallRow& e3, const int j, const uint32_t v) {
    int q = 0;
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        x[q++] = (j > 0 ? side[f]->v[j > 0 ? j - 1 : 0] : Lc[f]) ^ v;
        if (f > 0) x[q++] = side[f]->v[j] ^ v;
        x[q++] = (j < WNJ - 1 ? side[f]->v[j < WNJ - 1 ? j + 1 : j] : Rc[f]) ^ v;
    }
    x[q++] = e0.v[j] ^ v; x[q++] = e1.v[j] ^ v; x[q++] = e2.v[j] ^ v; x[q++] = e3.v[j] ^ v;
}

// the smallest x above d: d + 1 + min_q((x_q - d - 1) mod 2^32) -- an x at or below d wraps to the top, above every x
// that does not -- 18 subtracts and a min3 tree, no compares
__device__ __forceinline__ uint32_t wall_next_above(const uint32_t (&x)[18], uint32_t d) {
    const uint32_t e = d + 1u;
    uint32_t m = umin3(umin3(x[0] - e, x[1] - e, x[2] - e), umin3(x[3] - e, x[4] - e, x[5] - e),
                       umin3(x[6] - e, x[7] - e, x[8] - e));
    m = umin3(m, umin3(x[9] - e, x[10] - e, x[11] - e), umin3(x[12] - e, x[13] - e, x[14] - e));
    m = umin3(m, x[15] - e, umin3(x[16] - e, x[17] - e, 0xFFFFFFFFu));
    return m + e;
}

}  // namespace

struct WallArgs {
    const void* vol;
    int64_t n0, n1, n2;
    int32_t nstrips, rows_per_wave;
    int32_t quads_ok;              // rows start 4-element aligned and hold a multiple of 4 columns: one vector load per lane
    uint32_t* counts;              // [n0 * n1 * nstrips] records of each (row, strip): written by COUNT, read by EMIT
    uint8_t* lane_counts;          // [cells][64] records of each lane of each non-empty (row, strip): likewise
    const uint64_t* offsets;       // EMIT: their exclusive scan
    uint2* out_pairs;              // [n] (lo, hi)
    int32_t* out_coords;           // [n][3], array-axis order
    int32_t inv[3];                // inv[i] = memory axis of array axis i
};

template <typename T, bool EMIT>
__global__ void __launch_bounds__(256) wall_rows_kernel(WallArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t wi = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t chunks_b = (A.n1 + A.rows_per_wave - 1) / A.rows_per_wave;
    const int64_t per_plane = chunks_b * A.nstrips;
    if (wi >= A.n0 * per_plane) return;
    const int64_t a = wi / per_plane, rem = wi - a * per_plane, cb = rem / A.nstrips;
    const int32_t s = (int32_t)(rem - cb * A.nstrips);
    const int64_t b0 = cb * A.rows_per_wave, b1 = b0 + A.rows_per_wave < A.n1 ? b0 + A.rows_per_wave : A.n1;
    const uint32_t c0 = (uint32_t)s * WSC, last = (uint32_t)(A.n2 - 1);
    const uint32_t colq = c0 + (uint32_t)WNJ * (uint32_t)lane;
    const bool quad_ok = A.quads_ok && colq + (WNJ - 1) <= last;
    const uint32_t col_left = c0 > 0u ? c0 - 1u : 0u, col_right = c0 + WSC <= last ? c0 + WSC : last;
    const T* vol = (const T*)A.vol;
    const T* plane[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        int64_t ap = a + p - 1;
        ap = ap < 0 ? 0 : (ap >= A.n0 ? A.n0 - 1 : ap);
        plane[p] = vol + ap * A.n1 * A.n2;
    }
    auto rowp = [&](int p, int64_t b) {
        b = b < 0 ? 0 : (b >= A.n1 ? A.n1 - 1 : b);
        return plane[p] + b * A.n2;
    };
    const T* ahead[3];                 // row b + 2 of each plane, clamped: advanced by one row per step
#pragma unroll
    for (int p = 0; p < 3; ++p) ahead[p] = rowp(p, b0 + 1);

    WallRow W[3][3], nxt[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        wall_load_row<T>(W[p][1], rowp(p, b0 - 1), colq, last, quad_ok, col_left, col_right);
        wall_load_row<T>(W[p][2], rowp(p, b0), colq, last, quad_ok, col_left, col_right);
        wall_load_row<T>(nxt[p], ahead[p], colq, last, quad_ok, col_left, col_right);
    }
    for (int64_t b = b0; b < b1; ++b) {
#pragma unroll
        for (int p = 0; p < 3; ++p) { W[p][0] = W[p][1]; W[p][1] = W[p][2]; W[p][2] = nxt[p]; }
        if (b + 1 < b1) {
            const int64_t adv = b + 2 < A.n1 ? A.n2 : 0;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                ahead[p] += adv;
                wall_load_row<T>(nxt[p], ahead[p], colq, last, quad_ok, col_left, col_right);
            }
        }
        const int64_t cell = (a * A.n1 + b) * A.nstrips + s;
        uint64_t base = 0;
        if (EMIT) {
            if (A.counts[cell] == 0u) continue;         // the count pass found nothing here
            base = A.offsets[cell];
        }
        // the whole 3 x 3 x (strip + 2) block one label: nothing to do (background, cell interiors)
        {
            const uint32_t ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)W[1][1].v[0]);
            uint32_t u = 0;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int j = 0; j < WNJ; ++j) u |= W[p][r].v[j] ^ ref;
                    u |= (W[p][r].hl ^ ref) | (W[p][r].hr ^ ref);
                }
            if (!__any(u != 0u)) {
                if (!EMIT && lane == 0) A.counts[cell] = 0u;
                continue;
            }
        }
        // the columns beside each lane's four, for the five rows whose column neighbours count (the row itself and its
        // four face rows): from the next lane through DPP, from the scalar halo loads at the ends of the strip
        const WallRow* const side[5] = {&W[1][1], &W[0][1], &W[2][1], &W[1][0], &W[1][2]};
        uint32_t Lc[5], Rc[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            Lc[f] = lane_shr1(side[f]->v[WNJ - 1], side[f]->hl);
            Rc[f] = lane_shl1(side[f]->v[0], side[f]->hr);
        }
        // EMIT: where this lane's records go -- the row strip's offset plus the lanes before it (their totals were
        // written by the count pass), so a column's labels are stored as soon as its rounds are done
        uint32_t pos = 0;                       // relative to the strip's first record: 32-bit offsets from a scalar base
        uint2* out_pairs = nullptr;
        int32_t* out_coords = nullptr;
        if (EMIT) {
            const uint32_t mine = A.lane_counts[cell * 64 + lane];
            pos = wall_scan_add(mine) - mine;
            out_pairs = A.out_pairs + base;
            out_coords = A.out_coords + 3 * base;
        }
        const int32_t ma = (int32_t)a, mb = (int32_t)b;
        struct __attribute__((packed, aligned(4))) Int3 { int32_t x, y, z; };
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < WNJ; ++j) {
            const uint32_t v = W[1][1].v[j];
            uint32_t x[18];
            wall_neighbours(x, side, Lc, Rc, W[0][0], W[0][2], W[2][0], W[2][2], j, v);
            uint32_t mx = umax3(umax3(x[0], x[1], x[2]), umax3(x[3], x[4], x[5]), umax3(x[6], x[7], x[8]));
            mx = umax3(mx, umax3(x[9], x[10], x[11]), umax3(x[12], x[13], x[14]));
            mx = umax3(mx, umax3(x[15], x[16], x[17]), 0u);
            if (colq + (uint32_t)j > last) mx = 0u;            // a column past the end of the row (clamped copies)
            if (!__any(mx != 0u)) continue;
            Int3 xyz;                                          // the voxel in array-axis order: one 12-byte store per record
            if (EMIT) {
                const int32_t mc = (int32_t)(colq + (uint32_t)j);
                xyz.x = A.inv[0] == 0 ? ma : (A.inv[0] == 1 ? mb : mc);
                xyz.y = A.inv[1] == 0 ? ma : (A.inv[1] == 1 ? mb : mc);
                xyz.z = A.inv[2] == 0 ? ma : (A.inv[2] == 1 ? mb : mc);
            }
            auto put = [&](uint32_t xd) {
                const uint32_t m = v ^ xd;
                out_pairs[pos] = make_uint2(v < m ? v : m, v < m ? m : v);
                *reinterpret_cast<Int3*>(out_coords + 3u * pos) = xyz;
                ++pos;
            };
            // distinct x in increasing order, one per round.  The first round (d = 0) gives the smallest non-zero x;
            // where it equals the largest there is one label and the voxel is done -- most wall voxels.  Further rounds
            // run only while some lane of the wave still has labels between its last one and its largest.  EMIT keeps
            // the first three labels in registers and stores them after the rounds (one pass of stores per column, not
            // one per round); a fourth and later label -- rare -- is stored where it is found.
            uint32_t d = mx ? wall_next_above(x, 0u) : 0u;
            const uint32_t k0 = d;
            uint32_t k1 = 0u, k2 = 0u, n = mx ? 1u : 0u;
            if (__any(d < mx)) {
                const uint32_t nd = wall_next_above(x, d);
                if (d < mx) { d = nd; k1 = nd; ++n; }
                if (__any(d < mx)) {
                    const uint32_t nd2 = wall_next_above(x, d);
                    if (d < mx) { d = nd2; k2 = nd2; ++n; }
                }
            }
            if (EMIT) {
                if (n > 0u) put(k0);
                if (n > 1u) put(k1);
                if (n > 2u) put(k2);
            }
            while (__any(d < mx)) {
                const uint32_t nd = wall_next_above(x, d);
                if (d < mx) { d = nd; ++n; if (EMIT) put(nd); }
            }
            total += n;
        }
        if (!EMIT) {
            A.lane_counts[cell * 64 + lane] = (uint8_t)total;          // <= 4 x 18
            const uint32_t incl = wall_scan_add(total);
            if (lane == 63) A.counts[cell] = incl;
        }
    }
}

// ---- exclusive scan of the (row, strip) counts: block sums, one block over the sums, apply ----------------------
constexpr int SCAN_PER_BLOCK = 2048;   // 256 threads x 8

__global__ void __launch_bounds__(256) wall_scan_sums_kernel(const uint32_t* counts, uint64_t n, uint64_t* block_sums) {
    __shared__ uint64_t part[4];
    const uint64_t lo = (uint64_t)blockIdx.x * SCAN_PER_BLOCK;
    uint64_t sum = 0;
    for (int k = 0; k < 8; ++k) {
        const uint64_t i = lo + (uint64_t)k * 256 + threadIdx.x;
        sum += i < n ? counts[i] : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += (uint64_t)__shfl_down((long long)sum, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// one block: block_sums -> exclusive, total out
__global__ void __launch_bounds__(256) wall_scan_top_kernel(uint64_t* block_sums, uint64_t nblocks, uint64_t* total) {
    __shared__ uint64_t sh[256];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t lo = 0; lo < nblocks; lo += 256) {
        const uint64_t i = lo + threadIdx.x;
        const uint64_t mine = i < nblocks ? block_sums[i] : 0ull;
        sh[threadIdx.x] = mine;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint64_t add = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0ull;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nblocks) block_sums[i] = carry + sh[threadIdx.x] - mine;
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(256) wall_scan_apply_kernel(const uint32_t* counts, uint64_t n, const uint64_t* block_sums,
                                                              uint64_t* offsets) {
    __shared__ uint64_t wsum[4];
    const uint64_t lo = (uint64_t)blockIdx.x * SCAN_PER_BLOCK + (uint64_t)threadIdx.x * 8;    // 8 consecutive per thread
    uint32_t c[8];
    uint64_t mine = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { c[k] = lo + k < n ? counts[lo + k] : 0u; mine += c[k]; }
    uint64_t incl = mine;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t up = (uint64_t)__shfl_up((long long)incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint64_t run = block_sums[blockIdx.x] + incl - mine;
    for (int i = 0; i < w; ++i) run += wsum[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (lo + k < n) offsets[lo + k] = run;
        run += c[k];
    }
}

// the same three kernels as a general exclusive scan of uint32 counts (the adjacency sort buckets its pairs with it)
uint64_t scan_u32_scratch_bytes(uint64_t n) { return ((n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK + 1) * 8 + 16; }

void launch_scan_u32_exclusive(hipStream_t s, const uint32_t* counts, uint64_t n, void* scratch, uint64_t* offsets) {
    if (n == 0) return;
    const uint64_t blocks = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    uint64_t* block_sums = (uint64_t*)scratch;
    uint64_t* total = block_sums + blocks;
    hipLaunchKernelGGL(wall_scan_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, s, counts, n, block_sums);
    hipLaunchKernelGGL(wall_scan_top_kernel, dim3(1), dim3(256), 0, s, block_sums, blocks, total);
    hipLaunchKernelGGL(wall_scan_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, counts, n, block_sums, offsets);
}

// ---- host side ------------------------------------------------------------------------------------------------
static int wall_rows_per_wave(int64_t n1) { return n1 < 16 ? (int)(n1 > 0 ? n1 : 1) : 16; }

WallPlan wall_plan(int64_t n0, int64_t n1, int64_t n2) {
    WallPlan p;
    p.nstrips = (int32_t)((n2 + WSC - 1) / WSC);
    p.rows_per_wave = wall_rows_per_wave(n1);
    p.cells = (uint64_t)n0 * (uint64_t)n1 * (uint64_t)p.nstrips;
    p.scan_blocks = (p.cells + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    const uint64_t chunks_b = (uint64_t)((n1 + p.rows_per_wave - 1) / p.rows_per_wave);
    p.waves = (uint64_t)n0 * chunks_b * (uint64_t)p.nstrips;
    return p;
}

static WallArgs wall_args(const void* vol, int itemsize, int64_t n0, int64_t n1, int64_t n2, const WallPlan& p) {
    WallArgs a;
    a.vol = vol; a.n0 = n0; a.n1 = n1; a.n2 = n2;
    a.quads_ok = ((uintptr_t)vol % (uintptr_t)(WNJ * itemsize) == 0) && (n2 % WNJ == 0);
    a.nstrips = p.nstrips; a.rows_per_wave = p.rows_per_wave;
    a.counts = nullptr; a.lane_counts = nullptr; a.offsets = nullptr; a.out_pairs = nullptr; a.out_coords = nullptr;
    a.inv[0] = 0; a.inv[1] = 1; a.inv[2] = 2;
    return a;
}

void launch_wall_count(hipStream_t s, const void* vol, int itemsize, int64_t n0, int64_t n1, int64_t n2,
                       uint32_t* counts, uint8_t* lane_counts, uint64_t* offsets, uint64_t* block_sums, uint64_t* total) {
    const WallPlan p = wall_plan(n0, n1, n2);
    if (p.cells == 0) { (void)hipMemsetAsync(total, 0, 8, s); return; }
    WallArgs a = wall_args(vol, itemsize, n0, n1, n2, p);
    a.counts = counts; a.lane_counts = lane_counts;
    const unsigned blocks = (unsigned)((p.waves + 3) / 4);
    if (itemsize == 2) hipLaunchKernelGGL((wall_rows_kernel<uint16_t, false>), dim3(blocks), dim3(256), 0, s, a);
    else               hipLaunchKernelGGL((wall_rows_kernel<uint32_t, false>), dim3(blocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(wall_scan_sums_kernel, dim3((unsigned)p.scan_blocks), dim3(256), 0, s, counts, p.cells, block_sums);
    hipLaunchKernelGGL(wall_scan_top_kernel, dim3(1), dim3(256), 0, s, block_sums, p.scan_blocks, total);
    hipLaunchKernelGGL(wall_scan_apply_kernel, dim3((unsigned)p.scan_blocks), dim3(256), 0, s, counts, p.cells, block_sums, offsets);
}

void launch_wall_emit(hipStream_t s, const void* vol, int itemsize, int64_t n0, int64_t n1, int64_t n2,
                      const uint32_t* counts, const uint8_t* lane_counts, const uint64_t* offsets, uint32_t* out_pairs,
                      int32_t* out_coords, const int perm[3]) {
    const WallPlan p = wall_plan(n0, n1, n2);
    if (p.cells == 0) return;
    WallArgs a = wall_args(vol, itemsize, n0, n1, n2, p);
    a.counts = const_cast<uint32_t*>(counts); a.lane_counts = const_cast<uint8_t*>(lane_counts); a.offsets = offsets; a.out_pairs = (uint2*)out_pairs; a.out_coords = out_coords;
    for (int k = 0; k < 3; ++k) a.inv[perm[k]] = k;          // perm[k] = array axis of memory axis k
    const unsigned blocks = (unsigned)((p.waves + 3) / 4);
    if (itemsize == 2) hipLaunchKernelGGL((wall_rows_kernel<uint16_t, true>), dim3(blocks), dim3(256), 0, s, a);
    else               hipLaunchKernelGGL((wall_rows_kernel<uint32_t, true>), dim3(blocks), dim3(256), 0, s, a);
}

}  // namespace ta
