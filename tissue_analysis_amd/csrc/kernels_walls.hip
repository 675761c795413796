// kernels_walls.hip -- wall voxels of every label pair (SIA:759-880, 1049-1111) on gfx950.
//
// A voxel p of label l is a wall voxel of the pair (l, m) when one of its 18 neighbours (faces + edges: scipy's
// generate_binary_structure(3, 2), the structure the reference dilates with) carries m != l.  One record
// {lo, hi, coordinates of p} per distinct m, records ordered by the position of p in memory.
//
// Layout of the work.  A wave walks DOWN axis 1 over `rows_per_wave` rows of one plane, for one strip of 256 columns:
// lane i holds the four columns c0 + 4 i .. + 3 of the 3 x 3 rows around the current one in registers -- a ring of three
// row slots per plane, so nothing is moved; a step loads the three rows the next one will need (one 8- or 16-byte load
// per row and lane) and leaves them as memory returned them until then -- every row is read three times in all, by the
// walkers of its own plane and of the two planes next to it, from L2; three of the four column neighbours are the lane's
// own registers and the fourth comes from the next lane through DPP.  Out-of-volume neighbours are handled by CLAMPING
// the plane / row / column index: the clamped position is itself one of the 18 neighbours (or the voxel), so it adds no
// label.
// A (row, strip) of the walk is a CELL: 256 voxels whose records are consecutive in the output.
//
// Distinct labels per voxel without an 18 x 18 compare: with t_q = (neighbour_q XOR l) - 1 (one v_xad_u32; a neighbour
// of the voxel's own label wraps to 0xFFFFFFFF), the smallest t is the first label (unsigned min3 tree) and the largest
// one is the last (SIGNED max3 tree: the wrapped ones are -1; labels from 2^31 up take the unsigned route, see WIDE):
// 36 instructions per voxel, and the voxel is done where the two agree -- one neighbour label, most wall voxels.
// Further labels come out in increasing order, one per round, by "smallest t above the last one" (18 subtracts + a
// min3 tree); a round runs only while some lane of the wave still has one left.
//
// ONE compute pass.  COUNT finds the labels of a cell, takes room for its records in a STAGING area (a wave takes 512
// records at a time with one returning atomic on one of 256 cursors that each own a region: no hot word, an atomic
// every eighth cell or so) and stores them there as {label, the other label, column} in memory order (lane, column,
// label), through the wave's slice of LDS; it leaves the cell's record count and its place in the staging area.  An
// exclusive scan ON THE DEVICE turns the counts into offsets; the host reads back ONE line (total, cells not staged,
// wide labels seen) to size the output; COPY moves every cell's records from the staging area to its offset, adding
// the coordinates in array-axis order -- no atomics, no second look at the volume.  A cell whose records did not fit
// its region, or that holds a voxel with more than four neighbour labels, is NOT staged but listed: EMIT, the same
// code with one listed cell per wave, recomputes exactly those cells and stores their records where they belong (a few
// cells around points where five cells meet; noise volumes; a staging area sized for tissue, half a record per voxel).
#include "ta_kernels.h"
#include "ta_sweep_common.h"

namespace ta {

namespace {

constexpr int WNJ = 4;                 // columns per lane
constexpr int WSC = 64 * WNJ;          // columns per strip
constexpr int WRING = 3;               // row slots per plane: above, current, below
constexpr uint32_t WALL_NONE = 0xFFFFFFFFu;

struct WallQuad { uint32_t v[WNJ]; };

__device__ __forceinline__ uint32_t lane_shl1(uint32_t src, uint32_t lane63_value) {
    // lane i <- src of lane i+1 ; lane 63 keeps lane63_value   (DPP wave_shl:1)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane63_value, (int)src, 0x130, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }
__device__ __forceinline__ int32_t smax3(uint32_t a, uint32_t b, uint32_t c) { return max(max((int32_t)a, (int32_t)b), (int32_t)c); }

// inclusive add-scan over the 64 lanes (row_shr 1,2,4,8 then the two row broadcasts)
__device__ __forceinline__ uint32_t wall_scan_add(uint32_t x) {
#define TA_DPP_ADD(ctrl, rmask) \
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xf, false);
    TA_DPP_ADD(0x111, 0xf) TA_DPP_ADD(0x112, 0xf) TA_DPP_ADD(0x114, 0xf) TA_DPP_ADD(0x118, 0xf)
    TA_DPP_ADD(0x142, 0xa) TA_DPP_ADD(0x143, 0xc)
#undef TA_DPP_ADD
    return x;
}

// One row of the window: lane i takes the 4 consecutive columns colq .. + 3 -- ONE 8- or 16-byte load where rows are
// aligned for it (QUADS: a lane is then wholly inside the row or wholly past its end, and a lane past the end reads the
// row's last quad and keeps its last voxel four times) -- clamped to the last column of the row.  In two halves: the
// LOAD leaves what memory returned in WallRaw, untouched, so that nothing waits for it; SETTLE, one step later, makes
// the four labels of it.
struct WallRaw { uint32_t r[WNJ]; };

template <typename T, bool QUADS>
__device__ __forceinline__ void wall_load_raw(WallRaw& n, const T* row, uint32_t colq, uint32_t last) {
    if (QUADS) {
        const uint32_t cl = colq <= last ? colq : last - (WNJ - 1);
        if (sizeof(T) == 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(row + cl);
            n.r[0] = q.x; n.r[1] = q.y; n.r[2] = q.z; n.r[3] = q.w;
        } else {
            const uint2 q = *reinterpret_cast<const uint2*>(row + cl);
            n.r[0] = q.x; n.r[1] = q.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < WNJ; ++j) {
            const uint32_t c = colq + (uint32_t)j;
            n.r[j] = (uint32_t)row[c <= last ? c : last];
        }
    }
}

template <typename T, bool QUADS>
__device__ __forceinline__ void wall_settle(WallQuad& q, const WallRaw& n, bool partial, bool past) {
    if (QUADS && sizeof(T) == 2) {
        q.v[0] = n.r[0] & 0xffffu; q.v[1] = n.r[0] >> 16; q.v[2] = n.r[1] & 0xffffu; q.v[3] = n.r[1] >> 16;
    } else {
#pragma unroll
        for (int j = 0; j < WNJ; ++j) q.v[j] = n.r[j];
    }
    if (QUADS && partial) {
        q.v[0] = past ? q.v[3] : q.v[0]; q.v[1] = past ? q.v[3] : q.v[1]; q.v[2] = past ? q.v[3] : q.v[2];
    }
}

// the smallest t above d: d + 1 + min_q((t_q - d - 1) mod 2^32) -- a t at or below d wraps to the top, above every t
// that does not -- 18 subtracts and a min3 tree, no compares
__device__ __forceinline__ uint32_t wall_next_above(const uint32_t (&t)[18], uint32_t d) {
    const uint32_t e = d + 1u;
    uint32_t m = umin3(umin3(t[0] - e, t[1] - e, t[2] - e), umin3(t[3] - e, t[4] - e, t[5] - e),
                       umin3(t[6] - e, t[7] - e, t[8] - e));
    m = umin3(m, umin3(t[9] - e, t[10] - e, t[11] - e), umin3(t[12] - e, t[13] - e, t[14] - e));
    m = umin3(m, t[15] - e, umin3(t[16] - e, t[17] - e, 0xFFFFFFFFu));
    return m + e;
}

}  // namespace

// a record in the staging area, as COUNT holds it: the voxel's label, t = (the other label XOR it) - 1, its column in the
// cell's strip; COPY, where every lane has a record, makes (lo, hi) of it
struct WallStaged { uint32_t own, t, column; };
// uint16 volumes: label and t fit 16 bits each (a staged t is never the wrapped one) -- 8 bytes a record instead of 12
struct WallStagedNarrow { uint32_t own_t, column; };

struct WallArgs {
    const void* vol;
    int64_t n0, n1, n2;
    int32_t nstrips, rows_per_wave;
    uint32_t* counts;              // [cells] records of each cell: written by COUNT
    uint32_t* cell_base;           // [cells] first staged record of the cell, WALL_NONE = not staged: written by COUNT
    uint8_t* lane_counts;          // [cells][64] records of each lane, cells that are not staged only
    const uint64_t* offsets;       // COPY / EMIT: exclusive scan of counts
    void* stage;                   // [WALL_CURSORS * region] WallStaged (uint32 volumes) / WallStagedNarrow (uint16)
    uint32_t* cursors;             // [WALL_CURSORS][32] (one 128-byte line each): records taken from each region
    uint32_t region;               // records per region
    uint32_t* status;              // [0] cells not staged  [1] a label >= 2^31 was seen by a kernel that cannot take it  [2] OR of all labels
    uint32_t* todo;                // [cells] the cells that are not staged, in no order: written by COUNT, EMIT takes one per wave
    uint32_t ntodo;                // EMIT: how many
    uint2* out_pairs;              // [n] (lo, hi)
    int32_t* out_coords;           // [n][3], array-axis order
    int32_t inv[3];                // inv[i] = memory axis of array axis i
    // the grouped fetch (kernels_wallsort.hip): key_bits != 0 -> COPY / EMIT write SORT KEYS (lo << key_bits | hi: 4 bytes where two
    // labels fit 32 bits, else 8) to `out_pairs` and the voxel's memory-order linear index (u32) to `out_coords` instead of records
    uint32_t key_bits;
};
// a record as the grouped fetch wants it (see WallArgs::key_bits)
__device__ __forceinline__ void wall_put_key(const WallArgs& A, uint64_t at, uint32_t lo, uint32_t hi, int32_t ma, int32_t mb, int32_t mc) {
    const uint64_t key = ((uint64_t)lo << A.key_bits) | hi;
    if (2u * A.key_bits <= 32u) reinterpret_cast<uint32_t*>(A.out_pairs)[at] = (uint32_t)key;
    else                        reinterpret_cast<uint64_t*>(A.out_pairs)[at] = key;
    reinterpret_cast<uint32_t*>(A.out_coords)[at] = (uint32_t)(((uint64_t)ma * (uint64_t)A.n1 + (uint64_t)mb) * (uint64_t)A.n2 + (uint64_t)mc);
}

constexpr int WALL_CURSORS = 256;
constexpr uint32_t WALL_COPY_CELLS = 16;
constexpr int WALL_KEPT = 4;           // labels of a voxel COUNT keeps in registers; a voxel with more leaves its cell to EMIT
constexpr uint32_t WALL_PIECE = 512;   // records a wave takes from its region at a time (a cell of tissue holds ~60)

// T: voxel type.  QUADS: rows start 4-element aligned and hold a multiple of 4 columns.  WIDE: labels may reach 2^31
// (the signed max is replaced by XOR + unsigned max: 54 instructions per voxel instead of 36).  EMIT: the second walk.
template <typename T, bool QUADS, bool WIDE, bool EMIT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) wall_cells_kernel(WallArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t wi = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t a, b0, b1;
    int32_t s;
    if (EMIT) {                        // one listed cell per wave
        if (wi >= (int64_t)A.ntodo) return;
        const uint32_t listed = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.todo[wi]);
        const uint32_t row = listed / (uint32_t)A.nstrips;
        s = (int32_t)(listed - row * (uint32_t)A.nstrips);
        a = (int64_t)(row / (uint32_t)A.n1);
        b0 = (int64_t)row - a * A.n1;
        b1 = b0 + 1;
    } else {                           // rows_per_wave cells, down axis 1
        const int64_t chunks_b = (A.n1 + A.rows_per_wave - 1) / A.rows_per_wave;
        const int64_t per_plane = chunks_b * A.nstrips;
        if (wi >= A.n0 * per_plane) return;
        a = wi / per_plane;
        const int64_t rem = wi - a * per_plane, cb = rem / A.nstrips;
        s = (int32_t)(rem - cb * A.nstrips);
        b0 = cb * A.rows_per_wave;
        b1 = b0 + A.rows_per_wave < A.n1 ? b0 + A.rows_per_wave : A.n1;
    }
    const uint32_t c0 = (uint32_t)s * WSC, last = (uint32_t)(A.n2 - 1);
    const uint32_t colq = c0 + (uint32_t)WNJ * (uint32_t)lane;
    const bool partial = c0 + (uint32_t)(WSC - 1) > last;          // the strip reaches past the end of the rows
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

    // the window: W[p][k] = the lane's four columns of one row of plane p, a ring of three slots per plane (above, the row,
    // below); H[p][k] = the two columns beside the strip, the left one in lanes 0 - 31 and the right one in lanes 32 - 63
    // (one more small load per row: lane 0 / lane 63 is where the DPP shifts want them); U3 / R3[k] = "the three rows of
    // slot k, those columns included, are all one label".  N / HN: the rows in flight -- loaded during one step, settled
    // into the slot of the row that is no longer needed at the top of the next.
    WallQuad W[3][WRING];
    uint32_t H[3][WRING], R3[WRING];
    bool U3[WRING];
    WallRaw N[3];
    uint32_t HN[3];
    uint32_t seen = 0;                 // OR of every label of the wave's own plane (WIDE detection; how many bits a label takes)
    const uint32_t hcol = lane < 32 ? col_left : col_right;
    const bool lane_past = colq > last;
    auto load_rows = [&](const T* const (&rows)[3], WallRaw (&n)[3], uint32_t (&hn)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            wall_load_raw<T, QUADS>(n[p], rows[p], colq, last);
            hn[p] = (uint32_t)rows[p][hcol];
        }
    };
    auto settle = [&](const WallRaw (&n)[3], const uint32_t (&hn)[3], const int k) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            wall_settle<T, QUADS>(W[p][k], n[p], partial, lane_past);
            H[p][k] = hn[p];
        }
        const uint32_t ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)W[1][k].v[0]);
        uint32_t d = 0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const WallQuad& q = W[p][k];
            d |= ((q.v[0] ^ ref) | (q.v[1] ^ ref)) | ((q.v[2] ^ ref) | (q.v[3] ^ ref)) | (H[p][k] ^ ref);
            if (p == 1 && !EMIT) seen |= (q.v[0] | q.v[1]) | (q.v[2] | q.v[3]);
        }
        R3[k] = ref;
        U3[k] = !__any(d != 0u);
    };
    const T* ahead[3];                 // the row the next load takes, per plane
    {
        const T* r0[3] = {rowp(0, b0 - 1), rowp(1, b0 - 1), rowp(2, b0 - 1)};
        const T* r1[3] = {rowp(0, b0), rowp(1, b0), rowp(2, b0)};
        const T* r2[3] = {rowp(0, b0 + 1), rowp(1, b0 + 1), rowp(2, b0 + 1)};
        WallRaw n0[3], n1[3];
        uint32_t h0[3], h1[3];
        load_rows(r0, n0, h0); load_rows(r1, n1, h1); load_rows(r2, N, HN);
        settle(n0, h0, 0); settle(n1, h1, 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) ahead[p] = r2[p];
    }
    const uint64_t cell0 = (uint64_t)((a * A.n1 + b0) * A.nstrips + s);
    uint64_t cell = cell0;
    const uint32_t region_of_wave = ((uint32_t)wi * 0x9E3779B1u) >> 24;          // 0 .. WALL_CURSORS - 1, scattered
    // COUNT keeps what it finds in the wave's slice of LDS and in two registers (lane r: the cell of the wave's r-th row)
    // and writes to memory when a piece is full and when the walk is over: a step then issues no store, and the wait for
    // the rows it prefetched is not a wait for the stores before them (one counter for both, in issue order).
    __shared__ uint32_t piece_lds[EMIT ? 1 : 4 * WALL_PIECE * 3];
    uint32_t* const piece = piece_lds + (EMIT ? 0u : (threadIdx.x >> 6) * (WALL_PIECE * 3u));
    uint32_t piece_at = 0, piece_used = 0, piece_size = 0;     // where the piece lies in the staging area, records in it, its size
    uint32_t cell_counts = 0, cell_firsts = WALL_NONE;
    auto flush_piece = [&]() {
        for (uint32_t r = (uint32_t)lane; r < piece_used; r += 64u) {
            if (sizeof(T) == 2) {
                WallStagedNarrow rec;
                rec.own_t = piece[r]; rec.column = piece[WALL_PIECE + r];
                reinterpret_cast<WallStagedNarrow*>(A.stage)[piece_at + r] = rec;
            } else {
                WallStaged rec;
                rec.own = piece[r]; rec.t = piece[WALL_PIECE + r]; rec.column = piece[2u * WALL_PIECE + r];
                reinterpret_cast<WallStaged*>(A.stage)[piece_at + r] = rec;
            }
        }
    };
    const int32_t ma = (int32_t)a;
    struct __attribute__((packed, aligned(4))) Int3 { int32_t x, y, z; };

    // one step: the cell of row b; ring slots PH = above, PH + 1 = the row, PH + 2 = below
    auto step = [&](auto ph, const int64_t b) {
        constexpr int PH = decltype(ph)::value;
        constexpr int KA = PH % WRING, KC = (PH + 1) % WRING, KB = (PH + 2) % WRING;
        settle(N, HN, KB);                               // the rows loaded during the previous step: first use
        if (b + 1 < b1) {
            const int64_t adv = b + 2 < A.n1 ? A.n2 : 0;
#pragma unroll
            for (int p = 0; p < 3; ++p) ahead[p] += adv;
            load_rows(ahead, N, HN);
        }
        const uint64_t this_cell = cell;
        cell += (uint64_t)A.nstrips;
        uint64_t base = 0;
        if (EMIT) {
            base = A.offsets[this_cell];
        } else {
            // the whole 3 x 3 x (strip + 2) block one label: nothing to do (background, cell interiors)
            if (U3[KA] && U3[KC] && U3[KB] && R3[KA] == R3[KC] && R3[KB] == R3[KC]) return;
        }
        // the columns beside each lane's four, for the five rows whose column neighbours count (the row itself and its
        // four face rows): from the next lane through DPP, at the ends of the strip what lane 0 / lane 63 of H hold
        const WallQuad* const side[5] = {&W[1][KC], &W[0][KC], &W[2][KC], &W[1][KA], &W[1][KB]};
        const uint32_t beside[5] = {H[1][KC], H[0][KC], H[2][KC], H[1][KA], H[1][KB]};
        uint32_t Lc[5], Rc[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            Lc[f] = lane_shr1(side[f]->v[WNJ - 1], beside[f]);
            Rc[f] = lane_shl1(side[f]->v[0], beside[f]);
        }
        const WallQuad& e0 = W[0][KA]; const WallQuad& e1 = W[0][KB]; const WallQuad& e2 = W[2][KA]; const WallQuad& e3 = W[2][KB];
        const bool past = partial && colq > last;         // QUADS: a lane is wholly in or wholly out

        uint32_t pos = 0;                       // EMIT: relative to the cell's first record
        uint2* out_pairs = nullptr;
        int32_t* out_coords = nullptr;
        if (EMIT) {
            const uint32_t mine = A.lane_counts[this_cell * 64 + lane];
            pos = wall_scan_add(mine) - mine;
            out_pairs = A.out_pairs + base;
            out_coords = A.out_coords + 3 * base;
        }
        const uint64_t emit_base = EMIT ? base : 0ull;
        uint32_t K[WNJ][WALL_KEPT];             // COUNT: the first four labels of each column, as t = (label XOR v) - 1
        uint32_t nn = 0;                        // records of column j in byte j
        bool many = false;                      // a voxel of this lane has more than four labels
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < WNJ; ++j) {
            const uint32_t v = W[1][KC].v[j];
            uint32_t t[18];
            {
                uint32_t nb[18];
                int q = 0;
#pragma unroll
                for (int f = 0; f < 5; ++f) {
                    nb[q++] = j > 0 ? side[f]->v[j > 0 ? j - 1 : 0] : Lc[f];
                    if (f > 0) nb[q++] = side[f]->v[j];
                    nb[q++] = j < WNJ - 1 ? side[f]->v[j < WNJ - 1 ? j + 1 : j] : Rc[f];
                }
                nb[q++] = e0.v[j]; nb[q++] = e1.v[j]; nb[q++] = e2.v[j]; nb[q++] = e3.v[j];
#pragma unroll
                for (int i = 0; i < 18; ++i) t[i] = (nb[i] ^ v) + 0xFFFFFFFFu;
            }
            uint32_t tmin = umin3(umin3(t[0], t[1], t[2]), umin3(t[3], t[4], t[5]), umin3(t[6], t[7], t[8]));
            tmin = umin3(tmin, umin3(t[9], t[10], t[11]), umin3(t[12], t[13], t[14]));
            tmin = umin3(tmin, t[15], umin3(t[16], t[17], 0xFFFFFFFFu));
            uint32_t tmax;                       // the largest t of a neighbour with another label, WALL_NONE without one
            if (!WIDE) {
                int32_t m = smax3((uint32_t)smax3(t[0], t[1], t[2]), (uint32_t)smax3(t[3], t[4], t[5]), (uint32_t)smax3(t[6], t[7], t[8]));
                m = smax3((uint32_t)m, (uint32_t)smax3(t[9], t[10], t[11]), (uint32_t)smax3(t[12], t[13], t[14]));
                m = smax3((uint32_t)m, (uint32_t)smax3(t[15], t[16], t[17]), 0xFFFFFFFFu);
                tmax = (uint32_t)m;
            } else {
                uint32_t m = umax3(umax3(t[0] + 1u, t[1] + 1u, t[2] + 1u), umax3(t[3] + 1u, t[4] + 1u, t[5] + 1u),
                                   umax3(t[6] + 1u, t[7] + 1u, t[8] + 1u));
                m = umax3(m, umax3(t[9] + 1u, t[10] + 1u, t[11] + 1u), umax3(t[12] + 1u, t[13] + 1u, t[14] + 1u));
                m = umax3(m, umax3(t[15] + 1u, t[16] + 1u, t[17] + 1u), 0u);
                tmax = m - 1u;
            }
            if (QUADS ? partial : true) {       // (a strip that ends inside the rows only: the test is per wave)
                if (QUADS ? past : colq + (uint32_t)j > last) { tmin = WALL_NONE; tmax = WALL_NONE; }   // a column past the end of the row (clamped copies)
            }
            if (!EMIT) K[j][0] = tmin;          // (the others are read only where n says they were found)
            if (!__any(tmax != WALL_NONE)) continue;
            Int3 xyz;                                          // the voxel in array-axis order: one 12-byte store per record
            if (EMIT) {
                const int32_t mb = (int32_t)b, mc = (int32_t)(colq + (uint32_t)j);
                xyz.x = A.inv[0] == 0 ? ma : (A.inv[0] == 1 ? mb : mc);
                xyz.y = A.inv[1] == 0 ? ma : (A.inv[1] == 1 ? mb : mc);
                xyz.z = A.inv[2] == 0 ? ma : (A.inv[2] == 1 ? mb : mc);
            }
            auto put = [&](uint32_t td) {
                const uint32_t m = v ^ (td + 1u);
                if (A.key_bits) {
                    wall_put_key(A, emit_base + pos, v < m ? v : m, v < m ? m : v, (int32_t)ma, (int32_t)b, (int32_t)(colq + (uint32_t)j));
                } else {
                    out_pairs[pos] = make_uint2(v < m ? v : m, v < m ? m : v);
                    *reinterpret_cast<Int3*>(out_coords + 3u * pos) = xyz;
                }
                ++pos;
            };
            // distinct t in increasing order, one per round.  The first one is tmin; where it equals the largest there is
            // one label and the voxel is done -- most wall voxels.  Further rounds run only while some lane of the wave
            // still has labels between its last one and its largest.  (Unsigned compares: a lane without any holds
            // WALL_NONE in both.)
            uint32_t d = tmin, n = tmax != WALL_NONE ? 1u : 0u;
            if (EMIT && n) put(d);
            if (__any(d < tmax)) {
                const uint32_t nd = wall_next_above(t, d);
                const bool more = d < tmax;
                if (EMIT) { if (more) put(nd); } else K[j][1] = nd;          // (K: read only where n says so)
                d = more ? nd : d; n += more ? 1u : 0u;
                if (__any(d < tmax)) {
                    const uint32_t nd2 = wall_next_above(t, d);
                    const bool more2 = d < tmax;
                    if (EMIT) { if (more2) put(nd2); } else K[j][2] = nd2;
                    d = more2 ? nd2 : d; n += more2 ? 1u : 0u;
                    // (at most 18 labels: the bound also ends the walk of the kernel without WIDE over labels from 2^31 up,
                    // whose signed maximum need not be one of the t -- its counts are thrown away, the host runs WIDE)
                    for (int round = 3; round < 19 && __any(d < tmax); ++round) {
                        const uint32_t nd3 = wall_next_above(t, d);
                        const bool more3 = d < tmax;
                        if (EMIT) { if (more3) put(nd3); } else if (round == 3) K[j][3] = nd3;
                        d = more3 ? nd3 : d; n += more3 ? 1u : 0u; many = many || (more3 && round > 3);
                    }
                }
            }
            total += n;
            nn |= n << (8 * j);
        }
        if (EMIT) return;
        const uint32_t incl = wall_scan_add(total);
        const uint32_t cell_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (cell_total == 0u) return;
        // room in the staging area: in the piece this wave holds; a new piece (one returning atomic on this wave's cursor)
        // when the cell does not fit what is left of it, the full one goes to memory first
        uint32_t first = WALL_NONE;
        if (!__any(many) && cell_total <= WALL_PIECE) {
            if (cell_total > piece_size - piece_used) {
                flush_piece();
                piece_used = 0; piece_size = 0;
                uint32_t got = 0;
                if (lane == 0) got = __hip_atomic_fetch_add(A.cursors + region_of_wave * 32u, WALL_PIECE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                if (got <= A.region && WALL_PIECE <= A.region - got) { piece_at = region_of_wave * A.region + got; piece_size = WALL_PIECE; }
            }
            if (cell_total <= piece_size - piece_used) first = piece_at + piece_used;
        }
        const int rth = (int)(b - b0);
        cell_counts = lane == rth ? cell_total : cell_counts;
        cell_firsts = lane == rth ? first : cell_firsts;
        if (first == WALL_NONE) {
            A.lane_counts[this_cell * 64 + lane] = (uint8_t)total;          // <= 4 x 18
            if (lane == 0) A.todo[atomicAdd(A.status, 1u)] = (uint32_t)this_cell;
            return;
        }
        uint32_t at = piece_used + incl - total;
        piece_used += cell_total;
        const bool several = __any((nn & 0xFEFEFEFEu) != 0u);          // some voxel of the cell has more than one label
        auto keep = [&](const uint32_t where, const uint32_t v, const uint32_t tk, const uint32_t column) {
            if (sizeof(T) == 2) { piece[where] = v | (tk << 16); piece[WALL_PIECE + where] = column; }
            else { piece[where] = v; piece[WALL_PIECE + where] = tk; piece[2u * WALL_PIECE + where] = column; }    // (one address, three offsets)
        };
#pragma unroll
        for (int j = 0; j < WNJ; ++j) {
            const uint32_t v = W[1][KC].v[j], nj = (nn >> (8 * j)) & 0xffu, column = (uint32_t)(WNJ * lane + j);
            if (nj > 0u) keep(at, v, K[j][0], column);
            if (several) {
#pragma unroll
                for (int i = 1; i < WALL_KEPT; ++i) {
                    if (!__any(nj > (uint32_t)i)) break;
                    if (nj > (uint32_t)i) keep(at + (uint32_t)i, v, K[j][i], column);
                }
            }
            at += nj;
        }
    };

    for (int64_t b = b0; b < b1; b += WRING) {
        step(std::integral_constant<int, 0>(), b);
        if (b + 1 < b1) step(std::integral_constant<int, 1>(), b + 1);
        if (b + 2 < b1) step(std::integral_constant<int, 2>(), b + 2);
    }
    if (!EMIT) {
        flush_piece();
        if ((int64_t)lane < b1 - b0) {
            const uint64_t mine = cell0 + (uint64_t)lane * (uint64_t)A.nstrips;
            A.counts[mine] = cell_counts;
            A.cell_base[mine] = cell_firsts;
        }
        if (sizeof(T) == 4 && !WIDE) {
            if (__any((int32_t)seen < 0) && lane == 0) A.status[1] = 1u;
        }
        {   // the bits labels occupy (the grouping sort packs its keys by it): only lanes that bring a new bit touch the word
            const uint32_t known = *reinterpret_cast<const volatile uint32_t*>(A.status + 2);
            if (seen & ~known) atomicOr(A.status + 2, seen);
        }
    }
}

// COPY: a wave takes 16 cells -- one per lane for the bookkeeping, then four cells at a time with all lanes on their records
template <bool NARROW>
__global__ void __launch_bounds__(256) wall_copy_kernel(WallArgs A, uint32_t ncells) {
    const int lane = threadIdx.x & 63;
    const uint32_t first_cell = (blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * WALL_COPY_CELLS;
    if (first_cell >= ncells) return;
    const uint32_t cell = first_cell + (uint32_t)lane;
    uint32_t cnt = 0, from = WALL_NONE;
    uint64_t to = 0;
    if (lane < (int)WALL_COPY_CELLS && cell < ncells) { cnt = A.counts[cell]; from = A.cell_base[cell]; to = A.offsets[cell]; }
    if (cnt == 0u || from == WALL_NONE) cnt = 0u;
    const uint32_t row = cell / (uint32_t)A.nstrips, strip = cell - row * (uint32_t)A.nstrips;
    const uint32_t pa = row / (uint32_t)A.n1, pb = row - pa * (uint32_t)A.n1;
    struct __attribute__((packed, aligned(4))) Int3 { int32_t x, y, z; };
    struct Cell { uint32_t n, src, c0; int32_t ma, mb; uint64_t dst; };
    auto take = [&](const int L) {
        Cell c;
        c.n = (uint32_t)__builtin_amdgcn_readlane((int)cnt, L);
        c.src = (uint32_t)__builtin_amdgcn_readlane((int)from, L);
        c.dst = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(to >> 32), L) << 32) |
                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)to, L);
        c.ma = __builtin_amdgcn_readlane((int)pa, L);
        c.mb = __builtin_amdgcn_readlane((int)pb, L);
        c.c0 = (uint32_t)__builtin_amdgcn_readlane((int)strip, L) * WSC;
        return c;
    };
    auto staged = [&](const uint32_t at) {
        WallStaged rec;
        if (NARROW) {
            const WallStagedNarrow n = reinterpret_cast<const WallStagedNarrow*>(A.stage)[at];
            rec.own = n.own_t & 0xffffu; rec.t = n.own_t >> 16; rec.column = n.column;
        } else {
            rec = reinterpret_cast<const WallStaged*>(A.stage)[at];
        }
        return rec;
    };
    auto put = [&](const Cell& c, const uint32_t r, const WallStaged& rec) {
        const int32_t mc = (int32_t)(c.c0 + rec.column);
        Int3 xyz;
        xyz.x = A.inv[0] == 0 ? c.ma : (A.inv[0] == 1 ? c.mb : mc);
        xyz.y = A.inv[1] == 0 ? c.ma : (A.inv[1] == 1 ? c.mb : mc);
        xyz.z = A.inv[2] == 0 ? c.ma : (A.inv[2] == 1 ? c.mb : mc);
        const uint32_t other = rec.own ^ (rec.t + 1u);
        if (A.key_bits) { wall_put_key(A, c.dst + r, rec.own < other ? rec.own : other, rec.own < other ? other : rec.own, c.ma, c.mb, mc); return; }
        A.out_pairs[c.dst + r] = make_uint2(rec.own < other ? rec.own : other, rec.own < other ? other : rec.own);
        *reinterpret_cast<Int3*>(A.out_coords + 3 * (c.dst + r)) = xyz;
    };
    uint64_t todo = __ballot(cnt != 0u);
    while (todo) {
        // four cells at a time: their loads are in flight together (a cell of tissue holds ~60 records: one per lane)
        constexpr int DEPTH = 4;
        Cell c[DEPTH];
        WallStaged rec[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            c[u].n = 0u;
            if (todo) {
                c[u] = take(__builtin_ctzll(todo));
                todo &= todo - 1;
            }
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            if ((uint32_t)lane < c[u].n) rec[u] = staged(c[u].src + (uint32_t)lane);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            if ((uint32_t)lane < c[u].n) put(c[u], (uint32_t)lane, rec[u]);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            for (uint32_t r = (uint32_t)lane + 64u; r < c[u].n; r += 64u) put(c[u], r, staged(c[u].src + r));
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

uint64_t* scan_u32_total(void* scratch, uint64_t n) { return (uint64_t*)scratch + (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK; }

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
static int wall_rows_per_wave(int64_t n1) { return n1 < 16 ? (int)(n1 > 0 ? n1 : 1) : 16; }     // (8, 32, 64 rows: slower on C2)

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

uint64_t wall_stage_bytes(uint64_t records_per_region, int itemsize) {
    return (uint64_t)WALL_CURSORS * records_per_region * (itemsize == 2 ? sizeof(WallStagedNarrow) : sizeof(WallStaged));
}
uint64_t wall_cursor_bytes() { return (uint64_t)WALL_CURSORS * 128; }
uint32_t wall_stage_regions() { return WALL_CURSORS; }

static WallArgs wall_args(const void* vol, int64_t n0, int64_t n1, int64_t n2, const WallPlan& p, const WallBuffers& b) {
    WallArgs a;
    a.vol = vol; a.n0 = n0; a.n1 = n1; a.n2 = n2;
    a.nstrips = p.nstrips; a.rows_per_wave = p.rows_per_wave;
    a.counts = b.counts; a.cell_base = b.cell_base; a.lane_counts = b.lane_counts; a.offsets = b.offsets;
    a.stage = b.stage; a.cursors = b.cursors; a.region = b.stage ? b.region : 0u; a.status = b.status;
    a.todo = b.todo; a.ntodo = 0;
    a.out_pairs = nullptr; a.out_coords = nullptr; a.key_bits = 0u;
    a.inv[0] = 0; a.inv[1] = 1; a.inv[2] = 2;
    return a;
}

template <bool EMIT>
static void launch_wall_cells(hipStream_t s, const WallArgs& a, int itemsize, bool wide, unsigned blocks) {
    const bool quads = ((uintptr_t)a.vol % (uintptr_t)(WNJ * itemsize) == 0) && (a.n2 % WNJ == 0);
#define TA_WALL_GO(T, Q, W) hipLaunchKernelGGL((wall_cells_kernel<T, Q, W, EMIT>), dim3(blocks), dim3(256), 0, s, a)
    if (itemsize == 2) { if (quads) TA_WALL_GO(uint16_t, true, false); else TA_WALL_GO(uint16_t, false, false); }
    else if (!wide)    { if (quads) TA_WALL_GO(uint32_t, true, false); else TA_WALL_GO(uint32_t, false, false); }
    else               { if (quads) TA_WALL_GO(uint32_t, true, true);  else TA_WALL_GO(uint32_t, false, true); }
#undef TA_WALL_GO
}

void launch_wall_count(hipStream_t s, const void* vol, int itemsize, int64_t n0, int64_t n1, int64_t n2, const WallBuffers& b,
                       bool wide) {
    const WallPlan p = wall_plan(n0, n1, n2);
    // total | status[2] | cursors: one clear (the layout of ta_api.hip's wall_bufs keeps them together)
    (void)hipMemsetAsync(b.total, 0, (size_t)((char*)b.cursors - (char*)b.total) + (p.cells ? wall_cursor_bytes() : 0), s);
    if (p.cells == 0) return;
    const WallArgs a = wall_args(vol, n0, n1, n2, p, b);
    launch_wall_cells<false>(s, a, itemsize, wide, (unsigned)((p.waves + 3) / 4));
    hipLaunchKernelGGL(wall_scan_sums_kernel, dim3((unsigned)p.scan_blocks), dim3(256), 0, s, b.counts, p.cells, b.block_sums);
    hipLaunchKernelGGL(wall_scan_top_kernel, dim3(1), dim3(256), 0, s, b.block_sums, p.scan_blocks, b.total);
    hipLaunchKernelGGL(wall_scan_apply_kernel, dim3((unsigned)p.scan_blocks), dim3(256), 0, s, b.counts, p.cells, b.block_sums, b.offsets);
}

void launch_wall_fetch(hipStream_t s, const void* vol, int itemsize, int64_t n0, int64_t n1, int64_t n2, const WallBuffers& b,
                       bool wide, uint32_t not_staged, uint32_t* out_pairs, int32_t* out_coords, const int perm[3], int key_bits) {
    const WallPlan p = wall_plan(n0, n1, n2);
    if (p.cells == 0) return;
    WallArgs a = wall_args(vol, n0, n1, n2, p, b);
    a.out_pairs = (uint2*)out_pairs; a.out_coords = out_coords; a.key_bits = (uint32_t)key_bits;
    for (int k = 0; k < 3; ++k) a.inv[perm[k]] = k;          // perm[k] = array axis of memory axis k
    if (a.region) {
        const dim3 grid((unsigned)((p.cells + 4 * WALL_COPY_CELLS - 1) / (4 * WALL_COPY_CELLS)));
        if (itemsize == 2) hipLaunchKernelGGL(wall_copy_kernel<true>, grid, dim3(256), 0, s, a, (uint32_t)p.cells);
        else               hipLaunchKernelGGL(wall_copy_kernel<false>, grid, dim3(256), 0, s, a, (uint32_t)p.cells);
    }
    a.ntodo = not_staged;
    if (not_staged) launch_wall_cells<true>(s, a, itemsize, wide, (not_staged + 3u) / 4u);
}

}  // namespace ta
