// kernels_rle.hip -- adjacency from a run-length encoding of the volume (TA_OPT_IMPL = 5).
//
// The row-run sweep (kernels_rowrun.hip) already cuts every row of every plane into runs of equal label
// to sum the moments.  Here it also KEEPS them: the same 64-record consumer passes copy the records
// {closing label, voxel to the right, c0 | n | b | a} to a per-wave-tile region in HBM, and a per-row
// directory {first record, count, label of an end-to-end uniform row} is staged in LDS and written at the
// end of the tile.  That is a run-length encoding of the volume, ~13x smaller than the volume on tissue
// (3.9 % of the voxels carry a record).  The adjacency is then computed FROM THE RUNS by a second kernel,
// with work proportional to the runs instead of the voxels:
//   axis 2: a record's two labels are a face; across a tile's left edge: the last run of the row in the tile
//           to the left against the run that starts at column 0;
//   axis 1: a run of row (a, b) against the runs of row (a, b-1): every overlap of [c0, c0+n) with a run of
//           another label is `overlap` faces of that pair (a uniform row is one run of the whole segment);
//   axis 0: the same against row (a-1, b).
// The streaming kernel carries no previous-plane / row-above registers, no face compares, no face records:
// it is the moments-only kernel (60 % of the HBM roofline) plus ~1 store per consumer pass.
#include "ta_sweep_common.h"

namespace ta {

#ifndef TA_RLE_RQCAP
#define TA_RLE_RQCAP 128
#endif
constexpr int RLE_RQCAP = TA_RLE_RQCAP;
constexpr uint32_t NOLABEL = INVALID_LABEL;

struct __attribute__((aligned(16))) RleWaveLds {
    uint32_t cqv[RLE_RQCAP], cqp[RLE_RQCAP], cqc[RLE_RQCAP];   // right voxel, closing label, c0 | n << 9 | b << 19 | a << 23
};

template <int NW, int RB>
struct __attribute__((aligned(16))) RleLds {
    RleWaveLds wave[WAVES];
    uint4 dir[WAVES][MAX_TILE_PLANES * RB];                     // per row: first record, records, uniform label, -
    uint64_t lsum[LSLOTS * NW];
    uint32_t lbox[LSLOTS * 8];
    uint32_t lkeys[LSLOTS];
    uint64_t pkeys[1];                                          // (no pair table here; members the shared helpers name)
    uint32_t pcnt[3];
};

// drains complete groups of 64 records: moments into the label table, a copy into the wave tile's region
template <bool MOM2, typename LDS>
__device__ __forceinline__ void consume_rle_ring(const SweepArgs& A, LDS& S, const TileFrame& F, int w, int lane,
                                                 int& chead, int ctail, bool all, uint32_t* __restrict__ region,
                                                 uint32_t rcap) {
    auto& W = S.wave[w];
    for (;;) {
        const int cnt = ctail - chead;
        if (cnt < 64 && !(all && cnt > 0)) break;
        const int qi = (chead + lane) & (RLE_RQCAP - 1);
        const uint32_t v = W.cqv[qi], label = W.cqp[qi], code = W.cqc[qi];
        const bool act = lane < cnt;
        const uint32_t gi = (uint32_t)(chead + lane);
        chead += cnt < 64 ? cnt : 64;
        if (act) {
            if (gi < rcap) { reinterpret_cast<uint2*>(region)[gi] = make_uint2(label, code); region[2u * rcap + gi] = v; }
            consume_row_run<MOM2, LDS>(A, S, F, label, code);
        }
    }
}

template <typename T, int VPL, int RB, bool MOM2, typename LDS>
__device__ __forceinline__ void wave_rle(const SweepArgs& A, LDS& S, const bool EDGE, const int lane, const int w,
                                         const int64_t c_tile0, const int64_t b_tile0, const int64_t p_lo,
                                         const int64_t p_hi, uint32_t* __restrict__ region, const uint32_t rcap,
                                         uint4* __restrict__ gdir, uint32_t* __restrict__ ghdr) {
    constexpr int TC = 64 * VPL;
    static_assert(TC <= 512, "the run code holds c0 in 9 bits and n in 10");
    auto& W = S.wave[w];
    uint4* D = S.dir[w];

    const T* vol = reinterpret_cast<const T*>(A.vol);
    const int64_t n1 = A.n1, n2 = A.n2, plane = n1 * n2;
    const int64_t b_wave0 = b_tile0 + (int64_t)w * RB;
    const int64_t c0g = c_tile0 + (int64_t)lane * VPL;
    TileFrame F;
    F.A0 = (uint64_t)(A.a_origin + (p_lo - A.first_owned)); F.B0 = (uint64_t)b_tile0; F.C0 = (uint64_t)c_tile0;
    const uint32_t lane_c = (uint32_t)lane * VPL;
    const uint32_t lane_off = lane_c * (uint32_t)sizeof(T);

    uint32_t cur[RB][VPL], nxt[RB][VPL];

    auto load_rows = [&](int64_t p, uint32_t (&d)[RB][VPL]) {
        const T* pbase = vol + p * plane;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int64_t b = b_wave0 + r;
            const bool row_ok = b < n1;
            const T* row = pbase + (EDGE ? (row_ok ? b : 0) : b) * n2 + c_tile0;
            load_strip<T, VPL>(EDGE, row, row_ok, lane_off, c0g, n2, d[r]);
        }
    };

    for (int i = lane; i < MAX_TILE_PLANES * RB; i += 64) D[i] = make_uint4(0u, 0u, NOLABEL, 0u);
    load_rows(p_lo, cur);
    if (p_lo + 1 < p_hi) load_rows(p_lo + 1, nxt);

    int chead = 0, ctail = 0;                     // free-running cursors = record numbers inside the region
    uint32_t ulab = INVALID_LABEL, un = 0, ua = 0, ub = 0, uaa = 0, uab = 0, ubb = 0;
    uint32_t uamin = 0xffffffffu, uamax = 0, ubmin = 0xffffffffu, ubmax = 0;

    for (int64_t p = p_lo; p < p_hi; ++p) {
        const uint32_t ploc = (uint32_t)(p - p_lo);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const uint32_t bloc = (uint32_t)(w * RB + r);
            uint64_t mc[VPL];
            uint32_t pcv[VPL];
            int nrun = 0;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const uint32_t v = cur[r][j];
                pcv[j] = j > 0 ? cur[r][j > 0 ? j - 1 : 0] : lane_shr1(cur[r][VPL - 1], cur[r][0]);
                mc[j] = __builtin_amdgcn_ballot_w64(v != pcv[j]);
                nrun += __popcll(mc[j]);
            }
            uint64_t inrow = mc[0];                 // (lane 0 compares its first voxel with itself: never a boundary)
#pragma unroll
            for (int j = 1; j < VPL; ++j) inrow |= mc[j];
            const uint32_t rowlab = __builtin_amdgcn_readfirstlane(cur[r][0]);
            const bool uniform = inrow == 0ull;
            const bool outside = uniform && rowlab == INVALID_LABEL;
            const bool summed = uniform && !outside && (ulab == INVALID_LABEL || ulab == rowlab);
            if (summed) {
                ulab = rowlab; un += 1u; ua += ploc; ub += bloc; uaa += ploc * ploc; uab += ploc * bloc; ubb += bloc * bloc;
                uamin = ploc < uamin ? ploc : uamin; uamax = ploc > uamax ? ploc : uamax;
                ubmin = bloc < ubmin ? bloc : ubmin; ubmax = bloc > ubmax ? bloc : ubmax;
            }
            // a uniform row is ONE run of the whole segment for the adjacency kernel, whoever sums its moments
            const uint32_t dirlab = (uniform && !outside) ? rowlab : NOLABEL;
            const bool need_end = !summed && !outside;
            if (nrun == 0 && !need_end) {
                if (lane == 0) D[ploc * RB + r] = make_uint4((uint32_t)ctail, 0u, dirlab, 0u);
                continue;
            }
            uint32_t s;
            {
                uint32_t lastpos = 0u;
#pragma unroll
                for (int j = 0; j < VPL; ++j) lastpos = (cur[r][j] != pcv[j]) ? lane_c + (uint32_t)j : lastpos;
                s = lane_shr1(wave_scan_max(lastpos), 0u);
            }
            const int c_before = ctail;
            // A row's records are written SORTED by column: the slot of the boundary at (lane, j) is its rank among
            // the row's boundaries = boundaries in lower lanes (one mask-prefix per position, shared) + the lane's own
            // earlier ones.  A row with more boundaries than the ring can take at once (never on tissue) gives the
            // volume back to the fused sweep.
            if ((ctail - chead) + nrun + 1 > RLE_RQCAP) {
                if (lane == 0) atomicOr(&A.flags[FLAG_REGION_OVERFLOW], 1u);
            } else {
                uint32_t rank = 0u;
#pragma unroll
                for (int j = 0; j < VPL; ++j) rank += mbcnt64(mc[j]);
                uint32_t s_ = s;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const uint32_t v = cur[r][j];
                    const uint32_t k = lane_c + (uint32_t)j;
                    if (v != pcv[j]) {
                        const int q_ = (ctail + (int)rank) & (RLE_RQCAP - 1);
                        W.cqv[q_] = v; W.cqp[q_] = pcv[j];
                        W.cqc[q_] = s_ | ((k - s_) << 9) | (bloc << 19) | (ploc << 23);
                        rank += 1u;
                        s_ = k;
                    }
                }
                if (need_end && lane == 63) {
                    const int q_ = (ctail + nrun) & (RLE_RQCAP - 1);
                    W.cqv[q_] = INVALID_LABEL; W.cqp[q_] = cur[r][VPL - 1];
                    W.cqc[q_] = s_ | (((uint32_t)TC - s_) << 9) | (bloc << 19) | (ploc << 23);
                }
                ctail += nrun + (need_end ? 1 : 0);
                if (ctail - chead >= 64) {
                    __builtin_amdgcn_wave_barrier();
                    consume_rle_ring<MOM2, LDS>(A, S, F, w, lane, chead, ctail, false, region, rcap);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (lane == 0) D[ploc * RB + r] = make_uint4((uint32_t)c_before, (uint32_t)(ctail - c_before), dirlab, 0u);
        }
        if (p + 1 < p_hi) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int j = 0; j < VPL; ++j) cur[r][j] = nxt[r][j];
            }
            if (p + 2 < p_hi) load_rows(p + 2, nxt);
        }
    }

    // ---- end of tile: drain the ring, the uniform rows' moments in one closed form, the row directory
    __builtin_amdgcn_wave_barrier();
    consume_rle_ring<MOM2, LDS>(A, S, F, w, lane, chead, ctail, true, region, rcap);
    if (lane == 0 && ulab != INVALID_LABEL) {
        const uint64_t nc = TC, t1c = range_sum1(0, nc), t2c = range_sum2(0, nc);
        LocalSums L;
        L.n = (uint64_t)un * nc; L.sa = (uint64_t)ua * nc; L.sb = (uint64_t)ub * nc; L.sc = (uint64_t)un * t1c;
        if (MOM2) {
            L.saa = (uint64_t)uaa * nc; L.sab = (uint64_t)uab * nc; L.sbb = (uint64_t)ubb * nc;
            L.sac = (uint64_t)ua * t1c; L.sbc = (uint64_t)ub * t1c; L.scc = (uint64_t)un * t2c;
        } else {
            L.saa = L.sab = L.sac = L.sbb = L.sbc = L.scc = 0;
        }
        lds_label_add<MOM2, LDS, LocalSums>(A, S, F, ulab, L, uamin, uamax, ubmin, ubmax, 0u, (uint32_t)(nc - 1));
    }
    __builtin_amdgcn_wave_barrier();
    const int nrows = A.tile_planes * RB;
    for (int i = lane; i < nrows; i += 64) gdir[i] = D[i];
    if (lane == 0) {
        *ghdr = (uint32_t)ctail;
        if ((uint32_t)ctail > rcap) atomicOr(&A.flags[FLAG_REGION_OVERFLOW], 1u);
    }
}

template <typename T, int VPL, int RB, bool MOM2>
__global__ void __launch_bounds__(WAVES * 64, TA_MINWAVES) rle_sweep_kernel(RleArgs P) {
    const SweepArgs& A = P.a;
    constexpr int NW = MOM2 ? 6 : 2;
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    static_assert(TB <= 16 && TC <= 512, "packed LDS moment words assume <= 16 rows x 512 columns per tile");
    using LDS = RleLds<NW, RB>;
    __shared__ LDS S;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < LSLOTS; i += WAVES * 64) {
        S.lkeys[i] = INVALID_LABEL;
#pragma unroll
        for (int k = 0; k < NW; ++k) S.lsum[i * NW + k] = 0ull;
        S.lbox[i * 8 + 0] = 0xFFFFFFFFu; S.lbox[i * 8 + 1] = 0xFFFFFFFFu; S.lbox[i * 8 + 2] = 0xFFFFFFFFu;
        S.lbox[i * 8 + 3] = 0u; S.lbox[i * 8 + 4] = 0u; S.lbox[i * 8 + 5] = 0u;
    }
    hot_row_init(A, tid);
    __syncthreads();

    const int64_t tiles_c = (A.n2 + TC - 1) / TC, tiles_b = (A.n1 + TB - 1) / TB;
    int64_t t = blockIdx.x;
    const int64_t tc = t % tiles_c; t /= tiles_c;
    const int64_t tb = t % tiles_b;
    const int64_t ta_ = t / tiles_b;
    const int64_t c_tile0 = tc * TC, b_tile0 = tb * TB;
    const int64_t p_lo = A.first_owned + ta_ * A.tile_planes;
    int64_t p_hi = p_lo + A.tile_planes;
    if (p_hi > A.n0) p_hi = A.n0;
    const uint64_t wt = (uint64_t)blockIdx.x * WAVES + (uint64_t)w;
    if (p_lo < p_hi) {
        const bool interior = A.vec_ok && (c_tile0 + TC <= A.n2) && (b_tile0 + (int64_t)(w + 1) * RB <= A.n1);
        wave_rle<T, VPL, RB, MOM2>(A, S, !interior, lane, w, c_tile0, b_tile0, p_lo, p_hi,
                                   P.rle + wt * 3ull * P.rcap, P.rcap,
                                   P.dir + wt * (uint64_t)(A.tile_planes * RB), P.hdr + wt);
    }
    __syncthreads();
    flush_tables<NW, false, MOM2, true>(A, S, tid, (uint64_t)(A.a_origin + (p_lo - A.first_owned)), (uint64_t)b_tile0,
                                        (uint64_t)c_tile0, hot_label_of<T>(A));
}

// ---- adjacency from the runs ---------------------------------------------------------------------
struct __attribute__((aligned(16))) PairLds {
    uint64_t pkeys[PSLOTS];
    uint32_t pcnt[PSLOTS * 3];
};

// faces of the run [c0, c0+n) of label L with the row described by `d` (records in `reg`, capacity rcap).
// The other row's records are read 8 at a time with independent loads (they are contiguous in its region),
// so a join costs one or two memory round trips instead of one per record.
template <typename LDS>
__device__ __forceinline__ void join_row(const SweepArgs& A, LDS& S, const uint4 d, const uint32_t* __restrict__ reg,
                                         const uint32_t rcap, const uint32_t L, const uint32_t c0, const uint32_t n,
                                         const uint32_t axis, const uint32_t rank) {
    if (d.z != NOLABEL) {                                    // the other row is one label from end to end
        if (d.z != L && d.z < LABEL_LIMIT) lds_pair_add(A, S, L, d.z, axis, n);
        return;
    }
    const uint32_t c1 = c0 + n;
    const uint2* rec = reinterpret_cast<const uint2*>(reg);
    // Rows are sorted by column and neighbouring rows look alike: the records around the same rank usually
    // cover [c0, c1) -- four reads instead of the whole row
    uint32_t kbeg = 0u, kend = d.y;
    {
        const uint32_t g0 = rank > 0u ? rank - 1u : 0u;
        uint2 wrec[4];
        uint32_t w0 = 0xffffffffu, w1 = 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t k = g0 + (uint32_t)u, i = d.x + k;
            const bool ok = k < d.y && i < rcap;
            wrec[u] = ok ? rec[i] : make_uint2(NOLABEL, 0u);
            const uint32_t o0 = wrec[u].y & 511u, on = (wrec[u].y >> 9) & 1023u;
            if (ok && on != 0u) { w0 = o0 < w0 ? o0 : w0; w1 = o0 + on > w1 ? o0 + on : w1; }
        }
        const bool left_ok = w0 <= c0 || g0 == 0u, right_ok = w1 >= c1 || g0 + 4u >= d.y;
        if (left_ok && right_ok) {                           // the window is all that can overlap: done after these four
            uint32_t len[4], pend = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t o0 = wrec[u].y & 511u, on = (wrec[u].y >> 9) & 1023u, o1 = o0 + on;
                const uint32_t lo = o0 > c0 ? o0 : c0, hi = o1 < c1 ? o1 : c1;
                const bool face = on != 0u && hi > lo && wrec[u].x != L && wrec[u].x < LABEL_LIMIT;
                len[u] = face ? hi - lo : 0u;
                pend |= face ? (1u << u) : 0u;
            }
            while (__builtin_amdgcn_ballot_w64(pend != 0u)) {
                if (pend) {
                    const uint32_t u = (uint32_t)__builtin_ctz(pend);
                    pend &= pend - 1u;
                    uint32_t sl = wrec[0].x, sn = len[0];
#pragma unroll
                    for (int q = 1; q < 4; ++q) { sl = u == (uint32_t)q ? wrec[q].x : sl; sn = u == (uint32_t)q ? len[q] : sn; }
                    lds_pair_add(A, S, L, sl, axis, sn);
                }
            }
            kend = 0u;                                       // nothing left for the full scan
        }
    }
    for (uint32_t k0 = kbeg; k0 < kend; k0 += 8u) {
        uint32_t lab[8], code[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = d.x + k0 + (uint32_t)u;
            const bool ok = k0 + (uint32_t)u < d.y && i < rcap;
            const uint2 rr = ok ? rec[i] : make_uint2(NOLABEL, 0u);
            lab[u] = rr.x;
            code[u] = rr.y;
        }
        // overlaps of all eight first (plain VALU), then one pair add per round for the lanes that still have
        // one pending: rounds = the largest number of overlaps any lane has (2-3), not eight divergent probes
        uint32_t len[8], pend = 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t o0 = code[u] & 511u, on = (code[u] >> 9) & 1023u, o1 = o0 + on;
            const uint32_t lo = o0 > c0 ? o0 : c0, hi = o1 < c1 ? o1 : c1;
            const bool face = on != 0u && hi > lo && lab[u] != L && lab[u] < LABEL_LIMIT;
            len[u] = face ? hi - lo : 0u;
            pend |= face ? (1u << u) : 0u;
        }
        while (__builtin_amdgcn_ballot_w64(pend != 0u)) {
            if (pend) {
                const uint32_t u = (uint32_t)__builtin_ctz(pend);
                pend &= pend - 1u;
                uint32_t sl = lab[0], sn = len[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) { sl = u == (uint32_t)q ? lab[q] : sl; sn = u == (uint32_t)q ? len[q] : sn; }
                lds_pair_add(A, S, L, sl, axis, sn);
            }
        }
    }
}

template <int VPL, int RB>
__global__ void __launch_bounds__(WAVES * 64) rle_adjacency_kernel(RleArgs P) {
    const SweepArgs& A = P.a;
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    __shared__ PairLds S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < PSLOTS; i += WAVES * 64) {
        S.pkeys[i] = EMPTY_KEY;
        S.pcnt[i * 3 + 0] = 0u; S.pcnt[i * 3 + 1] = 0u; S.pcnt[i * 3 + 2] = 0u;
    }
    __syncthreads();

    const int64_t tiles_c = (A.n2 + TC - 1) / TC, tiles_b = (A.n1 + TB - 1) / TB;
    int64_t t = blockIdx.x;
    const int64_t tc = t % tiles_c; t /= tiles_c;
    const int64_t tb = t % tiles_b;
    const int64_t ta_ = t / tiles_b;
    const int64_t owned = A.n0 - A.first_owned;
    int64_t planes = owned - ta_ * A.tile_planes;
    if (planes > A.tile_planes) planes = A.tile_planes;
    const uint32_t rcap = P.rcap, dstride = (uint32_t)(A.tile_planes * RB);
    const uint64_t wt = (uint64_t)blockIdx.x * WAVES + (uint64_t)w;
    const uint32_t* reg = P.rle + wt * 3ull * rcap;
    const uint4* dir = P.dir + wt * dstride;

    // where the row above (a, b-1) and the row of the previous plane (a-1, b) live: wave tile + row index
    auto up_of = [&](uint32_t aloc, uint32_t bloc, uint64_t& owt, uint32_t& orow) -> bool {
        if (bloc > 0u) { owt = (uint64_t)blockIdx.x * WAVES + (bloc - 1u) / RB; orow = aloc * RB + (bloc - 1u) % RB; return true; }
        if (tb == 0) return false;
        owt = ((uint64_t)blockIdx.x - (uint64_t)tiles_c) * WAVES + (WAVES - 1); orow = aloc * RB + (RB - 1);
        return true;
    };
    auto prev_of = [&](uint32_t aloc, uint32_t bloc, uint64_t& owt, uint32_t& orow) -> bool {
        if (aloc > 0u) { owt = (uint64_t)blockIdx.x * WAVES + bloc / RB; orow = (aloc - 1u) * RB + bloc % RB; return true; }
        if (ta_ == 0) return false;
        owt = ((uint64_t)blockIdx.x - (uint64_t)(tiles_b * tiles_c)) * WAVES + bloc / RB;
        orow = (uint32_t)(A.tile_planes - 1) * RB + bloc % RB;
        return true;
    };
    auto both_joins = [&](uint32_t L, uint32_t c0, uint32_t n, uint32_t aloc, uint32_t bloc, uint32_t rank) {
        uint64_t owt; uint32_t orow;
        if (up_of(aloc, bloc, owt, orow))
            join_row(A, S, P.dir[owt * dstride + orow], P.rle + owt * 3ull * rcap, rcap, L, c0, n, 1u, rank);
        if (prev_of(aloc, bloc, owt, orow))
            join_row(A, S, P.dir[owt * dstride + orow], P.rle + owt * 3ull * rcap, rcap, L, c0, n, 0u, rank);
    };

    if (planes > 0) {
        // 1. every record: its axis-2 face, and its run against the row above and the previous plane's row
        const uint32_t nrec = P.hdr[wt] < rcap ? P.hdr[wt] : rcap;
        for (uint32_t i = 0; i < nrec; i += 64u) {
            const uint32_t idx = i + (uint32_t)lane;
            if (idx < nrec) {
                const uint2 rc = reinterpret_cast<const uint2*>(reg)[idx];
                const uint32_t L = rc.x, code = rc.y, v = reg[2u * rcap + idx];
                if (L < LABEL_LIMIT && v < LABEL_LIMIT) lds_pair_add(A, S, L, v, 2u, 1u);
                const uint32_t n = (code >> 9) & 1023u;
                if (n != 0u && L < LABEL_LIMIT) {
                    const uint32_t aloc = (code >> 23) & 63u, bloc = (code >> 19) & 15u;
                    const uint32_t rowstart = dir[aloc * RB + (bloc - (uint32_t)w * RB)].x;      // rank of this run in its row
                    both_joins(L, code & 511u, n, aloc, bloc, idx - rowstart);
                }
            }
        }
        // 2. the rows that are one label from end to end: one run of the whole segment each
        const uint32_t nrows = (uint32_t)planes * RB;
        for (uint32_t i = (uint32_t)lane; i < nrows; i += 64u) {
            const uint4 d = dir[i];
            if (d.z < LABEL_LIMIT) both_joins(d.z, 0u, (uint32_t)TC, i / RB, (uint32_t)w * RB + i % RB, 0u);
        }
        // 3. the axis-2 face across the tile's left edge: last run of the row in the tile to the left (its END
        //    record is the last one the row wrote) against this row's run that starts at column 0
        if (tc > 0) {
            const uint64_t lwt = wt - WAVES;
            const uint32_t* lreg = P.rle + lwt * 3ull * rcap;
            const uint4* ldir = P.dir + lwt * dstride;
            for (uint32_t i = (uint32_t)lane; i < nrows; i += 64u) {
                const uint4 d = dir[i], dl = ldir[i];
                uint32_t first = d.z, last = dl.z;
                if (first == NOLABEL && d.y != 0u && d.x < rcap)       // rows are sorted: the first record is the run from column 0
                    first = reinterpret_cast<const uint2*>(reg)[d.x].x;
                if (last == NOLABEL && dl.y != 0u && dl.x + dl.y - 1u < rcap)
                    last = reinterpret_cast<const uint2*>(lreg)[dl.x + dl.y - 1u].x;
                if (first < LABEL_LIMIT && last < LABEL_LIMIT && first != last) lds_pair_add(A, S, first, last, 2u, 1u);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < PSLOTS; i += WAVES * 64) {
        const uint64_t key = S.pkeys[i];
        if (key == EMPTY_KEY) continue;
        pair_add_global(A.pairs, (uint32_t)(key >> 32), (uint32_t)key, S.pcnt[i * 3 + 0], S.pcnt[i * 3 + 1],
                        S.pcnt[i * 3 + 2], A.flags);
    }
}

template <int VPL, int RB>
static void rle_shape_t(const SweepArgs& a, uint64_t* wave_tiles, uint32_t* rcap, uint32_t* dir_rows) {
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    const int64_t owned = a.n0 - a.first_owned;
    const int64_t tiles = owned <= 0 ? 0 : ((a.n2 + TC - 1) / TC) * ((a.n1 + TB - 1) / TB) *
                                               ((owned + a.tile_planes - 1) / a.tile_planes);
    const uint64_t vox = (uint64_t)RB * TC * (uint64_t)a.tile_planes;
    *wave_tiles = (uint64_t)tiles * WAVES;
    *rcap = (uint32_t)(vox / 8 + (uint64_t)RB * a.tile_planes);       // tissue: ~0.04 records per voxel
    *dir_rows = (uint32_t)(a.tile_planes * RB);
}

void rle_region_shape(const SweepArgs& a, int itemsize, uint64_t* wave_tiles, uint32_t* rcap, uint32_t* dir_rows) {
    if (itemsize == 2) rle_shape_t<8, 2>(a, wave_tiles, rcap, dir_rows);
    else               rle_shape_t<4, TA_RB32>(a, wave_tiles, rcap, dir_rows);
}

template <typename T, int VPL, int RB>
static void launch_rle_t(hipStream_t s, const RleArgs& p, uint32_t fm) {
    uint64_t wave_tiles; uint32_t rcap, drows;
    rle_shape_t<VPL, RB>(p.a, &wave_tiles, &rcap, &drows);
    if (wave_tiles == 0 || p.a.n1 <= 0 || p.a.n2 <= 0) return;
    const dim3 grid((unsigned)(wave_tiles / WAVES)), block(WAVES * 64);
    if (fm & 8u) hipLaunchKernelGGL((rle_sweep_kernel<T, VPL, RB, true>), grid, block, 0, s, p);
    else         hipLaunchKernelGGL((rle_sweep_kernel<T, VPL, RB, false>), grid, block, 0, s, p);
    if (fm & 16u) hipLaunchKernelGGL((rle_adjacency_kernel<VPL, RB>), grid, block, 0, s, p);
}

void launch_rle(hipStream_t s, const RleArgs& p, int itemsize, uint32_t feature_mask) {
    if (itemsize == 2) launch_rle_t<uint16_t, 8, 2>(s, p, feature_mask);
    else               launch_rle_t<uint32_t, 4, TA_RB32>(s, p, feature_mask);
}

}  // namespace ta
