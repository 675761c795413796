// kernels_rowrun.hip -- the sweep with runs along the CONTIGUOUS axis (TA_OPT_IMPL = 3).
//
// Same decomposition, register-resident neighbours, LDS rings and workgroup tables as the fused sweep
// of kernels_sweep.hip, but a label's voxels are summed as runs along memory axis 2 INSIDE one row of
// one plane instead of runs along axis 0 through the tile:
//   * a run ends exactly where an axis-2 face event fires, so ONE record {right voxel, closing label,
//     c0 | n | b | a} serves both (the pair's axis-2 face and the run's ten sums, closed form in c);
//   * nothing about a run outlives its row: no per-column open-run state, no run closes at the end of
//     a tile (they are 43 % of the fused sweep's run records on tissue) -- run records drop from
//     6.7 % to 3.9 % of the voxels;
//   * the start of a run is the previous boundary of the row: a 6-step DPP max-scan over the lanes;
//   * a row that is one label from end to end (background, cell interiors wider than the wave tile)
//     is not a record at all: it is added to six wave-uniform SGPR sums and flushed once per tile.
#include "ta_sweep_common.h"

namespace ta {

#ifndef TA_RQCAP
#define TA_RQCAP 128
#endif
constexpr int RQCAP = TA_RQCAP;     // combined ring (axis-2 face + run): a tissue row adds ~10, one block <= 64

struct __attribute__((aligned(16))) RowWaveLds {
    uint32_t fqv[QCAP], fqp[QCAP];                    // axis-0/1 faces: voxel, neighbour | axis << 30
    uint32_t cqv[RQCAP], cqp[RQCAP], cqc[RQCAP];      // combined: right voxel, closing label, c0 | n << 9 | b << 19 | a << 23
};

template <int NW>
struct __attribute__((aligned(16))) RowLds {
    RowWaveLds wave[WAVES];
    uint64_t lsum[LSLOTS * NW];
    uint64_t pkeys[PSLOTS];
    uint32_t lbox[LSLOTS * 8];
    uint32_t lkeys[LSLOTS];
    uint32_t pcnt[PSLOTS * 3];
};

template <bool ADJ, bool MOM2, typename LDS>
__device__ __forceinline__ void consume_row_rings(const SweepArgs& A, LDS& S, const TileFrame& F, int w, int lane,
                                                  int& fhead, int ftail, int& chead, int ctail, bool all) {
    auto& W = S.wave[w];
    if (ADJ) {
        for (;;) {
            const int cnt = ftail - fhead;
            if (cnt < 64 && !(all && cnt > 0)) break;
            const int qi = (fhead + lane) & (QCAP - 1);
            const uint32_t v = W.fqv[qi], recy = W.fqp[qi];
            const bool act = lane < cnt;
            fhead += cnt < 64 ? cnt : 64;
            if (act) {
                const uint32_t pv = recy & 0x3fffffffu, axis = recy >> 30;
                if (v != INVALID_LABEL && pv < LABEL_LIMIT) lds_pair_add(A, S, pv, v, axis, 1u);
            }
        }
    }
    for (;;) {
        const int cnt = ctail - chead;
        if (cnt < 64 && !(all && cnt > 0)) break;
        const int qi = (chead + lane) & (RQCAP - 1);
        const uint32_t v = W.cqv[qi], label = W.cqp[qi], code = W.cqc[qi];
        const bool act = lane < cnt;
        chead += cnt < 64 ? cnt : 64;
        if (act) {
            if (ADJ && v < LABEL_LIMIT && label < LABEL_LIMIT) lds_pair_add(A, S, label, v, 2u, 1u);
            consume_row_run<MOM2, LDS>(A, S, F, label, code);
        }
    }
}

template <typename T, int VPL, int RB, bool ADJ, bool MOM2, typename LDS>
__device__ __forceinline__ void wave_rowrun(const SweepArgs& A, LDS& S, const bool EDGE, const int lane, const int w,
                                            const int64_t c_tile0, const int64_t b_tile0,
                                            const int64_t p_lo, const int64_t p_hi) {
    constexpr int TC = 64 * VPL;
    static_assert(TC <= 512, "the run code holds c0 in 9 bits and n in 10");
    static_assert(QCAP >= 128 && RQCAP >= 128, "rings must hold a leftover (<64) plus one block (<=64)");
    auto& W = S.wave[w];

    const T* vol = reinterpret_cast<const T*>(A.vol);
    const int64_t n1 = A.n1, n2 = A.n2, plane = n1 * n2;
    const int64_t b_wave0 = b_tile0 + (int64_t)w * RB;
    const int64_t c0g = c_tile0 + (int64_t)lane * VPL;
    const bool has_up = ADJ && b_wave0 > 0;
    const bool has_left = ADJ && c_tile0 > 0;
    const bool has_prev = ADJ && p_lo > 0;
    TileFrame F;
    F.A0 = (uint64_t)(A.a_origin + (p_lo - A.first_owned)); F.B0 = (uint64_t)b_tile0; F.C0 = (uint64_t)c_tile0;
    const uint32_t lane_c = (uint32_t)lane * VPL;
    const uint32_t lane_off = lane_c * (uint32_t)sizeof(T);

    uint32_t cur[RB][VPL], nxt[RB][VPL], prv[RB][VPL];
    uint32_t up[VPL], nxt_up[VPL], left[RB], nxt_left[RB];

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
    auto load_halo = [&](int64_t p, uint32_t (&dup)[VPL], uint32_t (&dl)[RB]) {
        const T* pbase = vol + p * plane;
        if (has_left) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int64_t b = b_wave0 + r;
                dl[r] = b < n1 ? load_uniform_voxel<T>(pbase + b * n2 + c_tile0 - 1) : INVALID_LABEL;
            }
        }
        if (has_up) {
            const bool row_ok = (b_wave0 - 1) < n1;
            const T* row = pbase + (EDGE ? (row_ok ? (b_wave0 - 1) : 0) : (b_wave0 - 1)) * n2 + c_tile0;
            load_strip<T, VPL>(EDGE, row, row_ok, lane_off, c0g, n2, dup);
        }
    };

#pragma unroll
    for (int j = 0; j < VPL; ++j) { up[j] = INVALID_LABEL; nxt_up[j] = INVALID_LABEL; }
#pragma unroll
    for (int r = 0; r < RB; ++r) { left[r] = INVALID_LABEL; nxt_left[r] = INVALID_LABEL; }
    load_rows(p_lo, cur);
    load_halo(p_lo, up, left);
    if (has_prev) {
        load_rows(p_lo - 1, prv);          // the plane before the tile (another tile's, or the slab's halo plane)
    } else {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < VPL; ++j) prv[r][j] = cur[r][j];
    }
    if (p_lo + 1 < p_hi) { load_rows(p_lo + 1, nxt); load_halo(p_lo + 1, nxt_up, nxt_left); }

    int fhead = 0, ftail = 0, chead = 0, ctail = 0;       // free-running ring cursors (wave-uniform)
    // rows that are one label from end to end: wave-uniform sums over (a, b), flushed once per tile
    uint32_t ulab = INVALID_LABEL, un = 0, ua = 0, ub = 0, uaa = 0, uab = 0, ubb = 0;
    uint32_t uamin = 0xffffffffu, uamax = 0, ubmin = 0xffffffffu, ubmax = 0;

    for (int64_t p = p_lo; p < p_hi; ++p) {
        const uint32_t ploc = (uint32_t)(p - p_lo);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const uint32_t bloc = (uint32_t)(w * RB + r);
            // ---- 1. compares: masks land in SGPRs, no branch yet
            uint64_t mb[VPL], mc[VPL], ma[VPL];
            uint32_t pcv[VPL];
            int nface = 0, nrun = 0;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const uint32_t v = cur[r][j];
                pcv[j] = j > 0 ? cur[r][j > 0 ? j - 1 : 0]
                               : lane_shr1(cur[r][VPL - 1], has_left ? left[r] : cur[r][0]);
                mc[j] = __builtin_amdgcn_ballot_w64(v != pcv[j]);
                nrun += __popcll(mc[j]);
                if (ADJ) {
                    if (r > 0) mb[j] = __builtin_amdgcn_ballot_w64(v != cur[r > 0 ? r - 1 : 0][j]);
                    else mb[j] = has_up ? __builtin_amdgcn_ballot_w64(v != up[j]) : 0ull;
                    ma[j] = __builtin_amdgcn_ballot_w64(v != prv[r][j]);
                    nface += __popcll(mb[j]) + __popcll(ma[j]);
                } else {
                    mb[j] = 0ull; ma[j] = 0ull;
                }
            }
            // the row is one label when no boundary fires inside it (bit 0 of mc[0] is the halo compare)
            const uint32_t rowlab = __builtin_amdgcn_readfirstlane(cur[r][0]);
            const bool uniform = ((mc[0] & ~1ull) | (VPL > 1 ? mc[1 % VPL] : 0ull) | (VPL > 2 ? mc[2 % VPL] : 0ull) |
                                  (VPL > 3 ? mc[3 % VPL] : 0ull) | (VPL > 4 ? mc[4 % VPL] : 0ull) |
                                  (VPL > 5 ? mc[5 % VPL] : 0ull) | (VPL > 6 ? mc[6 % VPL] : 0ull) |
                                  (VPL > 7 ? mc[7 % VPL] : 0ull)) == 0ull;
            const bool outside = uniform && rowlab == INVALID_LABEL;           // a row beyond the volume
            const bool summed = uniform && !outside && (ulab == INVALID_LABEL || ulab == rowlab);
            if (summed) {
                ulab = rowlab; un += 1u; ua += ploc; ub += bloc; uaa += ploc * ploc; uab += ploc * bloc; ubb += bloc * bloc;
                uamin = ploc < uamin ? ploc : uamin; uamax = ploc > uamax ? ploc : uamax;
                ubmin = bloc < ubmin ? bloc : ubmin; ubmax = bloc > ubmax ? bloc : ubmax;
            }
            const bool need_end = !summed && !outside;                          // the row's last run closes by a record
            if (nface + nrun == 0 && !need_end) continue;                       // the common case: one branch per row

            // ---- 2. start of every run = previous boundary of the row (0 when there is none)
            uint32_t s;
            {
                uint32_t lastpos = 0u;
#pragma unroll
                for (int j = 0; j < VPL; ++j) lastpos = (cur[r][j] != pcv[j]) ? lane_c + (uint32_t)j : lastpos;
                s = lane_shr1(wave_scan_max(lastpos), 0u);
            }
            const bool overflow = (ftail - fhead) + nface > QCAP || (ctail - chead) + nrun + 1 > RQCAP;
#define TA_EMIT_ROW(GUARD)                                                                                  \
            {                                                                                                \
                int fo = ftail, co = ctail;                                                                  \
                uint32_t s_ = s;                                                                             \
                _Pragma("unroll") for (int j = 0; j < VPL; ++j) {                                            \
                    const uint32_t v = cur[r][j];                                                            \
                    if (ADJ) {                                                                               \
                        if ((r > 0 || has_up) && GUARD(3 * j)) {                                             \
                            const uint32_t pv = r > 0 ? cur[r > 0 ? r - 1 : 0][j] : up[j];                   \
                            if (v != pv) {                                                                   \
                                const int q_ = (fo + (int)mbcnt64(mb[j])) & (QCAP - 1);                      \
                                W.fqv[q_] = v; W.fqp[q_] = pv | (1u << 30);                                  \
                            }                                                                                \
                            fo += __popcll(mb[j]);                                                           \
                        }                                                                                    \
                        if (GUARD(3 * j + 1)) {                                                              \
                            const uint32_t o = prv[r][j];                                                    \
                            if (v != o) {                                                                    \
                                const int q_ = (fo + (int)mbcnt64(ma[j])) & (QCAP - 1);                      \
                                W.fqv[q_] = v; W.fqp[q_] = o;                                                \
                            }                                                                                \
                            fo += __popcll(ma[j]);                                                           \
                        }                                                                                    \
                    }                                                                                        \
                    {   /* axis 2: the boundary closes the run [s_, k) of the left label */                  \
                        const uint32_t k = lane_c + (uint32_t)j;                                             \
                        if (GUARD(3 * j + 2)) {                                                              \
                            if (v != pcv[j]) {                                                               \
                                const int q_ = (co + (int)mbcnt64(mc[j])) & (RQCAP - 1);                     \
                                W.cqv[q_] = v; W.cqp[q_] = pcv[j];                                           \
                                W.cqc[q_] = s_ | ((k - s_) << 9) | (bloc << 19) | (ploc << 23);              \
                            }                                                                                \
                            co += __popcll(mc[j]);                                                           \
                        }                                                                                    \
                        s_ = (v != pcv[j]) ? k : s_;                                                         \
                    }                                                                                        \
                }                                                                                            \
                if (need_end && GUARD(3 * VPL)) {                                                            \
                    if (lane == 63) {                                                                        \
                        const int q_ = co & (RQCAP - 1);                                                     \
                        W.cqv[q_] = INVALID_LABEL; W.cqp[q_] = cur[r][VPL - 1];                              \
                        W.cqc[q_] = s_ | (((uint32_t)TC - s_) << 9) | (bloc << 19) | (ploc << 23);           \
                    }                                                                                        \
                    co += 1;                                                                                 \
                }                                                                                            \
                ftail = fo; ctail = co;                                                                      \
            }
#define TA_GUARD_ALL(k) true
#define TA_GUARD_PASS(k) (pass == (k))
            if (!overflow) {
                TA_EMIT_ROW(TA_GUARD_ALL)
                if (ftail - fhead >= 64 || ctail - chead >= 64) {
                    __builtin_amdgcn_wave_barrier();
                    consume_row_rings<ADJ, MOM2, LDS>(A, S, F, w, lane, fhead, ftail, chead, ctail, false);
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                // never seen on tissue, only on noise: one block per pass so the rings cannot overflow
#pragma nounroll
                for (int pass = 0; pass <= 3 * VPL; ++pass) {
                    TA_EMIT_ROW(TA_GUARD_PASS)
                    if (ftail - fhead >= 64 || ctail - chead >= 64) {
                        __builtin_amdgcn_wave_barrier();
                        consume_row_rings<ADJ, MOM2, LDS>(A, S, F, w, lane, fhead, ftail, chead, ctail, false);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
#undef TA_EMIT_ROW
#undef TA_GUARD_ALL
#undef TA_GUARD_PASS
        }
        // ---- advance: rotate the register planes, keep one plane of loads in flight
        if (p + 1 < p_hi) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    if (ADJ) prv[r][j] = cur[r][j];
                    cur[r][j] = nxt[r][j];
                }
                left[r] = nxt_left[r];
            }
#pragma unroll
            for (int j = 0; j < VPL; ++j) up[j] = nxt_up[j];
            if (p + 2 < p_hi) { load_rows(p + 2, nxt); load_halo(p + 2, nxt_up, nxt_left); }
        }
    }

    // ---- end of tile: drain the rings, then the uniform rows in one closed form
    __builtin_amdgcn_wave_barrier();
    consume_row_rings<ADJ, MOM2, LDS>(A, S, F, w, lane, fhead, ftail, chead, ctail, true);
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
}

template <typename T, int VPL, int RB, bool ADJ, bool MOM2>
__global__ void __launch_bounds__(WAVES * 64, TA_MINWAVES) rowrun_kernel(SweepArgs A) {
    constexpr int NW = MOM2 ? 6 : 2;
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    static_assert(TB <= 16 && TC <= 512, "packed LDS moment words assume <= 16 rows x 512 columns per tile");
    using LDS = RowLds<NW>;
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
    if (ADJ) {
        for (int i = tid; i < PSLOTS; i += WAVES * 64) {
            S.pkeys[i] = EMPTY_KEY;
            S.pcnt[i * 3 + 0] = 0u; S.pcnt[i * 3 + 1] = 0u; S.pcnt[i * 3 + 2] = 0u;
        }
    }
    if (!ADJ) hot_row_init(A, tid);          // the hot-label rows are used by the variants without adjacency only
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

    if (p_lo < p_hi) {
        const bool interior = A.vec_ok && (c_tile0 + TC <= A.n2) && (b_tile0 + (int64_t)(w + 1) * RB <= A.n1);
        wave_rowrun<T, VPL, RB, ADJ, MOM2>(A, S, !interior, lane, w, c_tile0, b_tile0, p_lo, p_hi);
    }
    __syncthreads();
    flush_tables<NW, ADJ, MOM2, !ADJ>(A, S, tid, (uint64_t)(A.a_origin + (p_lo - A.first_owned)), (uint64_t)b_tile0,
                                      (uint64_t)c_tile0, ADJ ? 0u : hot_label_of<T>(A));
}

template <typename T, int VPL, int RB>
static void launch_rowrun_t(hipStream_t s, const SweepArgs& a, uint32_t fm) {
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    const int64_t owned = a.n0 - a.first_owned;
    if (owned <= 0 || a.n1 <= 0 || a.n2 <= 0) return;
    const int64_t tiles = ((a.n2 + TC - 1) / TC) * ((a.n1 + TB - 1) / TB) *
                          ((owned + a.tile_planes - 1) / a.tile_planes);
    const dim3 grid((unsigned)tiles), block(WAVES * 64);
    const bool adj = fm & 16u, mom2 = fm & 8u;
    if (adj && mom2)       hipLaunchKernelGGL((rowrun_kernel<T, VPL, RB, true, true>), grid, block, 0, s, a);
    else if (adj && !mom2) hipLaunchKernelGGL((rowrun_kernel<T, VPL, RB, true, false>), grid, block, 0, s, a);
    else if (!adj && mom2) hipLaunchKernelGGL((rowrun_kernel<T, VPL, RB, false, true>), grid, block, 0, s, a);
    else                   hipLaunchKernelGGL((rowrun_kernel<T, VPL, RB, false, false>), grid, block, 0, s, a);
}

void launch_rowrun(hipStream_t s, const SweepArgs& a, int itemsize, uint32_t feature_mask) {
    if (itemsize == 2) launch_rowrun_t<uint16_t, 8, 2>(s, a, feature_mask);
    else               launch_rowrun_t<uint32_t, 4, TA_RB32>(s, a, feature_mask);
}

}  // namespace ta
