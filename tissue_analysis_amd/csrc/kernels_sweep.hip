// kernels_sweep.hip -- the fused single-sweep feature extractor for gfx950 (MI355X).
//
// One coalesced pass over the labelled volume produces, per label, the exact integer
// accumulators behind SpatialImageAnalysis.volume / boundingbox / center_of_mass / inertia_axis
// (SIA:1197-1292, 417-535) and, per unordered label pair, the per-axis shared-face counts
// behind neighbors / cell_wall_area / wall_areas (SIA:538-660, 908-993).
//
// Work decomposition (memory axes: 0 slowest ... 2 fastest)
//   workgroup = WAVES waves stacked along axis 1; wave tile = RB rows x (64 lanes * VPL voxels)
//   along axis 2; each lane keeps its RB x VPL voxels of the current plane in VGPRs (16-byte
//   loads, next plane prefetched) and the workgroup walks `tile_planes` planes along axis 0.
//   * Neighbours never leave the register file: axis 0 = the lane's registers from the previous
//     plane (they double as the label of the column's open run), axis 1 = the row above in the
//     same lane (+ one halo row per wave), axis 2 = the previous voxel of the strip (+ one DPP
//     wave shift for voxel 0).
//   * Every (voxel slot, axis) is one straight-line BLOCK: v_cmp gives the 64-lane event mask for
//     free; when it is non-zero the firing lanes append an 8-byte record {voxel, neighbour|axis}
//     (faces) and, for axis 0, {closed label, column, first plane, length} (runs) to small per-wave
//     LDS rings at mask-prefix (mbcnt) offsets.  Empty blocks cost one compare and one branch --
//     that is the whole cost of the ~60 % of a tissue-in-ellipsoid volume that is background or
//     cell interior.
//   * The rings are consumed 64 records at a time with every lane busy.  A closed column run
//     (label, a0, n, b, c) yields all ten moments in closed form in TILE-LOCAL coordinates, two
//     32-bit sums per u64 LDS atomic, into a workgroup-shared LDS hash table; faces go to a
//     second LDS hash keyed by (lo,hi) with three counters.
//   * The wave body is a resumable fall-through `switch`: blocks run back to back and there is
//     exactly ONE consumer site, entered when a ring holds >= 64 records.
//   * Tables are flushed once per workgroup tile with global atomics (u64 add / i32 min) after
//     shifting the local sums to global coordinates; integer sums and minima make the result
//     independent of tiling, scheduling and order.
//   No MFMA anywhere: integer compare/reduce work bound by the HBM read of the volume.
#include "ta_sweep_common.h"

namespace ta {

struct __attribute__((aligned(16))) WaveLds {     // struct-of-arrays rings: two dword stores, no 64-bit register pairs
    uint32_t fqv[QCAP], fqp[QCAP];   // faces: voxel, neighbour | axis << 30
    uint32_t rql[QCAP], rqc[QCAP];   // runs : label, c_loc | b_loc << 10 | a0 << 14 | n << 20
};

// Workgroup tables.  Moments are kept in TILE-LOCAL coordinates (a < 64 planes, b < 16 rows,
// c < 512 columns, so N <= 2^19 voxels of one label per tile) and two sums share one u64 LDS
// atomic with a free 32/32 split; every half is a hard maximum below 2^32:
//   w0 = N | Sb<<32   w1 = Sa | Sc<<32   w2 = Saa | Sab<<32   w3 = Sbb | Sbc<<32   w4 = Sac   w5 = Scc
// They are unpacked and shifted to global coordinates (exact u64) once per tile at the flush.
template <int NW>
struct __attribute__((aligned(16))) SweepLds {
    WaveLds wave[WAVES];
    uint64_t lsum[LSLOTS * NW];
    uint64_t pkeys[PSLOTS];
    uint32_t lbox[LSLOTS * 8];    // per slot: min a,b,c | max a,b,c (tile-local) | 2 pad
    uint32_t lkeys[LSLOTS];
    uint32_t pcnt[PSLOTS * 3];
#ifdef TA_LDS_PAD
    uint32_t pad_[TA_LDS_PAD];
#endif
};

// one run record -> the ten tile-local sums (every term fits 32 bits) -> the workgroup label table.
// Every factor is below 2^24 (n <= 64, a < 64, b < 16, c < 1024, partial products < 2^22), so the
// full-rate 24-bit multiplier (v_mul_u32_u24 / v_mad_u32_u24) is exact here; v_mul_lo_u32 is quarter rate.
template <bool MOM2, typename LDS>
__device__ __forceinline__ void consume_run_record(const SweepArgs& A, LDS& S, const TileFrame& F, uint32_t label,
                                                   uint32_t code) {
    const uint32_t cl = code & 1023u, bl = (code >> 10) & 15u, a0l = (code >> 14) & 63u, n = (code >> 20) & 127u;
    if (label == INVALID_LABEL || n == 0u) return;
    const uint32_t a1l = a0l + n - 1u;
    const uint32_t sa1 = __umul24(n, a0l + a1l) >> 1;                       // sum a over the run
    const uint32_t nb = __umul24(n, bl), nc = __umul24(n, cl);
    RunSums L;
    L.n = n; L.sa = sa1; L.sb = nb; L.sc = nc;
    if (MOM2) {
        const uint32_t t1 = __umul24(n, n - 1u);                            // n (n - 1), even
        L.saa = __umul24(__umul24(n, a0l), a0l) + __umul24(a0l, t1) + __umul24(t1 >> 1, 2u * n - 1u) / 3u;
        L.sab = __umul24(bl, sa1); L.sac = __umul24(sa1, cl);
        L.sbb = __umul24(nb, bl); L.sbc = __umul24(nb, cl); L.scc = __umul24(nc, cl);
    } else {
        L.saa = L.sab = L.sac = L.sbb = L.sbc = L.scc = 0;
    }
    lds_label_add<MOM2, LDS, RunSums>(A, S, F, label, L, a0l, a1l, bl, bl, cl, cl);
}

// ---- the consumer ----------------------------------------------------------------------------
// Drains complete groups of 64 records from both rings of a wave (everything when `all`), every
// lane busy.  fhead/rhead are advanced; all cursors stay wave-uniform.
template <bool ADJ, bool MOM2, typename LDS>
__device__ __forceinline__ void consume_rings(const SweepArgs& A, LDS& S, const TileFrame& F, int w, int lane,
                                              int& fhead, int ftail, int& rhead, int rtail, bool all) {
    auto& W = S.wave[w];
    if (ADJ) {
        for (;;) {
            const int cnt = ftail - fhead;
            if (cnt < 64 && !(all && cnt > 0)) break;
            const int qi = (fhead + lane) & (QCAP - 1);
            const uint32_t recx = W.fqv[qi], recy = W.fqp[qi];
            const bool act = lane < cnt;
            fhead += cnt < 64 ? cnt : 64;
            if (TA_ABLATE == 0 && act) {
                const uint32_t v = recx, pv = recy & 0x3fffffffu, axis = recy >> 30;
                if (v != INVALID_LABEL && pv < LABEL_LIMIT) lds_pair_add(A, S, pv, v, axis, 1u);
            }
        }
    }
    for (;;) {
        const int cnt = rtail - rhead;
        if (cnt < 64 && !(all && cnt > 0)) break;
        const int qi = (rhead + lane) & (QCAP - 1);
        const uint32_t label = W.rql[qi], code = W.rqc[qi];
        const bool act = lane < cnt;
        rhead += cnt < 64 ? cnt : 64;
        if (TA_ABLATE == 0 && act) consume_run_record<MOM2, LDS>(A, S, F, label, code);
    }
}

// ---- the wave body ---------------------------------------------------------------------------
template <typename T, int VPL, int RB, bool ADJ, bool MOM2, typename LDS>
__device__ __forceinline__ void wave_sweep(const SweepArgs& A, LDS& S, const bool EDGE, const int lane, const int w,
                                           const int64_t c_tile0, const int64_t b_tile0,
                                           const int64_t p_lo, const int64_t p_hi) {
    constexpr int TC = 64 * VPL;
    static_assert(QCAP >= 128, "rings must hold a leftover (<64) plus one block (<=64)");
    auto& W = S.wave[w];

    const T* vol = reinterpret_cast<const T*>(A.vol);
    const int64_t n1 = A.n1, n2 = A.n2, plane = n1 * n2;
    const int64_t b_wave0 = b_tile0 + (int64_t)w * RB;
    const int64_t c0 = c_tile0 + (int64_t)lane * VPL;
    const bool has_up = ADJ && b_wave0 > 0;
    const bool has_left = ADJ && c_tile0 > 0;
    const bool has_prev = p_lo > 0;
    TileFrame F;
    F.A0 = (uint64_t)(A.a_origin + (p_lo - A.first_owned)); F.B0 = (uint64_t)b_tile0; F.C0 = (uint64_t)c_tile0;
    const uint32_t lane_c = (uint32_t)lane * VPL;
    const uint32_t lane_off = lane_c * (uint32_t)sizeof(T);

    uint32_t cur[RB][VPL], nxt[RB][VPL], nx2[RB][VPL], runlab[RB][VPL];
    uint32_t a0w[RB][VPL / 4];                 // first plane of each column's open run, one byte per column
    uint32_t up[VPL], nxt_up[VPL], nx2_up[VPL], left[RB], nxt_left[RB], nx2_left[RB];

    auto load_rows = [&](int64_t p, uint32_t (&d)[RB][VPL]) {
        const T* pbase = vol + p * plane;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int64_t b = b_wave0 + r;
            const bool row_ok = b < n1;
            const T* row = pbase + (EDGE ? (row_ok ? b : 0) : b) * n2 + c_tile0;
            load_strip<T, VPL>(EDGE, row, row_ok, lane_off, c0, n2, d[r]);
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
            load_strip<T, VPL>(EDGE, row, row_ok, lane_off, c0, n2, dup);
        }
    };

    // ---- prologue ------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < VPL; ++j) { up[j] = INVALID_LABEL; nxt_up[j] = INVALID_LABEL; nx2_up[j] = INVALID_LABEL; }
#pragma unroll
    for (int r = 0; r < RB; ++r) { left[r] = INVALID_LABEL; nxt_left[r] = INVALID_LABEL; nx2_left[r] = INVALID_LABEL; }
    load_rows(p_lo, cur);
    load_halo(p_lo, up, left);
    if (ADJ && has_prev) {
        load_rows(p_lo - 1, runlab);       // the plane before the tile: faces only (its runs have length 0)
    } else {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < VPL; ++j) runlab[r][j] = cur[r][j];
    }
    if (p_lo + 1 < p_hi) { load_rows(p_lo + 1, nxt); load_halo(p_lo + 1, nxt_up, nxt_left); }
    if (TA_PREFETCH >= 2 && p_lo + 2 < p_hi) { load_rows(p_lo + 2, nx2); load_halo(p_lo + 2, nx2_up, nx2_left); }
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int q = 0; q < VPL / 4; ++q) a0w[r][q] = 0u;

    int fhead = 0, ftail = 0, rhead = 0, rtail = 0;       // free-running ring cursors (wave-uniform)
    bool any_event = false;
#ifdef TA_STAMPS
    uint64_t tk_cmp = 0, tk_emit = 0, tk_cons = 0, tk_adv = 0, tk_rows = 0, tk_evrows = 0;
    const uint64_t tk_begin = __builtin_amdgcn_s_memtime();
#define TA_T() __builtin_amdgcn_s_memtime()
#endif
    const uint32_t first_label = __builtin_amdgcn_readfirstlane(cur[0][0]);
    const uint32_t last = (uint32_t)(p_hi - 1 - p_lo);

    for (int64_t p = p_lo; p < p_hi; ++p) {
        const uint32_t ploc = (uint32_t)(p - p_lo);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            // ---- 1. the row's 3*VPL compares, batched: masks land in SGPRs, no branch yet
#ifdef TA_STAMPS
            const uint64_t t0 = TA_T();
#endif
            uint64_t mb[VPL], mc[VPL], ma[VPL];
            uint32_t pcv[VPL];
            int nface = 0, nrun = 0;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const uint32_t v = cur[r][j];
                if (ADJ) {
                    if (r > 0) mb[j] = __builtin_amdgcn_ballot_w64(v != cur[r > 0 ? r - 1 : 0][j]);
                    else mb[j] = has_up ? __builtin_amdgcn_ballot_w64(v != up[j]) : 0ull;
                    pcv[j] = j > 0 ? cur[r][j > 0 ? j - 1 : 0]
                                   : lane_shr1(cur[r][VPL - 1], has_left ? left[r] : cur[r][0]);
                    mc[j] = __builtin_amdgcn_ballot_w64(v != pcv[j]);
                    nface += __popcll(mb[j]) + __popcll(mc[j]);
                } else {
                    mb[j] = 0ull; mc[j] = 0ull; pcv[j] = 0u;
                }
                ma[j] = __builtin_amdgcn_ballot_w64(v != runlab[r][j]);
                nrun += __popcll(ma[j]);
            }
            if (ADJ) nface += nrun;
#ifdef TA_STAMPS
            const uint64_t t1 = TA_T();
            tk_cmp += t1 - t0; tk_rows += 1;
#endif
            if (nface + nrun == 0) continue;              // the common case: one branch per row
            any_event = true;
            // ---- 2. emit + 3. consume.  Firing lanes append value-carrying records at mask-prefix
            // offsets; the block list is straight-line and predicated.  Normally the whole row is one
            // pass.  A row whose records might not fit in the rings (never seen on tissue, only on
            // noise) runs one block per pass, each followed by the same single consumer site.
            const bool overflow = (ftail - fhead) + nface > QCAP || (rtail - rhead) + nrun > QCAP;
// one row's blocks; GUARD(k) selects which blocks run in this pass (always true on the fast path)
#define TA_EMIT_ROW(GUARD)                                                                                  \
            {                                                                                                \
                int fo = ftail, ro = rtail;                                                                  \
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
                            if (v != pcv[j]) {                                                               \
                                const int q_ = (fo + (int)mbcnt64(mc[j])) & (QCAP - 1);                      \
                                W.fqv[q_] = v; W.fqp[q_] = pcv[j] | (2u << 30);                              \
                            }                                                                                \
                            fo += __popcll(mc[j]);                                                           \
                        }                                                                                    \
                    }                                                                                        \
                    if (GUARD(3 * j + 2)) {                                                                  \
                        /* axis 0: closes the column's open run (length ploc - a0; 0 for the plane before the tile) */ \
                        const uint32_t o = runlab[r][j];                                                     \
                        if (v != o) {                                                                        \
                            const uint32_t i = mbcnt64(ma[j]);                                               \
                            const uint32_t sh = 8u * (j % 4);                                                \
                            const uint32_t a0 = (a0w[r][j / 4] >> sh) & 0xffu;                               \
                            if (ADJ) { const int q_ = (fo + (int)i) & (QCAP - 1); W.fqv[q_] = v; W.fqp[q_] = o; } \
                            const int q2_ = (ro + (int)i) & (QCAP - 1);                                      \
                            W.rql[q2_] = o;                                                                  \
                            W.rqc[q2_] = (lane_c + j) | ((uint32_t)(w * RB + r) << 10) | (a0 << 14) | ((ploc - a0) << 20); \
                            a0w[r][j / 4] = (a0w[r][j / 4] & ~(0xffu << sh)) | (ploc << sh);                 \
                            runlab[r][j] = v;                                                                \
                        }                                                                                    \
                        const int n = __popcll(ma[j]);                                                       \
                        if (ADJ) fo += n;                                                                    \
                        ro += n;                                                                             \
                    }                                                                                        \
                }                                                                                            \
                ftail = fo; rtail = ro;                                                                      \
            }
#define TA_GUARD_ALL(k) true
#define TA_GUARD_PASS(k) (pass == (k))
            if (!overflow) {
                TA_EMIT_ROW(TA_GUARD_ALL)
#ifdef TA_STAMPS
                const uint64_t t2 = TA_T();
                tk_emit += t2 - t1; tk_evrows += 1;
#endif
                if (ftail - fhead >= 64 || rtail - rhead >= 64) {
                    __builtin_amdgcn_wave_barrier();
                    consume_rings<ADJ, MOM2, LDS>(A, S, F, w, lane, fhead, ftail, rhead, rtail, false);
                    __builtin_amdgcn_wave_barrier();
                }
#ifdef TA_STAMPS
                tk_cons += TA_T() - t2;
#endif
            } else {
                // never seen on tissue, only on noise: one block per pass so the rings cannot overflow
#pragma nounroll
                for (int pass = 0; pass < 3 * VPL; ++pass) {
                    TA_EMIT_ROW(TA_GUARD_PASS)
                    if (ftail - fhead >= 64 || rtail - rhead >= 64) {
                        __builtin_amdgcn_wave_barrier();
                        consume_rings<ADJ, MOM2, LDS>(A, S, F, w, lane, fhead, ftail, rhead, rtail, false);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
#undef TA_EMIT_ROW
#undef TA_GUARD_ALL
#undef TA_GUARD_PASS
        }
#ifdef TA_STAMPS
        const uint64_t t4 = TA_T();
#endif
        // ---- advance: rotate the register planes, keep TA_PREFETCH planes of loads in flight
        if (p + 1 < p_hi) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    cur[r][j] = nxt[r][j];
                    if (TA_PREFETCH >= 2) nxt[r][j] = nx2[r][j];
                }
                left[r] = nxt_left[r];
                if (TA_PREFETCH >= 2) nxt_left[r] = nx2_left[r];
            }
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                up[j] = nxt_up[j];
                if (TA_PREFETCH >= 2) nxt_up[j] = nx2_up[j];
            }
            if (TA_PREFETCH >= 2) {
                if (p + 3 < p_hi) { load_rows(p + 3, nx2); load_halo(p + 3, nx2_up, nx2_left); }
            } else {
                if (p + 2 < p_hi) { load_rows(p + 2, nxt); load_halo(p + 2, nxt_up, nxt_left); }
            }
        }
#ifdef TA_STAMPS
        {   // make the wait for the rotated registers visible here, not in the next compare
            uint32_t sink = cur[0][0];
            asm volatile("" :: "v"(sink));
            tk_adv += TA_T() - t4;
        }
#endif
    }
#ifdef TA_STAMPS
    if (lane == 0) {
        atomicAdd(&A.flags[8], (uint32_t)(tk_cmp >> 8)); atomicAdd(&A.flags[9], (uint32_t)(tk_emit >> 8));
        atomicAdd(&A.flags[10], (uint32_t)(tk_cons >> 8)); atomicAdd(&A.flags[11], (uint32_t)(tk_adv >> 8));
        atomicAdd(&A.flags[12], (uint32_t)((TA_T() - tk_begin) >> 8));
        atomicAdd(&A.flags[13], (uint32_t)tk_rows); atomicAdd(&A.flags[14], (uint32_t)tk_evrows);
    }
#endif

    // ---- end of tile: close every open run --------------------------------------------------
    if (ADJ && !any_event) {
        // no event at all (needs the in-plane compares of ADJ): the whole wave tile is one label
        // (or lies outside the volume): one closed-form box in tile-local coordinates
        if (lane == 0 && first_label != INVALID_LABEL) {
            const uint64_t na = last + 1u, nb = RB, nc = TC, b0 = (uint64_t)w * RB;
            const uint64_t sa = range_sum1(0, na), sb = range_sum1(b0, nb), sc = range_sum1(0, nc);
            LocalSums L;
            L.n = na * nb * nc; L.sa = sa * nb * nc; L.sb = sb * na * nc; L.sc = sc * na * nb;
            if (MOM2) {
                L.saa = range_sum2(0, na) * nb * nc; L.sab = sa * sb * nc; L.sac = sa * sc * nb;
                L.sbb = range_sum2(b0, nb) * na * nc; L.sbc = sb * sc * na; L.scc = range_sum2(0, nc) * na * nb;
            } else {
                L.saa = L.sab = L.sac = L.sbb = L.sbc = L.scc = 0;
            }
            lds_label_add<MOM2, LDS, LocalSums>(A, S, F, first_label, L, 0u, last, (uint32_t)b0, (uint32_t)(b0 + nb - 1), 0u,
                                                (uint32_t)(nc - 1));
        }
    } else {
        // every column closes its run [a0, last]: RB*VPL dense blocks of 64 records, consumed as they come
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const uint32_t a0 = (a0w[r][j / 4] >> (8u * (j % 4))) & 0xffu;
                const int q_ = (rtail + lane) & (QCAP - 1);
                W.rql[q_] = runlab[r][j];
                W.rqc[q_] = (lane_c + j) | ((uint32_t)(w * RB + r) << 10) | (a0 << 14) | ((last + 1u - a0) << 20);
                rtail += 64;
                if (QCAP < 64 * (VPL + 1) || j == VPL - 1) {
                    __builtin_amdgcn_wave_barrier();
                    consume_rings<ADJ, MOM2, LDS>(A, S, F, w, lane, fhead, ftail, rhead, rtail, r == RB - 1 && j == VPL - 1);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
}

template <typename T, int VPL, int RB, bool ADJ, bool MOM2>
__global__ void __launch_bounds__(WAVES * 64, TA_MINWAVES) sweep_kernel(SweepArgs A) {
    constexpr int NW = MOM2 ? 6 : 2;
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    static_assert(TB <= 16 && TC <= 512, "packed LDS moment words assume <= 16 rows x 512 columns per tile");
    using LDS = SweepLds<NW>;
    __shared__ LDS S;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave index: make it provably wave-uniform
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
    __syncthreads();

    const int64_t tiles_c = (A.n2 + TC - 1) / TC, tiles_b = (A.n1 + TB - 1) / TB;
#ifdef TA_XCD_REMAP
    // Workgroups are dealt round-robin over the 8 XCDs (speed only, never correctness): give each
    // XCD a contiguous range of tile ids so tiles that share halo rows/planes share an L2.
    int64_t t;
    {
        const int64_t nwg = gridDim.x, orig = blockIdx.x, xcd = orig % 8, q = nwg / 8, rr = nwg % 8;
        t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    }
#else
    int64_t t = blockIdx.x;
#endif
    const int64_t tc = t % tiles_c; t /= tiles_c;
    const int64_t tb = t % tiles_b;
    const int64_t ta_ = t / tiles_b;
    const int64_t c_tile0 = tc * TC, b_tile0 = tb * TB;
    const int64_t p_lo = A.first_owned + ta_ * A.tile_planes;
    int64_t p_hi = p_lo + A.tile_planes;
    if (p_hi > A.n0) p_hi = A.n0;

    if (p_lo < p_hi) {
        const bool interior = A.vec_ok && (c_tile0 + TC <= A.n2) && (b_tile0 + (int64_t)(w + 1) * RB <= A.n1);
        wave_sweep<T, VPL, RB, ADJ, MOM2>(A, S, !interior, lane, w, c_tile0, b_tile0, p_lo, p_hi);
    }
    __syncthreads();

    flush_tables<NW, ADJ, MOM2, false>(A, S, tid, (uint64_t)(A.a_origin + (p_lo - A.first_owned)), (uint64_t)b_tile0, (uint64_t)c_tile0, 0u);
}

// =================================================================================================
// Split path (TA_OPT_IMPL = 2): the same events, but the streaming kernel only EMITS them -- straight
// to per-wave-tile record regions in HBM with coalesced mask-prefix stores, no LDS, no hash, no
// atomics -- and a second, dense kernel REDUCES the regions through the same workgroup LDS tables
// and flush.  Bit-exact like the fused sweep and kept as a DIAGNOSTIC: it lets rocprofv3 time the
// producer and the consumer half separately (C4: emit 1.58 ms, reduce 0.89 ms = slower than the
// fused 1.99 ms); a volume with more events than a region holds falls back to the
// fused sweep by itself (FLAG_REGION_OVERFLOW).
// =================================================================================================
#ifndef TA_EMIT_MINWAVES
#define TA_EMIT_MINWAVES 4        // 5 waves/SIMD would need <= 96 VGPRs: 59 spilled registers, 2.9 ms
#endif

template <typename T, int VPL, int RB, bool ADJ>
__device__ __forceinline__ void wave_emit(const SweepArgs& A, const uint32_t fcap, const uint32_t rcap,
                                          const bool EDGE, const int lane, const int w,
                                          const int64_t c_tile0, const int64_t b_tile0, const int64_t p_lo,
                                          const int64_t p_hi, uint2* __restrict__ fr, uint2* __restrict__ rr,
                                          uint32_t* __restrict__ hdr) {
    const T* vol = reinterpret_cast<const T*>(A.vol);
    const int64_t n1 = A.n1, n2 = A.n2, plane = n1 * n2;
    const int64_t b_wave0 = b_tile0 + (int64_t)w * RB;
    const int64_t c0 = c_tile0 + (int64_t)lane * VPL;
    const bool has_up = ADJ && b_wave0 > 0;
    const bool has_left = ADJ && c_tile0 > 0;
    const bool has_prev = p_lo > 0;
    const uint32_t lane_c = (uint32_t)lane * VPL;
    const uint32_t lane_off = lane_c * (uint32_t)sizeof(T);

    uint32_t cur[RB][VPL], nxt[RB][VPL], runlab[RB][VPL];
    uint32_t a0w[RB][VPL / 4];
    uint32_t up[VPL], nxt_up[VPL], left[RB], nxt_left[RB];

    auto load_rows = [&](int64_t p, uint32_t (&d)[RB][VPL]) {
        const T* pbase = vol + p * plane;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int64_t b = b_wave0 + r;
            const bool row_ok = b < n1;
            const T* row = pbase + (EDGE ? (row_ok ? b : 0) : b) * n2 + c_tile0;
            load_strip<T, VPL>(EDGE, row, row_ok, lane_off, c0, n2, d[r]);
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
            load_strip<T, VPL>(EDGE, row, row_ok, lane_off, c0, n2, dup);
        }
    };

#pragma unroll
    for (int j = 0; j < VPL; ++j) { up[j] = INVALID_LABEL; nxt_up[j] = INVALID_LABEL; }
#pragma unroll
    for (int r = 0; r < RB; ++r) { left[r] = INVALID_LABEL; nxt_left[r] = INVALID_LABEL; }
    load_rows(p_lo, cur);
    load_halo(p_lo, up, left);
    if (ADJ && has_prev) {
        load_rows(p_lo - 1, runlab);       // the plane before the tile: faces only (its runs have length 0)
    } else {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < VPL; ++j) runlab[r][j] = cur[r][j];
    }
    if (p_lo + 1 < p_hi) { load_rows(p_lo + 1, nxt); load_halo(p_lo + 1, nxt_up, nxt_left); }
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int q = 0; q < VPL / 4; ++q) a0w[r][q] = 0u;

    uint32_t fo = 0, ro = 0;                      // records written so far (wave-uniform)
    bool any_event = false;
    const uint32_t first_label = __builtin_amdgcn_readfirstlane(cur[0][0]);
    const uint32_t last = (uint32_t)(p_hi - 1 - p_lo);

    for (int64_t p = p_lo; p < p_hi; ++p) {
        const uint32_t ploc = (uint32_t)(p - p_lo);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            uint64_t mb[VPL], mc[VPL], ma[VPL];
            uint32_t pcv[VPL];
            int nev = 0;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const uint32_t v = cur[r][j];
                if (ADJ) {
                    if (r > 0) mb[j] = __builtin_amdgcn_ballot_w64(v != cur[r > 0 ? r - 1 : 0][j]);
                    else mb[j] = has_up ? __builtin_amdgcn_ballot_w64(v != up[j]) : 0ull;
                    pcv[j] = j > 0 ? cur[r][j > 0 ? j - 1 : 0]
                                   : lane_shr1(cur[r][VPL - 1], has_left ? left[r] : cur[r][0]);
                    mc[j] = __builtin_amdgcn_ballot_w64(v != pcv[j]);
                    nev += __popcll(mb[j]) + __popcll(mc[j]);
                } else {
                    mb[j] = 0ull; mc[j] = 0ull; pcv[j] = 0u;
                }
                ma[j] = __builtin_amdgcn_ballot_w64(v != runlab[r][j]);
                nev += __popcll(ma[j]);
            }
            if (nev == 0) continue;                       // the common case: one branch per row
            any_event = true;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const uint32_t v = cur[r][j];
                if (ADJ) {
                    if (r > 0 || has_up) {
                        const uint32_t pv = r > 0 ? cur[r > 0 ? r - 1 : 0][j] : up[j];
                        const uint32_t idx = fo + mbcnt64(mb[j]);
                        if (v != pv && idx < fcap) fr[idx] = make_uint2(v, pv | (1u << 30));
                        fo += (uint32_t)__popcll(mb[j]);
                    }
                    {
                        const uint32_t idx = fo + mbcnt64(mc[j]);
                        if (v != pcv[j] && idx < fcap) fr[idx] = make_uint2(v, pcv[j] | (2u << 30));
                        fo += (uint32_t)__popcll(mc[j]);
                    }
                }
                {   // axis 0: closes the column's open run (length ploc - a0; 0 for the plane before the tile)
                    const uint32_t o = runlab[r][j];
                    const uint32_t i = mbcnt64(ma[j]);
                    const uint32_t sh = 8u * (j % 4);
                    const uint32_t a0 = (a0w[r][j / 4] >> sh) & 0xffu;
                    if (v != o) {
                        if (ADJ && fo + i < fcap) fr[fo + i] = make_uint2(v, o);
                        if (ro + i < rcap)
                            rr[ro + i] = make_uint2(o, (lane_c + j) | ((uint32_t)(w * RB + r) << 10) | (a0 << 14) | ((ploc - a0) << 20));
                        a0w[r][j / 4] = (a0w[r][j / 4] & ~(0xffu << sh)) | (ploc << sh);
                        runlab[r][j] = v;
                    }
                    const uint32_t n = (uint32_t)__popcll(ma[j]);
                    if (ADJ) fo += n;
                    ro += n;
                }
            }
        }
        if (p + 1 < p_hi) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int j = 0; j < VPL; ++j) cur[r][j] = nxt[r][j];
                left[r] = nxt_left[r];
            }
#pragma unroll
            for (int j = 0; j < VPL; ++j) up[j] = nxt_up[j];
            if (p + 2 < p_hi) { load_rows(p + 2, nxt); load_halo(p + 2, nxt_up, nxt_left); }
        }
    }

    // ---- end of tile: a uniform wave tile is one header word, otherwise every column closes its run
    uint32_t ulabel = INVALID_LABEL;
    if (ADJ && !any_event) {
        ulabel = first_label;                 // may itself be INVALID_LABEL: the tile lies outside the volume
        if (ulabel == INVALID_LABEL) ulabel = INVALID_LABEL - 1u;     // "nothing here", distinct from "not uniform"
    } else {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const uint32_t a0 = (a0w[r][j / 4] >> (8u * (j % 4))) & 0xffu;
                const uint32_t idx = ro + (uint32_t)lane;
                if (idx < rcap)
                    rr[idx] = make_uint2(runlab[r][j], (lane_c + j) | ((uint32_t)(w * RB + r) << 10) | (a0 << 14) | ((last + 1u - a0) << 20));
                ro += 64u;
            }
        }
    }
    if (lane == 0) {
        *reinterpret_cast<uint4*>(hdr) = make_uint4(fo, ro, ulabel, last);
        if (fo > fcap || ro > rcap) atomicOr(&A.flags[FLAG_REGION_OVERFLOW], 1u);
    }
}

template <typename T, int VPL, int RB, bool ADJ>
__global__ void __launch_bounds__(WAVES * 64, TA_EMIT_MINWAVES) emit_kernel(SplitArgs P) {
    const SweepArgs& A = P.a;
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    uint32_t* hdr = P.rhdr + wt * 4;
    if (p_lo >= p_hi) {
        if (lane == 0) *reinterpret_cast<uint4*>(hdr) = make_uint4(0u, 0u, INVALID_LABEL - 1u, 0u);
        return;
    }
    const bool interior = A.vec_ok && (c_tile0 + TC <= A.n2) && (b_tile0 + (int64_t)(w + 1) * RB <= A.n1);
    wave_emit<T, VPL, RB, ADJ>(A, P.fcap, P.rcap, !interior, lane, w, c_tile0, b_tile0, p_lo, p_hi,
                               reinterpret_cast<uint2*>(P.frec + wt * P.fcap),
                               reinterpret_cast<uint2*>(P.rrec + wt * P.rcap), hdr);
}

template <int NW>
struct __attribute__((aligned(16))) ReduceLds {      // the workgroup tables of SweepLds, without the rings
    uint64_t lsum[LSLOTS * NW];
    uint64_t pkeys[PSLOTS];
    uint32_t lbox[LSLOTS * 8];
    uint32_t lkeys[LSLOTS];
    uint32_t pcnt[PSLOTS * 3];
};

template <int VPL, int RB, bool ADJ, bool MOM2>
__global__ void __launch_bounds__(WAVES * 64) reduce_kernel(SplitArgs P) {
    const SweepArgs& A = P.a;
    constexpr int NW = MOM2 ? 6 : 2;
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    using LDS = ReduceLds<NW>;
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
    __syncthreads();

    const int64_t tiles_c = (A.n2 + TC - 1) / TC, tiles_b = (A.n1 + TB - 1) / TB;
    int64_t t = blockIdx.x;
    const int64_t tc = t % tiles_c; t /= tiles_c;
    const int64_t tb = t % tiles_b;
    const int64_t ta_ = t / tiles_b;
    const int64_t p_lo = A.first_owned + ta_ * A.tile_planes;
    TileFrame F;
    F.A0 = (uint64_t)(A.a_origin + (p_lo - A.first_owned)); F.B0 = (uint64_t)(tb * TB); F.C0 = (uint64_t)(tc * TC);

    const uint64_t wt = (uint64_t)blockIdx.x * WAVES + (uint64_t)w;
    const uint4 hdr = *reinterpret_cast<const uint4*>(P.rhdr + wt * 4);
    const uint32_t nf = hdr.x < P.fcap ? hdr.x : P.fcap, nr = hdr.y < P.rcap ? hdr.y : P.rcap;
    const uint32_t ulabel = hdr.z, last = hdr.w;
    if (ADJ) {
        const uint2* fr = reinterpret_cast<const uint2*>(P.frec + wt * P.fcap);
        uint2 nxt = (uint32_t)lane < nf ? fr[lane] : make_uint2(INVALID_LABEL, 0u);
        for (uint32_t i = 0; i < nf; i += 64u) {
            const uint2 rec = nxt;
            const uint32_t nidx = i + 64u + (uint32_t)lane;
            nxt = nidx < nf ? fr[nidx] : make_uint2(INVALID_LABEL, 0u);          // next batch in flight
            const uint32_t v = rec.x, pv = rec.y & 0x3fffffffu, axis = rec.y >> 30;
            if (v != INVALID_LABEL && pv < LABEL_LIMIT) lds_pair_add(A, S, pv, v, axis, 1u);
        }
    }
    {
        const uint2* rr = reinterpret_cast<const uint2*>(P.rrec + wt * P.rcap);
        uint2 nxt = (uint32_t)lane < nr ? rr[lane] : make_uint2(INVALID_LABEL, 0u);
        for (uint32_t i = 0; i < nr; i += 64u) {
            const uint2 rec = nxt;
            const uint32_t nidx = i + 64u + (uint32_t)lane;
            nxt = nidx < nr ? rr[nidx] : make_uint2(INVALID_LABEL, 0u);
            consume_run_record<MOM2, LDS>(A, S, F, rec.x, rec.y);
        }
    }
    if (lane == 0 && ulabel < LABEL_LIMIT) {
        // the wave tile is one label: one closed-form box in tile-local coordinates
        const uint64_t na = last + 1u, nb = RB, nc = TC, b0 = (uint64_t)w * RB;
        const uint64_t sa = range_sum1(0, na), sb = range_sum1(b0, nb), sc = range_sum1(0, nc);
        LocalSums L;
        L.n = na * nb * nc; L.sa = sa * nb * nc; L.sb = sb * na * nc; L.sc = sc * na * nb;
        if (MOM2) {
            L.saa = range_sum2(0, na) * nb * nc; L.sab = sa * sb * nc; L.sac = sa * sc * nb;
            L.sbb = range_sum2(b0, nb) * na * nc; L.sbc = sb * sc * na; L.scc = range_sum2(0, nc) * na * nb;
        } else {
            L.saa = L.sab = L.sac = L.sbb = L.sbc = L.scc = 0;
        }
        lds_label_add<MOM2, LDS, LocalSums>(A, S, F, ulabel, L, 0u, last, (uint32_t)b0, (uint32_t)(b0 + nb - 1), 0u,
                                            (uint32_t)(nc - 1));
    }
    __syncthreads();
    flush_tables<NW, ADJ, MOM2, false>(A, S, tid, F.A0, F.B0, F.C0, 0u);
}

template <int VPL, int RB>
static void split_shape_t(const SweepArgs& a, uint64_t* wave_tiles, uint32_t* fcap, uint32_t* rcap) {
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    const int64_t owned = a.n0 - a.first_owned;
    const int64_t tiles = owned <= 0 ? 0 : ((a.n2 + TC - 1) / TC) * ((a.n1 + TB - 1) / TB) *
                                               ((owned + a.tile_planes - 1) / a.tile_planes);
    const uint64_t vox = (uint64_t)RB * TC * (uint64_t)a.tile_planes;       // voxels of one wave tile
    *wave_tiles = (uint64_t)tiles * WAVES;
    *fcap = (uint32_t)(vox / 4);                                            // tissue: ~0.10 face events per voxel
    *rcap = (uint32_t)(vox / 8 + (uint64_t)RB * VPL * 64);                  // ~0.04 closes per voxel + the tile-end closes
}

void split_region_shape(const SweepArgs& a, int itemsize, uint64_t* wave_tiles, uint32_t* fcap, uint32_t* rcap) {
    if (itemsize == 2) split_shape_t<8, 2>(a, wave_tiles, fcap, rcap);
    else               split_shape_t<4, TA_RB32>(a, wave_tiles, fcap, rcap);
}

template <typename T, int VPL, int RB>
static void launch_split_t(hipStream_t s, const SplitArgs& a, uint32_t fm) {
    uint64_t wave_tiles; uint32_t fcap, rcap;
    split_shape_t<VPL, RB>(a.a, &wave_tiles, &fcap, &rcap);
    if (wave_tiles == 0 || a.a.n1 <= 0 || a.a.n2 <= 0) return;
    const dim3 grid((unsigned)(wave_tiles / WAVES)), block(WAVES * 64);
    const bool adj = fm & 16u, mom2 = fm & 8u;
    if (adj) hipLaunchKernelGGL((emit_kernel<T, VPL, RB, true>), grid, block, 0, s, a);
    else     hipLaunchKernelGGL((emit_kernel<T, VPL, RB, false>), grid, block, 0, s, a);
    if (adj && mom2)       hipLaunchKernelGGL((reduce_kernel<VPL, RB, true, true>), grid, block, 0, s, a);
    else if (adj && !mom2) hipLaunchKernelGGL((reduce_kernel<VPL, RB, true, false>), grid, block, 0, s, a);
    else if (!adj && mom2) hipLaunchKernelGGL((reduce_kernel<VPL, RB, false, true>), grid, block, 0, s, a);
    else                   hipLaunchKernelGGL((reduce_kernel<VPL, RB, false, false>), grid, block, 0, s, a);
}

void launch_split(hipStream_t s, const SplitArgs& a, int itemsize, uint32_t feature_mask) {
    if (itemsize == 2) launch_split_t<uint16_t, 8, 2>(s, a, feature_mask);
    else               launch_split_t<uint32_t, 4, TA_RB32>(s, a, feature_mask);
}

uint64_t sweep_grid_size(const SweepArgs& a, int itemsize) {
    uint64_t wave_tiles; uint32_t f, r;
    split_region_shape(a, itemsize, &wave_tiles, &f, &r);
    return wave_tiles / WAVES;
}

int sweep_default_tile_planes() { return 32; }   // 64 planes start to overflow the 128-slot label table
int sweep_max_tile_planes() { return MAX_TILE_PLANES; }

template <typename T, int VPL, int RB>
static void launch_sweep_t(hipStream_t s, const SweepArgs& a, uint32_t fm) {
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    const int64_t owned = a.n0 - a.first_owned;
    if (owned <= 0 || a.n1 <= 0 || a.n2 <= 0) return;
    const int64_t tiles = ((a.n2 + TC - 1) / TC) * ((a.n1 + TB - 1) / TB) *
                          ((owned + a.tile_planes - 1) / a.tile_planes);
    const dim3 grid((unsigned)tiles), block(WAVES * 64);
    const bool adj = fm & 16u, mom2 = fm & 8u;
    if (adj && mom2)       hipLaunchKernelGGL((sweep_kernel<T, VPL, RB, true, true>), grid, block, 0, s, a);
    else if (adj && !mom2) hipLaunchKernelGGL((sweep_kernel<T, VPL, RB, true, false>), grid, block, 0, s, a);
    else if (!adj && mom2) hipLaunchKernelGGL((sweep_kernel<T, VPL, RB, false, true>), grid, block, 0, s, a);
    else                   hipLaunchKernelGGL((sweep_kernel<T, VPL, RB, false, false>), grid, block, 0, s, a);
}

void launch_sweep(hipStream_t s, const SweepArgs& a, int itemsize, uint32_t feature_mask) {
    if (itemsize == 2) launch_sweep_t<uint16_t, 8, 2>(s, a, feature_mask);
    else               launch_sweep_t<uint32_t, 4, TA_RB32>(s, a, feature_mask);
}

}  // namespace ta
