// kernels_sweep.hip -- the fused single-sweep feature extractor for gfx950 (MI355X).
//
// One coalesced pass over the labelled volume produces, per label, the exact integer
// accumulators behind SpatialImageAnalysis.volume / boundingbox / center_of_mass / inertia_axis
// (SIA:1197-1292, 417-535) and, per unordered label pair, the per-axis shared-face counts
// behind neighbors / cell_wall_area / wall_areas (SIA:538-660, 908-993).
//
// Work decomposition (memory axes: 0 slowest ... 2 fastest)
//   workgroup = 4 waves stacked along axis 1; wave = RB rows x (64 lanes * VPL voxels) along
//   axis 2; each lane keeps its RB x VPL voxels of the current plane in VGPRs (16-byte loads)
//   and the workgroup walks `tile_planes` planes along axis 0 with the next plane prefetched.
//   * axis-0 neighbours are the lane's own registers from the previous plane,
//     axis-1 neighbours the row above in the same lane (+ one halo row per wave),
//     axis-2 neighbours the previous voxel of the strip (+ one DPP wave-shift for voxel 0).
//   * a label change along axis 0 closes a *column run*; a run (label, a0..a1, b, c) yields all
//     ten moments in closed form, so moments cost one event per ~cell-diameter voxels, and that
//     event is the same compare that detects the axis-0 face.
//   * events are sparse per lane, so they are compacted (v_cmp mask -> mbcnt) into a small
//     per-wave LDS ring and consumed 64 at a time with every lane busy.  The consumer updates
//     two workgroup-shared LDS hash tables with LDS atomics: label -> {10 x u64 sums, bbox} and
//     (lo,hi) -> 3 face counters.  The tables are flushed once per tile with global atomics
//     (u64 add / i32 min), which is what makes the result independent of tiling and of order.
//   * the kernel body is a resumable "event pump": producer phases (one per voxel slot and
//     axis) are straight-line code with static register indices and there is exactly ONE
//     consumer site, reached by a forward jump when the ring holds >= 64 records.
//   * no MFMA: this is integer compare/reduce work bound by the HBM read of the volume.
#include "ta_kernels.h"

namespace ta {

constexpr int WAVES = 4;          // waves per workgroup, stacked along axis 1
constexpr int QCAP = 128;         // per-wave ring capacity (records); a phase adds <= 64
constexpr int LSLOTS = 128;       // label table slots per workgroup
constexpr int PSLOTS = 512;       // pair table slots per workgroup
constexpr int LPROBE = 16;        // max probes before spilling to global atomics
constexpr int PPROBE = 32;

constexpr uint32_t META_RUN = 1u << 18, META_FACE = 1u << 19;

template <int NS>
struct __attribute__((aligned(16))) SweepLds {
    uint4 q[WAVES * QCAP];        // event rings: {a/old, b/new, run(a0|a1<<16), meta}
    uint64_t lsum[LSLOTS * NS];
    uint64_t pkeys[PSLOTS];
    uint32_t lkeys[LSLOTS];
    uint32_t lbox[LSLOTS * 6];    // min0,min1,min2 (u32 min) | max0,max1,max2 (u32 max), global coords
    uint32_t pcnt[PSLOTS * 3];
};

__device__ __forceinline__ uint32_t lane_shr1(uint32_t src, uint32_t lane0_value) {
    // lane i <- src of lane i-1 ; lane 0 keeps lane0_value   (DPP wave_shr:1)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0_value, (int)src, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---- strip loads ---------------------------------------------------------------------------
template <typename T, int VPL, bool EDGE>
__device__ __forceinline__ void load_strip(const T* row, bool row_ok, int64_t c, int64_t n2,
                                           uint32_t (&dst)[VPL]) {
    if (!EDGE) {
        const uint4 x = *reinterpret_cast<const uint4*>(row + c);
        if (sizeof(T) == 4) {
            dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
        } else {
            dst[0] = x.x & 0xffffu; dst[1] = x.x >> 16; dst[2] = x.y & 0xffffu; dst[3] = x.y >> 16;
            dst[4 % VPL] = x.z & 0xffffu; dst[5 % VPL] = x.z >> 16;
            dst[6 % VPL] = x.w & 0xffffu; dst[7 % VPL] = x.w >> 16;
        }
    } else {
#pragma unroll
        for (int j = 0; j < VPL; ++j)
            dst[j] = (row_ok && c + j < n2) ? (uint32_t)row[c + j] : INVALID_LABEL;
    }
}

// ---- the wave body ---------------------------------------------------------------------------
template <typename T, int VPL, int RB, bool ADJ, bool MOM2, bool EDGE>
__device__ __forceinline__ void wave_sweep(const SweepArgs& A, SweepLds<MOM2 ? 10 : 4>& S,
                                           const int lane, const int w, const int64_t c_tile0,
                                           const int64_t b_tile0, const int64_t p_lo,
                                           const int64_t p_hi) {
    constexpr int NS = MOM2 ? 10 : 4;
    constexpr int NSLOT = RB * VPL;
    // phases: [0, 2*NSLOT) face phases (axis 1 then axis 2 per slot), [PA, PA+NSLOT) axis-0 phases,
    // P_ADV plane advance, [PE, PE+NSLOT) end-of-tile run flush, P_DONE.
    constexpr int PA = 2 * NSLOT, P_ADV = PA + NSLOT, PE = P_ADV + 1, P_DONE = PE + NSLOT;

    const T* vol = reinterpret_cast<const T*>(A.vol);
    const int64_t n1 = A.n1, n2 = A.n2, plane = n1 * n2;
    const int64_t b_wave0 = b_tile0 + (int64_t)w * RB;
    const int64_t c0 = c_tile0 + (int64_t)lane * VPL;
    const bool has_up = b_wave0 > 0;
    const bool has_left = c_tile0 > 0;
    const bool has_prev = p_lo > 0;
    const int qbase = w * QCAP;

    uint32_t cur[RB][VPL], nxt[RB][VPL], runlab[RB][VPL], a0s[RB][VPL];
    uint32_t up[VPL], nxt_up[VPL], left[RB], nxt_left[RB];

    auto load_plane = [&](int64_t p, uint32_t (&d)[RB][VPL], uint32_t (&dup)[VPL], uint32_t (&dl)[RB]) {
        const T* pbase = vol + p * plane;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int64_t b = b_wave0 + r;
            const bool row_ok = b < n1;
            const T* row = pbase + (EDGE ? (row_ok ? b : 0) : b) * n2;
            load_strip<T, VPL, EDGE>(row, row_ok, c0, n2, d[r]);
            if (ADJ) {
                dl[r] = INVALID_LABEL;
                if (has_left && lane == 0 && row_ok) dl[r] = (uint32_t)row[c_tile0 - 1];
            }
        }
        if (ADJ) {
            if (has_up) {
                const bool row_ok = (b_wave0 - 1) < n1;
                const T* row = pbase + (EDGE ? (row_ok ? (b_wave0 - 1) : 0) : (b_wave0 - 1)) * n2;
                load_strip<T, VPL, EDGE>(row, row_ok, c0, n2, dup);
            } else {
#pragma unroll
                for (int j = 0; j < VPL; ++j) dup[j] = INVALID_LABEL;
            }
        }
    };

    // ---- prologue: plane before the tile (faces only), first plane, prefetch of the second
    if (has_prev) {
        uint32_t tmp_up[VPL], tmp_left[RB];
        // only the voxels themselves matter for the previous plane
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int64_t b = b_wave0 + r;
            const bool row_ok = b < n1;
            const T* row = vol + (p_lo - 1) * plane + (EDGE ? (row_ok ? b : 0) : b) * n2;
            load_strip<T, VPL, EDGE>(row, row_ok, c0, n2, runlab[r]);
        }
        (void)tmp_up; (void)tmp_left;
    } else {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < VPL; ++j) runlab[r][j] = INVALID_LABEL;
    }
    load_plane(p_lo, cur, up, left);
    if (p_lo + 1 < p_hi) load_plane(p_lo + 1, nxt, nxt_up, nxt_left);
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int j = 0; j < VPL; ++j) a0s[r][j] = 0;

    int64_t p = p_lo;           // plane being processed
    bool first = true;          // processing the first owned plane of the tile
    bool finished = false;
    int phase = ADJ ? 0 : PA;
    int head = 0, tail = 0;     // wave-uniform ring cursors (free-running)

#define TA_EMIT(EV, RA, RBV, RRUN, RMETA)                                                   \
    do {                                                                                    \
        const uint64_t m_ = __builtin_amdgcn_ballot_w64(EV);                                \
        if (m_) {                                                                           \
            const uint32_t i_ = mbcnt64(m_);                                                \
            if (EV) S.q[qbase + ((tail + (int)i_) & (QCAP - 1))] = make_uint4((RA), (RBV), (RRUN), (RMETA)); \
            tail += __popcll(m_);                                                           \
        }                                                                                   \
    } while (0)

    for (;;) {
        // ------------------------------------------------------------ producer: face phases
        if (ADJ && phase < PA) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const int s = r * VPL + j;
                    if (phase == 2 * s) {                      // axis 1: voxel vs the row above
                        const uint32_t v = cur[r][j];
                        const uint32_t pv = (r == 0) ? up[j] : cur[r == 0 ? 0 : r - 1][j];
                        bool ev = v != pv;
                        if (EDGE || r == 0) ev = ev && (v != INVALID_LABEL) && (pv != INVALID_LABEL);
                        TA_EMIT(ev, pv, v, 0u, (1u << 16) | META_FACE);
                        phase = 2 * s + 1;
                        if (tail - head >= 64) goto consume;
                    }
                    if (phase == 2 * s + 1) {                  // axis 2: voxel vs its predecessor
                        const uint32_t v = cur[r][j];
                        uint32_t pv;
                        if (j == 0) pv = lane_shr1(cur[r][VPL - 1], left[r]);
                        else pv = cur[r][j == 0 ? 0 : j - 1];
                        bool ev = v != pv;
                        if (EDGE || j == 0) ev = ev && (v != INVALID_LABEL) && (pv != INVALID_LABEL);
                        TA_EMIT(ev, pv, v, 0u, (2u << 16) | META_FACE);
                        phase = 2 * s + 2;
                        if (tail - head >= 64) goto consume;
                    }
                }
            }
        }
        // ------------------------------------------------------------ producer: axis-0 phases
        if (phase >= PA && phase < P_ADV) {
            const uint32_t ploc = (uint32_t)(p - p_lo);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const int s = r * VPL + j;
                    if (phase == PA + s) {
                        const uint32_t v = cur[r][j], old = runlab[r][j];
                        const uint32_t pos = (uint32_t)(lane * VPL + j) | ((uint32_t)(w * RB + r) << 10);
                        bool ev = v != old;
                        if (first) {
                            // `old` is the plane before the tile: a face, never a run of this tile
                            if (ADJ && has_prev) {
                                ev = ev && (v != INVALID_LABEL) && (old != INVALID_LABEL);
                                TA_EMIT(ev, old, v, 0u, META_FACE);
                            }
                        } else {
                            TA_EMIT(ev, old, v, a0s[r][j] | ((ploc - 1u) << 16),
                                    pos | META_RUN | (ADJ ? META_FACE : 0u));
                            a0s[r][j] = ev ? ploc : a0s[r][j];
                        }
                        runlab[r][j] = v;
                        phase = PA + s + 1;
                        if (tail - head >= 64) goto consume;
                    }
                }
            }
        }
        // ------------------------------------------------------------ plane advance
        if (phase == P_ADV) {
            first = false;
            ++p;
            if (p < p_hi) {
#pragma unroll
                for (int r = 0; r < RB; ++r) {
#pragma unroll
                    for (int j = 0; j < VPL; ++j) cur[r][j] = nxt[r][j];
                    left[r] = nxt_left[r];
                }
#pragma unroll
                for (int j = 0; j < VPL; ++j) up[j] = nxt_up[j];
                if (p + 1 < p_hi) load_plane(p + 1, nxt, nxt_up, nxt_left);
                phase = ADJ ? 0 : PA;
                continue;
            }
            phase = PE;
        }
        // ------------------------------------------------------------ end of tile: close every run
        if (phase >= PE && phase < P_DONE) {
            const uint32_t last = (uint32_t)(p_hi - 1 - p_lo);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const int s = r * VPL + j;
                    if (phase == PE + s) {
                        const uint32_t pos = (uint32_t)(lane * VPL + j) | ((uint32_t)(w * RB + r) << 10);
                        const bool ev = runlab[r][j] != INVALID_LABEL;
                        TA_EMIT(ev, runlab[r][j], 0u, a0s[r][j] | (last << 16), pos | META_RUN);
                        phase = PE + s + 1;
                        if (tail - head >= 64) goto consume;
                    }
                }
            }
        }
        if (phase == P_DONE) finished = true;

    consume:
        // ------------------------------------------------------------ the single consumer site
        for (;;) {
            const int cnt = tail - head;
            if (cnt < 64 && !(finished && cnt > 0)) break;
            __builtin_amdgcn_wave_barrier();
            const uint4 rec = S.q[qbase + ((head + lane) & (QCAP - 1))];
            const bool act = lane < cnt;
            head += cnt < 64 ? cnt : 64;
            if (act) {
                const uint32_t meta = rec.w;
                if (ADJ && (meta & META_FACE)) {
                    const uint32_t lo = rec.x < rec.y ? rec.x : rec.y, hi = rec.x < rec.y ? rec.y : rec.x;
                    const uint32_t axis = (meta >> 16) & 3u;
                    const uint64_t key = ((uint64_t)lo << 32) | hi;
                    uint32_t h = hash_pair(lo, hi) & (PSLOTS - 1);
                    int slot = -1;
                    for (int probe = 0; probe < PPROBE; ++probe) {
                        uint64_t k = S.pkeys[h];
                        if (k == EMPTY_KEY) {
                            k = atomicCAS((unsigned long long*)&S.pkeys[h], (unsigned long long)EMPTY_KEY,
                                          (unsigned long long)key);
                            if (k == EMPTY_KEY) k = key;
                        }
                        if (k == key) { slot = (int)h; break; }
                        h = (h + 1) & (PSLOTS - 1);
                    }
                    if (slot >= 0) {
                        atomicAdd(&S.pcnt[slot * 3 + axis], 1u);
                    } else {
                        pair_add_global(A.pairs, lo, hi, axis == 0, axis == 1, axis == 2, A.flags);
                        atomicOr(&A.flags[FLAG_LDS_PAIR_SPILL], 1u);
                    }
                }
                if (meta & META_RUN) {
                    const uint32_t label = rec.x;
                    const uint32_t a0l = rec.z & 0xffffu, a1l = rec.z >> 16;
                    const uint64_t gc = (uint64_t)(c_tile0 + (int64_t)(meta & 1023u));
                    const uint64_t gb = (uint64_t)(b_tile0 + (int64_t)((meta >> 10) & 63u));
                    const uint64_t ga0 = (uint64_t)(A.a_origin + (p_lo + a0l - A.first_owned));
                    const uint32_t n = a1l - a0l + 1u;
                    uint32_t h = hash_u32(label) & (LSLOTS - 1);
                    int slot = -1;
                    for (int probe = 0; probe < LPROBE; ++probe) {
                        uint32_t k = S.lkeys[h];
                        if (k == INVALID_LABEL) {
                            k = atomicCAS(&S.lkeys[h], INVALID_LABEL, label);
                            if (k == INVALID_LABEL) k = label;
                        }
                        if (k == label) { slot = (int)h; break; }
                        h = (h + 1) & (LSLOTS - 1);
                    }
                    if (slot >= 0) {
                        uint64_t sv[NSUM];
                        run_moments<MOM2>(ga0, n, gb, gc, sv);
                        unsigned long long* row = (unsigned long long*)&S.lsum[slot * NS];
#pragma unroll
                        for (int k = 0; k < NS; ++k) atomicAdd(row + k, (unsigned long long)sv[k]);
                        uint32_t* box = &S.lbox[slot * 6];
                        atomicMin(box + 0, (uint32_t)ga0); atomicMax(box + 3, (uint32_t)(ga0 + n - 1));
                        atomicMin(box + 1, (uint32_t)gb);  atomicMax(box + 4, (uint32_t)gb);
                        atomicMin(box + 2, (uint32_t)gc);  atomicMax(box + 5, (uint32_t)gc);
                    } else {
                        run_add_global<MOM2>(A, label, ga0, n, gb, gc);
                        atomicOr(&A.flags[FLAG_LDS_LABEL_SPILL], 1u);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (finished) break;
    }
#undef TA_EMIT
}

template <typename T, int VPL, int RB, bool ADJ, bool MOM2>
__global__ void __launch_bounds__(WAVES * 64) sweep_kernel(SweepArgs A) {
    constexpr int NS = MOM2 ? 10 : 4;
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    __shared__ SweepLds<NS> S;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < LSLOTS; i += WAVES * 64) {
        S.lkeys[i] = INVALID_LABEL;
#pragma unroll
        for (int k = 0; k < NS; ++k) S.lsum[i * NS + k] = 0ull;
        S.lbox[i * 6 + 0] = 0xFFFFFFFFu; S.lbox[i * 6 + 1] = 0xFFFFFFFFu; S.lbox[i * 6 + 2] = 0xFFFFFFFFu;
        S.lbox[i * 6 + 3] = 0u; S.lbox[i * 6 + 4] = 0u; S.lbox[i * 6 + 5] = 0u;
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
    const int64_t c_tile0 = tc * TC, b_tile0 = tb * TB;
    const int64_t p_lo = A.first_owned + ta_ * A.tile_planes;
    int64_t p_hi = p_lo + A.tile_planes;
    if (p_hi > A.n0) p_hi = A.n0;

    if (p_lo < p_hi) {
        const bool interior = A.vec_ok && (c_tile0 + TC <= A.n2) && (b_tile0 + (int64_t)(w + 1) * RB <= A.n1);
        if (interior) wave_sweep<T, VPL, RB, ADJ, MOM2, false>(A, S, lane, w, c_tile0, b_tile0, p_lo, p_hi);
        else          wave_sweep<T, VPL, RB, ADJ, MOM2, true>(A, S, lane, w, c_tile0, b_tile0, p_lo, p_hi);
    }
    __syncthreads();

    // ---- flush the workgroup tables with global atomics
    for (int i = tid; i < LSLOTS; i += WAVES * 64) {
        const uint32_t label = S.lkeys[i];
        if (label == INVALID_LABEL) continue;
        if (label > A.max_label) { atomicOr(&A.flags[FLAG_RANGE], 1u); continue; }
        unsigned long long* row = (unsigned long long*)&A.sums[(uint64_t)label * NSUM];
#pragma unroll
        for (int k = 0; k < NS; ++k) atomicAdd(row + k, (unsigned long long)S.lsum[i * NS + k]);
        int32_t* box = &A.boxes[(uint64_t)label * NBOX];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            atomicMin(box + d, (int32_t)S.lbox[i * 6 + d]);
            atomicMin(box + 3 + d, -(int32_t)S.lbox[i * 6 + 3 + d]);
        }
    }
    if (ADJ) {
        for (int i = tid; i < PSLOTS; i += WAVES * 64) {
            const uint64_t key = S.pkeys[i];
            if (key == EMPTY_KEY) continue;
            pair_add_global(A.pairs, (uint32_t)(key >> 32), (uint32_t)key, S.pcnt[i * 3 + 0],
                            S.pcnt[i * 3 + 1], S.pcnt[i * 3 + 2], A.flags);
        }
    }
}

int sweep_default_tile_planes() { return 32; }

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
    else               launch_sweep_t<uint32_t, 4, 4>(s, a, feature_mask);
}

}  // namespace ta
