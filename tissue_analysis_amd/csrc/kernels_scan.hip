// kernels_scan.hip -- THE sweep: one coalesced pass over the labelled volume (TA_OPT_IMPL = 0).
//
// One pass produces, per label, the exact integer accumulators behind SpatialImageAnalysis.volume /
// boundingbox / center_of_mass / inertia_axis (SIA:1197-1292, 417-535) and, per unordered label pair, the
// per-axis shared-face counts behind neighbors / cell_wall_area / wall_areas (SIA:538-660, 908-993).
//
// Work decomposition (memory axes: 0 slowest ... 2 fastest): workgroup = WAVES waves stacked along axis 1;
// wave tile = RB rows x (64 lanes * VPL voxels) along axis 2; each lane keeps its RB x VPL voxels of the
// current and the previous plane in VGPRs (16-byte loads, the next plane in flight) and the workgroup walks
// `tile_planes` planes along axis 0.  Neighbours never leave the register file: axis 0 = the previous plane's
// registers, axis 1 = the row above in the same lane (+ one halo row per wave), axis 2 = the previous voxel
// of the strip (+ one DPP wave shift, + one halo voxel per row).  A label's voxels are summed as RUNS along
// axis 2 inside a row (closed forms in the column), rows of one label from end to end are not even records.
// No MFMA anywhere: integer compare / reduce work bound by the HBM read of the volume.
//
// How the (sparse: ~2 of 64 lanes per compare) events reach the dense 64-wide consumer:
//   * per row every lane counts its events (one v_addc per compare), ONE packed DPP add-scan over the
//     lanes gives each lane the offset of its first record, and every lane then stores at every compare
//     without touching the exec mask: at its own running offset when the compare fired, into a trash slot
//     when it did not.  No per-compare ballot / mask-prefix / popcount / SGPR cursor arithmetic, no live
//     64-bit masks; the axis-0 faces of a plane ride on the scan of its first row;
//   * the per-wave buffers are LINEAR and drained completely (no ring wrap arithmetic); the drain is
//     checked once per row, BEFORE the row is emitted, from the totals the scan produced (one scalar test);
//   * records of a row land sorted by column, so a run record only carries the column where its run
//     ENDS: the consumer takes the start from the record before it (same row) -- no max-scan for run
//     starts in the producer, no per-compare bookkeeping.
// Records: faces of axis 0/1 {voxel, neighbour | axis << 30}; runs {closing label, k | b << 10 | a << 14} (k = end column of
// the run = column of the voxel right of it, whose label is the closing label of the row's NEXT record).
#include "ta_sweep_common.h"

#include <type_traits>

#include <hip/hip_ext.h>

namespace ta {

// the hot (most common) label of the volume gets a private row per workgroup, also with adjacency
constexpr int FCAP = TA_FCAP, RCAP = TA_RCAP;           // record capacities of a wave's buffers
constexpr int FTRASH = FCAP, RTRASH = RCAP + 1;          // the trash slots of the branch-free stores, behind the buffers
constexpr uint32_t ROWID_MASK = 0x7FFFFC00u;            // bits of a run code that name the row (b, a, and the zero bits above)
constexpr uint32_t ROW_END = 0x80000000u;               // code bit of the record that closes a row's last run (no voxel to its right)
constexpr uint32_t NO_ROW = 0xFFFFFFFFu;                // code of the sentinel: never equal to a record's row

typedef __attribute__((address_space(3))) uint32_t* lds_u32;

template <bool ADJ>
struct __attribute__((aligned(16))) ScanWaveLds {
    uint2 frec[ADJ ? FCAP + 1 : 1];                     // faces of axis 0/1: voxel, neighbour | axis << 30; [FCAP] = trash slot
    uint32_t cql[RCAP + 2], cqc[RCAP + 2];              // runs; record i lives in slot i + 1, slot 0 = sentinel / carry, slot RCAP + 1 = trash
};

// NW packed words of sums per label slot and replica (SumPack); REP_ replicas of every slot: the runs of one label in a group of
// 64 records come from consecutive rows, and rows of different parity add into different replicas (drain_run_group)
template <int NW, bool ADJ, int REP_, typename PACK>
struct __attribute__((aligned(16))) ScanLds {
    static constexpr int REP = REP_;
    using Pack = PACK;
    ScanWaveLds<ADJ> wave[WAVES];
    uint64_t lsum[LSLOTS * NW * REP_];
    uint64_t pkeys[ADJ ? PSLOTS : 2];
#if TA_PCNT64
    uint64_t pcnt[ADJ ? PSLOTS : 1];  // three 21-bit face counts per pair (PCNT_BITS)
#else
    uint32_t pcnt[ADJ ? PSLOTS * 3 : 1];
#endif
    uint32_t lbox[LSLOTS * 8];
    uint32_t lkeys[LSLOTS];
    uint32_t frame[4];                // origin (axes 0, 1, 2) of the tile-local coordinates: only the spill paths need it
    uint32_t fcnt[2];                 // the flush: occupied label / pair slots (flush_tables gathers them first)
#ifdef TA_LDS_PAD
    uint32_t pad_[TA_LDS_PAD];        // experiments only: fewer workgroups per CU
#endif
};

// one face of `axis` for the pair in `slot`.  (Three u32 counters: an LDS atomic on 64 bits costs the CU's LDS twice the
// cycles per lane that shares its address -- profiles/r04_lds_atomics_microbench.txt -- and a pass of 64 faces holds few pairs.)
template <typename LDS>
__device__ __forceinline__ void pcnt_add(LDS& S, const uint32_t slot, const uint32_t axis) {
#if TA_PCNT64
    atomicAdd((unsigned long long*)&S.pcnt[slot], 1ull << (PCNT_BITS * axis));
#else
    atomicAdd(&S.pcnt[slot * 3u + axis], 1u);
#endif
}

// inclusive add-scan over the 64 lanes: row_shr 1,2,4,8 then the two row broadcasts
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {
#define TA_DPP_ADD(ctrl, rmask) \
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xf, false);
    TA_DPP_ADD(0x111, 0xf) TA_DPP_ADD(0x112, 0xf) TA_DPP_ADD(0x114, 0xf) TA_DPP_ADD(0x118, 0xf)
    TA_DPP_ADD(0x142, 0xa) TA_DPP_ADD(0x143, 0xc)
#undef TA_DPP_ADD
    return x;
}

// ---- record stores by the exec mask itself ----------------------------------------------------------
// A compare position of the emit pass used to cost five vector instructions whether or not it fired: compare, select the
// address (own offset or trash slot), select the step (8 or 0), store, add the step -- plus the wait states between a
// VCC write and its readers.  The sweep is bound by its vector instructions wherever there is tissue (profiles/NOTES.md,
// round 5), so the stores are predicated the cheap way instead: v_cmpx writes the compare straight into EXEC, the store and
// the step of the offset run in the lanes that fired, one scalar move puts EXEC back: two vector instructions a position.
// (Inline asm: the compiler never sees EXEC change.  No manual wait states are needed between a VALU write of EXEC and
// LDS / VALU instructions that run under it; DPP instructions are the exception -- emit_done() pads for them.)
// FULL: every lane of the wave is live at the call (EXEC is put back to all ones), else `live` is the mask to put back
#define TA_CMPX(a, b) "v_cmpx_ne_u32_e32 vcc, " a ", " b "\n\t"
#define TA_CMPX_PART(a, b) TA_CMPX(a, b)
template <bool FULL>
__device__ __forceinline__ void store_face_if_ne(uint32_t& off, const uint32_t v, const uint32_t pv, const uint64_t live) {
    if constexpr (FULL)
        asm volatile(TA_CMPX("%1", "%2") "ds_write2_b32 %0, %1, %2 offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, -1"
                     : "+v"(off) : "v"(v), "v"(pv) : "vcc", "memory");
    else
        asm volatile(TA_CMPX_PART("%1", "%2") "ds_write2_b32 %0, %1, %2 offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, %3"
                     : "+v"(off) : "v"(v), "v"(pv), "s"(live) : "vcc", "memory");
}
// ... the neighbour word tagged with the axis (`tag`: wave-uniform)
template <bool FULL>
__device__ __forceinline__ void store_face_tagged_if_ne(uint32_t& off, const uint32_t v, const uint32_t pv, const uint32_t tag, const uint64_t live) {
    uint32_t t;
    if constexpr (FULL)
        asm volatile(TA_CMPX("%2", "%3") "v_or_b32_e32 %1, %4, %3\n\tds_write2_b32 %0, %2, %1 offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, -1"
                     : "+v"(off), "=&v"(t) : "v"(v), "v"(pv), "s"(tag) : "vcc", "memory");
    else
        asm volatile(TA_CMPX_PART("%2", "%3") "v_or_b32_e32 %1, %4, %3\n\tds_write2_b32 %0, %2, %1 offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, %5"
                     : "+v"(off), "=&v"(t) : "v"(v), "v"(pv), "s"(tag), "s"(live) : "vcc", "memory");
}
// a run record {closing label, code0 | J} where v != pcv; the two run arrays lie STRIDE dwords apart
template <bool FULL, int J, int STRIDE>
__device__ __forceinline__ void store_run_if_ne(uint32_t& off, const uint32_t v, const uint32_t pcv, const uint32_t code0, const uint64_t live) {
    static_assert(STRIDE < 256, "ds_write2_b32 offsets are 8 bits of dwords");
    uint32_t t;
    if constexpr (FULL)
        asm volatile(TA_CMPX("%2", "%3") "v_or_b32_e32 %1, %5, %4\n\tds_write2_b32 %0, %3, %1 offset1:%6\n\tv_add_u32_e32 %0, 4, %0\n\ts_mov_b64 exec, -1"
                     : "+v"(off), "=&v"(t) : "v"(v), "v"(pcv), "v"(code0), "n"(J), "n"(STRIDE) : "vcc", "memory");
    else
        asm volatile(TA_CMPX_PART("%2", "%3") "v_or_b32_e32 %1, %5, %4\n\tds_write2_b32 %0, %3, %1 offset1:%6\n\tv_add_u32_e32 %0, 4, %0\n\ts_mov_b64 exec, %7"
                     : "+v"(off), "=&v"(t) : "v"(v), "v"(pcv), "v"(code0), "n"(J), "n"(STRIDE), "s"(live) : "vcc", "memory");
}
// Four positions in ONE asm statement (between separate statements the compiler pads a wait state it cannot rule out: an s_nop per
// position).  `EX`: what EXEC is put back to after each position.
#define TA_FACE_POS(v, pv, EX) TA_CMPX(v, pv) "ds_write2_b32 %0, " v ", " pv " offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, " EX "\n\t"
#define TA_FACE_POS_PART(v, pv, EX) TA_CMPX_PART(v, pv) "ds_write2_b32 %0, " v ", " pv " offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, " EX "\n\t"
template <bool FULL>
__device__ __forceinline__ void store_faces4_if_ne(uint32_t& off, const uint32_t v0, const uint32_t p0, const uint32_t v1, const uint32_t p1,
                                                   const uint32_t v2, const uint32_t p2, const uint32_t v3, const uint32_t p3, const uint64_t live) {
    if constexpr (FULL)
        asm volatile(TA_FACE_POS("%1", "%2", "-1") TA_FACE_POS("%3", "%4", "-1") TA_FACE_POS("%5", "%6", "-1") TA_FACE_POS("%7", "%8", "-1")
                     : "+v"(off) : "v"(v0), "v"(p0), "v"(v1), "v"(p1), "v"(v2), "v"(p2), "v"(v3), "v"(p3) : "vcc", "memory");
    else
        asm volatile(TA_FACE_POS_PART("%1", "%2", "%9") TA_FACE_POS_PART("%3", "%4", "%9") TA_FACE_POS_PART("%5", "%6", "%9") TA_FACE_POS_PART("%7", "%8", "%9")
                     : "+v"(off) : "v"(v0), "v"(p0), "v"(v1), "v"(p1), "v"(v2), "v"(p2), "v"(v3), "v"(p3), "s"(live) : "vcc", "memory");
}
// ... the neighbour words tagged with the axis (%1: scratch, %10: the tag)
#define TA_TFACE_POS(v, pv, EX) TA_CMPX(v, pv) "v_or_b32_e32 %1, %10, " pv "\n\tds_write2_b32 %0, " v ", %1 offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, " EX "\n\t"
#define TA_TFACE_POS_PART(v, pv, EX) TA_CMPX_PART(v, pv) "v_or_b32_e32 %1, %10, " pv "\n\tds_write2_b32 %0, " v ", %1 offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, " EX "\n\t"
template <bool FULL>
__device__ __forceinline__ void store_faces4_tagged_if_ne(uint32_t& off, const uint32_t v0, const uint32_t p0, const uint32_t v1, const uint32_t p1,
                                                          const uint32_t v2, const uint32_t p2, const uint32_t v3, const uint32_t p3,
                                                          const uint32_t tag, const uint64_t live) {
    uint32_t t;
    if constexpr (FULL)
        asm volatile(TA_TFACE_POS("%2", "%3", "-1") TA_TFACE_POS("%4", "%5", "-1") TA_TFACE_POS("%6", "%7", "-1") TA_TFACE_POS("%8", "%9", "-1")
                     : "+v"(off), "=&v"(t) : "v"(v0), "v"(p0), "v"(v1), "v"(p1), "v"(v2), "v"(p2), "v"(v3), "v"(p3), "s"(tag) : "vcc", "memory");
    else
        asm volatile(TA_TFACE_POS_PART("%2", "%3", "%11") TA_TFACE_POS_PART("%4", "%5", "%11") TA_TFACE_POS_PART("%6", "%7", "%11") TA_TFACE_POS_PART("%8", "%9", "%11")
                     : "+v"(off), "=&v"(t) : "v"(v0), "v"(p0), "v"(v1), "v"(p1), "v"(v2), "v"(p2), "v"(v3), "v"(p3), "s"(tag), "s"(live) : "vcc", "memory");
}
// ... four run records: positions J .. J + 3 of the strip; v_k against the voxel before it (p0 = the voxel left of v0); %1: scratch,
// %7: code0, %8 .. %11: J .. J + 3, %12: the dword stride between the two run arrays
#define TA_RUN_POS(v, pv, j, EX) TA_CMPX(v, pv) "v_or_b32_e32 %1, " j ", %7\n\tds_write2_b32 %0, " pv ", %1 offset1:%12\n\tv_add_u32_e32 %0, 4, %0\n\ts_mov_b64 exec, " EX "\n\t"
#define TA_RUN_POS_PART(v, pv, j, EX) TA_CMPX_PART(v, pv) "v_or_b32_e32 %1, " j ", %7\n\tds_write2_b32 %0, " pv ", %1 offset1:%12\n\tv_add_u32_e32 %0, 4, %0\n\ts_mov_b64 exec, " EX "\n\t"
template <bool FULL, int J, int STRIDE>
__device__ __forceinline__ void store_runs4_if_ne(uint32_t& off, const uint32_t p0, const uint32_t v0, const uint32_t v1, const uint32_t v2, const uint32_t v3,
                                                  const uint32_t code0, const uint64_t live) {
    static_assert(STRIDE < 256, "ds_write2_b32 offsets are 8 bits of dwords");
    uint32_t t;
    if constexpr (FULL)
        asm volatile(TA_RUN_POS("%3", "%2", "%8", "-1") TA_RUN_POS("%4", "%3", "%9", "-1") TA_RUN_POS("%5", "%4", "%10", "-1") TA_RUN_POS("%6", "%5", "%11", "-1")
                     : "+v"(off), "=&v"(t) : "v"(p0), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(code0), "n"(J), "n"(J + 1), "n"(J + 2), "n"(J + 3), "n"(STRIDE)
                     : "vcc", "memory");
    else
        asm volatile(TA_RUN_POS_PART("%3", "%2", "%8", "%13") TA_RUN_POS_PART("%4", "%3", "%9", "%13") TA_RUN_POS_PART("%5", "%4", "%10", "%13") TA_RUN_POS_PART("%6", "%5", "%11", "%13")
                     : "+v"(off), "=&v"(t) : "v"(p0), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(code0), "n"(J), "n"(J + 1), "n"(J + 2), "n"(J + 3), "n"(STRIDE), "s"(live)
                     : "vcc", "memory");
}
// behind the last predicated store of a row: a DPP instruction needs five wait states after a VALU write of EXEC
__device__ __forceinline__ void emit_done() { asm volatile("s_nop 4" ::: "memory"); }

// ---- workgroup tables, with the kernel arguments kept OUT of the hot loop -----------------------
// The sweep kernels are short of SGPRs: the twenty-odd kernel arguments that only the cold paths need
// (global rows, pair table, flags) are re-read from the kernarg segment where they are used.  The
// pointer is laundered through an empty asm so that the loads cannot be hoisted into the hot loop.
__device__ __forceinline__ const SweepArgs* cold_args(const SweepArgs* kp) {
    asm volatile("" : "+s"(kp));
    return kp;
}

// ... and where MANY of them are needed -- the flush -- the whole argument block is fetched once through the CONSTANT address
// space: a handful of s_load_dwordx4 into SGPRs.  (Through the generic pointer every field access is a per-lane FLAT load, a
// vector-memory round trip the wave waits for; the flush re-read the pair table's pointers and mask that way inside its probe
// loop and spent 9 % (C4) to 18 % (tissue everywhere) of a workgroup's life: profiles/NOTES.md, round 5.)
struct KernelArgs { SweepArgs a; uint32_t split[6]; uint32_t wg0; };      // the kernarg segment of every sweep kernel: (SweepArgs, ScanSplit, uint32_t)
__device__ __forceinline__ KernelArgs scalar_args(const SweepArgs* kp) {
    typedef const __attribute__((address_space(4))) uint32_t* cu32;
    static_assert(sizeof(SweepArgs) % 4 == 0, "(SweepArgs, ScanSplit of six words, uint32_t) back to back in the kernarg segment");
    const cu32 w = (cu32)(uintptr_t)cold_args(kp);
    KernelArgs k;
    uint32_t buf[(sizeof(SweepArgs) + 28) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(buf) / 4); ++i) buf[i] = w[i];
    __builtin_memcpy(&k.a, buf, sizeof(SweepArgs));
#pragma unroll
    for (int i = 0; i < 6; ++i) k.split[i] = buf[sizeof(SweepArgs) / 4 + i];
    k.wg0 = buf[sizeof(SweepArgs) / 4 + 6];
    return k;
}

__device__ __forceinline__ const SweepArgs* kernarg_args(const SweepArgs& by_value) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)by_value;        // the struct is the first (only explicit) kernel argument: it sits at offset 0 of the segment
    return (const SweepArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return &by_value;
#endif
}

// 24-bit multiplies by hand: written with __umul24 the compiler sees that only a few product bits are used and falls back to
// v_mul_lo_u32 / v_mad_u64_u32 -- quarter-rate instructions -- for the two hashes of every record
__device__ __forceinline__ uint32_t mul24_const(uint32_t x, uint32_t c) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(c), "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t mad24_const(uint32_t x, uint32_t c, uint32_t acc) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(c), "v"(acc));
    return r;
}

__device__ __forceinline__ uint32_t scan_label_hash(uint32_t label) {
    return (mul24_const(label, 0x9E3779u) >> (24 - LSLOTS_LOG2)) & (LSLOTS - 1);
}

// `h`, `k` = the label's home slot and the key read from it: the caller issues that read early, together with the
// other table's, so the two LDS round trips overlap.  (A key read as empty may be taken by now: the CAS tells.)
template <bool MOM2, typename LDS, typename SUMS>
__device__ __forceinline__ void scan_label_add(const SweepArgs* kp, LDS& S,
                                               uint32_t label, const SUMS& L, uint32_t mna, uint32_t mxa, uint32_t mnb,
                                               uint32_t mxb, uint32_t mnc, uint32_t mxc, uint32_t h, uint32_t k) {
    constexpr int NW = MOM2 ? 4 : 2;
    int slot = -1;
#pragma nounroll
    for (int probe = 0; probe < LPROBE; ++probe) {
        if (k == INVALID_LABEL) {
            k = atomicCAS(&S.lkeys[h], INVALID_LABEL, label);
            if (k == INVALID_LABEL) k = label;
        }
        if (k == label) { slot = (int)h; break; }
        h = (h + 1) & (LSLOTS - 1);
        k = S.lkeys[h];
    }
    if (slot >= 0) {
        unsigned long long* row = (unsigned long long*)&S.lsum[slot * NW * LDS::REP];      // (replica 0)
        uint64_t w[4];
        LDS::Pack::template pack<MOM2>(L, w);
#ifndef TA_ABL_NOSUMS          // (ablations: results wrong by construction, only the time matters)
#pragma unroll
        for (int k = 0; k < NW; ++k) atomicAdd(row + k, (unsigned long long)w[k]);
#endif
        // bounding box: read first, touch the atomics only when this contribution extends it
        uint32_t* box = &S.lbox[slot * 8];
        const uint4 cur = *reinterpret_cast<const uint4*>(box);          // min a,b,c | max a
        const uint2 cur2 = *reinterpret_cast<const uint2*>(box + 4);     // max b,c
        // (one test for "this contribution extends the box somewhere": six separately guarded atomics are six exec-mask
        //  round trips per pass, and almost every run lies inside the box its label already has)
#ifdef TA_ABL_NOBOX
        const bool grows = (mna + cur.x + cur2.x == 0x12345u);
#else
        const bool grows = (mna < cur.x) | (mnb < cur.y) | (mnc < cur.z) | (mxa > cur.w) | (mxb > cur2.x) | (mxc > cur2.y);
#endif
        if (grows) {
            atomicMin(box + 0, mna); atomicMin(box + 1, mnb); atomicMin(box + 2, mnc);
            atomicMax(box + 3, mxa); atomicMax(box + 4, mxb); atomicMax(box + 5, mxc);
        }
    } else {                                       // table full: straight to the global rows
        const SweepArgs* A = cold_args(kp);
        LocalSums Lc;
        Lc.n = L.n; Lc.sa = L.sa; Lc.sb = L.sb; Lc.sc = L.sc; Lc.saa = L.saa; Lc.sab = L.sab;
        Lc.sac = L.sac; Lc.sbb = L.sbb; Lc.sbc = L.sbc; Lc.scc = L.scc;
        uint32_t bx[6];
        bx[0] = mna; bx[1] = mnb; bx[2] = mnc; bx[3] = mxa; bx[4] = mxb; bx[5] = mxc;
        label_spill_global(A->sums, A->boxes, A->flags, A->max_label, label, &Lc, S.frame[0], S.frame[1], S.frame[2], bx);
    }
}

__device__ __forceinline__ uint32_t scan_pair_hash(uint32_t lo, uint32_t hi) {
    const uint32_t h = mad24_const(hi, 0x85EBCBu, mul24_const(lo, 0x9E3779u));     // two full-rate 24-bit multiplies, modulo 2^24
    return (h >> (24 - PSLOTS_LOG2)) & (PSLOTS - 1);
}

// (h, k: the pair's home slot and the key read from it by the caller, see scan_label_add)
template <typename LDS>
__device__ __forceinline__ void scan_pair_add(const SweepArgs* kp, LDS& S, uint32_t lo, uint32_t hi, uint32_t axis, uint32_t h,
                                              uint64_t k) {
    const uint64_t key = ((uint64_t)lo << 32) | hi;
    int slot = -1;
#pragma nounroll
    for (int probe = 0; probe < PPROBE; ++probe) {
        if (k == EMPTY_KEY) {
            k = atomicCAS((unsigned long long*)&S.pkeys[h], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (k == EMPTY_KEY) k = key;
        }
        if (k == key) { slot = (int)h; break; }
        h = (h + 1) & (PSLOTS - 1);
        k = S.pkeys[h];
    }
#ifdef TA_ABL_NOPCNT
    if (slot >= 0) { if (axis == 77u) S.pcnt[slot] = 0; }
#else
    if (slot >= 0) pcnt_add(S, (uint32_t)slot, axis);
#endif
    else {
        const SweepArgs* A = cold_args(kp);
        pair_spill_global(A->pairs, A->flags, lo, hi, axis, 1u);
    }
}

// the ten sums of one run [c0, c0 + n) of row (al, bl), tile-local: every term < 2^32, every factor < 2^24
template <bool MOM2>
__device__ __forceinline__ RunSums run_sums(const uint32_t c0, const uint32_t n, const uint32_t al, const uint32_t bl) {
    const uint32_t t1 = __umul24(n, n - 1u);                                 // n (n - 1), even
    const uint32_t nc0 = __umul24(n, c0);
    const uint32_t sc = nc0 + (t1 >> 1);                                     // sum c over the run
    const uint32_t na = __umul24(n, al), nb = __umul24(n, bl);
    RunSums L;
    L.n = n; L.sa = na; L.sb = nb; L.sc = sc;
    if (MOM2) {
        L.saa = __umul24(na, al); L.sab = __umul24(na, bl); L.sbb = __umul24(nb, bl);
        L.sac = __umul24(al, sc); L.sbc = __umul24(bl, sc);
        L.scc = __umul24(nc0, c0) + __umul24(c0, t1) + __umul24(t1 >> 1, 2u * n - 1u) / 3u;
    } else {
        L.saa = L.sab = L.sac = L.sbb = L.sbc = L.scc = 0;
    }
    return L;
}

// one run [s, k) of a row (tile-local a, b)
template <bool MOM2, typename LDS>
__device__ __forceinline__ void consume_scan_run(const SweepArgs* kp, LDS& S,
                                                 const bool EDGE, uint32_t label, uint32_t s, uint32_t code, uint32_t lh,
                                                 uint32_t lk) {
    const uint32_t c0 = s, k = code & 1023u, bl = (code >> 10) & 15u, al = (code >> 14) & 63u;
    const uint32_t n = k - c0;
    if (label >= LABEL_LIMIT) {
        // a real voxel the records cannot carry; INVALID_LABEL is the outside-the-volume filler of edge tiles only
        if (label != INVALID_LABEL || !EDGE) atomicOr(&cold_args(kp)->flags[FLAG_RANGE], 1u);
        return;
    }
    if (n == 0u) return;                  // the boundary at column 0: the run it closes belongs to the tile on the left
    const RunSums L = run_sums<MOM2>(c0, n, al, bl);
    scan_label_add<MOM2, LDS, RunSums>(kp, S, label, L, al, al, bl, bl, c0, k - 1u, lh, lk);
}

// Drain both buffers of a wave completely, 64 records per pass, every lane busy but in the last pass.
template <bool ADJ, bool MOM2, typename LDS, typename WLDS>
__device__ __forceinline__ void drain_buffers(const SweepArgs* kp, LDS& S, WLDS& W,
                                              const bool EDGE, int lane, uint32_t& fcount, uint32_t& rcount,
                                              const uint32_t lead_label, const bool may_keep) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (TA_ABLATE >= 1) { fcount = 0u; rcount = 0u; return; }
    if (ADJ) {
        // two passes of 64 faces per iteration: the home-slot reads of both are in flight together
        for (uint32_t i = 0; i < fcount; i += 128u) {
            const uint32_t idx0 = i + (uint32_t)lane, idx1 = idx0 + 64u;
            const uint2 rec0 = W.frec[idx0 < (uint32_t)FCAP ? idx0 : 0u], rec1 = W.frec[idx1 < (uint32_t)FCAP ? idx1 : 0u];
            const uint32_t v0 = rec0.x, pv0 = rec0.y & 0x3fffffffu, v1 = rec1.x, pv1 = rec1.y & 0x3fffffffu;
            // records that touch the outside-the-volume filler are dropped; a label the record words cannot
            // carry (>= LABEL_LIMIT) raises FLAG_RANGE through the run record of its own voxel
            const bool live0 = idx0 < fcount && v0 < LABEL_LIMIT && pv0 < LABEL_LIMIT;
            const bool live1 = idx1 < fcount && v1 < LABEL_LIMIT && pv1 < LABEL_LIMIT;
            const uint32_t lo0 = v0 < pv0 ? v0 : pv0, hi0 = v0 < pv0 ? pv0 : v0, lo1 = v1 < pv1 ? v1 : pv1, hi1 = v1 < pv1 ? pv1 : v1;
            const uint32_t h0 = scan_pair_hash(lo0, hi0), h1 = scan_pair_hash(lo1, hi1);
            const uint64_t k0 = S.pkeys[h0], k1 = S.pkeys[h1];
            if (live0) scan_pair_add(kp, S, lo0, hi0, rec0.y >> 30, h0, k0);
            if (live1) scan_pair_add(kp, S, lo1, hi1, rec1.y >> 30, h1, k1);
        }
        fcount = 0u;
    }
    // The voxel right of a boundary is read from the record that FOLLOWS it in its row, so a boundary record whose follower
    // is not in the buffer yet stays behind: the buffers are drained between rows, and then the last record closes a row --
    // except when a row too big for the buffers is placed a lane range at a time (place_in_pieces).
    // (`may_keep`: only there.  Everywhere else a last record without a follower is complete as it is: the boundary with the
    //  tile on the left in a row that has no closing record.)
    uint32_t keep = 0u;
    if (ADJ && may_keep && rcount) keep = (__builtin_amdgcn_readfirstlane((int)W.cqc[rcount]) & ROW_END) ? 0u : 1u;
    const uint32_t nrun = rcount - keep;
    for (uint32_t i = 0; i < nrun; i += 64u) {
        const uint32_t idx = i + (uint32_t)lane;
        const uint32_t slot = idx < (uint32_t)RCAP ? idx : 0u;
        const uint32_t prev = W.cqc[slot], code = W.cqc[slot + 1u], label = W.cql[slot + 1u];
        // (the voxel right of the boundary = the label of the run that starts there = the closing label of the next record;
        //  the record that closes a row has none.  A row WITHOUT a closing record -- one of the tile's leading one-label rows -- can
        //  still hold the boundary with the tile on the left, at column 0: the voxel right of it is the leading label.)
        const uint32_t ncode = ADJ ? W.cqc[slot + 2u] : 0u;
        const uint32_t v = !ADJ ? 0u : (code & ROW_END) ? INVALID_LABEL
                                     : (((code ^ ncode) & ROWID_MASK) == 0u && idx + 1u < rcount) ? W.cql[slot + 2u] : lead_label;
        // the home slots of both tables are read before either is worked on
        const uint32_t lo = label < v ? label : v, hi = label < v ? v : label;
        const uint32_t ph = ADJ ? scan_pair_hash(lo, hi) : 0u, lh = scan_label_hash(label);
        const uint64_t pk = ADJ ? S.pkeys[ph] : 0ull;
        const uint32_t lk = S.lkeys[lh];
        if (idx < nrun) {
            if (ADJ && v < LABEL_LIMIT && label < LABEL_LIMIT) scan_pair_add(kp, S, lo, hi, 2u, ph, pk);
            const uint32_t s = ((prev ^ code) & ROWID_MASK) == 0u ? (prev & 1023u) : 0u;
            consume_scan_run<MOM2, LDS>(kp, S, EDGE, label, s, code, lh, lk);
        }
    }
    if (rcount) {
        // carry: the record that follows (if it belongs to the same row) starts where the last consumed one ended; a record
        // that stays behind moves to the front
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) {
            const uint32_t carry = W.cqc[nrun];
            if (keep) { const uint32_t kl = W.cql[rcount], kc = W.cqc[rcount]; W.cql[1] = kl; W.cqc[1] = kc; }
            W.cqc[0] = carry;
        }
        rcount = keep;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}


// ---- the hot drains: several groups of 64 records in flight at once -------------------------------------------------
// drain_buffers above walks a buffer 64 (128) records at a time, and every group is a chain of dependent LDS round trips:
// record -> home slot -> (compare-and-swap, next slot ...) -> counters.  Measured (profiles/r04_pmc_ablations.md): the drain
// adds 17 % vector instructions and +0.34 ms to the sweep -- it waits.  Here a lane takes one record of EVERY group that is
// drained: one round trip reads all of them, one more reads all their home slots, and each further probe round -- a
// compare-and-swap where the slot read was EMPTY, the next slot where it held another key -- serves all lookups still
// open in lockstep.  Same tables, same probing order, same spill rules as scan_pair_add / scan_label_add; sums and minima
// commute, so the result is the same.
// These drains run at the TOP of a plane -- where a wave holds nothing but the plane before, and the next plane's loads
// are in flight anyway -- and take FULL groups only: what is left (at most 64 records) moves to the front of the buffer, so
// no pass runs with idle lanes and a run record's follower is always in the buffer (no record has to stay behind).  The
// drains inside a plane (a row that does not fit: drain_buffers) are the rare path.
//
// A lookup starts with a plain read of its home slot (most keys are there: same-address reads broadcast, they cost nothing).
// What is still open after that goes through PROBE ROUNDS, and a round is ONE compare-and-swap of the EMPTY key by the wanted
// one -- on the same slot when the read found it EMPTY, on the next slot when it held another key: what comes back is EMPTY
// (the slot is ours now), the key (it is there) or another key (on to the next slot).  No read-then-CAS pair, no branch on
// what was read: slot and state move by selects, the only divergent instruction is the masked compare-and-swap itself.
// (Per-lane branches are what made the probe loops of drain_buffers slow: ~100 scalar / branch instructions a round.)
template <typename LDS>
__device__ __forceinline__ uint64_t pair_probe_issue(LDS& S, const bool pend, uint32_t& slot, const uint64_t k, const uint64_t key) {
#if TA_PROBE_NORTN
    // a compare-and-swap that returns nothing where the slot was EMPTY, then a plain read of the slot (the same one, or the
    // next one where another key sits): LDS keeps a wave's operations in order, so the read sees what the swap left
    if (pend && k == EMPTY_KEY) (void)atomicCAS((unsigned long long*)&S.pkeys[slot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
    slot = (pend && k != EMPTY_KEY) ? ((slot + 1u) & (PSLOTS - 1)) : slot;
    uint64_t r = k;
    if (pend) r = S.pkeys[slot];
    return r;
#else
    slot = (pend && k != EMPTY_KEY) ? ((slot + 1u) & (PSLOTS - 1)) : slot;
    uint64_t r = k;
    if (pend) r = atomicCAS((unsigned long long*)&S.pkeys[slot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
    return r;
#endif
}
template <typename LDS>
__device__ __forceinline__ uint32_t label_probe_issue(LDS& S, const bool pend, uint32_t& slot, const uint32_t k, const uint32_t label) {
#if TA_PROBE_NORTN
    if (pend && k == INVALID_LABEL) (void)atomicCAS(&S.lkeys[slot], INVALID_LABEL, label);
    slot = (pend && k != INVALID_LABEL) ? ((slot + 1u) & (LSLOTS - 1)) : slot;
    uint32_t r = k;
    if (pend) r = S.lkeys[slot];
    return r;
#else
    slot = (pend && k != INVALID_LABEL) ? ((slot + 1u) & (LSLOTS - 1)) : slot;
    uint32_t r = k;
    if (pend) r = atomicCAS(&S.lkeys[slot], INVALID_LABEL, label);
    return r;
#endif
}

// TA_ABL_HOT (ablations of the hot drains, cumulative, results wrong by construction): 1 = nothing is added to the tables
// (no count / sum / box atomics), 2 = ... and no probe rounds, 3 = ... and no home-slot reads, 4 = ... and no record reads
// the LAST NF groups of 64 records of a face buffer that holds at least that many: face records carry no order, so the drain
// takes the top of the buffer and nothing has to move (round 4 took the front and moved what was left down: one LDS read and
// one write a drain, and a buffer that could not hold more than 64 records behind the groups)
template <int NF, typename LDS, typename WLDS>
__device__ __forceinline__ void drain_face_groups(const SweepArgs* kp, LDS& S, WLDS& W, int lane, uint32_t& fcount) {
    static_assert(NF == 1 || NF == 2, "one or two lookups per lane in lockstep");
    static_assert(FCAP >= 64 * NF, "the buffer holds the groups");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (TA_ABLATE >= 1) { fcount = 0u; return; }
#ifdef TA_ABL_NOHOT
    fcount = 0u; return;
#endif
    asm volatile("" : "+v"(lane));                // (keeps the addresses below out of the registers that live across the sweep)
    uint2 rec[NF];
#pragma unroll
    for (int g = 0; g < NF; ++g) {
        if (TA_ABL_HOT < 4) rec[g] = W.frec[fcount - 64u * (uint32_t)NF + (uint32_t)(lane + 64 * g)];
        else { rec[g].x = (uint32_t)lane * 2654435761u; rec[g].y = rec[g].x >> 7; asm volatile("" : "+v"(rec[g].x), "+v"(rec[g].y)); }
    }
    const uint32_t rem = fcount - 64u * NF;                               // what stays, where it is
    uint32_t lo[NF], hi[NF], slot[NF];
    uint64_t k[NF], key[NF];
    bool live[NF], pend[NF];
#pragma unroll
    for (int g = 0; g < NF; ++g) {
        const uint32_t v = rec[g].x, pv = rec[g].y & 0x3fffffffu;
        // records that touch the outside-the-volume filler are dropped; a label the record words cannot carry (>= LABEL_LIMIT)
        // raises FLAG_RANGE through the run record of its own voxel
        lo[g] = v < pv ? v : pv; hi[g] = v < pv ? pv : v;
        key[g] = ((uint64_t)lo[g] << 32) | hi[g];
        live[g] = v < LABEL_LIMIT && pv < LABEL_LIMIT;
        slot[g] = scan_pair_hash(lo[g], hi[g]);
    }
    // the home slot AND the one behind it (reads are cheap, a probe round is not: profiles/r04_ablation_ladder.txt): a key that
    // another one has pushed off its home slot sits one slot further nine times in ten
    uint64_t k1[NF];
#pragma unroll
    for (int g = 0; g < NF; ++g) {
        if (TA_ABL_HOT < 3) { k[g] = S.pkeys[slot[g]]; k1[g] = S.pkeys[(slot[g] + 1u) & (PSLOTS - 1)]; }
        else { k[g] = key[g]; k1[g] = key[g]; asm volatile("" :: "v"(slot[g])); }
    }
#pragma unroll
    for (int g = 0; g < NF; ++g) {
        const bool home = k[g] == key[g] || k[g] == EMPTY_KEY;            // (it is there, or it goes there)
        slot[g] = home ? slot[g] : ((slot[g] + 1u) & (PSLOTS - 1));
        k[g] = home ? k[g] : k1[g];
        pend[g] = live[g] && k[g] != key[g];
    }
    bool spill = false;
#if !defined(TA_ABL_NOLOOP) && TA_ABL_HOT < 2
#pragma nounroll
    for (uint32_t round = 1u; __builtin_amdgcn_ballot_w64(pend[0] | pend[NF - 1]); ++round) {
        if (round >= (uint32_t)PPROBE) { spill = true; break; }
        const uint64_t r0 = pair_probe_issue(S, pend[0], slot[0], k[0], key[0]);
        // (a compare-and-swap that gave EMPTY back has put the key there; a plain read that gives EMPTY has found a free slot)
        if constexpr (NF == 2) {
            const uint64_t r1 = pair_probe_issue(S, pend[1], slot[1], k[1], key[1]);
            k[1] = (!TA_PROBE_NORTN && r1 == EMPTY_KEY) ? key[1] : r1;
            pend[1] = pend[1] && k[1] != key[1];
        }
        k[0] = (!TA_PROBE_NORTN && r0 == EMPTY_KEY) ? key[0] : r0;
        pend[0] = pend[0] && k[0] != key[0];
    }
#endif
#pragma unroll
    for (int g = 0; g < NF; ++g) {
        const bool ok = live[g] && !pend[g];
#ifdef TA_ABL_SHARE1
        if (TA_ABL_HOT < 1) { if (ok) pcnt_add(S, (slot[g] + (uint32_t)lane * 7u) & (PSLOTS - 1), rec[g].y >> 30); }
#else
        if (TA_ABL_HOT < 1) { if (ok) pcnt_add(S, slot[g], rec[g].y >> 30); }
#endif
        if (TA_ABL_HOT >= 1) asm volatile("" :: "v"(slot[g]), "s"(__builtin_amdgcn_ballot_w64(ok)));
        if (spill) {          // (wave-uniform: the probe limit was reached with lookups still open)
            if (pend[g]) { const SweepArgs* A = cold_args(kp); pair_spill_global(A->pairs, A->flags, lo[g], hi[g], rec[g].y >> 30, 1u); }
        }
    }
    fcount = rem;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// the first 64 run records of a buffer that holds MORE than 64 (the follower of record 63 is there)
template <bool MOM2, typename LDS, typename WLDS>
__device__ __forceinline__ void drain_run_group(const SweepArgs* kp, LDS& S, WLDS& W, const bool EDGE, int lane, uint32_t& rcount,
                                                const uint32_t lead_label) {
    static_assert(RCAP <= 128, "what is left after one group moves to the front one record per lane");
    constexpr int NW = MOM2 ? 4 : 2;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (TA_ABLATE >= 1) { rcount = 0u; return; }
#ifdef TA_ABL_NOHOT
    rcount = 0u; if (lane == 0) W.cqc[0] = NO_ROW; return;
#endif
    asm volatile("" : "+v"(lane));
    uint32_t prev, code, ncode, label, nlabel;
    if (TA_ABL_HOT < 4) {
        prev = W.cqc[lane]; code = W.cqc[lane + 1]; ncode = W.cqc[lane + 2];
        label = W.cql[lane + 1]; nlabel = W.cql[lane + 2];
    } else {
        prev = (uint32_t)lane * 3u; code = (uint32_t)lane * 3u + 2u; ncode = code + 3u; label = (uint32_t)lane >> 3; nlabel = label + 1u;
        asm volatile("" : "+v"(prev), "+v"(code), "+v"(ncode), "+v"(label), "+v"(nlabel));
    }
    // what stays: records 64 .. rcount - 1 move to the front, the code of record 63 becomes the carry
    const uint32_t rem = rcount - 64u;
    const uint32_t tsrc = (uint32_t)lane < rem ? 65u + (uint32_t)lane : 0u;
    const uint32_t tail_l = W.cql[tsrc], tail_c = W.cqc[tsrc], carry = W.cqc[64];
    // the voxel right of the boundary: the closing label of the row's next record (it is in the buffer); none behind the
    // record that closes the row; the leading label where the row has no closing record (see drain_buffers)
    const uint32_t v = (code & ROW_END) ? INVALID_LABEL : (((code ^ ncode) & ROWID_MASK) == 0u) ? nlabel : lead_label;
    const uint32_t plo = label < v ? label : v, phi = label < v ? v : label;
    const uint64_t pkey = ((uint64_t)plo << 32) | phi;
    const bool plive = v < LABEL_LIMIT && label < LABEL_LIMIT;
    uint32_t pslot = scan_pair_hash(plo, phi);
    const uint32_t c0 = ((prev ^ code) & ROWID_MASK) == 0u ? (prev & 1023u) : 0u, kk = code & 1023u;
    // a real voxel the records cannot carry; INVALID_LABEL is the outside-the-volume filler of edge tiles only
    const bool range = label >= LABEL_LIMIT && (label != INVALID_LABEL || !EDGE);
    // (an empty run: the boundary at column 0 -- the run it closes belongs to the tile on the left)
    const bool llive = label < LABEL_LIMIT && kk != c0;
    uint32_t lslot = scan_label_hash(label);
    uint64_t pk = pkey, pk1 = pkey;
    uint32_t lk = label, lk1 = label;
    if (TA_ABL_HOT < 3) {             // (the home slots and the ones behind them: see drain_face_groups)
        pk = S.pkeys[pslot]; lk = S.lkeys[lslot];
        pk1 = S.pkeys[(pslot + 1u) & (PSLOTS - 1)]; lk1 = S.lkeys[(lslot + 1u) & (LSLOTS - 1)];
    } else asm volatile("" :: "v"(pslot), "v"(lslot));
    if ((uint32_t)lane < rem) { W.cql[1 + lane] = tail_l; W.cqc[1 + lane] = tail_c; }
    if (lane == 0) W.cqc[0] = carry;
    if (__builtin_amdgcn_ballot_w64(range)) { if (range) atomicOr(&cold_args(kp)->flags[FLAG_RANGE], 1u); }
    // the pair and the label lookup in lockstep
    {
        const bool phome = pk == pkey || pk == EMPTY_KEY, lhome = lk == label || lk == INVALID_LABEL;
        pslot = phome ? pslot : ((pslot + 1u) & (PSLOTS - 1)); pk = phome ? pk : pk1;
        lslot = lhome ? lslot : ((lslot + 1u) & (LSLOTS - 1)); lk = lhome ? lk : lk1;
    }
    bool ppend = plive && pk != pkey, lpend = llive && lk != label, spill = false;
#if !defined(TA_ABL_NOLOOP) && TA_ABL_HOT < 2
#pragma nounroll
    for (uint32_t round = 1u; __builtin_amdgcn_ballot_w64(ppend | lpend); ++round) {
        if (round >= (uint32_t)LPROBE) { spill = true; break; }          // (LPROBE <= PPROBE: both give up together)
        const uint64_t rp = pair_probe_issue(S, ppend, pslot, pk, pkey);
        const uint32_t rl = label_probe_issue(S, lpend, lslot, lk, label);
        pk = (!TA_PROBE_NORTN && rp == EMPTY_KEY) ? pkey : rp; lk = (!TA_PROBE_NORTN && rl == INVALID_LABEL) ? label : rl;
        ppend = ppend && pk != pkey; lpend = lpend && lk != label;
    }
#endif
    // -- what the record carries, into the slots found.  The box is read before the sums are added (LDS operations of a
    //    wave complete in order: a read behind six atomics would wait for them)
    const uint32_t* boxr = &S.lbox[lslot * 8];
    uint4 cur = {0u, 0u, 0u, 0xffffu};
    uint2 cur2 = {0xffffu, 0xffffu};
    if (TA_ABL_HOT < 1) {
#ifndef TA_ABL_NOBOXHOT
        cur = *reinterpret_cast<const uint4*>(boxr);              // min a,b,c | max a
        cur2 = *reinterpret_cast<const uint2*>(boxr + 4);         // max b,c
#endif
        if (plive && !ppend) pcnt_add(S, pslot, 2u);
    } else asm volatile("" :: "v"(pslot), "s"(__builtin_amdgcn_ballot_w64(plive && !ppend)));
    const uint32_t bl = (code >> 10) & 15u, al = (code >> 14) & 63u;
    const RunSums L = run_sums<MOM2>(c0, kk - c0, al, bl);
    if (TA_ABL_HOT >= 1) asm volatile("" :: "v"(L.n), "v"(L.sa), "v"(L.sb), "v"(L.sc), "v"(L.saa), "v"(L.sab), "v"(L.sac), "v"(L.sbb), "v"(L.sbc), "v"(L.scc), "v"(lslot), "s"(__builtin_amdgcn_ballot_w64(llive && !lpend)));
    if (TA_ABL_HOT < 1 && llive && !lpend) {
        // (records of one label in a group come from consecutive rows: rows of different parity take different replicas, which
        //  halves the lanes that share an address in each of these adds)
        const uint32_t rep = LDS::REP > 1 ? (bl & (uint32_t)(LDS::REP - 1)) : 0u;
#ifdef TA_ABL_SHARE1       // (ablation: every lane its own row -- no two lanes of an atomic share an address)
        unsigned long long* row = (unsigned long long*)&S.lsum[((lslot + (uint32_t)lane) & (LSLOTS - 1)) * NW * LDS::REP];
#else
        unsigned long long* row = (unsigned long long*)&S.lsum[(lslot * LDS::REP + rep) * NW];
#endif
        uint64_t w[4];
        LDS::Pack::template pack<MOM2>(L, w);
#pragma unroll
        for (int k = 0; k < NW; ++k) atomicAdd(row + k, (unsigned long long)w[k]);
        // bounding box: touched only when this run extends it (almost every run lies inside the box its label already has)
#ifdef TA_ABL_NOBOXHOT
        const bool grows = false;
#else
        const bool grows = (al < cur.x) | (bl < cur.y) | (c0 < cur.z) | (al > cur.w) | (bl > cur2.x) | (kk - 1u > cur2.y);
#endif
        if (grows) {
            uint32_t* box = &S.lbox[lslot * 8];
            atomicMin(box + 0, al); atomicMin(box + 1, bl); atomicMin(box + 2, c0);
            atomicMax(box + 3, al); atomicMax(box + 4, bl); atomicMax(box + 5, kk - 1u);
        }
    }
    if (spill) {              // (wave-uniform: the probe limit was reached with lookups still open)
        if (ppend) { const SweepArgs* A = cold_args(kp); pair_spill_global(A->pairs, A->flags, plo, phi, 2u, 1u); }
        if (lpend) {
            const SweepArgs* A = cold_args(kp);
            LocalSums Lc;
            Lc.n = L.n; Lc.sa = L.sa; Lc.sb = L.sb; Lc.sc = L.sc; Lc.saa = L.saa; Lc.sab = L.sab;
            Lc.sac = L.sac; Lc.sbb = L.sbb; Lc.sbc = L.sbc; Lc.scc = L.scc;
            uint32_t bx[6];
            bx[0] = al; bx[1] = bl; bx[2] = c0; bx[3] = al; bx[4] = bl; bx[5] = kk - 1u;
            label_spill_global(A->sums, A->boxes, A->flags, A->max_label, label, &Lc, S.frame[0], S.frame[1], S.frame[2], bx);
        }
    }
    rcount = rem;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// strip load of this file: like load_strip, but the scalar path of edge tiles also reports a REAL voxel that
// equals the outside-the-volume filler (0xFFFFFFFF in a uint32 volume: above any max_label)
template <typename T, int VPL>
__device__ __forceinline__ void scan_load_strip(const bool EDGE, const T* row_c0, bool row_ok, uint32_t lane_off,
                                                int64_t c, int64_t n2, uint32_t (&dst)[VPL], bool& bad) {
    if (!EDGE) {
        const uint4 x = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(row_c0) + lane_off);
        if (sizeof(T) == 4) {
            dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
        } else {
            dst[0] = x.x & 0xffffu; dst[1] = x.x >> 16; dst[2] = x.y & 0xffffu; dst[3] = x.y >> 16;
            if (VPL > 4) {           // (VPL = 4 of a uint16 volume: the lane's strip is the first half of what it loaded)
                dst[4 % VPL] = x.z & 0xffffu; dst[5 % VPL] = x.z >> 16;
                dst[6 % VPL] = x.w & 0xffffu; dst[7 % VPL] = x.w >> 16;
            }
        }
    } else {
        const T* lane_p = reinterpret_cast<const T*>(reinterpret_cast<const char*>(row_c0) + lane_off);
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const bool ok = row_ok && c + j < n2;
            dst[j] = ok ? (uint32_t)lane_p[j] : INVALID_LABEL;
            if (sizeof(T) == 4) bad |= ok && dst[j] == INVALID_LABEL;
        }
    }
}

// ---- interior tiles: the plane loads are issued by hand --------------------------------------------
// The compiler's vmcnt bookkeeping gives up at the loop's control-flow joins and waits for vmcnt(0) at the top
// of every plane -- right after the next plane's loads were issued, so the "prefetch" never overlaps anything
// and every wave eats a full HBM round trip per plane.  Here the loads are inline asm the compiler does not
// count: raw 16-byte strips land in `nraw` one plane ahead and ONE s_waitcnt vmcnt(0) sits just before they are
// unpacked into the working registers, a whole plane of compute after they were issued.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The plane in flight lives in registers the compiler never allocates: the kernels are compiled with
// amdgpu_num_vgpr(BASE), and v[BASE ..] are named explicitly in Pin<BASE>: the first quads = the rows of the
// wave tile, the next quad = the row above it, then one register for lanes 0..RB-1: the voxel left of each row.
// (Loading into ordinary asm outputs does not work: the register allocator copies a loop-carried output at the
// back edge, i.e. reads it while the load is still in flight; accumulation registers make the compiler split the
// unified file in halves.)
#ifndef TA_PIN_ADJ
#define TA_PIN_ADJ 104        // kernels with adjacency: 104 compiler-allocated VGPRs + 21 pinned = 125: four waves per SIMD
#endif
#define TA_PIN_MOM 76         // kernels without (rows only: no row above, no voxel to the left): 76 + 16 = 92: five waves
// amdgpu_num_vgpr is a request the allocator overshoots when it would have to spill: ask for less than the first
// pinned register; the build checks the generated code (tissue_analysis_amd/build.py).
#ifndef TA_CAP_ADJ
#define TA_CAP_ADJ 100
#endif
// uint32 volumes with adjacency run TWO rows per wave (RB = 2: half the plane state of the four-row tiles): 82 compiler-allocated
// VGPRs + 13 pinned (two rows, the row above, the voxel to the left) = 95: FIVE waves per SIMD
#define TA_PIN_ADJ2 82
#define TA_CAP_ADJ2 78
// TA_PLANES_IN_FLIGHT = 2 (experiment): a second landing zone behind the first, 82 + 27 = 109 registers, FOUR waves per SIMD
#define TA_PIN_ADJ2B 96
#define TA_CAP_MOM 72
// the PADDED kernels (partial tiles of a volume whose rows are 16-byte aligned: interior-style loads, edge-style
// semantics) carry more state: with adjacency 116 + 21 = 137 (three waves per SIMD), without 80 + 16 = 96 (five, like the
// interior kernel)
// two rows of EIGHT uint32 voxels a lane (TA_U32_SHAPE = 1): 25 pinned registers (an even base: 16-byte loads want even register pairs), 102 + 25 = 127: four waves per SIMD
#define TA_PIN_ADJ8 102
#define TA_CAP_ADJ8 98
#define TA_PIN_ADJ8_PAD 116
#define TA_CAP_ADJ8_PAD 112
#define TA_PIN_ADJ_PAD 120
#define TA_PIN_MOM_PAD 80
#define TA_CAP_ADJ_PAD 116
#define TA_CAP_MOM_PAD 76
// ... and with two rows per wave (uint32 volumes with adjacency) the landing zone is 13 registers: 112 + 13 = 125, FOUR waves
// per SIMD for the partial tiles (1000^3 is a quarter partial tiles; real volumes are rarely multiples of 8 x 256)
#define TA_PIN_ADJ2_PAD 112
#define TA_CAP_ADJ2_PAD 108
template <int BASE> struct Pin;
#include "ta_pin_tables.inc"        // Pin<104>, Pin<120>, Pin<76>, Pin<80>, Pin<82>, Pin<112>: generated by scripts/gen_pin_tables.py
template <typename T, int VPL>
__device__ __forceinline__ void unpack_strip(const u32x4& x, uint32_t (&dst)[VPL]) {
    if (sizeof(T) == 4) {
        dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
    } else {
        dst[0] = x.x & 0xffffu; dst[1] = x.x >> 16; dst[2] = x.y & 0xffffu; dst[3] = x.y >> 16;
        if (VPL > 4) {               // (VPL = 4 of a uint16 volume: 8-byte strips, loaded as the first half of 16 bytes)
            dst[4 % VPL] = x.z & 0xffffu; dst[5 % VPL] = x.z >> 16;
            dst[6 % VPL] = x.w & 0xffffu; dst[7 % VPL] = x.w >> 16;
        }
    }
}

// EDGE: the tile may hold positions outside the volume (they carry the filler INVALID_LABEL) and may lack a row above / a
// column to the left.  PINB: 0 = plain guarded loads (volumes whose rows are not 16-byte aligned), else the first of the
// hand-pinned registers the plane in flight lands in.  EDGE with PINB != 0 is the PADDED variant: interior-style loads from
// clamped addresses, the positions outside the volume overwritten with the filler when the plane lands.
// PINB2 != 0 (experiment, TA_PLANES_IN_FLIGHT = 2): a SECOND landing zone -- two planes in flight per wave, the plane loop
// unrolled by two, `s_waitcnt vmcnt(4)` where the next plane's four loads stay in flight.
template <typename T, int VPL, int RB, bool ADJ, bool MOM2, bool EDGE, int PINB, int PINB2, typename LDS>
__device__ __forceinline__ void wave_scan(const SweepArgs& A, const SweepArgs* kp, LDS& S, const int lane, const int w,
                                          const uint32_t c_tile0, const uint32_t b_tile0,
                                          const int32_t p_lo, const int32_t p_hi) {
    constexpr int TC = 64 * VPL;
    static_assert(TC <= 512, "the run code holds the end column in 10 bits");
    static_assert(FCAP >= RB * VPL && RCAP >= VPL + 1, "the records of one lane must fit a buffer");
    auto& W = S.wave[w];
    // the hot drain sites take every record of a buffer at once (drain_faces_all / drain_runs_all); the rare ones -- a row too
    // big for the buffers placed in pieces, the end of the tile -- keep the compact group-by-group drain_buffers
// TA_FDRAIN1: tiles of eight voxels a lane put twice the records of a plane step into the same buffers; with 64 .. 127 faces
// left after a step, the next one's often do not fit and the whole buffer goes through the in-plane drain (29 % of C4's face
// records, 43 % of the tissue-filled volume's, against 13 / 14 % with four voxels a lane).  With one group of 64 leaving at
// the top of the plane those shares fall to 7 / 16 % -- and the sweep takes the same time (C4 1.01 / 1.02 vs 1.01 / 1.02 ms,
// filled 1.38 - 1.44 vs 1.39 - 1.46, C5 6.79 / 6.83 vs 6.78 / 6.86): where a record is drained does not matter.  On since
// round 5 (with the predicated stores: C4 0.972 -> 0.965, filled 1.392 -> 1.367 / 1.390 ms in one call -- within the noise, and
// the in-plane drain with its per-lane probe loops becomes the rare path it is meant to be).
    constexpr bool DRAIN_ALL = ADJ && TA_DRAIN_ALL && FCAP % 64 == 0 && RCAP % 64 == 0 && (TA_DRAIN_ALL_U16 || VPL == 4) && RB == 2 && !EDGE;

    const T* vol = reinterpret_cast<const T*>(A.vol);
    const int64_t n1 = A.n1, n2 = A.n2, plane = n1 * n2;
    const int64_t b_wave0 = (int64_t)b_tile0 + (int64_t)w * RB;
    const int64_t c0g = (int64_t)c_tile0 + (int64_t)lane * VPL;
#ifdef TA_ABL_NOHALO                                        // 1: a workgroup's first wave reads no row above (the row another tile owns); 2: no wave does
    const bool has_up = ADJ && b_wave0 > 0 && (TA_ABL_NOHALO == 1 ? w != 0 : false);
#else
    const bool has_up = ADJ && b_wave0 > 0;
#endif
    const bool has_left = ADJ && c_tile0 > 0;
    const bool has_prev = ADJ && p_lo > 0;
    const uint32_t lane_c = (uint32_t)lane * VPL;
    constexpr bool PAD = EDGE && PINB != 0;
    // PADDED: a lane whose strip lies past the end of the row loads the tile's first strip instead (a strip that straddles
    // the end is loaded where it is: the few bytes past the row are the next row's, or the slack behind the library's own
    // volume buffer); a row past the last one loads the wave's first row (the last row of the volume when the whole wave
    // is outside).  `nin`: how many voxels of the lane's strip are inside the row.
    const int64_t nin64 = n2 - c0g;
    const uint32_t nin = !PAD ? (uint32_t)VPL : (nin64 <= 0 ? 0u : (nin64 >= VPL ? (uint32_t)VPL : (uint32_t)nin64));
    const uint32_t lane_off = (nin ? lane_c : 0u) * (uint32_t)sizeof(T);
    const int64_t b_base = PAD && b_wave0 >= n1 ? n1 - 1 : b_wave0;
    bool row_in[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) row_in[r] = !PAD || b_wave0 + r < n1;

    // The working set of a lane: the RB x VPL voxels of the CURRENT plane, the strip of the row above, the voxel to
    // the left.  The previous plane is never kept: its faces with the current one are taken at the moment the new
    // plane lands, while both are in registers (that is what leaves room for a fifth wave per SIMD).
    uint32_t cur[RB][VPL], up[VPL];
    uint32_t leftv = INVALID_LABEL;                        // lane r: the voxel left of row r of the tile
    static_assert(RB == 2 || RB == 4, "the pinned-register layout is written out for 2 and 4 rows");
    bool bad = false;

    // -- edge tiles: plain loads with bounds, the compiler waits as it sees fit
    auto load_rows = [&](int64_t p, uint32_t (&d)[RB][VPL]) {
        const T* pbase = vol + p * plane;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int64_t b = b_wave0 + r;
            const bool row_ok = b < n1;
            const T* row = pbase + (row_ok ? b : 0) * n2 + c_tile0;
            scan_load_strip<T, VPL>(true, row, row_ok, lane_off, c0g, n2, d[r], bad);
        }
    };
    auto load_halo = [&](int64_t p, uint32_t (&dup)[VPL], uint32_t& dl) {
        const T* pbase = vol + p * plane;
        if (has_left) {
            const int64_t b = b_wave0 + lane;
            dl = (lane < RB && b < n1) ? (uint32_t)pbase[b * n2 + c_tile0 - 1] : INVALID_LABEL;
        }
        if (has_up) {
            const bool row_ok = (b_wave0 - 1) < n1;
            const T* row = pbase + (row_ok ? (b_wave0 - 1) : 0) * n2 + c_tile0;
            scan_load_strip<T, VPL>(true, row, row_ok, lane_off, c0g, n2, dup, bad);
        }
    };
    // -- interior tiles: every load is unconditional, RB + 2 loads per plane.  A tile without a row above / a column
    //    to the left reads its own first row / column instead: those compares can never fire, so the hot loop does
    //    not test has_up / has_left at all.  One running pointer walks the planes.
    const uint32_t rowbytes = (uint32_t)(n2 * (int64_t)sizeof(T));
    const uint32_t left_off = (uint32_t)(PAD && b_wave0 + (lane & (RB - 1)) >= n1 ? 0 : (lane & (RB - 1))) * rowbytes;    // (the VGPR offset of a load is unsigned)
    const int64_t plane_bytes = plane * (int64_t)sizeof(T);
    const char* next_row0 = reinterpret_cast<const char*>(vol + (int64_t)(has_prev ? p_lo - 1 : p_lo) * plane + b_base * n2 + c_tile0);
    auto issue_plane_to = [&](auto zone) {                // issues the plane at `next_row0` into a landing zone and steps it
        const char* row0 = next_row0;
#ifndef TA_ABL_L2
        next_row0 += plane_bytes;
#endif
        using P = Pin<decltype(zone)::value>;
        if constexpr (sizeof(T) == 4 && VPL == 8) {      // 32-byte strips: two quads a row, two rows, two quads the row above
            static_assert(RB == 2 && ADJ, "eight uint32 voxels a lane: two rows a wave, the 25-register landing zone");
            const char* row1 = row_in[1] ? row0 + rowbytes : row0;
            P::template issue_strip<0>(lane_off, row0);
            P::template issue_strip<1>(lane_off, row0 + 16);
            P::template issue_strip<2>(lane_off, row1);
            P::template issue_strip<3>(lane_off, row1 + 16);
            const bool up_there = has_up && (!PAD || b_wave0 < n1);
            const char* upr = up_there ? row0 - rowbytes : row0;
            P::template issue_strip<4>(lane_off, upr);
            P::template issue_strip<5>(lane_off, upr + 16);
            P::template issue_voxel<T, RB>(left_off, has_left ? row0 - sizeof(T) : row0);
        } else if constexpr (sizeof(T) == 2 && VPL == 4) {      // 8-byte strips: the first half of each landing quad
            static_assert(RB == 2, "four uint16 voxels a lane: the two-row tiles");
            P::template issue_half<0>(lane_off, row0);
            P::template issue_half<1>(lane_off, row_in[1] ? row0 + rowbytes : row0);
            if (ADJ) {
                const bool up_there = has_up && (!PAD || b_wave0 < n1);
                P::template issue_half<2>(lane_off, up_there ? row0 - rowbytes : row0);
                P::template issue_voxel<T, RB>(left_off, has_left ? row0 - sizeof(T) : row0);
            }
        } else {
        P::template issue_strip<0>(lane_off, row0);
        P::template issue_strip<1>(lane_off, row_in[1] ? row0 + rowbytes : row0);
        if (RB > 2) {
            P::template issue_strip<2>(lane_off, row_in[RB > 2 ? 2 : 0] ? row0 + 2 * (int64_t)rowbytes : row0);
            P::template issue_strip<3>(lane_off, row_in[RB > 3 ? 3 : 0] ? row0 + 3 * (int64_t)rowbytes : row0);
        }
        if (ADJ) {       // (without adjacency nobody looks at the row above or the voxel to the left)
            // (PADDED, the whole wave past the last row: row0 is the last row of the volume; the row "above" it is not asked for)
            const bool up_there = has_up && (!PAD || b_wave0 < n1);
            if (RB > 2) P::template issue_strip<4>(lane_off, up_there ? row0 - rowbytes : row0);
            else        P::template issue_strip<2>(lane_off, up_there ? row0 - rowbytes : row0);
            P::template issue_voxel<T, RB>(left_off, has_left ? row0 - sizeof(T) : row0);
        }
        }
    };
    auto issue_plane = [&]() { issue_plane_to(std::integral_constant<int, (PINB ? PINB : TA_PIN_ADJ)>{}); };

    // PADDED: what landed for positions outside the volume is overwritten with the filler; a REAL voxel equal to the
    // filler (0xFFFFFFFF in a uint32 volume: above any max_label) is reported like the plain edge loads report it
    auto pad_plane = [&](uint32_t (&nw)[RB][VPL], uint32_t (&nup)[VPL], uint32_t& nl) {
        if constexpr (PAD) {
            if (sizeof(T) == 4) {
                uint32_t mx = 0u;
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int j = 0; j < VPL; ++j) mx = max(mx, (row_in[r] && (uint32_t)j < nin) ? nw[r][j] : 0u);
                bad = bad || mx == INVALID_LABEL;
            }
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int j = 0; j < VPL; ++j) nw[r][j] = (row_in[r] && (uint32_t)j < nin) ? nw[r][j] : INVALID_LABEL;
#pragma unroll
            for (int j = 0; j < VPL; ++j) nup[j] = (has_up && (uint32_t)j < nin && b_wave0 < n1) ? nup[j] : INVALID_LABEL;
            nl = (has_left && lane < RB && b_wave0 + lane < n1) ? nl : INVALID_LABEL;
        }
    };

    uint32_t fcount = 0u, rcount = 0u;                    // records in the buffers (wave-uniform)
#ifdef TA_STAMPS
    uint64_t tk_cmp = 0, tk_emit = 0, tk_drain = 0, tk_adv = 0, tk_land = 0, tk_evrows = 0, tk_drains = 0;
#endif
    if (lane == 0) W.cqc[0] = NO_ROW;
    // LDS byte offsets of the wave's buffers (the low half of a flat LDS address is the LDS offset)
    const uint32_t fbase = (uint32_t)(uintptr_t)&W.frec[0];
    const uint32_t rbase = (uint32_t)(uintptr_t)&W.cql[1];
    constexpr uint32_t RSTRIDE = (RCAP + 2) * 4u;         // bytes between the two run arrays
    [[maybe_unused]] const uint32_t ftrash = (uint32_t)(uintptr_t)&W.frec[FTRASH];   // where the stores of compares that did not fire go (TA_MASKED_STORES: nowhere)
    [[maybe_unused]] const uint32_t rtrash = (uint32_t)(uintptr_t)&W.cql[RTRASH];

    // Leading rows of the tile that are one label (the label of its first voxel) from end to end are not records:
    // they are counted and added in closed form at the end -- that is the whole cost of background.
    uint32_t first_label = INVALID_LABEL;
    bool leading = true;
    uint32_t nlead = 0u;
    auto drain = [&](const bool may_keep) {
#ifdef TA_ABL_NOSLOW      // (ablation, results wrong by construction: the in-plane drains and the end of the tile consume nothing)
        fcount = 0u; rcount = 0u; if (lane == 0) W.cqc[0] = NO_ROW; return;
#endif
#ifdef TA_STAMPS
        const uint64_t td0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef TA_RECCOUNT
        if (lane == 0) { atomicAdd(&cold_args(kp)->flags[8], fcount); atomicAdd(&cold_args(kp)->flags[9], rcount); atomicAdd(&cold_args(kp)->flags[10], 1u); }
#endif
        drain_buffers<ADJ, MOM2, LDS>(kp, S, W, EDGE, lane, fcount, rcount, first_label, may_keep);
#ifdef TA_STAMPS
        tk_drain += __builtin_amdgcn_s_memtime() - td0; tk_drains += 1;
#endif
    };
    // A set of records too big even for empty buffers (a flat wall between two planes: up to RB * VPL faces per lane; noise)
    // goes a lane range at a time: halve the range until its records fit, place them, go on with the rest; the buffers are
    // drained whenever the next range does not fit behind what they hold (a boundary record whose follower has not been
    // placed yet stays behind in a drain, see drain_buffers).
    auto place_in_pieces = [&](const uint32_t cnt, auto&& emit) {
        uint32_t lo = 0u, hi = 64u;
        for (;;) {
            const bool insel = ((uint32_t)lane - lo) < (hi - lo);
            const uint32_t mine = insel ? cnt : 0u;
            const uint32_t incl = wave_scan_add(mine);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t rowf = tot & 0xffffu, rowr = tot >> 16;
            if (fcount + rowf > (uint32_t)FCAP || rcount + rowr > (uint32_t)RCAP) {
                if (fcount != 0u || rcount > 1u) drain(true);              // make room first (one record may stay behind)
                else hi = lo + ((hi - lo) >> 1);                           // too big even then: half the lanes
                continue;
            }
            const uint32_t excl = incl - mine;
            if (insel) emit(fbase + ((fcount + (excl & 0xffffu)) << 3), rbase + ((rcount + (excl >> 16)) << 2), std::false_type{});
            fcount += rowf; rcount += rowr;
            if (hi >= 64u) break;
            lo = hi; hi = 64u;
        }
    };
    // One packed add-scan over the lanes gives every lane the offset of its first record (faces in the low half of `cnt`,
    // runs in the high half); the totals say whether the records fit behind what the buffers hold -- if not, the buffers are
    // drained first.  `emit(offf, offr)` stores the lane's records at its offsets.  The common path is straight-line: scan,
    // two scalar compares, the stores, one threshold test.
    auto place_records = [&](const uint32_t cnt, auto&& emit) {
        const uint32_t incl = wave_scan_add(cnt);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t rowf = tot & 0xffffu, rowr = tot >> 16;
        if (TA_ABLATE >= 3) return;
        // (one drain site, one scalar test: the buffers are drained on the way IN, when these records do not fit behind what
        //  they hold; draining earlier, at a fill threshold, measured the same from 64 to 190 records)
        const int32_t over = max((int32_t)(fcount + rowf) - (int32_t)FCAP, (int32_t)(rcount + rowr) - (int32_t)RCAP);
        if (over > 0) {
            drain(false);
            if (rowf > (uint32_t)FCAP || rcount + rowr > (uint32_t)RCAP) { place_in_pieces(cnt, emit); return; }
        }
        const uint32_t excl = incl - cnt;
        const uint32_t offf = fbase + ((fcount + (excl & 0xffffu)) << 3);     // LDS address of the lane's next face record
        const uint32_t offr = rbase + ((rcount + (excl >> 16)) << 2);         // ... and of its next run record (first array)
        fcount += rowf; rcount += rowr;
        if (TA_ABLATE < 2) emit(offf, offr, std::true_type{});      // (straight-line code: every lane is live)
    };
    // A new plane has landed: its faces with the plane before it (axis 0) are counted and stored TOGETHER with the events of
    // its first row -- one scan of the lanes' counts instead of two (`old` = the plane before, kept until that row is done).
    uint32_t old[RB][VPL];
    auto count_plane_faces = [&]() {
        uint32_t cf = 0u;
#ifndef TA_ABL_NOFACE0
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < VPL; ++j) cf += (cur[r][j] != old[r][j]) ? 1u : 0u;
#endif
        return cf;
    };
    auto emit_plane_faces = [&](uint32_t& offf, const uint64_t live, auto full) {
#ifndef TA_ABL_NOFACE0
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
#if TA_MASKED_STORES & 1
                if constexpr (VPL % 4 == 0) {
                    if (j % 4 == 0)
                        store_faces4_if_ne<decltype(full)::value>(offf, cur[r][j], old[r][j], cur[r][(j + 1) % VPL], old[r][(j + 1) % VPL],
                                                                 cur[r][(j + 2) % VPL], old[r][(j + 2) % VPL], cur[r][(j + 3) % VPL], old[r][(j + 3) % VPL], live);
                } else {
                    store_face_if_ne<decltype(full)::value>(offf, cur[r][j], old[r][j], live);
                }
                continue;
#endif
                uint32_t v = cur[r][j];
                asm volatile("" : "+v"(v));               // (compare again: see the row emission)
                const bool f = v != old[r][j];
                const uint32_t at = f ? offf : ftrash;
                *(lds_u32)(uintptr_t)at = v; *(lds_u32)(uintptr_t)(at + 4u) = old[r][j];
                offf += f ? 8u : 0u;
            }
#endif
    };

    // ---- prologue: the plane before the tile (another tile's, or the slab's halo plane) only gives faces
    uint32_t nxt[RB][VPL], nxt_up[VPL], nxt_leftv = INVALID_LABEL;           // edge tiles: the plane in flight, unpacked
    u32x4 nraw[RB], nup_raw;                                                  // interior tiles: the plane just read back
    u32x4 nraw_hi[RB], nup_hi;                                                // (eight uint32 voxels a lane: their second halves)
    constexpr bool WIDE8 = sizeof(T) == 4 && VPL == 8;
    auto land = [&](auto zone, auto keep4) {               // the plane in flight -> nraw / nup_raw / nxt_leftv
        using P = Pin<decltype(zone)::value>;
        if constexpr (WIDE8) {
            u32x4 t[6];
            P::landed_wide(t, nxt_leftv);
            nraw[0] = t[0]; nraw_hi[0] = t[1]; nraw[RB - 1] = t[2]; nraw_hi[RB - 1] = t[3]; nup_raw = t[4]; nup_hi = t[5];
        } else if constexpr (decltype(keep4)::value) {
            P::template landed_keep4<RB>(nraw, nup_raw, nxt_leftv);
        } else {
            P::template landed<RB>(nraw, nup_raw, nxt_leftv);
        }
    };
    auto unpack_row = [&](const int r, uint32_t (&dst)[VPL]) {
        if constexpr (WIDE8) {
            dst[0] = nraw[r].x; dst[1] = nraw[r].y; dst[2] = nraw[r].z; dst[3] = nraw[r].w;
            dst[4 % VPL] = nraw_hi[r].x; dst[5 % VPL] = nraw_hi[r].y; dst[6 % VPL] = nraw_hi[r].z; dst[7 % VPL] = nraw_hi[r].w;
        } else {
            unpack_strip<T, VPL>(nraw[r], dst);
        }
    };
    auto unpack_up = [&](uint32_t (&dst)[VPL]) {
        if constexpr (WIDE8) {
            dst[0] = nup_raw.x; dst[1] = nup_raw.y; dst[2] = nup_raw.z; dst[3] = nup_raw.w;
            dst[4 % VPL] = nup_hi.x; dst[5 % VPL] = nup_hi.y; dst[6 % VPL] = nup_hi.z; dst[7 % VPL] = nup_hi.w;
        } else {
            unpack_strip<T, VPL>(nup_raw, dst);
        }
    };
    using ZoneA = std::integral_constant<int, (PINB ? PINB : TA_PIN_ADJ)>;
#pragma unroll
    for (int j = 0; j < VPL; ++j) { up[j] = INVALID_LABEL; nxt_up[j] = INVALID_LABEL; }
    if constexpr (PINB == 0) {
        if (has_prev) load_rows(p_lo - 1, cur);
        load_rows(p_lo, nxt); load_halo(p_lo, nxt_up, nxt_leftv);
    } else {
        if constexpr (!PAD && TA_PROLOGUE_OVERLAP && RB == 2 && VPL == 8) {       // (the tiles of eight voxels a lane: C5 6.79 -> 6.63 ms, C4 unchanged; the narrow tiles measured +1 %: not there)
            // The plane BEFORE the tile and the tile's first plane in flight TOGETHER (round 4 landed one, then issued the other:
            // a workgroup began its life with two memory latencies back to back, ~13 k cycles of a background tile's 210 k).
            // The older plane goes into ordinary registers -- by hand-issued loads as well: a load the compiler knows about in
            // front of this loop makes it wait for `vmcnt(0)` inside every iteration, behind the next plane's hand-issued loads
            // (measured: +6 ... +9 %) -- and ONE counted wait lets the landing zone's loads, which are younger, stay in flight.
            if (has_prev) {
                const char* row0 = next_row0;
#ifndef TA_ABL_L2
                next_row0 += plane_bytes;
#endif
                constexpr int NQ = WIDE8 ? 2 : 1;                    // 16-byte quads a row
                u32x4 t[RB * NQ];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(t[r * NQ + q]) : "v"(lane_off), "s"(row0 + (int64_t)r * rowbytes + 16 * q) : "memory");
                issue_plane();
                // the landing zone's loads: eight voxels a lane 6 strips + the left voxel, else two rows + the row above + the left voxel
                if constexpr (WIDE8) asm volatile("s_waitcnt vmcnt(7)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[NQ]), "+v"(t[2 * NQ - 1]) :: "memory");
                else                 asm volatile("s_waitcnt vmcnt(4)" : "+v"(t[0]), "+v"(t[RB - 1]) :: "memory");
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    if constexpr (WIDE8) {
                        const u32x4 a = t[r * NQ], c4 = t[r * NQ + NQ - 1];
                        cur[r][0] = a.x; cur[r][1] = a.y; cur[r][2] = a.z; cur[r][3] = a.w;
                        cur[r][4 % VPL] = c4.x; cur[r][5 % VPL] = c4.y; cur[r][6 % VPL] = c4.z; cur[r][7 % VPL] = c4.w;
                    } else {
                        unpack_strip<T, VPL>(t[r * NQ], cur[r]);
                    }
                }
            } else {
                issue_plane();
            }
        } else if (has_prev) {
            issue_plane();
            land(ZoneA{}, std::false_type{});
#pragma unroll
            for (int r = 0; r < RB; ++r) unpack_row(r, cur[r]);
            if constexpr (PAD) {
                uint32_t dump_up[VPL], dump_l = 0u;
#pragma unroll
                for (int j = 0; j < VPL; ++j) dump_up[j] = 0u;
                pad_plane(cur, dump_up, dump_l);
            }
            issue_plane();
        } else {
            issue_plane();
        }
    }

#ifdef TA_STAMPS
    const uint64_t tk_begin = __builtin_amdgcn_s_memtime();
#define TA_T() __builtin_amdgcn_s_memtime()
#endif
    // `process_plane(p)`: plane p has landed (interior tiles: in `nraw` / `nup_raw` / `nxt_leftv`; edge tiles: in `nxt`...)
    auto process_plane = [&](const int32_t p) {
        const uint32_t ploc = (uint32_t)(p - p_lo);
#ifdef TA_STAMPS
        const uint64_t t4 = TA_T();
#endif
        if constexpr (PINB == 0) {
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int j = 0; j < VPL; ++j) { old[r][j] = cur[r][j]; cur[r][j] = nxt[r][j]; }
            leftv = nxt_leftv;
#pragma unroll
            for (int j = 0; j < VPL; ++j) up[j] = nxt_up[j];
            if (p + 1 < p_hi) { load_rows(p + 1, nxt); load_halo(p + 1, nxt_up, nxt_leftv); }
        } else {
            uint32_t nw[RB][VPL], nup[VPL];
#pragma unroll
            for (int r = 0; r < RB; ++r) unpack_row(r, nw[r]);
            if (ADJ) unpack_up(nup);
            pad_plane(nw, nup, nxt_leftv);
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int j = 0; j < VPL; ++j) { old[r][j] = cur[r][j]; cur[r][j] = nw[r][j]; }
            if (ADJ) {
#pragma unroll
                for (int j = 0; j < VPL; ++j) up[j] = nup[j];
            }
            leftv = nxt_leftv;
        }
        if (p == p_lo) first_label = __builtin_amdgcn_readfirstlane(cur[0][0]);
        const bool with_plane_faces = ADJ && (p > p_lo || has_prev);
        const uint32_t cf_plane = with_plane_faces ? count_plane_faces() : 0u;
#ifdef TA_STAMPS
        tk_adv += TA_T() - t4;
#endif
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const uint32_t bloc = (uint32_t)(w * RB + r);
#ifdef TA_STAMPS
            const uint64_t t0 = TA_T();
#endif
            // ---- 1. compares + per-lane event counts
            uint32_t pcv[VPL];
            uint32_t cf = 0u, cr = 0u;
            uint64_t inner = 0ull;                        // boundaries inside the row (the halo compare of lane 0 left out)
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const uint32_t v = cur[r][j];
                if (j > 0) pcv[j] = cur[r][j > 0 ? j - 1 : 0];
                else pcv[j] = lane_shr1(cur[r][VPL - 1], (ADJ && (!EDGE || has_left)) ? (uint32_t)__builtin_amdgcn_readlane((int)leftv, r) : cur[r][0]);
                const bool fc = v != pcv[j];
                cr += fc ? 1u : 0u;
                const uint64_t mc = __builtin_amdgcn_ballot_w64(fc);
                inner |= j == 0 ? (mc & ~1ull) : mc;
                if (ADJ && (r > 0 || !EDGE || has_up)) cf += (v != (r > 0 ? cur[r > 0 ? r - 1 : 0][j] : up[j])) ? 1u : 0u;
            }
#ifdef TA_ABL_NOFACE1
            cf = 0u;                                      // (ablation: results wrong by construction)
#endif
            if (ADJ && r == 0) cf += cf_plane;            // the plane's axis-0 faces ride on its first row's scan
            const uint32_t rowlab = __builtin_amdgcn_readfirstlane(cur[r][0]);
            const bool uniform = inner == 0ull;
            bool need_end;                                // the row's last run closes by a record of lane 63
            if (leading && uniform && rowlab == first_label) { nlead += 1u; need_end = false; }
            else { leading = false; need_end = !(uniform && rowlab == INVALID_LABEL); }
            if (!EDGE && uniform && rowlab == INVALID_LABEL) bad = true;      // not a filler: a voxel above any max_label
            if (need_end && lane == 63) cr += 1u;
            const uint32_t cnt = cf | (cr << 16);         // faces in the low half, runs in the high half
#ifdef TA_STAMPS
            const bool anyev_ = __builtin_amdgcn_ballot_w64(cnt != 0u) != 0ull;
            const uint64_t t1 = TA_T();
            tk_cmp += t1 - t0;
            if (!anyev_) continue;
            tk_evrows += 1;
#else
            if (__builtin_amdgcn_ballot_w64(cnt != 0u) == 0ull) continue;       // the common case: one branch per row
#endif
            // ---- 2. + 3. offsets, then a firing lane stores at its own offset and steps it
            place_records(cnt, [&](uint32_t offf, uint32_t offr, auto full) {
                // (laundered: keeps the tagged / coded copies of the row's registers out of long-lived registers)
                uint32_t tag1 = 1u << 30;
                uint32_t rowcode = (uint32_t)__builtin_amdgcn_readfirstlane((int)((bloc << 10) | (ploc << 14)));
                if (ADJ) asm volatile("" : "+s"(tag1));
                asm volatile("" : "+s"(rowcode));
                constexpr bool FULL = decltype(full)::value;
                // (the lanes that are live here -- all of them on the straight-line path, a lane range in place_in_pieces.  Laundered: the
                //  predicated stores put EXEC back from this COPY; handed the ballot itself, the compiler passes `exec` as the operand
                //  and the restore becomes `s_mov_b64 exec, exec`)
                uint64_t live = FULL ? ~0ull : __builtin_amdgcn_ballot_w64(true);
                if (!FULL) asm volatile("" : "+s"(live));
#ifdef TA_DBG_EMIT
                const uint32_t dbg_off0 = offf;
#endif
                if (ADJ && r == 0) { if (with_plane_faces) emit_plane_faces(offf, live, full); }
#ifdef TA_DBG_EMIT
                if (ADJ && r == 0 && with_plane_faces) {
                    const uint32_t wrote = (offf - dbg_off0) >> 3, want = count_plane_faces();
                    if (wrote != want || (wrote != 0u && (offf > fbase + 8u * (uint32_t)FCAP || dbg_off0 < fbase))) {
                        uint32_t* fl = cold_args(kp)->flags;
                        if (atomicAdd(&fl[15], 1u) == 0u) { fl[8] = wrote; fl[9] = want; fl[10] = (uint32_t)lane; fl[11] = (dbg_off0 - fbase) >> 3; fl[12] = fcount; fl[13] = FULL ? 1u : 0u; fl[14] = (uint32_t)__builtin_amdgcn_ballot_w64(true); }
                    }
                }
#endif
                const uint32_t code0 = lane_c | rowcode;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
#ifndef TA_ABL_NOFACE1
                    if (ADJ && (r > 0 || !EDGE || has_up))
#else
                    if (false)
#endif
                    {
                        const uint32_t pv = r > 0 ? cur[r > 0 ? r - 1 : 0][j] : up[j];
#if TA_MASKED_STORES & 2
                        if constexpr (VPL % 4 == 0) {
                            if (j % 4 == 0) {
                                const uint32_t pv1 = r > 0 ? cur[r > 0 ? r - 1 : 0][(j + 1) % VPL] : up[(j + 1) % VPL];
                                const uint32_t pv2 = r > 0 ? cur[r > 0 ? r - 1 : 0][(j + 2) % VPL] : up[(j + 2) % VPL];
                                const uint32_t pv3 = r > 0 ? cur[r > 0 ? r - 1 : 0][(j + 3) % VPL] : up[(j + 3) % VPL];
                                store_faces4_tagged_if_ne<FULL>(offf, cur[r][j], pv, cur[r][(j + 1) % VPL], pv1, cur[r][(j + 2) % VPL], pv2,
                                                                cur[r][(j + 3) % VPL], pv3, tag1, live);
                            }
                        } else {
                            store_face_tagged_if_ne<FULL>(offf, cur[r][j], pv, tag1, live);
                        }
#else
                        // (every lane stores at every compare without touching the exec mask: at its own offset when the compare
                        //  fired, into a trash slot when it did not -- the pre-round-5 stores, kept for comparison)
                        uint32_t v = cur[r][j];
                        asm volatile("" : "+v"(v));
                        const bool f = v != pv;
                        const uint32_t at = f ? offf : ftrash;
                        *(lds_u32)(uintptr_t)at = v; *(lds_u32)(uintptr_t)(at + 4u) = pv | tag1;
                        offf += f ? 8u : 0u;
#endif
                    }
                }
                // (the runs behind the faces; records of a row in column order)
#if TA_MASKED_STORES & 4
                static_assert(VPL == 4 || VPL == 8, "the run stores are written out for four or eight voxels a lane");
                store_runs4_if_ne<FULL, 0, (int)(RSTRIDE / 4u)>(offr, pcv[0], cur[r][0], cur[r][1 % VPL], cur[r][2 % VPL], cur[r][3 % VPL], code0, live);
                if constexpr (VPL == 8)
                    store_runs4_if_ne<FULL, 4, (int)(RSTRIDE / 4u)>(offr, cur[r][3 % VPL], cur[r][4 % VPL], cur[r][5 % VPL], cur[r][6 % VPL], cur[r][7 % VPL], code0, live);
                static_assert(VPL <= 8, "the run stores are written out for eight voxels a lane");
#else
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    uint32_t v = cur[r][j];
                    asm volatile("" : "+v"(v));
                    const bool g = v != pcv[j];
                    const uint32_t at = g ? offr : rtrash;
                    *(lds_u32)(uintptr_t)at = pcv[j];
                    *(lds_u32)(uintptr_t)(at + RSTRIDE) = code0 | (uint32_t)j;
                    offr += g ? 4u : 0u;
                }
#endif
                if (TA_MASKED_STORES) emit_done();
                if (need_end && lane == 63) {
                    *(lds_u32)(uintptr_t)offr = cur[r][VPL - 1];
                    *(lds_u32)(uintptr_t)(offr + RSTRIDE) = (uint32_t)TC | rowcode | ROW_END;
                }
            });
#ifdef TA_STAMPS
            tk_emit += TA_T() - t1;
#endif
        }
    };
    // ---- the planes of the tile: the plane in flight lands (it was issued a whole plane of compute ago), the next one is
    //      issued into the registers it leaves, the landed one is processed.  (Measured and dropped: TWO planes in flight for
    //      the two-row tiles of uint16 volumes, in the two halves of the sixteen pinned registers -- C2 stayed at 0.070 ms:
    //      that kernel is bound by its instructions per voxel, not by the bytes it has in flight.)
    auto top_drains = [&]() {
        if constexpr (DRAIN_ALL) {
            if (fcount >= (uint32_t)TA_FDRAIN) drain_face_groups<2, LDS>(kp, S, W, lane, fcount);
            else if (TA_FDRAIN1 && VPL == 8 && fcount >= 64u) drain_face_groups<1, LDS>(kp, S, W, lane, fcount);   // (the wide tiles: see TA_FDRAIN1)
            if (rcount > 64u) drain_run_group<MOM2, LDS>(kp, S, W, EDGE, lane, rcount, first_label);
        }
    };
    if constexpr (PINB != 0 && PINB2 != 0) {
        // two planes in flight: plane p lands from zone A (PINB) for even p - p_lo, from zone B (PINB2) for odd; each landing
        // leaves the other zone's four loads in flight and re-issues its own zone two planes ahead
        using ZA = std::integral_constant<int, PINB>;
        using ZB = std::integral_constant<int, PINB2>;
        if (p_lo + 1 < p_hi) issue_plane_to(ZB{});
        for (int32_t p = p_lo; p < p_hi; p += 2) {
            top_drains();
            if (p + 1 < p_hi) land(ZA{}, std::true_type{});
            else land(ZA{}, std::false_type{});
            if (p + 2 < p_hi) issue_plane_to(ZA{});
            process_plane(p);
            if (p + 1 < p_hi) {
                top_drains();
                if (p + 2 < p_hi) land(ZB{}, std::true_type{});
                else land(ZB{}, std::false_type{});
                if (p + 3 < p_hi) issue_plane_to(ZB{});
                process_plane(p + 1);
            }
        }
    } else {
        for (int32_t p = p_lo; p < p_hi; ++p) {
            if constexpr (DRAIN_ALL) {
                // the hot drains: between two planes a wave holds nothing but the plane before (and the next one is in flight)
#ifdef TA_RECCOUNT
                if (lane == 0 && fcount >= (uint32_t)TA_FDRAIN) atomicAdd(&cold_args(kp)->flags[11], 128u);
                else if (lane == 0 && TA_FDRAIN1 && VPL == 8 && fcount >= 64u) atomicAdd(&cold_args(kp)->flags[11], 64u);
                if (lane == 0 && rcount > 64u) atomicAdd(&cold_args(kp)->flags[12], 64u);
#endif
                if (fcount >= (uint32_t)TA_FDRAIN) drain_face_groups<2, LDS>(kp, S, W, lane, fcount);
                else if (TA_FDRAIN1 && VPL == 8 && fcount >= 64u) drain_face_groups<1, LDS>(kp, S, W, lane, fcount);
                if (rcount > 64u) drain_run_group<MOM2, LDS>(kp, S, W, EDGE, lane, rcount, first_label);
            }
            if constexpr (PINB != 0) {
#ifdef TA_STAMPS
                const uint64_t t4 = TA_T();
#endif
                land(ZoneA{}, std::false_type{});
#ifdef TA_STAMPS
                tk_land += TA_T() - t4;
#endif
                if (p + 1 < p_hi) issue_plane();
            }
            process_plane(p);
        }
    }
#ifdef TA_STAMPS
    if (lane == 0) {
        uint32_t* fl = cold_args(kp)->flags;
        atomicAdd(&fl[8], (uint32_t)(tk_cmp >> 8)); atomicAdd(&fl[9], (uint32_t)(tk_emit >> 8));
        atomicAdd(&fl[10], (uint32_t)(tk_drain >> 8)); atomicAdd(&fl[11], (uint32_t)(tk_adv >> 8));
        atomicAdd(&fl[12], (uint32_t)((TA_T() - tk_begin) >> 8));
        atomicAdd(&fl[13], (uint32_t)(tk_land >> 8)); atomicAdd(&fl[14], (uint32_t)tk_evrows); atomicAdd(&fl[15], (uint32_t)tk_drains);
    }
#endif

    // ---- end of tile: drain the buffers, then the leading one-label rows in one closed form
    drain(false);
    if (__builtin_amdgcn_ballot_w64(bad)) { if (lane == 0) atomicOr(&cold_args(kp)->flags[FLAG_RANGE], 1u); }
    if (lane == 0 && nlead != 0u && first_label != INVALID_LABEL) {
        // rows in (plane, row) order: P full planes of RB rows, then R rows of plane P
        const uint64_t P = nlead / RB, R = nlead % RB, nc = TC, b0 = (uint64_t)w * RB;
        const uint64_t t1c = range_sum1(0, nc), t2c = range_sum2(0, nc);
        const uint64_t rows = P * RB + R;                                        // = nlead
        const uint64_t sa_rows = range_sum1(0, P) * RB + P * R;                  // sum of a over the rows
        const uint64_t saa_rows = range_sum2(0, P) * RB + P * P * R;
        const uint64_t sb_rows = P * range_sum1(b0, RB) + range_sum1(b0, R);
        const uint64_t sbb_rows = P * range_sum2(b0, RB) + range_sum2(b0, R);
        const uint64_t sab_rows = range_sum1(0, P) * range_sum1(b0, RB) + P * range_sum1(b0, R);
        LocalSums L;
        L.n = rows * nc; L.sa = sa_rows * nc; L.sb = sb_rows * nc; L.sc = rows * t1c;
        if (MOM2) {
            L.saa = saa_rows * nc; L.sab = sab_rows * nc; L.sbb = sbb_rows * nc;
            L.sac = sa_rows * t1c; L.sbc = sb_rows * t1c; L.scc = rows * t2c;
        } else {
            L.saa = L.sab = L.sac = L.sbb = L.sbc = L.scc = 0;
        }
        const uint32_t mxa = (uint32_t)(R ? P : P - 1u);
        const uint32_t mxb = (uint32_t)(b0 + (P ? RB : R) - 1u);
        const uint32_t fh = scan_label_hash(first_label);
        scan_label_add<MOM2, LDS, LocalSums>(kp, S, first_label, L, 0u, mxa, (uint32_t)b0, mxb, 0u,
                                             (uint32_t)(nc - 1), fh, S.lkeys[fh]);
    }
}

// Tiles of a volume: `fc` x `fb` full tiles per plane band go to the interior kernel (hand-issued 16-byte loads, no
// bounds); the partial tiles of the last tile column / tile row go to the PADDED kernel (the same loads from clamped
// addresses, filler written over what lies outside) when 16-byte loads are allowed (SweepArgs::vec_ok), else everything
// goes to the plain edge kernel (guarded scalar loads).  All write the same tables; `wg0` numbers the second launch's
// workgroups after the interior kernel's (private hot-label rows).
struct ScanSplit { uint32_t tiles_c, tiles_b, fc, fb, nbands, padded; };

template <int VPL, int RB>
static ScanSplit scan_split(const SweepArgs& a, int itemsize) {
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    ScanSplit s;
    const int64_t owned = a.n0 - a.first_owned;
    s.tiles_c = (uint32_t)((a.n2 + TC - 1) / TC); s.tiles_b = (uint32_t)((a.n1 + TB - 1) / TB);
    s.nbands = owned <= 0 ? 0u : (uint32_t)((owned + a.tile_planes - 1) / a.tile_planes);
    const bool fast = a.vec_ok && a.n2 * itemsize * (int64_t)RB < (1ll << 31);
    s.fc = fast ? (uint32_t)(a.n2 / TC) : 0u; s.fb = fast ? (uint32_t)(a.n1 / TB) : 0u;
    if (s.fc == 0 || s.fb == 0) { s.fc = 0; s.fb = 0; }
    s.padded = fast ? 1u : 0u;         // the partial tiles go to the padded kernel (vector loads allowed), else to the plain edge kernel
    return s;
}

// The next tile of a persistent workgroup: from the list of its own XCD (workgroups are dealt round-robin over the 8 XCDs;
// an XCD's list is every eighth CHUNK of TA_XCD_CHUNK consecutive tiles -- neighbours along axes 2 and 1, so that the halo
// row a tile reads is the row its neighbour on the same XCD / L2 reads at about the same time -- plus its share of the last,
// partial group of chunks), then from the other XCDs' lists.  Speed only: any workgroup may take any tile.
__device__ __forceinline__ uint32_t claim_tile(uint32_t* queues, const uint32_t xcd, const uint32_t ntiles) {
    constexpr uint32_t G = TA_XCD_CHUNK > 0 ? TA_XCD_CHUNK : 1;
    const uint32_t per = (ntiles / (8u * G)) * G, full = per * 8u;          // tiles of the whole groups of chunks: per XCD, in all
#if TA_PERSIST_SINGLE_QUEUE      // (experiment: one queue, tiles in plain order)
    {
        const uint32_t j = __hip_atomic_fetch_add(&queues[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return j < ntiles ? j : NO_TILE;
    }
#endif
#pragma nounroll
    for (uint32_t s = 0; s < 8u; ++s) {
        const uint32_t x = (xcd + s) & 7u;
        const uint32_t nx = per + (ntiles - full + 7u - x) / 8u;            // ... plus every eighth tile of the rest
        // (an empty list is seen by a plain load: at the end every workgroup walks all eight lists, and a thousand
        //  read-modify-writes of one word take their time)
        if (__hip_atomic_load(&queues[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nx) continue;
        const uint32_t j = __hip_atomic_fetch_add(&queues[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (j < nx) return j < per ? ((j / G) * 8u + x) * G + (j % G) : full + (j - per) * 8u + x;
    }
    return NO_TILE;
}

// PERSIST: the workgroup walks tiles until the queues are empty: its tables are emptied by the flush itself, the flush's global
// atomics are left in flight while the next tile starts, and no workgroup has to be launched (and waited out) per tile.
template <typename T, int VPL, int RB, bool ADJ, bool MOM2, bool EDGE, int PINB, bool PERSIST = false, int PINB2 = 0>
__device__ __forceinline__ void scan_kernel_body(const SweepArgs& A, const ScanSplit& sp, const uint32_t wg0) {
    constexpr int NW = MOM2 ? 4 : 2;
    constexpr int TC = 64 * VPL, TB = WAVES * RB;
    static_assert(TB <= 16 && TC <= 512, "packed LDS moment words assume <= 16 rows x 512 columns per tile");
    static_assert(!PERSIST || !EDGE, "the persistent kernel walks the full tiles");
    // two replicas of the label sums where the LDS has room at the kernel's occupancy (the tiles of eight voxels a lane: four
    // workgroups a CU); the narrow uint32 tiles sit at five workgroups a CU and 31 KB, the moments-only kernels are not bound here
    constexpr int REP = (ADJ && VPL == 8 && TA_LSUM_REP > 1) ? TA_LSUM_REP : 1;
    using LDS = ScanLds<NW, ADJ, REP, SumPack<tile_planes_cap(ADJ, (int)sizeof(T), VPL), TB, TC>>;
    __shared__ LDS S;
    // the arguments only the cold paths need are re-read from the kernarg segment there (see cold_args)
    const SweepArgs* kp = kernarg_args(A);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < LSLOTS; i += WAVES * 64) {
        S.lkeys[i] = INVALID_LABEL;
#pragma unroll
        for (int k = 0; k < NW * REP; ++k) S.lsum[i * NW * REP + k] = 0ull;
        S.lbox[i * 8 + 0] = 0xFFFFFFFFu; S.lbox[i * 8 + 1] = 0xFFFFFFFFu; S.lbox[i * 8 + 2] = 0xFFFFFFFFu;
        S.lbox[i * 8 + 3] = 0u; S.lbox[i * 8 + 4] = 0u; S.lbox[i * 8 + 5] = 0u;
    }
    if (ADJ) {
        for (int i = tid; i < PSLOTS; i += WAVES * 64) {
            S.pkeys[i] = EMPTY_KEY;
#if TA_PCNT64
            S.pcnt[i] = 0ull;
#else
            S.pcnt[i * 3 + 0] = 0u; S.pcnt[i * 3 + 1] = 0u; S.pcnt[i * 3 + 2] = 0u;
#endif
        }
    }
    uint32_t tile = blockIdx.x;
    if (PERSIST) {
        if (tid == 0) S.frame[3] = claim_tile(cold_args(kp)->flags + QUEUE_WORD, blockIdx.x & 7u, sp.fc * sp.fb * sp.nbands);
        __syncthreads();
        tile = S.frame[3];
    }
    for (;;) {
        if (PERSIST && tile == NO_TILE) break;
        uint32_t t = tile, tc, tb, band;
#if TA_XCD_CHUNK > 0
        // Workgroups are dealt round-robin over the 8 XCDs (speed only, never correctness: placement is not promised).
        // Give each XCD CHUNKS of consecutive tiles -- neighbours along axes 2 and 1 -- so that the halo row a tile reads is
        // the row its neighbour on the same XCD reads at about the same time (one HBM fetch, one L2), while chunks stay
        // small enough that tissue and background tiles still spread evenly over the XCDs.  (PERSIST: claim_tile does it.)
        if (!EDGE && !PERSIST) {
            constexpr uint32_t G = TA_XCD_CHUNK;
            const uint32_t full = (gridDim.x / (8u * G)) * (8u * G);
            if (t < full) {
                const uint32_t xcd = t % 8u, k = t / 8u;
                t = ((k / G) * 8u + xcd) * G + (k % G);
            }
        }
#endif
        if (!EDGE) {
            tc = t % sp.fc; t /= sp.fc; tb = t % sp.fb; band = t / sp.fb;
        } else {
            const uint32_t per_band = sp.tiles_c * sp.tiles_b - sp.fc * sp.fb, strip = (sp.tiles_c - sp.fc) * sp.tiles_b;
            band = t / per_band; t -= band * per_band;
            if (t < strip) { tc = sp.fc + t % (sp.tiles_c - sp.fc); tb = t / (sp.tiles_c - sp.fc); }       // the last tile column(s)
            else { t -= strip; tc = t % sp.fc; tb = sp.fb + t / sp.fc; }                                    // the last tile row(s)
        }
        // (the private hot-label row of a tile: numbered by the tile, whichever workgroup walks it)
        if (TA_HOT_ADJ || !ADJ) hot_row_init(A, tid, wg0 + tile);
        const uint32_t c_tile0 = tc * TC, b_tile0 = tb * TB;
        const int32_t p_lo = A.first_owned + (int32_t)band * A.tile_planes;
        int32_t p_hi = p_lo + A.tile_planes;
        if (p_hi > (int32_t)A.n0) p_hi = (int32_t)A.n0;
        const uint64_t A0 = (uint64_t)(A.a_origin + (p_lo - A.first_owned));
        if (tid == 0) { S.frame[0] = (uint32_t)A0; S.frame[1] = b_tile0; S.frame[2] = c_tile0; }
        __syncthreads();

#ifdef TA_BARSTAMP       // (instrumentation: where a workgroup's life goes -- flags[8..12] = cycles >> 8 summed over the waves)
        const uint64_t tb_start = __builtin_amdgcn_s_memtime();
#endif
        if (p_lo < p_hi)
            wave_scan<T, VPL, RB, ADJ, MOM2, EDGE, PINB, PINB2>(A, kp, S, lane, w, c_tile0, b_tile0, p_lo, p_hi);
#ifdef TA_BARSTAMP
        const uint64_t tb_scan = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef TA_BARSTAMP
        const uint64_t tb_bar = __builtin_amdgcn_s_memtime();
#endif
        // (everything the flush needs is re-read -- arguments from the kernarg segment, the tile origin from LDS --
        //  rather than kept in scarce SGPRs across the sweep)
        const KernelArgs ka = scalar_args(kp);
        const SweepArgs& Ac = ka.a;
        const uint32_t wg_ = ka.wg0 + tile;
        uint32_t next = NO_TILE;
        if (PERSIST && tid == 0)            // (asked for before the flush, needed after it; split: tiles_c, tiles_b, fc, fb, nbands, padded)
            next = claim_tile(Ac.flags + QUEUE_WORD, blockIdx.x & 7u, ka.split[2] * ka.split[3] * ka.split[4]);
#ifndef TA_ABL_NOFLUSH
        flush_tables<NW, ADJ, MOM2, (TA_HOT_ADJ || !ADJ), LDS, WAVES * 64, PERSIST>(Ac, S, threadIdx.x, (uint64_t)S.frame[0], (uint64_t)S.frame[1], (uint64_t)S.frame[2],
                                          (TA_HOT_ADJ || !ADJ) ? hot_label_of<T>(Ac) : 0u, wg_);
#else
        if (wg_ == 0xffffffffu) S.frame[3] = 1;
#endif
#ifdef TA_BARSTAMP
        if (lane == 0) {
            uint32_t* fl = cold_args(kp)->flags;
            const uint64_t tb_end = __builtin_amdgcn_s_memtime();
            atomicAdd(&fl[8], (uint32_t)((tb_scan - tb_start) >> 8)); atomicAdd(&fl[9], (uint32_t)((tb_bar - tb_scan) >> 8));
            atomicAdd(&fl[10], (uint32_t)((tb_end - tb_bar) >> 8)); atomicAdd(&fl[11], 1u);
        }
#endif
        if (!PERSIST) break;
        __syncthreads();                     // (the tile origin has been read by every thread's flush)
        if (tid == 0) S.frame[3] = next;
        __syncthreads();
        tile = S.frame[3];
    }
}

// (several entry points only because the VGPR budget is an attribute and must be a literal)
template <typename T, int VPL, int RB, bool MOM2, bool EDGE>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_vgpr(TA_CAP_ADJ))) scan_kernel(SweepArgs A, ScanSplit sp, uint32_t wg0) {
    scan_kernel_body<T, VPL, RB, true, MOM2, EDGE, EDGE ? 0 : TA_PIN_ADJ>(A, sp, wg0);
}
// the full tiles of a uint32 volume with adjacency: two rows per wave, five waves per SIMD.
// TA_PERSIST (measured, NOT adopted: profiles/r04_NOTES.md): 1280 persistent workgroups that take tiles from per-XCD queues and
// leave the flush's global atomics in flight.  C4 1.30 ms against 1.07 ms with a workgroup per tile -- the same instructions
// (SQ_INSTS_* equal to 1 %), but the waves are parked twice as long (SQ_WAIT_ANY 2.3e9 against 1.1e9 quad-cycles).
template <typename T, int VPL, int RB, bool MOM2>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_vgpr(TA_CAP_ADJ2))) scan_two_rows_kernel(SweepArgs A, ScanSplit sp, uint32_t wg0) {
    static_assert(RB == 2 && VPL == 4, "the 13-register landing zone holds two rows of four voxels a lane");
    scan_kernel_body<T, VPL, RB, true, MOM2, false, TA_PIN_ADJ2, TA_PERSIST != 0, (TA_PLANES_IN_FLIGHT == 2 ? TA_PIN_ADJ2B : 0)>(A, sp, wg0);
}
template <typename T, int VPL, int RB, bool MOM2, bool EDGE>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_vgpr(TA_CAP_MOM))) scan_noadj_kernel(SweepArgs A, ScanSplit sp, uint32_t wg0) {
    scan_kernel_body<T, VPL, RB, false, MOM2, EDGE, EDGE ? 0 : TA_PIN_MOM>(A, sp, wg0);
}
// the partial tiles of a volume with 16-byte aligned rows: hand-issued loads like the interior tiles, filler like the edge tiles
template <typename T, int VPL, int RB, bool MOM2>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_vgpr(TA_CAP_ADJ_PAD))) scan_pad_kernel(SweepArgs A, ScanSplit sp, uint32_t wg0) {
    scan_kernel_body<T, VPL, RB, true, MOM2, true, TA_PIN_ADJ_PAD>(A, sp, wg0);
}
template <typename T, int VPL, int RB, bool MOM2, bool EDGE>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_vgpr(TA_CAP_ADJ8))) scan_wide_kernel(SweepArgs A, ScanSplit sp, uint32_t wg0) {
    scan_kernel_body<T, VPL, RB, true, MOM2, EDGE, EDGE ? 0 : TA_PIN_ADJ8>(A, sp, wg0);
}
template <typename T, int VPL, int RB, bool MOM2>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_vgpr(TA_CAP_ADJ8_PAD))) scan_wide_pad_kernel(SweepArgs A, ScanSplit sp, uint32_t wg0) {
    scan_kernel_body<T, VPL, RB, true, MOM2, true, TA_PIN_ADJ8_PAD>(A, sp, wg0);
}
template <typename T, int VPL, int RB, bool MOM2>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_vgpr(TA_CAP_ADJ2_PAD))) scan_pad2_kernel(SweepArgs A, ScanSplit sp, uint32_t wg0) {
    static_assert(RB == 2 && VPL == 4, "the 13-register landing zone holds two rows of four voxels a lane");
    scan_kernel_body<T, VPL, RB, true, MOM2, true, TA_PIN_ADJ2_PAD>(A, sp, wg0);
}
template <typename T, int VPL, int RB, bool MOM2>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_vgpr(TA_CAP_MOM_PAD))) scan_noadj_pad_kernel(SweepArgs A, ScanSplit sp, uint32_t wg0) {
    scan_kernel_body<T, VPL, RB, false, MOM2, true, TA_PIN_MOM_PAD>(A, sp, wg0);
}

// ev_start / ev_stop (optional): HIP events attached to the launches themselves (hipExtLaunchKernelGGL) -- they carry the
// dispatch's own begin / end timestamps, and cost no separate event-record packet (~4 us each) on the queue.
template <typename T, int VPL, int RB, bool ADJ, bool MOM2>
static void launch_scan_tt(hipStream_t s, const SweepArgs& a, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const ScanSplit sp = scan_split<VPL, RB>(a, (int)sizeof(T));
    if (sp.nbands == 0 || a.n1 <= 0 || a.n2 <= 0) return;
    const uint32_t n_in = sp.fc * sp.fb * sp.nbands;
    const uint32_t n_ed = (sp.tiles_c * sp.tiles_b - sp.fc * sp.fb) * sp.nbands;
    const dim3 block(WAVES * 64);
    hipEvent_t in0 = ev_start, in1 = n_ed ? nullptr : ev_stop;              // interior launch
    hipEvent_t ed0 = n_in ? nullptr : ev_start, ed1 = ev_stop;              // edge launch
    if constexpr (ADJ && sizeof(T) == 4 && VPL == 8) {
        if (n_in) hipExtLaunchKernelGGL((scan_wide_kernel<T, VPL, RB, MOM2, false>), dim3(n_in), block, 0, s, in0, in1, 0, a, sp, 0u);
        if (n_ed && sp.padded) hipExtLaunchKernelGGL((scan_wide_pad_kernel<T, VPL, RB, MOM2>), dim3(n_ed), block, 0, s, ed0, ed1, 0, a, sp, n_in);
        if (n_ed && !sp.padded) hipExtLaunchKernelGGL((scan_wide_kernel<T, VPL, RB, MOM2, true>), dim3(n_ed), block, 0, s, ed0, ed1, 0, a, sp, n_in);
    } else if constexpr (ADJ) {
        if constexpr (VPL == 4 && RB == 2) {
            const uint32_t grid = TA_PERSIST ? (n_in < (uint32_t)TA_PERSIST_WGS ? n_in : (uint32_t)TA_PERSIST_WGS) : n_in;
            if (n_in) hipExtLaunchKernelGGL((scan_two_rows_kernel<T, VPL, RB, MOM2>), dim3(grid), block, 0, s, in0, in1, 0, a, sp, 0u);
        } else {
            if (n_in) hipExtLaunchKernelGGL((scan_kernel<T, VPL, RB, MOM2, false>), dim3(n_in), block, 0, s, in0, in1, 0, a, sp, 0u);
        }
        if constexpr (VPL == 4 && RB == 2) {
            if (n_ed && sp.padded) hipExtLaunchKernelGGL((scan_pad2_kernel<T, VPL, RB, MOM2>), dim3(n_ed), block, 0, s, ed0, ed1, 0, a, sp, n_in);
        } else {
            if (n_ed && sp.padded) hipExtLaunchKernelGGL((scan_pad_kernel<T, VPL, RB, MOM2>), dim3(n_ed), block, 0, s, ed0, ed1, 0, a, sp, n_in);
        }
        if (n_ed && !sp.padded) hipExtLaunchKernelGGL((scan_kernel<T, VPL, RB, MOM2, true>), dim3(n_ed), block, 0, s, ed0, ed1, 0, a, sp, n_in);
    } else {
        if (n_in) hipExtLaunchKernelGGL((scan_noadj_kernel<T, VPL, RB, MOM2, false>), dim3(n_in), block, 0, s, in0, in1, 0, a, sp, 0u);
        if (n_ed && sp.padded) hipExtLaunchKernelGGL((scan_noadj_pad_kernel<T, VPL, RB, MOM2>), dim3(n_ed), block, 0, s, ed0, ed1, 0, a, sp, n_in);
        else if (n_ed) hipExtLaunchKernelGGL((scan_noadj_kernel<T, VPL, RB, MOM2, true>), dim3(n_ed), block, 0, s, ed0, ed1, 0, a, sp, n_in);
    }
}

// Rows per wave: uint16 volumes 2 x 512 columns; uint32 volumes 4 x 256 without adjacency (four rows in flight per wave: the
// moments-only kernel is bound by the stream) and 2 x 256 with it (half the plane state: 95 VGPRs and 31 KB of LDS make five
// waves per SIMD of a kernel that is bound by latency at four -- C4 1.13 -> 1.06 ms, tissue-filled 1.51 -> 1.40 ms at 48-plane tiles).
// uint32 volumes with adjacency: TA_U32_SHAPE 0 = two rows of 256 columns a wave (four voxels a lane, the 13-register landing
// zone, 95 VGPRs, five waves per SIMD); 1 = two rows of 512 columns (eight voxels a lane through two 16-byte loads a row, a
// 25-register zone, 128 VGPRs, four waves): the shape of the uint16 kernel, twice the voxels per plane step of a wave
constexpr int RB32_ADJ = 2, RB32_MOM = 4;
// uint16 volumes with adjacency: 4 voxels a lane like the uint32 kernel (8-byte strips read as the first half of a 16-byte
// load: the same 13-register landing zone, the same plane state, five waves per SIMD) or 8 (the pre-round-3 shape, 125 VGPRs)
constexpr int VPL16_ADJ = TA_U16_VPL;
constexpr int RB16_MOM = TA_U16_MOM_RB;

uint64_t sweep_grid_size(const SweepArgs& a, int itemsize, bool adjacency) {
    const ScanSplit sp = itemsize == 2 ? (adjacency ? scan_split<VPL16_ADJ, 2>(a, 2) : scan_split<8, RB16_MOM>(a, 2)) : (adjacency ? (a.shape ? scan_split<8, RB32_ADJ>(a, 4) : scan_split<4, RB32_ADJ>(a, 4)) : scan_split<4, RB32_MOM>(a, 4));
    return (uint64_t)sp.tiles_c * sp.tiles_b * sp.nbands;
}
// measured on C4 / C5 (profiles/r03_ablations.txt, gpurun_out/r3_rb2_tp.txt): shorter tiles = more workgroups to balance over
// the CUs against more table inits / flushes.  The two-row tiles of uint32 volumes with adjacency cover half the rows, so
// they walk twice the planes for the same voxels per workgroup (40 and 48 measure the same, 24 is 3 % slower, 64 1 %).
// (The wide uint32 tiles -- 8 rows x 512 columns -- hold twice the voxels a plane: 32 planes measure best on C4 and C5, and at
//  48 the tissue-filled volume overflows the workgroup tables.)
// (Round 5, after the flush got cheaper: the narrow tiles measure best at 40 planes -- tissue-filled 1.221 against 1.236 ms at 48 and 1.241 at 32, C4
//  0.970 against 0.978; the wide tiles at 24 .. 32 on C4 (within the noise), 32 on C5, 24 where every tile is tissue: profiles/r05_tile_planes.txt.)
// (uint16 volumes: 1024^3 0.681 / 0.684 ms at 28 planes against 0.686 / 0.688 at 32, 0.683 / 0.687 at 24; C2 is cut shorter by the launch anyway)
int sweep_default_tile_planes(bool adjacency, int itemsize, int shape) { return adjacency ? (itemsize == 4 ? (shape ? 32 : 40) : 28) : 16; }
int sweep_tile_planes_limit() { return MAX_TILE_PLANES; }
// the tile height a kernel's packed sums are sized for (SumPack; results do not depend on the tile height)
int sweep_max_tile_planes(bool adjacency, int itemsize, int shape) {
    const int vpl = adjacency ? (itemsize == 2 ? VPL16_ADJ : (shape ? 8 : 4)) : (itemsize == 2 ? 8 : 4);
    const int cap = tile_planes_cap(adjacency, itemsize, vpl);
    return cap < MAX_TILE_PLANES ? cap : MAX_TILE_PLANES;
}

void launch_scan(hipStream_t s, const SweepArgs& a, int itemsize, uint32_t feature_mask, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const bool adj = feature_mask & 16u, mom2 = feature_mask & 8u;
    if (itemsize == 2) {
        if (adj && mom2)       launch_scan_tt<uint16_t, VPL16_ADJ, 2, true, true>(s, a, ev_start, ev_stop);
        else if (adj)          launch_scan_tt<uint16_t, VPL16_ADJ, 2, true, false>(s, a, ev_start, ev_stop);
        else if (mom2)         launch_scan_tt<uint16_t, 8, RB16_MOM, false, true>(s, a, ev_start, ev_stop);
        else                   launch_scan_tt<uint16_t, 8, RB16_MOM, false, false>(s, a, ev_start, ev_stop);
    } else {
        if (adj && mom2)       { if (a.shape) launch_scan_tt<uint32_t, 8, RB32_ADJ, true, true>(s, a, ev_start, ev_stop); else launch_scan_tt<uint32_t, 4, RB32_ADJ, true, true>(s, a, ev_start, ev_stop); }
        else if (adj)          { if (a.shape) launch_scan_tt<uint32_t, 8, RB32_ADJ, true, false>(s, a, ev_start, ev_stop); else launch_scan_tt<uint32_t, 4, RB32_ADJ, true, false>(s, a, ev_start, ev_stop); }
        else if (mom2)         launch_scan_tt<uint32_t, 4, RB32_MOM, false, true>(s, a, ev_start, ev_stop);
        else                   launch_scan_tt<uint32_t, 4, RB32_MOM, false, false>(s, a, ev_start, ev_stop);
    }
}

}  // namespace ta
