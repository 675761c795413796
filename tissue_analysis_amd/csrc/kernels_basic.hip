// kernels_basic.hip -- small gfx950 kernels around the fused sweep:
//   init of the per-label accumulators, the per-voxel-atomics cross-check kernel (TA_OPT_IMPL=1),
//   max-label reduction, adjacency hash collect / insert / clear, and the synthetic generator.
#include "ta_kernels.h"

namespace ta {

// ------------------------------------------------------------------------------------------
// accumulator init: sums = 0, boxes = INT32_MAX, flags = 0, cursor = 0 (one launch, 16 B stores)
__global__ void __launch_bounds__(256) init_kernel(uint64_t* sums, int32_t* boxes, uint64_t nlabels,
                                                   uint32_t* flags, uint32_t* cursor, uint64_t* hot_rows) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nsum2 = nlabels * NSUM / 2;              // NSUM is even: ulong2 stores
    ulonglong2* s2 = reinterpret_cast<ulonglong2*>(sums);
    for (uint64_t i = tid; i < nsum2; i += nthreads) s2[i] = make_ulonglong2(0ull, 0ull);
    const uint64_t nbox2 = nlabels * NBOX / 2;              // NBOX is even: int2 stores
    int2* b2 = reinterpret_cast<int2*>(boxes);
    for (uint64_t i = tid; i < nbox2; i += nthreads) b2[i] = make_int2(INT32_MAX, INT32_MAX);
    if (tid < NFLAGS) flags[tid] = 0u;
    if (tid < NQUEUES) flags[QUEUE_WORD + tid] = 0u;        // the tile queues of the persistent sweep kernel
    if (tid == 0) {
        *cursor = 0u;
        *reinterpret_cast<uint64_t**>(flags + HOT_PTR_WORD) = hot_rows;      // parked for the sweep kernels
    }
}

void launch_init_accumulators(hipStream_t s, uint64_t* sums, int32_t* boxes, uint64_t nlabels,
                              uint32_t* flags, uint32_t* pair_cursor, uint64_t* hot_rows) {
    uint64_t work = nlabels * NSUM / 2;
    int blocks = (int)((work + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(init_kernel, dim3(blocks), dim3(256), 0, s, sums, boxes, nlabels, flags,
                       pair_cursor, hot_rows);
}

// ------------------------------------------------------------------------------------------
// Label changes along the fast axis, per plane of axis 0: the weight a Z-slab partition balances (a record of the sweep per
// change; distributed.balanced_cuts).  One pass at streaming speed, run once per resident volume -- not part of a step.
template <typename T>
__global__ void __launch_bounds__(256) plane_events_kernel(const T* vol, int64_t n1, int64_t n2, unsigned long long* out) {
    const int64_t plane = blockIdx.y, rows_per_block = (n1 + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < n1 ? r0 + rows_per_block : n1;
    const T* p = vol + plane * n1 * n2;
    uint32_t mine = 0;
    for (int64_t i = r0 * n2 + threadIdx.x; i < r1 * n2; i += 256) {
        const int64_t c = i % n2;
        mine += (c > 0 && p[i] != p[i - 1]) ? 1u : 0u;
    }
    __shared__ uint32_t part[256];
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && part[0]) atomicAdd(out + plane, (unsigned long long)part[0]);
}

void launch_plane_events(hipStream_t s, const void* vol, int itemsize, int64_t planes, int64_t n1, int64_t n2, uint64_t* out_dev) {
    (void)hipMemsetAsync(out_dev, 0, (size_t)planes * sizeof(uint64_t), s);
    if (planes <= 0 || n1 <= 0 || n2 <= 0) return;
    const unsigned bx = (unsigned)(n1 < 16 ? n1 : 16);
    for (int64_t done = 0; done < planes; done += 32768) {          // (gridDim.y <= 65535)
        const unsigned by = (unsigned)(planes - done < 32768 ? planes - done : 32768);
        const char* base = (const char*)vol + (size_t)done * n1 * n2 * itemsize;
        if (itemsize == 2)
            hipLaunchKernelGGL(plane_events_kernel<uint16_t>, dim3(bx, by), dim3(256), 0, s, (const uint16_t*)base, n1, n2,
                               (unsigned long long*)(out_dev + done));
        else
            hipLaunchKernelGGL(plane_events_kernel<uint32_t>, dim3(bx, by), dim3(256), 0, s, (const uint32_t*)base, n1, n2,
                               (unsigned long long*)(out_dev + done));
    }
}

// ------------------------------------------------------------------------------------------
// Cross-check kernel: one thread per voxel, every contribution a global atomic.  Slow by design;
// it shares no logic with the fused sweep beyond the accumulator layout, so the two check each
// other on the GPU (and both are checked against the CPU oracle by the tests).
template <typename T>
__global__ void __launch_bounds__(256) naive_kernel(SweepArgs A, uint32_t feature_mask) {
    const int64_t plane = A.n1 * A.n2;
    const int64_t nown = (A.n0 - A.first_owned) * plane;
    const bool adj = feature_mask & 16u, mom2 = feature_mask & 8u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nown;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = i + A.first_owned * plane;
        const int64_t pa = idx / plane, rem = idx - pa * plane;
        const int64_t b = rem / A.n2, c = rem - b * A.n2;
        const uint32_t v = load_label<T>(A.vol, idx);
        const uint64_t ga = (uint64_t)(A.a_origin + pa - A.first_owned);
        if (mom2) run_add_global<true>(A, v, ga, 1u, (uint64_t)b, (uint64_t)c);
        else      run_add_global<false>(A, v, ga, 1u, (uint64_t)b, (uint64_t)c);
        if (adj) {
            if (pa > 0) {
                uint32_t u = load_label<T>(A.vol, idx - plane);
                if (u != v) pair_add_global(A.pairs, u < v ? u : v, u < v ? v : u, 1, 0, 0, A.flags);
            }
            if (b > 0) {
                uint32_t u = load_label<T>(A.vol, idx - A.n2);
                if (u != v) pair_add_global(A.pairs, u < v ? u : v, u < v ? v : u, 0, 1, 0, A.flags);
            }
            if (c > 0) {
                uint32_t u = load_label<T>(A.vol, idx - 1);
                if (u != v) pair_add_global(A.pairs, u < v ? u : v, u < v ? v : u, 0, 0, 1, A.flags);
            }
        }
    }
}

void launch_naive(hipStream_t s, const SweepArgs& a, int itemsize, uint32_t feature_mask) {
    const int64_t nown = (a.n0 - a.first_owned) * a.n1 * a.n2;
    if (nown <= 0) return;
    int64_t blocks = (nown + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (itemsize == 2)
        hipLaunchKernelGGL(naive_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a, feature_mask);
    else
        hipLaunchKernelGGL(naive_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, s, a, feature_mask);
}

// ------------------------------------------------------------------------------------------
// max label: 16-byte loads, wave max via DPP-free shuffles, one atomicMax per wave
template <typename T>
__global__ void __launch_bounds__(256) max_label_kernel(const T* vol, uint64_t nvox, uint32_t* out) {
    constexpr int VEC = 16 / sizeof(T);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    uint32_t m = 0;
    const uint64_t nvec = ((reinterpret_cast<uintptr_t>(vol) & 15) == 0) ? nvox / VEC : 0;
    const uint4* v4 = reinterpret_cast<const uint4*>(vol);
    for (uint64_t i = tid; i < nvec; i += nthreads) {
        uint4 x = v4[i];
        if (sizeof(T) == 4) {
            m = max(m, max(max(x.x, x.y), max(x.z, x.w)));
        } else {
            m = max(m, max(max(x.x & 0xffffu, x.x >> 16), max(x.y & 0xffffu, x.y >> 16)));
            m = max(m, max(max(x.z & 0xffffu, x.z >> 16), max(x.w & 0xffffu, x.w >> 16)));
        }
    }
    for (uint64_t i = nvec * VEC + tid; i < nvox; i += nthreads) m = max(m, (uint32_t)vol[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

void launch_max_label(hipStream_t s, const void* vol, int itemsize, uint64_t nvox, uint32_t* out_dev) {
    (void)hipMemsetAsync(out_dev, 0, sizeof(uint32_t), s);
    if (nvox == 0) return;
    uint64_t blocks = (nvox / (16 / itemsize) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    if (itemsize == 2)
        hipLaunchKernelGGL(max_label_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s,
                           (const uint16_t*)vol, nvox, out_dev);
    else
        hipLaunchKernelGGL(max_label_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, s,
                           (const uint32_t*)vol, nvox, out_dev);
}

// ------------------------------------------------------------------------------------------
// adjacency hash: collect (and self-clean), insert, clear
// Each block owns a contiguous slot range: count its occupied slots, reserve output space with
// ONE returning atomic per block (a single hot word sustains only ~88 atomics/us), then write.
constexpr int COLLECT_PER_THREAD = 8;
// `publish`: host-mapped mirror of the flag words (+ cursor): written by a kernel whose predecessors have fixed them, so the
// step needs no device-to-host copy (a blit kernel + a queue barrier) afterwards.
__device__ __forceinline__ void publish_small(const uint32_t* small, uint32_t* publish, int nwords, int tid) {
    if (tid < nwords) {
        const uint32_t v = __hip_atomic_load(&small[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&publish[tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__device__ __forceinline__ void pairs_collect_block(const PairTable& pt, uint64_t* out_keys, uint64_t* out_faces,
                                                    uint32_t* cursor, const uint32_t block) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t block_base;
    const uint64_t cap = (uint64_t)pt.mask + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t lo = (uint64_t)block * (256 * COLLECT_PER_THREAD);
    // the thread's keys are read ONCE, all loads in flight together (coalesced 8-byte reads), and kept in registers
    uint64_t k[COLLECT_PER_THREAD];
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < COLLECT_PER_THREAD; ++i) {
        const uint64_t h = lo + (uint64_t)i * 256 + tid;
        k[i] = h < cap ? pt.keys[h] : EMPTY_KEY;
    }
#pragma unroll
    for (int i = 0; i < COLLECT_PER_THREAD; ++i) mine += k[i] != EMPTY_KEY;
    // wave inclusive scan, then block offsets
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    if (tid == 0) {
        const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        block_base = total ? atomicAdd(cursor, total) : 0u;
    }
    __syncthreads();
    uint32_t pos = block_base + incl - mine;
    for (int i = 0; i < w; ++i) pos += wave_tot[i];
    // emit and self-clean (order inside the list does not matter: the host getter sorts)
#pragma unroll
    for (int i = 0; i < COLLECT_PER_THREAD; ++i) {
        if (k[i] == EMPTY_KEY) continue;
        const uint64_t h = lo + (uint64_t)i * 256 + tid;
        out_keys[pos] = k[i];
        out_faces[3ull * pos + 0] = pt.faces[3 * h + 0];
        out_faces[3ull * pos + 1] = pt.faces[3 * h + 1];
        out_faces[3ull * pos + 2] = pt.faces[3 * h + 2];
        ++pos;
        pt.keys[h] = EMPTY_KEY;                              // leave the table clean for the next call
        pt.faces[3 * h + 0] = 0; pt.faces[3 * h + 1] = 0; pt.faces[3 * h + 2] = 0;
    }
}

__global__ void __launch_bounds__(256) pairs_collect_kernel(PairTable pt, uint64_t* out_keys,
                                                            uint64_t* out_faces, uint32_t* cursor) {
    pairs_collect_block(pt, out_keys, out_faces, cursor, blockIdx.x);
}

void launch_pairs_collect(hipStream_t s, const PairTable& pt, uint64_t* out_keys, uint64_t* out_faces,
                          uint32_t* cursor) {
    const uint64_t cap = (uint64_t)pt.mask + 1;
    const uint64_t per_block = 256 * COLLECT_PER_THREAD;
    const uint64_t blocks = (cap + per_block - 1) / per_block;
    hipLaunchKernelGGL(pairs_collect_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pt, out_keys,
                       out_faces, cursor);
}

__global__ void __launch_bounds__(256) pairs_insert_kernel(PairTable pt, const uint64_t* keys,
                                                           const uint64_t* faces, uint64_t n,
                                                           uint32_t* flags) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        if (k == EMPTY_KEY) continue;                        // padding of gathered lists
        pair_add_global(pt, (uint32_t)(k >> 32), (uint32_t)k, faces[3 * i], faces[3 * i + 1],
                        faces[3 * i + 2], flags);
    }
}

void launch_pairs_insert(hipStream_t s, const PairTable& pt, const uint64_t* keys, const uint64_t* faces,
                         uint64_t n, uint32_t* flags) {
    if (n == 0) return;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pairs_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pt, keys, faces, n,
                       flags);
}

// ---- multi-GPU exchange: fixed-capacity blocks, no host round trip (layout in ta_device.h) ----
__global__ void __launch_bounds__(256) pairs_pack_kernel(const uint64_t* keys, const uint64_t* faces,
                                                         const uint32_t* cursor, const uint32_t* flags,
                                                         uint64_t* block, uint64_t cap) {
    const uint64_t n = *cursor;
    uint64_t* bk = block + XHDR;
    uint64_t* bf = block + XHDR + cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        block[0] = n;
        block[1] = (flags[FLAG_RANGE] ? XSTATUS_RANGE : 0) | (flags[FLAG_PAIR_OVERFLOW] ? XSTATUS_PAIR_OVERFLOW : 0);
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const bool live = i < n;
        bk[i] = live ? keys[i] : EMPTY_KEY;
        bf[3 * i + 0] = live ? faces[3 * i + 0] : 0;
        bf[3 * i + 1] = live ? faces[3 * i + 1] : 0;
        bf[3 * i + 2] = live ? faces[3 * i + 2] : 0;
    }
}

void launch_pairs_pack(hipStream_t s, const uint64_t* keys, const uint64_t* faces, const uint32_t* cursor,
                       const uint32_t* flags, uint64_t* block, uint64_t cap) {
    uint64_t blocks = (cap + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pairs_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, keys, faces, cursor, flags,
                       block, cap);
}

// Slab-exclusive labels: with the per-label boxes already reduced over all ranks, a label whose axis-0 extent lies inside
// [lo, hi - 2] can touch no other rank's planes nor its halo (the next rank's halo is plane hi - 1).  A pair with such a
// label is final on this rank: it goes straight back into the (clean) table.  Only the other pairs -- the ones a wall
// crossing a slab face can split -- travel.  A label without a box (no TA_F_BBOX) is never exclusive: everything travels.
__device__ __forceinline__ bool slab_exclusive(const int32_t* boxes, uint32_t label, uint32_t max_label, int64_t lo, int64_t hi) {
    if (label > max_label) return false;
    const int32_t mn = boxes[(uint64_t)label * NBOX + 0], negmx = boxes[(uint64_t)label * NBOX + 3];
    return mn != INT32_MAX && (int64_t)mn >= lo && -(int64_t)negmx <= hi - 2;
}

__global__ void __launch_bounds__(256) pairs_pack_shared_kernel(PairTable pt, const uint64_t* keys, const uint64_t* faces,
                                                                const uint32_t* cursor, uint32_t* flags,
                                                                const int32_t* boxes, uint32_t max_label, int64_t lo,
                                                                int64_t hi, uint64_t* block, uint64_t cap) {
    const uint64_t n = *cursor;
    uint64_t* bk = block + XHDR;
    uint64_t* bf = block + XHDR + cap;
    if (blockIdx.x == 0 && threadIdx.x == 0)        // block[0] (the count) was zeroed on the stream before this launch
        block[1] = (flags[FLAG_RANGE] ? XSTATUS_RANGE : 0) | (flags[FLAG_PAIR_OVERFLOW] ? XSTATUS_PAIR_OVERFLOW : 0);
    const int lane = threadIdx.x & 63;
    const uint64_t nround = (n + 63) & ~63ull;      // whole waves stay in the loop (ballot below)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nround;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const bool live = i < n;
        const uint64_t k = live ? keys[i] : EMPTY_KEY;
        const uint32_t a = (uint32_t)(k >> 32), b = (uint32_t)k;
        const bool valid = live && k != EMPTY_KEY;
        const bool travels = valid && !slab_exclusive(boxes, a, max_label, lo, hi) && !slab_exclusive(boxes, b, max_label, lo, hi);
        if (valid && !travels) pair_add_global(pt, a, b, faces[3 * i + 0], faces[3 * i + 1], faces[3 * i + 2], flags);
        const uint64_t m = __ballot(travels);
        if (m == 0ull) continue;
        uint64_t base = 0;
        if (lane == __ffsll((long long)m) - 1)
            base = atomicAdd((unsigned long long*)&block[0], (unsigned long long)__popcll(m));
        base = __shfl((long long)base, __ffsll((long long)m) - 1, 64);
        if (travels) {
            const uint64_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < cap) {
                bk[pos] = k;
                bf[3 * pos + 0] = faces[3 * i + 0]; bf[3 * pos + 1] = faces[3 * i + 1]; bf[3 * pos + 2] = faces[3 * i + 2];
            }
        }
    }
}

void launch_pairs_pack_shared(hipStream_t s, const PairTable& pt, const uint64_t* keys, const uint64_t* faces,
                              const uint32_t* cursor, uint32_t* flags, const int32_t* boxes, uint32_t max_label,
                              int64_t lo, int64_t hi, uint64_t* block, uint64_t cap, uint64_t npairs_bound) {
    uint64_t blocks = (npairs_bound + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pairs_pack_shared_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pt, keys, faces, cursor, flags,
                       boxes, max_label, lo, hi, block, cap);
}

__global__ void __launch_bounds__(256) pairs_insert_blocks_kernel(PairTable pt, const uint64_t* blocks,
                                                                  int nblocks, uint64_t cap, uint32_t* flags) {
    const uint64_t stride = XHDR + 4 * cap, total = (uint64_t)nblocks * cap;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = idx / cap, i = idx - b * cap;
        const uint64_t* blk = blocks + b * stride;
        const uint64_t n = blk[0];
        if (i == 0) {                                     // one thread per block carries the header over
            const uint64_t st = blk[1];
            if (st & XSTATUS_RANGE) atomicOr(&flags[FLAG_RANGE], 1u);
            if (st & XSTATUS_PAIR_OVERFLOW) atomicOr(&flags[FLAG_PAIR_OVERFLOW], 1u);
            if (n > cap) atomicOr(&flags[FLAG_EXCHANGE_OVERFLOW], 1u);
        }
        if (i >= n) continue;
        const uint64_t k = blk[XHDR + i];
        if (k == EMPTY_KEY) continue;
        const uint64_t* f = blk + XHDR + cap + 3 * i;
        pair_add_global(pt, (uint32_t)(k >> 32), (uint32_t)k, f[0], f[1], f[2], flags);
    }
}

void launch_pairs_insert_blocks(hipStream_t s, const PairTable& pt, const uint64_t* blocks, int nblocks,
                                uint64_t cap, uint32_t* flags) {
    const uint64_t total = (uint64_t)nblocks * cap;
    if (total == 0) return;
    uint64_t grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pairs_insert_blocks_kernel, dim3((unsigned)grid), dim3(256), 0, s, pt, blocks, nblocks,
                       cap, flags);
}

__global__ void __launch_bounds__(256) pairs_clear_kernel(PairTable pt) {
    const uint64_t cap = (uint64_t)pt.mask + 1;
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < cap;
         h += (uint64_t)gridDim.x * blockDim.x) {
        pt.keys[h] = EMPTY_KEY;
        pt.faces[3 * h + 0] = 0; pt.faces[3 * h + 1] = 0; pt.faces[3 * h + 2] = 0;
    }
}

void launch_pairs_clear(hipStream_t s, const PairTable& pt) {
    uint64_t cap = (uint64_t)pt.mask + 1;
    uint64_t blocks = (cap + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pairs_clear_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pt);
}

// ------------------------------------------------------------------------------------------
// Fold the per-workgroup private rows of the hot label (ta_sweep_common.h: flush_tables) into its
// global row.  Row = 128 bytes: sums u64[NSUM] | boxes i32[NBOX] | padding.
struct HotFold {                      // what the fold of the private hot-label rows needs
    const uint64_t* rows; uint32_t nrows;
    const void* vol; int itemsize; int64_t corner;
    uint64_t* sums; int32_t* boxes; uint32_t max_label;
};

__device__ __forceinline__ void hot_reduce_block(const HotFold& H, const uint32_t block, const uint32_t nblocks) {
    __shared__ uint64_t part[256];
    const uint32_t hot = H.itemsize == 2 ? (uint32_t)((const uint16_t*)H.vol)[H.corner] : ((const uint32_t*)H.vol)[H.corner];
    if (hot > H.max_label) return;                         // the sweep has raised FLAG_RANGE already
    const int col = threadIdx.x & (HOTW - 1), lanegrp = threadIdx.x / HOTW;          // 16 row-lanes per block
    const bool is_sum = col < NSUM;
    int64_t acc = is_sum ? 0 : (int64_t)INT32_MAX;
    for (uint32_t r = block * (256 / HOTW) + lanegrp; r < H.nrows; r += nblocks * (256 / HOTW)) {
        if (is_sum) acc += (int64_t)H.rows[(uint64_t)r * HOTW + col];
        else {
            const int32_t v = reinterpret_cast<const int32_t*>(H.rows + (uint64_t)r * HOTW + NSUM)[col - NSUM];
            acc = v < acc ? v : acc;
        }
    }
    part[threadIdx.x] = (uint64_t)acc;
    __syncthreads();
    if (lanegrp == 0) {
        for (int g = 1; g < 256 / HOTW; ++g) {
            const int64_t v = (int64_t)part[g * HOTW + col];
            if (is_sum) acc += v;
            else acc = v < acc ? v : acc;
        }
        if (is_sum) { if (acc) atomicAdd((unsigned long long*)&H.sums[(uint64_t)hot * NSUM + col], (unsigned long long)acc); }
        else atomicMin(&H.boxes[(uint64_t)hot * NBOX + (col - NSUM)], (int32_t)acc);
    }
}

__global__ void __launch_bounds__(256) hot_reduce_kernel(HotFold H, const uint32_t* small, uint32_t* publish, int nwords) {
    // last kernel of a step without adjacency: the sweep's flags are final, block 0 mirrors them to the host
    if (publish && blockIdx.x == 0) publish_small(small, publish, nwords, (int)threadIdx.x);
    hot_reduce_block(H, blockIdx.x, gridDim.x);
}

// With adjacency the fold rides along with the collect: one launch (the first `collect_blocks` blocks collect, the rest
// fold), one kernel boundary less per step.
__global__ void __launch_bounds__(256) pairs_collect_hot_kernel(PairTable pt, uint64_t* out_keys, uint64_t* out_faces,
                                                                uint32_t* cursor, uint32_t collect_blocks, HotFold H) {
    if (blockIdx.x < collect_blocks) pairs_collect_block(pt, out_keys, out_faces, cursor, blockIdx.x);
    else hot_reduce_block(H, blockIdx.x - collect_blocks, gridDim.x - collect_blocks);
}

static HotFold hot_fold(const SweepArgs& a, int itemsize, const uint64_t* hot_rows, uint32_t nrows) {
    HotFold H;
    H.rows = hot_rows; H.nrows = nrows; H.vol = a.vol; H.itemsize = itemsize;
    H.corner = (int64_t)a.first_owned * a.n1 * a.n2; H.sums = a.sums; H.boxes = a.boxes; H.max_label = a.max_label;
    return H;
}

static uint32_t hot_fold_blocks(uint32_t nrows) {
    uint32_t blocks = (nrows + 15) / 16;
    return blocks > 128 ? 128u : blocks;
}

void launch_pairs_collect_hot(hipStream_t s, const PairTable& pt, uint64_t* out_keys, uint64_t* out_faces, uint32_t* cursor,
                              const SweepArgs& a, int itemsize, const uint64_t* hot_rows, uint32_t nrows) {
    const uint64_t cap = (uint64_t)pt.mask + 1;
    const uint64_t per_block = 256 * COLLECT_PER_THREAD;
    const uint32_t cb = (uint32_t)((cap + per_block - 1) / per_block);
    hipLaunchKernelGGL(pairs_collect_hot_kernel, dim3(cb + hot_fold_blocks(nrows)), dim3(256), 0, s, pt, out_keys, out_faces,
                       cursor, cb, hot_fold(a, itemsize, hot_rows, nrows));
}

void launch_hot_reduce(hipStream_t s, const SweepArgs& a, int itemsize, const uint64_t* hot_rows, uint32_t nrows,
                       uint32_t* publish, int nwords) {
    if (!hot_rows || nrows == 0) return;
    hipLaunchKernelGGL(hot_reduce_kernel, dim3(hot_fold_blocks(nrows)), dim3(256), 0, s, hot_fold(a, itemsize, hot_rows, nrows),
                       a.flags, publish, nwords);
}

// ------------------------------------------------------------------------------------------
// Label lookup-table sweeps: v -> lut[v].  HBM-bound gathers: the table (4 bytes
// per label, ~200 KB at 50k labels) lives in L2 / the vector L1, neighbouring voxels mostly share a
// label so a wave's gather touches a handful of cache lines.
// relabel_kernel: in place, 16-byte loads and stores (SIA:1114-1165 fuse / remove, done per label
// with bounding-box crops in the reference).  map_kernel: to an output image of another word size
// (PSI:207-221 create_property_image), labels beyond the table get `fill`.
template <typename T>
__global__ void __launch_bounds__(256) relabel_kernel(const T* src, T* vol, uint64_t n, uint64_t nvec,
                                                      const uint32_t* __restrict__ lut, uint32_t lut_len) {
    // src == vol: in place.  src = the volume's RANK copy (a compacted context): the table has one entry per rank.
    constexpr int PER = 16 / (int)sizeof(T);
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* v4 = reinterpret_cast<uint4*>(vol);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 x = s4[i];
        uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (sizeof(T) == 4) {
                if (w[k] < lut_len) w[k] = lut[w[k]];
            } else {
                uint32_t lo = w[k] & 0xffffu, hi = w[k] >> 16;
                if (lo < lut_len) lo = lut[lo];
                if (hi < lut_len) hi = lut[hi];
                w[k] = (lo & 0xffffu) | (hi << 16);
            }
        }
        v4[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    // tail: fewer than PER voxels, or the whole volume when the buffer is not 16-byte aligned (nvec = 0)
    for (uint64_t t = nvec * PER + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = src[t];
        if (v < lut_len) vol[t] = (T)lut[v];
    }
}

void launch_relabel(hipStream_t s, const void* src, void* vol, int itemsize, uint64_t n, const uint32_t* lut, uint32_t lut_len) {
    if (n == 0) return;
    const uint64_t nvec = (((uintptr_t)vol | (uintptr_t)src) & 15) ? 0 : n / (16 / itemsize);
    uint64_t blocks = ((nvec ? nvec : n) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 16384) blocks = 16384;
    if (itemsize == 2) hipLaunchKernelGGL(relabel_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)vol, n, nvec, lut, lut_len);
    else               hipLaunchKernelGGL(relabel_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)vol, n, nvec, lut, lut_len);
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) map_kernel(const TI* __restrict__ vol, TO* __restrict__ out, uint64_t n,
                                                  const TO* __restrict__ lut, uint32_t lut_len, TO fill) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = vol[i];
        out[i] = v < lut_len ? lut[v] : fill;
    }
}

template <typename TI>
static void launch_map_t(hipStream_t s, const void* vol, void* out, int out_itemsize, uint64_t n, const void* lut,
                         uint32_t lut_len, uint64_t fill) {
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const dim3 g((unsigned)blocks), b(256);
    switch (out_itemsize) {
        case 1: hipLaunchKernelGGL((map_kernel<TI, uint8_t>), g, b, 0, s, (const TI*)vol, (uint8_t*)out, n, (const uint8_t*)lut, lut_len, (uint8_t)fill); break;
        case 2: hipLaunchKernelGGL((map_kernel<TI, uint16_t>), g, b, 0, s, (const TI*)vol, (uint16_t*)out, n, (const uint16_t*)lut, lut_len, (uint16_t)fill); break;
        case 4: hipLaunchKernelGGL((map_kernel<TI, uint32_t>), g, b, 0, s, (const TI*)vol, (uint32_t*)out, n, (const uint32_t*)lut, lut_len, (uint32_t)fill); break;
        default: hipLaunchKernelGGL((map_kernel<TI, uint64_t>), g, b, 0, s, (const TI*)vol, (uint64_t*)out, n, (const uint64_t*)lut, lut_len, (uint64_t)fill); break;
    }
}

// ------------------------------------------------------------------------------------------
// Read-bandwidth probe ("measure the box's achievable read-only streaming bandwidth with a trivial
// reduction kernel"): 16-byte loads, four in flight per lane, an XOR reduction nobody needs; the sink store can
// never happen (the data are labels < 2^32, never all-ones four times over) but keeps the loads alive.
__global__ void __launch_bounds__(256) read_probe_kernel(const uint4* __restrict__ p, uint64_t n16, uint32_t* sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0u, 0u, 0u, 0u);
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y;
        acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = p[i]; acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w; }
    if ((acc.x & acc.y & acc.z & acc.w) == 0xFFFFFFFFu && (acc.x ^ acc.y) == 0x12345678u) *sink = acc.x;
}

void launch_read_probe(hipStream_t s, const void* p, uint64_t bytes, uint32_t* sink) {
    const uint64_t n16 = bytes / 16;
    if (n16 == 0) return;
    uint64_t blocks = (n16 + 255) / 256;
    if (blocks > 256ull * 32) blocks = 256ull * 32;          // 32 workgroups per CU: a persistent grid-stride sweep
    hipLaunchKernelGGL(read_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4*)p, n16, sink);
}

// ------------------------------------------------------------------------------------------
// First voxel layer (SIA:1024-1046): a voxel that is not background and has a background voxel among
// its six face neighbours keeps its label, every other tissue voxel becomes 0, background becomes 1
// (keep_background) or 0 -- `image * (dilate6(image == bg) - (image == bg)) + (image == bg)`, one stencil pass.
// One lane = VEC consecutive voxels of a row (16 bytes): the row itself plus the four neighbouring rows come as
// 16-byte loads (L1/L2 hits for all but one of them), the two voxels across the strip's ends as scalars.
template <typename T>
__global__ void __launch_bounds__(256) first_layer_kernel(const T* __restrict__ vol, T* __restrict__ out, int64_t n0,
                                                          int64_t n1, int64_t n2, uint32_t bg, int keep_bg) {
    constexpr int VEC = 16 / sizeof(T);
    const int64_t strips = (n2 + VEC - 1) / VEC, total = n0 * n1 * strips;
    const bool vec_ok = (n2 % VEC) == 0 && ((reinterpret_cast<uintptr_t>(vol) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i % strips, row = i / strips, b = row % n1, a = row / n1, c0 = s * VEC;
        const T* r = vol + row * n2;
        T v[VEC], up[VEC], dn[VEC], pv[VEC], nx[VEC];
        auto load = [&](const T* rp, bool ok, T (&d)[VEC]) {
            if (!ok) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) d[j] = (T)0;
            } else if (vec_ok) {
                *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(rp + c0);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) d[j] = c0 + j < n2 ? rp[c0 + j] : (T)0;
            }
        };
        load(r, true, v);
        load(r - n2, b > 0, up); load(r + n2, b + 1 < n1, dn);
        load(r - n1 * n2, a > 0, pv); load(r + n1 * n2, a + 1 < n0, nx);
        const bool lb = c0 > 0 && (uint32_t)r[c0 - 1] == bg, rb = c0 + VEC < n2 && (uint32_t)r[c0 + VEC] == bg;
        T o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool self_bg = (uint32_t)v[j] == bg;
            bool near = (b > 0 && (uint32_t)up[j] == bg) || (b + 1 < n1 && (uint32_t)dn[j] == bg) ||
                        (a > 0 && (uint32_t)pv[j] == bg) || (a + 1 < n0 && (uint32_t)nx[j] == bg);
            near = near || (j > 0 ? (uint32_t)v[j > 0 ? j - 1 : 0] == bg : lb);
            near = near || (j + 1 < VEC ? (c0 + j + 1 < n2 && (uint32_t)v[j + 1 < VEC ? j + 1 : 0] == bg) : rb);
            o[j] = self_bg ? (T)(keep_bg ? 1 : 0) : (near ? v[j] : (T)0);
        }
        if (vec_ok) {
            *reinterpret_cast<uint4*>(out + row * n2 + c0) = *reinterpret_cast<const uint4*>(o);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) if (c0 + j < n2) out[row * n2 + c0 + j] = o[j];
        }
    }
}

void launch_first_layer(hipStream_t s, const void* vol, int itemsize, void* out, int64_t n0, int64_t n1, int64_t n2,
                        uint32_t background, int keep_background) {
    const int64_t strips = (n2 + 16 / itemsize - 1) / (16 / itemsize), total = n0 * n1 * strips;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (itemsize == 2)
        hipLaunchKernelGGL(first_layer_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint16_t*)vol,
                           (uint16_t*)out, n0, n1, n2, background, keep_background);
    else
        hipLaunchKernelGGL(first_layer_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t*)vol,
                           (uint32_t*)out, n0, n1, n2, background, keep_background);
}

// ------------------------------------------------------------------------------------------
// hollow_out_cells (SIA:74-95): `image * (laplace(image) != 0)`, then `* (that != background)`.  scipy's laplace keeps
// the image's integer type: each axis' v[-1] - 2 v + v[+1] is cast to it and the three are added in it, i.e. the whole
// is (sum of the six face neighbours - 6 v) modulo 2^(bits of that type), with the edge voxel repeated outside the image
// (mode 'reflect').  A wall voxel whose neighbours happen to cancel (v - 1 on one side, v + 1 on the other) is NOT kept,
// like in the reference.  Same layout of the work as first_layer_kernel.
template <typename T>
__global__ void __launch_bounds__(256) hollow_kernel(const T* __restrict__ vol, T* __restrict__ out, int64_t n0, int64_t n1,
                                                     int64_t n2, uint32_t bg, int remove_bg, int drop_bits) {
    constexpr int VEC = 16 / sizeof(T);
    const int64_t strips = (n2 + VEC - 1) / VEC, total = n0 * n1 * strips;
    const bool vec_ok = (n2 % VEC) == 0 && ((reinterpret_cast<uintptr_t>(vol) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i % strips, row = i / strips, b = row % n1, a = row / n1, c0 = s * VEC;
        const T* r = vol + row * n2;
        T v[VEC], up[VEC], dn[VEC], pv[VEC], nx[VEC];
        auto load = [&](const T* rp, T (&d)[VEC]) {
            if (vec_ok) {
                *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(rp + c0);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) d[j] = rp[c0 + j < n2 ? c0 + j : n2 - 1];
            }
        };
        load(r, v);
        load(b > 0 ? r - n2 : r, up); load(b + 1 < n1 ? r + n2 : r, dn);
        load(a > 0 ? r - n1 * n2 : r, pv); load(a + 1 < n0 ? r + n1 * n2 : r, nx);
        const T left = r[c0 > 0 ? c0 - 1 : 0], right = r[c0 + VEC < n2 ? c0 + VEC : n2 - 1];
        T o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const uint32_t l = j > 0 ? (uint32_t)v[j > 0 ? j - 1 : 0] : (uint32_t)left;
            const uint32_t rr = j + 1 < VEC ? (uint32_t)(c0 + j + 1 < n2 ? v[j + 1 < VEC ? j + 1 : 0] : v[j]) : (uint32_t)right;
            // exact in 64 bits, then modulo 2^(64 - drop_bits): the width of the image the caller holds
            const uint64_t sum = (uint64_t)up[j] + (uint64_t)dn[j] + (uint64_t)pv[j] + (uint64_t)nx[j] + l + rr - 6ull * (uint64_t)v[j];
            const bool keep = (sum << drop_bits) != 0ull && !(remove_bg && (uint32_t)v[j] == bg);
            o[j] = keep ? v[j] : (T)0;
        }
        if (vec_ok) {
            *reinterpret_cast<uint4*>(out + row * n2 + c0) = *reinterpret_cast<const uint4*>(o);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) if (c0 + j < n2) out[row * n2 + c0 + j] = o[j];
        }
    }
}

void launch_hollow(hipStream_t s, const void* vol, int itemsize, void* out, int64_t n0, int64_t n1, int64_t n2, uint32_t background,
                   int remove_background, int label_bits) {
    const int drop_bits = 64 - label_bits;
    const int64_t strips = (n2 + 16 / itemsize - 1) / (16 / itemsize), total = n0 * n1 * strips;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (itemsize == 2)
        hipLaunchKernelGGL(hollow_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint16_t*)vol, (uint16_t*)out, n0, n1,
                           n2, background, remove_background, drop_bits);
    else
        hipLaunchKernelGGL(hollow_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t*)vol, (uint32_t*)out, n0, n1,
                           n2, background, remove_background, drop_bits);
}

// ------------------------------------------------------------------------------------------
// The voxel layer of every cell at once (cells_voxel_layer, SIA:1399-1448: `mask - binary_erosion(mask, 18-structure)`
// per label and crop): out[p] = 1 when one of the 18 neighbours of p (faces + edges, inside the image) carries another
// label, else 0.  The erosion of one label's mask inside a crop removes exactly these voxels plus the label's voxels on
// the faces of the crop, which the host adds.  One lane = VEC consecutive voxels of a row; the 3 x 3 rows around it come
// as 16-byte loads, clamped at the image faces (a clamped row repeats a neighbour or the row itself: no new label).
template <typename T>
__global__ void __launch_bounds__(256) layer18_kernel(const T* __restrict__ vol, uint8_t* __restrict__ out, int64_t n0, int64_t n1,
                                                      int64_t n2) {
    constexpr int VEC = 16 / sizeof(T);
    const int64_t strips = (n2 + VEC - 1) / VEC, total = n0 * n1 * strips;
    const bool vec_ok = (n2 % VEC) == 0 && (reinterpret_cast<uintptr_t>(vol) & 15) == 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i % strips, row = i / strips, b = row % n1, a = row / n1, c0 = s * VEC;
        const int64_t cl = c0 > 0 ? c0 - 1 : 0, cr = c0 + VEC < n2 ? c0 + VEC : n2 - 1;
        T v[VEC];
        uint32_t other[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) other[j] = 0;
        auto rowp = [&](int da, int db) {
            int64_t aa = a + da, bb = b + db;
            aa = aa < 0 ? 0 : (aa >= n0 ? n0 - 1 : aa);
            bb = bb < 0 ? 0 : (bb >= n1 ? n1 - 1 : bb);
            return vol + (aa * n1 + bb) * n2;
        };
        auto load = [&](const T* rp, T (&d)[VEC]) {
            if (vec_ok) {
                *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(rp + c0);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) d[j] = rp[c0 + j < n2 ? c0 + j : n2 - 1];
            }
        };
        load(rowp(0, 0), v);
#pragma unroll
        for (int da = -1; da <= 1; ++da)
#pragma unroll
            for (int db = -1; db <= 1; ++db) {
                const bool face = da == 0 || db == 0;         // the row itself and its four face rows: column neighbours count
                const T* rp = rowp(da, db);
                T d[VEC];
                load(rp, d);
                if (face) {
                    const T left = rp[cl], right = rp[cr];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const T l = j > 0 ? d[j > 0 ? j - 1 : 0] : left;
                        const T r = j + 1 < VEC ? (c0 + j + 1 < n2 ? d[j + 1 < VEC ? j + 1 : 0] : d[j]) : right;
                        other[j] |= (uint32_t)(l ^ v[j]) | (uint32_t)(r ^ v[j]) | (uint32_t)(d[j] ^ v[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) other[j] |= (uint32_t)(d[j] ^ v[j]);
                }
            }
        uint8_t o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = other[j] ? 1 : 0;
        uint8_t* op = out + row * n2 + c0;
        if (vec_ok && VEC == 8) {
            *reinterpret_cast<uint2*>(op) = *reinterpret_cast<const uint2*>(o);
        } else if (vec_ok && VEC == 4) {
            *reinterpret_cast<uint32_t*>(op) = *reinterpret_cast<const uint32_t*>(o);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) if (c0 + j < n2) op[j] = o[j];
        }
    }
}

void launch_layer18(hipStream_t s, const void* vol, int itemsize, uint8_t* out, int64_t n0, int64_t n1, int64_t n2) {
    const int64_t strips = (n2 + 16 / itemsize - 1) / (16 / itemsize), total = n0 * n1 * strips;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (itemsize == 2)
        hipLaunchKernelGGL(layer18_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint16_t*)vol, out, n0, n1, n2);
    else
        hipLaunchKernelGGL(layer18_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t*)vol, out, n0, n1, n2);
}

void launch_map(hipStream_t s, const void* vol, int itemsize, void* out, int out_itemsize, uint64_t n,
                const void* lut, uint32_t lut_len, uint64_t fill) {
    if (n == 0) return;
    if (itemsize == 2) launch_map_t<uint16_t>(s, vol, out, out_itemsize, n, lut, lut_len, fill);
    else               launch_map_t<uint32_t>(s, vol, out, out_itemsize, n, lut, lut_len, fill);
}

// ------------------------------------------------------------------------------------------
// Synthetic jittered-grid Voronoi tissue (tissue_analysis_amd/synth.py is the definition).
template <typename T>
__global__ void __launch_bounds__(256) synth_kernel(T* out, int64_t d0, int64_t d1, int64_t d2,
                                                    int64_t a_begin, int64_t a_count,
                                                    const int32_t* seeds, int g0, int g1, int g2,
                                                    const int64_t* ell) {
    const int64_t n = a_count * d1 * d2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pa = i / (d1 * d2), rem = i - pa * d1 * d2;
        const int x0 = (int)(a_begin + pa), x1 = (int)(rem / d2), x2 = (int)(rem - (rem / d2) * d2);
        uint32_t label;
        bool inside = true;
        if (ell) inside = (ell[x0] + ell[d0 + x1] + ell[d0 + d1 + x2]) <= (1ll << 24);
        if (!inside) {
            label = 1u;
        } else {
            const int i0 = (int)(((int64_t)x0 * g0) / d0), i1 = (int)(((int64_t)x1 * g1) / d1),
                      i2 = (int)(((int64_t)x2 * g2) / d2);
            int64_t best_d = INT64_MAX;
            uint32_t best_l = 0xFFFFFFFFu;
            for (int j0 = max(i0 - 2, 0); j0 <= min(i0 + 2, g0 - 1); ++j0)
                for (int j1 = max(i1 - 2, 0); j1 <= min(i1 + 2, g1 - 1); ++j1)
                    for (int j2 = max(i2 - 2, 0); j2 <= min(i2 + 2, g2 - 1); ++j2) {
                        const int cell = (j0 * g1 + j1) * g2 + j2;
                        const int64_t e0 = x0 - seeds[3 * cell + 0], e1 = x1 - seeds[3 * cell + 1],
                                      e2 = x2 - seeds[3 * cell + 2];
                        const int64_t d = e0 * e0 + e1 * e1 + e2 * e2;
                        const uint32_t l = (uint32_t)cell + 2u;
                        if (d < best_d || (d == best_d && l < best_l)) { best_d = d; best_l = l; }
                    }
            label = best_l;
        }
        out[i] = (T)label;
    }
}

void launch_synth(hipStream_t s, void* out, int itemsize, const int64_t dims[3], int64_t a_begin,
                  int64_t a_count, const int32_t* seeds_dev, const int32_t grid[3],
                  const int64_t* ell_dev) {
    const int64_t n = a_count * dims[1] * dims[2];
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (itemsize == 2)
        hipLaunchKernelGGL(synth_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s,
                           (uint16_t*)out, dims[0], dims[1], dims[2], a_begin, a_count, seeds_dev,
                           grid[0], grid[1], grid[2], ell_dev);
    else
        hipLaunchKernelGGL(synth_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, s,
                           (uint32_t*)out, dims[0], dims[1], dims[2], a_begin, a_count, seeds_dev,
                           grid[0], grid[1], grid[2], ell_dev);
}

}  // namespace ta
