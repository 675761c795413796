// kernels_pairsort.hip -- the adjacency list sorted by (lo, hi) for the host getter (ta_adjacency_get), hand-written.
//
// The unique pairs come out of the device-global hash in table order.  Labels are dense row indices, so "sorted by lo" is a
// counting sort over the label rows -- count the pairs of each lo, scan, scatter into the buckets -- and a bucket (the pairs
// of one label with its larger neighbours: a handful; a few thousand for the background) is put in order by RANK: hi is
// unique inside a bucket, so the position of a pair is the number of pairs of its bucket with a smaller hi.  Every thread
// of the rank kernel walks its own bucket; the threads of a wave mostly share one, so the walk is a broadcast read.
// Seven small launches instead of the eight radix passes of a 64-bit library sort (round 2: hipCUB DeviceRadixSort).
#include "ta_kernels.h"

namespace ta {

// One label -- the background -- holds thousands of pairs as their lo: thousands of atomics on ONE counter serialise in one
// L2 channel (~90 per microsecond).  `hot` (the label of the volume's first voxel, like the sweep's private hot rows; any
// value is correct) is therefore counted per BLOCK in LDS and added with one atomic per block.  One pair per thread.
__device__ __forceinline__ uint32_t pairsort_hot(const void* vol, int itemsize, int64_t corner) {
    if (!vol) return 0xFFFFFFFFu;
    return itemsize == 2 ? (uint32_t)((const uint16_t*)vol)[corner] : ((const uint32_t*)vol)[corner];
}

__global__ void __launch_bounds__(256) pairsort_count_kernel(const uint64_t* keys, uint64_t n, uint32_t* counts, uint32_t max_label,
                                                             const void* vol, int itemsize, int64_t corner) {
    __shared__ uint32_t hot_here;
    const uint32_t hot = pairsort_hot(vol, itemsize, corner);
    if (threadIdx.x == 0) hot_here = 0u;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const uint32_t lo = (uint32_t)(keys[i] >> 32), b = lo <= max_label ? lo : max_label + 1u;      // (a foreign key above max_label: last bucket)
        if (b == hot) atomicAdd(&hot_here, 1u);
        else atomicAdd(&counts[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0 && hot_here) atomicAdd(&counts[hot], hot_here);
}

// bucket order is whatever the atomics give; `cursor` = a copy of the counts, counted down
__global__ void __launch_bounds__(256) pairsort_scatter_kernel(const uint64_t* keys, const uint64_t* faces, uint64_t n,
                                                               const uint64_t* offsets, uint32_t* cursor, uint32_t max_label,
                                                               uint64_t* keys_tmp, uint64_t* faces_tmp, const void* vol, int itemsize,
                                                               int64_t corner) {
    __shared__ uint32_t hot_here, hot_base;
    const uint32_t hot = pairsort_hot(vol, itemsize, corner);
    if (threadIdx.x == 0) hot_here = 0u;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t k = i < n ? keys[i] : 0ull;
    const uint32_t lo = (uint32_t)(k >> 32), b = lo <= max_label ? lo : max_label + 1u;
    const bool is_hot = i < n && b == hot;
    const uint32_t mine = is_hot ? atomicAdd(&hot_here, 1u) : 0u;        // this block's hot pairs get consecutive places
    __syncthreads();
    if (threadIdx.x == 0 && hot_here) hot_base = atomicSub(&cursor[hot], hot_here) - hot_here;
    __syncthreads();
    if (i < n) {
        const uint64_t pos = offsets[b] + (uint64_t)(is_hot ? hot_base + mine : atomicSub(&cursor[b], 1u) - 1u);
        keys_tmp[pos] = k;
        faces_tmp[3 * pos + 0] = faces[3 * i + 0]; faces_tmp[3 * pos + 1] = faces[3 * i + 1]; faces_tmp[3 * pos + 2] = faces[3 * i + 2];
    }
}

// One thread per pair of the bucketed list (block t covers positions 256 t ..): the rank of a pair = the pairs of its bucket
// with a smaller key.  A big bucket (the background's: thousands of pairs) is walked through LDS, a tile of 256 keys at a
// time, by every block that holds a piece of it (each compare a broadcast read); small buckets by their own threads.
__global__ void __launch_bounds__(256) pairsort_rank_kernel(const uint64_t* keys_tmp, const uint64_t* faces_tmp, uint64_t n,
                                                            const uint64_t* offsets, uint32_t max_label, uint64_t* keys_out,
                                                            uint64_t* faces_out) {
    __shared__ uint64_t tile[256];
    __shared__ uint32_t ends[2];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const uint64_t k = valid ? keys_tmp[i] : ~0ull;
    const uint32_t lo = (uint32_t)(k >> 32), b = lo <= max_label ? lo : max_label + 1u;
    const uint64_t last_i = (uint64_t)blockIdx.x * 256 + 255 < n ? (uint64_t)blockIdx.x * 256 + 255 : n - 1;
    if (threadIdx.x == 0) ends[0] = b;
    if (i == last_i) ends[1] = b;
    __syncthreads();
    const uint64_t first = valid ? offsets[b] : 0, last = valid ? offsets[b + 1u] : 0;
    uint64_t rank = 0;
    // the buckets at the two ends of the block may reach far outside it (the list is grouped by bucket): those are walked
    // by the whole block, through LDS; a bucket in between lies inside the block (< 256 pairs): its threads walk it themselves
    for (int e = 0; e < 2; ++e) {
        if (e == 1 && ends[1] == ends[0]) break;
        const uint32_t be = ends[e];
        const uint64_t f0 = offsets[be], l0 = offsets[be + 1u];
        for (uint64_t j0 = f0; j0 < l0; j0 += 256) {
            tile[threadIdx.x] = j0 + threadIdx.x < l0 ? keys_tmp[j0 + threadIdx.x] : ~0ull;
            __syncthreads();
            if (valid && b == be) {
#pragma unroll 16
                for (int u = 0; u < 256; ++u) rank += tile[u] < k ? 1u : 0u;
            }
            __syncthreads();
        }
    }
    if (valid && b != ends[0] && b != ends[1])
        for (uint64_t j = first; j < last; ++j) rank += keys_tmp[j] < k ? 1u : 0u;
    if (valid) {
        const uint64_t pos = first + rank;
        keys_out[pos] = k;
        faces_out[3 * pos + 0] = faces_tmp[3 * i + 0]; faces_out[3 * pos + 1] = faces_tmp[3 * i + 1]; faces_out[3 * pos + 2] = faces_tmp[3 * i + 2];
    }
}

uint64_t pairs_sort_scratch_bytes(uint64_t n, uint32_t max_label) {
    const uint64_t rows = (uint64_t)max_label + 3;                       // buckets 0 .. max_label + 1, one more offset behind them
    return 2 * ((rows * 4 + 15) & ~15ull) + ((rows * 8 + 15) & ~15ull) + scan_u32_scratch_bytes(rows) + n * 8 + n * 24 + 64;
}

// keys / faces: the n unique pairs in any order.  keys_out / faces_out: the same pairs sorted by (lo, hi).  scratch:
// pairs_sort_scratch_bytes(n, max_label) bytes of device memory; vol / itemsize / corner: where the label that holds the most
// pairs can be read (the volume's first owned voxel), or vol = NULL.  Only enqueues work.
hipError_t launch_pairs_sort(hipStream_t s, const uint64_t* keys, const uint64_t* faces, uint64_t n, uint32_t max_label, void* scratch,
                             uint64_t* keys_out, uint64_t* faces_out, const void* vol, int itemsize, int64_t corner) {
    if (n == 0) return hipSuccess;
    const uint64_t rows = (uint64_t)max_label + 3;
    char* p = (char*)scratch;
    uint32_t* counts = (uint32_t*)p; p += (rows * 4 + 15) & ~15ull;
    uint32_t* cursor = (uint32_t*)p; p += (rows * 4 + 15) & ~15ull;
    uint64_t* offsets = (uint64_t*)p; p += (rows * 8 + 15) & ~15ull;
    void* scan_scratch = p; p += scan_u32_scratch_bytes(rows);
    uint64_t* keys_tmp = (uint64_t*)p; p += n * 8;
    uint64_t* faces_tmp = (uint64_t*)p;
    hipError_t e = hipMemsetAsync(counts, 0, rows * 4, s);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(pairsort_count_kernel, dim3(blocks), dim3(256), 0, s, keys, n, counts, max_label, vol, itemsize, corner);
    e = hipMemcpyAsync(cursor, counts, rows * 4, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    launch_scan_u32_exclusive(s, counts, rows, scan_scratch, offsets);        // offsets[b] = pairs in the buckets before b; offsets[rows - 1] = n
    hipLaunchKernelGGL(pairsort_scatter_kernel, dim3(blocks), dim3(256), 0, s, keys, faces, n, offsets, cursor, max_label, keys_tmp, faces_tmp, vol, itemsize, corner);
    hipLaunchKernelGGL(pairsort_rank_kernel, dim3(blocks), dim3(256), 0, s, keys_tmp, faces_tmp, n, offsets, max_label, keys_out, faces_out);
    return hipGetLastError();
}

}  // namespace ta
