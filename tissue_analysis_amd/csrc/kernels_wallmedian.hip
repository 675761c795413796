// kernels_wallmedian.hip -- the median voxel of every wall ON THE DEVICE (TGI:210-242 through SIA:1586-1635).
//
// `_graph_from_image(..., 'wall_median')` wants ONE voxel per wall: the geometric median of the wall's voxels by the
// reference's Weiszfeld rules, truncated, then the wall voxel nearest to it.  Rounds 2-3 shipped ~20 bytes per wall-voxel
// record (300 MB on C2) to pageable host memory for that; here the records stay where the wall kernels left them -- grouped
// by pair -- and E x 3 integers come back.
//
// The arithmetic is the host routine's (tissue_analysis_amd/geometry.py::weiszfeld_segments, which restates the reference's),
// operation for operation in IEEE double: sums taken point by point in record order (the truncation that follows makes the
// last bit count on symmetric walls, whose median sits ON an integer), no fused multiply-add anywhere (numpy's ufuncs multiply
// and add separately), correctly rounded sqrt and division.  The centroid is a sum of integers (exact in any order) divided
// once.  One wave per wall: the terms of 64 voxels at a time in parallel, the sums lane-sequential.
#include "ta_kernels.h"

#pragma clang fp contract(off)

namespace ta {

// flags[i] = 1 where record i opens a wall (the records are grouped by pair)
__global__ void __launch_bounds__(256) wall_open_flags_kernel(const uint2* pairs, uint64_t n, uint32_t* flags) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t f = 1u;
        if (i > 0) { const uint2 a = pairs[i], b = pairs[i - 1]; f = (a.x != b.x || a.y != b.y) ? 1u : 0u; }
        flags[i] = f;
    }
}

// starts[rank of the wall] = its first record
__global__ void __launch_bounds__(256) wall_starts_kernel(const uint32_t* flags, const uint64_t* rank, uint64_t n, uint32_t* starts) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (flags[i]) starts[rank[i]] = (uint32_t)i;
}

// One WAVE per wall.  The lanes take 64 consecutive voxels of the wall at a time and compute their terms (a square root and
// five divisions in double each -- what the time goes into); the SUMS stay sequential: lanes 0..4 each add one of the five
// terms, voxel after voxel, out of LDS -- ((acc + t_0) + t_1) + ... in record order, exactly like the host routine.
__global__ void __launch_bounds__(64) wall_median_kernel(const uint2* pairs, const int32_t* coords, const uint32_t* starts, uint32_t nwalls,
                                                         uint64_t n, int max_iter, uint2* out_pairs, uint32_t* out_sizes, int32_t* out_medians,
                                                         uint32_t* status) {
    const uint32_t w = blockIdx.x;
    const int lane = threadIdx.x;
    __shared__ double T[5][64];
    const uint64_t s = starts[w], e = w + 1 < nwalls ? (uint64_t)starts[w + 1] : n;
    const int32_t* P = coords + 3 * s;
    const uint64_t m = e - s;
    // centroid: np.mean of integer coordinates -- an exact sum (any order), one division
    long long si[3] = {0, 0, 0};
    for (uint64_t i = lane; i < m; i += 64) { si[0] += P[3 * i]; si[1] += P[3 * i + 1]; si[2] += P[3 * i + 2]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        si[0] += __shfl_xor(si[0], off); si[1] += __shfl_xor(si[1], off); si[2] += __shfl_xor(si[2], off);
    }
    double y[3] = {(double)si[0] / (double)m, (double)si[1] / (double)m, (double)si[2] / (double)m};
    // rule (1): nudged by +0.1 on all axes while, on EVERY axis, its coordinate is some sample's coordinate on that axis
    for (;;) {
        bool on0 = false, on1 = false, on2 = false;
        for (uint64_t i = lane; i < m; i += 64) {
            on0 = on0 || (double)P[3 * i] == y[0];
            on1 = on1 || (double)P[3 * i + 1] == y[1];
            on2 = on2 || (double)P[3 * i + 2] == y[2];
        }
        if (!(__builtin_amdgcn_ballot_w64(on0) && __builtin_amdgcn_ballot_w64(on1) && __builtin_amdgcn_ballot_w64(on2))) break;
        y[0] += 0.1; y[1] += 0.1; y[2] += 0.1;
    }
    double cost_1 = 0.0, cost_2 = 0.0;
    bool stopped = false;
    for (int it = 0; it < max_iter; ++it) {
        double acc = 0.0;                                        // lane j < 5: the running sum of term j
        for (uint64_t base = 0; base < m; base += 64) {
            const uint64_t i = base + (uint64_t)lane;
            if (i < m) {
                const double p0 = (double)P[3 * i], p1 = (double)P[3 * i + 1], p2 = (double)P[3 * i + 2];
                const double d0 = p0 - y[0], d1 = p1 - y[1], d2 = p2 - y[2];
                const double dist = sqrt((d0 * d0 + d1 * d1) + d2 * d2);
                T[0][lane] = p0 / dist; T[1][lane] = p1 / dist; T[2][lane] = p2 / dist;
                T[3][lane] = 1.0 / dist;
                T[4][lane] = dist * dist;
            }
            __syncthreads();
            const int cnt = (int)(m - base < 64 ? m - base : 64);
            if (lane < 5)
                for (int k = 0; k < cnt; ++k) acc = acc + T[lane][k];
            __syncthreads();
        }
        const double wsum = __shfl(acc, 3), cost = __shfl(acc, 4);
        const bool dead = wsum == 0.0;
        const double a0 = __shfl(acc, 0), a1 = __shfl(acc, 1), a2 = __shfl(acc, 2);
        y[0] = dead ? 0.0 : a0 / wsum; y[1] = dead ? 0.0 : a1 / wsum; y[2] = dead ? 0.0 : a2 / wsum;
        // rule (2): from the fifth pass on, stop when the cost differs by less than 0.1 from its value TWO passes back;
        // rule (3), as written: settling on the very last pass still counts as failure
        bool stop = dead || (it > 3 && fabs(cost - cost_2) < 0.1);
        if (it == max_iter - 1) stop = dead;
        cost_2 = cost_1; cost_1 = cost;
        if (stop) { stopped = true; break; }
    }
    if (!stopped && lane == 0) atomicAdd(status, 1u);        // still moving after max_iter passes: counted, and marked in its size word
    // the wall voxel nearest to the truncated position, the first one on ties (a position that is not a number: the first voxel)
    const double o0 = trunc(y[0]), o1 = trunc(y[1]), o2 = trunc(y[2]);
    double best = INFINITY;
    uint64_t at = ~0ull;
    for (uint64_t i = lane; i < m; i += 64) {
        const double d0 = (double)P[3 * i] - o0, d1 = (double)P[3 * i + 1] - o1, d2 = (double)P[3 * i + 2] - o2;
        double sq = (d0 * d0 + d1 * d1) + d2 * d2;
        if (sq != sq) sq = INFINITY;
        if (sq < best || (sq == best && i < at)) { best = sq; at = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const uint64_t oa = __shfl_xor((unsigned long long)at, off);
        if (ob < best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if (lane == 0) {
        out_pairs[w] = pairs[s];
        out_sizes[w] = (uint32_t)m | (stopped ? 0u : 0x80000000u);      // (bit 31: this wall's iteration did not settle)
        out_medians[3 * w] = P[3 * at]; out_medians[3 * w + 1] = P[3 * at + 1]; out_medians[3 * w + 2] = P[3 * at + 2];
    }
}

uint64_t wall_median_scratch_bytes(uint64_t n) { return ((n * 4 + 15) & ~15ull) + ((n * 8 + 15) & ~15ull) + scan_u32_scratch_bytes(n) + 64; }

// pairs / coords: the n records grouped by pair (device).  scratch: wall_median_scratch_bytes(n).  starts: room for n entries.
// Enqueues the flag + rank passes and leaves the number of walls in *nwalls_dev (uint64, device): the caller reads it back and
// then calls launch_wall_medians.
void launch_wall_starts(hipStream_t s, const uint32_t* pairs, uint64_t n, void* scratch, uint32_t* starts, uint64_t** nwalls_dev) {
    char* p = (char*)scratch;
    uint32_t* flags = (uint32_t*)p; p += (n * 4 + 15) & ~15ull;
    uint64_t* rank = (uint64_t*)p; p += (n * 8 + 15) & ~15ull;
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wall_open_flags_kernel, dim3(blocks), dim3(256), 0, s, (const uint2*)pairs, n, flags);
    launch_scan_u32_exclusive(s, flags, n, p, rank);
    hipLaunchKernelGGL(wall_starts_kernel, dim3(blocks), dim3(256), 0, s, flags, rank, n, starts);
    *nwalls_dev = scan_u32_total(p, n);              // (the scan leaves its total behind its block sums)
}

void launch_wall_medians(hipStream_t s, const uint32_t* pairs, const int32_t* coords, const uint32_t* starts, uint32_t nwalls, uint64_t n,
                         int max_iter, uint32_t* out_pairs, uint32_t* out_sizes, int32_t* out_medians, uint32_t* status) {
    if (nwalls == 0) return;
    hipLaunchKernelGGL(wall_median_kernel, dim3(nwalls), dim3(64), 0, s, (const uint2*)pairs, coords, starts, nwalls, n, max_iter,
                       (uint2*)out_pairs, out_sizes, out_medians, status);
}

}  // namespace ta
