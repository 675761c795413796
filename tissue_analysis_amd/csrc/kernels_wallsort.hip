// kernels_wallsort.hip -- the wall-voxel records grouped by label pair ON THE DEVICE (what every caller of the table
// wants: WallTable used to do a stable host sort of ~10^7 records).  A stable LSD radix sort (hipCUB / rocPRIM, a plain
// library sort) of key = lo << 32 | hi with the record index as value keeps the memory order inside each pair; a gather
// then writes the records in that order.  Kept in its own file: the library headers are slow to compile.
#include "ta_kernels.h"

#include <hipcub/hipcub.hpp>

namespace ta {

__global__ void __launch_bounds__(256) wall_sort_keys_kernel(const uint2* pairs, uint64_t n, uint64_t* keys, uint32_t* index) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 p = pairs[i];
        keys[i] = ((uint64_t)p.x << 32) | p.y;
        index[i] = (uint32_t)i;
    }
}

struct __attribute__((packed, aligned(4))) WallInt3 { int32_t x, y, z; };

__global__ void __launch_bounds__(256) wall_gather_kernel(const uint64_t* keys, const uint32_t* index, const WallInt3* coords,
                                                          uint64_t n, uint2* pairs_out, WallInt3* coords_out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        pairs_out[i] = make_uint2((uint32_t)(k >> 32), (uint32_t)k);
        coords_out[i] = coords[index[i]];
    }
}

uint64_t wall_sort_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    hipcub::DoubleBuffer<uint64_t> dk(nullptr, nullptr);
    hipcub::DoubleBuffer<uint32_t> dv(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dk, dv, (int64_t)n, 0, 64, nullptr);
    return (uint64_t)bytes;
}

// pairs / coords: the records in memory order (n < 2^32).  keys[2], index[2]: double buffers of n entries; temp: wall_sort_temp_bytes.
// Writes the grouped records to pairs_out / coords_out.  Only enqueues work.
hipError_t launch_wall_group_by_pair(hipStream_t s, const uint32_t* pairs, const int32_t* coords, uint64_t n, uint64_t* keys0,
                                     uint64_t* keys1, uint32_t* index0, uint32_t* index1, void* temp, uint64_t temp_bytes,
                                     int key_bits_lo, uint32_t* pairs_out, int32_t* coords_out) {
    if (n == 0) return hipSuccess;
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wall_sort_keys_kernel, dim3(blocks), dim3(256), 0, s, (const uint2*)pairs, n, keys0, index0);
    hipcub::DoubleBuffer<uint64_t> dk(keys0, keys1);
    hipcub::DoubleBuffer<uint32_t> dv(index0, index1);
    size_t bytes = (size_t)temp_bytes;
    // hi sits in bits [0, 32), lo in [32, 32 + key_bits_lo): the passes above the largest label's bits are skipped
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, bytes, dk, dv, (int64_t)n, 0, 32 + key_bits_lo, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wall_gather_kernel, dim3(blocks), dim3(256), 0, s, dk.Current(), dv.Current(), (const WallInt3*)coords, n,
                       (uint2*)pairs_out, (WallInt3*)coords_out);
    return hipGetLastError();
}

}  // namespace ta
