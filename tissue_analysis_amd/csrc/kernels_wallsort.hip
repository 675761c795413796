// kernels_wallsort.hip -- the wall-voxel records grouped by label pair ON THE DEVICE (what every caller of the table
// wants: WallTable used to do a stable host sort of ~10^7 records).  A stable LSD radix sort (hipCUB / rocPRIM, a plain
// library sort) of key = lo << bits | hi -- bits = what a label of this volume takes, found by the count pass -- with the
// record index as value keeps the memory order inside each pair; a gather then writes the records in that order.  Kept in its own file: the library headers are slow to compile.
#include "ta_kernels.h"

#include <hipcub/hipcub.hpp>

namespace ta {

// key = lo << bits | hi with `bits` = the bits a label of this volume takes: the sort runs over 2 x bits instead of 48 / 64,
// and over 32-bit keys where that fits (labels below 2^16: every uint16 volume, most uint32 ones)
template <typename K>
__global__ void __launch_bounds__(256) wall_sort_keys_kernel(const uint2* pairs, uint64_t n, K* keys, uint32_t* index, int bits) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 p = pairs[i];
        keys[i] = (K)(((uint64_t)p.x << bits) | p.y);
        index[i] = (uint32_t)i;
    }
}

struct __attribute__((packed, aligned(4))) WallInt3 { int32_t x, y, z; };

template <typename K>
__global__ void __launch_bounds__(256) wall_gather_kernel(const K* keys, const uint32_t* index, const WallInt3* coords, uint64_t n,
                                                          uint2* pairs_out, WallInt3* coords_out, int bits) {
    const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = (uint64_t)keys[i];
        pairs_out[i] = make_uint2((uint32_t)(k >> bits), (uint32_t)(k & mask));
        coords_out[i] = coords[index[i]];
    }
}

uint64_t wall_sort_temp_bytes(uint64_t n) {
    size_t wide = 0, narrow = 0;
    hipcub::DoubleBuffer<uint64_t> dk(nullptr, nullptr);
    hipcub::DoubleBuffer<uint32_t> dn(nullptr, nullptr);
    hipcub::DoubleBuffer<uint32_t> dv(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, wide, dk, dv, (int64_t)n, 0, 64, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, narrow, dn, dv, (int64_t)n, 0, 32, nullptr);
    return (uint64_t)(wide > narrow ? wide : narrow);
}

template <typename K>
static hipError_t wall_group(hipStream_t s, const uint32_t* pairs, const int32_t* coords, uint64_t n, K* keys0, K* keys1,
                             uint32_t* index0, uint32_t* index1, void* temp, uint64_t temp_bytes, int bits, uint32_t* pairs_out,
                             int32_t* coords_out) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wall_sort_keys_kernel<K>, dim3(blocks), dim3(256), 0, s, (const uint2*)pairs, n, keys0, index0, bits);
    hipcub::DoubleBuffer<K> dk(keys0, keys1);
    hipcub::DoubleBuffer<uint32_t> dv(index0, index1);
    size_t bytes = (size_t)temp_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, bytes, dk, dv, (int64_t)n, 0, 2 * bits, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wall_gather_kernel<K>, dim3(blocks), dim3(256), 0, s, dk.Current(), dv.Current(), (const WallInt3*)coords, n,
                       (uint2*)pairs_out, (WallInt3*)coords_out, bits);
    return hipGetLastError();
}

// pairs / coords: the records in memory order (n < 2^32).  keys[2], index[2]: double buffers of n entries (keys: 8 bytes an
// entry, used as 4-byte ones where two labels fit 32 bits); temp: wall_sort_temp_bytes; label_bits: the bits a label takes
// (1 .. 32).  Writes the grouped records to pairs_out / coords_out.  Only enqueues work.
hipError_t launch_wall_group_by_pair(hipStream_t s, const uint32_t* pairs, const int32_t* coords, uint64_t n, uint64_t* keys0,
                                     uint64_t* keys1, uint32_t* index0, uint32_t* index1, void* temp, uint64_t temp_bytes,
                                     int label_bits, uint32_t* pairs_out, int32_t* coords_out) {
    if (n == 0) return hipSuccess;
    if (label_bits < 1) label_bits = 1;
    if (label_bits > 32) label_bits = 32;
    if (2 * label_bits <= 32)
        return wall_group<uint32_t>(s, pairs, coords, n, (uint32_t*)keys0, (uint32_t*)keys1, index0, index1, temp, temp_bytes, label_bits,
                                    pairs_out, coords_out);
    return wall_group<uint64_t>(s, pairs, coords, n, keys0, keys1, index0, index1, temp, temp_bytes, label_bits, pairs_out, coords_out);
}

}  // namespace ta
