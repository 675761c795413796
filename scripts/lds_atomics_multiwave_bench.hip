// LDS atomic THROUGHPUT on gfx950: W waves of one workgroup (one CU) issue atomics back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ void k(uint64_t* out, int share, int reps) {
    __shared__ uint64_t tab[8192];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 8192; i += blockDim.x) tab[i] = 0ull;
    __syncthreads();
    // lanes in groups of `share` hit the same word; every wave has its own 64-word region (no cross-wave conflicts) unless OP says so
    const int slot = ((w * 64) & 8191) + (lane / share) * 1;
    uint64_t acc = 0;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (OP == 0) atomicAdd((unsigned long long*)&tab[slot], 1ull);
        if (OP == 1) atomicAdd((uint32_t*)&tab[slot], 1u);
        if (OP == 2) acc += tab[(slot + r) & 8191];
        if (OP == 3) tab[(slot + 64 * 0) & 8191] = (uint64_t)r;
        if (OP == 4) atomicMin((uint32_t*)&tab[slot], (uint32_t)r);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { out[0] = t1 - t0; out[1] = acc; }
    if (acc == 12345) out[2] = acc;
}

int main() {
    uint64_t* d; (void)hipMalloc(&d, 64);
    const char* names[] = {"ds_add_u64", "ds_add_u32", "ds_read_b64", "ds_write_b64", "ds_min_u32"};
    const int reps = 512;
    for (int op = 0; op < 5; ++op)
        for (int share : {1, 8})
            for (int waves : {1, 2, 4, 8, 16}) {
                uint64_t h[2];
                for (int it = 0; it < 2; ++it) {
                    switch (op) {
                        case 0: hipLaunchKernelGGL(k<0>, 1, 64 * waves, 0, 0, d, share, reps); break;
                        case 1: hipLaunchKernelGGL(k<1>, 1, 64 * waves, 0, 0, d, share, reps); break;
                        case 2: hipLaunchKernelGGL(k<2>, 1, 64 * waves, 0, 0, d, share, reps); break;
                        case 3: hipLaunchKernelGGL(k<3>, 1, 64 * waves, 0, 0, d, share, reps); break;
                        case 4: hipLaunchKernelGGL(k<4>, 1, 64 * waves, 0, 0, d, share, reps); break;
                    }
                    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                }
                printf("%-14s share %d waves %2d: %8.1f cycles per wave-instruction per wave, %7.1f cycles per instruction on the CU\n", names[op], share, waves,
                       (double)h[0] / reps, (double)h[0] / reps / waves);
            }
    return 0;
}
