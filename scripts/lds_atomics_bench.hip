// LDS atomic costs on gfx950: cycles per wave-instruction for one wave alone, by how many lanes share an address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(64) k(uint64_t* out, int share, int reps, int stride_words) {
    __shared__ uint64_t tab[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) tab[i] = OP == 2 ? ~0ull : 0ull;
    __syncthreads();
    // lanes in groups of `share` hit the same word; groups are `stride_words` u64 apart
    const int slot = (lane / share) * stride_words;
    uint64_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (OP == 0) { atomicAdd((unsigned long long*)&tab[slot], 1ull); }                         // ds_add_u64, no return
        if (OP == 1) { atomicAdd((uint32_t*)&tab[slot], 1u); }                                     // ds_add_u32
        if (OP == 2) { acc += atomicCAS((unsigned long long*)&tab[slot], ~0ull, (unsigned long long)(slot + 1)); }   // ds_cmpst_rtn_b64 (dependent chain)
        if (OP == 3) { acc += tab[(slot + r) & 4095]; }                                            // ds_read_b64 (dependent through acc? no: independent)
        if (OP == 4) { atomicMin((uint32_t*)&tab[slot], (uint32_t)r); }
        if (OP == 5) { acc += atomicAdd((unsigned long long*)&tab[slot], 1ull); }                  // returning add
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = t1 - t0; out[1] = acc; }
}

int main() {
    uint64_t* d; hipMalloc(&d, 64);
    const char* names[] = {"ds_add_u64", "ds_add_u32", "ds_cmpst_rtn_b64", "ds_read_b64", "ds_min_u32", "ds_add_rtn_u64"};
    const int reps = 256;
    for (int op = 0; op < 6; ++op) {
        for (int share : {1, 2, 4, 8, 16, 64}) {
            for (int stride : {1, 6}) {
                uint64_t h[2];
                for (int it = 0; it < 2; ++it) {
                    switch (op) {
                        case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, d, share, reps, stride); break;
                        case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, d, share, reps, stride); break;
                        case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, d, share, reps, stride); break;
                        case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, d, share, reps, stride); break;
                        case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, d, share, reps, stride); break;
                        case 5: hipLaunchKernelGGL(k<5>, 1, 64, 0, 0, d, share, reps, stride); break;
                    }
                    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                }
                printf("%-18s share %2d stride %d u64: %7.1f cycles per wave-instruction\n", names[op], share, stride, (double)h[0] / reps);
            }
        }
    }
    return 0;
}
