// VALU issue rate on gfx950 for the integer ops of the sweep: cycles per wave-instruction per SIMD, by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ void k(uint64_t* out, int reps, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u, e = b + 9u, f = c + 11u, g = d + 13u, h = e + 15u;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { a += b; c += d; e += f; g += h; b += a; d += c; f += e; h += g; }                          // v_add_u32 x8
            if (OP == 1) { a = a < b ? a : c; c = c < d ? c : e; e = e < f ? e : g; g = g < h ? g : a;                // v_cmp + v_cndmask x4 (8 instr)
                           b ^= a; d ^= c; f ^= e; h ^= g; }                                                               // + 4 xor
            if (OP == 2) { a = __builtin_amdgcn_update_dpp(0, a, 0x111, 0xf, 0xf, false) + b; c = __builtin_amdgcn_update_dpp(0, c, 0x111, 0xf, 0xf, false) + d;
                           e = __builtin_amdgcn_update_dpp(0, e, 0x111, 0xf, 0xf, false) + f; g = __builtin_amdgcn_update_dpp(0, g, 0x111, 0xf, 0xf, false) + h; }   // v_add_dpp x4
            if (OP == 3) { a = __umul24(a, b) + c; c = __umul24(c, d) + e; e = __umul24(e, f) + g; g = __umul24(g, h) + a; }   // v_mad_u32_u24 x4
            if (OP == 4) { a += (b != c) ? 1u : 0u; c += (d != e) ? 1u : 0u; e += (f != g) ? 1u : 0u; g += (h != a) ? 1u : 0u; } // v_cmp + v_addc x4
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a + b + c + d + e + f + g + h == 0x12345u) out[1] = a;
}

int main() {
    uint64_t* d; (void)hipMalloc(&d, 64);
    const char* names[] = {"v_add_u32 x8", "cmp+cndmask x4 + xor x4 (12)", "v_add_dpp x4 (dpp mov+add: 8?)", "v_mad_u32_u24 x4", "v_cmp + v_addc x4 (8)"};
    const int per_iter[] = {8, 12, 8, 4, 8};
    const int reps = 2000;
    for (int op = 0; op < 5; ++op)
        for (int waves_per_simd : {1, 2, 4, 5, 8}) {
            uint64_t h[2];
            const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;      // one workgroup per CU: 4 SIMDs x waves
            const int blocks = 256 * waves_per_simd > 1024 ? (256 * waves_per_simd) / 1024 : 1;
            for (int it = 0; it < 2; ++it) {
                switch (op) {
                    case 0: hipLaunchKernelGGL(k<0>, blocks, threads, 0, 0, d, reps, 1u); break;
                    case 1: hipLaunchKernelGGL(k<1>, blocks, threads, 0, 0, d, reps, 1u); break;
                    case 2: hipLaunchKernelGGL(k<2>, blocks, threads, 0, 0, d, reps, 1u); break;
                    case 3: hipLaunchKernelGGL(k<3>, blocks, threads, 0, 0, d, reps, 1u); break;
                    case 4: hipLaunchKernelGGL(k<4>, blocks, threads, 0, 0, d, reps, 1u); break;
                }
                (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            }
            const double instr = (double)reps * 8 * per_iter[op];
            printf("%-34s waves/SIMD %d (blocks %d): %6.2f cycles per wave-instruction per wave -> %5.2f cycles per instruction on the SIMD\n", names[op], waves_per_simd, blocks,
                   (double)h[0] / instr, (double)h[0] / instr / waves_per_simd);
        }
    return 0;
}
