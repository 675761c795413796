// Pipe throughput on gfx950 at the sweep's occupancy (4 workgroups of 256 threads a CU = 4 waves a SIMD, 16 waves sharing one LDS):
// whole-kernel times (hipEvents), every CU busy -- NOT one wave's own clock (an older microbenchmark of this repository read wave 0's
// s_memtime only, and the oldest wave of a SIMD is served first: it measured one wave's issue rate whatever else ran).
//   hipcc --offload-arch=gfx950 -O3 scripts/pipes_bench.hip -o /tmp/pipes_bench && /tmp/pipes_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int CUS = 256, WGS_PER_CU = 4, THREADS = 256;

// ---- VALU: N independent integer instructions per iteration
template <int OP>
__global__ void __launch_bounds__(THREADS) valu_k(uint32_t* out, int reps, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u, e = b + 9u, f = c + 11u, g = d + 13u, h = e + 15u;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { a += b; c += d; e += f; g += h; b += a; d += c; f += e; h += g; }                      // 8 v_add_u32
            if (OP == 1) { a = a < b ? a : c; c = c < d ? c : e; e = e < f ? e : g; g = g < h ? g : a; b ^= a; d ^= c; f ^= e; h ^= g; }   // 4 cmp + 4 cndmask + 4 xor
            if (OP == 2) { a = __umul24(a, b) + c; c = __umul24(c, d) + e; e = __umul24(e, f) + g; g = __umul24(g, h) + a; }              // 4 v_mad_u32_u24
        }
    }
    if (a + b + c + d + e + f + g + h == 0x12345u) out[0] = a;
}

// ---- LDS: one instruction kind per kernel, `active` lanes of each wave take part, `share` lanes use one address
// KIND 0 ds_write2_b32, 1 ds_write_b64, 2 ds_write_b32, 3 ds_add_u32 (no return), 4 ds_add_u64, 5 ds_read_b64, 6 ds_read_b128, 7 ds_min_u32
template <int KIND>
__global__ void __launch_bounds__(THREADS) lds_k(uint32_t* out, int reps, int active, int share, int stride_dw) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[8192];        // 32 KB a workgroup like the sweep
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += THREADS) buf[i] = i;
    __syncthreads();
    // address: lanes lane / share share one slot; slots `stride_dw` dwords apart; each wave its own 2048-dword quarter
    const uint32_t slot = (uint32_t)(lane / share);
    uint32_t addr = (uint32_t)(uintptr_t)&buf[w * 2048] + ((slot * (uint32_t)stride_dw * 4u) & 8191u & ~15u);
    uint32_t acc = 0;
    uint32_t v0 = lane, v1 = lane * 3;
    uint64_t q = 0; uint4 qq = {0, 0, 0, 0};
    if (lane < active) {
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == 0) asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(addr), "v"(v0), "v"(v1) : "memory");
                if (KIND == 1) asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(q) : "memory");
                if (KIND == 2) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v0) : "memory");
                if (KIND == 3) asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(v0) : "memory");
                if (KIND == 4) asm volatile("ds_add_u64 %0, %1" :: "v"(addr), "v"(q) : "memory");
                if (KIND == 5) { asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(addr) : "memory"); }
                if (KIND == 6) { asm volatile("ds_read_b128 %0, %1" : "=v"(qq) : "v"(addr) : "memory"); }
                if (KIND == 7) asm volatile("ds_min_u32 %0, %1" :: "v"(addr), "v"(v0) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    acc = (uint32_t)q + qq.x + qq.w;
    __syncthreads();
    if (acc == 0x1234567u || buf[threadIdx.x] == 0xdeadbeefu) out[0] = acc;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); (void)hipDeviceSynchronize();
    double best = 1e9;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(a, 0); launch(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    uint32_t* d; (void)hipMalloc(&d, 64);
    int clk_khz = 0; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double ghz = clk_khz / 1e6;
    printf("clock %.2f GHz (attribute); grid %d workgroups of %d threads = %d waves a SIMD\n", ghz, CUS * WGS_PER_CU, THREADS, WGS_PER_CU);
    const char* vn[] = {"v_add_u32 x8 (8 a loop)", "cmp+cndmask x4 + xor x4 (12 a loop)", "v_mad_u32_u24 x4 (4 a loop)"};
    const int per[] = {8, 12, 4};
    for (int wgs : {1, 2, 4}) {
        for (int op = 0; op < 3; ++op) {
            const int reps = 4000;
            double ms = 0;
            if (op == 0) ms = time_ms([&] { hipLaunchKernelGGL(valu_k<0>, CUS * wgs, THREADS, 0, 0, d, reps, 1u); });
            if (op == 1) ms = time_ms([&] { hipLaunchKernelGGL(valu_k<1>, CUS * wgs, THREADS, 0, 0, d, reps, 1u); });
            if (op == 2) ms = time_ms([&] { hipLaunchKernelGGL(valu_k<2>, CUS * wgs, THREADS, 0, 0, d, reps, 1u); });
            const double instr_per_simd = (double)reps * 8 * per[op] * wgs;          // wave-instructions issued on one SIMD
            printf("VALU %-38s %d waves/SIMD: %.3f ms -> %.2f cycles per wave-instruction on the SIMD\n", vn[op], wgs, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
        }
    }
    const char* ln[] = {"ds_write2_b32", "ds_write_b64", "ds_write_b32", "ds_add_u32", "ds_add_u64", "ds_read_b64", "ds_read_b128", "ds_min_u32"};
    for (int kind = 0; kind < 8; ++kind) {
        struct Cfg { int active, share, stride; };
        std::vector<Cfg> cfgs;
        if (kind <= 2) cfgs = {{64, 1, 2}, {2, 1, 2}, {8, 1, 2}, {32, 1, 2}, {64, 64, 2}};
        else if (kind == 3 || kind == 4 || kind == 7) cfgs = {{64, 1, 2}, {64, 2, 2}, {64, 4, 2}, {64, 8, 2}, {64, 16, 2}, {64, 1, 6}, {64, 4, 12}, {16, 4, 2}};
        else cfgs = {{64, 1, 2}, {64, 1, 4}, {64, 4, 2}, {64, 1, 13}};
        for (const Cfg& c : cfgs) {
            const int reps = 2000;
            double ms = 0;
            auto L = [&](auto kfn) { ms = time_ms([&] { hipLaunchKernelGGL(kfn, CUS * WGS_PER_CU, THREADS, 0, 0, d, reps, c.active, c.share, c.stride); }); };
            switch (kind) {
                case 0: L(lds_k<0>); break; case 1: L(lds_k<1>); break; case 2: L(lds_k<2>); break; case 3: L(lds_k<3>); break;
                case 4: L(lds_k<4>); break; case 5: L(lds_k<5>); break; case 6: L(lds_k<6>); break; case 7: L(lds_k<7>); break;
            }
            const double instr_per_cu = (double)reps * 8 * WGS_PER_CU * 4;           // wave-instructions through one CU's LDS
            printf("LDS  %-14s active lanes %2d, lanes per address %2d, slot stride %2d dwords: %.3f ms -> %.1f cycles per wave-instruction on the CU\n",
                   ln[kind], c.active, c.share, c.stride, ms, ms * 1e-3 * ghz * 1e9 / instr_per_cu);
        }
    }
    return 0;
}
