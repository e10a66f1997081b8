// Does a gather whose rows are PARTITIONED BY XCD come from the XCD L2s?    The k-hop pull gathers one
// 128 B row of X per entry of A'; every XCD gathers from the whole of X, so the eight 4 MiB L2s all hold the same hottest rows
// (hit rate 0.22).  If workgroup b only gathers rows of partition b % 8 — workgroups are dealt to the XCDs round-robin — a hot
// set of 8 x P rows is cached ONCE across the chip.  This micro measures the same reference stream three ways over a table of
// 8 x P rows:  (A) every workgroup draws from the whole table;  (B) workgroup b draws from partition b % 8;  (C) as B, but the
// partition is (b / 8) % 8 — the control: same per-workgroup footprint, NOT aligned with the dispatch order.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/xcdgather.hip -o tools/micro/xcdgather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u64 mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// MODE 0: whole table; 1: partition blockIdx % 8; 2: partition (blockIdx / 8) % 8.  Row ids are generated in registers (no index
// stream) so that only the gathers touch memory; 8 lanes x 16 B per row, G = 8 gathers in flight per lane (the pull's shape).
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const char* __restrict__ x, u32 prow, u32 rounds, u64* __restrict__ out) {
    const u32 lane = threadIdx.x & 63u, wl = lane & 7u, slot = lane >> 3;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 part = MODE == 1 ? (blockIdx.x & 7u) : ((blockIdx.x >> 3) & 7u);
    u32x4 acc = {0, 0, 0, 0};
    for (u32 r = 0; r < rounds; ++r) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u64 h = mix64((wave * rounds + r) * 64 + k * 8 + slot);
            const u32 row = MODE == 0 ? (u32)(h % (8ull * prow)) : part * prow + (u32)(h % prow);
            const char* p = x + (size_t)row * 128 + wl * 16;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int k = 0; k < 8; ++k) acc |= v[k];
    }
    if ((acc.x | acc.y | acc.z | acc.w) == 0x12345678u) out[0] = acc.x;
}

// the same question for 4-byte gathers (PageRank's w[col], a BFS level's bitmap probes): every lane its own random word, 8 in
// flight; table = 8 x pwords words, MODE as above
template <int MODE>
__global__ __launch_bounds__(256) void gather4_kernel(const u32* __restrict__ x, u32 pwords, u32 rounds, u64* __restrict__ out) {
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 part = MODE == 1 ? (blockIdx.x & 7u) : ((blockIdx.x >> 3) & 7u);
    u32 acc = 0;
    for (u32 r = 0; r < rounds; ++r) {
        u32 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u64 h = mix64((tid * rounds + r) * 8 + k);
            const u32 idx = MODE == 0 ? (u32)(h % (8ull * pwords)) : part * pwords + (u32)(h % pwords);
            asm volatile("global_load_dword %0, %1, off" : "=v"(v[k]) : "v"(x + idx) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int k = 0; k < 8; ++k) acc |= v[k];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int MODE>
static float run4(const u32* x, u32 pwords, u32 rounds, u64* out, int grid) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((gather4_kernel<MODE>), dim3(grid), dim3(256), 0, 0, x, pwords, rounds, out);
    (void)hipEventRecord(a);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((gather4_kernel<MODE>), dim3(grid), dim3(256), 0, 0, x, pwords, rounds, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

template <int MODE>
static float run(const char* x, u32 prow, u32 rounds, u64* out, int grid) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((gather_kernel<MODE>), dim3(grid), dim3(256), 0, 0, x, prow, rounds, out);
    (void)hipEventRecord(a);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((gather_kernel<MODE>), dim3(grid), dim3(256), 0, 0, x, prow, rounds, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

int main() {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount, grid = cus * 32;
    char* x; u64* out;
    (void)hipMalloc(&x, (size_t)8 * 8388608 * 4);   // 268 MB: the largest table of either sweep
    (void)hipMalloc(&out, 8);
    (void)hipMemset(x, 1, (size_t)8 * 8388608 * 4);
    const u32 rounds = 64;                                       // 64 gathers per wave and round
    const double refs = (double)grid * 4 * rounds * 64;
    printf("%s: %d CUs, grid %d x 256, %.1f M row gathers per launch (%.2f GB)\n", pr.gcnArchName, cus, grid, refs / 1e6, refs * 128 / 1e9);
    for (u32 prow : {2048u, 4096u, 8192u, 16384u, 24576u, 32768u, 65536u}) {
        const float a = run<0>(x, prow, rounds, out, grid), b = run<1>(x, prow, rounds, out, grid), c = run<2>(x, prow, rounds, out, grid);
        printf("partition %6u rows (%5.2f MB), table %6.1f MB:  whole-table %.3f ms %6.1f G rows/s | blockIdx%%8 %.3f ms %6.1f G rows/s (%.1f TB/s) | control %.3f ms %6.1f G rows/s\n",
               prow, prow * 128 / 1e6, 8.0 * prow * 128 / 1e6, a, refs / a / 1e6, b, refs / b / 1e6, refs * 128 / b / 1e9, c, refs / c / 1e6);
    }
    // 4-byte gathers: 8 x pwords words (pwords x 4 B per partition)
    const u32 r4 = 8;
    const double refs4 = (double)grid * 256 * r4 * 8;
    printf("4-byte gathers: %.1f M per launch\n", refs4 / 1e6);
    for (u32 pw : {65536u, 262144u, 524288u, 1048576u, 2097152u, 8388608u}) {
        const float a = run4<0>((const u32*)x, pw, r4, out, grid), b = run4<1>((const u32*)x, pw, r4, out, grid), c = run4<2>((const u32*)x, pw, r4, out, grid);
        printf("partition %8u words (%5.2f MB), table %6.1f MB:  whole-table %.3f ms %6.1f G/s | blockIdx%%8 %.3f ms %6.1f G/s | control %.3f ms %6.1f G/s\n",
               pw, pw * 4 / 1e6, 8.0 * pw * 4 / 1e6, a, refs4 / a / 1e6, b, refs4 / b / 1e6, c, refs4 / c / 1e6);
    }
    return 0;
}
