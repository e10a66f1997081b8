// gatherq.hip — which queue caps a random ROW gather on gfx950?  
//
// The last hop of a count-only k-hop chain (bitexpand.hip bp_pull_kernel<.., dense, count>) gathers one 64-byte row of the bit
// state X per entry of A': 64 G rows/s at RMAT-22 whatever the row width.  This micro reads off, with the index stream
// generated in registers so that ONLY the gathers touch memory:
//   sweep   rows/s against (table residency: L2 / Infinity Cache / HBM) x (wavefronts per CU) x (gathers in flight per lane)
//           for 64- and 128-byte rows on the vector path                                  -> the in-flight ceiling per CU
//   cus     rows/s against the number of ACTIVE CUs (one 1024-thread workgroup per CU)  -> per-CU limit or shared limit?
//   paths   the same rows through s_load_dwordx16 (scalar cache path), through global_load_lds_dwordx4 (LDS-DMA), and
//           vector + scalar mixed in one wavefront                                        -> does a second path add requests?
//   pairs   64-byte rows whose 128-byte line-mates are gathered in the same instruction / one round later / never
//                                                                                         -> is the fill granule 64 or 128 B?
//   calib   one launch of each named kernel with a known gather count, to be run under rocprofv3 --pmc (tools/pmc_pass.py)
//           so FETCH_SIZE / TCC_EA0_RDREQ can be calibrated per access width
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/gatherq.hip -o tools/micro/gatherq
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ u32 h32(u32 x) {   // murmur3 finaliser: cheap enough not to bound an L2-resident gather
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

// ---- vector path -------------------------------------------------------------------------------------------------------
// ROWB bytes per row (4: a lane per word; 64 / 128 / 256: ROWB / 16 lanes of 16 B), G gathers in flight per lane.
// PAIR (64-byte rows): 0 independent rows; 1 the two rows of a 128-byte line in ONE instruction (adjacent lane groups);
// 2 the line-mate gathered one round (G instructions) after its partner.
template <int ROWB, int G, int PAIR>
__global__ __launch_bounds__(1024) void vgather(const char* __restrict__ x, u32 rowmask, u32 rounds, u64* __restrict__ out) {
    constexpr int LPR = ROWB >= 16 ? ROWB / 16 : 1, RPI = 64 / LPR;
    constexpr int SH = ROWB == 4 ? 2 : ROWB == 64 ? 6 : ROWB == 128 ? 7 : 8;
    const u32 lane = threadIdx.x & 63u, wl = lane % LPR, slot = lane / LPR;
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    u32x4 acc = {0, 0, 0, 0};
    for (u32 r = 0; r < rounds; ++r) {
        u32x4 v[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            u32 row;
            if (PAIR == 0) row = h32(((wave * rounds + r) * G + k) * RPI + slot) & rowmask;
            else if (PAIR == 1) row = ((h32(((wave * rounds + r) * G + k) * RPI + (slot >> 1)) << 1) | (slot & 1u)) & rowmask;
            else row = ((h32(((wave * rounds + (r & ~1u)) * G + k) * RPI + slot) << 1) | (r & 1u)) & rowmask;
            const char* p = x + ((size_t)row << SH) + wl * 16;
            if (ROWB == 4) {
                asm volatile("global_load_dword %0, %1, off" : "=v"(v[k].x) : "v"(p) : "memory");
                v[k].y = v[k].z = v[k].w = 0;
            } else {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(p) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < G; ++k) {
            asm volatile("" : "+v"(v[k]));
            acc |= v[k];
        }
    }
    if ((acc.x | acc.y | acc.z | acc.w) == 0x12345678u) out[0] = acc.x;
}

// ---- scalar path: one 64-byte row per s_load_dwordx16, S rows in flight per wavefront ---------------------------------------
template <int S>
__global__ __launch_bounds__(1024) void sgather(const char* __restrict__ x, u32 rowmask, u32 rounds, u64* __restrict__ out) {
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    u32 acc = 0;
    for (u32 r = 0; r < rounds; ++r) {
        u32x16 v[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const u32 row = h32((wave * rounds + r) * S + k) & rowmask;
            const u64 a = (u64)x + ((u64)row << 6);
            const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)a), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(a >> 32));
            const u64 p = ((u64)hi << 32) | lo;
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v[k]) : "s"(p) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < S; ++k) {
            asm volatile("" : "+s"(v[k]));
            acc |= v[k][0] | v[k][5] | v[k][10] | v[k][15];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// ---- vector + scalar in one wavefront: G vector instructions (16 rows of 64 B each) and S scalar rows per round -------------
template <int G, int S>
__global__ __launch_bounds__(1024) void mixgather(const char* __restrict__ x, u32 rowmask, u32 rounds, u64* __restrict__ out) {
    const u32 lane = threadIdx.x & 63u, wl = lane & 3u, slot = lane >> 2;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    u32x4 acc = {0, 0, 0, 0};
    u32 sacc = 0;
    for (u32 r = 0; r < rounds; ++r) {
        u32x4 v[G];
        u32x16 s[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const u32 row = h32(0x40000000u + (wave * rounds + r) * S + k) & rowmask;
            const u64 a = (u64)x + ((u64)row << 6);
            const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)a), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(a >> 32));
            const u64 p = ((u64)hi << 32) | lo;
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(s[k]) : "s"(p) : "memory");
        }
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const u32 row = h32(((wave * rounds + r) * G + k) * 16 + slot) & rowmask;
            const char* p = x + ((size_t)row << 6) + wl * 16;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < G; ++k) { asm volatile("" : "+v"(v[k])); acc |= v[k]; }
#pragma unroll
        for (int k = 0; k < S; ++k) { asm volatile("" : "+s"(s[k])); sacc |= s[k][0] | s[k][7] | s[k][15]; }
    }
    if ((acc.x | acc.y | acc.z | acc.w | sacc) == 0x12345678u) out[0] = acc.x;
}

// ---- LDS-DMA path: global_load_lds_dwordx4, every lane its own global address, 1 KiB of LDS per instruction -------------------
template <int G>
__global__ __launch_bounds__(1024) void dmagather(const char* __restrict__ x, u32 rowmask, u32 rounds, u64* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char s_dma[];
    const u32 lane = threadIdx.x & 63u, wl = lane & 3u, slot = lane >> 2;
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, wib = threadIdx.x >> 6;
    char* mine = s_dma + (size_t)wib * G * 1024;
    u32x4 acc = {0, 0, 0, 0};
    for (u32 r = 0; r < rounds; ++r) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const u32 row = h32(((wave * rounds + r) * G + k) * 16 + slot) & rowmask;
            const char* p = x + ((size_t)row << 6) + wl * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(mine + k * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < G; ++k) acc |= *reinterpret_cast<const u32x4*>(mine + k * 1024 + lane * 16);
    }
    if ((acc.x | acc.y | acc.z | acc.w) == 0x12345678u) out[0] = acc.x;
}

// ---- coalesced stream (calibration of FETCH_SIZE on a known byte count) ------------------------------------------------------
__global__ __launch_bounds__(256) void stream_read(const u32x4* __restrict__ x, size_t n16, u64* __restrict__ out) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) acc |= x[i];
    if ((acc.x | acc.y | acc.z | acc.w) == 0x12345678u) out[0] = acc.x;
}

static int g_cus = 256;
static u64* g_out;

template <typename K, typename... A>
static float timeit(K kern, dim3 grid, dim3 block, size_t lds, int reps, A... args) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, grid, block, lds, 0, args...);
    (void)hipEventRecord(a);
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(kern, grid, block, lds, 0, args...);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("  launch error: %s\n", hipGetErrorString(e)); return -1.f; }
    return ms / reps;
}

struct Table { const char* name; size_t bytes; };
static const Table TABLES[3] = {{"L2 (2 MiB)", (size_t)2 << 20}, {"MALL (128 MiB)", (size_t)128 << 20}, {"HBM (8 GiB)", (size_t)8 << 30}};

// one sweep cell: 256-thread workgroups, `bpc` resident per CU (by the LDS request), G in flight per lane
template <int ROWB, int G>
static void cell(const char* x, size_t tbytes, int bpc, double target_rows, char* buf) {
    auto kern = vgather<ROWB, G, 0>;
    size_t lds = (size_t)(160 * 1024 / bpc) & ~(size_t)1023;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds);
    constexpr int RPI = ROWB >= 16 ? 64 / (ROWB / 16) : 64;
    const int grid = g_cus * occ;
    const double per_round = (double)grid * 4 * G * RPI;
    u32 rounds = (u32)(target_rows / per_round);
    if (rounds < 2) rounds = 2;
    const u32 rowmask = (u32)(tbytes / ROWB) - 1u;
    const float ms = timeit(kern, dim3(grid), dim3(256), lds, 2, x, rowmask, rounds, g_out);
    const double rows = per_round * rounds;
    sprintf(buf, "%5.1f(w%d)", rows / ms / 1e6, occ * 4);
}

template <int ROWB>
static void sweep(const char* x) {
    for (const Table& t : TABLES) {
        printf("\n[sweep] %d-byte rows, table %s: G rows/s (w = resident wavefronts per CU)\n", ROWB, t.name);
        printf("%-14s %12s %12s %12s %12s %12s\n", "blocks/CU", "G=1", "G=2", "G=4", "G=8", "G=16");
        for (int bpc : {1, 2, 4, 6, 8}) {
            char c[5][32];
            const double target = t.bytes <= ((size_t)2 << 20) ? 4e8 : 1.6e8;
            cell<ROWB, 1>(x, t.bytes, bpc, target / 4, c[0]);
            cell<ROWB, 2>(x, t.bytes, bpc, target / 2, c[1]);
            cell<ROWB, 4>(x, t.bytes, bpc, target, c[2]);
            cell<ROWB, 8>(x, t.bytes, bpc, target, c[3]);
            cell<ROWB, 16>(x, t.bytes, bpc, target, c[4]);
            printf("%-14d %12s %12s %12s %12s %12s\n", bpc, c[0], c[1], c[2], c[3], c[4]);
        }
    }
}

// rows/s against the number of active CUs: one 1024-thread workgroup (16 wavefronts) per CU, pinned by a 96 KiB LDS request
template <int ROWB>
static void cus(const char* x) {
    auto kern = vgather<ROWB, 8, 0>;
    const size_t lds = 96 * 1024;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    constexpr int RPI = 64 / (ROWB / 16);
    for (const Table& t : TABLES) {
        printf("\n[cus] %d-byte rows, table %s, 16 wavefronts x 8 in flight per CU: active CUs -> G rows/s (per CU M rows/s)\n", ROWB, t.name);
        for (int n : {8, 16, 32, 64, 96, 128, 160, 192, 224, 256}) {
            if (n > g_cus) continue;
            const double per_round = (double)n * 16 * 8 * RPI;
            u32 rounds = (u32)(1.2e8 * n / g_cus / per_round);
            if (rounds < 2) rounds = 2;
            const float ms = timeit(kern, dim3(n), dim3(1024), lds, 2, x, (u32)(t.bytes / ROWB) - 1u, rounds, g_out);
            printf("  %3d CUs: %6.1f G rows/s  (%6.1f M rows/s per CU)\n", n, per_round * rounds / ms / 1e6, per_round * rounds / ms / 1e3 / n);
        }
    }
}

static void paths(const char* x) {
    const size_t lds_small = 20 * 1024;   // 8 workgroups of 256 per CU
    for (const Table& t : TABLES) {
        const u32 rowmask = (u32)(t.bytes / 64) - 1u;
        printf("\n[paths] 64-byte rows, table %s, 32 wavefronts per CU (8 x 256 threads)\n", t.name);
        const int grid = g_cus * 8;
        {
            const double pr = (double)grid * 4 * 8 * 16;
            const u32 rounds = (u32)(1.6e8 / pr);
            const float ms = timeit(vgather<64, 8, 0>, dim3(grid), dim3(256), lds_small, 2, x, rowmask, rounds, g_out);
            printf("  vector  global_load_dwordx4, 8 in flight per lane (128 rows per wave): %6.1f G rows/s\n", pr * rounds / ms / 1e6);
        }
#define SCAL(S)                                                                                                                  \
        {                                                                                                                        \
            const double pr = (double)grid * 4 * S;                                                                              \
            const u32 rounds = (u32)(2e7 / pr) + 2;                                                                              \
            const float ms = timeit(sgather<S>, dim3(grid), dim3(256), lds_small, 2, x, rowmask, rounds, g_out);                \
            printf("  scalar  s_load_dwordx16, %d rows in flight per wave:                     %6.1f G rows/s\n", S, pr * rounds / ms / 1e6); \
        }
        SCAL(1) SCAL(2) SCAL(4)
#define MIX(G, S)                                                                                                                \
        {                                                                                                                        \
            const double pr = (double)grid * 4 * (G * 16 + S);                                                                   \
            const u32 rounds = (u32)(1.6e8 / pr) + 2;                                                                            \
            const float ms = timeit(mixgather<G, S>, dim3(grid), dim3(256), lds_small, 2, x, rowmask, rounds, g_out);           \
            printf("  mixed   %d vector instructions + %d scalar rows per round:                %6.1f G rows/s (%.1f %% of the rows on the scalar path)\n", \
                   G, S, pr * rounds / ms / 1e6, 100.0 * S / (G * 16 + S));                                                       \
        }
        MIX(8, 2) MIX(8, 4) MIX(4, 4)
#define DMA(G)                                                                                                                   \
        {                                                                                                                        \
            const int bpc = 160 / (G * 4) < 8 ? 160 / (G * 4) : 8;                                                               \
            const size_t lds = (size_t)G * 4 * 1024;                                                                             \
            (void)hipFuncSetAttribute((const void*)dmagather<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
            int occ = 0;                                                                                                         \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, dmagather<G>, 256, lds);                                    \
            const int g2 = g_cus * occ;                                                                                          \
            const double pr = (double)g2 * 4 * G * 16;                                                                           \
            const u32 rounds = (u32)(1.6e8 / pr) + 2;                                                                            \
            const float ms = timeit(dmagather<G>, dim3(g2), dim3(256), lds, 2, x, rowmask, rounds, g_out);                      \
            printf("  LDS-DMA global_load_lds_dwordx4, %d in flight per lane, %d waves per CU:  %6.1f G rows/s\n", G, occ * 4, pr * rounds / ms / 1e6); \
            (void)bpc;                                                                                                           \
        }
        DMA(4) DMA(8)
    }
}

static void pairs(const char* x) {
    const size_t lds_small = 20 * 1024;
    const int grid = g_cus * 8;
    for (const Table& t : TABLES) {
        const u32 rowmask = (u32)(t.bytes / 64) - 1u;
        const double pr = (double)grid * 4 * 8 * 16;
        const u32 rounds = ((u32)(1.6e8 / pr) + 2) & ~1u;
        const float a = timeit(vgather<64, 8, 0>, dim3(grid), dim3(256), lds_small, 2, x, rowmask, rounds, g_out);
        const float b = timeit(vgather<64, 8, 1>, dim3(grid), dim3(256), lds_small, 2, x, rowmask, rounds, g_out);
        const float c = timeit(vgather<64, 8, 2>, dim3(grid), dim3(256), lds_small, 2, x, rowmask, rounds, g_out);
        const float d = timeit(vgather<128, 8, 0>, dim3(grid), dim3(256), lds_small, 2, x, (u32)(t.bytes / 128) - 1u, rounds, g_out);
        printf("\n[pairs] table %s: 64-byte rows independent %6.1f G rows/s | line-mates in one instruction %6.1f | line-mate one round later %6.1f | 128-byte rows %6.1f G rows/s\n",
               t.name, pr * rounds / a / 1e6, pr * rounds / b / 1e6, pr * rounds / c / 1e6, pr * rounds / 2 / d / 1e6);
    }
}

// one launch of each kernel over the 8 GiB table with the gather counts printed — run under rocprofv3 --pmc
static void calib(const char* x) {
    const size_t tb = (size_t)8 << 30;
    const int grid = g_cus * 8;
    const size_t lds_small = 20 * 1024;
    const u32 rounds = 16;
    printf("[calib] table 8 GiB; every kernel once\n");
    hipLaunchKernelGGL(stream_read, dim3(g_cus * 8), dim3(256), 0, 0, (const u32x4*)x, ((size_t)1 << 30) / 16, g_out);
    printf("  stream_read: %zu bytes\n", (size_t)1 << 30);
    hipLaunchKernelGGL((vgather<4, 8, 0>), dim3(grid), dim3(256), lds_small, 0, x, (u32)(tb / 4) - 1u, rounds, g_out);
    printf("  vgather<4,8,0>: %.0f gathers of 4 B\n", (double)grid * 4 * 8 * 64 * rounds);
    hipLaunchKernelGGL((vgather<64, 8, 0>), dim3(grid), dim3(256), lds_small, 0, x, (u32)(tb / 64) - 1u, rounds, g_out);
    printf("  vgather<64,8,0>: %.0f gathers of 64 B (independent rows)\n", (double)grid * 4 * 8 * 16 * rounds);
    hipLaunchKernelGGL((vgather<64, 8, 1>), dim3(grid), dim3(256), lds_small, 0, x, (u32)(tb / 64) - 1u, rounds, g_out);
    printf("  vgather<64,8,1>: %.0f gathers of 64 B (line-mates in one instruction: %.0f lines)\n", (double)grid * 4 * 8 * 16 * rounds, (double)grid * 4 * 8 * 8 * rounds);
    hipLaunchKernelGGL((vgather<64, 8, 2>), dim3(grid), dim3(256), lds_small, 0, x, (u32)(tb / 64) - 1u, rounds, g_out);
    printf("  vgather<64,8,2>: %.0f gathers of 64 B (line-mate one round later: %.0f lines)\n", (double)grid * 4 * 8 * 16 * rounds, (double)grid * 4 * 8 * 8 * rounds);
    hipLaunchKernelGGL((vgather<128, 8, 0>), dim3(grid), dim3(256), lds_small, 0, x, (u32)(tb / 128) - 1u, rounds, g_out);
    printf("  vgather<128,8,0>: %.0f gathers of 128 B\n", (double)grid * 4 * 8 * 8 * rounds);
    hipLaunchKernelGGL((vgather<256, 8, 0>), dim3(grid), dim3(256), lds_small, 0, x, (u32)(tb / 256) - 1u, rounds, g_out);
    printf("  vgather<256,8,0>: %.0f gathers of 256 B\n", (double)grid * 4 * 8 * 4 * rounds);
    hipLaunchKernelGGL((sgather<4>), dim3(grid), dim3(256), lds_small, 0, x, (u32)(tb / 64) - 1u, rounds * 8, g_out);
    printf("  sgather<4>: %.0f scalar gathers of 64 B\n", (double)grid * 4 * 4 * rounds * 8);
    (void)hipDeviceSynchronize();
    printf("  last error: %s\n", hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    g_cus = pr.multiProcessorCount;
    char* x;
    const size_t tb = (size_t)8 << 30;
    if (hipMalloc(&x, tb) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMalloc(&g_out, 8);
    (void)hipMemset(x, 1, tb);
    (void)hipDeviceSynchronize();
    printf("%s: %d CUs, clock %d MHz, L2 %d KiB\n", pr.gcnArchName, g_cus, pr.clockRate / 1000, pr.l2CacheSize / 1024);
    const char* what = argc > 1 ? argv[1] : "all";
    const bool all = !strcmp(what, "all");
    if (!strcmp(what, "calib")) { calib(x); return 0; }
    if (all || !strcmp(what, "pairs")) pairs(x);
    if (all || !strcmp(what, "cus")) { cus<64>(x); cus<128>(x); }
    if (all || !strcmp(what, "paths")) paths(x);
    if (all || !strcmp(what, "sweep")) { sweep<64>(x); sweep<128>(x); }
    (void)hipDeviceSynchronize();
    printf("done: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
