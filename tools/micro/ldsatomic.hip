// LDS atomic throughput on gfx950: ds_or_b32 (no return) against ds_read_b32 / plain read-modify-write, for the address
// patterns of blocked.hip's output window (random words of a 2 KiB .. 64 KiB window, same-address collisions included).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/ldsatomic.hip -o tools/micro/ldsatomic
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32;

template <int MODE>   // 0 ds_or random, 1 ds_read random, 2 ds_or conflict-free (lane-private word), 3 ds_or with 1/8 of the lanes active, 4 plain rmw random
__global__ __launch_bounds__(1024) void k(u32* out, u32 words, int iters) {
    extern __shared__ u32 s[];
    for (u32 i = threadIdx.x; i < words; i += blockDim.x) s[i] = 0;
    __syncthreads();
    u32 r = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
    u32 acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            r = r * 1664525u + 1013904223u;
            const u32 a = MODE == 2 ? (threadIdx.x & (words - 1)) : ((r >> 8) & (words - 1));
            if (MODE == 0 || MODE == 2) atomicOr(&s[a], 1u << (r & 31));
            else if (MODE == 3) { if ((threadIdx.x & 7) == 0) atomicOr(&s[a], 1u << (r & 31)); }
            else if (MODE == 1) acc ^= s[a];
            else s[a] |= 1u << (r & 31);
        }
    }
    __syncthreads();
    if (acc == 0x12345u) out[0] = acc;
    if (threadIdx.x == 0) out[1 + blockIdx.x] = s[threadIdx.x];
}

template <int MODE>
static void run(const char* name, u32 words, u32* out) {
    const int cus = 256, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(1024), words * 4, 0, out, words, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(1024), words * 4, 0, out, words, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)cus * 1024 * iters * 8 * (MODE == 3 ? 0.125 : 1.0);
    printf("%-44s window %6u B: %8.3f ms  %7.2f G lane-ops/s chip-wide = %5.2f lane-ops / clk / CU (2.1 GHz)\n", name, words * 4, ms,
           ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.1);
}

int main() {
    u32* out; hipMalloc(&out, 4096);
    for (u32 words : {512u, 4096u, 16384u}) {
        run<1>("ds_read_b32 random", words, out);
        run<0>("ds_or_b32 random (all lanes)", words, out);
        run<3>("ds_or_b32 random (1 lane in 8 active)", words, out);
        run<2>("ds_or_b32 lane-private word", words, out);
        run<4>("plain read-modify-write random", words, out);
    }
    return 0;
}
