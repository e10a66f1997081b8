// What a BFS level launch costs before it does any work: back-to-back launches of a 7-workgroups-per-CU grid that
// (0) return at once, (1) zero a 512 KiB bitmap, (2) + take the two-level ticket, (3) + the last workgroup's wave
// reads 4 x 64 statistic slots with agent-scope atomic loads and writes a control block (the fused_ctrl shape).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/levelfloor.hip -o tools/micro/levelfloor
#include <hip/hip_runtime.h>
#include <stdio.h>

struct Ctrl { unsigned tick[64 * 32]; unsigned tick_mid[8 * 32]; unsigned tick_top; unsigned long long slot[4][64]; unsigned long long out[8]; unsigned done; };

template <int MODE, int PAD = 1, bool THREE = false, bool SLOT = true>
__global__ __launch_bounds__(256) void level(Ctrl* c, unsigned long long* bm, unsigned nw) {
    if (c->done) return;
    if (MODE == 0) return;
    for (unsigned w = blockIdx.x * 256 + threadIdx.x; w < nw; w += gridDim.x * 256) bm[w] = 0ull;
    if (MODE == 1) return;
    __shared__ unsigned s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (SLOT) {
            unsigned long long r = atomicAdd(&c->slot[0][blockIdx.x & 63], 1ull);
            asm volatile("" ::"v"(r));
        }
        const unsigned s = blockIdx.x & 63u;
        const unsigned expect = (gridDim.x + 63u - s) >> 6;
        bool last = false;
        if (atomicAdd(&c->tick[s * PAD], 1u) + 1u == expect) {
            c->tick[s * PAD] = 0;
            if (THREE) {
                if (atomicAdd(&c->tick_mid[(s >> 3) * PAD], 1u) + 1u == 8u) {
                    c->tick_mid[(s >> 3) * PAD] = 0;
                    if (atomicAdd(&c->tick_top, 1u) + 1u == 8u) { c->tick_top = 0; last = true; }
                }
            } else if (atomicAdd(&c->tick_top, 1u) + 1u == 64u) { c->tick_top = 0; last = true; }
        }
        s_last = last;
    }
    __syncthreads();
    if (MODE == 2) return;
    if (s_last && threadIdx.x < 64) {
        unsigned long long v[4];
        for (int k = 0; k < 4; ++k) {
            v[k] = __hip_atomic_load(&c->slot[k][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v[k]) c->slot[k][threadIdx.x] = 0;
            for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d, 64);
        }
        if (threadIdx.x == 0) for (int k = 0; k < 4; ++k) c->out[k] += v[k];
    }
}

template <int MODE, int PAD = 1, bool THREE = false, bool SLOT = true>
static void run(const char* name, Ctrl* c, unsigned long long* bm, unsigned nw, int grid) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((level<MODE, PAD, THREE, SLOT>), dim3(grid), dim3(256), 0, 0, c, bm, nw);
    (void)hipEventRecord(a);
    const int it = 200;
    for (int k = 0; k < it; ++k) hipLaunchKernelGGL((level<MODE, PAD, THREE, SLOT>), dim3(grid), dim3(256), 0, 0, c, bm, nw);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%-64s grid %5d  %6.2f us per launch\n", name, grid, ms / it * 1e3);
}

int main() {
    Ctrl* c; unsigned long long* bm;
    const unsigned nw = 1u << 16;
    (void)hipMalloc(&c, sizeof(Ctrl)); (void)hipMalloc(&bm, nw * 8);
    (void)hipMemset(c, 0, sizeof(Ctrl));
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    for (int per : {7, 1}) {
        const int grid = cus * per;
        run<0>("return at once", c, bm, nw, grid);
        run<1>("zero 512 KiB", c, bm, nw, grid);
        run<2>("zero + slot atomic + two-level ticket", c, bm, nw, grid);
        run<3>("zero + ticket + control wave (4 x 64 slot loads, reduce, store)", c, bm, nw, grid);
        run<2, 1, false, false>("ticket only (no slot atomic), counters packed", c, bm, nw, grid);
        run<2, 16, false, false>("ticket only, one counter per 64 B", c, bm, nw, grid);
        run<2, 32, false, false>("ticket only, one counter per 128 B", c, bm, nw, grid);
        run<2, 32, true, false>("ticket only, 128 B apart, three levels (64 / 8 / 1)", c, bm, nw, grid);
        run<2, 32, true, true>("slot atomic + three-level ticket, 128 B apart", c, bm, nw, grid);
    }
    return 0;
}
