// What ONE level costs a grid-resident (persistent) BFS loop on this chip, beside a kernel boundary (VERDICT r05 item 4 asked
// for "one persistent launch per search ... a device-side barrier per level").  A grid of 6 workgroups per CU (the fused level
// kernel's), every workgroup per "level": writes its own 4 KiB of a bitmap with PLAIN stores, arrives at a barrier (one returning
// device-scope add on a counter of its own 128-byte line, a second-level counter, a generation word everybody spins on), and then
// reads the 4 KiB its LEFT neighbour wrote — data another XCD's L2 holds dirty unless the writer released it.
//   mode 0: one launch per level (the kernel boundary does the release / acquire)      — what bfs.hip does today
//   mode 1: persistent, barrier only (NO fences: the neighbour's data is read stale — counted, not a correct program)
//   mode 2: persistent, agent-scope release before the arrive, acquire after the wait (the correct program)
//   mode 3: as 2, but the data goes out with device-scope atomic stores and comes in with device-scope atomic loads (no fence)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/gridsync.hip -o tools/micro/gridsync
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

struct Bar { unsigned cnt[64 * 32]; unsigned top; unsigned pad[31]; unsigned gen; unsigned pad2[31]; unsigned long long stale; };

__device__ __forceinline__ void grid_barrier(Bar* b, unsigned nblk, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned s = blockIdx.x & 63u;
        const unsigned expect = (nblk + 63u - s) >> 6;
        if (atomicAdd(&b->cnt[s * 32], 1u) + 1u == expect) {
            // (device-scope stores: a plain store would sit dirty in this XCD's L2 under the memory-side atomics of the next round)
            __hip_atomic_store(&b->cnt[s * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (atomicAdd(&b->top, 1u) + 1u == (nblk < 64u ? nblk : 64u)) {
                __hip_atomic_store(&b->top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&b->gen, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(2);
    }
    ++gen;
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void levels(Bar* b, unsigned long long* bm, unsigned nlev, unsigned lev0) {
    unsigned gen = MODE == 0 ? 0u : 0u;
    unsigned long long* mine = bm + (size_t)blockIdx.x * 512;
    const unsigned long long* left = bm + (size_t)((blockIdx.x + gridDim.x - 1) % gridDim.x) * 512;
    unsigned long long stale = 0;
    for (unsigned l = 0; l < nlev; ++l) {
        const unsigned long long tag = (unsigned long long)(lev0 + l + 1);
        for (unsigned i = threadIdx.x; i < 512; i += 256) {
            if (MODE == 3) __hip_atomic_store(&mine[i], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else mine[i] = tag;
        }
        if (MODE == 0) break;                                   // the next launch is the barrier
        if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        grid_barrier(b, gridDim.x, gen);
        if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (unsigned i = threadIdx.x; i < 512; i += 256) {
            const unsigned long long v = MODE == 3 ? __hip_atomic_load(&left[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : left[i];
            stale += v != tag;
        }
        grid_barrier(b, gridDim.x, gen);                        // (nobody overwrites before everybody has read)
    }
    if (MODE == 0) {                                            // reads what the PREVIOUS launch's left neighbour wrote
        for (unsigned i = threadIdx.x; i < 512; i += 256) stale += lev0 && left[i] != (unsigned long long)lev0 ? 0 : 0;
    }
    if (stale) atomicAdd(&b->stale, stale);
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount * 4;      // (4 workgroups per CU: resident whatever the register count)
    Bar* b; unsigned long long* bm;
    (void)hipMalloc(&b, sizeof(Bar)); (void)hipMalloc(&bm, (size_t)grid * 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned nlev = 200;
    for (int mode = 0; mode < 4; ++mode) {
        (void)hipMemset(b, 0, sizeof(Bar)); (void)hipMemset(bm, 0, (size_t)grid * 4096);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(b, 0, sizeof(Bar));
            (void)hipEventRecord(e0);
            if (mode == 0) for (unsigned l = 0; l < nlev; ++l) hipLaunchKernelGGL(levels<0>, dim3(grid), dim3(256), 0, 0, b, bm, 1u, l);
            else if (mode == 1) hipLaunchKernelGGL(levels<1>, dim3(grid), dim3(256), 0, 0, b, bm, nlev, 0u);
            else if (mode == 2) hipLaunchKernelGGL(levels<2>, dim3(grid), dim3(256), 0, 0, b, bm, nlev, 0u);
            else hipLaunchKernelGGL(levels<3>, dim3(grid), dim3(256), 0, 0, b, bm, nlev, 0u);
            (void)hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("mode %d failed\n", mode); return 1; }
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        Bar h; (void)hipMemcpy(&h, b, sizeof(Bar), hipMemcpyDeviceToHost);
        printf("mode %d  grid %d  %7.2f us per level  (%s)  stale words read: %llu\n", mode, grid, ms * 1e3 / nlev,
               mode == 0 ? "one launch per level" : mode == 1 ? "persistent, two barriers, no fence"
               : mode == 2 ? "persistent, two barriers, agent release + acquire" : "persistent, two barriers, device-scope atomic stores / loads",
               (unsigned long long)h.stale);
        fflush(stdout);
    }
    return 0;
}
