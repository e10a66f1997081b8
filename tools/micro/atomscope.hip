// Does a push DISCOVERY (returning atomicOr on visited, atomicOr on next, a 4-byte level store, a row-pointer pair read) get
// cheaper when the destination words are OWNED by the XCD that updates them, so that its atomics may stay in that XCD's L2
// (workgroup scope) instead of going to the memory side (agent scope: ~26 G/s chip-wide, tools/micro/atomicbw.hip)?
// Every workgroup reads the XCD it RUNS on (HW_REG_XCC_ID) and only draws destinations from that XCD's eighth of the vertex
// range; the eight L2s therefore never hold the same bitmap line.  The bitmaps are compared with an agent-scope run of the
// same draws afterwards: a lost update shows as a differing word.
//   mode 0: agent-scope atomics, destinations over the whole range            (what bfs.hip push_fused does today)
//   mode 1: agent-scope atomics, destinations in the own XCD's eighth
//   mode 2: workgroup-scope atomics, destinations in the own XCD's eighth
//   mode 3: as 2, plus the level store and the row-pointer pair read of a discovery
//   mode 4: as 0, plus the level store and the row-pointer pair read
//   mode 5: as 3 with plain load-modify-store instead of atomics is NOT measured: two workgroups of one XCD may race
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/atomscope.hip -o tools/micro/atomscope
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* __restrict__ vis, unsigned* __restrict__ nxt, int* __restrict__ level,
                                         const unsigned* __restrict__ rowptr, unsigned nbits, unsigned per_thread,
                                         unsigned long long* out, unsigned* xcd_of_block) {
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if (threadIdx.x == 0 && xcd_of_block) xcd_of_block[blockIdx.x] = xcc;
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    const unsigned range = nbits / 8;
    unsigned long long cnt = 0, mf = 0;
    for (unsigned i = 0; i < per_thread; i += 4) {
        unsigned u[4], r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned h = mix(tid * per_thread + i + j);
            u[j] = (MODE == 0 || MODE == 4) ? h % nbits : xcc * range + h % range;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned w = u[j] >> 5, bit = 1u << (u[j] & 31);
            if (MODE == 2 || MODE == 3) r[j] = __hip_atomic_fetch_or(&vis[w], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else r[j] = __hip_atomic_fetch_or(&vis[w], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned w = u[j] >> 5, bit = 1u << (u[j] & 31);
            if (r[j] & bit) continue;
            if (MODE == 2 || MODE == 3) __hip_atomic_fetch_or(&nxt[w], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_or(&nxt[w], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE >= 3) {
                level[u[j]] = 7;
                mf += rowptr[u[j] + 1] - rowptr[u[j]];
            }
            ++cnt;
        }
    }
    if (cnt | mf) atomicAdd(out, cnt + (mf << 40));
}

template <int MODE>
static void run(const char* name, unsigned* vis, unsigned* nxt, int* level, unsigned* rowptr, unsigned nbits, unsigned long long* out,
                int grid, unsigned per, std::vector<unsigned>* keep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    unsigned long long found = 0;
    for (int it = 0; it < 4; ++it) {
        hipMemset(vis, 0, nbits / 8); hipMemset(nxt, 0, nbits / 8); hipMemset(out, 0, 8);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, vis, nxt, level, rowptr, nbits, per, out, (unsigned*)nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(&found, out, 8, hipMemcpyDeviceToHost);
    }
    const double ops = (double)grid * 256 * per;
    std::vector<unsigned> hv(nbits / 32), hn(nbits / 32);
    hipMemcpy(hv.data(), vis, nbits / 8, hipMemcpyDeviceToHost);
    hipMemcpy(hn.data(), nxt, nbits / 8, hipMemcpyDeviceToHost);
    unsigned long long pc = 0, diffn = 0;
    for (size_t i = 0; i < hv.size(); ++i) { pc += __builtin_popcount(hv[i]); diffn += hv[i] != hn[i]; }
    printf("%-74s %9.1f us %7.2f G edges/s  discoveries %llu  bits set %llu  vis!=nxt words %llu\n", name, best * 1e3,
           ops / (best * 1e-3) / 1e9, found & ((1ull << 40) - 1), pc, diffn);
    if (keep) *keep = hv;
}

int main(int argc, char** argv) {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    for (unsigned lg : {22u, 26u}) {
        const unsigned nbits = 1u << lg;
        unsigned *vis, *nxt, *rowptr, *xob; int* level; unsigned long long* out;
        hipMalloc(&vis, nbits / 8); hipMalloc(&nxt, nbits / 8); hipMalloc(&out, 8);
        hipMalloc(&level, (size_t)nbits * 4); hipMalloc(&rowptr, ((size_t)nbits + 1) * 4);
        hipMemset(rowptr, 0, ((size_t)nbits + 1) * 4);
        const int grid = cus * 6;
        hipMalloc(&xob, grid * 4);
        // where do the workgroups of a launch run?
        hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, vis, nxt, level, rowptr, nbits, 4u, out, xob);
        std::vector<unsigned> hx(grid);
        hipMemcpy(hx.data(), xob, grid * 4, hipMemcpyDeviceToHost);
        unsigned rr = 0, hist[8] = {0};
        for (int b = 0; b < grid; ++b) { rr += hx[b] == (unsigned)(b & 7); hist[hx[b] & 7]++; }
        printf("== %u-bit bitmaps (%u KiB each), grid %d: workgroups on XCD blockIdx & 7: %u of %d; per XCD %u %u %u %u %u %u %u %u\n", nbits,
               nbits / 8 / 1024, grid, rr, grid, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
        for (unsigned total : {nbits / 4, nbits * 2}) {    // few repeats per bit (a heavy level's discoveries) / many (its edge checks)
            unsigned per = (total / (grid * 256) + 3) & ~3u; if (!per) per = 4;
            printf("-- %u edges over %d x 256 threads (%u per thread)\n", grid * 256 * per, grid, per);
            std::vector<unsigned> ref, got;
            run<0>("0 agent scope, whole range", vis, nxt, level, rowptr, nbits, out, grid, per, nullptr);
            run<1>("1 agent scope, own XCD's eighth", vis, nxt, level, rowptr, nbits, out, grid, per, &ref);
            run<2>("2 workgroup scope, own XCD's eighth", vis, nxt, level, rowptr, nbits, out, grid, per, &got);
            // (the draws of modes 1 and 2 depend on which XCD a workgroup lands on: compare only if the placement is the round-robin
            // one both times — otherwise the popcounts and vis == nxt are the check)
            size_t bad = 0;
            for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != got[i];
            printf("   words differing between modes 1 and 2: %zu\n", bad);
            run<4>("4 agent scope, whole range, + level store + row-pointer pair", vis, nxt, level, rowptr, nbits, out, grid, per, nullptr);
            run<3>("3 workgroup scope, own eighth, + level store + row-pointer pair", vis, nxt, level, rowptr, nbits, out, grid, per, nullptr);
        }
        hipFree(vis); hipFree(nxt); hipFree(out); hipFree(level); hipFree(rowptr); hipFree(xob);
    }
    return 0;
}
