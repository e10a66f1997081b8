// Which load policy keeps a HOT set of 128-byte rows resident in the XCD L2s while COLD rows of the same table stream
// through?  The k-hop pull (bitexpand.hip bp_pull_kernel) gathers one 128 B row of X per entry of A'; ~40-50 % of the
// references go to a few 10^4 hub rows, the rest to millions of cold rows that evict them under LRU (measured L2 hit
// rate 0.22).  This micro reproduces that stream: M references, a fraction p_hot of them uniform over the first H rows,
// the others uniform over the rest of a 2 GiB table; 16 lanes per row, 8 gathers in flight per lane (the pull's shape).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/gatherpolicy.hip -o tools/micro/gatherpolicy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ u64 mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// bit 31 of an index = "hot" (static class, as a flag bit in a private copy of the column ids would carry it)
__global__ void gen_kernel(u32* idx, size_t m, u32 hot_rows, u32 nrows, u32 p_hot_1024, u32 seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
        const u64 h = mix64(i * 0x9E3779B97F4A7C15ull + seed);
        const bool hot = (h & 1023u) < p_hot_1024;
        const u32 r = hot ? (u32)((h >> 10) % hot_rows) : hot_rows + (u32)((h >> 10) % (nrows - hot_rows));
        idx[i] = r | (hot ? 0x80000000u : 0u);
    }
}

enum { P_PLAIN = 0, P_NT = 1, P_SC1 = 2, P_SYS = 3, P_SC1NT = 4 };

// one 8-byte load per lane whose cache policy differs BY LANE: lanes in `hotmask` load with the default policy, the
// others with POL.  (A `cond ? plain_load : __builtin_nontemporal_load` is merged by hipcc into one plain load.)  The
// result is in flight when this returns: wait_loads() before use.
template <int POL>
__device__ __forceinline__ void ld_mixed(u64& d, const u64* p, u64 hotmask) {
    u64 sv;
    if (POL == P_PLAIN)
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
    else if (POL == P_NT)
        asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, %1, %3\n global_load_dwordx2 %0, %2, off\n"
                     "s_andn2_b64 exec, %1, %3\n global_load_dwordx2 %0, %2, off nt\n s_mov_b64 exec, %1"
                     : "=&v"(d), "=&s"(sv) : "v"(p), "s"(hotmask) : "memory");
    else if (POL == P_SC1)
        asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, %1, %3\n global_load_dwordx2 %0, %2, off\n"
                     "s_andn2_b64 exec, %1, %3\n global_load_dwordx2 %0, %2, off sc1\n s_mov_b64 exec, %1"
                     : "=&v"(d), "=&s"(sv) : "v"(p), "s"(hotmask) : "memory");
    else if (POL == P_SYS)
        asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, %1, %3\n global_load_dwordx2 %0, %2, off\n"
                     "s_andn2_b64 exec, %1, %3\n global_load_dwordx2 %0, %2, off sc0 sc1\n s_mov_b64 exec, %1"
                     : "=&v"(d), "=&s"(sv) : "v"(p), "s"(hotmask) : "memory");
    else
        asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, %1, %3\n global_load_dwordx2 %0, %2, off\n"
                     "s_andn2_b64 exec, %1, %3\n global_load_dwordx2 %0, %2, off sc1 nt\n s_mov_b64 exec, %1"
                     : "=&v"(d), "=&s"(sv) : "v"(p), "s"(hotmask) : "memory");
}

// COLD = policy of the cold-row gathers, IDXNT = the index stream is loaded non-temporally, SPLIT = hot rows come from
// their own compact table `xh` (row = idx), cold rows from `x`
template <int COLD, bool IDXNT, bool SPLIT>
__global__ __launch_bounds__(256) void gather_kernel(const u32* __restrict__ idx, size_t m, const u64* __restrict__ x,
                                                     const u64* __restrict__ xh, u64* __restrict__ out) {
    constexpr int LN = 16, SLOTS = 4, G = 8;
    const u32 lane = threadIdx.x & 63u;
    const u32 wl = lane % LN, slot = lane / LN;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    u64 acc = 0;
    // a wave takes "items" of 32 references (one round of G x SLOTS gathers), like an item of ~30 entries of A'
    for (size_t it = wave; it * 32 < m; it += nwaves) {
        const size_t q0 = it * 32;
        u32 u[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const size_t q = q0 + k * SLOTS + slot;
            u[k] = q < m ? (IDXNT ? __builtin_nontemporal_load(&idx[q]) : idx[q]) : 0xFFFFFFFFu;
        }
        u64 xv[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const bool valid = u[k] != 0xFFFFFFFFu;
            const bool hot = valid && (u[k] >> 31) != 0;
            const u32 r = valid ? (u[k] & 0x7FFFFFFFu) : 0u;
            const u64* p = ((SPLIT && hot) ? xh : x) + (size_t)r * LN + wl;
            ld_mixed<COLD>(xv[k], p, __ballot(hot));
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]), "+v"(xv[5]),
                     "+v"(xv[6]), "+v"(xv[7]));
#pragma unroll
        for (int k = 0; k < G; ++k) acc |= (u[k] != 0xFFFFFFFFu) ? xv[k] : 0ull;
    }
    if (acc == 0x123456789ull) out[0] = acc;
}


// ---- v2: items sorted hot-first (as the hub-first order of A' would give), policy chosen per INSTRUCTION -------------
// idx2: per item of 32 references, the hot ones first; nh[item] = number of hot references.
__global__ void gen_sorted_kernel(u32* idx, u32* nh, size_t nitems, u32 hot_rows, u32 nrows, u32 p_hot_1024, u32 seed) {
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < nitems; it += (size_t)gridDim.x * blockDim.x) {
        u32 h = 0, c = 0;
        u32 hot[32], cold[32];
        for (u32 j = 0; j < 32; ++j) {
            const u64 r = mix64((it * 32 + j) * 0x9E3779B97F4A7C15ull + seed);
            if ((r & 1023u) < p_hot_1024) hot[h++] = (u32)((r >> 10) % hot_rows);
            else cold[c++] = hot_rows + (u32)((r >> 10) % (nrows - hot_rows));
        }
        for (u32 j = 0; j < h; ++j) idx[it * 32 + j] = hot[j];
        for (u32 j = 0; j < c; ++j) idx[it * 32 + h + j] = cold[j];
        nh[it] = h;
    }
}

template <int B> struct VecT;
template <> struct VecT<8> { typedef u64 T; };
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
template <> struct VecT<16> { typedef u32x4 T; };

template <int B, bool NT>
__device__ __forceinline__ void ld_async(typename VecT<B>::T& d, const void* p) {
    if (B == 8) {
        if (NT) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(d) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
    } else {
        if (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(d) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
    }
}

// B bytes per lane (LN = 128 / B lanes per row, SLOTS = 64 / LN rows per instruction), G instructions per round;
// POLICY 0: all plain; 1: an instruction whose rows are all hot is plain, every other one nt; 2: all nt
template <int B, int G, int POLICY>
__global__ __launch_bounds__(256) void gather2_kernel(const u32* __restrict__ idx, const u32* __restrict__ nh, size_t nitems,
                                                      const char* __restrict__ x, u64* __restrict__ out) {
    constexpr int LN = 128 / B, SLOTS = 64 / LN;
    constexpr int PER_ROUND = G * SLOTS;            // references per round
    constexpr int ROUNDS = 32 / PER_ROUND > 0 ? 32 / PER_ROUND : 1;
    constexpr int ITEMS = PER_ROUND > 32 ? PER_ROUND / 32 : 1;   // items per round when a round covers more than one
    typedef typename VecT<B>::T V;
    const u32 lane = threadIdx.x & 63u;
    const u32 wl = lane % LN, slot = lane / LN;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    u64 acc = 0;
    for (size_t it = wave * ITEMS; it < nitems; it += nwaves * ITEMS) {
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const size_t q0 = it * 32 + (size_t)rd * PER_ROUND;
            u32 u[G];
#pragma unroll
            for (int k = 0; k < G; ++k) u[k] = __builtin_nontemporal_load(&idx[q0 + k * SLOTS + slot]);
            V xv[G];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const char* p = x + (size_t)u[k] * 128 + wl * B;
                bool nt = POLICY == 2;
                if (POLICY == 1) {
                    const size_t qi = q0 + (size_t)k * SLOTS;                       // first reference of this instruction
                    const u32 hcount = (u32)__builtin_amdgcn_readfirstlane((int)nh[qi >> 5]);
                    nt = ((qi & 31) + SLOTS) > hcount;                               // (wave-uniform)
                }
                if (nt) ld_async<B, true>(xv[k], p); else ld_async<B, false>(xv[k], p);
            }
            if (G == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]));
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4 % G]), "+v"(xv[5 % G]), "+v"(xv[6 % G]), "+v"(xv[7 % G]));
#pragma unroll
            for (int k = 0; k < G; ++k) {
                if (B == 8) acc |= *reinterpret_cast<u64*>(&xv[k]);
                else { const u32x4 t = *reinterpret_cast<u32x4*>(&xv[k]); acc |= ((u64)(t.x | t.z) << 32) | (t.y | t.w); }
            }
        }
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

template <int B, int G, int POLICY>
static float run2(const u32* idx, const u32* nh, size_t nitems, const u64* x, u64* out, int grid) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((gather2_kernel<B, G, POLICY>), dim3(grid), dim3(256), 0, 0, idx, nh, nitems, (const char*)x, out);
    (void)hipEventRecord(a);
    const int it = 3;
    for (int k = 0; k < it; ++k)
        hipLaunchKernelGGL((gather2_kernel<B, G, POLICY>), dim3(grid), dim3(256), 0, 0, idx, nh, nitems, (const char*)x, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / it;
}

struct Cfg { const char* name; int id; };

template <int COLD, bool IDXNT, bool SPLIT>
static float run(const u32* idx, size_t m, const u64* x, const u64* xh, u64* out, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather_kernel<COLD, IDXNT, SPLIT>), dim3(grid), dim3(256), 0, 0, idx, m, x, xh, out);
    hipEventRecord(a);
    const int it = 3;
    for (int k = 0; k < it; ++k)
        hipLaunchKernelGGL((gather_kernel<COLD, IDXNT, SPLIT>), dim3(grid), dim3(256), 0, 0, idx, m, x, xh, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / it;
}

int main(int argc, char** argv) {
    const u32 nrows = argc > 1 ? (u32)atoll(argv[1]) : (1u << 24);
    const size_t m = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)1 << 27;   // 134 M references
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    u64 *x, *xh, *xu, *out; u32* idx;
    hipMalloc(&x, (size_t)nrows * 128);
    hipMalloc(&xh, (size_t)(1u << 18) * 128);
    hipMalloc(&out, 8);
    hipMalloc(&idx, m * 4);
    hipMemset(x, 1, (size_t)nrows * 128);
    hipMemset(xh, 1, (size_t)(1u << 18) * 128);
    // an uncached (MTYPE UC) copy of the table: cold rows read from it never allocate in L2
    hipError_t eu = hipExtMallocWithFlags((void**)&xu, (size_t)nrows * 128, hipDeviceMallocUncached);
    if (eu != hipSuccess) { xu = nullptr; printf("uncached alloc failed: %s\n", hipGetErrorString(eu)); }
    else hipMemset(xu, 1, (size_t)nrows * 128);
    printf("%s: %d CUs; table %u rows x 128 B, %zu references per launch (= %.2f GB of row gathers)\n", pr.gcnArchName, cus,
           nrows, m, m * 128.0 / 1e9);
    const int grid = cus * 32;
    if (argc > 3) {   // v2: sorted items, per-instruction policy
        u32* nh; hipMalloc(&nh, (m / 32 + 1) * 4);
        const size_t nitems = m / 32;
        for (u32 hot_rows : {16384u, 32768u}) {
            for (u32 p : {410u, 512u}) {
                hipLaunchKernelGGL(gen_sorted_kernel, dim3(cus * 8), dim3(256), 0, 0, idx, nh, nitems, hot_rows, nrows, p, 777u);
                hipDeviceSynchronize();
                printf("v2 hot set %u rows (%.1f MB), p_hot %.2f:\n", hot_rows, hot_rows * 128.0 / 1e6, p / 1024.0);
                for (int g : {cus * 32, cus * 16, cus * 8}) {
                    struct { const char* name; float ms; } r[] = {
                        {"8 B/lane  G=8 plain", run2<8, 8, 0>(idx, nh, nitems, x, out, g)},
                        {"8 B/lane  G=8 hot plain / rest nt", run2<8, 8, 1>(idx, nh, nitems, x, out, g)},
                        {"8 B/lane  G=8 all nt", run2<8, 8, 2>(idx, nh, nitems, x, out, g)},
                        {"16 B/lane G=4 plain", run2<16, 4, 0>(idx, nh, nitems, x, out, g)},
                        {"16 B/lane G=4 hot plain / rest nt", run2<16, 4, 1>(idx, nh, nitems, x, out, g)},
                        {"16 B/lane G=4 all nt", run2<16, 4, 2>(idx, nh, nitems, x, out, g)},
                        {"16 B/lane G=8 plain (2 items / round)", run2<16, 8, 0>(idx, nh, nitems, x, out, g)},
                        {"16 B/lane G=8 hot plain / rest nt", run2<16, 8, 1>(idx, nh, nitems, x, out, g)},
                    };
                    for (auto& e : r)
                        printf("  grid %5d  %-40s %8.3f ms  %7.1f GB/s of rows\n", g, e.name, e.ms, m * 128.0 / 1e6 / e.ms);
                }
            }
        }
        return 0;
    }
    for (u32 hot_rows : {16384u, 32768u, 65536u}) {
        for (u32 p : {410u, 512u, 717u}) {
            hipLaunchKernelGGL(gen_kernel, dim3(cus * 8), dim3(256), 0, 0, idx, m, hot_rows, nrows, p, 12345u);
            hipDeviceSynchronize();
            printf("hot set %u rows (%.1f MB), p_hot %.2f:\n", hot_rows, hot_rows * 128.0 / 1e6, p / 1024.0);
            struct { const char* name; float ms; } r[] = {
                {"plain / plain idx", run<P_PLAIN, false, false>(idx, m, x, xh, out, grid)},
                {"plain / nt idx", run<P_PLAIN, true, false>(idx, m, x, xh, out, grid)},
                {"cold nt / nt idx", run<P_NT, true, false>(idx, m, x, xh, out, grid)},
                {"cold sc1 / nt idx", run<P_SC1, true, false>(idx, m, x, xh, out, grid)},
                {"cold sys / nt idx", run<P_SYS, true, false>(idx, m, x, xh, out, grid)},
                {"cold sc1 nt / nt idx", run<P_SC1NT, true, false>(idx, m, x, xh, out, grid)},
                {"cold nt, hot table / nt idx", run<P_NT, true, true>(idx, m, x, xh, out, grid)},
                {"cold from UNCACHED alloc, hot table / nt idx", xu ? run<P_PLAIN, true, true>(idx, m, xu, xh, out, grid) : 0.f},
                {"cold nt from UNCACHED alloc, hot table / nt idx", xu ? run<P_NT, true, true>(idx, m, xu, xh, out, grid) : 0.f},
            };
            for (auto& e : r)
                printf("  %-52s %8.3f ms  %7.1f GB/s of rows  (cold-only bytes at this time: %6.1f GB/s)\n", e.name, e.ms,
                       m * 128.0 / 1e6 / e.ms, m * 128.0 * (1.0 - p / 1024.0) / 1e6 / e.ms);
        }
    }
    return 0;
}
