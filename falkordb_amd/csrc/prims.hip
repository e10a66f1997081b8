// prims.hip — device primitives shared by the matrix builders and the ANY_PAIR
// products: exclusive scans, per-segment sort+unique, segment compaction.
//
// segsort_unique is the engine's replacement for what GraphBLAS does inside
// GrB_Matrix_wait / the saxpy3 "sort the jumbled row" step (reference call sites:
// Matrix::wait matrix.rs:781-796, Matrix::build matrix.rs:1281-1303, lmxm
// matrix.rs:930-947): every row of a product or of a COO build ends up with its
// column ids ascending and unique.  Three size classes, all wave64-native:
//   <= 64 keys   one wavefront per segment, bitonic network through DPP/ds_bpermute
//   <= 4096 keys one 256-thread workgroup, bitonic network in LDS
//   larger       hierarchical global bitmap (key_bound bits) — sorted order falls
//                out of the bit positions; HBM capacity (288 GB) makes this cheap.
#include "common.hpp"

namespace fgpu {

// ---------------------------------------------------------------------------------
// scans
// ---------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_SUBTILES = 16;                        // 256 * 16 = 4096 items per block
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_SUBTILES;

template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T x) {
    const u32 lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T y = __shfl_up(x, d, 64);
        if (lane >= (u32)d) x += y;
    }
    return x;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive
// prefix, *total receives the block sum.  s_wave must hold 4 entries.
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T x, T* s_wave, T* total) {
    const u32 lane = lane_id();
    const u32 w = threadIdx.x >> 6;
    T inc = wave_inclusive_scan(x);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_THREADS / 64; ++i) {
        T v = s_wave[i];
        if ((u32)i < w) base += v;
        tot += v;
    }
    __syncthreads();
    *total = tot;
    return base + inc - x;
}

template <typename Tin, typename Tout>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const Tin* __restrict__ in, u64 n,
                                                                  Tout* __restrict__ block_sums) {
    __shared__ Tout s_wave[SCAN_THREADS / 64];
    const u64 base = (u64)blockIdx.x * SCAN_TILE;
    Tout acc = 0;
#pragma unroll 4
    for (int j = 0; j < SCAN_SUBTILES; ++j) {
        u64 i = base + (u64)j * SCAN_THREADS + threadIdx.x;
        if (i < n) acc += (Tout)in[i];
    }
    // wave reduce
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane_id() == 0) s_wave[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Tout t = 0;
        for (int i = 0; i < SCAN_THREADS / 64; ++i) t += s_wave[i];
        block_sums[blockIdx.x] = t;
    }
}

template <typename Tin, typename Tout>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const Tin* __restrict__ in,
                                                                 Tout* __restrict__ out, u64 n,
                                                                 const Tout* __restrict__ block_prefix,
                                                                 Tout* __restrict__ total_out) {
    __shared__ Tout s_wave[SCAN_THREADS / 64];
    const u64 base = (u64)blockIdx.x * SCAN_TILE;
    Tout carry = block_prefix ? block_prefix[blockIdx.x] : (Tout)0;
    for (int j = 0; j < SCAN_SUBTILES; ++j) {
        u64 i = base + (u64)j * SCAN_THREADS + threadIdx.x;
        if (base + (u64)j * SCAN_THREADS >= n) break;  // uniform
        Tout x = (i < n) ? (Tout)in[i] : (Tout)0;
        Tout tot;
        Tout ex = block_exclusive_scan<Tout>(x, s_wave, &tot);
        if (i < n) out[i] = carry + ex;
        carry += tot;
    }
    if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = carry;
}

template <typename Tin, typename Tout>
static fgpu_info scan_impl(fgpu_ctx* ctx, const Tin* in, Tout* out, u64 n, Tout* total_dev) {
    if (n == 0) {
        if (total_dev) FGPU_HIP(hipMemsetAsync(total_dev, 0, sizeof(Tout), ctx->stream()));
        return FGPU_OK;
    }
    u32 nblocks = cdiv(n, SCAN_TILE);
    if (nblocks == 1) {
        hipLaunchKernelGGL((scan_apply_kernel<Tin, Tout>), dim3(1), dim3(SCAN_THREADS), 0, ctx->stream(), in, out,
                           n, (const Tout*)nullptr, total_dev);
        FGPU_HIP(hipGetLastError());
        return FGPU_OK;
    }
    DevBuf<Tout> sums;
    FGPU_TRY(sums.alloc(ctx, nblocks));
    hipLaunchKernelGGL((scan_reduce_kernel<Tin, Tout>), dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream(), in,
                       n, sums.p);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY((scan_impl<Tout, Tout>(ctx, sums.p, sums.p, nblocks, (Tout*)nullptr)));
    hipLaunchKernelGGL((scan_apply_kernel<Tin, Tout>), dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream(), in,
                       out, n, (const Tout*)sums.p, total_dev);
    FGPU_HIP(hipGetLastError());
    return FGPU_OK;
}

fgpu_info scan_u32(fgpu_ctx* ctx, const u32* in, u32* out, u64 n, u32* total_dev) {
    return scan_impl<u32, u32>(ctx, in, out, n, total_dev);
}
fgpu_info scan_u32_to_u64(fgpu_ctx* ctx, const u32* in, u64* out, u64 n, u64* total_dev) {
    return scan_impl<u32, u64>(ctx, in, out, n, total_dev);
}

// ---------------------------------------------------------------------------------
// segsort_unique
// ---------------------------------------------------------------------------------
constexpr u32 SEG_WAVE_MAX = 64;
constexpr u32 SEG_BLOCK_MAX = 4096;
constexpr u32 KEY_INF = 0xFFFFFFFFu;

// One wavefront per segment.  Segments longer than 64 keys are appended to the
// mid / big work lists for the following kernels.
__global__ __launch_bounds__(256) void segsort_wave_kernel(u32* __restrict__ data, const u64* __restrict__ off,
                                                          u32 nseg, u32* __restrict__ cnt,
                                                          u32* __restrict__ mid_list, u32* __restrict__ big_list,
                                                          u32* __restrict__ list_counts,
                                                          const uint8_t* __restrict__ dirty, u32 seg_base) {
    const u32 lane = lane_id();
    const u32 seg = seg_base + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    if (dirty && !dirty[seg]) return;  // caller pre-filled cnt[seg]; keys already sorted unique
    const u64 b = off[seg];
    const u64 len64 = off[seg + 1] - b;
    if (len64 > SEG_WAVE_MAX) {
        if (lane == 0) {
            if (len64 <= SEG_BLOCK_MAX) mid_list[atomicAdd(&list_counts[0], 1u)] = seg;
            else big_list[atomicAdd(&list_counts[1], 1u)] = seg;
        }
        return;
    }
    const u32 len = (u32)len64;
    if (len <= 1) {
        if (lane == 0) cnt[seg] = len;
        return;
    }
    u32 x = (lane < len) ? data[b + lane] : KEY_INF;
    // bitonic sort across the 64 lanes
#pragma unroll
    for (u32 k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (u32 j = k >> 1; j >= 1; j >>= 1) {
            u32 y = __shfl_xor(x, j, 64);
            bool up = ((lane & k) == 0);
            bool lower = ((lane & j) == 0);
            u32 mn = x < y ? x : y, mx = x < y ? y : x;
            x = (lower == up) ? mn : mx;
        }
    }
    u32 prev = __shfl_up(x, 1, 64);
    bool keep = (lane < len) && (lane == 0 || x != prev);
    // note: KEY_INF is a legal key only if key_bound == 2^32, which the ABI excludes
    u64 mask = __ballot(keep);
    if (keep) {
        u32 pos = __popcll(mask & ((1ull << lane) - 1ull));
        data[b + pos] = x;
    }
    if (lane == 0) cnt[seg] = (u32)__popcll(mask);
}

// One workgroup per mid segment (65..4096 keys): bitonic sort in LDS.
__global__ __launch_bounds__(256) void segsort_block_kernel(u32* __restrict__ data, const u64* __restrict__ off,
                                                           const u32* __restrict__ mid_list,
                                                           const u32* __restrict__ list_counts,
                                                           u32* __restrict__ cnt) {
    __shared__ u32 s[SEG_BLOCK_MAX];
    __shared__ u32 s_wave[4];
    const u32 nmid = list_counts[0];
    for (u32 it = blockIdx.x; it < nmid; it += gridDim.x) {
        const u32 seg = mid_list[it];
        const u64 b = off[seg];
        const u32 len = (u32)(off[seg + 1] - b);
        u32 p2 = 128;
        while (p2 < len) p2 <<= 1;
        for (u32 i = threadIdx.x; i < p2; i += 256) s[i] = (i < len) ? data[b + i] : KEY_INF;
        __syncthreads();
        for (u32 k = 2; k <= p2; k <<= 1) {
            for (u32 j = k >> 1; j >= 1; j >>= 1) {
                for (u32 t = threadIdx.x; t < (p2 >> 1); t += 256) {
                    u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
                    u32 l = i | j;
                    u32 a = s[i], c = s[l];
                    bool up = ((i & k) == 0);
                    if ((a > c) == up) { s[i] = c; s[l] = a; }
                }
                __syncthreads();
            }
        }
        // unique + compaction, 256 keys per round
        u32 outpos = 0;
        for (u32 base = 0; base < len; base += 256) {
            u32 i = base + threadIdx.x;
            u32 x = (i < len) ? s[i] : KEY_INF;
            bool keep = (i < len) && (i == 0 || s[i - 1] != x);
            u32 tot;
            u32 ex = block_exclusive_scan<u32>(keep ? 1u : 0u, s_wave, &tot);
            if (keep) data[b + outpos + ex] = x;
            outpos += tot;
        }
        if (threadIdx.x == 0) cnt[seg] = outpos;
        __syncthreads();
    }
}

// Big segments: scatter bits into a per-slot bitmap of `words` u32 words...
__global__ __launch_bounds__(256) void segsort_big_scatter_kernel(const u32* __restrict__ data,
                                                                 const u64* __restrict__ off,
                                                                 const u32* __restrict__ big_list, u32 first,
                                                                 u32 nslots, u32* __restrict__ bitmaps,
                                                                 u64 words) {
    const u32 slot = blockIdx.y;
    if (slot >= nslots) return;
    const u32 seg = big_list[first + slot];
    const u64 b = off[seg];
    const u64 len = off[seg + 1] - b;
    u32* bm = bitmaps + (u64)slot * words;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < len; i += (u64)gridDim.x * 256) {
        u32 k = data[b + i];
        u32 w = k >> 5, bit = 1u << (k & 31);
        if ((bm[w] & bit) == 0) atomicOr(&bm[w], bit);
    }
}

// ... then one 1024-thread workgroup per slot turns the bitmap back into an ascending
// key list (two sweeps: count per wave, then emit) and clears the words it used.
__global__ __launch_bounds__(1024) void segsort_big_emit_kernel(u32* __restrict__ data, const u64* __restrict__ off,
                                                               const u32* __restrict__ big_list, u32 first,
                                                               u32* __restrict__ bitmaps, u64 words,
                                                               u32* __restrict__ cnt) {
    __shared__ u32 s_wave_tot[16];
    const u32 slot = blockIdx.x;
    const u32 seg = big_list[first + slot];
    const u64 b = off[seg];
    u32* bm = bitmaps + (u64)slot * words;
    const u32 lane = lane_id();
    const u32 w = threadIdx.x >> 6;
    // contiguous word range per wave
    const u64 per = (words + 15) / 16;
    const u64 w0 = (u64)w * per;
    const u64 w1 = (w0 + per < words) ? (w0 + per) : words;
    u32 total = 0;
    for (u64 i = w0 + lane; i < w1; i += 64) total += __popc(bm[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d, 64);
    if (lane == 0) s_wave_tot[w] = total;
    __syncthreads();
    u32 base = 0, all = 0;
    for (u32 i = 0; i < 16; ++i) {
        u32 v = s_wave_tot[i];
        if (i < w) base += v;
        all += v;
    }
    for (u64 i0 = w0; i0 < w1; i0 += 64) {
        u64 i = i0 + lane;
        u32 word = (i < w1) ? bm[i] : 0u;
        u32 c = __popc(word);
        u32 inc = wave_inclusive_scan<u32>(c);
        u32 pos = base + inc - c;
        base += __shfl(inc, 63, 64);
        if (word) {
            bm[i] = 0u;
            u32 key0 = (u32)(i << 5);
            while (word) {
                u32 t = __builtin_ctz(word);
                word &= word - 1;
                data[b + pos++] = key0 + t;
            }
        }
    }
    if (threadIdx.x == 0) cnt[seg] = all;
}

fgpu_info segsort_unique(fgpu_ctx* ctx, u32* data, const u64* off, u32 nseg, u32 key_bound, u32* cnt,
                         const uint8_t* dirty) {
    if (nseg == 0) return FGPU_OK;
    DevBuf<u32> mid, big, counts;
    FGPU_TRY(mid.alloc(ctx, nseg));
    FGPU_TRY(big.alloc(ctx, nseg));
    FGPU_TRY(counts.alloc(ctx, 2));
    FGPU_HIP(hipMemsetAsync(counts.p, 0, 2 * sizeof(u32), ctx->stream()));
    // a wavefront per segment: 2^26 rows (RMAT-26) would be 2^32 threads in one grid, past what a launch accepts
    for (u64 base = 0; base < nseg; base += (1ull << 24)) {
        const u32 part = (u32)((nseg - base < (1ull << 24)) ? nseg - base : (1ull << 24));
        hipLaunchKernelGGL(segsort_wave_kernel, dim3(cdiv(part, 4)), dim3(256), 0, ctx->stream(), data, off, nseg, cnt,
                           mid.p, big.p, counts.p, dirty, (u32)base);
        FGPU_HIP(hipGetLastError());
    }
    // mid segments: grid-stride over the device-side list, no host round trip
    {
        u32 grid = ctx->cus * 8;
        if (grid > nseg) grid = nseg;
        hipLaunchKernelGGL(segsort_block_kernel, dim3(grid), dim3(256), 0, ctx->stream(), data, off, mid.p, counts.p,
                           cnt);
        FGPU_HIP(hipGetLastError());
    }
    // big segments need their count on the host to size the bitmap workspace
    u32 nbig = 0;
    FGPU_TRY(read_u32(ctx, counts.p + 1, &nbig));
    if (nbig) {
        const u64 words = ((u64)key_bound + 31) / 32;
        const u64 budget_words = (1ull << 30) / 4;  // 1 GiB of bitmaps in flight
        u32 slots = (u32)(budget_words / (words ? words : 1));
        if (slots < 1) slots = 1;
        if (slots > nbig) slots = nbig;
        if (slots > 65535) slots = 65535;
        DevBuf<u32> bitmaps;
        FGPU_TRY(bitmaps.alloc(ctx, (size_t)slots * words));
        FGPU_HIP(hipMemsetAsync(bitmaps.p, 0, (size_t)slots * words * sizeof(u32), ctx->stream()));
        for (u32 first = 0; first < nbig; first += slots) {
            u32 ns = (nbig - first < slots) ? (nbig - first) : slots;
            hipLaunchKernelGGL(segsort_big_scatter_kernel, dim3(64, ns), dim3(256), 0, ctx->stream(), data, off,
                               big.p, first, ns, bitmaps.p, words);
            FGPU_HIP(hipGetLastError());
            hipLaunchKernelGGL(segsort_big_emit_kernel, dim3(ns), dim3(1024), 0, ctx->stream(), data, off, big.p,
                               first, bitmaps.p, words, cnt);
            FGPU_HIP(hipGetLastError());
        }
    }
    return FGPU_OK;
}

// ---------------------------------------------------------------------------------
// compaction of the unique prefixes into a dense CSR
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compact_segments_kernel(const u32* __restrict__ data,
                                                              const u64* __restrict__ off,
                                                              const u32* __restrict__ rowptr, u32 nseg,
                                                              u32* __restrict__ col_out) {
    // one wavefront per segment, grid-stride
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 seg = wave; seg < nseg; seg += nwaves) {
        const u64 b = off[seg];
        const u32 o = rowptr[seg];
        const u32 c = rowptr[seg + 1] - o;
        for (u32 i = lane; i < c; i += 64) col_out[o + i] = data[b + i];
    }
}

fgpu_info compact_segments(fgpu_ctx* ctx, const u32* data, const u64* off, const u32* rowptr, u32 nseg,
                           u32* col_out) {
    // rowptr = exclusive scan of the per-segment unique counts (caller did the scan to size col_out)
    if (nseg == 0) return FGPU_OK;
    u32 grid = cdiv(nseg, 4);
    u32 cap = ctx->cus * 16;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(compact_segments_kernel, dim3(grid), dim3(256), 0, ctx->stream(), data, off, rowptr, nseg,
                       col_out);
    FGPU_HIP(hipGetLastError());
    return FGPU_OK;
}

}  // namespace fgpu
