// spgemm.hip — ANY_PAIR (structural) sparse products and the CondTraverse expansion core.
//
// Reference call sites (graph/src/graph/graphblas/matrix.rs unless noted):
//   Matrix::lmxm        :930-947   C = F x B, GrB_mxm(GxB_ANY_PAIR_BOOL), no mask
//   Matrix::delta_lmxm  :1317-1402 (F x (m U dp)) with (F x dm) masked out of the F x m part
//   expand_batch        runtime/ops/cond_traverse.rs:452-751 (F build :600-601, chain :602-605,
//                       ascending (row, dest) read-out :644, dst label post-filter :647-651)
//
// The semiring never reads values, so a product row is the sorted union of the B-rows named
// by the F-row.  Device formulation: (1) per F-entry degree + prefix sum = output upper bound
// ("flops"), (2) one wavefront per F-entry copies its B-row coalesced into the row's segment,
// (3) segments with more than one source go through segsort_unique (prims.hip), single-source
// segments (hop 1 of every traversal) are already sorted and unique, (4) compaction to CSR.
#include <algorithm>
#include <string>
#include <thread>

#include "common.hpp"

namespace fgpu {

__global__ void entry_deg_kernel(CsrView f, CsrView b, u32 nnzf, u32* __restrict__ deg) {
    u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnzf) { if (e == nnzf) deg[e] = 0; return; }
    u32 s = f.colidx[e], rb, re;
    row_range(b, s, rb, re);
    deg[e] = re - rb;
}

__global__ void gather_u64_kernel(const u64* __restrict__ eoff, const u32* __restrict__ frp, u32 k,
                                  u64* __restrict__ roff, u32* __restrict__ maxlen) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > k) return;
    roff[i] = eoff[frp[i]];
    if (i < k) {
        u32 len = frp[i + 1] - frp[i];
        u32 m = len;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { u32 o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
        if (lane_id() == 0 && m > 1) atomicMax(maxlen, m);
        else if (lane_id() == 0 && m == 1) atomicMax(maxlen, 1u);
    }
}

// one wavefront per F entry: copy the B row it names into the product row's segment
__global__ __launch_bounds__(256) void gather_rows_kernel(CsrView f, CsrView b, u32 nnzf,
                                                         const u64* __restrict__ eoff, u32* __restrict__ tmp) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 e = wave; e < nnzf; e += nwaves) {
        u32 s = f.colidx[e], rb, re;
        row_range(b, s, rb, re);
        const u64 o = eoff[e];
        u32 i = rb + lane;
        // 4 x 64 per trip keeps several loads in flight for long rows
        for (; i + 192 < re; i += 256) {
            u32 x0 = b.colidx[i], x1 = b.colidx[i + 64], x2 = b.colidx[i + 128], x3 = b.colidx[i + 192];
            u64 p = o + (i - rb);
            tmp[p] = x0; tmp[p + 64] = x1; tmp[p + 128] = x2; tmp[p + 192] = x3;
        }
        for (; i < re; i += 64) tmp[o + (i - rb)] = b.colidx[i];
    }
}

__global__ void seg_len_kernel(const u64* __restrict__ roff, u32 k, u32* __restrict__ cnt) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > k) return;
    cnt[i] = (i < k) ? (u32)(roff[i + 1] - roff[i]) : 0u;
}

// keep entries whose column bit is set in `bitmap`
__global__ __launch_bounds__(256) void bitmap_filter_fill_kernel(CsrView c, const u64* __restrict__ bitmap,
                                                                u32 nrows, u32* __restrict__ tmp,
                                                                u32* __restrict__ cnt) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32* __restrict__ b32 = (const u32*)bitmap;
    for (u32 r = wave; r < nrows; r += nwaves) {
        const u32 rb = c.rowptr[r], re = c.rowptr[r + 1];
        u32 outn = 0;
        for (u32 i0 = rb; i0 < re; i0 += 64) {
            u32 i = i0 + lane, x = 0;
            bool keep = false;
            if (i < re) {
                x = c.colidx[i];
                keep = (b32[x >> 5] >> (x & 31)) & 1u;
            }
            u64 mask = __ballot(keep);
            if (keep) tmp[rb + outn + __popcll(mask & ((1ull << lane) - 1ull))] = x;
            outn += (u32)__popcll(mask);
        }
        if (lane == 0) cnt[r] = outn;
    }
}

__global__ __launch_bounds__(256) void compact_rows_kernel(const u32* __restrict__ tmp, const u32* __restrict__ srcoff,
                                                          const u32* __restrict__ rowptr, u32 nrows,
                                                          u32* __restrict__ col) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 r = wave; r < nrows; r += nwaves) {
        u32 s = srcoff[r], o = rowptr[r], c = rowptr[r + 1] - o;
        for (u32 i = lane; i < c; i += 64) col[o + i] = tmp[s + i];
    }
}

// (rowmap, nullable: row r of c stands for row rowmap[r] of the call — a pass of a whole-frontier call)
__global__ __launch_bounds__(256) void checksum_kernel(CsrView c, u32 nrows, unsigned long long* __restrict__ acc,
                                                       const u32* __restrict__ rowmap) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    u64 sum = 0;
    for (u32 r = wave; r < nrows; r += nwaves) {
        const u32 rb = c.rowptr[r], re = c.rowptr[r + 1];
        const u64 hr = cs_row_hash(rowmap ? rowmap[r] : r);
        for (u32 i = rb + lane; i < re; i += 64) sum += hr * cs_dest_hash(c.colidx[i]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0 && sum) atomicAdd(acc, (unsigned long long)sum);
}

// sum_{(i,s) in F} deg_B(s): the entries a sorted-CSR product would gather (exact, one small pass)
static fgpu_info mxm_flops(fgpu_ctx* ctx, const fgpu_mat* F, const fgpu_mat* B, u64* T) {
    *T = 0;
    const u32 nnzf = (u32)F->nnz;
    if (nnzf == 0 || B->nnz == 0) return FGPU_OK;
    DevBuf<u32> deg;
    DevBuf<u64> eoff, tot;
    FGPU_TRY(deg.alloc(ctx, (size_t)nnzf + 1));
    FGPU_TRY(eoff.alloc(ctx, (size_t)nnzf + 1));
    FGPU_TRY(tot.alloc(ctx, 1));
    hipLaunchKernelGGL(entry_deg_kernel, dim3(cdiv((u64)nnzf + 1, 256)), dim3(256), 0, ctx->stream(), view_of(F),
                       view_of(B), nnzf, deg.p);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(scan_u32_to_u64(ctx, deg.p, eoff.p, (u64)nnzf + 1, tot.p));
    return read_u64(ctx, tot.p, T);
}

// entry-parallel form for results with few, very long rows (dense k-hop results: ~10^6 entries per row):
// each lane takes one entry and finds its row in the short row-pointer array
__global__ __launch_bounds__(256) void checksum_entries_kernel(CsrView c, u32 nrows, u32 nnz,
                                                              unsigned long long* __restrict__ acc, const u32* __restrict__ rowmap) {
    u64 sum = 0;
    for (u32 q = blockIdx.x * 256 + threadIdx.x; q < nnz; q += gridDim.x * 256) {
        u32 lo = 0, hi = nrows - 1;
        while (lo < hi) {
            u32 mid = (lo + hi + 1) >> 1;
            if (c.rowptr[mid] <= q) lo = mid; else hi = mid - 1;
        }
        sum += cs_row_hash(rowmap ? rowmap[lo] : lo) * cs_dest_hash(c.colidx[q]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane_id() == 0 && sum) atomicAdd(acc, (unsigned long long)sum);
}

static fgpu_info empty_dense(fgpu_ctx* ctx, fgpu_mat** out, u64 nrows, u64 ncols) {
    fgpu_mat* o = nullptr;
    FGPU_TRY(mat_alloc(ctx, &o, nrows, ncols, 0, false, 0, false));
    hipError_t e = hipMemsetAsync(o->rowptr, 0, (nrows + 1) * sizeof(u32), ctx->stream());
    if (e != hipSuccess) { mat_release(o); set_error("memset failed: %s", hipGetErrorString(e)); return FGPU_DEVICE; }
    *out = o;
    return FGPU_OK;
}

// C = F x B (structural).  *flops (nullable) += sum_{(i,s) in F} deg_B(s).
fgpu_info mxm_device(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* F, const fgpu_mat* B, u64* flops) {
    FGPU_REQUIRE(F->ncols == B->nrows, FGPU_DIM_MISMATCH, "mxm: F is %llu x %llu but B is %llu x %llu",
                 (unsigned long long)F->nrows, (unsigned long long)F->ncols, (unsigned long long)B->nrows,
                 (unsigned long long)B->ncols);
    const u64 k = F->nrows;
    const u32 nnzf = (u32)F->nnz;
    if (nnzf == 0 || B->nnz == 0) return empty_dense(ctx, out, k, B->ncols);
    DevBuf<u32> frp, deg, cnt, rowptr, maxlen;
    DevBuf<u64> eoff, roff, tot;
    FGPU_TRY(dense_rowptr(ctx, F, frp));
    FGPU_TRY(deg.alloc(ctx, (size_t)nnzf + 1));
    FGPU_TRY(eoff.alloc(ctx, (size_t)nnzf + 1));
    FGPU_TRY(tot.alloc(ctx, 1));
    hipLaunchKernelGGL(entry_deg_kernel, dim3(cdiv((u64)nnzf + 1, 256)), dim3(256), 0, ctx->stream(), view_of(F),
                       view_of(B), nnzf, deg.p);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(scan_u32_to_u64(ctx, deg.p, eoff.p, (u64)nnzf + 1, tot.p));
    FGPU_TRY(roff.alloc(ctx, k + 1));
    FGPU_TRY(maxlen.alloc(ctx, 1));
    FGPU_HIP(hipMemsetAsync(maxlen.p, 0, sizeof(u32), ctx->stream()));
    hipLaunchKernelGGL(gather_u64_kernel, dim3(cdiv(k + 1, 256)), dim3(256), 0, ctx->stream(), (const u64*)eoff.p,
                       (const u32*)frp.p, (u32)k, roff.p, maxlen.p);
    FGPU_HIP(hipGetLastError());
    u64 T = 0;
    FGPU_TRY(read_u64(ctx, tot.p, &T));
    u32 ml = 0;
    FGPU_TRY(read_u32(ctx, maxlen.p, &ml));
    if (flops) *flops += T;
    if (T == 0) return empty_dense(ctx, out, k, B->ncols);
    FGPU_REQUIRE(T < (1ull << 33), FGPU_OOM,
                 "mxm: %llu gathered entries exceed the per-call workspace; batch the source rows",
                 (unsigned long long)T);
    DevBuf<u32> tmp;
    FGPU_TRY(tmp.alloc(ctx, T));
    {
        ProfScope ps(ctx, "gather_rows_kernel", 12 * (u64)nnzf + 8 * T);   // F entry + B row-pointer pair, B row read + written
        u32 grid = cdiv(nnzf, 4);
        if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
        hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(F), view_of(B), nnzf,
                           (const u64*)eoff.p, tmp.p);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(cnt.alloc(ctx, k + 1));
    if (ml <= 1) {
        hipLaunchKernelGGL(seg_len_kernel, dim3(cdiv(k + 1, 256)), dim3(256), 0, ctx->stream(), (const u64*)roff.p,
                           (u32)k, cnt.p);
        FGPU_HIP(hipGetLastError());
    } else {
        FGPU_HIP(hipMemsetAsync(cnt.p, 0, (k + 1) * sizeof(u32), ctx->stream()));
        ProfScope ps(ctx, "segsort_unique (product rows)", 8 * T);
        FGPU_TRY(segsort_unique(ctx, tmp.p, roff.p, (u32)k, (u32)B->ncols, cnt.p, nullptr));
    }
    FGPU_TRY(rowptr.alloc(ctx, k + 1));
    FGPU_TRY(scan_u32(ctx, cnt.p, rowptr.p, k + 1, nullptr));
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, rowptr.p + k, &nnz));
    fgpu_mat* o = nullptr;
    FGPU_TRY(mat_alloc(ctx, &o, k, B->ncols, nnz, false, 0, false));
    FGPU_HIP(hipMemcpyAsync(o->rowptr, rowptr.p, (k + 1) * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream()));
    fgpu_info i;
    {
        ProfScope ps(ctx, "compact_segments_kernel", 8 * (u64)nnz);
        i = compact_segments(ctx, tmp.p, roff.p, o->rowptr, (u32)k, o->colidx);
    }
    if (i != FGPU_OK) { mat_release(o); return i; }
    // hub lists are only needed by BFS; products skip mat_finalize (no extra sync per hop)
    *out = o;
    return FGPU_OK;
}

fgpu_info delta_lmxm_device(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* F, const fgpu_mat* m, const fgpu_mat* dp,
                            const fgpu_mat* dm, u64* flops) {
    const bool has_dp = dp && dp->nnz, has_dm = dm && dm->nnz;
    if (!has_dp && !has_dm) return mxm_device(ctx, out, F, m, flops);  // hot path, matrix.rs:1333-1337
    fgpu_mat *mask = nullptr, *acc = nullptr, *c = nullptr;
    fgpu_info i = FGPU_OK;
    if (has_dm) {
        i = mxm_device(ctx, &mask, F, dm, nullptr);
        if (i == FGPU_OK && mask->nnz == 0) { mat_release(mask); mask = nullptr; }
    }
    if (i == FGPU_OK && has_dp) {
        i = mxm_device(ctx, &acc, F, dp, flops);
        if (i == FGPU_OK && acc->nnz == 0) { mat_release(acc); acc = nullptr; }
    }
    if (i == FGPU_OK) i = mxm_device(ctx, &c, F, m, flops);
    if (i == FGPU_OK && (mask || acc)) {
        fgpu_mat* merged = nullptr;
        // (F.m with MASK removed) U ACCUM — the accumulated dp product is not masked (matrix.rs:1382-1400)
        i = mat_merge_entries(ctx, &merged, c, acc, mask, false, c->nrows, c->ncols, true);
        if (i == FGPU_OK) { mat_release(c); c = merged; }
    }
    mat_release(mask);
    mat_release(acc);
    if (i != FGPU_OK) { mat_release(c); return i; }
    *out = c;
    return FGPU_OK;
}

static fgpu_info filter_by_bitmap(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* c, const u64* bitmap_dev) {
    const u64 nrows = c->nrows;
    DevBuf<u32> tmp, cnt, rowptr;
    FGPU_TRY(tmp.alloc(ctx, c->nnz));
    FGPU_TRY(cnt.alloc(ctx, nrows + 1));
    FGPU_HIP(hipMemsetAsync(cnt.p, 0, (nrows + 1) * sizeof(u32), ctx->stream()));
    u32 grid = cdiv(nrows ? nrows : 1, 4);
    if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
    if (nrows && c->nnz) {
        hipLaunchKernelGGL(bitmap_filter_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(c), bitmap_dev,
                           (u32)nrows, tmp.p, cnt.p);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(rowptr.alloc(ctx, nrows + 1));
    FGPU_TRY(scan_u32(ctx, cnt.p, rowptr.p, nrows + 1, nullptr));
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, rowptr.p + nrows, &nnz));
    fgpu_mat* o = nullptr;
    FGPU_TRY(mat_alloc(ctx, &o, nrows, c->ncols, nnz, false, 0, false));
    FGPU_HIP(hipMemcpyAsync(o->rowptr, rowptr.p, (nrows + 1) * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream()));
    if (nnz) {
        hipLaunchKernelGGL(compact_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream(), (const u32*)tmp.p,
                           (const u32*)c->rowptr, (const u32*)o->rowptr, (u32)nrows, o->colidx);
        FGPU_HIP(hipGetLastError());
    }
    *out = o;
    return FGPU_OK;
}

// F: one row per source, at most one entry per row (cond_traverse.rs:600-601); UINT64_MAX = row left empty
static fgpu_info upload_sources(fgpu_ctx* ctx, fgpu_mat** out, const uint64_t* src_ids, uint64_t nsrc, u64 ncols0) {
    std::vector<u32> rp(nsrc + 1), ci;
    ci.reserve(nsrc);
    rp[0] = 0;
    for (u64 i = 0; i < nsrc; ++i) {
        if (src_ids[i] != UINT64_MAX) {
            FGPU_REQUIRE(src_ids[i] < ncols0, FGPU_OUT_OF_BOUNDS, "expand: source %llu = %llu >= %llu",
                         (unsigned long long)i, (unsigned long long)src_ids[i], (unsigned long long)ncols0);
            ci.push_back((u32)src_ids[i]);
        }
        rp[i + 1] = (u32)ci.size();
    }
    fgpu_mat* f = nullptr;
    FGPU_TRY(mat_alloc(ctx, &f, nsrc, ncols0, ci.size(), false, 0, false));
    fgpu_info u = ctx->h2d(f->rowptr, rp.data(), rp.size() * sizeof(u32));
    if (u == FGPU_OK && !ci.empty()) u = ctx->h2d(f->colidx, ci.data(), ci.size() * sizeof(u32));
    if (u != FGPU_OK) { mat_release(f); return u; }
    *out = f;
    return FGPU_OK;
}

static fgpu_info check_hops(const fgpu_mat* const* m, const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops,
                            uint64_t nsrc) {
    FGPU_REQUIRE(nhops >= 1 && m, FGPU_INVALID, "expand: need at least one hop");
    FGPU_REQUIRE(nsrc < 0xFFFFFFFFull, FGPU_INVALID, "expand: too many source rows");
    for (int h = 0; h < nhops; ++h) {
        FGPU_REQUIRE(m[h], FGPU_NULL_POINTER, "expand: hop %d base matrix is NULL", h);
        FGPU_REQUIRE(!dp || !dp[h] || (dp[h]->nrows == m[h]->nrows && dp[h]->ncols == m[h]->ncols),
                     FGPU_DIM_MISMATCH, "expand: hop %d dp dims differ from m", h);
        FGPU_REQUIRE(!dm || !dm[h] || (dm[h]->nrows == m[h]->nrows && dm[h]->ncols == m[h]->ncols),
                     FGPU_DIM_MISMATCH, "expand: hop %d dm dims differ from m", h);
        FGPU_REQUIRE(h == 0 || m[h]->nrows == m[h - 1]->ncols, FGPU_DIM_MISMATCH,
                     "expand: hop %d rows do not match hop %d columns", h, h - 1);
    }
    return FGPU_OK;
}

// ---- first hop from one-entry rows over a clean layer --------------------------------------------------------------------
// F0 has at most one entry per row (cond_traverse.rs:600-601: F[i, src_i] = 1), so F0 x m is "row i = row src_i of m": the
// rows are copied as they are — sorted, unique, nothing to sort or collapse.  The general product (degrees per entry, two
// scans, row offsets, gather, segment lengths, compaction: ~14 launches and three read-backs) is replaced by ONE workgroup that
// scans the <= 4096 source degrees, one read-back of the size, and one copy kernel that also sums the NEXT hop's traversed
// edges (sum over the copied entries of deg_next(col): what mxm_flops would compute for the form decision of hop 2).
constexpr u32 FH_MAX_ROWS = 4096;
__global__ __launch_bounds__(1024) void first_hop_scan_kernel(CsrView f, CsrView m, u32 k, u32* __restrict__ rp, u32* __restrict__ total,
                                                              unsigned long long* __restrict__ tnext_slots) {
    __shared__ u32 s_wave[16];
    __shared__ u32 s_base;
    tnext_slots[threadIdx.x] = 0ull;              // the 64 x 16 words first_hop_copy_kernel sums into (one launch less than a memset)
    u32 d[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                 // thread t owns rows 4 t .. 4 t + 3 (consecutive: the scan stays in order)
        const u32 i = threadIdx.x * 4 + j;
        u32 dg = 0;
        if (i < k && f.rowptr[i + 1] > f.rowptr[i]) {
            u32 b, e;
            row_range(m, f.colidx[f.rowptr[i]], b, e);
            dg = e - b;
        }
        d[j] = dg;
        sum += dg;
    }
    u32 inc = sum;                                 // inclusive scan of the per-thread sums inside the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 v = (u32)__shfl_up((int)inc, o, 64); if ((int)lane_id() >= o) inc += v; }
    if (lane_id() == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int w = 0; w < 16; ++w) { const u32 v = s_wave[w]; s_wave[w] = run; run += v; }
        s_base = run;
    }
    __syncthreads();
    u32 off = s_wave[threadIdx.x >> 6] + inc - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 i = threadIdx.x * 4 + j;
        if (i <= k) rp[i] = off;
        off += d[j];
    }
    if (threadIdx.x == 0) total[0] = s_base;
}
// a lane per RESULT entry (the row of an entry by a search of the <= 4097 row pointers in LDS): a hub source — one :P vertex in
// a thousand has 10^5 out-edges — was copied by ONE wavefront in the first version, 27 us of a 1.45 ms batch.  Sums deg_next
// over the copied column ids (next_rowptr nullable); workgroup 0 also writes the result's row pointers.
__global__ __launch_bounds__(256) void first_hop_copy_kernel(CsrView f, CsrView m, u32 k, const u32* __restrict__ rp, u32 nnz,
                                                             u32* __restrict__ out_rowptr, u32* __restrict__ col,
                                                             const u32* __restrict__ next_rowptr, u32 next_rows,
                                                             unsigned long long* __restrict__ tnext) {
    __shared__ u32 s_rp[FH_MAX_ROWS + 1];
    for (u32 i = threadIdx.x; i <= k; i += 256) {
        const u32 v = rp[i];
        s_rp[i] = v;
        if (blockIdx.x == 0) out_rowptr[i] = v;
    }
    __syncthreads();
    u64 t = 0;
    for (u32 q = blockIdx.x * 256 + threadIdx.x; q < nnz; q += gridDim.x * 256) {
        u32 lo = 0, hi = k - 1;                              // largest row with s_rp[row] <= q
        while (lo < hi) {
            const u32 mid = (lo + hi + 1) >> 1;
            if (s_rp[mid] <= q) lo = mid; else hi = mid - 1;
        }
        u32 b, e;
        row_range(m, f.colidx[f.rowptr[lo]], b, e);
        const u32 c = m.colidx[b + (q - s_rp[lo])];
        col[q] = c;
        if (next_rowptr && c < next_rows) t += next_rowptr[c + 1] - next_rowptr[c];
    }
    if (tnext) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
        if (lane_id() == 0 && t) atomicAdd(&tnext[(wave & 63u) * 16u], (unsigned long long)t);   // 64 slots, a 128-byte line apart
    }
}
// the 64 slots summed; lane 0 publishes the sum into the lane's mapped line itself when there is one (ctx.hip pub_begin)
__global__ void first_hop_fold_kernel(unsigned long long* __restrict__ tnext, u32* __restrict__ pub, u32 seq) {
    u64 v = tnext[(size_t)threadIdx.x * 16];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (threadIdx.x == 0) {
        if (pub) {
            __hip_atomic_store(pub + 0, (u32)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub + 1, (u32)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            tnext[1] = v;
        }
    }
}

// *out = F0 x m for a clean hop (no dp / dm); *T0 = its traversed edges; *Tnext (when `next` is a plain CSR) = the traversed
// edges of the following hop over `next`.  Returns FGPU_NO_VALUE when the shortcut does not apply (the caller takes the general path).
static fgpu_info first_hop_rows(fgpu_ctx* ctx, const fgpu_mat* f, const fgpu_mat* m, const fgpu_mat* next, fgpu_mat** out,
                                u64* T0, u64* Tnext, bool* have_next) {
    *have_next = false;
    const u32 k = (u32)f->nrows;
    // (k == FH_MAX_ROWS is out: the scan kernel's 1024 threads x 4 rows write rp[0 .. 4095], never rp[4096])
    if (k == 0 || k >= FH_MAX_ROWS || f->is_hyper() || f->nnz > f->nrows || m->nnz == 0 || f->nnz == 0) return FGPU_NO_VALUE;
    DevBuf<u32> rp, tot;
    DevBuf<u64> tn;
    FGPU_TRY(rp.alloc(ctx, (size_t)k + 1));
    FGPU_TRY(tot.alloc(ctx, 1));
    FGPU_TRY(tn.alloc(ctx, 64 * 16));
    hipLaunchKernelGGL(first_hop_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream(), view_of(f), view_of(m), k, rp.p, tot.p,
                       (unsigned long long*)tn.p);
    FGPU_HIP(hipGetLastError());
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, tot.p, &nnz));
    *T0 = nnz;
    fgpu_mat* c = nullptr;
    FGPU_TRY(mat_alloc(ctx, &c, k, m->ncols, nnz, false, 0, false));
    const bool sum_next = next && !next->is_hyper() && next->nnz && nnz;
    fgpu_info i = FGPU_OK;
    {
        u32 grid = cdiv(nnz ? nnz : 1, 256 * 4);
        if (grid > (u32)ctx->cus * 4) grid = ctx->cus * 4;
        hipLaunchKernelGGL(first_hop_copy_kernel, dim3(grid ? grid : 1), dim3(256), 0, ctx->stream(), view_of(f), view_of(m), k,
                           (const u32*)rp.p, nnz, c->rowptr, c->colidx, sum_next ? (const u32*)next->rowptr : (const u32*)nullptr,
                           sum_next ? (u32)next->nrows : 0u, sum_next ? (unsigned long long*)tn.p : (unsigned long long*)nullptr);
        if (hipGetLastError() != hipSuccess) i = FGPU_DEVICE;
    }
    if (i == FGPU_OK && sum_next) {
        u32* pub = nullptr;
        u32 seq = 0;
        const bool mapped = pub_begin(ctx, &pub, &seq);
        hipLaunchKernelGGL(first_hop_fold_kernel, dim3(1), dim3(64), 0, ctx->stream(), (unsigned long long*)tn.p, mapped ? pub : (u32*)nullptr, seq);
        if (mapped) {
            u32 w[2] = {0, 0};
            i = hipGetLastError() == hipSuccess ? pub_wait(ctx, seq, 2, w) : FGPU_DEVICE;
            *Tnext = (u64)w[0] | ((u64)w[1] << 32);
        } else {
            i = read_u64(ctx, tn.p + 1, Tnext);
        }
        *have_next = i == FGPU_OK;
    }
    if (i != FGPU_OK) { mat_release(c); if (i == FGPU_DEVICE) set_error("first hop: device call failed"); return i; }
    *out = c;
    return FGPU_OK;
}

// ---- the first hop from one-entry rows over DIRTY layers -----------------------------------------------------------------
// Row i of F holds one source u: (F·m)<not (F·dm)> U (F·dp) is then (m[u] \ dm[u]) U dp[u] row by row (the row-level mask of
// Matrix::delta_lmxm, matrix.rs:1343-1361, is the row's own tombstones).  The general product does this in ~75 launches of a
// few microseconds (three gathers, three sorts, the merge machinery); here: candidate offsets (one single-workgroup scan), keep
// flags a lane per candidate (binary searches in the sorted rows), two scans of the flags, row lengths, and a fill that places
// every kept candidate by its rank among the kept entries of both lists — the result row stays ascending and unique whatever
// the layers hold (no Delta invariant is assumed: a dp entry that m already has is dropped, a tombstone outside m does nothing).
__global__ __launch_bounds__(1024) void fhd_scan_kernel(CsrView f, CsrView m, CsrView dp, u32 k, u32* __restrict__ cm, u32* __restrict__ cp,
                                                        u32* __restrict__ totals, unsigned long long* __restrict__ tnext_slots) {
    __shared__ u32 s_wave[2][16];
    tnext_slots[threadIdx.x] = 0ull;
    u32 a[4], b[4], sa = 0, sb = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 i = threadIdx.x * 4 + j;
        u32 da = 0, db = 0;
        if (i < k && f.rowptr[i + 1] > f.rowptr[i]) {
            const u32 u = f.colidx[f.rowptr[i]];
            u32 x, y;
            row_range(m, u, x, y); da = y - x;
            row_range(dp, u, x, y); db = y - x;
        }
        a[j] = da; b[j] = db; sa += da; sb += db;
    }
    u32 ia = sa, ib = sb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 va = (u32)__shfl_up((int)ia, o, 64), vb = (u32)__shfl_up((int)ib, o, 64);
        if ((int)lane_id() >= o) { ia += va; ib += vb; }
    }
    if (lane_id() == 63) { s_wave[0][threadIdx.x >> 6] = ia; s_wave[1][threadIdx.x >> 6] = ib; }
    __syncthreads();
    if (threadIdx.x < 2) {
        u32 run = 0;
        for (int w = 0; w < 16; ++w) { const u32 v = s_wave[threadIdx.x][w]; s_wave[threadIdx.x][w] = run; run += v; }
        totals[threadIdx.x] = run;
    }
    __syncthreads();
    u32 oa = s_wave[0][threadIdx.x >> 6] + ia - sa, ob = s_wave[1][threadIdx.x >> 6] + ib - sb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 i = threadIdx.x * 4 + j;
        if (i <= k) { cm[i] = oa; cp[i] = ob; }
        oa += a[j]; ob += b[j];
    }
}
__device__ __forceinline__ u32 fhd_lower_bound(const u32* __restrict__ col, u32 b, u32 e, u32 x) {
    while (b < e) {
        const u32 mid = (b + e) >> 1;
        if (col[mid] < x) b = mid + 1; else e = mid;
    }
    return b;
}
// the row of candidate q: largest i with ptr[i] <= q (k + 1 offsets in LDS)
__device__ __forceinline__ u32 fhd_row_of(const u32* s_ptr, u32 k, u32 q) {
    u32 lo = 0, hi = k - 1;
    while (lo < hi) {
        const u32 mid = (lo + hi + 1) >> 1;
        if (s_ptr[mid] <= q) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// FILL = false: keep flags.  FILL = true: every kept candidate to its place (rank among the kept entries of both lists).
template <bool FILL>
__global__ __launch_bounds__(256) void fhd_cand_kernel(CsrView f, CsrView m, CsrView dp, CsrView dm, u32 k, const u32* __restrict__ cm,
                                                       const u32* __restrict__ cp, u32 ncm, u32 ncp, u32* __restrict__ keepm,
                                                       u32* __restrict__ keepp, const u32* __restrict__ pm, const u32* __restrict__ pp,
                                                       const u32* __restrict__ rowptr, u32* __restrict__ col,
                                                       const u32* __restrict__ next_rowptr, u32 next_rows,
                                                       unsigned long long* __restrict__ tnext) {
    __shared__ u32 s_cm[FH_MAX_ROWS + 1], s_cp[FH_MAX_ROWS + 1];
    for (u32 i = threadIdx.x; i <= k; i += 256) { s_cm[i] = cm[i]; s_cp[i] = cp[i]; }
    __syncthreads();
    u64 t = 0;
    for (u32 q = blockIdx.x * 256 + threadIdx.x; q < ncm + ncp; q += gridDim.x * 256) {
        const bool from_m = q < ncm;
        const u32 c = from_m ? q : q - ncm;
        const u32 i = fhd_row_of(from_m ? s_cm : s_cp, k, c);
        const u32 u = f.colidx[f.rowptr[i]];
        u32 mb, me, pb, pe, db, de;
        row_range(m, u, mb, me);
        row_range(dp, u, pb, pe);
        row_range(dm, u, db, de);
        const u32 j = c - (from_m ? s_cm[i] : s_cp[i]);
        const u32 x = from_m ? m.colidx[mb + j] : dp.colidx[pb + j];
        const u32 dl = fhd_lower_bound(dm.colidx, db, de, x);
        const bool tomb = dl < de && dm.colidx[dl] == x;
        bool keep;
        u32 other = 0;                                       // index of x's lower bound in the OTHER list
        if (from_m) {
            keep = !tomb;
            if (FILL) other = fhd_lower_bound(dp.colidx, pb, pe, x) - pb;
        } else {
            const u32 ml = fhd_lower_bound(m.colidx, mb, me, x);
            keep = !(ml < me && m.colidx[ml] == x && !tomb);  // (m keeps x itself: the dp copy is a duplicate)
            other = ml - mb;
        }
        if (!FILL) {
            (from_m ? keepm : keepp)[c] = keep ? 1u : 0u;
        } else if (keep) {
            const u32 pos = from_m ? (pm[c] - pm[s_cm[i]]) + (pp[s_cp[i] + other] - pp[s_cp[i]])
                                   : (pp[c] - pp[s_cp[i]]) + (pm[s_cm[i] + other] - pm[s_cm[i]]);
            col[rowptr[i] + pos] = x;
            if (next_rowptr && x < next_rows) t += next_rowptr[x + 1] - next_rowptr[x];
        }
    }
    if (FILL && tnext) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
        if (lane_id() == 0 && t) atomicAdd(&tnext[(wave & 63u) * 16u], (unsigned long long)t);
    }
}
// row lengths from the two flag scans, their exclusive scan, the total published (k <= 4095)
__global__ __launch_bounds__(1024) void fhd_rowptr_kernel(const u32* __restrict__ cm, const u32* __restrict__ cp, const u32* __restrict__ pm,
                                                          const u32* __restrict__ pp, u32 k, u32* __restrict__ rowptr, u32* __restrict__ pub,
                                                          u32 seq, u32* __restrict__ total) {
    __shared__ u32 s_wave[16];
    u32 d[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 i = threadIdx.x * 4 + j;
        d[j] = i < k ? (pm[cm[i + 1]] - pm[cm[i]]) + (pp[cp[i + 1]] - pp[cp[i]]) : 0u;
        sum += d[j];
    }
    u32 inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 v = (u32)__shfl_up((int)inc, o, 64); if ((int)lane_id() >= o) inc += v; }
    if (lane_id() == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int w = 0; w < 16; ++w) { const u32 v = s_wave[w]; s_wave[w] = run; run += v; }
    }
    __syncthreads();
    u32 off = s_wave[threadIdx.x >> 6] + inc - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 i = threadIdx.x * 4 + j;
        if (i <= k) rowptr[i] = off;
        if (i == k) {
            total[0] = off;
            if (pub) {
                __hip_atomic_store(pub + 0, off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(pub + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        off += d[j];
    }
}

static fgpu_info first_hop_rows_dirty(fgpu_ctx* ctx, const fgpu_mat* f, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm,
                                      const fgpu_mat* next, fgpu_mat** out, u64* T0, u64* Tnext, bool* have_next) {
    *have_next = false;
    const u32 k = (u32)f->nrows;
    if (k == 0 || k >= FH_MAX_ROWS || f->is_hyper() || f->nnz > f->nrows || f->nnz == 0 || m->is_hyper()) return FGPU_NO_VALUE;
    hipStream_t st = ctx->stream();
    DevBuf<u32> cm, cp, tot, empty;
    DevBuf<u64> tn;
    FGPU_TRY(cm.alloc(ctx, (size_t)k + 1));
    FGPU_TRY(cp.alloc(ctx, (size_t)k + 1));
    FGPU_TRY(tot.alloc(ctx, 4));
    FGPU_TRY(tn.alloc(ctx, 64 * 16));
    FGPU_TRY(empty.alloc(ctx, 4));
    FGPU_HIP(hipMemsetAsync(empty.p, 0, 4 * sizeof(u32), st));
    auto view_or_empty = [&](const fgpu_mat* a) {
        if (a && a->nnz) return view_of(a);
        CsrView v;                                           // hypersparse with no stored row: every row_range is empty
        v.rowptr = empty.p; v.colidx = empty.p; v.hrows = empty.p; v.nvec = 0; v.nrows = (u32)m->nrows;
        return v;
    };
    const CsrView vm = view_of(m), vdp = view_or_empty(dp), vdm = view_or_empty(dm), vf = view_of(f);
    hipLaunchKernelGGL(fhd_scan_kernel, dim3(1), dim3(1024), 0, st, vf, vm, vdp, k, cm.p, cp.p, tot.p, (unsigned long long*)tn.p);
    FGPU_HIP(hipGetLastError());
    u32 w2[2] = {0, 0};
    FGPU_TRY(read_words(ctx, tot.p, 2, w2));
    const u32 ncm = w2[0], ncp = w2[1];
    *T0 = (u64)ncm + ncp;                                    // traversed edges of the hop: deg over m and over dp
    DevBuf<u32> keepm, keepp, pm, pp;
    FGPU_TRY(keepm.alloc(ctx, (size_t)ncm + 1)); FGPU_TRY(keepp.alloc(ctx, (size_t)ncp + 1));
    FGPU_TRY(pm.alloc(ctx, (size_t)ncm + 1)); FGPU_TRY(pp.alloc(ctx, (size_t)ncp + 1));
    u32 grid = cdiv((u64)ncm + ncp ? (u64)ncm + ncp : 1, 256 * 2);
    if (grid > (u32)ctx->cus * 4) grid = ctx->cus * 4;
    FGPU_HIP(hipMemsetAsync(keepm.p + ncm, 0, sizeof(u32), st));
    FGPU_HIP(hipMemsetAsync(keepp.p + ncp, 0, sizeof(u32), st));
    hipLaunchKernelGGL(fhd_cand_kernel<false>, dim3(grid ? grid : 1), dim3(256), 0, st, vf, vm, vdp, vdm, k, (const u32*)cm.p, (const u32*)cp.p,
                       ncm, ncp, keepm.p, keepp.p, (const u32*)nullptr, (const u32*)nullptr, (const u32*)nullptr, (u32*)nullptr,
                       (const u32*)nullptr, 0u, (unsigned long long*)nullptr);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(scan_u32(ctx, keepm.p, pm.p, (u64)ncm + 1, nullptr));
    FGPU_TRY(scan_u32(ctx, keepp.p, pp.p, (u64)ncp + 1, nullptr));
    DevBuf<u32> rp;
    FGPU_TRY(rp.alloc(ctx, (size_t)k + 1));
    u32 nnz = 0;
    {
        u32* pub = nullptr;
        u32 seq = 0;
        const bool mapped = pub_begin(ctx, &pub, &seq);
        hipLaunchKernelGGL(fhd_rowptr_kernel, dim3(1), dim3(1024), 0, st, (const u32*)cm.p, (const u32*)cp.p, (const u32*)pm.p,
                           (const u32*)pp.p, k, rp.p, mapped ? pub : (u32*)nullptr, seq, tot.p + 2);
        FGPU_HIP(hipGetLastError());
        if (mapped) { u32 w[1] = {0}; FGPU_TRY(pub_wait(ctx, seq, 1, w)); nnz = w[0]; }
        else FGPU_TRY(read_u32(ctx, tot.p + 2, &nnz));
    }
    fgpu_mat* c = nullptr;
    FGPU_TRY(mat_alloc(ctx, &c, k, m->ncols, nnz, false, 0, false));
    const bool sum_next = next && !next->is_hyper() && next->nnz && nnz;
    fgpu_info i = FGPU_OK;
    if (hipMemcpyAsync(c->rowptr, rp.p, ((size_t)k + 1) * sizeof(u32), hipMemcpyDeviceToDevice, st) != hipSuccess) i = FGPU_DEVICE;
    if (i == FGPU_OK && nnz) {
        hipLaunchKernelGGL(fhd_cand_kernel<true>, dim3(grid ? grid : 1), dim3(256), 0, st, vf, vm, vdp, vdm, k, (const u32*)cm.p,
                           (const u32*)cp.p, ncm, ncp, (u32*)nullptr, (u32*)nullptr, (const u32*)pm.p, (const u32*)pp.p, (const u32*)rp.p,
                           c->colidx, sum_next ? (const u32*)next->rowptr : (const u32*)nullptr, sum_next ? (u32)next->nrows : 0u,
                           sum_next ? (unsigned long long*)tn.p : (unsigned long long*)nullptr);
        if (hipGetLastError() != hipSuccess) i = FGPU_DEVICE;
    }
    if (i == FGPU_OK && sum_next) {
        u32* pub = nullptr;
        u32 seq = 0;
        const bool mapped = pub_begin(ctx, &pub, &seq);
        hipLaunchKernelGGL(first_hop_fold_kernel, dim3(1), dim3(64), 0, st, (unsigned long long*)tn.p, mapped ? pub : (u32*)nullptr, seq);
        if (mapped) {
            u32 w[2] = {0, 0};
            i = hipGetLastError() == hipSuccess ? pub_wait(ctx, seq, 2, w) : FGPU_DEVICE;
            *Tnext = (u64)w[0] | ((u64)w[1] << 32);
        } else {
            i = read_u64(ctx, tn.p + 1, Tnext);
        }
        *have_next = i == FGPU_OK;
    }
    if (i != FGPU_OK) { mat_release(c); if (i == FGPU_DEVICE) set_error("dirty first hop: device call failed"); return i; }
    *out = c;
    return FGPU_OK;
}

// ---- dropping the empty source rows of a count-only chain before it goes to bits ------------------------------------------
// A source without out-edges (half of the vertices of an R-MAT graph, half of a `:P` batch) leaves an empty row after the
// first hop and can never contribute again, but keeps its bit in every 128-byte row of the bit state.  When the live rows
// fit HALF the words (1024 sources, 499 live: 16 -> 8 words per vertex) the frontier is renumbered to the live rows only:
// the dense last hop gathers 64-byte rows (two vertices per line, a state that fits the Infinity Cache at RMAT-22: -13 %),
// the middle hop writes half the bytes (-27 %).  The way back: the checksum's row hashes go through `map`, the emitted
// row pointers through `rank` (bp_to_csr).
__global__ void cr_flag_kernel(const u32* __restrict__ rowptr, u32 k, u32* __restrict__ flag) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= k) flag[i] = (i < k && rowptr[i + 1] > rowptr[i]) ? 1u : 0u;
}
// k <= 4096 rows (a child batch holds 1024): flags, their exclusive scan and the read-back of the live count in ONE
// single-workgroup launch (was: flag kernel, scan, a one-thread publish kernel)
__global__ __launch_bounds__(1024) void cr_rank_kernel(const u32* __restrict__ rowptr, u32 k, u32* __restrict__ rank, u32* __restrict__ pub,
                                                       u32 seq) {
    __shared__ u32 s_wave[16];
    u32 d[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 i = threadIdx.x * 4 + j;
        d[j] = (i < k && rowptr[i + 1] > rowptr[i]) ? 1u : 0u;
        sum += d[j];
    }
    u32 inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 v = (u32)__shfl_up((int)inc, o, 64); if ((int)lane_id() >= o) inc += v; }
    if (lane_id() == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int w = 0; w < 16; ++w) { const u32 v = s_wave[w]; s_wave[w] = run; run += v; }
    }
    __syncthreads();
    u32 off = s_wave[threadIdx.x >> 6] + inc - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 i = threadIdx.x * 4 + j;
        if (i <= k) rank[i] = off;
        if (i == k && pub) {
            __hip_atomic_store(pub + 0, off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        off += d[j];
    }
}
__global__ void cr_build_kernel(const u32* __restrict__ rowptr, const u32* __restrict__ colidx, u32 k, u32 nnz, const u32* __restrict__ rank,
                                u32* __restrict__ new_rowptr, u32* __restrict__ new_colidx, u32* __restrict__ map, u32 nlive) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k && rowptr[i + 1] > rowptr[i]) { new_rowptr[rank[i]] = rowptr[i]; map[rank[i]] = i; }
    if (i == k) new_rowptr[nlive] = rowptr[k];
    for (u32 q = i; q < nnz; q += gridDim.x * blockDim.x) new_colidx[q] = colidx[q];   // (the entries stay where they are)
}
static u32 bits_stride(u32 rows) {                         // = bp_layout's row stride for `rows` source rows
    u32 w = (rows + 63) / 64;
    if (w == 0) w = 1;
    if (w > 64) return ((w + 63) / 64) * 64;
    u32 p = 1;
    while (p < w) p <<= 1;
    return p;
}
// *out = f without its empty rows (nullptr: nothing to gain, f stays), map[new row] = old row
static fgpu_info compact_source_rows(fgpu_ctx* ctx, const fgpu_mat* f, fgpu_mat** out, DevBuf<u32>& map, DevBuf<u32>& rank_out) {
    *out = nullptr;
    const u32 k = (u32)f->nrows;
    if (f->is_hyper() || k < 128 || f->nnz == 0) return FGPU_OK;
    DevBuf<u32> flag, rank;
    FGPU_TRY(rank.alloc(ctx, (size_t)k + 1));
    u32 nlive = 0;
    u32* pub = nullptr;
    u32 seq = 0;
    if (k <= 4095 && pub_begin(ctx, &pub, &seq)) {
        hipLaunchKernelGGL(cr_rank_kernel, dim3(1), dim3(1024), 0, ctx->stream(), (const u32*)f->rowptr, k, rank.p, pub, seq);
        FGPU_HIP(hipGetLastError());
        u32 w[1] = {0};
        FGPU_TRY(pub_wait(ctx, seq, 1, w));
        nlive = w[0];
    } else {
        FGPU_TRY(flag.alloc(ctx, (size_t)k + 1));
        hipLaunchKernelGGL(cr_flag_kernel, dim3(cdiv((u64)k + 1, 256)), dim3(256), 0, ctx->stream(), (const u32*)f->rowptr, k, flag.p);
        FGPU_HIP(hipGetLastError());
        FGPU_TRY(scan_u32(ctx, flag.p, rank.p, (u64)k + 1, nullptr));
        FGPU_TRY(read_u32(ctx, rank.p + k, &nlive));
    }
    if (nlive == 0 || bits_stride(nlive) >= bits_stride(k)) return FGPU_OK;
    fgpu_mat* c = nullptr;
    FGPU_TRY(mat_alloc(ctx, &c, nlive, f->ncols, f->nnz, false, 0, false));
    fgpu_info i = map.alloc(ctx, nlive);
    if (i == FGPU_OK) {
        u32 grid = cdiv((u64)k + 1, 256);
        const u32 want = cdiv(f->nnz, 256 * 8);
        if (grid < want) grid = want < (u32)ctx->cus * 4 ? want : (u32)ctx->cus * 4;
        hipLaunchKernelGGL(cr_build_kernel, dim3(grid), dim3(256), 0, ctx->stream(), (const u32*)f->rowptr, (const u32*)f->colidx, k,
                           (u32)f->nnz, (const u32*)rank.p, c->rowptr, c->colidx, map.p, nlive);
        if (hipGetLastError() != hipSuccess) {
            set_error("compact_source_rows: device call failed");
            i = FGPU_DEVICE;
        }
    }
    if (i != FGPU_OK) { mat_release(c); return i; }
    rank_out = std::move(rank);        // rank[i] = live rows before source row i, rank[k] = nlive: the way back (bp_to_csr)
    *out = c;
    return FGPU_OK;
}

// One pass of a whole-frontier call (expand_count_scan below): F is already on the device — one entry per row, no empty
// rows — and row i of it is row rowmap[i] of the CALL (what the checksum's row hashes need).
struct ChainSources {
    fgpu_mat* f = nullptr;           // taken over by the chain
    DevBuf<u32>* rowmap = nullptr;   // moved into the bit state while the chain runs, handed back when it ends in CSR form
};

// shared front half of fgpu_expand / fgpu_expand_count: result stays on device
static fgpu_info expand_device(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                               const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops,
                               const uint64_t* dst_label_bitmap, fgpu_mat** result, u64* flops,
                               u64* count_only = nullptr /* [0] nnz, [1] checksum: no CSR is built when the
                                                            chain ends in bit form */,
                               bool want_checksum = true,
                               BitState* keep_bits = nullptr /* a chain that ends in bit form hands its state over
                                                                 (*result = nullptr) instead of emitting it */,
                               ChainSources* pre = nullptr /* the sources are on the device already (src_ids unused) */) {
    fgpu_mat* f = nullptr;
    if (pre) {
        f = pre->f;
        pre->f = nullptr;
    } else {
        FGPU_TRY(check_hops(m, dp, dm, nhops, nsrc));
        FGPU_TRY(upload_sources(ctx, &f, src_ids, nsrc, m[0]->nrows));
    }
    // Hops run on the sorted-CSR products until a hop's gather volume T makes the bit-parallel form
    // cheaper (bitexpand.hip): it costs one pass over A' gathering max(64, 8 W) bytes per entry,
    // whatever T is; the CSR product moves ~T entries several times.  Once dense, stay dense.
    const int mode = ctx->opt.expand_mode;
    bool bits = false;
    BitState bs;
    if (pre && pre->rowmap) bs.rowmap = std::move(*pre->rowmap);
    u64 T_known = 0;                      // traversed edges of f over T_for, when an earlier step already summed them
    const fgpu_mat *T_for = nullptr, *T_f = nullptr;
    for (int h = 0; h < nhops; ++h) {
        const fgpu_mat* mh = m[h];
        const fgpu_mat* dph = dp ? dp[h] : nullptr;
        const fgpu_mat* dmh = dm ? dm[h] : nullptr;
        if (h == 0 && mode != 2 && ctx->opt.expand_first_hop && !(dph && dph->nnz) && !(dmh && dmh->nnz)) {
            // clean first hop from one-entry rows: the source rows of m, copied (first_hop_rows above)
            fgpu_mat* c = nullptr;
            u64 T0 = 0, Tn = 0;
            bool have = false;
            const fgpu_info fi = first_hop_rows(ctx, f, mh, nhops > 1 ? m[1] : nullptr, &c, &T0, &Tn, &have);
            if (fi == FGPU_OK) {
                // (a first hop heavy enough for the bit form — T0 * ratio > nnz — is a batch of hub sources; it is still
                // correct as a copy, and the next hop's decision sees its true volume)
                if (flops) *flops += T0;
                mat_release(f);
                f = c;
                if (have) { T_known = Tn; T_for = m[1]; T_f = f; }
                continue;
            }
            if (fi != FGPU_NO_VALUE) { mat_release(f); return fi; }
        } else if (h == 0 && mode != 2 && ctx->opt.expand_first_hop) {
            // dirty first hop from one-entry rows: (m[u] \ dm[u]) U dp[u] row by row (first_hop_rows_dirty above)
            fgpu_mat* c = nullptr;
            u64 T0 = 0, Tn = 0;
            bool have = false;
            const fgpu_info fi = first_hop_rows_dirty(ctx, f, mh, dph, dmh, nhops > 1 ? m[1] : nullptr, &c, &T0, &Tn, &have);
            if (fi == FGPU_OK) {
                if (flops) *flops += T0;
                mat_release(f);
                f = c;
                // (the next hop's traversed edges over its BASE matrix only: with a dirty next layer its dp part is summed there)
                if (have) { T_known = Tn; T_for = m[1]; T_f = f; }
                continue;
            }
            if (fi != FGPU_NO_VALUE) { mat_release(f); return fi; }
        }
        if (!bits && mode != 1 && !mh->is_hyper() && mh->nnz && mh->nnz < 0x7FFFFFFFull && f->nnz) {
            const u64 w = (nsrc + 63) / 64;
            const u64 mem = 2ull * (mh->nrows > mh->ncols ? mh->nrows : mh->ncols) * (w <= 64 ? 2 * w : w + 64) * 8;
            bool go = (mode == 2);
            u64 T = 0;
            if ((mode == 0 && mem < (64ull << 30)) || go) {
                fgpu_info i = FGPU_OK;
                if (T_for == mh && T_f == f) T = T_known;          // summed while the first hop's rows were copied
                else i = mxm_flops(ctx, f, mh, &T);
                if (i != FGPU_OK) { mat_release(f); return i; }
                // measured: a sorted-CSR hop costs ~0.12-0.16 ns per gathered entry (RMAT-22: 6 ms at T = 36 M; RMAT-26:
                // 12 ms at T ~ 100 M), the first bit hop — the sparse pull, a flag probe per entry of A' — 5.5-6 ps per matrix
                // entry whatever the row width (1.45 ms at RMAT-24, 6.2 ms at RMAT-26): the chain leaves the CSR form once
                // T exceeds ~ nnz / 28.  (The round-1 rule, T * 1024 > nnz * row_bytes = nnz / 8 at 1024 rows, dated from a
                // 41 ps-per-entry pull and kept RMAT-26 batches in an 8.7 ms sort.)
                if (mode == 0) go = T * ctx->opt.expand_bits_ratio > mh->nnz;
            }
            if (go && ctx->opt.expand_compact && !pre) {
                // the empty source rows stay behind (compact_source_rows above); bp_to_csr finds the way back
                fgpu_mat* fc = nullptr;
                const u32 k_full = (u32)f->nrows;
                fgpu_info i = compact_source_rows(ctx, f, &fc, bs.rowmap, bs.rowrank);
                if (i != FGPU_OK) { mat_release(f); return i; }
                if (fc) { mat_release(f); f = fc; bs.nsrc_full = k_full; }
            }
            if (go) {
                // leaving the CSR form: a frontier whose out-edges are FEW beside the matrix is pushed into the bit state
                // (one 8-byte atomic per traversed edge: measured ~8 G/s on random words of a 2 GiB state — 33 M edges
                // took 4.06 ms where the sparse pull takes 2.64 ms, so the bar is a 32nd of the matrix, not a quarter),
                // a heavier one is scattered to a dense X and pulled
                const fgpu_mat* dph_ = dp ? dp[h] : nullptr;
                if (T * 32 <= mh->nnz) {
                    if (flops) {
                        *flops += T;
                        if (dph_ && dph_->nnz) {
                            u64 Tp = 0;
                            fgpu_info i = mxm_flops(ctx, f, dph_, &Tp);
                            if (i != FGPU_OK) { mat_release(f); return i; }
                            *flops += Tp;
                        }
                    }
                    fgpu_info i = bp_push_from_csr(ctx, bs, f, mh, dph_, dmh);
                    mat_release(f);
                    f = nullptr;
                    if (i != FGPU_OK) return i;
                    bits = true;
                    continue;                       // this hop is done
                }
                fgpu_info i = bp_from_csr(ctx, bs, f);
                mat_release(f);
                f = nullptr;
                if (i != FGPU_OK) return i;
                bs.pre_for = mh;                    // T is this hop's traversed-edge count over mh: no need to sum it again
                bs.pre_flops = T;
                bits = true;
            }
        }
        if (bits) {
            if (count_only && h == nhops - 1 && ctx->opt.expand_fuse_count) {
                // the last hop of a count-only chain counts its rows where they are produced: no result state is
                // written, zeroed or read back (bitexpand.hip bp_hop_count)
                DevBuf<u64> bm;
                if (dst_label_bitmap) {
                    const u64 nw = ((u64)mh->ncols + 63) / 64;
                    FGPU_TRY(bm.alloc(ctx, nw + 1));
                    FGPU_TRY(ctx->h2d(bm.p, dst_label_bitmap, nw * sizeof(u64)));
                }
                *result = nullptr;
                return bp_hop_count(ctx, bs, mh, dph, dmh, flops, dst_label_bitmap ? bm.p : nullptr, &count_only[0],
                                    want_checksum ? &count_only[1] : nullptr);
            }
            // (the hop right before a counting end of the chain writes its state in the layout that end gathers from)
            FGPU_TRY(bp_hop(ctx, bs, mh, dph, dmh, flops, (flops && h + 1 < nhops) ? m[h + 1] : nullptr,
                            (count_only && ctx->opt.expand_fuse_count && !keep_bits && h + 2 == nhops) ? m[h + 1] : nullptr));
            continue;
        }
        fgpu_mat* c = nullptr;
        fgpu_info i = delta_lmxm_device(ctx, &c, f, mh, dph, dmh, flops);
        mat_release(f);
        if (i != FGPU_OK) return i;
        f = c;
    }
    if (bits && keep_bits) {
        *keep_bits = std::move(bs);
        *result = nullptr;
        return FGPU_OK;
    }
    if (bits) {
        DevBuf<u64> bm;
        if (dst_label_bitmap) {
            const u64 nw = ((u64)bs.n + 63) / 64;
            FGPU_TRY(bm.alloc(ctx, nw + 1));
            FGPU_TRY(ctx->h2d(bm.p, dst_label_bitmap, nw * sizeof(u64)));
        }
        fgpu_info ri;
        if (count_only) {
            *result = nullptr;
            ri = bp_count(ctx, bs, dst_label_bitmap ? bm.p : nullptr, &count_only[0], want_checksum ? &count_only[1] : nullptr);
        } else {
            ri = bp_to_csr(ctx, bs, dst_label_bitmap ? bm.p : nullptr, result);
        }
        if (ri == FGPU_OK) bp_finish(ctx, bs);   // the next batch's state finds a zeroed block instead of a 2 GiB memset
        return ri;
    }
    if (dst_label_bitmap) {
        const u64 nc = f->ncols, nw = (nc + 63) / 64;
        DevBuf<u64> bm;
        fgpu_info i = bm.alloc(ctx, nw + 1);
        if (i == FGPU_OK) {
            i = ctx->h2d(bm.p, dst_label_bitmap, nw * sizeof(u64));
        }
        fgpu_mat* c = nullptr;
        if (i == FGPU_OK) i = filter_by_bitmap(ctx, &c, f, bm.p);
        if (i == FGPU_OK) i = (hipStreamSynchronize(ctx->stream()) == hipSuccess) ? FGPU_OK : FGPU_DEVICE;
        mat_release(f);
        if (i != FGPU_OK) return i;
        f = c;
    }
    if (pre && pre->rowmap) *pre->rowmap = std::move(bs.rowmap);
    *result = f;
    return FGPU_OK;
}

// ---- exact trail counts for k <= 2 -------------------------------------------------------------
// The reference evaluates `[*1..k]` with a per-row DFS over TRAILS (edge-unique paths, one row per path:
// CondVarLenTraverseOp, cond_var_len_traverse.rs:196-387).  For k <= 2 the number of trails between two nodes is a
// counting product over the effective adjacency: length 1 = the pair's multiplicity; length 2 = sum over the
// intermediates m of w(a, m) * w(m, b) — PLUS_PAIR on pattern matrices, PLUS_TIMES when the matrices carry per-pair
// multiplicities — minus the walks that use ONE edge twice, which at length 2 can only be a self-loop a -> a -> a
// (w(a, a) of them).  From k = 3 on walks that revisit an edge no longer have a product form, so the DFS stays.
__global__ __launch_bounds__(256) void row_weight_kernel(CsrView c1, CsrView a, const u64* __restrict__ aw,
                                                        const u32* __restrict__ src, u32 nrows, u64* __restrict__ w1) {
    // C1 = F0 x A with one source per row: row i of C1 IS row src[i] of A, entry for entry
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 i = wave; i < nrows; i += nwaves) {
        const u32 cb = c1.rowptr[i], ce = c1.rowptr[i + 1];
        if (cb == ce) continue;
        u32 ab, ae;
        row_range(a, src[i], ab, ae);
        for (u32 k = lane; k < ce - cb; k += 64) w1[cb + k] = aw ? aw[ab + k] : 1ull;
    }
}

// one wavefront per F1 entry (i, m): every b in A[m, :] adds w1 * w2 to the count of (i, b) in C2 (binary search in
// the sorted row), less the self-loop walk
__global__ __launch_bounds__(256) void trail2_count_kernel(CsrView f1, const u64* __restrict__ w1, CsrView a,
                                                          const u64* __restrict__ aw, CsrView c2,
                                                          const u32* __restrict__ src, u32 nnzf,
                                                          unsigned long long* __restrict__ counts) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 e = wave; e < nnzf; e += nwaves) {
        u32 lo = 0, hi = f1.nrows - 1;                 // row of entry e
        while (lo < hi) {
            const u32 mid = (lo + hi + 1) >> 1;
            if (f1.rowptr[mid] <= e) lo = mid; else hi = mid - 1;
        }
        const u32 i = lo, m = f1.colidx[e], s = src[i];
        const u64 wa = w1[e];
        u32 rb, re;
        row_range(a, m, rb, re);
        const u32 cb = c2.rowptr[i], ce = c2.rowptr[i + 1];
        for (u32 q = rb + lane; q < re; q += 64) {
            const u32 b = a.colidx[q];
            const u64 wb = aw ? aw[q] : 1ull;
            u64 add = wa * wb;
            if (m == s && b == s) add -= wa;           // a -> a -> a over the same edge: not a trail
            if (add == 0) continue;
            u32 l = cb, h = ce;
            while (l < h) {
                const u32 mid = (l + h) >> 1;
                if (c2.colidx[mid] < b) l = mid + 1; else h = mid;
            }
            atomicAdd(&counts[l], (unsigned long long)add);   // (i, b) is in C2 by construction
        }
    }
}

}  // namespace fgpu

using namespace fgpu;

// ---- (active_row, dest) columns built on the device (include/fgpu.h fgpu_expand_pairs) ------------------------------------
// rows the operator may keep: all of a free row, at most the one entry equal to the pinned destination of a pinned row
__global__ void pairs_len_kernel(const u32* __restrict__ rowptr, const u32* __restrict__ colidx, const u32* __restrict__ pin, u32 nsrc,
                                 u32* __restrict__ len, u32* __restrict__ pos) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nsrc) return;
    u32 l = 0, p = 0;
    if (i < nsrc) {
        const u32 b = rowptr[i], e = rowptr[i + 1];
        const u32 want = pin[i];
        if (want == 0xFFFFFFFFu) { l = e - b; p = b; }
        else if (want != 0xFFFFFFFEu) {                      // (0xFFFFFFFE: a pinned id beyond 32 bits — no vertex has it)
            u32 lo = b, hi = e;                              // destinations are ascending and unique per row
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (colidx[mid] < want) lo = mid + 1; else hi = mid;
            }
            if (lo < e && colidx[lo] == want) { l = 1; p = lo; }
        }
    }
    len[i] = l;
    pos[i] = p;
}
// a thread per OUTPUT entry: its row by a search of the (new) row pointers, its destination from the row's first kept entry
// (out_dest nullable: the 32-bit form of an unpinned batch ships the result's own column ids and only needs the row column)
template <typename RowT, typename DestT>
__global__ __launch_bounds__(256) void pairs_fill_kernel(const u32* __restrict__ newptr, const u32* __restrict__ pos,
                                                         const u32* __restrict__ colidx, u32 nsrc, u64 n,
                                                         RowT* __restrict__ out_row, DestT* __restrict__ out_dest) {
    for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; q < n; q += (u64)gridDim.x * 256) {
        u32 lo = 0, hi = nsrc - 1;                           // largest row with newptr[row] <= q
        while (lo < hi) {
            const u32 mid = (lo + hi + 1) >> 1;
            if (newptr[mid] <= q) lo = mid; else hi = mid - 1;
        }
        out_row[q] = (RowT)lo;
        if (out_dest) out_dest[q] = (DestT)colidx[pos[lo] + (u32)(q - newptr[lo])];
    }
}


// the last hop of an all-pinned batch from a SORTED-CSR frontier F: a wavefront per row i, a lane per entry u of F[i, :] — is
// dst[i] in m[u, :] / dm[u, :] / dp[u, :]?  (rows are ascending: binary search)
__global__ __launch_bounds__(256) void probe_csr_rows_kernel(CsrView f, CsrView m, CsrView dm, CsrView dp, const u32* __restrict__ dst,
                                                             u32 k, uint8_t* __restrict__ hit_m, uint8_t* __restrict__ hit_dm,
                                                             uint8_t* __restrict__ hit_dp) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    auto has = [](const CsrView& a, u32 u, u32 v) {
        if (!a.rowptr) return false;
        u32 b, e;
        row_range(a, u, b, e);
        while (b < e) {
            const u32 mid = (b + e) >> 1;
            if (a.colidx[mid] < v) b = mid + 1; else e = mid;
        }
        u32 b0, e0;
        row_range(a, u, b0, e0);
        return b < e0 && a.colidx[b] == v;
    };
    for (u32 i = wave; i < k; i += nwaves) {
        const u32 v = dst[i];
        if (v == 0xFFFFFFFFu) continue;
        const u32 fb = f.rowptr[i], fe = f.rowptr[i + 1];
        bool a = false, b = false, c = false;
        for (u32 q = fb + lane; q < fe; q += 64) {
            const u32 u = f.colidx[q];
            a = a || has(m, u, v);
            b = b || has(dm, u, v);
            c = c || has(dp, u, v);
        }
        if (__ballot(a) && lane == 0) hit_m[i] = 1;
        if (__ballot(b) && lane == 0) hit_dm[i] = 1;
        if (__ballot(c) && lane == 0) hit_dp[i] = 1;
    }
}

// nnz (+ checksum) of a chain that ended in CSR form; rowmap (nullable) = the call's row of every row of r
static fgpu_info count_csr_result(fgpu_ctx* ctx, const fgpu_mat* r, const u32* rowmap, u64* out_nnz, u64* checksum) {
    *out_nnz = r->nnz;
    if (!checksum) return FGPU_OK;
    *checksum = 0;
    if (!r->nnz) return FGPU_OK;
    DevBuf<u64> acc;
    FGPU_TRY(acc.alloc(ctx, 1));
    FGPU_HIP(hipMemsetAsync(acc.p, 0, sizeof(u64), ctx->stream()));
    if (r->nnz / r->nrows >= 1024 && r->nnz < 0xFFFFFFFFull) {
        u32 grid = cdiv(r->nnz, 256);
        if (grid > (u32)ctx->cus * 32) grid = ctx->cus * 32;
        hipLaunchKernelGGL(checksum_entries_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(r),
                           (u32)r->nrows, (u32)r->nnz, (unsigned long long*)acc.p, rowmap);
    } else {
        u32 grid = cdiv(r->nrows, 4);
        if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
        hipLaunchKernelGGL(checksum_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(r),
                           (u32)r->nrows, (unsigned long long*)acc.p, rowmap);
    }
    FGPU_HIP(hipGetLastError());
    return read_u64(ctx, acc.p, checksum);
}

// ---- whole-frontier calls (SURVEY.md §7 "hard part 1", §8d) ---------------------------------------------------------------
// The reference hands CondTraverse at most 1024 rows per mxm (BATCH_SIZE, graph/src/runtime/batch.rs:81; the loop of
// cond_traverse.rs:452-751) — at that size a launch of this chip is latency, and half of a label scan's sources have no
// out-edge at all (an R-MAT graph: 525 of the first 1024 :P sources), so a 1024-source batch fills 64 of the 128 bytes of
// the row the count hop gathers per entry of A' and a missed gather costs a whole 128-byte line either way.
// fgpu_expand_count therefore takes ALL sources of a scan (10^5 - 10^6 rows) in one call:
//   1. the sources are filtered on the device: a row takes part when its source has an entry in m[0] or dp[0] (every other
//      row's result is empty whatever follows), and the live rows are compacted — (source id, row of the call) pairs;
//   2. the live rows are cut into PASSES of `expand_scan_rows` rows (1024: 16 words = one 128-byte line per vertex) — the
//      width is picked from the LIVE rows, not from the caller's batch size;
//   3. the passes are dealt to `expand_scan_lanes` lanes of the context (the calling thread + worker threads, each on its own
//      stream and pool: a chain makes ~10 host-side decisions per pass — product sizes, the CSR -> bit-form switch, the
//      counts — during which ITS stream idles), so one pass's launch-bound head and tail run under another's count hop;
//   4. rows keep their identity: bit i of a pass stands for row rowmap[i] of the call, which is what the checksum hashes.
// The sums are order-independent (64-bit wrap-around adds), so the result equals the 1024-row calls' sums bit for bit.
__global__ void scan_live_kernel(const u32* __restrict__ ids, u32 n, CsrView m0, CsrView dp0, u32 has_dp, u32* __restrict__ flag) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    u32 live = 0;
    if (i < n && ids[i] != 0xFFFFFFFFu) {
        u32 b, e;
        row_range(m0, ids[i], b, e);
        live = e > b;
        if (!live && has_dp) { row_range(dp0, ids[i], b, e); live = e > b; }
    }
    flag[i] = live;
}
__global__ void scan_compact_kernel(const u32* __restrict__ ids, const u32* __restrict__ flag, const u32* __restrict__ pos, u32 n,
                                    u32* __restrict__ lid, u32* __restrict__ lrow) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) { lid[pos[i]] = ids[i]; lrow[pos[i]] = i; }
}
__global__ void scan_pass_kernel(const u32* __restrict__ lid, const u32* __restrict__ lrow, u32 first, u32 k, u32* __restrict__ rowptr,
                                 u32* __restrict__ colidx, u32* __restrict__ rowmap) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= k) rowptr[i] = i;
    if (i < k) { colidx[i] = lid[first + i]; rowmap[i] = lrow[first + i]; }
}

struct ScanJob {
    fgpu_ctx* ctx;
    const u32 *lid, *lrow;
    u32 nlive, pass_rows, npasses;
    const fgpu_mat* const* m; const fgpu_mat* const* dp; const fgpu_mat* const* dm;
    int nhops;
    const uint64_t* label;
    bool want_cs, want_flops;
    std::atomic<u32> next{0};
    std::atomic<bool> failed{false};
    std::mutex mu;                       // the sums and the first error
    u64 nnz = 0, cs = 0, flops = 0;
    fgpu_info err = FGPU_OK;
    std::string msg;
};

static fgpu_info scan_one_pass(ScanJob& j, u32 p, u64* nnz, u64* cs, u64* fl) {
    fgpu_ctx* ctx = j.ctx;
    const u32 first = p * j.pass_rows;
    const u32 k = std::min(j.pass_rows, j.nlive - first);
    fgpu_mat* f = nullptr;
    DevBuf<u32> rm;
    FGPU_TRY(rm.alloc(ctx, k));
    FGPU_TRY(mat_alloc(ctx, &f, k, j.m[0]->nrows, k, false, 0, false));
    hipLaunchKernelGGL(scan_pass_kernel, dim3(cdiv((u64)k + 1, 256)), dim3(256), 0, ctx->stream(), j.lid, j.lrow, first, k, f->rowptr,
                       f->colidx, rm.p);
    if (hipGetLastError() != hipSuccess) { mat_release(f); set_error("expand scan: device call failed"); return FGPU_DEVICE; }
    ChainSources pre;
    pre.f = f;
    pre.rowmap = &rm;
    fgpu_mat* r = nullptr;
    u64 cnt[2] = {0, 0};
    *fl = 0;
    FGPU_TRY(expand_device(ctx, nullptr, k, j.m, j.dp, j.dm, j.nhops, j.label, &r, j.want_flops ? fl : nullptr, cnt, j.want_cs, nullptr, &pre));
    if (!r) { *nnz = cnt[0]; *cs = cnt[1]; return FGPU_OK; }
    const fgpu_info i = count_csr_result(ctx, r, rm.p, nnz, j.want_cs ? cs : nullptr);
    mat_release(r);
    return i;
}

static void scan_worker(ScanJob* j) {
    u64 nnz = 0, cs = 0, flops = 0;
    fgpu_info err = FGPU_OK;
    for (;;) {
        if (j->failed.load(std::memory_order_relaxed)) break;
        const u32 p = j->next.fetch_add(1, std::memory_order_relaxed);
        if (p >= j->npasses) break;
        u64 a = 0, b = 0, c = 0;
        err = scan_one_pass(*j, p, &a, &b, &c);
        if (err != FGPU_OK) { j->failed.store(true); break; }
        nnz += a; cs += b; flops += c;
    }
    // (a worker thread's stream has nothing pending here: every pass ends with the read-back of its sums)
    std::lock_guard<std::mutex> g(j->mu);
    j->nnz += nnz; j->cs += cs; j->flops += flops;
    if (err != FGPU_OK && j->err == FGPU_OK) { j->err = err; j->msg = get_error(); }
}

static fgpu_info expand_count_scan(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                                   const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops,
                                   const uint64_t* dst_label_bitmap, uint64_t* out_nnz, uint64_t* checksum, uint64_t* flops) {
    FGPU_TRY(check_hops(m, dp, dm, nhops, nsrc));
    const u64 ncols0 = m[0]->nrows;
    const u32 n = (u32)nsrc;
    std::vector<u32> ids(n);
    for (u32 i = 0; i < n; ++i) {
        if (src_ids[i] == UINT64_MAX) { ids[i] = 0xFFFFFFFFu; continue; }
        FGPU_REQUIRE(src_ids[i] < ncols0, FGPU_OUT_OF_BOUNDS, "expand: source %u = %llu >= %llu", i,
                     (unsigned long long)src_ids[i], (unsigned long long)ncols0);
        ids[i] = (u32)src_ids[i];
    }
    DevBuf<u32> dids, flag, pos, lid, lrow;
    FGPU_TRY(dids.alloc(ctx, n));
    FGPU_TRY(flag.alloc(ctx, (size_t)n + 1));
    FGPU_TRY(pos.alloc(ctx, (size_t)n + 1));
    FGPU_TRY(ctx->h2d(dids.p, ids.data(), (size_t)n * sizeof(u32)));
    const fgpu_mat* dp0 = dp && dp[0] && dp[0]->nnz ? dp[0] : nullptr;
    hipLaunchKernelGGL(scan_live_kernel, dim3(cdiv((u64)n + 1, 256)), dim3(256), 0, ctx->stream(), (const u32*)dids.p, n, view_of(m[0]),
                       dp0 ? view_of(dp0) : view_of(m[0]), dp0 ? 1u : 0u, flag.p);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(scan_u32(ctx, flag.p, pos.p, (u64)n + 1, nullptr));
    u32 nlive = 0;
    FGPU_TRY(read_u32(ctx, pos.p + n, &nlive));
    *out_nnz = 0;
    if (checksum) *checksum = 0;
    ctx->scan_last_live.store(nlive, std::memory_order_relaxed);
    ctx->scan_last_passes.store(0, std::memory_order_relaxed);
    if (nlive == 0) return FGPU_OK;
    FGPU_TRY(lid.alloc(ctx, nlive));
    FGPU_TRY(lrow.alloc(ctx, nlive));
    hipLaunchKernelGGL(scan_compact_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream(), (const u32*)dids.p, (const u32*)flag.p,
                       (const u32*)pos.p, n, lid.p, lrow.p);
    FGPU_HIP(hipGetLastError());
    FGPU_HIP(hipStreamSynchronize(ctx->stream()));           // the other lanes read the live list
    ScanJob j;
    j.ctx = ctx; j.lid = lid.p; j.lrow = lrow.p; j.nlive = nlive;
    j.pass_rows = (u32)ctx->opt.expand_scan_rows;
    j.npasses = cdiv(nlive, j.pass_rows);
    j.m = m; j.dp = dp; j.dm = dm; j.nhops = nhops; j.label = dst_label_bitmap;
    j.want_cs = checksum != nullptr; j.want_flops = flops != nullptr;
    ctx->scan_last_passes.store(j.npasses, std::memory_order_relaxed);
    u32 lanes = (u32)ctx->opt.expand_scan_lanes;
    if (lanes > j.npasses) lanes = j.npasses;
    if (lanes < 1) lanes = 1;
    std::vector<std::thread> th;
    th.reserve(lanes);
    ctx->scan_active.fetch_add((int)lanes, std::memory_order_relaxed);
    for (u32 t = 1; t < lanes; ++t) {
        try { th.emplace_back(scan_worker, &j); } catch (...) { break; }   // (fewer lanes: the passes are pulled, not assigned)
    }
    scan_worker(&j);
    for (auto& t : th) t.join();
    ctx->scan_active.fetch_sub((int)lanes, std::memory_order_relaxed);
    if (j.err != FGPU_OK) { set_error("%s", j.msg.c_str()); return j.err; }
    *out_nnz = j.nnz;
    if (checksum) *checksum = j.cs;
    if (flops) *flops = j.flops;
    return FGPU_OK;
}

extern "C" {

static fgpu_info mxm_impl(fgpu_ctx* ctx, fgpu_mat** c, const fgpu_mat* f, const fgpu_mat* b) {
    FGPU_REQUIRE(ctx && c && f && b, FGPU_NULL_POINTER, "fgpu_mxm: NULL argument");
    return mxm_device(ctx, c, f, b, nullptr);
}

static fgpu_info delta_lmxm_impl(fgpu_ctx* ctx, fgpu_mat** c, const fgpu_mat* f, const fgpu_mat* m, const fgpu_mat* dp,
                          const fgpu_mat* dm) {
    FGPU_REQUIRE(ctx && c && f && m, FGPU_NULL_POINTER, "fgpu_delta_lmxm: NULL argument");
    FGPU_REQUIRE(!dp || (dp->nrows == m->nrows && dp->ncols == m->ncols), FGPU_DIM_MISMATCH,
                 "fgpu_delta_lmxm: dp dims differ from m");
    FGPU_REQUIRE(!dm || (dm->nrows == m->nrows && dm->ncols == m->ncols), FGPU_DIM_MISMATCH,
                 "fgpu_delta_lmxm: dm dims differ from m");
    return delta_lmxm_device(ctx, c, f, m, dp, dm, nullptr);
}

fgpu_info fgpu_expand(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                      const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops,
                      const uint64_t* dst_label_bitmap, uint64_t** out_rowptr, uint64_t** out_dest,
                      uint64_t* out_nnz, uint64_t* flops) {
    FGPU_REQUIRE(ctx && out_rowptr && out_dest && out_nnz, FGPU_NULL_POINTER, "fgpu_expand: NULL argument");
    FGPU_REQUIRE(nsrc == 0 || src_ids, FGPU_NULL_POINTER, "fgpu_expand: NULL src_ids");
    if (flops) *flops = 0;
    fgpu_mat* r = nullptr;
    FGPU_TRY(expand_device(ctx, src_ids, nsrc, m, dp, dm, nhops, dst_label_bitmap, &r, flops));
    uint64_t *vals = nullptr;
    fgpu_info i = fgpu_mat_export_csr(ctx, r, out_rowptr, out_dest, &vals, out_nnz);
    mat_release(r);
    return i;
}

static fgpu_info expand_pairs_impl(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                                   const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                                   const uint64_t* pinned_dest, int row_bits, int dest_bits, void** out_row, void** out_dest, uint64_t* out_n,
                                   uint64_t* flops) {
    FGPU_REQUIRE(ctx && out_row && out_dest && out_n, FGPU_NULL_POINTER, "fgpu_expand_pairs: NULL argument");
    FGPU_REQUIRE(nsrc == 0 || src_ids, FGPU_NULL_POINTER, "fgpu_expand_pairs: NULL src_ids");
    FGPU_REQUIRE(row_bits == 16 || row_bits == 32, FGPU_INVALID, "fgpu_expand_pairs: row_bits must be 16 or 32");
    FGPU_REQUIRE(row_bits == 32 || nsrc <= 65536, FGPU_INVALID, "fgpu_expand_pairs: %llu source rows do not fit 16-bit row indices",
                 (unsigned long long)nsrc);
    *out_row = nullptr; *out_dest = nullptr; *out_n = 0;
    if (flops) *flops = 0;
    fgpu_mat* r = nullptr;
    FGPU_TRY(expand_device(ctx, src_ids, nsrc, m, dp, dm, nhops, dst_label_bitmap, &r, flops));
    struct Rel { fgpu_mat* r; ~Rel() { if (r) mat_release(r); } } rel{r};
    if (nsrc == 0 || r->nnz == 0) return FGPU_OK;
    // (the kernels below index the result's row pointers densely, nsrc + 1 of them)
    FGPU_REQUIRE(!r->is_hyper() && r->nrows == nsrc, FGPU_INVALID, "fgpu_expand_pairs: the chain's result is not a dense-row CSR of %llu rows",
                 (unsigned long long)nsrc);
    hipStream_t st = ctx->stream();
    const u32 k = (u32)nsrc;
    DevBuf<u32> pin, len, pos, newptr, total;
    FGPU_TRY(pin.alloc(ctx, k + 1));
    FGPU_TRY(len.alloc(ctx, (size_t)k + 1));
    FGPU_TRY(pos.alloc(ctx, (size_t)k + 1));
    FGPU_TRY(newptr.alloc(ctx, (size_t)k + 1));
    FGPU_TRY(total.alloc(ctx, 1));
    u64 n = r->nnz;
    const u32* rowptr = r->rowptr;                           // (no pinned row: the result's own row pointers and entries)
    const u32* first = r->rowptr;
    if (pinned_dest) {
        std::vector<u32> hp(k);
        bool any = false;
        for (u32 i = 0; i < k; ++i) {
            hp[i] = pinned_dest[i] == ~0ull ? 0xFFFFFFFFu : pinned_dest[i] >= 0xFFFFFFFEull ? 0xFFFFFFFEu : (u32)pinned_dest[i];
            any = any || hp[i] != 0xFFFFFFFFu;
        }
        if (any) {
            FGPU_TRY(ctx->h2d(pin.p, hp.data(), (size_t)k * sizeof(u32)));
            hipLaunchKernelGGL(pairs_len_kernel, dim3(cdiv((u64)k + 1, 256)), dim3(256), 0, st, (const u32*)r->rowptr, (const u32*)r->colidx,
                               (const u32*)pin.p, k, len.p, pos.p);
            FGPU_HIP(hipGetLastError());
            FGPU_TRY(scan_u32(ctx, len.p, newptr.p, (u64)k + 1, total.p));
            u32 t = 0;
            FGPU_TRY(read_u32(ctx, total.p, &t));
            n = t;
            rowptr = newptr.p;
            first = pos.p;
        }
    }
    if (n == 0) return FGPU_OK;
    const size_t rb = row_bits / 8, db = dest_bits / 8;
    // the 32-bit form of a batch without pinned rows: the destination column IS the result's column-id array
    const bool dest_is_colidx = dest_bits == 32 && rowptr == r->rowptr;
    DevBuf<uint8_t> ddest, drow;
    if (!dest_is_colidx) FGPU_TRY(ddest.alloc(ctx, n * db));
    FGPU_TRY(drow.alloc(ctx, n * rb));
    {
        ProfScope ps(ctx, "pairs_fill_kernel", n * (4 + (dest_is_colidx ? 0 : db) + rb));
        u32 grid = cdiv(n, 256 * 4);
        if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
        if (!grid) grid = 1;
#define PAIRS_FILL(RT, DT)                                                                                                     \
        hipLaunchKernelGGL((pairs_fill_kernel<RT, DT>), dim3(grid), dim3(256), 0, st, rowptr, first, (const u32*)r->colidx, k, n, \
                           (RT*)drow.p, (DT*)ddest.p)
        if (row_bits == 16) { if (dest_bits == 64) PAIRS_FILL(uint16_t, u64); else PAIRS_FILL(uint16_t, u32); }
        else { if (dest_bits == 64) PAIRS_FILL(u32, u64); else PAIRS_FILL(u32, u32); }
#undef PAIRS_FILL
        FGPU_HIP(hipGetLastError());
    }
    void* hrow = ctx->result_alloc(n * rb);
    void* hdest = ctx->result_alloc(n * db);
    if (!hrow || !hdest) {
        ctx->host_free(hrow); ctx->host_free(hdest);
        set_error("fgpu_expand_pairs: host allocation failed");
        return FGPU_OOM;
    }
    fgpu_info i = ctx->d2h(hdest, dest_is_colidx ? (const void*)r->colidx : (const void*)ddest.p, n * db);
    if (i == FGPU_OK) i = ctx->d2h(hrow, drow.p, n * rb);
    if (i != FGPU_OK) { ctx->host_free(hrow); ctx->host_free(hdest); return i; }
    *out_row = hrow;
    *out_dest = hdest;
    *out_n = n;
    return FGPU_OK;
}

fgpu_info fgpu_expand_pairs(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                            const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                            const uint64_t* pinned_dest, int row_bits, void** out_row, uint64_t** out_dest, uint64_t* out_n,
                            uint64_t* flops) {
    return expand_pairs_impl(ctx, src_ids, nsrc, m, dp, dm, nhops, dst_label_bitmap, pinned_dest, row_bits, 64, out_row, (void**)out_dest,
                             out_n, flops);
}

fgpu_info fgpu_expand_pairs32(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                              const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                              const uint64_t* pinned_dest, int row_bits, void** out_row, uint32_t** out_dest, uint64_t* out_n,
                              uint64_t* flops) {
    return expand_pairs_impl(ctx, src_ids, nsrc, m, dp, dm, nhops, dst_label_bitmap, pinned_dest, row_bits, 32, out_row, (void**)out_dest,
                             out_n, flops);
}

fgpu_info fgpu_expand_probe(fgpu_ctx* ctx, const uint64_t* src_ids, const uint64_t* dst_ids, uint64_t nsrc, const fgpu_mat* const* m,
                            const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                            uint8_t* present, uint64_t* flops) {
    FGPU_REQUIRE(ctx && present && (nsrc == 0 || (src_ids && dst_ids)), FGPU_NULL_POINTER, "fgpu_expand_probe: NULL argument");
    if (flops) *flops = 0;
    FGPU_TRY(check_hops(m, dp, dm, nhops, nsrc));
    if (nsrc == 0) return FGPU_OK;
    const u32 k = (u32)nsrc;
    memset(present, 0, k);
    const fgpu_mat* ml = m[nhops - 1];
    const fgpu_mat* dpl = dp ? dp[nhops - 1] : nullptr;
    const fgpu_mat* dml = dm ? dm[nhops - 1] : nullptr;
    // the chain up to the last hop: a sorted-CSR frontier, or the bit state it ended in
    fgpu_mat* f = nullptr;
    BitState bs;
    if (nhops == 1) FGPU_TRY(upload_sources(ctx, &f, src_ids, nsrc, m[0]->nrows));
    else FGPU_TRY(expand_device(ctx, src_ids, nsrc, m, dp, dm, nhops - 1, nullptr, &f, flops, nullptr, true, &bs));
    struct Rel { fgpu_mat* f; ~Rel() { if (f) mat_release(f); } } rel{f};
    // destinations (a vertex the last matrix does not have, or one the label filter drops: no match), the bit of every row
    const u64 ncols = ml->ncols;
    std::vector<u32> hdst(k), hbit(k), order(k), sdst(k), srow(k);
    std::vector<u32> rank;
    if (!f && bs.nsrc_full) {
        rank.resize((size_t)bs.nsrc_full + 1);
        FGPU_TRY(ctx->d2h(rank.data(), bs.rowrank.p, rank.size() * sizeof(u32)));
    }
    for (u32 i = 0; i < k; ++i) {
        u64 v = dst_ids[i];
        if (v >= ncols || src_ids[i] == UINT64_MAX) v = ~0ull;
        else if (dst_label_bitmap && !((dst_label_bitmap[v >> 6] >> (v & 63)) & 1ull)) v = ~0ull;
        hdst[i] = v == ~0ull ? 0xFFFFFFFFu : (u32)v;
        if (rank.empty()) hbit[i] = i;
        else hbit[i] = rank[i + 1] > rank[i] ? rank[i] : 0xFFFFFFFFu;   // (a source row dropped before the chain went to bits is empty)
        order[i] = i;
    }
    std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return hdst[a] < hdst[b] || (hdst[a] == hdst[b] && a < b); });
    for (u32 p = 0; p < k; ++p) { sdst[p] = hdst[order[p]]; srow[p] = order[p]; }
    DevBuf<u32> d_dst, d_bit, d_sdst, d_srow;
    DevBuf<uint8_t> hits;
    FGPU_TRY(d_dst.alloc(ctx, k)); FGPU_TRY(d_bit.alloc(ctx, k)); FGPU_TRY(d_sdst.alloc(ctx, k)); FGPU_TRY(d_srow.alloc(ctx, k));
    FGPU_TRY(hits.alloc(ctx, (size_t)3 * k));
    FGPU_TRY(ctx->h2d(d_dst.p, hdst.data(), (size_t)k * 4));
    FGPU_TRY(ctx->h2d(d_bit.p, hbit.data(), (size_t)k * 4));
    FGPU_TRY(ctx->h2d(d_sdst.p, sdst.data(), (size_t)k * 4));
    FGPU_TRY(ctx->h2d(d_srow.p, srow.data(), (size_t)k * 4));
    FGPU_HIP(hipMemsetAsync(hits.p, 0, (size_t)3 * k, ctx->stream()));
    if (f) {
        if (f->nnz) {
            CsrView none;
            none.rowptr = nullptr; none.colidx = nullptr; none.hrows = nullptr; none.nvec = 0; none.nrows = 0;
            u32 grid = cdiv(k, 4);
            if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
            hipLaunchKernelGGL(probe_csr_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(f), view_of(ml),
                               (dml && dml->nnz) ? view_of(dml) : none, (dpl && dpl->nnz) ? view_of(dpl) : none, (const u32*)d_dst.p, k,
                               hits.p, hits.p + k, hits.p + 2 * (size_t)k);
            FGPU_HIP(hipGetLastError());
        }
    } else {
        FGPU_TRY(bp_probe_rows(ctx, bs, ml, dpl, dml, d_dst.p, d_bit.p, d_sdst.p, d_srow.p, k, hits.p, hits.p + k, hits.p + 2 * (size_t)k));
    }
    std::vector<uint8_t> h((size_t)3 * k);
    FGPU_TRY(ctx->d2h(h.data(), hits.p, (size_t)3 * k));
    if (!f) bp_finish(ctx, bs);
    // (F·m)<not (F·dm)> U (F·dp), one entry of it: the row-level mask of Matrix::delta_lmxm (matrix.rs:1343-1361)
    for (u32 i = 0; i < k; ++i) present[i] = ((h[i] && !h[k + i]) || h[2 * (size_t)k + i]) ? 1 : 0;
    return FGPU_OK;
}

// ---- streamed result (include/fgpu.h fgpu_expand_stream_*) ----------------------------------------------------------
struct fgpu_expand_stream {
    static constexpr int NS = 4;
    fgpu_ctx* ctx = nullptr;
    hipStream_t st = nullptr;        // the opening thread's lane
    fgpu_mat* r = nullptr;           // F on the device
    std::vector<u32> rp;             // its row pointers (nsrc + 1)
    u64 nsrc = 0, chunk_rows = 0;
    int width = 8;                   // bytes per destination id handed out
    size_t cap = 0;                  // entries per slot
    struct Slot {
        void* host = nullptr;        // pinned, cap * width bytes
        u64* wide = nullptr;         // device staging of the widened ids (width 8)
        std::vector<u64> rowptr;     // relative offsets of the chunk's rows
        hipEvent_t ev = nullptr;
        u64 first = 0, nrows = 0;
    } slot[NS];
    u64 enq_row = 0;                 // first row not yet enqueued
    int head = 0, inflight = 0;      // slot of the oldest enqueued chunk; chunks enqueued and not yet handed out
    bool held = false;               // the caller is reading slot (head - 1)
};

static fgpu_info stream_enqueue(fgpu_expand_stream* s, int k) {
    auto& sl = s->slot[k];
    const u64 first = s->enq_row;
    const u64 nr = s->nsrc - first < s->chunk_rows ? s->nsrc - first : s->chunk_rows;
    const u64 b = s->rp[first], e = s->rp[first + nr], n = e - b;
    sl.first = first; sl.nrows = nr;
    sl.rowptr.resize(nr + 1);
    for (u64 i = 0; i <= nr; ++i) sl.rowptr[i] = (u64)s->rp[first + i] - b;
    if (n) {
        if (s->width == 8) {
            FGPU_TRY(fgpu::widen_on_device(s->ctx, sl.wide, s->r->colidx + b, n));
            FGPU_HIP(hipMemcpyAsync(sl.host, sl.wide, n * 8, hipMemcpyDeviceToHost, s->st));
        } else {
            FGPU_HIP(hipMemcpyAsync(sl.host, s->r->colidx + b, n * 4, hipMemcpyDeviceToHost, s->st));
        }
    }
    FGPU_HIP(hipEventRecord(sl.ev, s->st));
    s->enq_row = first + nr;
    ++s->inflight;
    return FGPU_OK;
}

fgpu_info fgpu_expand_stream_close(fgpu_expand_stream* s) {
    if (!s) return FGPU_OK;
    fgpu_ctx* ctx = s->ctx;
    if (s->st) (void)hipStreamSynchronize(s->st);
    for (auto& sl : s->slot) {
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        if (sl.host) ctx->host_free(sl.host);
        if (sl.wide) ctx->dev_free(sl.wide);
    }
    if (s->r) mat_release(s->r);
    delete s;
    return FGPU_OK;
}

fgpu_info fgpu_expand_stream_open(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                                  const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops,
                                  const uint64_t* dst_label_bitmap, uint64_t chunk_rows, int dest_bits,
                                  fgpu_expand_stream** out, uint64_t* nnz, uint64_t* flops) {
    FGPU_REQUIRE(ctx && out, FGPU_NULL_POINTER, "fgpu_expand_stream_open: NULL argument");
    FGPU_REQUIRE(nsrc == 0 || src_ids, FGPU_NULL_POINTER, "fgpu_expand_stream_open: NULL src_ids");
    FGPU_REQUIRE(dest_bits == 32 || dest_bits == 64, FGPU_INVALID, "fgpu_expand_stream_open: dest_bits must be 32 or 64");
    FGPU_REQUIRE(chunk_rows >= 1, FGPU_INVALID, "fgpu_expand_stream_open: chunk_rows must be >= 1");
    *out = nullptr;
    if (flops) *flops = 0;
    fgpu_mat* r = nullptr;
    FGPU_TRY(expand_device(ctx, src_ids, nsrc, m, dp, dm, nhops, dst_label_bitmap, &r, flops));
    fgpu_expand_stream* s = new (std::nothrow) fgpu_expand_stream();
    if (!s) { mat_release(r); set_error("fgpu_expand_stream_open: out of host memory"); return FGPU_OOM; }
    s->ctx = ctx; s->st = ctx->stream(); s->r = r; s->nsrc = nsrc; s->chunk_rows = chunk_rows; s->width = dest_bits / 8;
    fgpu_info i = FGPU_OK;
    s->rp.assign(nsrc + 1, 0);
    if (r->nnz) {
        if (r->is_hyper()) {          // (expand_device returns one stored row per source row; guard the other form)
            std::vector<u32> hr(r->nvec), srp((size_t)r->nvec + 1);
            i = ctx->d2h(srp.data(), r->rowptr, srp.size() * sizeof(u32));
            if (i == FGPU_OK && r->nvec) i = ctx->d2h(hr.data(), r->hrows, (size_t)r->nvec * sizeof(u32));
            if (i == FGPU_OK) {
                for (u32 k = 0; k < r->nvec; ++k) s->rp[hr[k] + 1] = srp[k + 1] - srp[k];
                for (u64 q = 0; q < nsrc; ++q) s->rp[q + 1] += s->rp[q];
            }
        } else {
            i = ctx->d2h(s->rp.data(), r->rowptr, (nsrc + 1) * sizeof(u32));
        }
    }
    if (i == FGPU_OK) {
        for (u64 f = 0; f < nsrc; f += chunk_rows) {
            const u64 l = f + chunk_rows < nsrc ? f + chunk_rows : nsrc;
            const size_t n = (size_t)s->rp[l] - s->rp[f];
            if (n > s->cap) s->cap = n;
        }
        const u64 nchunks = nsrc ? (nsrc + chunk_rows - 1) / chunk_rows : 0;
        const int use = nchunks < (u64)fgpu_expand_stream::NS ? (int)nchunks : fgpu_expand_stream::NS;
        for (int k = 0; k < use && i == FGPU_OK; ++k) {
            auto& sl = s->slot[k];
            sl.host = ctx->pinned_alloc((s->cap ? s->cap : 1) * s->width);
            if (!sl.host) { set_error("fgpu_expand_stream_open: pinned host memory exhausted"); i = FGPU_OOM; break; }
            if (s->width == 8) i = ctx->dev_alloc((void**)&sl.wide, (s->cap ? s->cap : 1) * 8);
            if (i == FGPU_OK && hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) {
                set_error("fgpu_expand_stream_open: hipEventCreate failed");
                i = FGPU_DEVICE;
            }
        }
        for (int k = 0; k < use && i == FGPU_OK && s->enq_row < nsrc; ++k) i = stream_enqueue(s, k);
    }
    if (i != FGPU_OK) { fgpu_expand_stream_close(s); return i; }
    if (nnz) *nnz = r->nnz;
    *out = s;
    return FGPU_OK;
}

fgpu_info fgpu_expand_stream_next(fgpu_expand_stream* s, uint64_t* first_row, uint64_t* nrows, const uint64_t** rowptr,
                                  const void** dest) {
    FGPU_REQUIRE(s && first_row && nrows && rowptr && dest, FGPU_NULL_POINTER, "fgpu_expand_stream_next: NULL argument");
    FGPU_REQUIRE(s->ctx->stream() == s->st, FGPU_INVALID, "fgpu_expand_stream_next: a stream belongs to the thread that opened it");
    constexpr int NS = fgpu_expand_stream::NS;
    if (s->held) {                         // the chunk handed out last is done with: its slot takes the next one
        s->held = false;
        const int freed = (s->head + NS - 1) % NS;
        if (s->enq_row < s->nsrc) FGPU_TRY(stream_enqueue(s, freed));
    }
    *nrows = 0; *first_row = s->nsrc; *rowptr = nullptr; *dest = nullptr;
    if (s->inflight == 0) return FGPU_NO_VALUE;
    auto& sl = s->slot[s->head];
    FGPU_HIP(hipEventSynchronize(sl.ev));
    *first_row = sl.first; *nrows = sl.nrows; *rowptr = sl.rowptr.data(); *dest = sl.host;
    s->head = (s->head + 1) % NS;
    --s->inflight;
    s->held = true;
    return FGPU_OK;
}

fgpu_info fgpu_expand32(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                        const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops,
                        const uint64_t* dst_label_bitmap, uint32_t** out_rowptr, uint32_t** out_dest,
                        uint64_t* out_nnz, uint64_t* flops) {
    FGPU_REQUIRE(ctx && out_rowptr && out_dest && out_nnz, FGPU_NULL_POINTER, "fgpu_expand32: NULL argument");
    FGPU_REQUIRE(nsrc == 0 || src_ids, FGPU_NULL_POINTER, "fgpu_expand32: NULL src_ids");
    *out_rowptr = nullptr; *out_dest = nullptr; *out_nnz = 0;
    if (flops) *flops = 0;
    fgpu_mat* r = nullptr;
    FGPU_TRY(expand_device(ctx, src_ids, nsrc, m, dp, dm, nhops, dst_label_bitmap, &r, flops));
    struct Rel { fgpu_mat* r; ~Rel() { if (r) mat_release(r); } } rel{r};
    FGPU_REQUIRE(!r->is_hyper() && r->nrows == nsrc, FGPU_INVALID, "fgpu_expand32: the chain's result is not a dense-row CSR");
    // the device arrays as they are: two DMAs into pinned result blocks, nothing widened anywhere
    u32* hrp = (u32*)ctx->result_alloc((nsrc + 1) * sizeof(u32));
    u32* hci = (u32*)ctx->result_alloc((r->nnz ? r->nnz : 1) * sizeof(u32));
    fgpu_info i = (hrp && hci) ? FGPU_OK : FGPU_OOM;
    if (i == FGPU_OK) i = ctx->d2h(hrp, r->rowptr, (nsrc + 1) * sizeof(u32));
    if (i == FGPU_OK && r->nnz) i = ctx->d2h(hci, r->colidx, r->nnz * sizeof(u32));
    if (i != FGPU_OK) {
        ctx->host_free(hrp); ctx->host_free(hci);
        if (i == FGPU_OOM) set_error("fgpu_expand32: host allocation failed");
        return i;
    }
    *out_rowptr = hrp; *out_dest = hci; *out_nnz = r->nnz;
    return FGPU_OK;
}

fgpu_info fgpu_expand_mat(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                          const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops,
                          const uint64_t* dst_label_bitmap, fgpu_mat** out, uint64_t* flops) {
    FGPU_REQUIRE(ctx && out, FGPU_NULL_POINTER, "fgpu_expand_mat: NULL argument");
    FGPU_REQUIRE(nsrc == 0 || src_ids, FGPU_NULL_POINTER, "fgpu_expand_mat: NULL src_ids");
    if (flops) *flops = 0;
    fgpu_mat* r = nullptr;
    FGPU_TRY(expand_device(ctx, src_ids, nsrc, m, dp, dm, nhops, dst_label_bitmap, &r, flops));
    fgpu_info i = ctx->publish();   // the new handle may go to another thread
    if (i != FGPU_OK) { mat_release(r); return i; }
    *out = r;
    return FGPU_OK;
}

fgpu_info fgpu_expand_count(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                            const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops,
                            const uint64_t* dst_label_bitmap, uint64_t* out_nnz, uint64_t* checksum,
                            uint64_t* flops) {
    FGPU_REQUIRE(ctx && out_nnz, FGPU_NULL_POINTER, "fgpu_expand_count: NULL argument");
    FGPU_REQUIRE(nsrc == 0 || src_ids, FGPU_NULL_POINTER, "fgpu_expand_count: NULL src_ids");
    if (flops) *flops = 0;
    if (ctx->opt.expand_scan_min > 0 && nsrc > (u64)ctx->opt.expand_scan_min && ctx->opt.expand_mode != 1)
        return expand_count_scan(ctx, src_ids, nsrc, m, dp, dm, nhops, dst_label_bitmap, out_nnz, checksum, flops);
    fgpu_mat* r = nullptr;
    u64 cnt[2] = {0, 0};
    FGPU_TRY(expand_device(ctx, src_ids, nsrc, m, dp, dm, nhops, dst_label_bitmap, &r, flops, cnt, checksum != nullptr));
    if (!r) {   // the chain ended in bit form: counted there, no CSR was materialized
        *out_nnz = cnt[0];
        if (checksum) *checksum = cnt[1];
        return FGPU_OK;
    }
    const fgpu_info i = count_csr_result(ctx, r, nullptr, out_nnz, checksum);
    mat_release(r);
    return i;
}

fgpu_info fgpu_expand_levels(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc, const fgpu_mat* const* m,
                             const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops,
                             const uint64_t* dst_label_bitmap, uint64_t* hop_nnz, uint64_t* hop_checksum,
                             uint64_t* union_nnz, uint64_t* union_checksum, uint64_t* flops) {
    FGPU_REQUIRE(ctx && hop_nnz, FGPU_NULL_POINTER, "fgpu_expand_levels: NULL argument");
    FGPU_REQUIRE(nsrc == 0 || src_ids, FGPU_NULL_POINTER, "fgpu_expand_levels: NULL src_ids");
    FGPU_TRY(check_hops(m, dp, dm, nhops, nsrc));
    for (int h = 0; h < nhops; ++h)
        FGPU_REQUIRE(!m[h]->is_hyper() && m[h]->nnz < 0x7FFFFFFFull, FGPU_INVALID,
                     "fgpu_expand_levels: hop %d needs a non-hypersparse base matrix with nnz < 2^31", h);
    if (flops) *flops = 0;
    if (union_nnz) *union_nnz = 0;
    if (union_checksum) *union_checksum = 0;
    fgpu_mat* f = nullptr;
    FGPU_TRY(upload_sources(ctx, &f, src_ids, nsrc, m[0]->nrows));
    // the whole chain runs in bit form (one bit per source row): every hop is one pass over the cached
    // transpose whatever the frontier size, and the union over the hops is a word-wise OR
    BitState bs, un;
    fgpu_info i = bp_from_csr(ctx, bs, f);
    mat_release(f);
    if (i != FGPU_OK) return i;
    DevBuf<u64> bm;
    const u64* label_dev = nullptr;
    for (int h = 0; h < nhops; ++h) {
        FGPU_TRY(bp_hop(ctx, bs, m[h], dp ? dp[h] : nullptr, dm ? dm[h] : nullptr, flops));
        if (dst_label_bitmap && h == 0) {   // the destination label applies to every reported set
            const u64 nw = ((u64)bs.n + 63) / 64;
            FGPU_TRY(bm.alloc(ctx, nw + 1));
            FGPU_TRY(ctx->h2d(bm.p, dst_label_bitmap, nw * sizeof(u64)));
            label_dev = bm.p;
        }
        u64 n = 0, cs = 0;
        FGPU_TRY(bp_count(ctx, bs, label_dev, &n, hop_checksum ? &cs : nullptr));
        hop_nnz[h] = n;
        if (hop_checksum) hop_checksum[h] = cs;
        if (union_nnz || union_checksum) FGPU_TRY(bp_accumulate(ctx, un, bs));
    }
    if (union_nnz || union_checksum) {
        u64 n = 0, cs = 0;
        FGPU_TRY(bp_count(ctx, un, label_dev, &n, union_checksum ? &cs : nullptr));
        if (union_nnz) *union_nnz = n;
        if (union_checksum) *union_checksum = cs;
    }
    bp_finish(ctx, bs);
    return FGPU_OK;
}

}  // extern "C"

extern "C" fgpu_info fgpu_expand_trail_counts(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc,
                                              const fgpu_mat* const* m, const fgpu_mat* const* dp,
                                              const fgpu_mat* const* dm, int nhops, int weighted,
                                              uint64_t** out_rowptr, uint64_t** out_dest, uint64_t** out_count,
                                              uint64_t* out_nnz) {
    FGPU_REQUIRE(ctx && out_rowptr && out_dest && out_count && out_nnz, FGPU_NULL_POINTER, "fgpu_expand_trail_counts: NULL argument");
    FGPU_REQUIRE(nsrc == 0 || src_ids, FGPU_NULL_POINTER, "fgpu_expand_trail_counts: NULL src_ids");
    FGPU_REQUIRE(nhops == 1 || nhops == 2, FGPU_INVALID,
                 "fgpu_expand_trail_counts: trail counts have a product form for 1 or 2 hops only (cond_var_len_traverse.rs keeps the DFS beyond)");
    FGPU_TRY(check_hops(m, dp, dm, nhops, nsrc));
    for (u64 i = 0; i < nsrc; ++i)
        FGPU_REQUIRE(src_ids[i] != UINT64_MAX, FGPU_INVALID, "fgpu_expand_trail_counts: every row needs a source");
    // `-[:T*1..2]->` walks ONE relationship: the "same edge twice" correction of the two-hop count (the a -> a -> a walk
    // over one self-loop) is only meaningful when both hops read the same layers.  With different layers per hop no
    // edge can repeat and the subtraction would undercount, so that form is refused rather than answered wrongly.
    FGPU_REQUIRE(nhops == 1 || (m[0] == m[1] && (dp ? dp[0] : nullptr) == (dp ? dp[1] : nullptr) &&
                                (dm ? dm[0] : nullptr) == (dm ? dm[1] : nullptr)),
                 FGPU_INVALID, "fgpu_expand_trail_counts: both hops must read the same relationship layers (one var-length relationship)");
    // effective layers (m \ dm) U dp — real edges, no row-level mask quirk: these are counts of paths, not a delta_lmxm
    std::vector<const fgpu_mat*> eff(nhops, nullptr);
    std::vector<fgpu_mat*> owned;
    auto cleanup = [&]() { for (auto* x : owned) mat_release(x); };
    fgpu_info i = FGPU_OK;
    for (int h = 0; h < nhops && i == FGPU_OK; ++h) {
        const fgpu_mat* dph = dp ? dp[h] : nullptr;
        const fgpu_mat* dmh = dm ? dm[h] : nullptr;
        const bool dirty = (dph && dph->nnz) || (dmh && dmh->nnz);
        if (!dirty && !m[h]->is_hyper()) { eff[h] = m[h]; continue; }
        fgpu_mat* e = nullptr;
        i = mat_merge_entries(ctx, &e, m[h], dph, dmh, false, m[h]->nrows, m[h]->ncols, !weighted);
        if (i == FGPU_OK) { owned.push_back(e); eff[h] = e; }
    }
    if (i != FGPU_OK) { cleanup(); return i; }
    if (weighted)
        for (int h = 0; h < nhops; ++h)
            if (!eff[h]->vals) { cleanup(); set_error("fgpu_expand_trail_counts: weighted counts need UINT64 multiplicity layers"); return FGPU_INVALID; }
    fgpu_mat *f0 = nullptr, *c1 = nullptr, *c2 = nullptr;
    DevBuf<u32> dsrc;
    DevBuf<u64> w1, cnt;
    std::vector<u32> s32(nsrc);
    for (u64 k = 0; k < nsrc; ++k) s32[k] = (u32)src_ids[k];
    auto run = [&]() -> fgpu_info {
        FGPU_TRY(upload_sources(ctx, &f0, src_ids, nsrc, eff[0]->nrows));
        FGPU_TRY(dsrc.alloc(ctx, nsrc));
        FGPU_TRY(ctx->h2d(dsrc.p, s32.data(), nsrc * sizeof(u32)));
        FGPU_TRY(mxm_device(ctx, &c1, f0, eff[0], nullptr));
        FGPU_TRY(w1.alloc(ctx, c1->nnz));
        if (c1->nnz) {
            u32 grid = cdiv(nsrc, 4);
            if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
            hipLaunchKernelGGL(row_weight_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(c1), view_of(eff[0]),
                               weighted ? (const u64*)eff[0]->vals : (const u64*)nullptr, (const u32*)dsrc.p, (u32)nsrc, w1.p);
            FGPU_HIP(hipGetLastError());
        }
        const fgpu_mat* res = c1;
        const u64* res_cnt = w1.p;
        if (nhops == 2) {
            FGPU_TRY(mxm_device(ctx, &c2, c1, eff[1], nullptr));
            FGPU_TRY(cnt.alloc(ctx, c2->nnz));
            FGPU_HIP(hipMemsetAsync(cnt.p, 0, (size_t)(c2->nnz ? c2->nnz : 1) * sizeof(u64), ctx->stream()));
            if (c1->nnz && c2->nnz) {
                u32 grid = cdiv(c1->nnz, 4);
                if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
                hipLaunchKernelGGL(trail2_count_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(c1), (const u64*)w1.p,
                                   view_of(eff[1]), weighted ? (const u64*)eff[1]->vals : (const u64*)nullptr, view_of(c2),
                                   (const u32*)dsrc.p, (u32)c1->nnz, (unsigned long long*)cnt.p);
                FGPU_HIP(hipGetLastError());
            }
            res = c2;
            res_cnt = cnt.p;
        }
        // host hand-over: rowptr / dest as fgpu_expand does, counts beside them.  Pairs reached only through the
        // self-loop walk keep a zero count: they are walks, not trails, and are dropped here.
        std::vector<u32> rp((size_t)nsrc + 1), ci(res->nnz);
        std::vector<u64> cv(res->nnz);
        FGPU_TRY(ctx->d2h(rp.data(), res->rowptr, rp.size() * sizeof(u32)));
        if (res->nnz) {
            FGPU_TRY(ctx->d2h(ci.data(), res->colidx, res->nnz * sizeof(u32)));
            FGPU_TRY(ctx->d2h(cv.data(), res_cnt, res->nnz * sizeof(u64)));
        }
        FGPU_HIP(hipStreamSynchronize(ctx->stream()));
        u64 keep = 0;
        for (u64 k = 0; k < res->nnz; ++k) keep += cv[k] != 0;
        u64* orp = (u64*)ctx->host_alloc((nsrc + 1) * sizeof(u64));
        u64* od = (u64*)ctx->host_alloc((keep ? keep : 1) * sizeof(u64));
        u64* oc = (u64*)ctx->host_alloc((keep ? keep : 1) * sizeof(u64));
        if (!orp || !od || !oc) {
            ctx->host_free(orp); ctx->host_free(od); ctx->host_free(oc);
            set_error("fgpu_expand_trail_counts: host allocation failed");
            return FGPU_OOM;
        }
        u64 o = 0;
        for (u64 r = 0; r < nsrc; ++r) {
            orp[r] = o;
            for (u32 k = rp[r]; k < rp[r + 1]; ++k)
                if (cv[k]) { od[o] = ci[k]; oc[o] = cv[k]; ++o; }
        }
        orp[nsrc] = o;
        *out_rowptr = orp; *out_dest = od; *out_count = oc; *out_nnz = o;
        return FGPU_OK;
    };
    i = run();
    if (f0) mat_release(f0);
    if (c1) mat_release(c1);
    if (c2) mat_release(c2);
    cleanup();
    return i;
}

// Public producers of snapshots: the implementation above, then fgpu_ctx::publish().
extern "C" {

fgpu_info fgpu_mxm(fgpu_ctx* ctx, fgpu_mat** c, const fgpu_mat* f, const fgpu_mat* b) {
    fgpu_info i_ = mxm_impl(ctx, c, f, b);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

fgpu_info fgpu_delta_lmxm(fgpu_ctx* ctx, fgpu_mat** c, const fgpu_mat* f, const fgpu_mat* m, const fgpu_mat* dp,
                          const fgpu_mat* dm) {
    fgpu_info i_ = delta_lmxm_impl(ctx, c, f, m, dp, dm);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

}  // extern "C"
