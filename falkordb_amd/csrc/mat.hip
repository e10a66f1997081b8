// mat.hip — device-resident matrix snapshots: builders, export, transpose, probes,
// Delta merge, pattern intersection, slabs, and the on-device R-MAT generator.
//
// Reference surface mirrored here (graph/src/graph/graphblas/matrix.rs unless noted):
//   new :1214-1235 | build :1281-1303 (bool), :1186-1210 (u64) | nvals :722-729
//   transpose :633-662 | get / isStoredElement :731-737, 1158-1172, 1248-1262
//   Iter :1471-1605 | eWiseAdd / eWiseMult / select / set_pattern / remove_all :824-924
//   GxB_Container export :508-546
//   VersionedMatrix::extract / flush  versioned_matrix.rs:609-620, 892-938
#include <algorithm>
#include <numeric>

#include "common.hpp"

namespace fgpu {

// ---------------------------------------------------------------------------------
// allocation / finalisation
// ---------------------------------------------------------------------------------
std::mutex& bfs_link_mu() {
    static std::mutex mu;
    return mu;
}

// the cached one-call BFS plan (fgpu_bfs, bfs.hip) goes before either of its matrices does
void mat_drop_bfs_plan(const fgpu_mat* a) {
    if (!a->bfs_plan) return;
    fgpu_ctx* pc = a->bfs_plan_ctx;
    (void)fgpu_bfs_plan_free(a->bfs_plan);
    if (a->bfs_plan_at && a->bfs_plan_at->bfs_cached_in == a) a->bfs_plan_at->bfs_cached_in = nullptr;
    if (pc && pc->bfs_cache_owner == a) pc->bfs_cache_owner = nullptr;
    a->bfs_plan = nullptr;
    a->bfs_plan_at = nullptr;
    a->bfs_plan_ctx = nullptr;
}

void mat_release(fgpu_mat* m) {
    if (!m) return;
    {
        // link mutex first, then the matrices' own: the owner of a back-link cannot be released (its own mat_release needs
        // the link mutex) between reading m->bfs_cached_in and locking it
        std::lock_guard<std::mutex> link(bfs_link_mu());
        {
            std::lock_guard<std::mutex> g(m->bfs_mu);
            mat_drop_bfs_plan(m);
        }
        if (const fgpu_mat* owner = m->bfs_cached_in) {   // m is the transpose of a cached plan
            std::lock_guard<std::mutex> g(owner->bfs_mu);
            if (owner->bfs_plan_at == m) mat_drop_bfs_plan(owner);
            m->bfs_cached_in = nullptr;
        }
    }
    fgpu_ctx* c = m->ctx;
    if (c) {
        c->dev_free(m->rowptr);
        c->dev_free(m->colidx);
        c->dev_free(m->vals);
        c->dev_free(m->hrows);
        c->dev_free(m->hub_chunks);
        c->dev_free(m->push_chunks);
        c->dev_free(m->wordrow);
        c->dev_free(m->pull_col);
    }
    tiles_release(m->tiles);
    if (c) c->dev_free(m->bp_items);
    if (c) c->dev_free(m->bp_sitems);
    if (c) c->dev_free(m->bp_split_bits);
    if (m->bp_xplan) bp_xplan_release(c, m->bp_xplan);
    if (m->pr_parts) pr_parts_release(c, m->pr_parts);
    if (m->tcache) mat_release(m->tcache);
    delete m;
}

fgpu_info mat_alloc(fgpu_ctx* ctx, fgpu_mat** out, u64 nrows, u64 ncols, u64 nnz, bool with_vals, u32 nvec_hyper,
                    bool hyper) {
    FGPU_REQUIRE(nrows < 0xFFFFFFFFull && ncols < 0xFFFFFFFFull, FGPU_INVALID,
                 "matrix dims %llu x %llu exceed the 32-bit id space of the device format",
                 (unsigned long long)nrows, (unsigned long long)ncols);
    FGPU_REQUIRE(nnz < 0xFFFFFFFFull, FGPU_INVALID, "nnz %llu exceeds the 32-bit row-pointer space",
                 (unsigned long long)nnz);
    fgpu_mat* m = new (std::nothrow) fgpu_mat();
    FGPU_REQUIRE(m != nullptr, FGPU_OOM, "out of host memory");
    m->ctx = ctx;
    m->nrows = nrows;
    m->ncols = ncols;
    m->nnz = nnz;
    m->nvec = hyper ? nvec_hyper : (u32)nrows;
    fgpu_info i;
    if ((i = ctx->dev_alloc((void**)&m->rowptr, ((size_t)m->nvec + 1) * sizeof(u32))) != FGPU_OK) goto fail;
    if ((i = ctx->dev_alloc((void**)&m->colidx, (size_t)(nnz ? nnz : 1) * sizeof(u32))) != FGPU_OK) goto fail;
    if (with_vals)
        if ((i = ctx->dev_alloc((void**)&m->vals, (size_t)(nnz ? nnz : 1) * sizeof(u64))) != FGPU_OK) goto fail;
    if (hyper)
        if ((i = ctx->dev_alloc((void**)&m->hrows, (size_t)(m->nvec ? m->nvec : 1) * sizeof(u32))) != FGPU_OK)
            goto fail;
    *out = m;
    return FGPU_OK;
fail:
    mat_release(m);
    return i;
}

// max degree + static hub chunk list (rows with >= HUB_DEG entries).
__global__ void hub_scan_kernel(const u32* __restrict__ rowptr, u32 nvec, u32* __restrict__ max_deg,
                                u32* __restrict__ n_chunks, u32* __restrict__ chunks, u32 cap,
                                const u32* __restrict__ hrows, u32 HUB_DEG, u32 HUB_CHUNK) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    u32 deg = 0, b = 0;
    if (r < nvec) { b = rowptr[r]; deg = rowptr[r + 1] - b; }
    u32 m = deg;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { u32 o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    if (lane_id() == 0 && m) atomicMax(max_deg, m);
    if (deg >= HUB_DEG) {
        u32 nch = (deg + HUB_CHUNK - 1) / HUB_CHUNK;
        u32 at = atomicAdd(n_chunks, nch);
        u32 row = hrows ? hrows[r] : r;
        for (u32 k = 0; k < nch; ++k) {
            if (at + k < cap) {
                u32 cb = b + k * HUB_CHUNK;
                u32 ce = cb + HUB_CHUNK < b + deg ? cb + HUB_CHUNK : b + deg;
                chunks[3 * (at + k) + 0] = row;
                chunks[3 * (at + k) + 1] = cb;
                chunks[3 * (at + k) + 2] = ce;
            }
        }
    }
}

static fgpu_info build_hub_list(const fgpu_mat* m, u32 deg_min, u32 chunk, u32** out, u32* n_out) {
    fgpu_ctx* ctx = m->ctx;
    // every hub chunk holds >= 1 edge and at most nnz / chunk + (#hub rows) chunks exist
    u32 cap = (u32)(m->nnz / chunk + m->nnz / deg_min + 1);
    DevBuf<u32> meta, chunks;
    FGPU_TRY(meta.alloc(ctx, 2));
    FGPU_TRY(chunks.alloc(ctx, (size_t)cap * 3));
    FGPU_HIP(hipMemsetAsync(meta.p, 0, 2 * sizeof(u32), ctx->stream()));
    hipLaunchKernelGGL(hub_scan_kernel, dim3(cdiv(m->nvec, 256)), dim3(256), 0, ctx->stream(), m->rowptr, m->nvec,
                       meta.p, meta.p + 1, chunks.p, cap, m->hrows, deg_min, chunk);
    FGPU_HIP(hipGetLastError());
    u32 h[2];
    FGPU_HIP(hipMemcpyAsync(ctx->pinned(), meta.p, 2 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream()));
    FGPU_HIP(hipStreamSynchronize(ctx->stream()));
    memcpy(h, ctx->pinned(), sizeof(h));
    m->max_deg = h[0];
    *n_out = h[1] < cap ? h[1] : cap;
    if (*n_out) {
        FGPU_TRY(ctx->dev_alloc((void**)out, (size_t)*n_out * 3 * sizeof(u32)));
        FGPU_HIP(hipMemcpyAsync(*out, chunks.p, (size_t)*n_out * 3 * sizeof(u32), hipMemcpyDeviceToDevice,
                                ctx->stream()));
    }
    return FGPU_OK;
}

// Called on a matrix no other thread can see yet (its builder) or under m->idx_mu (mat_ensure_finalized).
fgpu_info mat_finalize(const fgpu_mat* m) {
    m->max_deg = 0;
    m->n_hub_chunks = 0;
    m->n_push_chunks = 0;
    if (m->nnz && m->nvec) {
        FGPU_TRY(build_hub_list(m, HUB_DEG, HUB_CHUNK, &m->hub_chunks, &m->n_hub_chunks));
        if (m->max_deg >= PUSH_HUB_DEG)
            FGPU_TRY(build_hub_list(m, PUSH_HUB_DEG, PUSH_HUB_CHUNK, &m->push_chunks, &m->n_push_chunks));
        FGPU_HIP(hipStreamSynchronize(m->ctx->stream()));   // the chunk copies are complete before the flag is raised
    }
    m->finalized.store(true, std::memory_order_release);
    return FGPU_OK;
}

fgpu_info mat_ensure_finalized(const fgpu_mat* m) {
    if (m->finalized.load(std::memory_order_acquire)) return FGPU_OK;
    std::lock_guard<std::mutex> idx_guard(m->idx_mu);
    if (m->finalized.load(std::memory_order_acquire)) return FGPU_OK;
    return mat_finalize(m);
}

// ---------------------------------------------------------------------------------
// device COO -> CSR
// ---------------------------------------------------------------------------------
constexpr u32 ROW_INVALID = 0xFFFFFFFFu;

__global__ void coo_hist_kernel(const u32* __restrict__ rows, u64 n, u32* __restrict__ hist) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u32 r = rows[i];
        if (r != ROW_INVALID) atomicAdd(&hist[r], 1u);
    }
}
__global__ void coo_scatter_kernel(const u32* __restrict__ rows, const u32* __restrict__ cols, u64 n,
                                   const u64* __restrict__ off, u32* __restrict__ cursor, u32* __restrict__ tmp) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u32 r = rows[i];
        if (r != ROW_INVALID) {
            u32 k = atomicAdd(&cursor[r], 1u);
            tmp[off[r] + k] = cols[i];
        }
    }
}

fgpu_info mat_from_device_coo(fgpu_ctx* ctx, fgpu_mat** out, u64 nrows, u64 ncols, const u32* rows, const u32* cols,
                              u64 n) {
    if (ctx->opt.transpose_mode != 1) {   // two stable counting sorts, no atomics, no per-row sort (transpose.hip)
        fgpu_info ci = mat_from_device_coo_counting(ctx, out, nrows, ncols, rows, cols, n);
        if (ci != FGPU_NO_VALUE) return ci;
    }
    DevBuf<u32> hist, cnt, tmp, rowptr;
    DevBuf<u64> off, tot;
    FGPU_TRY(hist.alloc(ctx, nrows + 1));
    FGPU_TRY(off.alloc(ctx, nrows + 1));
    FGPU_TRY(tot.alloc(ctx, 1));
    FGPU_HIP(hipMemsetAsync(hist.p, 0, (nrows + 1) * sizeof(u32), ctx->stream()));
    u32 grid = ctx->cus * 16;
    if (n) {
        hipLaunchKernelGGL(coo_hist_kernel, dim3(grid), dim3(256), 0, ctx->stream(), rows, n, hist.p);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(scan_u32_to_u64(ctx, hist.p, off.p, nrows + 1, tot.p));
    u64 nvalid = 0;
    FGPU_TRY(read_u64(ctx, tot.p, &nvalid));
    FGPU_TRY(tmp.alloc(ctx, nvalid));
    FGPU_TRY(cnt.alloc(ctx, nrows + 1));
    FGPU_HIP(hipMemsetAsync(hist.p, 0, (nrows + 1) * sizeof(u32), ctx->stream()));
    FGPU_HIP(hipMemsetAsync(cnt.p, 0, (nrows + 1) * sizeof(u32), ctx->stream()));
    if (n) {
        hipLaunchKernelGGL(coo_scatter_kernel, dim3(grid), dim3(256), 0, ctx->stream(), rows, cols, n, off.p, hist.p,
                           tmp.p);
        FGPU_HIP(hipGetLastError());
    }
    hist.release();
    FGPU_TRY(segsort_unique(ctx, tmp.p, off.p, (u32)nrows, (u32)ncols, cnt.p, nullptr));
    FGPU_TRY(rowptr.alloc(ctx, nrows + 1));
    FGPU_TRY(scan_u32(ctx, cnt.p, rowptr.p, nrows + 1, nullptr));  // cnt[nrows] == 0
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, rowptr.p + nrows, &nnz));
    fgpu_mat* m = nullptr;
    FGPU_TRY(mat_alloc(ctx, &m, nrows, ncols, nnz, false, 0, false));
    FGPU_HIP(hipMemcpyAsync(m->rowptr, rowptr.p, (nrows + 1) * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream()));
    fgpu_info i = compact_segments(ctx, tmp.p, off.p, m->rowptr, (u32)nrows, m->colidx);
    if (i == FGPU_OK) i = mat_finalize(m);
    if (i != FGPU_OK) { mat_release(m); return i; }
    *out = m;
    return FGPU_OK;
}

// ---------------------------------------------------------------------------------
// host CSR upload (shared by from_csr / from_coo host path)
// ---------------------------------------------------------------------------------
static fgpu_info upload_host_csr(fgpu_ctx* ctx, fgpu_mat** out, u64 nrows, u64 ncols,
                                 const std::vector<u32>& hrows_or_empty, bool hyper, const std::vector<u32>& rowptr,
                                 const std::vector<u32>& colidx, const std::vector<u64>* vals) {
    u64 nnz = colidx.size();
    fgpu_mat* m = nullptr;
    FGPU_TRY(mat_alloc(ctx, &m, nrows, ncols, nnz, vals != nullptr, (u32)hrows_or_empty.size(), hyper));
    // (h2d returns once the host vectors have been consumed: they die with the caller)
    fgpu_info u = ctx->h2d(m->rowptr, rowptr.data(), rowptr.size() * sizeof(u32));
    if (u == FGPU_OK && nnz) u = ctx->h2d(m->colidx, colidx.data(), nnz * sizeof(u32));
    if (u == FGPU_OK && vals && nnz) u = ctx->h2d(m->vals, vals->data(), nnz * sizeof(u64));
    if (u == FGPU_OK && hyper && !hrows_or_empty.empty())
        u = ctx->h2d(m->hrows, hrows_or_empty.data(), hrows_or_empty.size() * sizeof(u32));
    if (u != FGPU_OK) {
        mat_release(m);
        return u;
    }
    fgpu_info i = mat_finalize(m);
    if (i != FGPU_OK) { mat_release(m); return i; }
    *out = m;
    return FGPU_OK;
}

// Decide the storage form of a host-built matrix: hypersparse when few rows are populated
// (delta layers are pinned hypersparse in the reference, versioned_matrix.rs Delta::new).
static bool prefer_hyper(u64 nrows, u64 nvec) { return nrows > 4096 && nvec * 16 <= nrows; }

// ---------------------------------------------------------------------------------
// probes, fills, merges (kernels)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ bool row_contains(const u32* __restrict__ col, u32 b, u32 e, u32 key, u32* pos) {
    u32 lo = b, hi = e;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if (col[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (pos) *pos = lo;
    return lo < e && col[lo] == key;
}

__global__ void probe_kernel(CsrView a, const u64* __restrict__ avals, const u64* __restrict__ rows,
                             const u64* __restrict__ cols, u64 n, u64 nrows, u64 ncols,
                             uint8_t* __restrict__ present, u64* __restrict__ vals) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 r = rows[i], c = cols[i];
    uint8_t p = 0;
    u64 v = 0;
    if (r < nrows && c < ncols) {
        u32 b, e, pos;
        row_range(a, (u32)r, b, e);
        if (row_contains(a.colidx, b, e, (u32)c, &pos)) {
            p = 1;
            if (avals) v = avals[pos];
        }
    }
    present[i] = p;
    if (vals) vals[i] = v;
}

// upper bound of merged row lengths: ub[r] = len_m(r) (+ len_dp(r) added by the scatter kernel)
__global__ void row_len_kernel(CsrView m, u32 nrows, u32* __restrict__ ub) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    u32 len = 0;
    if (r < nrows) {
        u32 b, e;
        row_range(m, r, b, e);
        len = e - b;
    }
    ub[r] = len;
}
__global__ void add_stored_row_len_kernel(CsrView d, u32* __restrict__ ub, uint8_t* __restrict__ dirty) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.nvec) return;
    u32 len = d.rowptr[i + 1] - d.rowptr[i];
    if (len == 0) return;
    u32 r = d.hrows ? d.hrows[i] : i;
    ub[r] += len;  // stored rows are unique: no race
    if (dirty) dirty[r] = 1;
}

// one wavefront per row: tmp[off[r]..] = (m_r \ dm_r) ++ (dp_r [\ dm_r])
__global__ __launch_bounds__(256) void merge_fill_kernel(CsrView m, CsrView dp, CsrView dm, bool has_dp, bool has_dm,
                                                        bool dm_masks_dp, u32 nrows, const u64* __restrict__ off,
                                                        u32* __restrict__ tmp, u32* __restrict__ cnt) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 r = wave; r < nrows; r += nwaves) {
        u32 mb, me, db = 0, de = 0, pb = 0, pe = 0;
        row_range(m, r, mb, me);
        if (has_dm) row_range(dm, r, db, de);
        if (has_dp) row_range(dp, r, pb, pe);
        const u64 o = off[r];
        u32 outn = 0;
        if (db == de) {
            for (u32 i = mb + lane; i < me; i += 64) tmp[o + (i - mb)] = m.colidx[i];
            outn = me - mb;
        } else {
            for (u32 i0 = mb; i0 < me; i0 += 64) {
                u32 i = i0 + lane;
                u32 x = 0;
                bool keep = false;
                if (i < me) {
                    x = m.colidx[i];
                    keep = !row_contains(dm.colidx, db, de, x, nullptr);
                }
                u64 mask = __ballot(keep);
                if (keep) tmp[o + outn + __popcll(mask & ((1ull << lane) - 1ull))] = x;
                outn += (u32)__popcll(mask);
            }
        }
        if (pb != pe) {
            for (u32 i0 = pb; i0 < pe; i0 += 64) {
                u32 i = i0 + lane;
                u32 x = 0;
                bool keep = false;
                if (i < pe) {
                    x = dp.colidx[i];
                    keep = !(dm_masks_dp && db != de && row_contains(dm.colidx, db, de, x, nullptr));
                }
                u64 mask = __ballot(keep);
                if (keep) tmp[o + outn + __popcll(mask & ((1ull << lane) - 1ull))] = x;
                outn += (u32)__popcll(mask);
            }
        }
        if (lane == 0) cnt[r] = outn;  // for clean rows this is final; dirty rows get re-counted by the sort
    }
}

// segment lengths after fill may be shorter than the reserved span; the sort needs exact
// segment bounds, so dirty rows are sorted through a (begin, begin+cnt) offset pair.
__global__ void tight_off_kernel(const u64* __restrict__ off, const u32* __restrict__ cnt,
                                 const uint8_t* __restrict__ dirty, u32 nrows, u64* __restrict__ off2,
                                 u32* __restrict__ cnt2, uint8_t* __restrict__ dirty2) {
    // segment 2r = row r's filled extent, segment 2r+1 = the unused tail of its reservation
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    off2[2 * (u64)r] = off[r];
    off2[2 * (u64)r + 1] = off[r] + cnt[r];
    cnt2[2 * (u64)r] = cnt[r];
    cnt2[2 * (u64)r + 1] = 0;
    dirty2[2 * (u64)r] = dirty[r];
    dirty2[2 * (u64)r + 1] = 0;
}
__global__ void take_even_kernel(const u32* __restrict__ cnt2, u32 nrows, u32* __restrict__ cnt) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrows) cnt[r] = cnt2[2 * (u64)r];
}

// a ∩ b per row: keep a's entries found in b (values come from b when present)
__global__ __launch_bounds__(256) void intersect_fill_kernel(CsrView a, CsrView b, const u64* __restrict__ bvals,
                                                            u32 nrows, u32* __restrict__ tmp,
                                                            u64* __restrict__ tmpv, u32* __restrict__ cnt,
                                                            const u32* __restrict__ aoff) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 r = wave; r < nrows; r += nwaves) {
        u32 ab, ae, bb, be;
        row_range(a, r, ab, ae);
        row_range(b, r, bb, be);
        u32 outn = 0;
        const u32 o = aoff[r];
        if (bb != be) {
            for (u32 i0 = ab; i0 < ae; i0 += 64) {
                u32 i = i0 + lane;
                u32 x = 0, pos = 0;
                bool keep = false;
                if (i < ae) {
                    x = a.colidx[i];
                    keep = row_contains(b.colidx, bb, be, x, &pos);
                }
                u64 mask = __ballot(keep);
                if (keep) {
                    u32 w = o + outn + __popcll(mask & ((1ull << lane) - 1ull));
                    tmp[w] = x;
                    if (tmpv) tmpv[w] = bvals[pos];
                }
                outn += (u32)__popcll(mask);
            }
        }
        if (lane == 0) cnt[r] = outn;
    }
}

__global__ void compact32_kernel(const u32* __restrict__ tmp, const u64* __restrict__ tmpv,
                                 const u32* __restrict__ srcoff, const u32* __restrict__ rowptr, u32 nrows,
                                 u32* __restrict__ col, u64* __restrict__ val) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 r = wave; r < nrows; r += nwaves) {
        u32 s = srcoff[r], o = rowptr[r], c = rowptr[r + 1] - o;
        for (u32 i = lane; i < c; i += 64) {
            col[o + i] = tmp[s + i];
            if (val) val[o + i] = tmpv[s + i];
        }
    }
}

__global__ void row_degree_kernel(const u32* __restrict__ rowptr, u32 nrows, u32* __restrict__ deg) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrows) deg[r] = rowptr[r + 1] - rowptr[r];
}

// dense rowptr for a (possibly hypersparse) view
__global__ void dense_rowptr_len_kernel(CsrView a, u32* __restrict__ len) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nvec) return;
    u32 r = a.hrows ? a.hrows[i] : i;
    len[r] = a.rowptr[i + 1] - a.rowptr[i];
}

// slab extraction: count / fill entries of each row with lo <= col < hi
__global__ __launch_bounds__(256) void col_slab_count_kernel(CsrView a, u32 lo, u32 hi, u32* __restrict__ cnt) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.nrows) return;
    u32 b = a.rowptr[r], e = a.rowptr[r + 1], p0, p1;
    row_contains(a.colidx, b, e, lo, &p0);
    row_contains(a.colidx, b, e, hi, &p1);
    cnt[r] = p1 - p0;
}
__global__ __launch_bounds__(256) void col_slab_fill_kernel(CsrView a, u32 lo, u32 hi, const u32* __restrict__ rowptr,
                                                           u32* __restrict__ col) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 r = wave; r < a.nrows; r += nwaves) {
        u32 b = a.rowptr[r], e = a.rowptr[r + 1], p0;
        u32 o = rowptr[r], c = rowptr[r + 1] - o;
        if (c == 0) continue;
        row_contains(a.colidx, b, e, lo, &p0);
        for (u32 i = lane; i < c; i += 64) col[o + i] = a.colidx[p0 + i];
    }
}
__global__ void row_slab_count_kernel(CsrView a, u32 lo, u32 hi, u32* __restrict__ cnt) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > a.nrows) return;
    cnt[r] = (r < a.nrows && r >= lo && r < hi) ? a.rowptr[r + 1] - a.rowptr[r] : 0u;
}

// COO expansion of a CSR (for transpose): rows_out[i] = col, cols_out[i] = row
__global__ __launch_bounds__(256) void csr_to_coo_t_kernel(CsrView a, u32* __restrict__ rows_out,
                                                          u32* __restrict__ cols_out) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 i = wave; i < a.nvec; i += nwaves) {
        u32 r = a.hrows ? a.hrows[i] : i;
        u32 b = a.rowptr[i], e = a.rowptr[i + 1];
        for (u32 k = b + lane; k < e; k += 64) {
            rows_out[k] = a.colidx[k];
            cols_out[k] = r;
        }
    }
}

// uniform sample of the stored entries (bench / test data: "0.1 % random tombstones"):
// entry (r, c) is kept iff mix64(seed ^ mix64(r << 32 | c)) % denom == 0 — a function of the coordinate, so the
// CPU oracle draws the same sample from its own copy of the matrix
__global__ __launch_bounds__(256) void sample_coo_kernel(CsrView a, u64 seed, u32 denom, u32* __restrict__ rows_out,
                                                        u32* __restrict__ cols_out) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 i = wave; i < a.nvec; i += nwaves) {
        const u32 r = a.hrows ? a.hrows[i] : i;
        const u32 b = a.rowptr[i], e = a.rowptr[i + 1];
        for (u32 k = b + lane; k < e; k += 64) {
            const u32 c = a.colidx[k];
            const bool keep = mix64(seed ^ mix64(((u64)r << 32) | c)) % denom == 0;
            rows_out[k] = keep ? r : ROW_INVALID;
            cols_out[k] = c;
        }
    }
}

// ---------------------------------------------------------------------------------
// R-MAT generator
// ---------------------------------------------------------------------------------
__host__ __device__ __forceinline__ u32 rmat_scramble(u32 x, int scale) {
    // bijection on [0, 2^scale): odd multiply, xor-shift, odd multiply, xor-shift (all mod 2^scale)
    const u32 mask = (scale >= 32) ? 0xFFFFFFFFu : ((1u << scale) - 1u);
    const int sh = scale / 2 + 1;
    x = (x * 0x9E3779B1u + 0x7F4A7C15u) & mask;
    x ^= x >> sh;
    x = (x * 0x85EBCA6Bu) & mask;
    x ^= x >> sh;
    x = (x * 0xC2B2AE35u + 0x165667B1u) & mask;
    return x;
}

__global__ void rmat_kernel(int scale, u64 nedges, u64 seed, u32 a32, u32 ab32, u32 abc32, u32* __restrict__ rows,
                            u32* __restrict__ cols) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nedges; i += (u64)gridDim.x * blockDim.x) {
        u32 u = 0, v = 0;
        const u64 base = seed + i * 0x9E3779B97F4A7C15ull;
        for (int l = 0; l < scale; ++l) {
            u32 r = (u32)(mix64(base + (u64)(l + 1) * 0xD1B54A32D192ED03ull) >> 32);
            u32 ub = (r >= ab32) ? 1u : 0u;
            u32 vb = (r >= a32 && r < ab32) || (r >= abc32) ? 1u : 0u;
            u = (u << 1) | ub;
            v = (v << 1) | vb;
        }
        u = rmat_scramble(u, scale);
        v = rmat_scramble(v, scale);
        if (u == v) u = ROW_INVALID;  // self-loops dropped
        rows[i] = u;
        cols[i] = v;
    }
}

}  // namespace fgpu

using namespace fgpu;

// ===================================================================================
// C ABI
// ===================================================================================
namespace fgpu {
// pattern-only transpose on device (values, if any, are ignored)
fgpu_info mat_transpose_pattern(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a) {
    if (a->nnz == 0) {
        if (a->is_hyper()) return fgpu_mat_new(ctx, out, a->ncols, a->nrows);
        // an empty matrix in the dense-row-pointer form stays in that form (an empty column slab of a partitioned
        // BFS — more ranks than populated vertex blocks — still needs a plan over dense row pointers)
        fgpu_mat* o = nullptr;
        FGPU_TRY(mat_alloc(ctx, &o, a->ncols, a->nrows, 0, false, 0, false));
        hipError_t e = hipMemsetAsync(o->rowptr, 0, (a->ncols + 1) * sizeof(u32), ctx->stream());
        if (e != hipSuccess) { mat_release(o); set_error("memset failed: %s", hipGetErrorString(e)); return FGPU_DEVICE; }
        *out = o;
        return FGPU_OK;
    }
    if (ctx->opt.transpose_mode != 1) {   // stable partition by column: rows of the result come out ascending, no sort
        fgpu_info ci = mat_transpose_counting(ctx, out, a);
        if (ci != FGPU_NO_VALUE) return ci;
    }
    DevBuf<u32> rows, cols;
    FGPU_TRY(rows.alloc(ctx, a->nnz));
    FGPU_TRY(cols.alloc(ctx, a->nnz));
    u32 grid = cdiv(a->nvec, 4);
    if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
    hipLaunchKernelGGL(csr_to_coo_t_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(a), rows.p, cols.p);
    FGPU_HIP(hipGetLastError());
    return mat_from_device_coo(ctx, out, a->ncols, a->nrows, rows.p, cols.p, a->nnz);
}
}  // namespace fgpu

extern "C" {

fgpu_info fgpu_mat_free(fgpu_mat* m) {
    // other lanes may still have reads of this snapshot in flight (asynchronous calls of other threads): order the
    // reuse of its blocks, which go to the calling lane's free-list, after everything queued so far
    if (m && m->ctx) m->ctx->fence_lanes();
    mat_release(m);
    return FGPU_OK;
}
fgpu_info fgpu_mat_nrows(const fgpu_mat* m, uint64_t* out) {
    FGPU_REQUIRE(m && out, FGPU_NULL_POINTER, "fgpu_mat_nrows: NULL argument");
    *out = m->nrows;
    return FGPU_OK;
}
fgpu_info fgpu_mat_ncols(const fgpu_mat* m, uint64_t* out) {
    FGPU_REQUIRE(m && out, FGPU_NULL_POINTER, "fgpu_mat_ncols: NULL argument");
    *out = m->ncols;
    return FGPU_OK;
}
fgpu_info fgpu_mat_nvals(const fgpu_mat* m, uint64_t* out) {
    FGPU_REQUIRE(m && out, FGPU_NULL_POINTER, "fgpu_mat_nvals: NULL argument");
    *out = m->nnz;
    return FGPU_OK;
}
fgpu_info fgpu_mat_has_values(const fgpu_mat* m, int32_t* out) {
    FGPU_REQUIRE(m && out, FGPU_NULL_POINTER, "fgpu_mat_has_values: NULL argument");
    *out = m->vals != nullptr;
    return FGPU_OK;
}

static fgpu_info mat_new_impl(fgpu_ctx* ctx, fgpu_mat** out, uint64_t nrows, uint64_t ncols) {
    FGPU_REQUIRE(ctx && out, FGPU_NULL_POINTER, "fgpu_mat_new: NULL argument");
    std::vector<u32> none, rowptr(1, 0), col;
    return upload_host_csr(ctx, out, nrows, ncols, none, true, rowptr, col, nullptr);
}

static fgpu_info mat_from_coo_impl(fgpu_ctx* ctx, fgpu_mat** out, uint64_t nrows, uint64_t ncols, const uint64_t* rows,
                            const uint64_t* cols, const uint64_t* vals, uint64_t n) {
    FGPU_REQUIRE(ctx && out, FGPU_NULL_POINTER, "fgpu_mat_from_coo: NULL argument");
    FGPU_REQUIRE(n == 0 || (rows && cols), FGPU_NULL_POINTER, "fgpu_mat_from_coo: NULL tuple arrays");
    FGPU_REQUIRE(nrows < 0xFFFFFFFFull && ncols < 0xFFFFFFFFull, FGPU_INVALID,
                 "fgpu_mat_from_coo: dims exceed the 32-bit id space");
    for (u64 i = 0; i < n; ++i)
        FGPU_REQUIRE(rows[i] < nrows && cols[i] < ncols, FGPU_OUT_OF_BOUNDS,
                     "fgpu_mat_from_coo: tuple %llu = (%llu, %llu) outside %llu x %llu", (unsigned long long)i,
                     (unsigned long long)rows[i], (unsigned long long)cols[i], (unsigned long long)nrows,
                     (unsigned long long)ncols);
    // Builds of >= DEVICE_BUILD_MIN tuples (valued or not) are sorted / deduplicated on the device; only a
    // short host-provided tuple list is ordered on the host while it is being marshalled for upload
    // (a kernel chain over nrows-sized histograms would cost more than the whole upload).
    const u64 DEVICE_BUILD_MIN = 4096;
    if (n >= DEVICE_BUILD_MIN) {
        std::vector<u32> r32(n), c32(n);
        for (u64 i = 0; i < n; ++i) { r32[i] = (u32)rows[i]; c32[i] = (u32)cols[i]; }
        DevBuf<u32> dr, dc;
        DevBuf<u64> dv;
        FGPU_TRY(dr.alloc(ctx, n));
        FGPU_TRY(dc.alloc(ctx, n));
        FGPU_TRY(ctx->h2d(dr.p, r32.data(), n * sizeof(u32)));
        FGPU_TRY(ctx->h2d(dc.p, c32.data(), n * sizeof(u32)));
        if (vals) {
            FGPU_TRY(dv.alloc(ctx, n));
            FGPU_TRY(ctx->h2d(dv.p, vals, n * sizeof(u64)));
        }
        FGPU_HIP(hipStreamSynchronize(ctx->stream()));
        if (vals) return mat_from_device_coo_vals(ctx, out, nrows, ncols, dr.p, dc.p, dv.p, n);
        return mat_from_device_coo(ctx, out, nrows, ncols, dr.p, dc.p, n);
    }
    // host path: stable sort by (row, col); duplicates collapse, last value wins
    std::vector<u64> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](u64 a, u64 b) {
        if (rows[a] != rows[b]) return rows[a] < rows[b];
        return cols[a] < cols[b];
    });
    std::vector<u32> hrows, rowptr, col;
    std::vector<u64> val;
    std::vector<u32> rowcount;  // per stored row
    u64 prev_r = ~0ull, prev_c = ~0ull;
    for (u64 k = 0; k < n; ++k) {
        u64 i = order[k];
        if (rows[i] == prev_r && cols[i] == prev_c) {
            if (vals) val.back() = vals[i];
            continue;
        }
        if (rows[i] != prev_r) { hrows.push_back((u32)rows[i]); rowcount.push_back(0); }
        rowcount.back()++;
        col.push_back((u32)cols[i]);
        if (vals) val.push_back(vals[i]);
        prev_r = rows[i];
        prev_c = cols[i];
    }
    bool hyper = prefer_hyper(nrows, hrows.size());
    if (hyper) {
        rowptr.assign(hrows.size() + 1, 0);
        for (size_t i = 0; i < hrows.size(); ++i) rowptr[i + 1] = rowptr[i] + rowcount[i];
        return upload_host_csr(ctx, out, nrows, ncols, hrows, true, rowptr, col, vals ? &val : nullptr);
    }
    rowptr.assign(nrows + 1, 0);
    for (size_t i = 0; i < hrows.size(); ++i) rowptr[hrows[i] + 1] = rowcount[i];
    for (u64 r = 0; r < nrows; ++r) rowptr[r + 1] += rowptr[r];
    std::vector<u32> none;
    return upload_host_csr(ctx, out, nrows, ncols, none, false, rowptr, col, vals ? &val : nullptr);
}

static fgpu_info mat_from_csr_impl(fgpu_ctx* ctx, fgpu_mat** out, uint64_t nrows, uint64_t ncols, uint64_t nnz,
                            const void* rowptr, int rowptr_bits, const void* colidx, int colidx_bits,
                            const uint64_t* vals, const uint64_t* hyper_rows, uint64_t nvec) {
    FGPU_REQUIRE(ctx && out && rowptr, FGPU_NULL_POINTER, "fgpu_mat_from_csr: NULL argument");
    FGPU_REQUIRE(nnz == 0 || colidx, FGPU_NULL_POINTER, "fgpu_mat_from_csr: NULL colidx");
    FGPU_REQUIRE((rowptr_bits == 32 || rowptr_bits == 64) && (colidx_bits == 32 || colidx_bits == 64),
                 FGPU_INVALID, "fgpu_mat_from_csr: index widths must be 32 or 64");
    FGPU_REQUIRE(nrows < 0xFFFFFFFFull && ncols < 0xFFFFFFFFull && nnz < 0xFFFFFFFFull, FGPU_INVALID,
                 "fgpu_mat_from_csr: dims/nnz exceed the 32-bit device format");
    const bool hyper = hyper_rows != nullptr;
    const u64 nv = hyper ? nvec : nrows;
    std::vector<u32> rp(nv + 1), ci(nnz), hr;
    for (u64 i = 0; i <= nv; ++i) {
        u64 v = rowptr_bits == 32 ? ((const u32*)rowptr)[i] : ((const u64*)rowptr)[i];
        FGPU_REQUIRE(v <= nnz && (i == 0 || v >= rp[i - 1]), FGPU_INVALID,
                     "fgpu_mat_from_csr: rowptr[%llu] = %llu is not monotone within nnz", (unsigned long long)i,
                     (unsigned long long)v);
        rp[i] = (u32)v;
    }
    FGPU_REQUIRE(rp[0] == 0 && rp[nv] == nnz, FGPU_INVALID, "fgpu_mat_from_csr: rowptr does not span [0, nnz]");
    for (u64 i = 0; i < nnz; ++i) {
        u64 v = colidx_bits == 32 ? ((const u32*)colidx)[i] : ((const u64*)colidx)[i];
        FGPU_REQUIRE(v < ncols, FGPU_OUT_OF_BOUNDS, "fgpu_mat_from_csr: colidx[%llu] = %llu >= ncols",
                     (unsigned long long)i, (unsigned long long)v);
        ci[i] = (u32)v;
    }
    for (u64 r = 0; r < nv; ++r)
        for (u32 k = rp[r] + 1; k < rp[r + 1]; ++k)
            FGPU_REQUIRE(ci[k - 1] < ci[k], FGPU_INVALID,
                         "fgpu_mat_from_csr: row %llu is not sorted/unique (pass the wait()ed state)",
                         (unsigned long long)r);
    if (hyper) {
        hr.resize(nvec);
        for (u64 i = 0; i < nvec; ++i) {
            FGPU_REQUIRE(hyper_rows[i] < nrows && (i == 0 || hyper_rows[i] > hyper_rows[i - 1]), FGPU_INVALID,
                         "fgpu_mat_from_csr: hyper row list must be ascending and < nrows");
            hr[i] = (u32)hyper_rows[i];
        }
    }
    std::vector<u64> vv;
    if (vals) vv.assign(vals, vals + nnz);
    return upload_host_csr(ctx, out, nrows, ncols, hr, hyper, rp, ci, vals ? &vv : nullptr);
}

static fgpu_info mat_rmat_impl(fgpu_ctx* ctx, fgpu_mat** out, int scale, int edge_factor, uint64_t seed, uint32_t a16,
                        uint32_t b16, uint32_t c16) {
    FGPU_REQUIRE(ctx && out, FGPU_NULL_POINTER, "fgpu_mat_rmat: NULL argument");
    FGPU_REQUIRE(scale >= 1 && scale <= 30 && edge_factor >= 1, FGPU_INVALID, "fgpu_mat_rmat: scale in [1,30]");
    if (a16 == 0 && b16 == 0 && c16 == 0) { a16 = 37356; b16 = 12452; c16 = 12452; }  // .57 .19 .19 in 16.16
    FGPU_REQUIRE((u64)a16 + b16 + c16 <= 65536, FGPU_INVALID, "fgpu_mat_rmat: a+b+c must be <= 1");
    const u64 n = 1ull << scale;
    const u64 nedges = (u64)edge_factor << scale;
    FGPU_REQUIRE(nedges < 0xFFFFFFFFull, FGPU_INVALID, "fgpu_mat_rmat: too many edges for the 32-bit format");
    u64 a = (u64)a16 << 16, ab = ((u64)a16 + b16) << 16, abc = ((u64)a16 + b16 + c16) << 16;
    u32 a32 = (u32)(a > 0xFFFFFFFFull ? 0xFFFFFFFFull : a);
    u32 ab32 = (u32)(ab > 0xFFFFFFFFull ? 0xFFFFFFFFull : ab);
    u32 abc32 = (u32)(abc > 0xFFFFFFFFull ? 0xFFFFFFFFull : abc);
    DevBuf<u32> rows, cols;
    FGPU_TRY(rows.alloc(ctx, nedges));
    FGPU_TRY(cols.alloc(ctx, nedges));
    hipLaunchKernelGGL(rmat_kernel, dim3(ctx->cus * 16), dim3(256), 0, ctx->stream(), scale, nedges, seed, a32, ab32,
                       abc32, rows.p, cols.p);
    FGPU_HIP(hipGetLastError());
    return mat_from_device_coo(ctx, out, n, n, rows.p, cols.p, nedges);
}

fgpu_info fgpu_mat_sample(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t seed, uint32_t denom) {
    FGPU_REQUIRE(ctx && out && a, FGPU_NULL_POINTER, "fgpu_mat_sample: NULL argument");
    FGPU_REQUIRE(denom >= 1, FGPU_INVALID, "fgpu_mat_sample: denom must be >= 1");
    if (a->nnz == 0) return fgpu_mat_new(ctx, out, a->nrows, a->ncols);
    DevBuf<u32> rows, cols;
    FGPU_TRY(rows.alloc(ctx, a->nnz));
    FGPU_TRY(cols.alloc(ctx, a->nnz));
    u32 grid = cdiv(a->nvec, 4);
    if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
    hipLaunchKernelGGL(sample_coo_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(a), seed, denom, rows.p,
                       cols.p);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(mat_from_device_coo(ctx, out, a->nrows, a->ncols, rows.p, cols.p, a->nnz));
    return ctx->publish();
}

// ---- export -------------------------------------------------------------------------
static fgpu_info download_mat(fgpu_ctx* ctx, const fgpu_mat* m, std::vector<u32>& rp, std::vector<u32>& ci,
                              std::vector<u64>& vv, std::vector<u32>& hr) {
    rp.resize((size_t)m->nvec + 1);
    ci.resize(m->nnz);
    FGPU_TRY(ctx->d2h(rp.data(), m->rowptr, rp.size() * sizeof(u32)));
    if (m->nnz)
        FGPU_TRY(ctx->d2h(ci.data(), m->colidx, m->nnz * sizeof(u32)));
    if (m->vals && m->nnz) {
        vv.resize(m->nnz);
        FGPU_TRY(ctx->d2h(vv.data(), m->vals, m->nnz * sizeof(u64)));
    }
    if (m->hrows && m->nvec) {
        hr.resize(m->nvec);
        FGPU_TRY(ctx->d2h(hr.data(), m->hrows, (size_t)m->nvec * sizeof(u32)));
    }
    FGPU_HIP(hipStreamSynchronize(ctx->stream()));
    return FGPU_OK;
}

fgpu_info fgpu_mat_export_csr(fgpu_ctx* ctx, const fgpu_mat* m, uint64_t** rowptr, uint64_t** colidx,
                              uint64_t** vals, uint64_t* nnz) {
    FGPU_REQUIRE(ctx && m && rowptr && colidx && nnz, FGPU_NULL_POINTER, "fgpu_mat_export_csr: NULL argument");
    // (large arrays come pinned from the context's pool and are filled by one DMA each: ctx.hip result_alloc / d2h_widen)
    u64* orp = (u64*)ctx->result_alloc((m->nrows + 1) * sizeof(u64));
    u64* oci = (u64*)ctx->result_alloc((m->nnz ? m->nnz : 1) * sizeof(u64));
    u64* ov = (vals && m->vals) ? (u64*)ctx->result_alloc((m->nnz ? m->nnz : 1) * sizeof(u64)) : nullptr;
    if (!orp || !oci || (vals && m->vals && !ov)) {
        ctx->host_free(orp); ctx->host_free(oci); ctx->host_free(ov);
        set_error("fgpu_mat_export_csr: host allocation failed");
        return FGPU_OOM;
    }
    // ids are 32-bit on the device and 64-bit for the caller (GrB_Index): widened chunk by chunk on the way out of the
    // pinned staging halves, straight into the caller's arrays (no intermediate host copy)
    fgpu_info i = FGPU_OK;
    if (m->is_hyper()) {
        std::vector<u32> rp((size_t)m->nvec + 1), hr(m->nvec);
        i = ctx->d2h(rp.data(), m->rowptr, rp.size() * sizeof(u32));
        if (i == FGPU_OK && m->nvec) i = ctx->d2h(hr.data(), m->hrows, (size_t)m->nvec * sizeof(u32));
        if (i == FGPU_OK) {
            memset(orp, 0, (m->nrows + 1) * sizeof(u64));
            for (u32 k = 0; k < m->nvec; ++k) orp[hr[k] + 1] = rp[k + 1] - rp[k];
            for (u64 r = 0; r < m->nrows; ++r) orp[r + 1] += orp[r];
        }
    } else {
        i = ctx->d2h_widen(orp, m->rowptr, m->nrows + 1);
    }
    if (i == FGPU_OK && m->nnz) i = ctx->d2h_widen(oci, m->colidx, m->nnz);
    if (i == FGPU_OK && ov && m->nnz) i = ctx->d2h(ov, m->vals, m->nnz * sizeof(u64));
    if (i != FGPU_OK) {
        ctx->host_free(orp); ctx->host_free(oci); ctx->host_free(ov);
        return i;
    }
    *rowptr = orp;
    *colidx = oci;
    if (vals) *vals = ov;
    *nnz = m->nnz;
    return FGPU_OK;
}

fgpu_info fgpu_mat_extract(fgpu_ctx* ctx, const fgpu_mat* m, uint64_t min_row, uint64_t max_row, uint64_t** rows,
                           uint64_t** cols, uint64_t** vals, uint64_t* n) {
    FGPU_REQUIRE(ctx && m && rows && cols && n, FGPU_NULL_POINTER, "fgpu_mat_extract: NULL argument");
    *rows = *cols = nullptr;
    if (vals) *vals = nullptr;
    *n = 0;
    if (m->nvec == 0 || m->nnz == 0 || min_row > max_row || min_row >= m->nrows) return FGPU_OK;
    if (max_row >= m->nrows) max_row = m->nrows - 1;
    // stored-row window [i0, i1)
    std::vector<u32> hr;
    u32 i0, i1;
    if (m->is_hyper()) {
        hr.resize(m->nvec);
        FGPU_TRY(ctx->d2h(hr.data(), m->hrows, (size_t)m->nvec * sizeof(u32)));
        FGPU_HIP(hipStreamSynchronize(ctx->stream()));
        i0 = (u32)(std::lower_bound(hr.begin(), hr.end(), (u32)min_row) - hr.begin());
        i1 = (u32)(std::upper_bound(hr.begin(), hr.end(), (u32)max_row) - hr.begin());
    } else {
        i0 = (u32)min_row;
        i1 = (u32)max_row + 1;
    }
    if (i0 >= i1) return FGPU_OK;
    std::vector<u32> rp(i1 - i0 + 1);
    FGPU_TRY(ctx->d2h(rp.data(), m->rowptr + i0, rp.size() * sizeof(u32)));
    FGPU_HIP(hipStreamSynchronize(ctx->stream()));
    u64 b = rp.front(), e = rp.back(), cnt = e - b;
    if (cnt == 0) return FGPU_OK;
    std::vector<u32> ci(cnt);
    std::vector<u64> vv;
    FGPU_TRY(ctx->d2h(ci.data(), m->colidx + b, cnt * sizeof(u32)));
    if (vals && m->vals) {
        vv.resize(cnt);
        FGPU_TRY(ctx->d2h(vv.data(), m->vals + b, cnt * sizeof(u64)));
    }
    FGPU_HIP(hipStreamSynchronize(ctx->stream()));
    u64* orow = (u64*)ctx->host_alloc(cnt * sizeof(u64));
    u64* ocol = (u64*)ctx->host_alloc(cnt * sizeof(u64));
    u64* oval = (vals && m->vals) ? (u64*)ctx->host_alloc(cnt * sizeof(u64)) : nullptr;
    if (!orow || !ocol || (vals && m->vals && !oval)) {
        ctx->host_free(orow); ctx->host_free(ocol); ctx->host_free(oval);
        set_error("fgpu_mat_extract: host allocation failed");
        return FGPU_OOM;
    }
    u64 k = 0;
    for (u32 i = i0; i < i1; ++i) {
        u64 r = m->is_hyper() ? hr[i] : i;
        for (u32 p = rp[i - i0]; p < rp[i - i0 + 1]; ++p, ++k) {
            orow[k] = r;
            ocol[k] = ci[p - b];
        }
    }
    if (oval) memcpy(oval, vv.data(), cnt * sizeof(u64));
    *rows = orow;
    *cols = ocol;
    if (vals) *vals = oval;
    *n = cnt;
    return FGPU_OK;
}

// ---- transpose ---------------------------------------------------------------------
static fgpu_info mat_transpose_impl(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a) {
    FGPU_REQUIRE(ctx && out && a, FGPU_NULL_POINTER, "fgpu_mat_transpose: NULL argument");
    if (a->vals) return mat_transpose_vals(ctx, out, a);
    return mat_transpose_pattern(ctx, out, a);
}

// ---- probes ------------------------------------------------------------------------
fgpu_info fgpu_mat_probe(fgpu_ctx* ctx, const fgpu_mat* m, const uint64_t* rows, const uint64_t* cols, uint64_t n,
                         uint8_t* present, uint64_t* vals) {
    FGPU_REQUIRE(ctx && m && present, FGPU_NULL_POINTER, "fgpu_mat_probe: NULL argument");
    if (n == 0) return FGPU_OK;
    FGPU_REQUIRE(rows && cols, FGPU_NULL_POINTER, "fgpu_mat_probe: NULL coordinate arrays");
    DevBuf<u64> dr, dc, dv;
    DevBuf<uint8_t> dp;
    FGPU_TRY(dr.alloc(ctx, n));
    FGPU_TRY(dc.alloc(ctx, n));
    FGPU_TRY(dp.alloc(ctx, n));
    if (vals) FGPU_TRY(dv.alloc(ctx, n));
    FGPU_TRY(ctx->h2d(dr.p, rows, n * sizeof(u64)));
    FGPU_TRY(ctx->h2d(dc.p, cols, n * sizeof(u64)));
    hipLaunchKernelGGL(probe_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream(), view_of(m),
                       (const u64*)m->vals, (const u64*)dr.p, (const u64*)dc.p, n, m->nrows, m->ncols, dp.p,
                       vals ? dv.p : (u64*)nullptr);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(ctx->d2h(present, dp.p, n));
    if (vals) FGPU_TRY(ctx->d2h(vals, dv.p, n * sizeof(u64)));
    FGPU_HIP(hipStreamSynchronize(ctx->stream()));
    return FGPU_OK;
}

}  // extern "C"

namespace fgpu {

// (m \ dm) U dp on device, pattern only.  Shared by fgpu_mat_merge and delta_lmxm.
fgpu_info mat_merge_device(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm,
                           bool dm_masks_dp) {
    const u64 nrows = m->nrows;
    const bool has_dp = dp && dp->nnz, has_dm = dm && dm->nnz;
    DevBuf<u32> ub, cnt, tmp, rowptr;
    DevBuf<u64> off, tot;
    DevBuf<uint8_t> dirty;
    FGPU_TRY(ub.alloc(ctx, nrows + 1));
    FGPU_TRY(off.alloc(ctx, nrows + 1));
    FGPU_TRY(tot.alloc(ctx, 1));
    FGPU_TRY(dirty.alloc(ctx, nrows + 1));
    FGPU_HIP(hipMemsetAsync(dirty.p, 0, nrows + 1, ctx->stream()));
    hipLaunchKernelGGL(row_len_kernel, dim3(cdiv(nrows + 1, 256)), dim3(256), 0, ctx->stream(), view_of(m),
                       (u32)nrows, ub.p);
    FGPU_HIP(hipGetLastError());
    if (has_dp) {
        hipLaunchKernelGGL(add_stored_row_len_kernel, dim3(cdiv(dp->nvec, 256)), dim3(256), 0, ctx->stream(),
                           view_of(dp), ub.p, dirty.p);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(scan_u32_to_u64(ctx, ub.p, off.p, nrows + 1, tot.p));
    u64 total = 0;
    FGPU_TRY(read_u64(ctx, tot.p, &total));
    FGPU_TRY(tmp.alloc(ctx, total));
    FGPU_TRY(cnt.alloc(ctx, nrows + 1));
    FGPU_HIP(hipMemsetAsync(cnt.p, 0, (nrows + 1) * sizeof(u32), ctx->stream()));
    CsrView vm = view_of(m), vdp = has_dp ? view_of(dp) : vm, vdm = has_dm ? view_of(dm) : vm;
    if (nrows) {
        u32 grid = cdiv(nrows, 4);
        if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
        hipLaunchKernelGGL(merge_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream(), vm, vdp, vdm, has_dp, has_dm,
                           dm_masks_dp, (u32)nrows, (const u64*)off.p, tmp.p, cnt.p);
        FGPU_HIP(hipGetLastError());
    }
    if (has_dp && nrows) {
        // dirty rows (those that received dp entries) need a sort+unique over their exact extent;
        // clean rows keep the count written by the fill.  off2 interleaves (begin, end) so segment
        // 2r is row r and the odd segments (gaps) are skipped through the dirty mask.
        DevBuf<u64> off2;
        DevBuf<u32> cnt2;
        DevBuf<uint8_t> dirty2;
        FGPU_TRY(off2.alloc(ctx, 2 * nrows + 1));
        FGPU_TRY(cnt2.alloc(ctx, 2 * nrows));
        FGPU_TRY(dirty2.alloc(ctx, 2 * nrows));
        hipLaunchKernelGGL(tight_off_kernel, dim3(cdiv(nrows, 256)), dim3(256), 0, ctx->stream(), (const u64*)off.p,
                           (const u32*)cnt.p, (const uint8_t*)dirty.p, (u32)nrows, off2.p, cnt2.p, dirty2.p);
        FGPU_HIP(hipGetLastError());
        FGPU_HIP(hipMemcpyAsync(off2.p + 2 * nrows, off.p + nrows, sizeof(u64), hipMemcpyDeviceToDevice,
                                ctx->stream()));
        FGPU_TRY(segsort_unique(ctx, tmp.p, off2.p, (u32)(2 * nrows), (u32)m->ncols, cnt2.p, dirty2.p));
        hipLaunchKernelGGL(take_even_kernel, dim3(cdiv(nrows, 256)), dim3(256), 0, ctx->stream(), (const u32*)cnt2.p,
                           (u32)nrows, cnt.p);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(rowptr.alloc(ctx, nrows + 1));
    FGPU_TRY(scan_u32(ctx, cnt.p, rowptr.p, nrows + 1, nullptr));
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, rowptr.p + nrows, &nnz));
    fgpu_mat* o = nullptr;
    FGPU_TRY(mat_alloc(ctx, &o, nrows, m->ncols, nnz, false, 0, false));
    FGPU_HIP(hipMemcpyAsync(o->rowptr, rowptr.p, (nrows + 1) * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream()));
    fgpu_info i = compact_segments(ctx, tmp.p, off.p, o->rowptr, (u32)nrows, o->colidx);
    if (i == FGPU_OK) i = mat_finalize(o);
    if (i != FGPU_OK) { mat_release(o); return i; }
    *out = o;
    return FGPU_OK;
}

// dense (nrows+1) device rowptr of a possibly hypersparse matrix
fgpu_info dense_rowptr(fgpu_ctx* ctx, const fgpu_mat* a, DevBuf<u32>& rp) {
    FGPU_TRY(rp.alloc(ctx, a->nrows + 1));
    if (!a->is_hyper()) {
        FGPU_HIP(hipMemcpyAsync(rp.p, a->rowptr, (a->nrows + 1) * sizeof(u32), hipMemcpyDeviceToDevice,
                                ctx->stream()));
        return FGPU_OK;
    }
    FGPU_HIP(hipMemsetAsync(rp.p, 0, (a->nrows + 1) * sizeof(u32), ctx->stream()));
    if (a->nvec) {
        hipLaunchKernelGGL(dense_rowptr_len_kernel, dim3(cdiv(a->nvec, 256)), dim3(256), 0, ctx->stream(), view_of(a),
                           rp.p);
        FGPU_HIP(hipGetLastError());
    }
    return scan_u32(ctx, rp.p, rp.p, a->nrows + 1, nullptr);
}

}  // namespace fgpu

extern "C" {

static fgpu_info mat_merge_impl(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm,
                         int dm_masks_dp) {
    FGPU_REQUIRE(ctx && out && m, FGPU_NULL_POINTER, "fgpu_mat_merge: NULL argument");
    FGPU_REQUIRE(!dp || (dp->nrows == m->nrows && dp->ncols == m->ncols), FGPU_DIM_MISMATCH,
                 "fgpu_mat_merge: dp dims differ from m");
    FGPU_REQUIRE(!dm || (dm->nrows == m->nrows && dm->ncols == m->ncols), FGPU_DIM_MISMATCH,
                 "fgpu_mat_merge: dm dims differ from m");
    if (ctx->opt.merge_mode == 1 && !m->vals && !(dp && dp->vals))
        return mat_merge_device(ctx, out, m, dp, dm, dm_masks_dp != 0);  // one wavefront per row (A/B timing)
    return mat_merge_entries(ctx, out, m, dp, dm, dm_masks_dp != 0, m->nrows, m->ncols, false);
}

static fgpu_info intersect_impl(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, const fgpu_mat* b, uint64_t* nvals) {
    FGPU_REQUIRE(a->nrows == b->nrows && a->ncols == b->ncols, FGPU_DIM_MISMATCH, "intersect: dims differ");
    const u64 nrows = a->nrows;
    DevBuf<u32> arp, cnt, tmp, rowptr;
    DevBuf<u64> tmpv;
    FGPU_TRY(dense_rowptr(ctx, a, arp));
    FGPU_TRY(cnt.alloc(ctx, nrows + 1));
    FGPU_TRY(tmp.alloc(ctx, a->nnz));
    const bool with_vals = (out != nullptr) && b->vals;
    if (with_vals) FGPU_TRY(tmpv.alloc(ctx, a->nnz));
    FGPU_HIP(hipMemsetAsync(cnt.p, 0, (nrows + 1) * sizeof(u32), ctx->stream()));
    if (nrows && a->nnz && b->nnz) {
        u32 grid = cdiv(nrows, 4);
        if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
        hipLaunchKernelGGL(intersect_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(a), view_of(b),
                           (const u64*)b->vals, (u32)nrows, tmp.p, with_vals ? tmpv.p : (u64*)nullptr, cnt.p,
                           (const u32*)arp.p);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(rowptr.alloc(ctx, nrows + 1));
    FGPU_TRY(scan_u32(ctx, cnt.p, rowptr.p, nrows + 1, nullptr));
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, rowptr.p + nrows, &nnz));
    if (nvals) *nvals = nnz;
    if (!out) return FGPU_OK;
    fgpu_mat* o = nullptr;
    FGPU_TRY(mat_alloc(ctx, &o, nrows, a->ncols, nnz, with_vals, 0, false));
    FGPU_HIP(hipMemcpyAsync(o->rowptr, rowptr.p, (nrows + 1) * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream()));
    if (nrows && nnz) {
        u32 grid = cdiv(nrows, 4);
        if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
        hipLaunchKernelGGL(compact32_kernel, dim3(grid), dim3(256), 0, ctx->stream(), (const u32*)tmp.p,
                           (const u64*)(with_vals ? tmpv.p : nullptr), (const u32*)arp.p, (const u32*)o->rowptr,
                           (u32)nrows, o->colidx, o->vals);
        FGPU_HIP(hipGetLastError());
    }
    fgpu_info i = mat_finalize(o);
    if (i != FGPU_OK) { mat_release(o); return i; }
    *out = o;
    return FGPU_OK;
}

static fgpu_info mat_intersect_impl(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, const fgpu_mat* b) {
    FGPU_REQUIRE(ctx && out && a && b, FGPU_NULL_POINTER, "fgpu_mat_intersect: NULL argument");
    return intersect_impl(ctx, out, a, b, nullptr);
}
fgpu_info fgpu_mat_intersect_nvals(fgpu_ctx* ctx, const fgpu_mat* a, const fgpu_mat* b, uint64_t* out) {
    FGPU_REQUIRE(ctx && out && a && b, FGPU_NULL_POINTER, "fgpu_mat_intersect_nvals: NULL argument");
    return intersect_impl(ctx, nullptr, a, b, out);
}

// ---- slabs (multi-GPU sharding helpers) ----------------------------------------------
fgpu_info fgpu_mat_row_degrees(fgpu_ctx* ctx, const fgpu_mat* a, uint32_t* out_dev) {
    FGPU_REQUIRE(ctx && a && out_dev, FGPU_NULL_POINTER, "fgpu_mat_row_degrees: NULL argument");
    DevBuf<u32> rp;
    FGPU_TRY(dense_rowptr(ctx, a, rp));
    if (a->nrows) {
        hipLaunchKernelGGL(row_degree_kernel, dim3(cdiv(a->nrows, 256)), dim3(256), 0, ctx->stream(),
                           (const u32*)rp.p, (u32)a->nrows, out_dev);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_HIP(hipStreamSynchronize(ctx->stream()));   // `rp` may return to the pool
    return FGPU_OK;
}

static fgpu_info mat_col_slab_impl(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t lo, uint64_t hi) {
    FGPU_REQUIRE(ctx && out && a, FGPU_NULL_POINTER, "fgpu_mat_col_slab: NULL argument");
    FGPU_REQUIRE(!a->is_hyper() && !a->vals, FGPU_INVALID, "fgpu_mat_col_slab: needs a non-hypersparse pattern matrix");
    if (hi > a->ncols) hi = a->ncols;
    if (lo > hi) lo = hi;
    const u64 nrows = a->nrows;
    DevBuf<u32> cnt, rowptr;
    FGPU_TRY(cnt.alloc(ctx, nrows + 1));
    FGPU_HIP(hipMemsetAsync(cnt.p, 0, (nrows + 1) * sizeof(u32), ctx->stream()));
    if (nrows) {
        hipLaunchKernelGGL(col_slab_count_kernel, dim3(cdiv(nrows, 256)), dim3(256), 0, ctx->stream(), view_of(a),
                           (u32)lo, (u32)hi, cnt.p);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(rowptr.alloc(ctx, nrows + 1));
    FGPU_TRY(scan_u32(ctx, cnt.p, rowptr.p, nrows + 1, nullptr));
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, rowptr.p + nrows, &nnz));
    fgpu_mat* o = nullptr;
    FGPU_TRY(mat_alloc(ctx, &o, nrows, a->ncols, nnz, false, 0, false));
    FGPU_HIP(hipMemcpyAsync(o->rowptr, rowptr.p, (nrows + 1) * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream()));
    if (nrows && nnz) {
        u32 grid = cdiv(nrows, 4);
        if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
        hipLaunchKernelGGL(col_slab_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(a), (u32)lo,
                           (u32)hi, (const u32*)o->rowptr, o->colidx);
        FGPU_HIP(hipGetLastError());
    }
    fgpu_info i = mat_finalize(o);
    if (i != FGPU_OK) { mat_release(o); return i; }
    *out = o;
    return FGPU_OK;
}

static fgpu_info mat_row_slab_impl(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t lo, uint64_t hi) {
    FGPU_REQUIRE(ctx && out && a, FGPU_NULL_POINTER, "fgpu_mat_row_slab: NULL argument");
    FGPU_REQUIRE(!a->is_hyper() && !a->vals, FGPU_INVALID, "fgpu_mat_row_slab: needs a non-hypersparse pattern matrix");
    if (hi > a->nrows) hi = a->nrows;
    if (lo > hi) lo = hi;
    const u64 nrows = a->nrows;
    DevBuf<u32> cnt, rowptr;
    FGPU_TRY(cnt.alloc(ctx, nrows + 1));
    hipLaunchKernelGGL(row_slab_count_kernel, dim3(cdiv(nrows + 1, 256)), dim3(256), 0, ctx->stream(), view_of(a),
                       (u32)lo, (u32)hi, cnt.p);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(rowptr.alloc(ctx, nrows + 1));
    FGPU_TRY(scan_u32(ctx, cnt.p, rowptr.p, nrows + 1, nullptr));
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, rowptr.p + nrows, &nnz));
    fgpu_mat* o = nullptr;
    FGPU_TRY(mat_alloc(ctx, &o, nrows, a->ncols, nnz, false, 0, false));
    FGPU_HIP(hipMemcpyAsync(o->rowptr, rowptr.p, (nrows + 1) * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream()));
    if (nnz) {
        // rows [lo,hi) are contiguous in a's colidx
        u32 b = 0;
        FGPU_TRY(read_u32(ctx, a->rowptr + lo, &b));
        FGPU_HIP(hipMemcpyAsync(o->colidx, a->colidx + b, (size_t)nnz * sizeof(u32), hipMemcpyDeviceToDevice,
                                ctx->stream()));
    }
    fgpu_info i = mat_finalize(o);
    if (i != FGPU_OK) { mat_release(o); return i; }
    *out = o;
    return FGPU_OK;
}

}  // extern "C"

// Public producers of snapshots: the implementation above, then fgpu_ctx::publish().
extern "C" {

fgpu_info fgpu_mat_new(fgpu_ctx* ctx, fgpu_mat** out, uint64_t nrows, uint64_t ncols) {
    fgpu_info i_ = mat_new_impl(ctx, out, nrows, ncols);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

fgpu_info fgpu_mat_from_coo(fgpu_ctx* ctx, fgpu_mat** out, uint64_t nrows, uint64_t ncols, const uint64_t* rows,
                            const uint64_t* cols, const uint64_t* vals, uint64_t n) {
    fgpu_info i_ = mat_from_coo_impl(ctx, out, nrows, ncols, rows, cols, vals, n);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

fgpu_info fgpu_mat_from_csr(fgpu_ctx* ctx, fgpu_mat** out, uint64_t nrows, uint64_t ncols, uint64_t nnz,
                            const void* rowptr, int rowptr_bits, const void* colidx, int colidx_bits,
                            const uint64_t* vals, const uint64_t* hyper_rows, uint64_t nvec) {
    fgpu_info i_ = mat_from_csr_impl(ctx, out, nrows, ncols, nnz, rowptr, rowptr_bits, colidx, colidx_bits, vals, hyper_rows, nvec);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

fgpu_info fgpu_mat_rmat(fgpu_ctx* ctx, fgpu_mat** out, int scale, int edge_factor, uint64_t seed, uint32_t a16,
                        uint32_t b16, uint32_t c16) {
    fgpu_info i_ = mat_rmat_impl(ctx, out, scale, edge_factor, seed, a16, b16, c16);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

fgpu_info fgpu_mat_transpose(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a) {
    fgpu_info i_ = mat_transpose_impl(ctx, out, a);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

fgpu_info fgpu_mat_merge(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm,
                         int dm_masks_dp) {
    fgpu_info i_ = mat_merge_impl(ctx, out, m, dp, dm, dm_masks_dp);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

fgpu_info fgpu_mat_intersect(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, const fgpu_mat* b) {
    fgpu_info i_ = mat_intersect_impl(ctx, out, a, b);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

fgpu_info fgpu_mat_col_slab(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t lo, uint64_t hi) {
    fgpu_info i_ = mat_col_slab_impl(ctx, out, a, lo, hi);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

fgpu_info fgpu_mat_row_slab(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t lo, uint64_t hi) {
    fgpu_info i_ = mat_row_slab_impl(ctx, out, a, lo, hi);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

}  // extern "C"
