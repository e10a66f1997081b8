// merge.hip — entry-parallel Delta merge on device, pattern or UINT64-valued layers.
//
//   out = (m \ dm) U dp            (dm_masks_dp == false: VersionedMatrix::extract,
//                                   versioned_matrix.rs:609-620; Tensor structure, tensor.rs:793-808)
//   out = (m U dp) \ dm            (dm_masks_dp == true: the (true,true) arm of flush,
//                                   versioned_matrix.rs:905-911 / tensor.rs:724-731)
// dp's value wins on a coordinate stored in both m and dp (GrB_SECOND_UINT64, matrix.rs:852-874);
// entries of a BOOL layer merged into a valued result carry the value 1 (iso true).
// The same machinery serves GrB_Matrix_resize (Matrix::resize / grown, matrix.rs:576-598,
// tensor.rs:613-667): entries at or past the new dims are dropped, rows are extended.
//
// Every layer is walked ENTRY-parallel (one lane per stored entry, 64 consecutive entries per
// wavefront), so R-MAT hub rows cost no more than any other 64 entries:
//   1. mark    : keep bit per entry (ballot -> one 64-bit word per wavefront step)
//   2. scan    : exclusive prefix of the per-word popcounts
//   3. rowlen  : out_len[r] = kept(m row r) + kept(dp row r)  ->  scan  -> out rowptr
//   4. scatter : kept entry -> out_rowptr[r] + own rank + rank of its column in the other layer's
//                kept entries (binary search in the other layer's row; both rows are sorted)
// HBM traffic (pattern): 2 x 4 nnz(m) read + 4 nnz(out) written + O(N) row arrays, against
// B_alg = 4(nnz(m)+nnz(dp)+nnz(dm)) + 4 nnz(out) + 8(N+1).
#include "common.hpp"

namespace fgpu {

struct Layer {        // dense-rowptr CSR view of one layer
    const u32* rp;    // nrows_layer + 1
    const u32* col;
    const u64* val;   // nullable
    u32 nrows;
    u32 nnz;
};

// largest r in [lo, hi] with rp[r] <= p (rows may be empty: equal row pointers)
__device__ __forceinline__ u32 row_of(const u32* __restrict__ rp, u32 lo, u32 hi, u32 p) {
    while (lo < hi) {
        u32 mid = (lo + hi + 1) >> 1;
        if (rp[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ u32 lower_bound_col(const u32* __restrict__ col, u32 b, u32 e, u32 key) {
    while (b < e) {
        u32 mid = (b + e) >> 1;
        if (col[mid] < key) b = mid + 1; else e = mid;
    }
    return b;
}

__device__ __forceinline__ bool layer_has(const Layer& l, u32 r, u32 c) {
    if (r >= l.nrows) return false;
    u32 b = l.rp[r], e = l.rp[r + 1];
    if (b == e) return false;
    u32 p = lower_bound_col(l.col, b, e, c);
    return p < e && l.col[p] == c;
}

// kept entries before position p of a layer: ks[p >> 6] + popcount(kb[p >> 6] below bit p & 63)
__device__ __forceinline__ u32 kept_before(const u64* __restrict__ kb, const u32* __restrict__ ks, u32 p) {
    u64 w = kb[p >> 6];
    u32 s = p & 63;
    return ks[p >> 6] + (u32)__popcll(s ? (w & ((1ull << s) - 1ull)) : 0ull);
}

// rows of the 64 entries [base, base+64) of `x`: scalar search for the wave's row window, then a
// short per-lane search inside it.
__device__ __forceinline__ u32 lane_row(const Layer& x, u32 base, u32 p, bool valid) {
    u32 last = base + 63 < x.nnz ? base + 63 : x.nnz - 1;
    u32 rlo = row_of(x.rp, 0, x.nrows - 1, base);
    u32 rhi = row_of(x.rp, rlo, x.nrows - 1, last);
    return valid ? row_of(x.rp, rlo, rhi, p) : rlo;
}

// IS_M: x = m, other = dp (a coordinate also stored in dp is dropped from m: dp's value wins),
//       and dm always masks m.   !IS_M: x = dp, masked by dm only when dm_masks_dp.
template <bool IS_M>
__global__ __launch_bounds__(256) void merge_mark_kernel(Layer x, Layer other, Layer dm, bool has_other, bool has_dm,
                                                        bool dm_masks_dp, u32 out_nrows, u32 out_ncols,
                                                        u64* __restrict__ kb, u32* __restrict__ kc) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (x.nnz + 63) >> 6;
    for (u32 w = wave; w < nwords; w += nwaves) {
        const u32 base = __builtin_amdgcn_readfirstlane(w << 6);
        const u32 p = base + lane;
        const bool valid = p < x.nnz;
        const u32 r = lane_row(x, base, p, valid);
        bool keep = false;
        if (valid) {
            const u32 c = x.col[p];
            keep = r < out_nrows && c < out_ncols;
            if (IS_M) {
                if (keep && has_dm) keep = !layer_has(dm, r, c);
                if (keep && has_other) keep = !layer_has(other, r, c);
            } else {
                if (keep && dm_masks_dp && has_dm) keep = !layer_has(dm, r, c);
            }
        }
        const u64 mask = __ballot(keep);
        if (lane == 0) {
            kb[w] = mask;
            kc[w] = (u32)__popcll(mask);
        }
    }
}

__global__ void merge_rowlen_kernel(Layer m, Layer dp, bool has_dp, const u64* __restrict__ kbm,
                                    const u32* __restrict__ ksm, const u64* __restrict__ kbp,
                                    const u32* __restrict__ ksp, u32 out_nrows, u32* __restrict__ len) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > out_nrows) return;
    u32 n = 0;
    if (r < out_nrows) {
        if (r < m.nrows && m.nnz) {
            u32 b = m.rp[r], e = m.rp[r + 1];
            if (b != e) n += kept_before(kbm, ksm, e) - kept_before(kbm, ksm, b);
        }
        if (has_dp && r < dp.nrows) {
            u32 b = dp.rp[r], e = dp.rp[r + 1];
            if (b != e) n += kept_before(kbp, ksp, e) - kept_before(kbp, ksp, b);
        }
    }
    len[r] = n;
}

template <bool IS_M>
__global__ __launch_bounds__(256) void merge_scatter_kernel(Layer x, Layer other, bool has_other,
                                                           const u64* __restrict__ kbx, const u32* __restrict__ ksx,
                                                           const u64* __restrict__ kbo, const u32* __restrict__ kso,
                                                           const u32* __restrict__ out_rp, u32* __restrict__ out_col,
                                                           u64* __restrict__ out_val) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (x.nnz + 63) >> 6;
    for (u32 w = wave; w < nwords; w += nwaves) {
        const u32 base = __builtin_amdgcn_readfirstlane(w << 6);
        const u64 mask = kbx[w];
        if (mask == 0) continue;
        const u32 p = base + lane;
        const bool valid = p < x.nnz;
        const u32 r = lane_row(x, base, p, valid);
        if (!((mask >> lane) & 1ull)) continue;
        const u32 c = x.col[p];
        const u32 own = ksx[w] + (u32)__popcll(lane ? (mask & ((1ull << lane) - 1ull)) : 0ull) -
                        kept_before(kbx, ksx, x.rp[r]);
        u32 cross = 0;
        if (has_other && r < other.nrows) {
            u32 b = other.rp[r], e = other.rp[r + 1];
            if (b != e) {
                u32 q = lower_bound_col(other.col, b, e, c);
                cross = kept_before(kbo, kso, q) - kept_before(kbo, kso, b);
            }
        }
        const u32 pos = out_rp[r] + own + cross;
        out_col[pos] = c;
        if (out_val) out_val[pos] = x.val ? x.val[p] : 1ull;
    }
}

// dense-rowptr view of a (possibly hypersparse) matrix; `tmp` keeps the expanded row pointers alive
static fgpu_info layer_of(fgpu_ctx* ctx, const fgpu_mat* a, DevBuf<u32>& tmp, Layer& l) {
    l.col = a->colidx;
    l.val = a->vals;
    l.nrows = (u32)a->nrows;
    l.nnz = (u32)a->nnz;
    if (!a->is_hyper()) {
        l.rp = a->rowptr;
        return FGPU_OK;
    }
    FGPU_TRY(dense_rowptr(ctx, a, tmp));
    l.rp = tmp.p;
    return FGPU_OK;
}

struct Keep {  // keep bits + exclusive prefix of their per-word popcounts
    DevBuf<u64> kb;
    DevBuf<u32> ks;
    fgpu_info alloc(fgpu_ctx* ctx, u32 nnz) {
        u32 nwords = (nnz + 63) >> 6;
        FGPU_TRY(kb.alloc(ctx, (size_t)nwords + 1));
        FGPU_TRY(ks.alloc(ctx, (size_t)nwords + 1));
        // the word past the end is read by kept_before(nnz) when nnz is a multiple of 64
        FGPU_HIP(hipMemsetAsync(kb.p + nwords, 0, sizeof(u64), ctx->stream));
        FGPU_HIP(hipMemsetAsync(ks.p + nwords, 0, sizeof(u32), ctx->stream));
        return FGPU_OK;
    }
};

static u32 entry_grid(fgpu_ctx* ctx, u32 nnz) {
    u32 g = cdiv(((u64)nnz + 63) >> 6, 4);
    u32 cap = (u32)ctx->cus * 32;
    return g < 1 ? 1 : (g > cap ? cap : g);
}

fgpu_info mat_merge_entries(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm,
                            bool dm_masks_dp, u64 out_nrows, u64 out_ncols, bool pattern_only) {
    const bool has_dp = dp && dp->nnz, has_dm = dm && dm->nnz;
    const bool with_vals = !pattern_only && (m->vals != nullptr || (dp && dp->vals != nullptr));
    DevBuf<u32> trm, trp, trd;
    Layer lm{}, lp{}, ld{};
    FGPU_TRY(layer_of(ctx, m, trm, lm));
    if (has_dp) FGPU_TRY(layer_of(ctx, dp, trp, lp));
    if (has_dm) FGPU_TRY(layer_of(ctx, dm, trd, ld));
    Keep km, kp;
    FGPU_TRY(km.alloc(ctx, lm.nnz));
    FGPU_TRY(kp.alloc(ctx, has_dp ? lp.nnz : 0));
    if (lm.nnz) {
        hipLaunchKernelGGL(merge_mark_kernel<true>, dim3(entry_grid(ctx, lm.nnz)), dim3(256), 0, ctx->stream, lm, lp,
                           ld, has_dp, has_dm, dm_masks_dp, (u32)out_nrows, (u32)out_ncols, km.kb.p, km.ks.p);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(scan_u32(ctx, km.ks.p, km.ks.p, ((u64)(lm.nnz + 63) >> 6) + 1, nullptr));
    if (has_dp) {
        hipLaunchKernelGGL(merge_mark_kernel<false>, dim3(entry_grid(ctx, lp.nnz)), dim3(256), 0, ctx->stream, lp, lm,
                           ld, lm.nnz != 0, has_dm, dm_masks_dp, (u32)out_nrows, (u32)out_ncols, kp.kb.p, kp.ks.p);
        FGPU_HIP(hipGetLastError());
        FGPU_TRY(scan_u32(ctx, kp.ks.p, kp.ks.p, ((u64)(lp.nnz + 63) >> 6) + 1, nullptr));
    }
    DevBuf<u32> orp;
    FGPU_TRY(orp.alloc(ctx, out_nrows + 1));
    hipLaunchKernelGGL(merge_rowlen_kernel, dim3(cdiv(out_nrows + 1, 256)), dim3(256), 0, ctx->stream, lm, lp, has_dp,
                       (const u64*)km.kb.p, (const u32*)km.ks.p, (const u64*)kp.kb.p, (const u32*)kp.ks.p,
                       (u32)out_nrows, orp.p);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(scan_u32(ctx, orp.p, orp.p, out_nrows + 1, nullptr));
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, orp.p + out_nrows, &nnz));
    fgpu_mat* o = nullptr;
    FGPU_TRY(mat_alloc(ctx, &o, out_nrows, out_ncols, nnz, with_vals, 0, false));
    hipError_t e = hipMemcpyAsync(o->rowptr, orp.p, (out_nrows + 1) * sizeof(u32), hipMemcpyDeviceToDevice,
                                  ctx->stream);
    if (e == hipSuccess && lm.nnz && nnz) {
        hipLaunchKernelGGL(merge_scatter_kernel<true>, dim3(entry_grid(ctx, lm.nnz)), dim3(256), 0, ctx->stream, lm,
                           lp, has_dp, (const u64*)km.kb.p, (const u32*)km.ks.p, (const u64*)kp.kb.p,
                           (const u32*)kp.ks.p, (const u32*)o->rowptr, o->colidx, o->vals);
        e = hipGetLastError();
    }
    if (e == hipSuccess && has_dp && nnz) {
        hipLaunchKernelGGL(merge_scatter_kernel<false>, dim3(entry_grid(ctx, lp.nnz)), dim3(256), 0, ctx->stream, lp,
                           lm, lm.nnz != 0, (const u64*)kp.kb.p, (const u32*)kp.ks.p, (const u64*)km.kb.p,
                           (const u32*)km.ks.p, (const u32*)o->rowptr, o->colidx, o->vals);
        e = hipGetLastError();
    }
    fgpu_info i = FGPU_OK;
    if (e != hipSuccess) {
        set_error("merge launch failed: %s", hipGetErrorString(e));
        i = FGPU_DEVICE;
    }
    if (i == FGPU_OK) i = mat_finalize(o);
    if (i != FGPU_OK) { fgpu_mat_free(o); return i; }
    *out = o;
    return FGPU_OK;
}

// ---------------------------------------------------------------------------------------
// value fills by search: the structure comes from the pattern builders, the values are
// placed by locating each source entry in the finished CSR.
// ---------------------------------------------------------------------------------------
// COO build with values: the LAST duplicate of a coordinate wins (a legal GxB_ANY_UINT64 choice,
// matrix.rs:1186-1210, made deterministic): win[pos] = max tuple index, then vals[pos] = in[win].
__global__ void coo_winner_kernel(const u32* __restrict__ rows, const u32* __restrict__ cols, u64 n, Layer a,
                                  u32* __restrict__ win) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u32 r = rows[i], c = cols[i];
        if (r == 0xFFFFFFFFu) continue;  // tuple dropped by the generator (mat_from_device_coo's ROW_INVALID)
        u32 b = a.rp[r], e = a.rp[r + 1];
        u32 p = lower_bound_col(a.col, b, e, c);
        atomicMax(&win[p], (u32)i);
    }
}
__global__ void coo_take_winner_kernel(const u32* __restrict__ win, const u64* __restrict__ in, u32 nnz,
                                       u64* __restrict__ out) {
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) out[p] = in[win[p]];
}

fgpu_info mat_from_device_coo_vals(fgpu_ctx* ctx, fgpu_mat** out, u64 nrows, u64 ncols, const u32* rows,
                                   const u32* cols, const u64* vals, u64 n) {
    fgpu_mat* a = nullptr;
    FGPU_TRY(mat_from_device_coo(ctx, &a, nrows, ncols, rows, cols, n));
    fgpu_info i = FGPU_OK;
    do {
        if ((i = ctx->dev_alloc((void**)&a->vals, (size_t)(a->nnz ? a->nnz : 1) * sizeof(u64))) != FGPU_OK) break;
        if (a->nnz == 0) break;
        DevBuf<u32> win;
        if ((i = win.alloc(ctx, a->nnz)) != FGPU_OK) break;
        hipError_t e = hipMemsetAsync(win.p, 0, a->nnz * sizeof(u32), ctx->stream);
        if (e == hipSuccess) {
            Layer la{a->rowptr, a->colidx, nullptr, (u32)a->nrows, (u32)a->nnz};
            hipLaunchKernelGGL(coo_winner_kernel, dim3(ctx->cus * 16), dim3(256), 0, ctx->stream, rows, cols, n, la,
                               win.p);
            hipLaunchKernelGGL(coo_take_winner_kernel, dim3(cdiv(a->nnz, 256)), dim3(256), 0, ctx->stream,
                               (const u32*)win.p, vals, (u32)a->nnz, a->vals);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // `win` returns to the pool
        if (e != hipSuccess) {
            set_error("valued COO build failed: %s", hipGetErrorString(e));
            i = FGPU_DEVICE;
        }
    } while (0);
    if (i != FGPU_OK) { fgpu_mat_free(a); return i; }
    *out = a;
    return FGPU_OK;
}

// transpose with values: structure from the pattern transpose, then every entry (r, c, v) of `a`
// is dropped at (c, r) of the result.
__global__ __launch_bounds__(256) void transpose_vals_kernel(Layer a, Layer t, u64* __restrict__ tvals) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (a.nnz + 63) >> 6;
    for (u32 w = wave; w < nwords; w += nwaves) {
        const u32 base = __builtin_amdgcn_readfirstlane(w << 6);
        const u32 p = base + lane;
        const bool valid = p < a.nnz;
        const u32 r = lane_row(a, base, p, valid);
        if (!valid) continue;
        const u32 c = a.col[p];
        u32 q = lower_bound_col(t.col, t.rp[c], t.rp[c + 1], r);
        tvals[q] = a.val[p];
    }
}

fgpu_info mat_transpose_vals(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a) {
    fgpu_mat* t = nullptr;
    FGPU_TRY(mat_transpose_pattern(ctx, &t, a));
    fgpu_info i = FGPU_OK;
    do {
        if (t->is_hyper()) {  // empty result: fgpu_mat_new form
            i = ctx->dev_alloc((void**)&t->vals, sizeof(u64));
            break;
        }
        if ((i = ctx->dev_alloc((void**)&t->vals, (size_t)(t->nnz ? t->nnz : 1) * sizeof(u64))) != FGPU_OK) break;
        if (a->nnz == 0) break;
        DevBuf<u32> tra;
        Layer la{}, lt{t->rowptr, t->colidx, nullptr, (u32)t->nrows, (u32)t->nnz};
        if ((i = layer_of(ctx, a, tra, la)) != FGPU_OK) break;
        hipLaunchKernelGGL(transpose_vals_kernel, dim3(entry_grid(ctx, la.nnz)), dim3(256), 0, ctx->stream, la, lt,
                           t->vals);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("valued transpose failed: %s", hipGetErrorString(e));
            i = FGPU_DEVICE;
        }
    } while (0);
    if (i != FGPU_OK) { fgpu_mat_free(t); return i; }
    *out = t;
    return FGPU_OK;
}

}  // namespace fgpu

extern "C" {

fgpu_info fgpu_mat_resize(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t nrows, uint64_t ncols) {
    FGPU_REQUIRE(ctx && out && a, FGPU_NULL_POINTER, "fgpu_mat_resize: NULL argument");
    FGPU_REQUIRE(nrows < 0xFFFFFFFFull && ncols < 0xFFFFFFFFull, FGPU_INVALID,
                 "fgpu_mat_resize: dims exceed the 32-bit id space");
    return fgpu::mat_merge_entries(ctx, out, a, nullptr, nullptr, false, nrows, ncols, false);
}

fgpu_info fgpu_mat_merge_pattern(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp,
                                 const fgpu_mat* dm, int dm_masks_dp) {
    FGPU_REQUIRE(ctx && out && m, FGPU_NULL_POINTER, "fgpu_mat_merge_pattern: NULL argument");
    FGPU_REQUIRE(!dp || (dp->nrows == m->nrows && dp->ncols == m->ncols), FGPU_DIM_MISMATCH,
                 "fgpu_mat_merge_pattern: dp dims differ from m");
    FGPU_REQUIRE(!dm || (dm->nrows == m->nrows && dm->ncols == m->ncols), FGPU_DIM_MISMATCH,
                 "fgpu_mat_merge_pattern: dm dims differ from m");
    return fgpu::mat_merge_entries(ctx, out, m, dp, dm, dm_masks_dp != 0, m->nrows, m->ncols, true);
}

}  // extern "C"
