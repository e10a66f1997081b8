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
// Every layer is walked ENTRY-parallel (one lane per stored entry, 64 consecutive entries = one "word"
// per wavefront step), so R-MAT hub rows cost no more than any other 64 entries:
//   0. wordrow : stored-row index of the first entry of every word (cached on the snapshot: it only
//                depends on the row pointers) — a lane finds its row with a 1-3 step search between
//                wordrow[w] and wordrow[w+1] instead of a log2(nrows) search
//      rowbits : one bit per row "dp or dm stores something in this row"; with 0.1 % deltas 99 % of
//                the base entries skip every delta lookup after one cached load
//   1. mark    : keep bit per entry (ballot -> one 64-bit word per wavefront step)
//   2. scan    : exclusive prefix of the per-word popcounts
//   3. rowlen  : out_len[r] = kept(m row r) + kept(dp row r)  ->  scan  -> out rowptr
//   4. scatter : kept entry -> out_rowptr[r] + own rank + rank of its column in the other layer's
//                kept entries (binary search in the other layer's row; both rows are sorted)
// HBM traffic (pattern): 2 x 4 nnz(m) read + 4 nnz(out) written + O(N) row arrays, against
// B_alg = 4(nnz(m)+nnz(dp)+nnz(dm)) + 4 nnz(out) + 8(N+1).
#include "common.hpp"

namespace fgpu {

struct Layer {        // one layer: the (possibly hypersparse) CSR view, its values and its wordrow index
    CsrView v;
    const u64* val;   // nullable
    const u32* wordrow;
    u32 nnz;
};

// largest i in [lo, hi] with rp[i] <= p (rows may be empty: equal row pointers)
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
    u32 b, e;
    row_range(l.v, r, b, e);
    if (b == e) return false;
    u32 p = lower_bound_col(l.v.colidx, b, e, c);
    return p < e && l.v.colidx[p] == c;
}

// kept entries before position p of a layer: ks[p >> 6] + popcount(kb[p >> 6] below bit p & 63)
__device__ __forceinline__ u32 kept_before(const u64* __restrict__ kb, const u32* __restrict__ ks, u32 p) {
    u64 w = kb[p >> 6];
    u32 s = p & 63;
    return ks[p >> 6] + (u32)__popcll(s ? (w & ((1ull << s) - 1ull)) : 0ull);
}

// stored-row index of the first entry of each 64-entry word; wordrow[nwords] = last stored row
__global__ void wordrow_kernel(const u32* __restrict__ rp, u32 nvec, u32 nwords, u32* __restrict__ wordrow) {
    u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwords) return;
    wordrow[w] = w < nwords ? row_of(rp, 0, nvec - 1, w << 6) : nvec - 1;
}

__global__ void rowbits_kernel(CsrView d, u32* __restrict__ bits) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.nvec) return;
    if (d.rowptr[i] == d.rowptr[i + 1]) return;
    u32 r = d.hrows ? d.hrows[i] : i;
    atomicOr(&bits[r >> 5], 1u << (r & 31));
}

// stored-row index and row id of entry p = (w << 6) + lane of layer x
__device__ __forceinline__ void lane_row(const Layer& x, u32 w, u32 p, bool valid, u32& i, u32& r) {
    const u32 ilo = x.wordrow[w], ihi = x.wordrow[w + 1];
    i = valid ? row_of(x.v.rowptr, ilo, ihi, p) : ilo;
    r = x.v.hrows ? x.v.hrows[i] : i;
}

// CLEAN words: no row from the first entry's row to the next word's first row is touched by dp or dm (one or two
// rowbits words, a lane each, one ballot).  Such a word's 64 entries are all kept and move by ONE offset
// (out_rowptr[r0] - rowptr[i0]: nothing is inserted or removed between its rows, empty rows included), so mark is a
// constant and scatter a shifted copy — ~10 instructions per word-lane against ~120 for the per-entry path, which
// is what bounds this kernel (VALU issue, not bytes).  With 0.1 % deltas on R-MAT-22 about 80 % of the words are clean.
__device__ __forceinline__ bool word_clean(const Layer& x, u32 w, const u32* __restrict__ rowbits, u32 lane) {
    if (!rowbits) return true;
    const u32 ilo = x.wordrow[w], ihi = x.wordrow[w + 1];
    const u32 rlo = x.v.hrows ? x.v.hrows[ilo] : ilo;
    const u32 rhi = x.v.hrows ? x.v.hrows[ihi] : ihi;
    const u32 b0 = rlo >> 5, b1 = rhi >> 5;
    if (b1 - b0 >= 64u) return false;   // a long run of rows (the last word of a layer): per-entry path
    u32 v = 0;
    if (lane <= b1 - b0) {
        v = rowbits[b0 + lane];
        if (lane == 0) v &= ~0u << (rlo & 31);
        if (lane == b1 - b0) v &= (rhi & 31) == 31 ? ~0u : ((1u << ((rhi & 31) + 1)) - 1u);
    }
    return __ballot(v != 0) == 0ull;
}

// IS_M: x = m, other = dp (a coordinate also stored in dp is dropped from m: dp's value wins),
//       and dm always masks m.   !IS_M: x = dp, masked by dm only when dm_masks_dp.
template <bool IS_M>
__global__ __launch_bounds__(256) void merge_mark_kernel(Layer x, Layer other, Layer dm, bool has_other, bool has_dm,
                                                        bool dm_masks_dp, const u32* __restrict__ rowbits,
                                                        u32 out_nrows, u32 out_ncols, u64* __restrict__ kb,
                                                        u32* __restrict__ kc, bool clip) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (x.nnz + 63) >> 6;
    for (u32 w = wave; w < nwords; w += nwaves) {
        const u32 p = (w << 6) + lane;
        const bool valid = p < x.nnz;
        if (IS_M && !clip && word_clean(x, w, rowbits, lane)) {   // wave-uniform
            const u64 all = __ballot(valid);
            if (lane == 0) {
                kb[w] = all;
                kc[w] = (u32)__popcll(all);
            }
            continue;
        }
        u32 i, r;
        lane_row(x, w, p, valid, i, r);
        bool keep = false;
        if (valid) {
            const u32 c = x.v.colidx[p];
            keep = r < out_nrows && c < out_ncols;
            if (keep && rowbits && ((rowbits[r >> 5] >> (r & 31)) & 1u)) {
                if (IS_M) {
                    if (has_dm) keep = !layer_has(dm, r, c);
                    if (keep && has_other) keep = !layer_has(other, r, c);
                } else {
                    if (dm_masks_dp && has_dm) keep = !layer_has(dm, r, c);
                }
            }
        }
        const u64 mask = __ballot(keep);
        if (lane == 0) {
            kb[w] = mask;
            kc[w] = (u32)__popcll(mask);
        }
    }
}

// The base layer the other way round (no clipping): every entry is kept until a delta entry says otherwise, so the
// keep bits start as all-ones and each dm / dp entry clears its own coordinate in m (one binary search in m's row per
// DELTA entry).  The per-entry form above asks the question from m's side — every entry of a touched row searches dm
// and dp — and R-MAT tombstones land on hub rows in proportion to their length: with 0.1 % uniformly random tombstones
// 47 % of the 64-entry words of RMAT-20 lie in a touched row, each of their entries paying a 16-step search of the
// hypersparse row list.
__global__ void merge_markall_kernel(u32 nnz, u64* __restrict__ kb) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nwords = (nnz + 63) >> 6;
    if (w >= nwords) return;
    const u32 left = nnz - (w << 6);
    kb[w] = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
}
__global__ __launch_bounds__(256) void merge_unmark_kernel(Layer d, Layer m, u64* __restrict__ kb) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (d.nnz + 63) >> 6;
    for (u32 w = wave; w < nwords; w += nwaves) {
        const u32 q = (w << 6) + lane;
        if (q >= d.nnz) continue;
        u32 i, r;
        lane_row(d, w, q, true, i, r);
        const u32 c = d.v.colidx[q];
        u32 b, e;
        row_range(m.v, r, b, e);
        if (b == e) continue;
        const u32 p = lower_bound_col(m.v.colidx, b, e, c);
        if (p < e && m.v.colidx[p] == c) atomicAnd((unsigned long long*)&kb[p >> 6], ~(1ull << (p & 63)));
    }
}
__global__ void merge_popc_kernel(const u64* __restrict__ kb, u32 nwords, u32* __restrict__ kc) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwords) kc[w] = (u32)__popcll(kb[w]);
}

__global__ void merge_rowlen_kernel(Layer m, Layer dp, bool has_dp, const u32* __restrict__ rowbits,
                                    const u64* __restrict__ kbm, const u32* __restrict__ ksm,
                                    const u64* __restrict__ kbp, const u32* __restrict__ ksp, u32 out_nrows,
                                    u32* __restrict__ len) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > out_nrows) return;
    u32 n = 0;
    if (r < out_nrows) {
        u32 b, e;
        if (m.nnz) {
            row_range(m.v, r, b, e);
            if (b != e) n += kept_before(kbm, ksm, e) - kept_before(kbm, ksm, b);
        }
        if (has_dp && ((rowbits[r >> 5] >> (r & 31)) & 1u)) {
            row_range(dp.v, r, b, e);
            if (b != e) n += kept_before(kbp, ksp, e) - kept_before(kbp, ksp, b);
        }
    }
    len[r] = n;
}

template <bool IS_M>
__global__ __launch_bounds__(256) void merge_scatter_kernel(Layer x, Layer other, bool has_other,
                                                           const u32* __restrict__ rowbits,
                                                           const u64* __restrict__ kbx, const u32* __restrict__ ksx,
                                                           const u64* __restrict__ kbo, const u32* __restrict__ kso,
                                                           const u32* __restrict__ out_rp, u32* __restrict__ out_col,
                                                           u64* __restrict__ out_val, bool clip,
                                                           const u32* __restrict__ crossbits /* rows the OTHER layer stores (nullable = rowbits) */) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (x.nnz + 63) >> 6;
    for (u32 w = wave; w < nwords; w += nwaves) {
        const u64 mask = kbx[w];
        if (mask == 0) continue;
        const u32 p = (w << 6) + lane;
        if (IS_M && !clip && word_clean(x, w, rowbits, lane)) {   // wave-uniform: a shifted copy
            const u32 i0 = x.wordrow[w];
            const u32 r0 = x.v.hrows ? x.v.hrows[i0] : i0;
            const u32 shift = out_rp[r0] - x.v.rowptr[i0];
            if (p < x.nnz) {
                out_col[p + shift] = x.v.colidx[p];
                if (out_val) out_val[p + shift] = x.val ? x.val[p] : 1ull;
            }
            continue;
        }
        // A word that lies inside ONE row (the words of a hub row a delta touches: a fifth of an R-MAT layer's entries
        // with 0.1 % deltas, while the rows touched are 3 %): the row, its kept-rank base and the other layer's row are
        // wave-uniform, and the other layer's few entries of the row are compared in registers — a dozen uniform or
        // coalesced loads per word where the per-entry path below makes a dozen dependent random loads per ENTRY
        // (row search, two kept-rank lookups, a binary search in the other row and two more lookups): that path held
        // this kernel at 499 us of the merge's 787 at RMAT-22.
        if (IS_M && !clip && x.wordrow[w] == x.wordrow[w + 1]) {
            const u32 i1 = x.wordrow[w];
            const u32 r1 = x.v.hrows ? x.v.hrows[i1] : i1;
            const u32 own1 = ksx[w] + (u32)__popcll(lane ? (mask & ((1ull << lane) - 1ull)) : 0ull) -
                             kept_before(kbx, ksx, x.v.rowptr[i1]);
            const u32 c1 = p < x.nnz ? x.v.colidx[p] : 0u;
            u32 cross1 = 0;
            bool done1 = true;
            const u32* __restrict__ xb1 = crossbits ? crossbits : rowbits;
            if (has_other && (!xb1 || ((xb1[r1 >> 5] >> (r1 & 31)) & 1u))) {
                u32 b, e;
                row_range(other.v, r1, b, e);
                if (e - b > 64u) done1 = false;   // (a long row on the other side too: per-entry path)
                else if (b != e) {
                    const u32 q = b + lane;
                    const bool live = q < e && ((kbo[q >> 6] >> (q & 63)) & 1ull);
                    const u32 oc = q < e ? other.v.colidx[q] : 0u;
                    u64 lv = __ballot(live);
                    while (lv) {   // wave-uniform: the kept entries of the other layer's row
                        const int j = (int)__builtin_ctzll(lv);
                        lv &= lv - 1ull;
                        cross1 += ((u32)__builtin_amdgcn_readlane((int)oc, j) < c1) ? 1u : 0u;
                    }
                }
            }
            if (done1) {
                if ((mask >> lane) & 1ull) {
                    const u32 pos1 = out_rp[r1] + own1 + cross1;
                    out_col[pos1] = c1;
                    if (out_val) out_val[pos1] = x.val ? x.val[p] : 1ull;
                }
                continue;
            }
        }
        if (!((mask >> lane) & 1ull)) continue;   // kept entries are valid entries
        u32 i, r;
        lane_row(x, w, p, true, i, r);
        const u32 c = x.v.colidx[p];
        const u32 own = ksx[w] + (u32)__popcll(lane ? (mask & ((1ull << lane) - 1ull)) : 0ull) -
                        kept_before(kbx, ksx, x.v.rowptr[i]);
        u32 cross = 0;
        const u32* __restrict__ xb = crossbits ? crossbits : rowbits;
        if (has_other && (!IS_M || !xb || ((xb[r >> 5] >> (r & 31)) & 1u))) {
            u32 b, e;
            row_range(other.v, r, b, e);
            if (b != e) {
                u32 q = lower_bound_col(other.v.colidx, b, e, c);
                cross = kept_before(kbo, kso, q) - kept_before(kbo, kso, b);
            }
        }
        const u32 pos = out_rp[r] + own + cross;
        out_col[pos] = c;
        if (out_val) out_val[pos] = x.val ? x.val[p] : 1ull;
    }
}

// ---- the scatter as a shifted copy with a handful of events per 2048 entries (round 4) ----------------------------------------
// out = the merged (row, col)-ordered sequence, so a kept entry p of m lands at  kept_before(p) + I(p),  I(p) = the number of
// dp entries inserted at or before p — and with 0.1 % deltas I changes every ~1000 entries.  merge_scatter_kernel asks the
// question per entry from m's side (row of the entry, the other layer's row, a search in it, two kept-rank look-ups: a dozen
// dependent loads for every entry of every word that shares a row with a delta — half of an R-MAT layer, whose hub rows always
// hold a tombstone) and stayed at 0.5 ms of the 0.72 ms merge through five rewrites.  Here the delta side answers once:
// Q[k] = the position in m before which dp entry k goes (one search per dp entry, ascending by construction); an item of
// MS_ITEM consecutive entries of m loads the <= 64 events that fall into it ONCE (ibase[item] = their first index) and every
// lane ranks its entries among them with shuffles — no row look-up at all; kb / ks words are read coalesced, 32 per item.
// dp entry k itself lands at kept_before(Q[k]) + k.  (Without dm_masks_dp every dp entry is kept; a coordinate stored in both
// m and dp has lost its keep bit in m already, merge_unmark_kernel.)
constexpr u32 MS_ITEM = 2048;
__global__ __launch_bounds__(256) void merge_qpos_kernel(Layer dp, Layer m, u32* __restrict__ Q) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (dp.nnz + 63) >> 6;
    for (u32 w = wave; w < nwords; w += nwaves) {
        const u32 q = (w << 6) + lane;
        if (q >= dp.nnz) continue;
        u32 i, r;
        lane_row(dp, w, q, true, i, r);
        u32 pos = m.nnz;                                        // a row past m's last one: behind everything
        if (r < m.v.nrows) {
            const u32 b = m.v.rowptr[r], e = m.v.rowptr[r + 1];  // (m is not hypersparse here: an empty row still has its place)
            pos = lower_bound_col(m.v.colidx, b, e, dp.v.colidx[q]);
        }
        Q[q] = pos;
    }
}
// ibase[it] = number of events before the item's first entry
__global__ void merge_ibase_kernel(const u32* __restrict__ Q, u32 nQ, u32 nitems, u32* __restrict__ ibase) {
    const u32 it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= nitems) return;
    const u32 p0 = it * MS_ITEM;
    u32 lo = 0, hi = nQ;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (Q[mid] < p0) lo = mid + 1; else hi = mid; }
    ibase[it] = lo;
}
__global__ __launch_bounds__(256) void merge_copy_items_kernel(Layer m, const u64* __restrict__ kb, const u32* __restrict__ ks,
                                                              const u32* __restrict__ Q, u32 nQ, const u32* __restrict__ ibase,
                                                              u32 nitems, u32* __restrict__ out_col, u64* __restrict__ out_val) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    constexpr u32 WPI = MS_ITEM / 64;                          // words per item: 32
    for (u32 it = wave; it < nitems; it += nwaves) {
        const u32 p0 = it * MS_ITEM;
        const u32 w0 = p0 >> 6;
        const u32 base = ibase[it];
        // the item's metadata in one round of loads: keep word + kept prefix of word w0 + lane (lanes < 32), events base + lane
        const u32 nwords = (m.nnz + 63) >> 6;
        const u64 kbw = (lane < WPI && w0 + lane < nwords) ? kb[w0 + lane] : 0ull;
        const u32 ksw = (lane < WPI && w0 + lane < nwords) ? ks[w0 + lane] : 0u;
        const u32 ev = base + lane < nQ ? Q[base + lane] : 0xFFFFFFFFu;
        const u32 last = p0 + MS_ITEM - 1;
        const bool dense = (u32)__builtin_amdgcn_readlane((int)ev, 63) <= last;   // more than 64 events in this item (wave-uniform)
#pragma unroll 4
        for (u32 t = 0; t < WPI; ++t) {
            const u64 mask = (u64)__shfl((long long)kbw, (int)t, 64);
            if (mask == 0ull) continue;                          // wave-uniform
            const u32 p = p0 + 64 * t + lane;
            const u32 own = (u32)__shfl((int)ksw, (int)t, 64) + (u32)__popcll(lane ? (mask & ((1ull << lane) - 1ull)) : 0ull);
            u32 ins;
            if (!dense) {                                        // events <= p among the 64 loaded ones (sorted): 6 shuffle steps
                u32 lo = 0, hi = 64;
#pragma unroll
                for (int st = 0; st < 7; ++st) {
                    const u32 mid = (lo + hi) >> 1;
                    const u32 v = (u32)__shfl((int)ev, (int)(mid & 63u), 64);
                    if (lo < hi) { if (v <= p) lo = mid + 1; else hi = mid; }
                }
                ins = base + lo;
            } else {                                             // a dense run of insertions: rank in the whole list
                u32 lo = base, hi = nQ;
                while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (Q[mid] <= p) lo = mid + 1; else hi = mid; }
                ins = lo;
            }
            if ((mask >> lane) & 1ull) {
                const u32 pos = own + ins;
                out_col[pos] = m.v.colidx[p];
                if (out_val) out_val[pos] = m.val ? m.val[p] : 1ull;
            }
        }
    }
}
__global__ void merge_dp_place_kernel(Layer dp, const u32* __restrict__ Q, const u64* __restrict__ kbm, const u32* __restrict__ ksm,
                                      u32* __restrict__ out_col, u64* __restrict__ out_val) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= dp.nnz) return;
    const u32 pos = kept_before(kbm, ksm, Q[k]) + k;
    out_col[pos] = dp.v.colidx[k];
    if (out_val) out_val[pos] = dp.val ? dp.val[k] : 1ull;
}

// the wordrow index of a snapshot: built on first use, owned (and freed) by the matrix
fgpu_info mat_wordrow(fgpu_ctx* ctx, const fgpu_mat* a, const u32** out) {
    *out = nullptr;
    if (a->nnz == 0) return FGPU_OK;
    std::lock_guard<std::mutex> idx_guard(a->idx_mu);
    if (!a->wordrow) {
        const u32 nwords = (u32)((a->nnz + 63) >> 6);
        u32* wr = nullptr;
        FGPU_TRY(ctx->dev_alloc((void**)&wr, ((size_t)nwords + 1) * sizeof(u32)));
        hipLaunchKernelGGL(wordrow_kernel, dim3(cdiv((u64)nwords + 1, 256)), dim3(256), 0, ctx->stream(),
                           (const u32*)a->rowptr, a->nvec, nwords, wr);
        hipError_t e = hipGetLastError();
        // complete on the device before it is published: other lanes read it from their own streams
        if (e == hipSuccess && ctx->multi_lane()) e = hipStreamSynchronize(ctx->stream());
        if (e != hipSuccess) {
            ctx->dev_free(wr);
            set_error("wordrow build failed: %s", hipGetErrorString(e));
            return FGPU_DEVICE;
        }
        a->wordrow = wr;
    }
    *out = a->wordrow;
    return FGPU_OK;
}

static fgpu_info layer_of(fgpu_ctx* ctx, const fgpu_mat* a, Layer& l) {
    l.v = view_of(a);
    l.val = a->vals;
    l.nnz = (u32)a->nnz;
    return mat_wordrow(ctx, a, &l.wordrow);
}

struct Keep {  // keep bits + exclusive prefix of their per-word popcounts
    DevBuf<u64> kb;
    DevBuf<u32> ks;
    fgpu_info alloc(fgpu_ctx* ctx, u32 nnz) {
        u32 nwords = (nnz + 63) >> 6;
        FGPU_TRY(kb.alloc(ctx, (size_t)nwords + 1));
        FGPU_TRY(ks.alloc(ctx, (size_t)nwords + 1));
        // the word past the end is read by kept_before(nnz) when nnz is a multiple of 64
        FGPU_HIP(hipMemsetAsync(kb.p + nwords, 0, sizeof(u64), ctx->stream()));
        FGPU_HIP(hipMemsetAsync(ks.p + nwords, 0, sizeof(u32), ctx->stream()));
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
    // entries at or past the new dims are dropped only when the result is smaller than m (Matrix::resize shrinking)
    const bool clip = out_nrows < m->nrows || out_ncols < m->ncols || (dp && (out_nrows < dp->nrows || out_ncols < dp->ncols));
    Layer lm{}, lp{}, ld{};
    FGPU_TRY(layer_of(ctx, m, lm));
    if (has_dp) FGPU_TRY(layer_of(ctx, dp, lp));
    if (has_dm) FGPU_TRY(layer_of(ctx, dm, ld));
    // rows touched by a delta layer
    DevBuf<u32> rowbits, rowbits_dp;
    const u64 max_rows = m->nrows > out_nrows ? m->nrows : out_nrows;
    if (has_dp || has_dm) {
        const size_t nb = (size_t)(max_rows >> 5) + 2;
        FGPU_TRY(rowbits.alloc(ctx, nb));
        FGPU_HIP(hipMemsetAsync(rowbits.p, 0, nb * sizeof(u32), ctx->stream()));
        if (has_dp) {
            hipLaunchKernelGGL(rowbits_kernel, dim3(cdiv(dp->nvec, 256)), dim3(256), 0, ctx->stream(), lp.v, rowbits.p);
            if (has_dm && !(ctx->opt.merge_items && !clip && ctx->opt.merge_mode != 2 && !dm_masks_dp && !m->is_hyper())) {
                // the rows dp stores, on their own: only they need a cross-rank lookup when m is scattered by merge_scatter_kernel
                FGPU_TRY(rowbits_dp.alloc(ctx, nb));
                FGPU_HIP(hipMemsetAsync(rowbits_dp.p, 0, nb * sizeof(u32), ctx->stream()));
                hipLaunchKernelGGL(rowbits_kernel, dim3(cdiv(dp->nvec, 256)), dim3(256), 0, ctx->stream(), lp.v, rowbits_dp.p);
            }
        }
        if (has_dm)
            hipLaunchKernelGGL(rowbits_kernel, dim3(cdiv(dm->nvec, 256)), dim3(256), 0, ctx->stream(), ld.v, rowbits.p);
        FGPU_HIP(hipGetLastError());
    }
    Keep km, kp;
    FGPU_TRY(km.alloc(ctx, lm.nnz));
    FGPU_TRY(kp.alloc(ctx, has_dp ? lp.nnz : 0));
    if (lm.nnz && !clip && ctx->opt.merge_mode != 2) {
        const u32 nwords = (lm.nnz + 63) >> 6;
        hipLaunchKernelGGL(merge_markall_kernel, dim3(cdiv(nwords, 256)), dim3(256), 0, ctx->stream(), lm.nnz, km.kb.p);
        for (const Layer* d : {has_dm ? &ld : nullptr, has_dp ? &lp : nullptr}) {
            if (!d) continue;
            hipLaunchKernelGGL(merge_unmark_kernel, dim3(entry_grid(ctx, d->nnz)), dim3(256), 0, ctx->stream(), *d, lm, km.kb.p);
        }
        hipLaunchKernelGGL(merge_popc_kernel, dim3(cdiv(nwords, 256)), dim3(256), 0, ctx->stream(), (const u64*)km.kb.p, nwords,
                           km.ks.p);
        FGPU_HIP(hipGetLastError());
    } else if (lm.nnz) {
        hipLaunchKernelGGL(merge_mark_kernel<true>, dim3(entry_grid(ctx, lm.nnz)), dim3(256), 0, ctx->stream(), lm, lp,
                           ld, has_dp, has_dm, dm_masks_dp, (const u32*)rowbits.p, (u32)out_nrows, (u32)out_ncols,
                           km.kb.p, km.ks.p, clip);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(scan_u32(ctx, km.ks.p, km.ks.p, ((u64)(lm.nnz + 63) >> 6) + 1, nullptr));
    if (has_dp) {
        hipLaunchKernelGGL(merge_mark_kernel<false>, dim3(entry_grid(ctx, lp.nnz)), dim3(256), 0, ctx->stream(), lp, lm,
                           ld, lm.nnz != 0, has_dm, dm_masks_dp, (const u32*)rowbits.p, (u32)out_nrows,
                           (u32)out_ncols, kp.kb.p, kp.ks.p, clip);
        FGPU_HIP(hipGetLastError());
        FGPU_TRY(scan_u32(ctx, kp.ks.p, kp.ks.p, ((u64)(lp.nnz + 63) >> 6) + 1, nullptr));
    }
    DevBuf<u32> orp;
    FGPU_TRY(orp.alloc(ctx, out_nrows + 1));
    hipLaunchKernelGGL(merge_rowlen_kernel, dim3(cdiv(out_nrows + 1, 256)), dim3(256), 0, ctx->stream(), lm, lp, has_dp,
                       (const u32*)rowbits.p, (const u64*)km.kb.p, (const u32*)km.ks.p, (const u64*)kp.kb.p,
                       (const u32*)kp.ks.p, (u32)out_nrows, orp.p);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(scan_u32(ctx, orp.p, orp.p, out_nrows + 1, nullptr));
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, orp.p + out_nrows, &nnz));
    fgpu_mat* o = nullptr;
    FGPU_TRY(mat_alloc(ctx, &o, out_nrows, out_ncols, nnz, with_vals, 0, false));
    hipError_t e = hipMemcpyAsync(o->rowptr, orp.p, (out_nrows + 1) * sizeof(u32), hipMemcpyDeviceToDevice,
                                  ctx->stream());
    // shifted-copy scatter (above) for the common shape: same dims, a plain CSR base, every dp entry kept
    const bool by_items = ctx->opt.merge_items && !clip && ctx->opt.merge_mode != 2 && !dm_masks_dp && !m->is_hyper() && lm.nnz && nnz;
    DevBuf<u32> Q, ibase;
    if (e == hipSuccess && by_items) {
        const u32 nQ = has_dp ? lp.nnz : 0u;
        const u32 nitems = cdiv(lm.nnz, MS_ITEM);
        fgpu_info ai = Q.alloc(ctx, (size_t)nQ + 64);
        if (ai == FGPU_OK) ai = ibase.alloc(ctx, (size_t)nitems + 1);
        if (ai != FGPU_OK) { mat_release(o); return ai; }
        if (nQ) hipLaunchKernelGGL(merge_qpos_kernel, dim3(entry_grid(ctx, nQ)), dim3(256), 0, ctx->stream(), lp, lm, Q.p);
        hipLaunchKernelGGL(merge_ibase_kernel, dim3(cdiv(nitems, 256)), dim3(256), 0, ctx->stream(), (const u32*)Q.p, nQ, nitems, ibase.p);
        u32 grid = cdiv(nitems, 4);
        if (grid > (u32)ctx->cus * 32) grid = ctx->cus * 32;
        hipLaunchKernelGGL(merge_copy_items_kernel, dim3(grid), dim3(256), 0, ctx->stream(), lm, (const u64*)km.kb.p, (const u32*)km.ks.p,
                           (const u32*)Q.p, nQ, (const u32*)ibase.p, nitems, o->colidx, o->vals);
        if (nQ) hipLaunchKernelGGL(merge_dp_place_kernel, dim3(cdiv(nQ, 256)), dim3(256), 0, ctx->stream(), lp, (const u32*)Q.p,
                                   (const u64*)km.kb.p, (const u32*)km.ks.p, o->colidx, o->vals);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !by_items && lm.nnz && nnz) {
        hipLaunchKernelGGL(merge_scatter_kernel<true>, dim3(entry_grid(ctx, lm.nnz)), dim3(256), 0, ctx->stream(), lm,
                           lp, has_dp, (const u32*)rowbits.p, (const u64*)km.kb.p, (const u32*)km.ks.p,
                           (const u64*)kp.kb.p, (const u32*)kp.ks.p, (const u32*)o->rowptr, o->colidx, o->vals, clip,
                           (const u32*)rowbits_dp.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !by_items && has_dp && nnz) {
        hipLaunchKernelGGL(merge_scatter_kernel<false>, dim3(entry_grid(ctx, lp.nnz)), dim3(256), 0, ctx->stream(), lp,
                           lm, lm.nnz != 0, (const u32*)rowbits.p, (const u64*)kp.kb.p, (const u32*)kp.ks.p,
                           (const u64*)km.kb.p, (const u32*)km.ks.p, (const u32*)o->rowptr, o->colidx, o->vals, clip,
                           (const u32*)nullptr);
        e = hipGetLastError();
    }
    // the scratch buffers above go back to the pool when this returns: the kernels reading them must be done
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream());
    if (e != hipSuccess) {
        set_error("merge failed: %s", hipGetErrorString(e));
        mat_release(o);
        return FGPU_DEVICE;
    }
    // hub list / max degree are computed when a BFS plan first needs them (mat_ensure_finalized)
    *out = o;
    return FGPU_OK;
}

// ---------------------------------------------------------------------------------------
// value fills by search: the structure comes from the pattern builders, the values are
// placed by locating each source entry in the finished CSR.
// ---------------------------------------------------------------------------------------
// COO build with values: the LAST duplicate of a coordinate wins (a legal GxB_ANY_UINT64 choice,
// matrix.rs:1186-1210, made deterministic): win[pos] = max tuple index, then vals[pos] = in[win].
__global__ void coo_winner_kernel(const u32* __restrict__ rows, const u32* __restrict__ cols, u64 n,
                                  const u32* __restrict__ rp, const u32* __restrict__ col, u32* __restrict__ win) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u32 r = rows[i], c = cols[i];
        if (r == 0xFFFFFFFFu) continue;  // tuple dropped by the generator (mat_from_device_coo's ROW_INVALID)
        u32 p = lower_bound_col(col, rp[r], rp[r + 1], c);
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
    FGPU_TRY(mat_from_device_coo(ctx, &a, nrows, ncols, rows, cols, n));   // dense row pointers
    fgpu_info i = FGPU_OK;
    do {
        if ((i = ctx->dev_alloc((void**)&a->vals, (size_t)(a->nnz ? a->nnz : 1) * sizeof(u64))) != FGPU_OK) break;
        if (a->nnz == 0) break;
        DevBuf<u32> win;
        if ((i = win.alloc(ctx, a->nnz)) != FGPU_OK) break;
        hipError_t e = hipMemsetAsync(win.p, 0, a->nnz * sizeof(u32), ctx->stream());
        if (e == hipSuccess) {
            hipLaunchKernelGGL(coo_winner_kernel, dim3(ctx->cus * 16), dim3(256), 0, ctx->stream(), rows, cols, n,
                               (const u32*)a->rowptr, (const u32*)a->colidx, win.p);
            hipLaunchKernelGGL(coo_take_winner_kernel, dim3(cdiv(a->nnz, 256)), dim3(256), 0, ctx->stream(),
                               (const u32*)win.p, vals, (u32)a->nnz, a->vals);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream());  // `win` returns to the pool
        if (e != hipSuccess) {
            set_error("valued COO build failed: %s", hipGetErrorString(e));
            i = FGPU_DEVICE;
        }
    } while (0);
    if (i != FGPU_OK) { mat_release(a); return i; }
    *out = a;
    return FGPU_OK;
}

// transpose with values: structure from the pattern transpose (dense row pointers), then every entry
// (r, c, v) of `a` is dropped at (c, r) of the result.
__global__ __launch_bounds__(256) void transpose_vals_kernel(Layer a, const u32* __restrict__ trp,
                                                            const u32* __restrict__ tcol, u64* __restrict__ tvals) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (a.nnz + 63) >> 6;
    for (u32 w = wave; w < nwords; w += nwaves) {
        const u32 p = (w << 6) + lane;
        if (p >= a.nnz) continue;
        u32 i, r;
        lane_row(a, w, p, true, i, r);
        const u32 c = a.v.colidx[p];
        u32 q = lower_bound_col(tcol, trp[c], trp[c + 1], r);
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
        Layer la{};
        if ((i = layer_of(ctx, a, la)) != FGPU_OK) break;
        hipLaunchKernelGGL(transpose_vals_kernel, dim3(entry_grid(ctx, la.nnz)), dim3(256), 0, ctx->stream(), la,
                           (const u32*)t->rowptr, (const u32*)t->colidx, t->vals);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream());
        if (e != hipSuccess) {
            set_error("valued transpose failed: %s", hipGetErrorString(e));
            i = FGPU_DEVICE;
        }
    } while (0);
    if (i != FGPU_OK) { mat_release(t); return i; }
    *out = t;
    return FGPU_OK;
}

}  // namespace fgpu

extern "C" {

static fgpu_info mat_resize_impl(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t nrows, uint64_t ncols) {
    FGPU_REQUIRE(ctx && out && a, FGPU_NULL_POINTER, "fgpu_mat_resize: NULL argument");
    FGPU_REQUIRE(nrows < 0xFFFFFFFFull && ncols < 0xFFFFFFFFull, FGPU_INVALID,
                 "fgpu_mat_resize: dims exceed the 32-bit id space");
    return fgpu::mat_merge_entries(ctx, out, a, nullptr, nullptr, false, nrows, ncols, false);
}

static fgpu_info mat_merge_pattern_impl(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp,
                                 const fgpu_mat* dm, int dm_masks_dp) {
    FGPU_REQUIRE(ctx && out && m, FGPU_NULL_POINTER, "fgpu_mat_merge_pattern: NULL argument");
    FGPU_REQUIRE(!dp || (dp->nrows == m->nrows && dp->ncols == m->ncols), FGPU_DIM_MISMATCH,
                 "fgpu_mat_merge_pattern: dp dims differ from m");
    FGPU_REQUIRE(!dm || (dm->nrows == m->nrows && dm->ncols == m->ncols), FGPU_DIM_MISMATCH,
                 "fgpu_mat_merge_pattern: dm dims differ from m");
    return fgpu::mat_merge_entries(ctx, out, m, dp, dm, dm_masks_dp != 0, m->nrows, m->ncols, true);
}

}  // extern "C"

// Public producers of snapshots: the implementation above, then fgpu_ctx::publish().
extern "C" {

fgpu_info fgpu_mat_merge_pattern(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp,
                                 const fgpu_mat* dm, int dm_masks_dp) {
    fgpu_info i_ = mat_merge_pattern_impl(ctx, out, m, dp, dm, dm_masks_dp);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

fgpu_info fgpu_mat_resize(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t nrows, uint64_t ncols) {
    fgpu_info i_ = mat_resize_impl(ctx, out, a, nrows, ncols);
    if (i_ == FGPU_OK && ctx) i_ = ctx->publish();   // the new handle may go to another thread
    return i_;
}

}  // extern "C"
