// transpose.hip — stable two-level counting sort of (key, value) pairs on the device, and the two builders that are
// nothing but that sort: the pattern transpose (GrB_transpose, Matrix::transpose matrix.rs:633-662; the cached
// `Tensor::matrix_t`, tensor.rs:814-816, 886-888) and the COO -> CSR build (Matrix::<bool>::build ->
// GxB_Matrix_build_Scalar, matrix.rs:1281-1303, duplicates collapsing as matrix.rs:1686-1695 pins).
//
// Why not atomics + a sort (the round-1 builder: one device-scope atomic per entry for the histogram, one for the
// scatter cursor, then a per-row sort of rows that arrive in random order): device-scope atomics run at ~26 G/s
// chip-wide here, so 2 x 67 M of them plus three sort passes took 13.3 ms for RMAT-22 — 1 % of HBM.  A sorted-unique
// CSR does not need sorting at all to be transposed: entries are already ordered by (row, col), so a STABLE
// partition by column leaves every column's rows ascending.  Keys are split in two digits:
//
//   pass 1  bucket = key >> wb            B = ceil(nkeys / 2^wb) <= 8192 buckets
//           count   : one workgroup per block of EB consecutive entries, LDS histogram over buckets
//           scan    : one flat exclusive scan of cnt[bucket][block] = the stable position of every (bucket, block) run
//           scatter : one WAVEFRONT per block walks its entries in order, 64 per trip; lanes that share a bucket find
//                     each other with log2(B) ballots, the lowest takes the run's slots from the LDS cursor
//                     (ds_add_rtn), everyone stores its packed (key, value) pair — stable by construction
//   pass 2  one workgroup per bucket (2^wb keys, counters in LDS): its four wavefronts count their quarter of the
//           bucket, a prefix over the keys gives every key's output range (written straight into the result's row
//           pointers) and every quarter's base, then each wavefront ranks its quarter exactly as in pass 1
//
// No global atomics, no sort; bytes: pass 1 reads the keys twice and writes 8 B per entry, pass 2 reads them twice
// and writes 4 B per entry — 28 B per entry against the 16 B a transpose must move.
#include "common.hpp"

namespace fgpu {

constexpr u32 KS_INVALID = 0xFFFFFFFFu;   // value marking a dropped tuple (self-loops of the R-MAT generator)
constexpr u32 KS_MAX_BUCKETS = 8192;      // pass-1 LDS cursors: 32 KiB
constexpr u32 KS_MAX_WB = 13;             // pass-2 LDS counters: 4 quarters x 2^13 x 4 B = 128 KiB

struct KsGeom {
    u32 wb;     // low-digit bits
    u32 B;      // buckets
    u32 bbits;  // ballots needed to tell buckets apart
    u32 EB;     // entries per pass-1 block
    u32 nblk;   // pass-1 blocks
};

static int g_ks_wb_override = 0;   // experiment knob (option "transpose_wb"): low-digit bits, 0 = pick
void ks_set_wb_override(int wb) { g_ks_wb_override = wb; }

static bool ks_geometry(u64 n, u64 nkeys, KsGeom& g) {
    u32 kb = 1;
    while (kb < 32 && (1ull << kb) < nkeys) ++kb;
    // split the key bits between the two digits so that neither LDS footprint starves its kernel of wavefronts:
    // pass 1 keeps B x 4 B per single-wavefront workgroup, pass 2 keeps 4 x 2^wb x 4 B per 4-wavefront workgroup
    // Sweep at 2^22 keys (RMAT-22, tools/transpose_wb_sweep.py; count / scatter / bucket ms): wb 10: .29 / 1.38 / .86;
    // 11: .18 / 1.46 / .99; 12: .13 / 1.01 / 1.23; 13: .11 / .66 / 2.26 — fewer pass-1 buckets mean longer runs per
    // bucket (the scattered 8 B pairs combine into lines before they leave L2 / MALL), more pass-2 keys mean LDS
    // counters that leave one workgroup per CU.  Smaller pass-1 blocks (4096, 2048 entries) changed nothing.
    u32 wb = kb > 22 ? kb - 12 : (kb > 12 ? 12 : (kb > 11 ? 11 : 8));   // 2^22 keys: 1024 x 4096; 2^24: 4096 x 4096; 2^26: 8192 x 8192
    if (wb < 8) wb = 8;                                // a pass-2 thread owns 2^wb / 256 keys
    if (kb > wb + 13) wb = kb - 13;
    if (g_ks_wb_override >= 8 && g_ks_wb_override <= (int)KS_MAX_WB && (u32)g_ks_wb_override + 13 >= kb) wb = (u32)g_ks_wb_override;
    if (wb > KS_MAX_WB) wb = KS_MAX_WB;
    const u64 B = (nkeys + (1ull << wb) - 1) >> wb;
    if (B > KS_MAX_BUCKETS) return false;      // > 2^26 keys: three digits would be needed
    g.wb = wb;
    g.B = (u32)(B ? B : 1);
    g.bbits = 0;
    while ((1u << g.bbits) < g.B) ++g.bbits;
    u64 eb = 16384;
    while ((n + eb - 1) / eb > 4096) eb <<= 1;  // the count matrix stays <= B x 4096

    g.EB = (u32)eb;
    g.nblk = (u32)((n + eb - 1) / eb);
    if (g.nblk == 0) g.nblk = 1;
    return true;
}

// row of CSR entry i (largest r with rowptr[r] <= i), searched in [lo, hi]
__device__ __forceinline__ u32 row_of_entry(const u32* __restrict__ rowptr, u32 lo, u32 hi, u32 i) {
    while (lo < hi) {
        const u32 mid = (lo + hi + 1) >> 1;
        if (rowptr[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// lanes of the wavefront holding the same `digit` (bits of it): the classic multi-split match
__device__ __forceinline__ u64 match_digit(u32 digit, u32 bits, bool active) {
    u64 peers = __ballot(active);
    for (u32 k = 0; k < bits; ++k) {
        const bool bit = (digit >> k) & 1u;
        const u64 b = __ballot(active && bit);
        peers &= bit ? b : ~b;
    }
    return peers;
}

// ---- pass 1 ------------------------------------------------------------------------------------------------
// IMPLICIT: the value of entry i is its CSR row (keys = colidx, `rowptr` given); otherwise val[i]
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void ks_count_kernel(const u32* __restrict__ key, const u32* __restrict__ val, u64 n,
                                                      KsGeom g, u32* __restrict__ cnt) {
    extern __shared__ u32 s_hist[];
    for (u32 b = threadIdx.x; b < g.B; b += 256) s_hist[b] = 0;
    __syncthreads();
    const u64 e0 = (u64)blockIdx.x * g.EB;
    const u64 e1 = e0 + g.EB < n ? e0 + g.EB : n;
    for (u64 i = e0 + threadIdx.x; i < e1; i += 256) {
        if (!IMPLICIT && val[i] == KS_INVALID) continue;
        atomicAdd(&s_hist[key[i] >> g.wb], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < g.B; b += 256) cnt[(size_t)b * g.nblk + blockIdx.x] = s_hist[b];
}

// rows of 64 consecutive CSR entries (lane j holds entry i0 + j, `on` = valid): the row boundaries from `rcur` on are
// fetched 64 at a time with ONE coalesced load and searched with shuffles — a per-lane binary search over the global
// row pointers is a chain of ~12 dependent loads, which is what a trip then costs (measured: 6-7 us per trip).
__device__ __forceinline__ u32 rows_of_trip(const u32* __restrict__ rowptr, u32 nrows, u32 rcur, u32 i, bool on, u32 lane) {
    u32 row = rcur, base = rcur;
    bool open = on;
    for (u32 guard = 0; __ballot(open) != 0ull; ++guard) {
        if (guard == 64) {   // thousands of empty rows inside one trip: finish with a plain search
            if (open) {
                u32 lo = base, hi = nrows - 1;
                while (lo < hi) {
                    const u32 mid = (lo + hi + 1) >> 1;
                    if (rowptr[mid] <= i) lo = mid; else hi = mid - 1;
                }
                row = lo;
            }
            break;
        }
        const u32 at = base + 1 + lane;
        const u32 bnd = rowptr[at < nrows ? at : nrows];   // rowptr[nrows] = nnz > every entry index
        u32 lo = 0, hi = 64;                                // #boundaries <= i among the 64 loaded (they ascend with the lane)
#pragma unroll
        for (int st = 0; st < 7; ++st) {                    // 65 possible answers: 7 halvings
            const u32 mid = (lo + hi) >> 1;
            const u32 v = (u32)__shfl((int)bnd, (int)(mid & 63u), 64);
            if (lo < hi) { if (v <= i) lo = mid + 1; else hi = mid; }
        }
        if (open && lo < 64) { row = base + lo; open = false; }
        base += 64;
    }
    return row;
}

template <bool IMPLICIT>
__global__ __launch_bounds__(64) void ks_scatter_kernel(const u32* __restrict__ key, const u32* __restrict__ val,
                                                       const u32* __restrict__ rowptr, u32 nrows, u64 n, KsGeom g,
                                                       const u32* __restrict__ pos, uint2* __restrict__ out) {
    extern __shared__ u32 s_cur[];
    const u32 lane = threadIdx.x;
    for (u32 b = lane; b < g.B; b += 64) s_cur[b] = pos[(size_t)b * g.nblk + blockIdx.x];
    __syncthreads();
    const u64 e0 = (u64)blockIdx.x * g.EB;
    const u64 e1 = e0 + g.EB < n ? e0 + g.EB : n;
    u32 rcur = 0;
    if (IMPLICIT) rcur = row_of_entry(rowptr, 0, nrows - 1, (u32)e0);
    constexpr int U = 4;   // trips whose loads are issued together
    for (u64 i0 = e0; i0 < e1; i0 += 64 * U) {
        u32 k[U], v[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u64 i = i0 + (u64)u * 64 + lane;
            on[u] = i < e1;
            k[u] = on[u] ? key[i] : 0u;
            v[u] = (!IMPLICIT && on[u]) ? val[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + (u64)u * 64 >= e1) break;   // wave-uniform
            const u64 i = i0 + (u64)u * 64 + lane;
            if (IMPLICIT) {
                v[u] = rows_of_trip(rowptr, nrows, rcur, (u32)i, on[u], lane);
                // rows only grow along the block: the next trip starts at the row of this trip's last entry
                const u64 m = __ballot(on[u]);
                rcur = (u32)__shfl((int)v[u], 63 - (int)__builtin_clzll(m), 64);
            } else {
                on[u] = on[u] && v[u] != KS_INVALID;
            }
            const u32 b = k[u] >> g.wb;
            const u64 peers = match_digit(b, g.bbits, on[u]);
            if (on[u]) {
                const u32 rank = (u32)__popcll(peers & ((1ull << lane) - 1ull));
                const u32 leader = (u32)__builtin_ctzll(peers);
                u32 base = 0;
                if (lane == leader) base = atomicAdd(&s_cur[b], (u32)__popcll(peers));
                base = (u32)__shfl((int)base, (int)leader, 64);
                out[base + rank] = make_uint2(k[u], v[u]);
            }
        }
    }
}

// ---- pass 2 ------------------------------------------------------------------------------------------------
// one workgroup (4 wavefronts) per bucket: counters s_cnt[q][key_low], q = the wavefront's quarter of the bucket
__global__ __launch_bounds__(256) void ks_bucket_kernel(const uint2* __restrict__ pairs, const u32* __restrict__ pos,
                                                       u32 n_valid, u64 nkeys, KsGeom g, u32* __restrict__ keyptr,
                                                       u32* __restrict__ out_val) {
    extern __shared__ u32 s_cnt[];
    __shared__ u32 s_wave[4];
    const u32 W = 1u << g.wb;
    const u32 b = blockIdx.x;
    const u32 s = pos[(size_t)b * g.nblk];
    const u32 e = (b + 1 < g.B) ? pos[(size_t)(b + 1) * g.nblk] : n_valid;
    const u32 lane = lane_id(), q = threadIdx.x >> 6;
    for (u32 i = threadIdx.x; i < 4 * W; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const u32 len = e - s;
    const u32 qlen = (len + 3) / 4;
    const u32 qs = s + q * qlen < e ? s + q * qlen : e;
    const u32 qe = qs + qlen < e ? qs + qlen : e;
    for (u32 i = qs + lane; i < qe; i += 64) atomicAdd(&s_cnt[q * W + (pairs[i].x & (W - 1))], 1u);
    __syncthreads();
    // exclusive prefix over the keys of the bucket; each thread owns W / 256 consecutive keys (W >= 256)
    const u32 per = W / 256;
    const u32 c0 = threadIdx.x * per;
    u32 mine = 0;
    for (u32 c = c0; c < c0 + per; ++c) mine += s_cnt[c] + s_cnt[W + c] + s_cnt[2 * W + c] + s_cnt[3 * W + c];
    // block exclusive scan of `mine`
    u32 inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = (u32)__shfl_up((int)inc, d, 64);
        if (lane >= (u32)d) inc += y;
    }
    if (lane == 63) s_wave[q] = inc;
    __syncthreads();
    u32 run = s + inc - mine;
    for (u32 w = 0; w < q; ++w) run += s_wave[w];
    const u64 kbase = (u64)b << g.wb;
    for (u32 c = c0; c < c0 + per; ++c) {
        if (kbase + c <= nkeys) keyptr[kbase + c] = run;   // entry nkeys (= n_valid) falls out of the last bucket
        u32 t0 = s_cnt[c], t1 = s_cnt[W + c], t2 = s_cnt[2 * W + c], t3 = s_cnt[3 * W + c];
        s_cnt[c] = run;
        s_cnt[W + c] = run + t0;
        s_cnt[2 * W + c] = run + t0 + t1;
        s_cnt[3 * W + c] = run + t0 + t1 + t2;
        run += t0 + t1 + t2 + t3;
    }
    __syncthreads();
    constexpr int U = 4;   // trips whose loads are issued together
    for (u32 i0 = qs; i0 < qe; i0 += 64 * U) {
        uint2 p[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32 i = i0 + u * 64 + lane;
            on[u] = i < qe;
            p[u] = on[u] ? pairs[i] : make_uint2(0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * 64 >= qe) break;   // wave-uniform
            const u32 c = p[u].x & (W - 1);
            const u64 peers = match_digit(c, g.wb, on[u]);
            if (on[u]) {
                const u32 rank = (u32)__popcll(peers & ((1ull << lane) - 1ull));
                const u32 leader = (u32)__builtin_ctzll(peers);
                u32 base = 0;
                if (lane == leader) base = atomicAdd(&s_cnt[q * W + c], (u32)__popcll(peers));
                base = (u32)__shfl((int)base, (int)leader, 64);
                out_val[base + rank] = p[u].y;
            }
        }
    }
}

// keyptr entries past the last bucket's keys (nkeys is not a multiple of 2^wb: none are missing; this only covers
// nkeys + 1 itself when nkeys is a multiple of 2^wb, which no bucket owns)
__global__ void ks_tail_kernel(u32* __restrict__ keyptr, u64 nkeys, u32 n_valid) { keyptr[nkeys] = n_valid; }

// Stable sort of n pairs by key (< nkeys): out_val = the values in (key, original position) order, keyptr[nkeys + 1] =
// the start of every key's run.  val == nullptr: the value of entry i is its row in the CSR `rowptr` (nrows rows).
// Pairs whose value is KS_INVALID are dropped.  *n_valid_out = pairs kept.  Returns FGPU_NO_VALUE when the key space
// is too wide for two digits (the caller falls back to the sorter).
fgpu_info sort_pairs_by_key(fgpu_ctx* ctx, const u32* key, const u32* val, const u32* rowptr, u32 nrows, u64 n,
                            u64 nkeys, u32* out_val, u32* keyptr, u32* n_valid_out) {
    KsGeom g;
    if (n == 0 || n >= 0xFFFFFFFFull || !ks_geometry(n, nkeys, g)) return FGPU_NO_VALUE;
    const bool implicit = val == nullptr;
    DevBuf<u32> cnt, pos, tot;
    DevBuf<uint2> pairs;
    const size_t ncnt = (size_t)g.B * g.nblk;
    FGPU_TRY(cnt.alloc(ctx, ncnt + 1));
    FGPU_TRY(pos.alloc(ctx, ncnt + 1));
    FGPU_TRY(tot.alloc(ctx, 1));
    FGPU_TRY(pairs.alloc(ctx, n));
    const size_t lds1 = (size_t)g.B * sizeof(u32);
    {
        ProfScope ps(ctx, "ks_count_kernel", 4 * n + 4 * ncnt);
        if (implicit)
            hipLaunchKernelGGL(ks_count_kernel<true>, dim3(g.nblk), dim3(256), lds1, ctx->stream(), key, val, n, g, cnt.p);
        else
            hipLaunchKernelGGL(ks_count_kernel<false>, dim3(g.nblk), dim3(256), lds1, ctx->stream(), key, val, n, g, cnt.p);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(scan_u32(ctx, cnt.p, pos.p, ncnt, tot.p));
    {
        ProfScope ps(ctx, "ks_scatter_kernel", (implicit ? 4 : 8) * n + 8 * n + 4 * ncnt);
        if (implicit)
            hipLaunchKernelGGL(ks_scatter_kernel<true>, dim3(g.nblk), dim3(64), lds1, ctx->stream(), key, val, rowptr,
                               nrows, n, g, (const u32*)pos.p, pairs.p);
        else
            hipLaunchKernelGGL(ks_scatter_kernel<false>, dim3(g.nblk), dim3(64), lds1, ctx->stream(), key, val, rowptr,
                               nrows, n, g, (const u32*)pos.p, pairs.p);
        FGPU_HIP(hipGetLastError());
    }
    u32 n_valid = (u32)n;   // implicit values are never dropped: no read-back, no host sync
    if (!implicit) FGPU_TRY(read_u32(ctx, tot.p, &n_valid));
    const size_t lds2 = (size_t)4 * (1u << g.wb) * sizeof(u32);
    if (lds2 > 48 * 1024)
        FGPU_HIP(hipFuncSetAttribute((const void*)ks_bucket_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    {
        ProfScope ps(ctx, "ks_bucket_kernel", 16 * (u64)n_valid + 4 * (u64)n_valid + 4 * (nkeys + 1));
        hipLaunchKernelGGL(ks_bucket_kernel, dim3(g.B), dim3(256), lds2, ctx->stream(), (const uint2*)pairs.p,
                           (const u32*)pos.p, n_valid, nkeys, g, keyptr, out_val);
        FGPU_HIP(hipGetLastError());
    }
    if ((nkeys & ((1ull << g.wb) - 1)) == 0) {
        hipLaunchKernelGGL(ks_tail_kernel, dim3(1), dim3(1), 0, ctx->stream(), keyptr, nkeys, n_valid);
        FGPU_HIP(hipGetLastError());
    }
    if (n_valid_out) *n_valid_out = n_valid;
    return FGPU_OK;
}

// ---- duplicate collapse of a sorted CSR (rows ascending, duplicates adjacent) --------------------------------------
__global__ __launch_bounds__(256) void dedup_flag_kernel(const u32* __restrict__ rowptr, u32 nrows,
                                                        const u32* __restrict__ col, u32 n, u32* __restrict__ keep) {
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        keep[i] = (i == 0 || col[i] != col[i - 1]) ? 1u : 0u;   // row starts are re-flagged below
    (void)rowptr; (void)nrows;
}
__global__ __launch_bounds__(256) void dedup_rowstart_kernel(const u32* __restrict__ rowptr, u32 nrows, u32 n,
                                                            u32* __restrict__ keep) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const u32 b = rowptr[r];
    if (b < rowptr[r + 1] && b < n) keep[b] = 1u;   // the first entry of a row is never a duplicate of the previous row's last
}
__global__ __launch_bounds__(256) void dedup_rowptr_kernel(const u32* __restrict__ rowptr, u32 nrows, u32 n,
                                                          const u32* __restrict__ newpos, u32 total,
                                                          u32* __restrict__ out_rowptr) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r > nrows) return;
    const u32 b = rowptr[r];
    out_rowptr[r] = b < n ? newpos[b] : total;
}
__global__ __launch_bounds__(256) void dedup_scatter_kernel(const u32* __restrict__ col, const u32* __restrict__ keep,
                                                           const u32* __restrict__ newpos, u32 n,
                                                           u32* __restrict__ out_col) {
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        if (keep[i]) out_col[newpos[i]] = col[i];
}

// pattern transpose of a non-hypersparse snapshot without a sort; FGPU_NO_VALUE = not applicable (caller falls back)
fgpu_info mat_transpose_counting(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a) {
    if (a->is_hyper() || a->nnz < 4096 || a->nrows == 0) return FGPU_NO_VALUE;
    KsGeom g;
    if (!ks_geometry(a->nnz, a->ncols, g)) return FGPU_NO_VALUE;
    fgpu_mat* t = nullptr;
    FGPU_TRY(mat_alloc(ctx, &t, a->ncols, a->nrows, a->nnz, false, 0, false));
    fgpu_info i = sort_pairs_by_key(ctx, a->colidx, nullptr, a->rowptr, (u32)a->nrows, a->nnz, a->ncols, t->colidx,
                                    t->rowptr, nullptr);
    // hub lists / max degree are built when a BFS plan, vxm or PageRank first asks (mat_ensure_finalized)
    if (i != FGPU_OK) { mat_release(t); return i; }
    *out = t;
    return FGPU_OK;
}

// device COO (rows may hold KS_INVALID = dropped tuple) -> CSR with duplicates collapsed, by two stable sorts:
// by column (values = rows), then by row of that column-major form (values = columns, implicit) — an LSD radix sort
// on (row, col) whose digit sorts are the counting sorts above.  FGPU_NO_VALUE = not applicable.
fgpu_info mat_from_device_coo_counting(fgpu_ctx* ctx, fgpu_mat** out, u64 nrows, u64 ncols, const u32* rows,
                                       const u32* cols, u64 n) {
    if (n < 4096 || n >= 0xFFFFFFFFull || nrows == 0 || ncols == 0) return FGPU_NO_VALUE;
    KsGeom g;
    if (!ks_geometry(n, ncols, g) || !ks_geometry(n, nrows, g)) return FGPU_NO_VALUE;
    DevBuf<u32> byc_row, colptr, byr_col, rowptr;
    FGPU_TRY(byc_row.alloc(ctx, n));
    FGPU_TRY(colptr.alloc(ctx, ncols + 1));
    u32 nv = 0;
    FGPU_TRY(sort_pairs_by_key(ctx, cols, rows, nullptr, 0, n, ncols, byc_row.p, colptr.p, &nv));
    if (nv == 0) return FGPU_NO_VALUE;   // nothing survived: let the generic path build the empty matrix
    FGPU_TRY(byr_col.alloc(ctx, nv));
    FGPU_TRY(rowptr.alloc(ctx, nrows + 1));
    // the column-major form is a CSR over `ncols` rows whose "column ids" are the original rows
    fgpu_info i = sort_pairs_by_key(ctx, byc_row.p, nullptr, colptr.p, (u32)ncols, nv, nrows, byr_col.p, rowptr.p, nullptr);
    if (i != FGPU_OK) return i;
    byc_row.release();
    DevBuf<u32> keep, newpos, tot;
    FGPU_TRY(keep.alloc(ctx, (size_t)nv + 1));
    FGPU_TRY(newpos.alloc(ctx, (size_t)nv + 1));
    FGPU_TRY(tot.alloc(ctx, 1));
    u32 grid = cdiv(nv, 256);
    if (grid > (u32)ctx->cus * 32) grid = ctx->cus * 32;
    hipLaunchKernelGGL(dedup_flag_kernel, dim3(grid), dim3(256), 0, ctx->stream(), (const u32*)rowptr.p, (u32)nrows,
                       (const u32*)byr_col.p, nv, keep.p);
    hipLaunchKernelGGL(dedup_rowstart_kernel, dim3(cdiv(nrows, 256)), dim3(256), 0, ctx->stream(), (const u32*)rowptr.p,
                       (u32)nrows, nv, keep.p);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(scan_u32(ctx, keep.p, newpos.p, nv, tot.p));
    u32 nnz = 0;
    FGPU_TRY(read_u32(ctx, tot.p, &nnz));
    fgpu_mat* m = nullptr;
    FGPU_TRY(mat_alloc(ctx, &m, nrows, ncols, nnz, false, 0, false));
    hipLaunchKernelGGL(dedup_rowptr_kernel, dim3(cdiv(nrows + 1, 256)), dim3(256), 0, ctx->stream(), (const u32*)rowptr.p,
                       (u32)nrows, nv, (const u32*)newpos.p, nnz, m->rowptr);
    hipLaunchKernelGGL(dedup_scatter_kernel, dim3(grid), dim3(256), 0, ctx->stream(), (const u32*)byr_col.p,
                       (const u32*)keep.p, (const u32*)newpos.p, nv, m->colidx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("COO build failed: %s", hipGetErrorString(e)); mat_release(m); return FGPU_DEVICE; }
    *out = m;
    return FGPU_OK;
}

}  // namespace fgpu
