// pagerank.hip — algo.pageRank's numeric core: LAGr_PageRank (LAGraph v1.x, called from
// graph/src/runtime/functions/algo_procedures.rs:741-752 with damping 0.85, tol 1e-4, itermax 100) over the
// adjacency matrix, the first floating-point semiring of the path (plus_second FP32).
//
// LAGraph is not vendored in the reference tree; the algorithm restated (the tests hold a numpy twin of it):
//     r = 1/n;  d = max(1/damping, out_degree / damping);  sinks = vertices without out-edges
//     repeat (iters < itermax and rdiff > tol):
//         teleport = (1 - damping)/n + (damping/n) * sum(r[sinks])
//         t = r;  w = t ./ d;  r = teleport + A' (+.second) w;  rdiff = sum |t - r|
// Vectors are FP32 like the reference's GrB_FP32 vectors; every SUM (a row of the SpMV, the sink mass, rdiff) is
// accumulated in FP64 and rounded to FP32 once.  The reference's FP32 sums run in a GraphBLAS-internal order, so its
// scores are defined only up to the rounding of that order (~ sqrt(deg) ulps on a hub row); the FP64 accumulation puts
// this engine at the centre of that cloud — within one FP32 rounding of the exact sum per iteration, whatever the
// order — which is what lets the parity test hold 1e-6 relative instead of a 2e-5 window.  Hub rows are reduced in two
// fixed-order stages (no float atomics between workgroups): the result is reproducible run to run.
//
// `active` (nullable bitmap) restricts the graph to the induced subgraph of the flagged vertices — what
// algo.pageRank does with a label that does not cover every node (build_compact_adj_from_tensors,
// algo_procedures.rs:725-733): n = |active|, edges with an inactive endpoint do not exist, inactive scores = 0.
//
// One iteration = three passes: (1) elementwise w = r/d + block partials of the sink mass, (2) the pull
// SpMV over CSR(A') — entry-parallel over the contiguous entry range of every 64-row word, per-row FP64 sums in LDS,
// rows >= HUB_DEG by the static hub chunk list: a partial per chunk, then one fixed-order sum per row — fused with
// the |t - r| partials, (3) fixed-order reductions of the partials.
// Bytes per iteration: 4 nnz (column ids) + gathers of w (16 MB at RMAT-22, L2 / MALL resident) + 6 n-vectors.
#include "common.hpp"

namespace fgpu {

__device__ __forceinline__ bool pr_active(const u64* __restrict__ act, u32 v) {
    return !act || ((act[v >> 6] >> (v & 63)) & 1ull);
}

// out-degree inside the active subgraph (only launched when `active` is given)
__global__ void pr_degree_kernel(CsrView a, const u64* __restrict__ act, u32 n, u32* __restrict__ deg) {
    const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    u32 d = 0;
    if (pr_active(act, v)) {
        u32 b, e;
        row_range(a, v, b, e);
        for (u32 i = b; i < e; ++i) d += pr_active(act, a.colidx[i]) ? 1u : 0u;
    }
    deg[v] = d;
}

__global__ void pr_init_kernel(CsrView a, const u64* __restrict__ act, const u32* __restrict__ deg_in, u32 n,
                               float r0, float damping, float* __restrict__ r, float* __restrict__ d,
                               unsigned char* __restrict__ sink) {
    const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const bool on = pr_active(act, v);
    u32 dg;
    if (deg_in) dg = deg_in[v];
    else { u32 b, e; row_range(a, v, b, e); dg = e - b; }
    r[v] = on ? r0 : 0.0f;
    const float dmin = 1.0f / damping;
    const float dv = (float)dg / damping;
    d[v] = dv > dmin ? dv : dmin;
    sink[v] = (on && dg == 0) ? 1 : 0;
}

// fixed-order tree: lane pairs by xor-shuffle, then the four wavefronts in order
__device__ __forceinline__ double block_sum_256(double x, double* s_red) {
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) x += __shfl_xor(x, k, 64);
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = x;
    __syncthreads();
    const double tot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    return tot;
}

// w = t ./ d (0 for inactive vertices) and the per-block sink mass
__global__ __launch_bounds__(256) void pr_prep_kernel(const float* __restrict__ t, const float* __restrict__ d,
                                                     const unsigned char* __restrict__ sink,
                                                     const u64* __restrict__ act, u32 n, float* __restrict__ w,
                                                     double* __restrict__ part, const int* __restrict__ stop) {
    __shared__ double s_red[4];
    if (*stop) return;   // converged earlier in this blind batch of iterations (fgpu_pagerank): the scores stay as they are
    const u32 v = blockIdx.x * 256 + threadIdx.x;
    double rs = 0.0;
    if (v < n) {
        const float tv = t[v];
        w[v] = pr_active(act, v) ? tv / d[v] : 0.0f;
        rs = sink[v] ? (double)tv : 0.0;
    }
    const double tot = block_sum_256(rs, s_red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// out[0] = base + scale * sum(part[0..np))  — one workgroup, fixed order
// `state` = {stop, iterations done}: the reduction that closes an iteration (`closing`) counts it and raises `stop` once the
// 1-norm of the change is within tol — the test LAGr_PageRank makes on the host, made here so that the host reads the
// state once per batch of iterations instead of synchronising after every one.
__global__ __launch_bounds__(256) void pr_reduce_kernel(const double* __restrict__ part, u32 np, float base, float scale,
                                                       float* __restrict__ out, int* __restrict__ state, int closing,
                                                       float tol) {
    __shared__ double s_red[4];
    if (state[0]) return;
    double x = 0.0;
    for (u32 i = threadIdx.x; i < np; i += 256) x += part[i];
    const double tot = block_sum_256(x, s_red);
    if (threadIdx.x == 0) {
        const float o = (float)((double)base + (double)scale * tot);
        out[0] = o;
        if (closing) {
            state[1] += 1;
            if (!(o > tol)) state[0] = 1;
        }
    }
}

// r[v] = teleport + sum_{u in in(v)} w[u]   (rows < HUB_DEG; hub rows get teleport only, the chunks add the rest).
// ENTRY-parallel like every kernel of this engine that meets R-MAT rows: the entries of a 64-row word are one
// contiguous range of CSR(A'); the wave walks it 256 entries per trip (a lane per entry, four trips' loads in
// flight), finds each entry's row among the word's 64 with a 6-step search over the row offsets in LDS, and adds
// w[col] into the row's LDS accumulator (ds_add_f32).  A lane-per-row / wave-per-long-row version of this kernel
// took 0.58 ms per pass at RMAT-22 whatever the locality of the gathers (column tiles of 2 MB changed nothing):
// it was a chain of dependent round trips per row, not a bandwidth problem.
__global__ __launch_bounds__(256) void pr_spmv_kernel(CsrView at, const u64* __restrict__ act, u32 n,
                                                     const float* __restrict__ w, const float* __restrict__ tele,
                                                     const float* __restrict__ t, float* __restrict__ r,
                                                     double* __restrict__ part, const int* __restrict__ stop) {
    __shared__ double s_red[4];
    if (*stop) return;
    __shared__ u32 s_off[4][65];     // exclusive prefix of the word's effective row lengths
    __shared__ u32 s_rb[4][64];      // first entry of each row
    __shared__ double s_acc[4][64];  // FP64 row sums (ds_add_f64): the order of the adds stops mattering at FP32
    const u32 lane = lane_id();
    const u32 wv = threadIdx.x >> 6;
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (n + 63) >> 6;
    const float tp = tele[0];
    const u32* __restrict__ col = at.colidx;
    u32* off = s_off[wv];
    u32* rbs = s_rb[wv];
    double* acc = s_acc[wv];
    double diff = 0.0;
    for (u32 g = wave; g < nwords; g += nwaves) {
        const u32 v = (g << 6) + lane;
        const u32 vc = v < n ? v : n - 1;
        const u32 rb = at.rowptr[vc];
        const u32 re = at.rowptr[vc + 1];
        const bool on = v < n && pr_active(act, v);
        const bool hub = re - rb >= HUB_DEG;
        const u32 deg = (on && !hub) ? re - rb : 0u;
        u32 inc = deg;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 y = __shfl_up(inc, d, 64);
            if (lane >= (u32)d) inc += y;
        }
        off[lane + 1] = inc;
        if (lane == 0) off[0] = 0;
        rbs[lane] = rb;
        acc[lane] = 0.0;
        const u32 total = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        for (u32 e0 = 0; e0 < total; e0 += 256) {
            u32 row[4], x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 e = e0 + 64 * k + lane;
                u32 lo = 0, hi = 64;               // largest lo with off[lo] <= e
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const u32 mid = (lo + hi) >> 1;
                    if (off[mid] <= e) lo = mid; else hi = mid;
                }
                row[k] = lo;
                x[k] = (e < total) ? col[rbs[lo] + (e - off[lo])] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x[k] != 0xFFFFFFFFu) atomicAdd(&acc[row[k]], (double)w[x[k]]);
        }
        const double sum = acc[lane];
        if (v < n) {
            const float rv = on ? (float)((double)tp + sum) : 0.0f;
            r[v] = rv;   // a hub row gets the teleport here; pr_hub_finish_kernel overwrites it with the full sum
            if (!hub) diff += fabs((double)t[v] - (double)rv);
        }
    }
    const double tot = block_sum_256(diff, s_red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// hub rows of A' (>= HUB_DEG in-neighbours), stage 1: one workgroup per chunk of the static list, the chunk's FP64
// partial goes to hpart[h] (no atomics: every chunk has its own slot)
__global__ __launch_bounds__(256) void pr_hub_kernel(const u32* __restrict__ hub, u32 n_hub, const u32* __restrict__ col,
                                                    const u64* __restrict__ act, const float* __restrict__ w,
                                                    double* __restrict__ hpart, const int* __restrict__ stop) {
    __shared__ double s_red[4];
    if (*stop) return;
    for (u32 h = blockIdx.x; h < n_hub; h += gridDim.x) {
        const u32 row = hub[3 * h], b = hub[3 * h + 1], e = hub[3 * h + 2];
        double s = 0.0;
        if (pr_active(act, row))   // block-uniform
            for (u32 i = b + threadIdx.x; i < e; i += 256) s += (double)w[col[i]];
        const double tot = block_sum_256(s, s_red);
        if (threadIdx.x == 0) hpart[h] = tot;
    }
}

// stage 2: a row's chunks sit next to each other in the list (hub_scan_kernel reserves them in one piece), in
// ascending entry order; the thread that meets a row's FIRST chunk adds the row's partials in that order, writes
// r[row] = teleport + sum and accounts |t - r|.  One workgroup, fixed assignment of rows to threads, fixed-order tree.
__global__ __launch_bounds__(256) void pr_hub_finish_kernel(const u32* __restrict__ hub, u32 n_hub,
                                                           const u32* __restrict__ rowptr, const u64* __restrict__ act,
                                                           const double* __restrict__ hpart, const float* __restrict__ tele,
                                                           const float* __restrict__ t, float* __restrict__ r,
                                                           double* __restrict__ part, const int* __restrict__ stop) {
    __shared__ double s_red[4];
    if (*stop) return;
    const float tp = tele[0];
    double x = 0.0;
    for (u32 h = threadIdx.x; h < n_hub; h += 256) {
        const u32 row = hub[3 * h];
        if (hub[3 * h + 1] != rowptr[row]) continue;        // not the row's first chunk
        // the row's chunks are consecutive list entries of one size (hub_scan_kernel, mat.hip): their number follows from the
        // row length, so the partials are added — in the same ascending order as ever — from independent loads (walking the
        // list entry by entry was a chain of dependent loads: 63 us for a 100-chunk hub row, a twelfth of an iteration)
        const u32 b0 = hub[3 * h + 1], re = rowptr[row + 1], cs = hub[3 * h + 2] - b0;
        u32 cnt = cs ? (re - b0 + cs - 1) / cs : 0u;
        if (cnt > n_hub - h) cnt = n_hub - h;
        double s = 0.0;
        for (u32 k = 0; k < cnt; ++k) s += hpart[h + k];
        const float rv = pr_active(act, row) ? (float)((double)tp + s) : 0.0f;
        r[row] = rv;
        x += fabs((double)t[row] - (double)rv);
    }
    const double tot = block_sum_256(x, s_red);
    if (threadIdx.x == 0) part[0] = tot;
}

// ---- the SpMV by column ranges, each cached by its own XCDs (round 4) ----------------------------------------------------------
// What bounds the pull is the random 4-byte gather of w[col]: 65 M of them per iteration at RMAT-22 from a 16 MB vector that
// every XCD's 4 MiB L2 caches a different quarter of — ~80-100 G gathers/s.  Workgroups are dealt to the XCDs round-robin, and
// the same gathers run at ~255 G/s when workgroup b only touches column range b mod 8 (tools/micro/xcdgather.hip: a 16.8 MB
// table, 2.1 MB per range — up to 4 MB per range at that rate, 8 MB at half of it).  So A' is split ONCE per snapshot into
// PR_NPARTS matrices by column range (rows keep their order inside a range: a row's entries are sorted, a range is a contiguous
// piece) — FOUR ranges, range k on XCDs k and k + 4: at RMAT-22 a quarter of w is the 4 MB one L2 still gathers at full rate,
// and half the partial sums and offsets of eight ranges (SpMV 0.456 -> 0.443 ms, the combine 0.066 -> 0.039; RMAT-24 equal) —
// stored range-major with one row-pointer array per range; per iteration workgroup (k, block) sums range k's entries of a block
// of PR_RB rows into FP64 accumulators in LDS — entry-parallel, the row of an entry by a search over the block's offsets in LDS;
// a wavefront whose 64 entries share one row (a hub row) adds them up with shuffles first — and stores the block's partial sums;
// a second kernel adds the partials of a row in range order, rounds to FP32 once, and accounts |t - r|.  No hub list, no
// hub passes.  (FP64 sums of FP32 terms: as before they are exact unless a row's terms span more than 2^29, so the order in which
// a range's terms meet does not show; across ranges the order is fixed.)
constexpr u32 PR_NPARTS = 4;      // a power of two that divides 8 (the XCDs)
constexpr u32 PR_PSHIFT = 2;      // log2(PR_NPARTS)
constexpr u32 PR_RB = 1024;       // rows per workgroup block
struct PrParts {
    u32* prp = nullptr;            // PR_NPARTS x (n + 1) offsets into pcol, range-major
    u32* pcol = nullptr;           // column ids, range-major
    u32 pw = 0;                    // columns per range
    u32 n = 0;
};
void pr_parts_release(fgpu_ctx* ctx, PrParts* p) {
    if (!p) return;
    if (ctx) { ctx->dev_free(p->prp); ctx->dev_free(p->pcol); }
    delete p;
}

// entries of row v in each column range (7 lower bounds in the sorted row)
__global__ void pr_part_count_kernel(CsrView at, u32 n, u32 pw, u32* __restrict__ cnt) {
    const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > n) return;
    u32 prev = 0, b = 0, e = 0;
    if (v < n) { b = at.rowptr[v]; e = at.rowptr[v + 1]; prev = b; }
    for (u32 p = 0; p < PR_NPARTS; ++p) {
        u32 hi = e;
        if (v < n && p + 1 < PR_NPARTS) {
            const u64 bound = (u64)(p + 1) * pw;              // first column of the next range
            u32 lo = prev;
            while (lo < hi) { const u32 mid = (lo + hi) >> 1; if ((u64)at.colidx[mid] < bound) lo = mid + 1; else hi = mid; }
            hi = lo;
        }
        cnt[(size_t)p * (n + 1) + v] = v < n ? hi - prev : 0u;
        prev = hi;
    }
}
// a wavefront per row: its entries to their ranges' arrays (a range's piece of the row is contiguous in both)
__global__ __launch_bounds__(256) void pr_part_fill_kernel(CsrView at, u32 n, u32 pw, const u32* __restrict__ prp, u32* __restrict__ pcol) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    for (u32 v = wave; v < n; v += nwaves) {
        const u32 b = at.rowptr[v], e = at.rowptr[v + 1];
        if (b == e) continue;
        // lane p < 8: where range p's piece of the row starts in the row (prefix of the lengths) and in pcol
        u32 len = 0, dst = 0;
        if (lane < PR_NPARTS) { dst = prp[(size_t)lane * (n + 1) + v]; len = prp[(size_t)lane * (n + 1) + v + 1] - dst; }
        u32 inc = len;
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) { const u32 y = (u32)__shfl_up((int)inc, d, 64); if (lane >= (u32)d) inc += y; }
        const u32 start = inc - len;                           // exclusive prefix (lanes 0..7)
        for (u32 q0 = b; q0 < e; q0 += 64) {                   // (the shuffles below read lanes 0..7: every lane stays in the loop)
            const u32 q = q0 + lane;
            const bool valid = q < e;
            const u32 c = valid ? at.colidx[q] : 0u;
            const u32 p = c / pw < PR_NPARTS ? c / pw : PR_NPARTS - 1;
            const u32 s0 = (u32)__shfl((int)start, (int)p, 64), d0 = (u32)__shfl((int)dst, (int)p, 64);
            if (valid) pcol[d0 + (q - b - s0)] = c;
        }
    }
}

static fgpu_info pr_parts_build(fgpu_ctx* ctx, const fgpu_mat* At, const PrParts** out) {
    std::lock_guard<std::mutex> idx_guard(At->idx_mu);
    if (At->pr_parts) { *out = At->pr_parts; return FGPU_OK; }
    const u32 n = (u32)At->nrows;
    PrParts* pp = new (std::nothrow) PrParts();
    FGPU_REQUIRE(pp, FGPU_OOM, "out of host memory");
    pp->n = n;
    pp->pw = (u32)(((u64)At->ncols + PR_NPARTS - 1) / PR_NPARTS);
    if (pp->pw == 0) pp->pw = 1;
    const size_t words = (size_t)PR_NPARTS * (n + 1);
    fgpu_info i = ctx->dev_alloc((void**)&pp->prp, (words + 1) * sizeof(u32));
    if (i == FGPU_OK) i = ctx->dev_alloc((void**)&pp->pcol, (size_t)(At->nnz ? At->nnz : 1) * sizeof(u32));
    if (i == FGPU_OK) {
        hipLaunchKernelGGL(pr_part_count_kernel, dim3(cdiv((u64)n + 1, 256)), dim3(256), 0, ctx->stream(), view_of(At), n, pp->pw, pp->prp);
        if (hipGetLastError() != hipSuccess) i = FGPU_DEVICE;
    }
    // one exclusive scan over the range-major counts IS the layout: range p's rows follow range p - 1's (the extra slot per
    // range holds 0, so prp[p][n] = prp[p + 1][0])
    if (i == FGPU_OK) i = scan_u32(ctx, pp->prp, pp->prp, words, nullptr);
    if (i == FGPU_OK && At->nnz) {
        u32 grid = cdiv(n, 4);
        if (grid > (u32)ctx->cus * 32) grid = ctx->cus * 32;
        hipLaunchKernelGGL(pr_part_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(At), n, pp->pw, (const u32*)pp->prp, pp->pcol);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream()) != hipSuccess) i = FGPU_DEVICE;
    }
    if (i != FGPU_OK) { pr_parts_release(ctx, pp); if (i == FGPU_DEVICE) set_error("pagerank: column-range layout build failed"); return i; }
    At->pr_parts = pp;
    *out = pp;
    return FGPU_OK;
}

// workgroup j: range k = j mod PR_NPARTS (dealt to XCDs k and k + 4), row block j / PR_NPARTS.  part[k][v] = sum over range k's
// entries of row v of w[col].
__global__ __launch_bounds__(256) void pr_part_spmv_kernel(const u32* __restrict__ prp, const u32* __restrict__ pcol, u32 n,
                                                          const float* __restrict__ w, double* __restrict__ part,
                                                          const int* __restrict__ stop) {
    __shared__ u32 s_off[PR_RB + 1];
    __shared__ double s_acc[PR_RB];
    if (*stop) return;
    const u32 k = blockIdx.x & (PR_NPARTS - 1u), blk = blockIdx.x >> PR_PSHIFT;
    const u32 v0 = blk * PR_RB;
    const u32 rows = n - v0 < PR_RB ? n - v0 : PR_RB;
    const u32* __restrict__ rp = prp + (size_t)k * (n + 1) + v0;
    // (the streams — offsets, column ids, partial sums — are loaded / stored non-temporally: the XCD's L2 is for range k of w)
    for (u32 i = threadIdx.x; i <= PR_RB; i += 256) s_off[i] = __builtin_nontemporal_load(&rp[i < rows ? i : rows]);
    for (u32 i = threadIdx.x; i < PR_RB; i += 256) s_acc[i] = 0.0;
    __syncthreads();
    const u32 e0 = s_off[0], e1 = s_off[rows];
    const u32 lane = lane_id();
    // a wavefront takes 64 consecutive entries per trip, FOUR trips' loads in flight (column ids, then the gathers): a block
    // holds ~2 K entries of its range, and one trip at a time made the kernel a chain of round trips (0.48 ms per pass)
    for (u32 eb = e0 + (threadIdx.x & ~63u); eb < e1; eb += 1024) {
        u32 c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 e = eb + 256 * j + lane;
            c[j] = e < e1 ? __builtin_nontemporal_load(&pcol[e]) : 0xFFFFFFFFu;
        }
        float wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = c[j] != 0xFFFFFFFFu ? w[c[j]] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 e = eb + 256 * j + lane;
            const bool on = e < e1;
            if (__ballot(on) == 0ull) break;                            // (wave-uniform: later trips are past the end too)
            u32 lo = 0, hi = rows;                                        // largest lo with s_off[lo] <= e (rows may be empty)
            if (on) {
                while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_off[mid] <= e) lo = mid; else hi = mid; }
            }
            double x = (double)wv[j];
            // a trip inside ONE row (hub rows: thousands of entries): 64 same-address LDS atomics would be served one by one
            const u32 r0 = (u32)__builtin_amdgcn_readfirstlane((int)lo);
            if (__ballot(on && lo != r0) == 0ull) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
                if (lane == 0) atomicAdd(&s_acc[r0], x);
            } else if (on) {
                atomicAdd(&s_acc[lo], x);
            }
        }
    }
    __syncthreads();
    double* __restrict__ dst = part + (size_t)k * n + v0;
    for (u32 i = threadIdx.x; i < rows; i += 256) __builtin_nontemporal_store(s_acc[i], &dst[i]);
}

// r[v] = teleport + the range partials in range order; |t - r| partials per workgroup.  The same pass prepares the NEXT
// iteration — w = r ./ d and the workgroup's share of the sink mass (what pr_prep_kernel does from t) — so an iteration of the
// range form is three launches (SpMV, this, one reduction of both partial arrays) instead of five.
__global__ __launch_bounds__(256) void pr_part_combine_kernel(const double* __restrict__ part, const u64* __restrict__ act, u32 n,
                                                             const float* __restrict__ tele, const float* __restrict__ t,
                                                             const float* __restrict__ d, const unsigned char* __restrict__ sink,
                                                             float* __restrict__ r, float* __restrict__ w,
                                                             double* __restrict__ dpart, double* __restrict__ spart,
                                                             const int* __restrict__ stop) {
    __shared__ double s_red[4];
    if (*stop) return;
    const float tp = tele[0];
    double diff = 0.0, rs = 0.0;
    for (u32 v = blockIdx.x * 256 + threadIdx.x; v < n; v += gridDim.x * 256) {
        double s = 0.0;
#pragma unroll
        for (u32 p = 0; p < PR_NPARTS; ++p) s += part[(size_t)p * n + v];
        const bool on = pr_active(act, v);
        const float rv = on ? (float)((double)tp + s) : 0.0f;
        r[v] = rv;
        diff += fabs((double)t[v] - (double)rv);
        w[v] = on ? rv / d[v] : 0.0f;
        rs += sink[v] ? (double)rv : 0.0;
    }
    const double tot = block_sum_256(diff, s_red);
    const double stot = block_sum_256(rs, s_red);
    if (threadIdx.x == 0) { dpart[blockIdx.x] = tot; spart[blockIdx.x] = stot; }
}
// closes an iteration of the range form: rdiff -> `stop` and the iteration count (as pr_reduce_kernel with `closing`), and the
// next iteration's teleport from the sink mass
__global__ __launch_bounds__(256) void pr_reduce2_kernel(const double* __restrict__ dpart, const double* __restrict__ spart, u32 np,
                                                        float teleport0, float damp_over_n, float* __restrict__ scal,
                                                        int* __restrict__ state, float tol) {
    __shared__ double s_red[4];
    if (state[0]) return;
    double x = 0.0, y = 0.0;
    for (u32 i = threadIdx.x; i < np; i += 256) { x += dpart[i]; y += spart[i]; }
    const double tx = block_sum_256(x, s_red);
    const double ty = block_sum_256(y, s_red);
    if (threadIdx.x == 0) {
        const float o = (float)(0.0 + 1.0 * tx);
        scal[1] = o;
        scal[0] = (float)((double)teleport0 + (double)damp_over_n * ty);
        state[1] += 1;
        if (!(o > tol)) state[0] = 1;
    }
}

}  // namespace fgpu

using namespace fgpu;

extern "C" fgpu_info fgpu_pagerank(fgpu_ctx* ctx, const fgpu_mat* A, const fgpu_mat* At, const uint64_t* active_bitmap,
                                   float damping, float tol, int32_t itermax, float* centrality, int32_t* iters) {
    return fgpu_pagerank_status(ctx, A, At, active_bitmap, damping, tol, itermax, centrality, iters, nullptr);
}

extern "C" fgpu_info fgpu_pagerank_status(fgpu_ctx* ctx, const fgpu_mat* A, const fgpu_mat* At, const uint64_t* active_bitmap,
                                          float damping, float tol, int32_t itermax, float* centrality, int32_t* iters,
                                          int32_t* converged) {
    FGPU_REQUIRE(ctx && A && centrality, FGPU_NULL_POINTER, "fgpu_pagerank: NULL argument");
    if (converged) *converged = 1;                // (the paths that run no iteration: nothing left to converge)
    FGPU_REQUIRE(A->nrows == A->ncols, FGPU_DIM_MISMATCH, "fgpu_pagerank: adjacency must be square");
    FGPU_REQUIRE(!At || (At->nrows == A->nrows && At->ncols == A->ncols), FGPU_DIM_MISMATCH,
                 "fgpu_pagerank: transpose has different dimensions");
    FGPU_REQUIRE(damping > 0.0f && damping <= 1.0f, FGPU_INVALID, "fgpu_pagerank: damping out of (0, 1]");
    const u32 n = (u32)A->nrows;
    if (iters) *iters = 0;
    if (n == 0) return FGPU_OK;
    if (A->nnz == 0) {
        // no edges: every vertex is a sink, the first iteration reproduces r = 1/n exactly (teleport = (1-d)/n + d/n)
        // and rdiff = 0 ends the loop — written out here so that empty / hypersparse-empty snapshots need no kernels
        u64 n_act = n;
        if (active_bitmap) {
            n_act = 0;
            for (u32 v = 0; v < n; ++v) n_act += (active_bitmap[v >> 6] >> (v & 63)) & 1ull;
        }
        for (u32 v = 0; v < n; ++v) {
            const bool on = !active_bitmap || ((active_bitmap[v >> 6] >> (v & 63)) & 1ull);
            centrality[v] = (on && n_act) ? 1.0f / (float)n_act : 0.0f;
        }
        if (iters) *iters = (n_act && itermax > 0 && tol < 1.0f) ? 1 : 0;
        if (converged) *converged = (itermax > 0 || !(tol < 1.0f)) ? 1 : 0;
        return FGPU_OK;
    }
    // dense row pointers are indexed directly below: hypersparse inputs are densified, a missing transpose is built
    fgpu_mat *dA = nullptr, *dAt = nullptr;
    fgpu_info info = FGPU_OK;
    if (A->is_hyper()) {
        info = mat_merge_entries(ctx, &dA, A, nullptr, nullptr, false, A->nrows, A->ncols, true);
        A = dA;
    }
    if (info == FGPU_OK && !At) {
        info = fgpu_mat_transpose(ctx, &dAt, A);
        At = dAt;
    }
    if (info == FGPU_OK && At->is_hyper()) {
        fgpu_mat* dense = nullptr;
        info = mat_merge_entries(ctx, &dense, At, nullptr, nullptr, false, At->nrows, At->ncols, true);
        if (dAt) mat_release(dAt);
        dAt = dense;
        At = dense;
    }
    auto run = [&]() -> fgpu_info {
        FGPU_TRY(mat_ensure_finalized(At));
        const u32 nb = cdiv(n, 256);
        const u32 grid = (u32)ctx->cus * 8 < cdiv(cdiv(n, 64), 4) ? (u32)ctx->cus * 8 : cdiv(cdiv(n, 64), 4);
        DevBuf<u64> act;
        DevBuf<u32> deg;
        DevBuf<float> r, t, w, d, scal;
        DevBuf<double> part, part2, hpart;
        DevBuf<unsigned char> sink;
        u64 n_act = n;
        if (active_bitmap) {
            const size_t words = ((size_t)n + 63) / 64;
            n_act = 0;
            for (size_t k = 0; k < words; ++k) {
                u64 x = active_bitmap[k];
                if (k == words - 1 && (n & 63)) x &= (1ull << (n & 63)) - 1ull;
                n_act += (u64)__builtin_popcountll(x);
            }
            FGPU_TRY(act.alloc(ctx, words));
            FGPU_TRY(ctx->h2d(act.p, active_bitmap, words * sizeof(u64)));
            FGPU_TRY(deg.alloc(ctx, n));
            hipLaunchKernelGGL(pr_degree_kernel, dim3(nb), dim3(256), 0, ctx->stream(), view_of(A), (const u64*)act.p, n,
                               deg.p);
            FGPU_HIP(hipGetLastError());
        }
        FGPU_TRY(r.alloc(ctx, n));
        FGPU_TRY(t.alloc(ctx, n));
        FGPU_TRY(w.alloc(ctx, n));
        FGPU_TRY(d.alloc(ctx, n));
        FGPU_TRY(sink.alloc(ctx, n));
        FGPU_TRY(part.alloc(ctx, nb));
        // the column-range form (PrParts above) when the score vector does not fit one XCD's L2 next to the stream
        const PrParts* parts = nullptr;
        if (ctx->opt.pagerank_parts == 2 || (ctx->opt.pagerank_parts == 1 && (u64)n * sizeof(float) > (2ull << 20)))
            FGPU_TRY(pr_parts_build(ctx, At, &parts));
        const u32 cgrid = parts ? ((u32)ctx->cus * 8 < nb ? (u32)ctx->cus * 8 : nb) : 0u;
        DevBuf<double> ppart;
        if (parts) FGPU_TRY(ppart.alloc(ctx, (size_t)PR_NPARTS * n));
        FGPU_TRY(part2.alloc(ctx, (size_t)(grid > cgrid ? grid : cgrid) + 1));
        FGPU_TRY(scal.alloc(ctx, 2));
        FGPU_TRY(hpart.alloc(ctx, (size_t)At->n_hub_chunks + 1));
        if (n_act == 0) {
            memset(centrality, 0, (size_t)n * sizeof(float));
            return FGPU_OK;
        }
        const float fn = (float)n_act;
        hipLaunchKernelGGL(pr_init_kernel, dim3(nb), dim3(256), 0, ctx->stream(), view_of(A), (const u64*)act.p,
                           (const u32*)deg.p, n, 1.0f / fn, damping, r.p, d.p, sink.p);
        FGPU_HIP(hipGetLastError());
        const float teleport0 = (1.0f - damping) / fn, damp_over_n = damping / fn;
        const CsrView vat = view_of(At);
        // Iterations are enqueued blind, PR_BATCH at a time; every kernel returns at once when the device-side `stop` is up,
        // and the closing reduction of an iteration counts it and raises `stop` on convergence (same test, same order of
        // operations as a host loop that reads rdiff after every iteration: the scores are bit-identical) — one
        // synchronisation per batch instead of one per iteration.
        constexpr int PR_BATCH = 4;
        DevBuf<int> state;
        FGPU_TRY(state.alloc(ctx, 2));
        FGPU_HIP(hipMemsetAsync(state.p, 0, 2 * sizeof(int), ctx->stream()));
        int it = 0;
        bool stopped = !(1.0f > tol);   // (the host loop started from rdiff = 1)
        float* rp = r.p;   // current scores
        float* tp = t.p;   // previous scores
        const bool timing = getenv("FGPU_PR_TIMING") != nullptr;
        hipEvent_t ev[6];
        float acc_ms[5] = {0, 0, 0, 0, 0};
        if (timing) for (auto& e : ev) (void)hipEventCreate(&e);
        if (parts && itermax > 0 && !stopped) {
            // the range form prepares iteration i + 1 inside iteration i's combine pass: only the first w / teleport come from here
            hipLaunchKernelGGL(pr_prep_kernel, dim3(nb), dim3(256), 0, ctx->stream(), (const float*)rp,
                               (const float*)d.p, (const unsigned char*)sink.p, (const u64*)act.p, n, w.p, part.p,
                               (const int*)state.p);
            hipLaunchKernelGGL(pr_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream(), (const double*)part.p, nb,
                               teleport0, damp_over_n, scal.p, state.p, 0, 0.0f);
            FGPU_HIP(hipGetLastError());
        }
        while (it < itermax && !stopped) {
            const int batch = timing ? 1 : (itermax - it < PR_BATCH ? itermax - it : PR_BATCH);
            for (int bi = 0; bi < batch; ++bi) {
                float* tmp = tp; tp = rp; rp = tmp;   // t = old r
                if (timing) (void)hipEventRecord(ev[0], ctx->stream());
                if (parts) {
                    if (timing) (void)hipEventRecord(ev[1], ctx->stream());
                    const u32 nblk = cdiv(n, PR_RB);
                    hipLaunchKernelGGL(pr_part_spmv_kernel, dim3(nblk * PR_NPARTS), dim3(256), 0, ctx->stream(), (const u32*)parts->prp,
                                       (const u32*)parts->pcol, n, (const float*)w.p, ppart.p, (const int*)state.p);
                    if (timing) (void)hipEventRecord(ev[2], ctx->stream());
                    hipLaunchKernelGGL(pr_part_combine_kernel, dim3(cgrid), dim3(256), 0, ctx->stream(), (const double*)ppart.p,
                                       (const u64*)act.p, n, (const float*)scal.p, (const float*)tp, (const float*)d.p,
                                       (const unsigned char*)sink.p, rp, w.p, part2.p, part.p, (const int*)state.p);
                    if (timing) (void)hipEventRecord(ev[3], ctx->stream());
                    hipLaunchKernelGGL(pr_reduce2_kernel, dim3(1), dim3(256), 0, ctx->stream(), (const double*)part2.p,
                                       (const double*)part.p, cgrid, teleport0, damp_over_n, scal.p, state.p, tol);
                    if (timing) (void)hipEventRecord(ev[4], ctx->stream());
                    FGPU_HIP(hipGetLastError());
                    continue;
                }
                hipLaunchKernelGGL(pr_prep_kernel, dim3(nb), dim3(256), 0, ctx->stream(), (const float*)tp,
                                   (const float*)d.p, (const unsigned char*)sink.p, (const u64*)act.p, n, w.p, part.p,
                                   (const int*)state.p);
                hipLaunchKernelGGL(pr_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream(), (const double*)part.p, nb,
                                   teleport0, damp_over_n, scal.p, state.p, 0, 0.0f);
                if (timing) (void)hipEventRecord(ev[1], ctx->stream());
                hipLaunchKernelGGL(pr_spmv_kernel, dim3(grid), dim3(256), 0, ctx->stream(), vat, (const u64*)act.p, n,
                                   (const float*)w.p, (const float*)scal.p, (const float*)tp, rp, part2.p + 1,
                                   (const int*)state.p);
                if (timing) (void)hipEventRecord(ev[2], ctx->stream());
                if (At->n_hub_chunks) {
                    const u32 hg = At->n_hub_chunks < (u32)ctx->cus * 8 ? At->n_hub_chunks : (u32)ctx->cus * 8;
                    hipLaunchKernelGGL(pr_hub_kernel, dim3(hg), dim3(256), 0, ctx->stream(), (const u32*)At->hub_chunks,
                                       At->n_hub_chunks, (const u32*)At->colidx, (const u64*)act.p, (const float*)w.p, hpart.p,
                                       (const int*)state.p);
                    hipLaunchKernelGGL(pr_hub_finish_kernel, dim3(1), dim3(256), 0, ctx->stream(), (const u32*)At->hub_chunks,
                                       At->n_hub_chunks, (const u32*)At->rowptr, (const u64*)act.p, (const double*)hpart.p,
                                       (const float*)scal.p, (const float*)tp, rp, part2.p, (const int*)state.p);
                } else {
                    FGPU_HIP(hipMemsetAsync(part2.p, 0, sizeof(double), ctx->stream()));
                }
                if (timing) (void)hipEventRecord(ev[3], ctx->stream());
                hipLaunchKernelGGL(pr_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream(), (const double*)part2.p, grid + 1,
                                   0.0f, 1.0f, scal.p + 1, state.p, 1, tol);
                if (timing) (void)hipEventRecord(ev[4], ctx->stream());
                FGPU_HIP(hipGetLastError());
            }
            int hstate[2] = {0, 0};
            if (timing) {   // (the events below must have completed)
                FGPU_HIP(hipMemcpyAsync(ctx->pinned(), state.p, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream()));
                FGPU_HIP(hipStreamSynchronize(ctx->stream()));
                memcpy(hstate, ctx->pinned(), sizeof(hstate));
            } else {        // one-thread publish kernel + a polled pinned line: 8 us instead of the runtime's 22 (ctx.hip read_words)
                FGPU_TRY(read_words(ctx, (const u32*)state.p, 2, (u32*)hstate));
            }
            stopped = hstate[0] != 0;
            it = hstate[1];
            if (timing) {
                for (int k = 0; k < 4; ++k) {
                    float ms = 0;
                    (void)hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
                    acc_ms[k] += ms;
                }
            }
        }
        // the scores of iteration `it` sit in t's buffer after an odd number of executed iterations, in r's after an even one
        rp = (it & 1) ? t.p : r.p;
        if (timing) {
            fprintf(stderr, "fgpu_pagerank timing over %d iterations (ms): prep+reduce %.3f  spmv %.3f  hubs %.3f  "
                            "final reduce %.3f  (n_hub_chunks %u)\n", it, acc_ms[0], acc_ms[1], acc_ms[2], acc_ms[3],
                    At->n_hub_chunks);
            for (auto& e : ev) (void)hipEventDestroy(e);
        }
        if (iters) *iters = it;
        if (converged) *converged = stopped ? 1 : 0;   // the last executed iteration moved the scores by no more than tol
        FGPU_TRY(ctx->d2h(centrality, rp, (size_t)n * sizeof(float)));
        FGPU_HIP(hipStreamSynchronize(ctx->stream()));
        return FGPU_OK;
    };
    if (info == FGPU_OK) info = run();
    if (dA) mat_release(dA);
    if (dAt) mat_release(dAt);
    return info;
}
