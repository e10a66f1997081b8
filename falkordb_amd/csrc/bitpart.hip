// bitpart.hip — the dense COUNTING hop of the bit-parallel k-hop chain with the rows of X partitioned BY XCD.
//
// What bounds the plain dense pull (bitexpand.hip bp_pull_kernel<.., dense, count>), measured in round 5
// (tools/micro/gatherq.hip, profiles/r05a_gatherq*.{txt,json}): a gather that misses the XCD's L2 costs one 128-byte LINE of
// the Infinity-Cache / HBM path whatever it wanted from it (4 B, a 64-byte row, a 128-byte row: 1 TCC_EA0_RDREQ each), and
// that path serves ~57 G lines/s (MALL-resident) / ~49.5 G lines/s (HBM) — 7.3 / 6.3 TB/s — saturated from 16 wavefronts x 1
// gather per CU, from ~100 of the 256 CUs, with no second path (scalar loads: 25 G rows/s, LDS-DMA: the same queue).  The hop
// ran AT that limit (6.0 GB of lines per launch in 1.01 ms at RMAT-22): it is bound by the NUMBER OF MISSED LINES.  Every XCD
// gathers from the whole of X, so the eight 4 MiB L2s hold the same hottest rows (simulated LRU hit rate 0.35, measured
// traffic within 5 % of the simulation's).  Here workgroup b only gathers rows of partition b & 7 — workgroups are dealt to
// the XCDs round-robin — so X is cached ONCE across the chip (32 MiB instead of 4): simulated hit rate 0.75, X traffic 5.4 ->
// 2.1 GB at RMAT-22.  A row v of A' then receives a PARTIAL row from each of the (up to 8) partitions its in-neighbours fall
// into: partial rows are written once, compact and in order (~1 GB written + read at RMAT-22: sequential traffic, not
// gathers), and a fold kernel ORs them per vertex and counts / check-sums the complete row exactly where the plain pull did.
//
// Round 4's BpHotPlan was the same idea restricted to "hot" (v, u) pairs, a wavefront per (row, partition) segment of 4-8
// entries: item-bound.  Here the entries of a partition, sorted by row, are ONE STREAM: a (partition, row) pair with entries
// is a run of the stream (its first entry carries a flag in bit 31 of the column id), run k of the whole stream produces
// partial row k, and the pull is a segmented OR-reduction over chunks of the stream (<= XP_SPAN entries and <= XP_RUNS runs,
// cut once per snapshot) — a wavefront per chunk, 64 entries per trip (a coalesced 4-byte load per lane), QL consecutive
// entries per slot, their rows of X gathered QL at a time with the NEXT trip's gathers already in flight; a slot ORs the
// entries of one run in registers and flushes into the chunk's tile of runs in LDS (ds_or_b64) when the run changes; the
// tile leaves as whole rows when the chunk ends.  The chunk loop holds NO global store: on gfx9 stores share vmcnt with
// loads and may complete out of order with them, so one pending store makes hipcc wait for vmcnt(0) — the gathers just
// issued — at every use of a load (the first version wrote finished runs every trip: 723 us).
// No row is long or short: an in-hub with 20 K entries in a partition is a few dozen chunks of one run each, whose pieces
// meet in the (pre-zeroed) partial row through atomics — the only atomics of the kernel, at most two rows per chunk.
#include "bitexpand.hpp"

namespace fgpu {

// partition(u) = u / prange with prange = ceil(ncols / 8) rounded up to 16 rows (whole 128-byte lines from 8-byte rows up): eight
// CONTIGUOUS ranges of X.  (The first version took bits 4..6 of u — every 8th block of 16 rows: the rows of a partition then
// share three address bits, the L2's channel interleave uses address bits of that order, and a partition reached only a
// fraction of its XCD's 16 channels: measured hit rate ~0.45 where the LRU simulation of a whole L2 gives 0.75.)
constexpr u32 XP_FIRST = 0x80000000u;   // bit 31 of a packed entry: first entry of its (partition, row) run
constexpr u32 XP_SPAN = 768;       // a chunk spans < XP_SPAN of the key (entry index + XP_RUNW x run index) inside its partition:
constexpr u32 XP_RUNW = 8;         //   <= XP_SPAN entries and <= XP_RUNS runs (the rows of its LDS tile)
constexpr int XP_FOLD_THREADS = 512;   // 8 wavefronts share one copy of the checksum tables
constexpr u32 XP_RUNS = XP_SPAN / XP_RUNW;

struct BpXPlan {
    bool usable = false;
    u32 n = 0, ng = 0;             // rows of A' (destinations), 64-row groups
    u64 nentries = 0;
    u32 nprows = 0;                // partial rows = runs = non-empty (partition, row) pairs
    u32 nchunks = 0;
    u32* pcol = nullptr;           // the entries, partition-major, rows ascending inside a partition: XP_FIRST | column id
    u32 pstart[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // first entry of partition k
    u32 cbase[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};    // first chunk of partition k
    u32* pstart_dev = nullptr;     // [9] + cbase [9]
    u32* cstart = nullptr;         // first entry of the chunk
    u32* crun0 = nullptr;          // run (= partial row) of the chunk's first entry
    uint8_t* cshared = nullptr;    // 1: that run began in an earlier chunk (its pieces meet through atomics)
    u32* zrows = nullptr;          // the partial rows that receive pieces from several chunks (zeroed before every hop)
    u32 nzrows = 0;
    u64* ne = nullptr;             // [8][ng] bit r: (partition, row 64 g + r) has entries
    u32* pbase = nullptr;          // [8][ng] partial row of the group's first non-empty row
    // slot of X's row u in the state this plan gathers from (nullptr: slot = u).  Rows are ranked by how often they are gathered
    // (out-degree of u in m, descending) and dealt to the partitions in turn — slot = (rank % 8) x range + rank / 8 — so every
    // partition holds an eighth of the hot rows, packed hot-first: a 128-byte line then carries two rows (64-byte rows) of about
    // the same heat, where vertex order puts a hot row next to a random one — the L2's capacity for hot rows doubles
    // (simulated: hit rate 0.75 -> 0.85 at RMAT-22).  The hop that PRODUCES the state writes row v at slot perm[v] (bitexpand.hip).
    u32* perm = nullptr;
};
const u32* bp_xplan_perm(const BpXPlan* p) { return p ? p->perm : nullptr; }

void bp_xplan_release(fgpu_ctx* ctx, BpXPlan* p) {
    if (!p) return;
    if (ctx) {
        ctx->dev_free(p->pcol); ctx->dev_free(p->pstart_dev); ctx->dev_free(p->cstart); ctx->dev_free(p->crun0); ctx->dev_free(p->cshared); ctx->dev_free(p->zrows);
        ctx->dev_free(p->ne); ctx->dev_free(p->pbase); ctx->dev_free(p->perm);
    }
    delete p;
}

// ---- plan build (once per snapshot) ----------------------------------------------------------------------------------------
// The entries of A' arrive partitioned by the slot range of their column (partition_csr_entries, transpose.hip: one
// LDS-staged stable partition pass, (slot, row) pairs — the walk that did this with one atomic and one scattered 4-byte store
// per entry took 3.5 + 4.1 ms at RMAT-22, three times the transpose itself).  What is left: the flag on the first entry of
// every (partition, row) run, and where every run starts — off[x * n + v] for ALL rows v, those without entries included,
// which a position that opens a run fills for the rows since the previous run.
__global__ __launch_bounds__(256) void xp_runs_kernel(const uint2* __restrict__ pairs, const u32* __restrict__ pstart, u32 n, u32 nnz,
                                                      u32* __restrict__ pcol, u32* __restrict__ off) {
    __shared__ u32 s_ps[9];
    if (threadIdx.x < 9) s_ps[threadIdx.x] = pstart[threadIdx.x];
    __syncthreads();
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    for (u32 p0 = wave * 64; p0 < nnz; p0 += nwaves * 64) {
        const u32 p = p0 + lane;
        u32 g0 = 0, g1 = 0;                       // rows (g0 .. g1] of partition x start at p
        size_t base = 0;
        if (p < nnz) {
            u32 x = 0;
#pragma unroll
            for (u32 k = 1; k < 8; ++k) x += p >= s_ps[k] ? 1u : 0u;
            const uint2 e = pairs[p];
            const bool head = p == s_ps[x];
            const u32 prev = head ? 0xFFFFFFFFu : pairs[p - 1].y;
            const bool first = head || e.y != prev;
            pcol[p] = e.x | (first ? XP_FIRST : 0u);
            if (first) { g0 = prev + 1u; g1 = e.y + 1u; base = (size_t)x * n; }   // (prev + 1 wraps to 0 at the head)
        }
        for (int j = 0; j < 2 && g0 < g1; ++j, ++g0) off[base + g0] = p;          // the usual case: the row itself, one empty row
        u64 more = __ballot(g0 < g1);
        while (more) {                            // a long stretch of rows without entries: the whole wavefront fills it
            const int l = (int)__builtin_ctzll(more);
            more &= more - 1ull;
            const u32 a = (u32)__builtin_amdgcn_readlane((int)g0, l), b = (u32)__builtin_amdgcn_readlane((int)g1, l);
            const u32 at = (u32)__builtin_amdgcn_readlane((int)p, l);
            const u64 bs = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)base >> 32), l) << 32) |
                           (u64)(u32)__builtin_amdgcn_readlane((int)(u32)base, l);
            for (u32 r = a + lane; r < b; r += 64) off[bs + r] = at;
        }
    }
}
// the rows after a partition's last run (all of them when it has none) start where the partition ends
__global__ __launch_bounds__(256) void xp_run_tails_kernel(const uint2* __restrict__ pairs, const u32* __restrict__ pstart, u32 n,
                                                           u32* __restrict__ off) {
    const u32 x = blockIdx.y;
    const u32 b = pstart[x], e = pstart[x + 1];
    const u32 from = e > b ? pairs[e - 1].y + 1u : 0u;
    for (u32 r = from + blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) off[(size_t)x * n + r] = e;
    if (x == 7 && blockIdx.x == 0 && threadIdx.x == 0) off[(size_t)8 * n] = pstart[8];
}
__global__ void xp_nonempty_kernel(const u32* __restrict__ off, u64 total, u32* __restrict__ nz) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i <= total; i += (u64)gridDim.x * 256) nz[i] = (i < total && off[i + 1] != off[i]) ? 1u : 0u;
}
// a wavefront per (partition, group): the non-empty bitmap and the partial row of the group's first non-empty row
__global__ __launch_bounds__(256) void xp_groups_kernel(const u32* __restrict__ off, const u32* __restrict__ ridx, u32 n, u32 ng,
                                                        u64* __restrict__ ne, u32* __restrict__ pbase) {
    const u32 lane = lane_id();
    const u32 wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (u32 t = wave; t < 8 * ng; t += nwaves) {
        const u32 x = t / ng, g = t % ng;
        const u32 v = g * 64 + lane;
        const size_t i = (size_t)x * n + (v < n ? v : n - 1);
        const u64 nw = __ballot(v < n && off[i + 1] != off[i]);
        if (lane == 0) {
            ne[t] = nw;
            pbase[t] = ridx[(size_t)x * n + (size_t)g * 64];
        }
    }
}
// chunk table: a thread per (partition, row) run.  Inside partition x the key of an entry is (its index in the partition) +
// XP_RUNW x (its run's index in the partition); chunk id = key / XP_SPAN, non-decreasing along the stream and never skipping
// an id (the key grows by <= XP_RUNW + 1 per entry).  A run announces every chunk that BEGINS inside it (or on its first
// entry): the chunk's first entry, the run's index, and whether the run began before the chunk.
__global__ void xp_chunks_kernel(const u32* __restrict__ off, const u32* __restrict__ ridx, u32 n, const u32* __restrict__ pstart,
                                 u32* __restrict__ cstart, u32* __restrict__ crun0, uint8_t* __restrict__ cshared) {
    const u32* cbase = pstart + 9;
    const u32* rbase = pstart + 18;
    const u64 total = 8ull * n;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += (u64)gridDim.x * 256) {
        const u32 b = off[i], e = off[i + 1];
        if (b == e) continue;
        const u32 x = (u32)(i / n);
        const u32 ps = pstart[x];
        const u64 rr = ridx[i] - rbase[x];                   // run index inside the partition
        const u64 kb = (u64)(b - ps) + XP_RUNW * rr;         // key of the run's first entry
        const u64 ke = (u64)(e - 1 - ps) + XP_RUNW * rr;     // ... of its last
        const long long prev = b == ps ? -1ll : (long long)((kb - 1 - XP_RUNW) / XP_SPAN);   // chunk of the entry before the run
        long long c = (long long)(kb / XP_SPAN);
        if (c <= prev) c = prev + 1;
        for (; c <= (long long)(ke / XP_SPAN); ++c) {
            const u64 first_key = (u64)c * XP_SPAN;          // smallest key of chunk c
            const u32 st = first_key > kb ? ps + (u32)(first_key - XP_RUNW * rr) : b;
            cstart[cbase[x] + c] = st;
            crun0[cbase[x] + c] = ridx[i];
            cshared[cbase[x] + c] = st != b;
        }
    }
}

__global__ void xp_shared_rows_kernel(const u32* __restrict__ crun0, const uint8_t* __restrict__ cshared, u32 nchunks,
                                      u32* __restrict__ zrows, u32* __restrict__ count) {
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c < nchunks && cshared[c]) zrows[atomicAdd(count, 1u)] = crun0[c];   // (a run over several chunks is listed once per chunk: harmless)
}

// rows of X ranked by out-degree, descending (ties by id): key = 65535 - min(degree, 65535), value = the row; the stable
// counting sort of transpose.hip orders them, and rank r goes to slot (r % 8) * prange + r / 8 — the ranks dealt to the 8
// partitions.  (One atomic per row on a histogram of degrees, then one on its cursor, was what this used to be: the two or
// three buckets of the small degrees take millions of same-address atomics at ~5.6 ns each — 60 ms of the 63 ms a new
// snapshot of RMAT-22 cost before its first batch.)
__global__ void xp_deg_key_kernel(const u32* __restrict__ rowptr, u32 n, u32* __restrict__ key, u32* __restrict__ val) {
    for (u32 u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) {
        const u32 d = rowptr[u + 1] - rowptr[u];
        key[u] = 65535u - (d < 65535u ? d : 65535u);
        val[u] = u;
    }
}
__global__ void xp_rank_kernel(const u32* __restrict__ by_rank, u32 n, u32 prange, u32* __restrict__ perm) {
    for (u32 r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) perm[by_rank[r]] = (r & 7u) * prange + (r >> 3);
}

static fgpu_info bp_xplan_build(fgpu_ctx* ctx, const fgpu_mat* m, const fgpu_mat* t, const BpXPlan** out);
// The plan is an optional accelerator: when its build fails (its temporaries are large — 64 bytes per vertex and 8 per entry) the
// half-built plan's buffers are freed, it stays attached as "not usable" (no second attempt per snapshot), the error is dropped
// and the hop falls back to the plain pull.
fgpu_info bp_xplan(fgpu_ctx* ctx, const fgpu_mat* m, const fgpu_mat* t, const BpXPlan** out) {
    const fgpu_info i = bp_xplan_build(ctx, m, t, out);
    if (i == FGPU_OK) return FGPU_OK;
    *out = nullptr;
    if (t && t->bp_xplan && !t->bp_xplan->usable) {
        BpXPlan* p = t->bp_xplan;
        ctx->dev_free(p->pcol); ctx->dev_free(p->pstart_dev); ctx->dev_free(p->cstart); ctx->dev_free(p->crun0); ctx->dev_free(p->cshared);
        ctx->dev_free(p->zrows); ctx->dev_free(p->ne); ctx->dev_free(p->pbase); ctx->dev_free(p->perm);
        p->pcol = nullptr; p->pstart_dev = nullptr; p->cstart = nullptr; p->crun0 = nullptr; p->cshared = nullptr; p->zrows = nullptr;
        p->ne = nullptr; p->pbase = nullptr; p->perm = nullptr;
        (void)hipGetLastError();
        set_error("%s", "");
        return FGPU_OK;
    }
    return i;                                                // (failed before a plan was attached: an argument / state error)
}
static fgpu_info bp_xplan_build(fgpu_ctx* ctx, const fgpu_mat* m, const fgpu_mat* t, const BpXPlan** out) {
    *out = nullptr;
    if (!ctx->opt.expand_xcd || !t || t->is_hyper()) return FGPU_OK;
    if (t->ncols >= (1ull << 31) || t->nrows >= 0xFFFFFFC0ull || t->nnz < 4096 || t->nnz >= 0x7FFFFFFFull) return FGPU_OK;
    std::lock_guard<std::mutex> idx_guard(m->idx_mu);
    if (t->bp_xplan) { if (t->bp_xplan->usable) *out = t->bp_xplan; return FGPU_OK; }
    BpXPlan* xp = new (std::nothrow) BpXPlan();
    FGPU_REQUIRE(xp, FGPU_OOM, "out of host memory");
    t->bp_xplan = xp;                                          // (published as "not usable" until the build below completes)
    const u32 n = (u32)t->nrows, ng = (n + 63) / 64;
    xp->n = n; xp->ng = ng; xp->nentries = t->nnz;
    hipStream_t st = ctx->stream();
    const u64 total = 8ull * n;
    DevBuf<u32> off, ridx;
    FGPU_TRY(off.alloc(ctx, total + 1));
    FGPU_TRY(ridx.alloc(ctx, total + 1));
    const u32 prange = (u32)((((t->ncols + 7) / 8) + 15) & ~15ull);
    if (ctx->opt.expand_xcd_relabel && !m->is_hyper() && m->nrows == t->ncols && t->ncols % 128 == 0 && (u64)prange * 8 == t->ncols) {
        // rank the rows of X by out-degree (descending, ties by id) and deal the ranks to the partitions
        const u32 nu = (u32)t->ncols;
        DevBuf<u32> dkey, dval, by_rank, kptr;
        FGPU_TRY(dkey.alloc(ctx, nu));
        FGPU_TRY(dval.alloc(ctx, nu));
        FGPU_TRY(by_rank.alloc(ctx, nu));
        FGPU_TRY(kptr.alloc(ctx, 65536 + 1));
        FGPU_TRY(ctx->dev_alloc((void**)&xp->perm, (size_t)nu * sizeof(u32)));
        hipLaunchKernelGGL(xp_deg_key_kernel, dim3(ctx->cus * 8), dim3(256), 0, st, (const u32*)m->rowptr, nu, dkey.p, dval.p);
        FGPU_HIP(hipGetLastError());
        FGPU_TRY(sort_u32_pairs_by_key(ctx, dkey.p, dval.p, nu, 65536, by_rank.p, kptr.p));
        hipLaunchKernelGGL(xp_rank_kernel, dim3(ctx->cus * 8), dim3(256), 0, st, (const u32*)by_rank.p, nu, prange, xp->perm);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(ctx->dev_alloc((void**)&xp->pstart_dev, 27 * sizeof(u32)));
    FGPU_TRY(ctx->dev_alloc((void**)&xp->pcol, ((size_t)t->nnz + 64) * sizeof(u32)));
    {
        DevBuf<uint2> pairs;
        FGPU_TRY(pairs.alloc(ctx, t->nnz));
        FGPU_TRY(partition_csr_entries(ctx, t->colidx, t->rowptr, n, t->nnz, xp->perm, prange, 8, pairs.p, xp->pstart_dev));
        u32 rgrid = cdiv(t->nnz, 256 * 4);
        if (rgrid > (u32)ctx->cus * 16) rgrid = ctx->cus * 16;
        hipLaunchKernelGGL(xp_runs_kernel, dim3(rgrid), dim3(256), 0, st, (const uint2*)pairs.p, (const u32*)xp->pstart_dev, n, (u32)t->nnz,
                           xp->pcol, off.p);
        hipLaunchKernelGGL(xp_run_tails_kernel, dim3(64, 8), dim3(256), 0, st, (const uint2*)pairs.p, (const u32*)xp->pstart_dev, n, off.p);
        FGPU_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(xp_nonempty_kernel, dim3(ctx->cus * 16), dim3(256), 0, st, (const u32*)off.p, total, ridx.p);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(scan_u32(ctx, ridx.p, ridx.p, total + 1, nullptr));
    FGPU_TRY(read_u32(ctx, ridx.p + total, &xp->nprows));
    u32 hp[27];                                              // pstart[9] | cbase[9] | first run of partition k [9]
    FGPU_TRY(read_words(ctx, xp->pstart_dev, 9, hp));
    for (int k = 0; k <= 8; ++k) FGPU_TRY(read_u32(ctx, ridx.p + (size_t)k * n, &hp[18 + k]));
    hp[9] = 0;
    for (int k = 0; k < 8; ++k) {
        const u64 len = hp[k + 1] - hp[k], runs = hp[19 + k] - hp[18 + k];
        hp[10 + k] = hp[9 + k] + (len ? (u32)((len - 1 + (u64)XP_RUNW * (runs - 1)) / XP_SPAN) + 1u : 0u);
    }
    for (int k = 0; k <= 8; ++k) { xp->pstart[k] = hp[k]; xp->cbase[k] = hp[9 + k]; }
    xp->nchunks = xp->cbase[8];
    FGPU_TRY(ctx->h2d(xp->pstart_dev, hp, 27 * sizeof(u32)));
    FGPU_TRY(ctx->dev_alloc((void**)&xp->ne, (size_t)8 * ng * sizeof(u64)));
    FGPU_TRY(ctx->dev_alloc((void**)&xp->pbase, ((size_t)8 * ng + 1) * sizeof(u32)));
    u32 ggrid = cdiv((u64)8 * ng, 4);
    if (ggrid > (u32)ctx->cus * 16) ggrid = ctx->cus * 16;
    hipLaunchKernelGGL(xp_groups_kernel, dim3(ggrid), dim3(256), 0, st, (const u32*)off.p, (const u32*)ridx.p, n, ng, xp->ne, xp->pbase);
    FGPU_HIP(hipGetLastError());
    FGPU_TRY(ctx->dev_alloc((void**)&xp->cstart, ((size_t)xp->nchunks + 1) * sizeof(u32)));
    FGPU_TRY(ctx->dev_alloc((void**)&xp->crun0, ((size_t)xp->nchunks + 1) * sizeof(u32)));
    FGPU_TRY(ctx->dev_alloc((void**)&xp->cshared, (size_t)xp->nchunks + 8));
    FGPU_HIP(hipMemsetAsync(xp->cshared, 0, (size_t)xp->nchunks + 8, st));
    hipLaunchKernelGGL(xp_chunks_kernel, dim3(ctx->cus * 16), dim3(256), 0, st, (const u32*)off.p, (const u32*)ridx.p, n,
                       (const u32*)xp->pstart_dev, xp->cstart, xp->crun0, xp->cshared);
    FGPU_HIP(hipGetLastError());
    {   // the shared rows, compacted once (a few thousand among ~10^5 chunks)
        DevBuf<u32> zc;
        FGPU_TRY(zc.alloc(ctx, 1));
        FGPU_HIP(hipMemsetAsync(zc.p, 0, sizeof(u32), st));
        FGPU_TRY(ctx->dev_alloc((void**)&xp->zrows, ((size_t)xp->nchunks + 1) * sizeof(u32)));
        hipLaunchKernelGGL(xp_shared_rows_kernel, dim3(cdiv(xp->nchunks ? xp->nchunks : 1, 256)), dim3(256), 0, st, (const u32*)xp->crun0,
                           (const uint8_t*)xp->cshared, xp->nchunks, xp->zrows, zc.p);
        FGPU_HIP(hipGetLastError());
        FGPU_TRY(read_u32(ctx, zc.p, &xp->nzrows));
        // ... and the spare row after the last run: the all-zero partial row the fold reads for (row, partition) pairs without one
        FGPU_TRY(ctx->h2d(xp->zrows + xp->nzrows, &xp->nprows, sizeof(u32)));
        xp->nzrows += 1;
    }
    FGPU_HIP(hipStreamSynchronize(st));
    xp->usable = true;
    *out = xp;
    return FGPU_OK;
}

// ---- per hop ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 or4(uint4 a, uint4 b) { return make_uint4(a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w); }
__device__ __forceinline__ bool any4(uint4 a) { return (a.x | a.y | a.z | a.w) != 0u; }
__device__ __forceinline__ void atomic_or4(uint4* dst, uint4 v) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
    if (v.x | v.y) atomicOr(d, ((unsigned long long)v.y << 32) | v.x);
    if (v.z | v.w) atomicOr(d + 1, ((unsigned long long)v.w << 32) | v.z);
}

// streaming hints (option expand_nt; A/B): a partial row is written once and read once by the fold, an entry of the column-id stream is
// read once per hop — neither should displace the partition's hot rows of X from its XCD's L2
typedef unsigned int xp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 xp_load_nt(const uint4* p) {
    const xp_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const xp_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void xp_store_nt(uint4* p, uint4 v) {
    xp_u32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<xp_u32x4*>(p));
}

// partial rows that receive pieces from several chunks start from zero
__global__ __launch_bounds__(256) void xp_zero_rows_kernel(const u32* __restrict__ rows, u32 nrows, u32 ql, uint4* __restrict__ partial) {
    const u32 per = 256 / ql;
    for (u32 i = blockIdx.x * per + threadIdx.x / ql; i < nrows; i += gridDim.x * per)
        partial[(size_t)rows[i] * ql + threadIdx.x % ql] = make_uint4(0, 0, 0, 0);
}

// the stream pull: QL lanes of 16 bytes per row of X (ws = 2 QL words), 64 / QL slots per wavefront, QL consecutive entries
// per slot and trip.  WIDE = false: the state is at most 4 GiB, a row's byte offset fits 32 bits (one multiply-add per gather
// address instead of 64-bit arithmetic: the kernel issues ~100 instructions per 64 entries and is not far from VALU-bound).
template <int QL, bool WIDE, int NT>
__global__ __launch_bounds__(256) void xp_stream_kernel(const u32* __restrict__ pcol, const u32* __restrict__ pstart,
                                                        const u32* __restrict__ cstart, const u32* __restrict__ crun0,
                                                        const uint8_t* __restrict__ cshared, const uint4* __restrict__ x,
                                                        uint4* __restrict__ partial) {
    constexpr int SLOTS = 64 / QL;
    extern __shared__ uint4 s_tile[];                        // per wavefront XP_RUNS rows (the runs of one chunk) x QL quads
    const u32 lane = lane_id(), wl = lane % QL, slot = lane / QL, wib = threadIdx.x >> 6;
    uint4* tile = s_tile + (size_t)wib * XP_RUNS * QL;
    unsigned long long* tile64 = reinterpret_cast<unsigned long long*>(tile) + wl * 2;
    for (u32 i = lane; i < XP_RUNS * QL; i += 64) tile[i] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32 part = blockIdx.x & 7u;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 3) * 4 + wib));
    const u32 nwaves = (gridDim.x >> 3) * 4;
    const u32 pe = pstart[part + 1];
    const u32 c0 = pstart[9 + part], nch = pstart[10 + part] - c0;
    const u64 upto = (2ull << (slot * QL)) - 1ull;            // entries 0 .. the slot's first one
    const char* xb = reinterpret_cast<const char*>(x) + wl * 16;
    for (u32 j = wave; j < nch; j += nwaves) {
        const u32 B = (u32)__builtin_amdgcn_readfirstlane((int)cstart[c0 + j]);
        const u32 E = j + 1 < nch ? (u32)__builtin_amdgcn_readfirstlane((int)cstart[c0 + j + 1]) : pe;
        const u32 R0 = (u32)__builtin_amdgcn_readfirstlane((int)crun0[c0 + j]);
        const bool first_atomic = cshared[c0 + j] != 0;      // the run of the chunk's first entry began in an earlier chunk
        const bool last_atomic = j + 1 < nch && cshared[c0 + j + 1] != 0;   // ... and the run of its last entry goes on in the next
        u32 base = 0;                                        // runs of the chunk begun before the current trip (tile row of a carried run)
        // One trip: `mask` = the run starts among its 64 entries (lane = entry), `xv` = the rows of X of this slot's QL
        // consecutive entries.  Entries past the chunk's end are copies of its last entry (the loads clamp): no run starts on
        // them, so they OR the last run's own bits into it once more — no validity test anywhere.
        auto trip = [&](u64 mask, const uint4 (&xv)[QL]) {
            const u32 fb = (u32)(mask >> (slot * QL));       // bit k: entry k of this slot starts a run
            u32 row = base + (u32)__popcll(mask & upto);     // tile row (= run of the chunk) of the slot's first entry
            uint4 acc = xv[0];
#pragma unroll
            for (int k = 1; k < QL; ++k) {
                if ((fb >> k) & 1u) {
                    atomicOr(&tile64[(size_t)row * QL * 2], ((unsigned long long)acc.y << 32) | acc.x);
                    atomicOr(&tile64[(size_t)row * QL * 2 + 1], ((unsigned long long)acc.w << 32) | acc.z);
                    ++row;
                    acc = xv[k];
                } else {
                    acc = or4(acc, xv[k]);
                }
            }
            atomicOr(&tile64[(size_t)row * QL * 2], ((unsigned long long)acc.y << 32) | acc.x);
            atomicOr(&tile64[(size_t)row * QL * 2 + 1], ((unsigned long long)acc.w << 32) | acc.z);
            base += (u32)__popcll(mask);
        };
        // software pipeline, two trips per iteration (ping-pong registers: a rotation by moves would wait for the loads it
        // moves): the column words run TWO trips ahead of the trip being consumed and the gathers ONE, and every wait is for
        // loads OLDER than the ones that should stay in flight (vmcnt counts in issue order: a column word loaded after the
        // previous trip's gathers would drain them).  Loads past the chunk's end repeat its last entry (no branch around a
        // load); the extra round of gathers per chunk hits the row it has just gathered.
#define XP_COL(T) ((NT & 2) ? __builtin_nontemporal_load(&pcol[(T) + lane < E ? (T) + lane : E - 1]) : pcol[(T) + lane < E ? (T) + lane : E - 1])
#define XP_GATHER(XV, CW)                                                                                   \
        _Pragma("unroll") for (int k = 0; k < QL; ++k) {                                                    \
            const u32 uk = (u32)__shfl((int)(CW), (int)(slot * QL + k), 64) & ~XP_FIRST;                    \
            if (WIDE) XV[k] = *reinterpret_cast<const uint4*>(xb + (size_t)uk * (QL * 16));                 \
            else XV[k] = *reinterpret_cast<const uint4*>(xb + (u32)(uk * (u32)(QL * 16)));                  \
        }
        u32 cwE = XP_COL(B);
        u32 cwO = XP_COL(B + 64);
        uint4 xvE[QL], xvO[QL];
        XP_GATHER(xvE, cwE)
        for (u32 t = B; t < E; t += 128) {
            // run starts in entry order (lane = entry); the chunk's first entry never opens a NEW tile row
            const u64 maskE = __ballot(t + lane < E && (cwE & XP_FIRST) && !(t == B && lane == 0));
            cwE = XP_COL(t + 128);
            XP_GATHER(xvO, cwO)
            trip(maskE, xvE);
            // (no early exit when the chunk ends in an even trip: the odd one then runs on copies of the last entry — a branch
            // here makes hipcc merge the wait states of the two paths at the loop head into vmcnt(0))
            const u64 maskO = __ballot(t + 64 + lane < E && (cwO & XP_FIRST));
            cwO = XP_COL(t + 192);
            XP_GATHER(xvE, cwE)
            trip(maskO, xvO);
        }
#undef XP_COL
#undef XP_GATHER
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the chunk's runs leave the tile: run R0 + r, SLOTS rows per step; the first / last one through atomics when other chunks
        // hold pieces of it
        const u32 nr = base + 1;
        for (u32 r0 = 0; r0 < nr; r0 += SLOTS) {
            const u32 r = r0 + slot;
            if (r < nr) {
                const uint4 v4 = tile[r * QL + wl];
                uint4* dst = partial + (size_t)(R0 + r) * QL + wl;
                if ((r == 0 && first_atomic) || (r + 1 == nr && last_atomic)) atomic_or4(dst, v4);
                else if (NT & 1) xp_store_nt(dst, v4);
                else *dst = v4;
                tile[r * QL + wl] = make_uint4(0, 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// the fold: Y[v] = OR over the partitions of v's partial rows, then exactly what the plain pull does with a finished row —
// a touched row (a delta layer names it) goes to its slot of the side buffer, any other is counted (MODE 2: and its checksum
// terms summed through the nibble tables in LDS) and never written.  A wavefront per 64-row group, 64 / QL rows per step,
// the (up to) 8 partial rows of a vertex in flight together.
__device__ __forceinline__ u64 xp_uniform64(u64 v) {
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
}
template <int QL, int MODE, bool NT>
__global__ __launch_bounds__(XP_FOLD_THREADS) void xp_fold_kernel(const u64* __restrict__ ne, const u32* __restrict__ pbase, u32 ng,
                                                      const uint4* __restrict__ partial, BpFinal fin, uint4* __restrict__ side,
                                                      u32 zrow /* an all-zero partial row */) {
    constexpr int SLOTS = 64 / QL;
    extern __shared__ u64 s_tab[];
    if (MODE == 2) {
        for (u32 i = threadIdx.x; i < fin.w * 256; i += blockDim.x) s_tab[i] = fin.tab[i];
        __syncthreads();
    }
    const u32 lane = lane_id(), wl = lane % QL, slot = lane / QL;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const u32 nwaves = (gridDim.x * blockDim.x) >> 6;
    constexpr u32 STR = 2 * QL + 1;                          // stage row stride in words (odd: the transposed reads spread over the banks)
    u64* stg = s_tab + (MODE == 2 ? (size_t)fin.w * 256 : 0) + (size_t)(threadIdx.x >> 6) * (32 * STR);   // MODE 2: 32 staged rows a wavefront
    u64 f_cnt = 0, f_sum = 0;
    for (u32 g = wave; g < ng; g += nwaves) {
        u64 nw[8];
        u32 pb[8];
        u64 any = 0ull;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            nw[k] = xp_uniform64(ne[(size_t)k * ng + g]);   // (wave-uniform: kept in scalar registers whatever the loads were)
            pb[k] = (u32)__builtin_amdgcn_readfirstlane((int)pbase[(size_t)k * ng + g]);
            any |= nw[k];
        }
        if (!any) continue;
        const u64 tw = fin.tbits ? xp_uniform64(fin.tbits[g]) : 0ull;      // (clean layers: no touched rows, no bitmap)
        const u32 tp = fin.tbits ? (u32)__builtin_amdgcn_readfirstlane((int)fin.tpref[g]) : 0u;
        const u64 lb = fin.label ? xp_uniform64(fin.label[g]) : ~0ull;
        // Where the time goes (PMC, round 6, checksum on): not the partial rows — two steps of loads in flight changed nothing —
        // but LDS: 83 % of the LDS-active cycles were bank conflicts.  With a row's words spread over QL lanes, one look-up
        // instruction reads 2 QL different tables, and entry e of EVERY table sits in the same bank pair: 64 random 8-byte reads
        // over 16 bank pairs.  The look-ups now run after a half group (32 rows) is staged in LDS with a lane per ROW: an
        // instruction then reads ONE table per half wavefront (16 addresses, broadcast, no conflict).  The index arithmetic of a
        // step is scalar where it is the same for the step; a row without a piece in partition k loads the plan's zero row.
#pragma unroll 1
        for (u32 hf0 = 0; hf0 < 64; hf0 += (SLOTS >= 64 ? 64 : 32)) {        // (SLOTS = 64: one step is the whole group)
        u32 staged = 0;                                       // (wave-uniform) rows of this half staged for the look-ups
#pragma unroll 1
        for (u32 r0 = hf0; r0 < hf0 + 32 && r0 < 64; r0 += SLOTS) {
            if (((any >> r0) & (SLOTS == 64 ? ~0ull : ((1ull << (SLOTS % 64)) - 1ull))) == 0ull) continue;   // (wave-uniform)
            const u32 r = r0 + slot;
            const u64 below = (1ull << r) - 1ull;
            const u64 before = (1ull << r0) - 1ull;          // (wave-uniform)
            uint4 pv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                u32 idx;
                if (SLOTS <= 32) {
                    const u32 fld = (u32)(nw[k] >> r0) & (SLOTS >= 32 ? 0xFFFFFFFFu : ((1u << (SLOTS % 32)) - 1u));   // scalar
                    const u32 first = pb[k] + (u32)__popcll(nw[k] & before);                                         // scalar
                    idx = ((fld >> slot) & 1u) ? first + (u32)__popc(fld & ((1u << slot) - 1u)) : zrow;
                } else {
                    idx = ((nw[k] >> r) & 1ull) ? pb[k] + (u32)__popcll(nw[k] & below) : zrow;
                }
                pv[k] = NT ? xp_load_nt(&partial[(size_t)idx * QL + wl]) : partial[(size_t)idx * QL + wl];
            }
            uint4 a = or4(or4(or4(pv[0], pv[1]), or4(pv[2], pv[3])), or4(or4(pv[4], pv[5]), or4(pv[6], pv[7])));
            const bool touched = (tw >> r) & 1ull;
            const bool counted = !touched && ((lb >> r) & 1ull);
            if (touched && any4(a)) {
                const u32 sl = tp + (u32)__popcll(tw & below);
                side[(size_t)sl * QL + wl] = a;              // the only writer of this slot before the delta fix-ups
            }
            const u32 pc = (u32)(__popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w));
            f_cnt += counted ? (u64)pc : 0ull;
            if (MODE == 2) {
                // (an all-zero word looks up entry 0 of its tables — zero — so nothing needs a test: words at or past fin.w are
                // zero by construction and their tables, past the end of s_tab, are never multiplied in)
                const u64 w0 = counted ? (((u64)a.y << 32) | a.x) : 0ull, w1 = counted ? (((u64)a.w << 32) | a.z) : 0ull;
                if (QL >= 2) {
                    u64* srow = stg + (size_t)(r & 31u) * STR + 2 * wl;
                    srow[0] = w0;
                    srow[1] = w1;
                    staged |= (u32)((SLOTS >= 32 ? 0xFFFFFFFFull : ((1ull << (SLOTS % 32)) - 1ull)) << (r0 & 31u));
                } else {
                    const u32 k0 = 2 * wl < fin.w ? 2 * wl : 0u, k1 = 2 * wl + 1 < fin.w ? 2 * wl + 1 : 0u;
                    const u64* t0 = s_tab + (size_t)k0 * 256;
                    const u64* t1 = s_tab + (size_t)k1 * 256;
                    u64 rs = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) rs += t0[j * 16 + (u32)((w0 >> (4 * j)) & 15ull)];
#pragma unroll
                    for (int j = 0; j < 16; ++j) rs += t1[j * 16 + (u32)((w1 >> (4 * j)) & 15ull)];
                    f_sum += rs * cs_dest_hash(g * 64 + r);
                }
            }
        }
        if (MODE == 2 && QL >= 2 && staged) {
            // the half group's look-ups, a lane per row: lanes 0-31 take the first QL words of rows hf0 .. hf0 + 31, lanes 32-63
            // the other QL — every instruction reads one table per half wavefront
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const u32 row = lane & 31u, hw = lane >> 5;
            const bool on = (staged >> row) & 1u;
            u64 rs = 0;
#pragma unroll 2
            for (u32 kk = 0; kk < (u32)QL; ++kk) {
                const u32 k = hw * QL + kk;
                const u64 w = on ? stg[(size_t)row * STR + k] : 0ull;
                const u64* tk = s_tab + (size_t)(k < fin.w ? k : 0u) * 256;
#pragma unroll
                for (int j = 0; j < 16; ++j) rs += tk[j * 16 + (u32)((w >> (4 * j)) & 15ull)];
            }
            rs += (u64)__shfl_xor((long long)rs, 32, 64);                 // the two halves of a row
            if (hw == 0 && on) f_sum += rs * cs_dest_hash(g * 64 + hf0 + row);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // (read before the next half overwrites the stage)
            __builtin_amdgcn_wave_barrier();
        }
        }
    }
    bp_block_add2(f_cnt, MODE == 2 ? f_sum : 0ull, fin.acc);
}

fgpu_info bp_xpull_count(fgpu_ctx* ctx, const BpXPlan* xp, const fgpu_mat* t, const u64* x, u32 ws, int mode, const BpFinal& fin,
                         u64* side, size_t lds_tables, u64 xrows) {
    FGPU_REQUIRE(xp && xp->usable && ws >= 2 && ws <= 16 && (ws & (ws - 1)) == 0, FGPU_INVALID, "partitioned pull: bad plan / row width");
    FGPU_REQUIRE(mode == 1 || mode == 2, FGPU_INVALID, "partitioned pull: counting hops only");
    const u32 ql = ws / 2;
    hipStream_t st = ctx->stream();
    DevBuf<uint4> partial;
    FGPU_TRY(partial.alloc(ctx, ((size_t)xp->nprows + 1) * ql));
    if (xp->nzrows) {
        u32 zg = cdiv(xp->nzrows, 256 / ql);
        if (zg > (u32)ctx->cus * 4) zg = ctx->cus * 4;
        hipLaunchKernelGGL(xp_zero_rows_kernel, dim3(zg), dim3(256), 0, st, (const u32*)xp->zrows, xp->nzrows, ql, partial.p);
        FGPU_HIP(hipGetLastError());
    }
    {
        // algorithmic bytes = what the HOP needs (SURVEY.md §8d's pull row): the entries of A' and its row pointers once, every
        // non-zero row of X once.  The partial rows are NOT in it — they exist only because of the partition, are written here and
        // read straight back by the fold (counting them made the kernel look 2.5 x closer to the roofline than the hop is)
        ProfScope ps(ctx, "xp_stream_kernel", 4 * xp->nentries + 4 * ((u64)xp->n + 1) + xrows * 8 * ws);
        const size_t lds = (size_t)4 * XP_RUNS * ql * sizeof(uint4);
        u32 per_cu = (u32)((size_t)ctx->opt.lds_limit / lds);
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        u32 most = 0;
        for (int k = 0; k < 8; ++k) most = std::max(most, xp->cbase[k + 1] - xp->cbase[k]);
        u32 grid = (u32)ctx->cus * per_cu;
        const u32 need = 8 * cdiv(most ? most : 1, 4);
        if (grid > need) grid = need;
        grid = (grid + 7) & ~7u;
        const bool wide = (u64)xp->n * ws * 8 > (1ull << 32) || (u64)t->ncols * ws * 8 > (1ull << 32);
#define XP_PULL3(Q, W, N) hipLaunchKernelGGL((xp_stream_kernel<Q, W, N>), dim3(grid), dim3(256), lds, st, (const u32*)xp->pcol,  \
                                          (const u32*)xp->pstart_dev, (const u32*)xp->cstart, (const u32*)xp->crun0,           \
                                          (const uint8_t*)xp->cshared, (const uint4*)x, partial.p)
#define XP_PULL2(Q, W) do { switch (ctx->opt.expand_nt & 3) { case 1: XP_PULL3(Q, W, 1); break; case 2: XP_PULL3(Q, W, 2); break;  \
                                                              case 3: XP_PULL3(Q, W, 3); break; default: XP_PULL3(Q, W, 0); break; } } while (0)
#define XP_PULL(Q) do { if (wide) XP_PULL2(Q, true); else XP_PULL2(Q, false); } while (0)
        switch (ql) {
            case 1: XP_PULL(1); break;
            case 2: XP_PULL(2); break;
            case 4: XP_PULL(4); break;
            default: XP_PULL(8); break;
        }
#undef XP_PULL3
#undef XP_PULL2
#undef XP_PULL
        FGPU_HIP(hipGetLastError());
    }
    {
        // (no algorithmic bytes of its own: everything it reads is the stream kernel's intermediate — bench.py quotes the hop,
        // stream + fold, against the stream kernel's bytes)
        ProfScope ps(ctx, "xp_fold_kernel", 0);
        // (4 or 8 wavefronts sharing one copy of the checksum tables, 16 or 32 resident per CU: the same 214 / 148 us with and
        // without the checksum at RMAT-22 — the kernel is not short of wavefronts)
        const u32 fthreads = XP_FOLD_THREADS;
        u32 grid = cdiv(xp->ng, fthreads / 64);
        if (grid > (u32)ctx->cus * (2048u / fthreads)) grid = ctx->cus * (2048u / fthreads);
        // MODE 2: the checksum tables + a stage of 32 rows per wavefront (xp_fold_kernel)
        const size_t lds = mode == 2 ? lds_tables + (size_t)(fthreads / 64) * 32 * (2 * ql + 1) * sizeof(u64) : 0;
        FGPU_REQUIRE(lds <= (size_t)ctx->opt.lds_limit, FGPU_INVALID, "partitioned pull: the fold needs %zu B of LDS", lds);
#define XP_FOLD3(Q, M, N)                                                                                                        \
        do {                                                                                                                     \
            if (lds > 48 * 1024)                                                                                                 \
                FGPU_HIP(hipFuncSetAttribute((const void*)xp_fold_kernel<Q, M, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((xp_fold_kernel<Q, M, N>), dim3(grid), dim3(fthreads), lds, st, (const u64*)xp->ne, (const u32*)xp->pbase, xp->ng,  \
                               (const uint4*)partial.p, fin, (uint4*)side, xp->nprows);                                                                         \
        } while (0)
#define XP_FOLD2(Q, M) do { if (ctx->opt.expand_nt & 4) XP_FOLD3(Q, M, true); else XP_FOLD3(Q, M, false); } while (0)
#define XP_FOLD(Q) do { if (mode == 2) XP_FOLD2(Q, 2); else XP_FOLD2(Q, 1); } while (0)
        switch (ql) {
            case 1: XP_FOLD(1); break;
            case 2: XP_FOLD(2); break;
            case 4: XP_FOLD(4); break;
            default: XP_FOLD(8); break;
        }
#undef XP_FOLD
#undef XP_FOLD2
#undef XP_FOLD3
        FGPU_HIP(hipGetLastError());
    }
    (void)t;
    return FGPU_OK;
}

}  // namespace fgpu
