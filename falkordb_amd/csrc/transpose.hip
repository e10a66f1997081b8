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
//
// From 2^17 keys on the sort runs as THREE such levels with every scattered store staged through LDS (kp_* below,
// "the same stable sort with every scattered store staged through LDS"): 60 B per entry, but in runs of whole lines — 1.30 ms
// against 2.43 ms for RMAT-22.  The two-level form above stays for narrow key spaces and as transpose_mode 3.
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

// ---- the same stable sort with every scattered store staged through LDS (transpose_mode 2) -------------------------------
// What held the two-level form at 2.5 ms for RMAT-22 (67 M entries; 0.55 GB algorithmic) is the shape of its stores, not
// their bytes: a wavefront's 64 pairs go to ~60 different 128-byte lines, and a line that leaves L2 before its other
// entries arrive costs one trip of the L2 miss path per STORE — the same ~57 G lines/s that bounds the row gather of the
// k-hop (67 M stores / 57 G/s = 1.2 ms, measured 1.0 + 1.2 ms for the two scatters).  Runs of whole lines
// need  entries per block >= run length x buckets,  and a block has to fit LDS: 4096 entries and <= 512 buckets give
// runs of 8-32 pairs.  So the key is cut into THREE digits (8 + 7 + 7 bits at 2^22 keys) and every level is the same
// most-significant-digit partition:
//
//   segments   the buckets of the digits already sorted (level 1: the whole input), contiguous in the level's input
//   blocks     <= KP_EB consecutive entries of ONE segment; block table = prefix of ceil(len / KP_EB) over the segments
//   count      one workgroup per block: LDS histogram of the level's digit -> cnt[(segment, digit, block)] laid out
//              segment-major, digit-major inside a segment: ONE flat exclusive scan gives every (block, digit) run its
//              place, and the place of (segment, digit, block 0) is the start of the next level's segment
//   scatter    the workgroup re-reads its block into registers (16 entries per thread, wavefront q owns the q-th quarter
//              so that order = stability), ranks every entry inside the block (per-wavefront cursors, ballot match as
//              above), writes it to its slot of an LDS copy of the block sorted by digit, and then copies that out:
//              consecutive threads -> consecutive slots -> consecutive addresses of a run
//
// The last level writes the values alone, and the key pointers are the starts of the "segments" one level further down.
constexpr u32 KP_EB = 2048;
constexpr u32 KP_MAX_D = 512;
constexpr bool KP_AUTO = true;    // transpose_mode 0 picks the staged levels from 2^17 keys
struct KpLevel {
    u32 shift;   // digit = (key >> shift) & (D - 1)
    u32 dbits, D;
    u32 S;       // segments of this level = 2^(bits above the digit); segment of a key = key >> (shift + dbits)
    u32 div;     // != 0 (single-level partitions): digit = key / div instead
};
template <bool DIV>
__device__ __forceinline__ u32 kp_digit(const KpLevel& lv, u32 key) {
    return DIV ? key / lv.div : (key >> lv.shift) & (lv.D - 1u);
}

// start of every segment of the NEXT level (q = segment * D + digit of this level; S * D + 1 entries wanted) out of this
// level's positions, and how many blocks each takes.  `limit`: entries written (the key pointers stop at nkeys).
__global__ __launch_bounds__(256) void kp_seg_kernel(const u32* __restrict__ pos, const u32* __restrict__ blkstart, KpLevel lv, u64 limit,
                                                     u32* __restrict__ segstart, u32* __restrict__ nblk) {
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    if (q > limit) return;
    const u32 NB = blkstart[lv.S];
    auto start_of = [&](u64 x) -> u32 {
        if (x >= (u64)lv.S * lv.D) return pos[(size_t)lv.D * NB];
        const u32 sg = (u32)(x >> lv.dbits), d = (u32)x & (lv.D - 1u);
        const u32 b0 = blkstart[sg], nbs = blkstart[sg + 1] - b0;
        return pos[(size_t)lv.D * b0 + (size_t)d * nbs];
    };
    const u32 a = start_of(q);
    segstart[q] = a;
    if (nblk) nblk[q] = q < limit ? (start_of(q + 1) - a + KP_EB - 1) / KP_EB : 0u;
}
__global__ void kp_first_kernel(u32 n, u32* __restrict__ segstart, u32* __restrict__ blkstart) {
    segstart[0] = 0; segstart[1] = n;
    blkstart[0] = 0; blkstart[1] = (n + KP_EB - 1) / KP_EB;
}
// exclusive scan of a short array (the blocks per segment) in ONE launch: 1024 threads, a contiguous piece each
constexpr u64 KP_SMALL_SCAN = 1u << 18;
__global__ __launch_bounds__(1024) void kp_small_scan_kernel(const u32* __restrict__ in, u32 n, u32* __restrict__ out) {
    __shared__ u32 s_w[16];
    const u32 per = (n + 1023u) / 1024u;
    const u32 b = threadIdx.x * per, e = b + per < n ? b + per : n;
    u32 sum = 0;
    for (u32 i = b; i < e; ++i) sum += in[i];
    u32 inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 v = (u32)__shfl_up((int)inc, o, 64); if ((int)lane_id() >= o) inc += v; }
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    u32 run = inc - sum;
    for (u32 w = 0; w < (threadIdx.x >> 6); ++w) run += s_w[w];
    for (u32 i = b; i < e; ++i) { const u32 v = in[i]; out[i] = run; run += v; }
}

// every block of a level: its entries [e0, e1) and where its counters are (digit d at cbase + d * cstride); cstride = ~0 marks
// the unused tail of the table (the grid is sized by the bound n / KP_EB + S, the blocks that exist are only known here)
struct KpBlock { u32 e0, e1, cbase, cstride; };
__global__ __launch_bounds__(256) void kp_blk_table_kernel(const u32* __restrict__ segstart, const u32* __restrict__ blkstart, KpLevel lv,
                                                           u32 nb_max, uint4* __restrict__ desc, const u32* __restrict__ rowptr, u32 nrows,
                                                           uint2* __restrict__ brows, u32* __restrict__ cnt) {
    const u32 b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb_max) return;
    if (b == 0) cnt[(size_t)lv.D * nb_max] = 0u;  // the scan runs over the bound: its last entry is the total
    if (b >= blkstart[lv.S]) { desc[b] = make_uint4(0u, 0u, 0u, 0xFFFFFFFFu); return; }
    u32 lo = 0, hi = lv.S - 1;                    // the last segment whose first block is <= b
    while (lo < hi) {
        const u32 mid = (lo + hi + 1) >> 1;
        if (blkstart[mid] <= b) lo = mid; else hi = mid - 1;
    }
    const u32 b0 = blkstart[lo], nbs = blkstart[lo + 1] - b0, j = b - b0;
    const u32 e0 = segstart[lo] + j * KP_EB, end = segstart[lo + 1];
    const u32 e1 = e0 + KP_EB < end ? e0 + KP_EB : end;
    desc[b] = make_uint4(e0, e1, lv.D * b0 + j, nbs);
    // implicit values (level 1 of a transpose): the rows of the block's first and last entry, searched here for all blocks at
    // once (in the scatter kernel the two searches were ~4 us of dependent loads at the head of every block)
    if (rowptr) brows[b] = make_uint2(row_of_entry(rowptr, 0, nrows - 1, e0), row_of_entry(rowptr, 0, nrows - 1, e1 - 1u));
}
// Workgroups are dealt to the 8 XCDs round-robin: hardware workgroup h takes logical block (h % 8) * (grid / 8) + h / 8, so
// that ONE XCD walks a contiguous range of blocks in order — neighbouring blocks write neighbouring pieces of every digit's
// run, and the line two pieces share is completed in that XCD's L2 instead of leaving two L2s half written.  (grid % 8 == 0)
__device__ __forceinline__ bool kp_block(const uint4* __restrict__ desc, u32 h, u32 grid, KpBlock& k) {
    const u32 b = (h & 7u) * (grid >> 3) + (h >> 3);
    const uint4 d = desc[b];
    k.e0 = d.x; k.e1 = d.y; k.cbase = d.z; k.cstride = d.w;
    return d.w != 0xFFFFFFFFu;
}

// FIRST: keys / values in two arrays (values nullable = implicit rows, never dropped); otherwise packed pairs
template <bool FIRST, bool DIV>
__global__ __launch_bounds__(256) void kp_count_kernel(const u32* __restrict__ key1, const u32* __restrict__ val1, const uint2* __restrict__ in,
                                                      KpLevel lv, const uint4* __restrict__ desc, u32* __restrict__ cnt,
                                                      const u32* __restrict__ keymap /* FIRST, nullable: the key is keymap[key1[i]] */,
                                                      u32* __restrict__ mapped /* nullable: the mapped keys, kept for the scatter */) {
    __shared__ u32 s_hist[KP_MAX_D];
    KpBlock k;
    if (!kp_block(desc, blockIdx.x, gridDim.x, k)) {
        // counters of the blocks that exist fill [0, D * NB); the rest of the bound is zeroed by the blocks that do not
        const u32 b = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        for (u32 d = threadIdx.x; d < lv.D; d += 256) cnt[(size_t)lv.D * b + d] = 0u;
        return;
    }
    for (u32 d = threadIdx.x; d < lv.D; d += 256) s_hist[d] = 0;
    __syncthreads();
    constexpr int T = KP_EB / 256;                // all of a thread's loads in flight before the first is used
    u32 key[T];
    bool on[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const u32 i = k.e0 + t * 256 + threadIdx.x;
        on[t] = i < k.e1;
        if (FIRST) {
            key[t] = on[t] ? key1[i] : 0u;
            if (val1 && on[t] && val1[i] == KS_INVALID) on[t] = false;
        } else {
            key[t] = on[t] ? in[i].x : 0u;
        }
    }
    if (FIRST && keymap) {
        // (a gather over a table wider than one L2 rides the miss path, ~57 G lines/s: it is done once, the scatter reads `mapped`)
#pragma unroll
        for (int t = 0; t < T; ++t) key[t] = on[t] ? keymap[key[t]] : 0u;
        if (mapped) {
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (on[t]) mapped[k.e0 + t * 256 + threadIdx.x] = key[t];
        }
    }
    if (lv.D <= 16) {
        // a handful of counters (the 8-way partition of the k-hop plan): 64 lanes on <= 16 LDS addresses serialise — count by
        // ballots, lane d keeps digit d's sum, one atomic a wavefront.  (That pass is bound by its slot gather all the same.)
        const u32 lane = lane_id();
        u32 acc = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const u32 dg = kp_digit<DIV>(lv, key[t]);
            for (u32 d = 0; d < lv.D; ++d) {
                const u32 c = (u32)__popcll(__ballot(on[t] && dg == d));
                if (lane == d) acc += c;
            }
        }
        if (lane < lv.D && acc) atomicAdd(&s_hist[lane], acc);
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (on[t]) atomicAdd(&s_hist[kp_digit<DIV>(lv, key[t])], 1u);
    }
    __syncthreads();
    for (u32 d = threadIdx.x; d < lv.D; d += 256) cnt[(size_t)k.cbase + (size_t)d * k.cstride] = s_hist[d];
}

template <bool FIRST, bool IMPLICIT, bool LAST, bool DIV>
__global__ __launch_bounds__(256) void kp_scatter_kernel(const u32* __restrict__ key1, const u32* __restrict__ val1,
                                                        const u32* __restrict__ rowptr, u32 nrows, const uint2* __restrict__ in,
                                                        KpLevel lv, const uint4* __restrict__ desc, const uint2* __restrict__ brows,
                                                        const u32* __restrict__ pos, uint2* __restrict__ out_pairs, u32* __restrict__ out_val,
                                                        const u32* __restrict__ keymap /* FIRST, nullable */) {
    extern __shared__ __attribute__((aligned(16))) u32 s_kp[];   // (64-bit LDS atomics below: the base must not sit at 4 mod 8)
    __shared__ u32 s_wave[4], s_total;
    const u32 D = lv.D;
    uint2* stage = (uint2*)s_kp;                  // KP_EB pairs
    u32* cur = s_kp + 2 * KP_EB;                  // [4][D]: per-wavefront counters, then cursors
    u32* lstart = cur + 4 * D;                    // [D]: first slot of a digit in the staged block
    u32* gpos = lstart + D;                       // [D]: where that run goes
    KpBlock k;
    if (!kp_block(desc, blockIdx.x, gridDim.x, k)) return;   // (uniform over the workgroup)
    const u32 lane = lane_id(), q = threadIdx.x >> 6;
    for (u32 i = threadIdx.x; i < 4 * D; i += 256) cur[i] = 0;
    // where this block's run of every digit goes: asked for now, used after the counting sweep (thread c owns digits c * per ..)
    const u32 per = D >= 256 ? D / 256 : 1u;
    const u32 c0 = threadIdx.x * per;
    u32 gp[KP_MAX_D / 256];
#pragma unroll
    for (u32 c = 0; c < KP_MAX_D / 256; ++c) gp[c] = (c < per && c0 + c < D) ? pos[(size_t)k.cbase + (size_t)(c0 + c) * k.cstride] : 0u;
    __syncthreads();
    constexpr int T = KP_EB / 256;                // trips of a wavefront
    const u32 qs = k.e0 + q * (KP_EB / 4);
    u32 key[T], val[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const u32 i = qs + t * 64 + lane;
        const bool on = i < k.e1;
        if (FIRST) {
            key[t] = on ? key1[i] : 0u;
            val[t] = (!IMPLICIT && on) ? val1[i] : 0u;
        } else {
            const uint2 p = on ? in[i] : make_uint2(0u, 0u);
            key[t] = p.x;
            val[t] = p.y;
        }
    }
    if (FIRST && keymap) {                        // (after all the key loads are out: T gathers in flight, not one at a time)
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (qs + t * 64 + lane < k.e1) key[t] = keymap[key[t]];
    }
    if (FIRST && IMPLICIT) {
        // rows of the block's entries, the other way round: every row that STARTS inside the block marks its first entry in
        // an LDS copy of the block (the last of several empty rows wins: it owns the entry), a running maximum then gives
        // every entry its row.  (Entry -> row by a window search per trip, rows_of_trip above, is a chain of 16 dependent
        // loads per wavefront: ~24 us of the 31 us a block took.)
        u32* mark = s_kp;                         // KP_EB words of the staging area, free until the ranks are known
        __shared__ u32 s_wmax[4];
        const uint2 ends = brows[(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)];
        for (u32 i = threadIdx.x; i < KP_EB; i += 256) mark[i] = 0;
        __syncthreads();
        const u32 rlo = ends.x, rhi = ends.y;
        for (u32 r = rlo + 1u + threadIdx.x; r <= rhi; r += 256) atomicMax(&mark[rowptr[r] - k.e0], r);   // (e0 < rowptr[r] < e1)
        __syncthreads();
        u32 carry = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            u32 x = mark[q * (KP_EB / 4) + t * 64 + lane];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u32 y = (u32)__shfl_up((int)x, d, 64);
                if (lane >= (u32)d && y > x) x = y;
            }
            if (carry > x) x = carry;
            carry = (u32)__builtin_amdgcn_readlane((int)x, 63);
            val[t] = x;
        }
        if (lane == 0) s_wmax[q] = carry;
        __syncthreads();
        u32 before = rlo;
        for (u32 w = 0; w < q; ++w) before = s_wmax[w] > before ? s_wmax[w] : before;
#pragma unroll
        for (int t = 0; t < T; ++t) val[t] = val[t] > before ? val[t] : before;
        __syncthreads();                          // the marks are read: the area is the staging buffer again
    }
    // count AND rank in one sweep: the wavefront walks its quarter in order; the lanes of a trip that hold the same digit find
    // each other through LDS — every lane ORs its bit into the wavefront's 64-bit word of that digit and reads the word back
    // (a wavefront's LDS instructions execute in order: the read sees the whole trip) — its rank is the popcount below it, the
    // group's place among the entries of (wavefront, digit) is the counter before the lowest lane adds the group's size.  (The
    // ballot match of the two-level form costs ~12 VALU per key BIT per trip, ~85 of the ~125 instructions a trip took: with 28
    // wavefronts a CU the kernel was bound by instruction issue, not by memory.)
    u64* seen = reinterpret_cast<u64*>(s_kp) + (size_t)q * D;      // [4][D] words in the staging area, free until the ranks are known
    for (u32 i = threadIdx.x; i < 4 * D; i += 256) reinterpret_cast<u64*>(s_kp)[i] = 0ull;
    __syncthreads();
    u32 live = 0;                                  // bit t: this lane's entry of trip t takes part
    u32 loc[T / 2];                                // place inside (wavefront, digit), two 16-bit fields a word
#pragma unroll
    for (int t = 0; t < T / 2; ++t) loc[t] = 0;
    const u64 mybit = 1ull << lane;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (qs + t * 64 >= k.e1) break;           // wave-uniform
        const u32 i = qs + t * 64 + lane;
        const bool on = i < k.e1 && !(FIRST && !IMPLICIT && val[t] == KS_INVALID);
        const u32 d = kp_digit<DIV>(lv, key[t]);
        if (on) atomicOr((unsigned long long*)&seen[d], (unsigned long long)mybit);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (on) {
            const u64 peers = seen[d];
            const u32 base = cur[q * D + d];
            live |= 1u << t;
            const u32 rank = (u32)__popcll(peers & (mybit - 1ull));
            loc[t >> 1] |= (base + rank) << ((t & 1) * 16);
            if ((peers & (mybit - 1ull)) == 0ull) {          // the lowest lane of the group tidies up for the next trip
                seen[d] = 0ull;
                cur[q * D + d] = base + (u32)__popcll(peers);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // slots: digits ascending, wavefronts ascending inside a digit; thread c owns digits c * per .. (per = D / 256, at least 1)
        u32 mine = 0;
        if (c0 < D)
            for (u32 c = c0; c < c0 + per; ++c) mine += cur[c] + cur[D + c] + cur[2 * D + c] + cur[3 * D + c];
        u32 inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 y = (u32)__shfl_up((int)inc, d, 64);
            if (lane >= (u32)d) inc += y;
        }
        if (lane == 63) s_wave[q] = inc;
        __syncthreads();
        u32 run = inc - mine;
        for (u32 w = 0; w < q; ++w) run += s_wave[w];
#pragma unroll
        for (u32 ci = 0; ci < KP_MAX_D / 256; ++ci) {
            const u32 c = c0 + ci;
            if (ci < per && c < D) {
                const u32 t0 = cur[c], t1 = cur[D + c], t2 = cur[2 * D + c], t3 = cur[3 * D + c];
                lstart[c] = run;
                gpos[c] = gp[ci];
                cur[c] = run;
                cur[D + c] = run + t0;
                cur[2 * D + c] = run + t0 + t1;
                cur[3 * D + c] = run + t0 + t1 + t2;
                run += t0 + t1 + t2 + t3;
            }
        }
        if (threadIdx.x == 255) s_total = run;     // (threads past the last digit carry the total along)
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (qs + t * 64 >= k.e1) break;           // wave-uniform
        if ((live >> t) & 1u) {
            const u32 d = kp_digit<DIV>(lv, key[t]);
            stage[cur[q * D + d] + ((loc[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu)] = make_uint2(key[t], val[t]);
        }
    }
    __syncthreads();
    const u32 total = s_total;
    for (u32 slot = threadIdx.x; slot < total; slot += 256) {
        const uint2 p = stage[slot];
        const u32 d = kp_digit<DIV>(lv, p.x);
        const u32 dst = gpos[d] + (slot - lstart[d]);
        if (LAST) out_val[dst] = p.y;
        else out_pairs[dst] = p;
    }
}

// digit widths, most significant first: as few levels as 9 bits a level allow, at least ... balanced
static int kp_widths(u64 nkeys, u32* w) {
    u32 kb = 1;
    while (kb < 32 && (1ull << kb) < nkeys) ++kb;
    int L = (int)((kb + 8) / 9);
    if (L < 1) L = 1;
    if (kb >= 17 && L < 3) L = 3;                 // (runs of whole lines need <= 512 buckets a level; 3 levels from 2^17 keys)
    if ((u32)L > kb) L = (int)kb;
    // the wider digits last: a level's runs are KP_EB / D entries long, and the last level moves 4-byte values in segments
    // that mostly fit one block (written as one contiguous piece whatever D is)
    for (int l = 0; l < L; ++l) w[l] = kb / L + ((u32)(L - 1 - l) < kb % L ? 1u : 0u);
    return L;
}

fgpu_info sort_pairs_by_key_staged(fgpu_ctx* ctx, const u32* key, const u32* val, const u32* rowptr, u32 nrows, u64 n,
                                   u64 nkeys, u32* out_val, u32* keyptr, u32* n_valid_out) {
    if (n == 0 || n >= 0xFFFFFFFFull - KP_EB || nkeys == 0 || nkeys > (1ull << 31)) return FGPU_NO_VALUE;
    const bool implicit = val == nullptr;
    u32 w[8];
    const int L = kp_widths(nkeys, w);
    u32 kb = 0;
    for (int l = 0; l < L; ++l) kb += w[l];
    hipStream_t st = ctx->stream();
    DevBuf<uint2> bufA, bufB;
    if (L >= 2) FGPU_TRY(bufA.alloc(ctx, n));
    if (L >= 3) FGPU_TRY(bufB.alloc(ctx, n));
    DevBuf<u32> segstart, nblk, blkstart, cnt, pos, segnext;
    DevBuf<uint4> desc;
    DevBuf<uint2> brows;
    // upper bounds: a level with S segments has at most n / KP_EB + S blocks
    u64 S_max = 1;
    {
        u32 done = 0;
        for (int l = 0; l + 1 < L; ++l) { done += w[l]; S_max = 1ull << done; }
    }
    const u64 nb_top = n / KP_EB + 9 + S_max;
    u32 dmax = 0;
    for (int l = 0; l < L; ++l) dmax = std::max(dmax, 1u << w[l]);
    FGPU_TRY(segstart.alloc(ctx, S_max + 2));
    FGPU_TRY(segnext.alloc(ctx, S_max + 2));
    FGPU_TRY(nblk.alloc(ctx, S_max + 2));
    FGPU_TRY(blkstart.alloc(ctx, S_max + 2));
    FGPU_TRY(desc.alloc(ctx, nb_top + 1));
    if (implicit) FGPU_TRY(brows.alloc(ctx, n / KP_EB + 16));
    FGPU_TRY(cnt.alloc(ctx, (size_t)dmax * nb_top + 2));
    FGPU_TRY(pos.alloc(ctx, (size_t)dmax * nb_top + 2));
    hipLaunchKernelGGL(kp_first_kernel, dim3(1), dim3(1), 0, st, (u32)n, segstart.p, blkstart.p);
    FGPU_HIP(hipGetLastError());
    u32 done = 0;
    const uint2* in = nullptr;
    for (int l = 0; l < L; ++l) {
        KpLevel lv;
        lv.dbits = w[l];
        lv.D = 1u << w[l];
        lv.shift = kb - done - w[l];
        lv.S = 1u << done;
        lv.div = 0;
        const bool first = l == 0, last = l == L - 1;
        const u64 nb_max = (n / KP_EB + 1 + lv.S + 7) & ~7ull;
        const size_t ncnt = (size_t)lv.D * nb_max + 1;
        hipLaunchKernelGGL(kp_blk_table_kernel, dim3(cdiv(nb_max, 256)), dim3(256), 0, st, (const u32*)segstart.p, (const u32*)blkstart.p, lv,
                           (u32)nb_max, desc.p, first && implicit ? rowptr : (const u32*)nullptr, nrows, brows.p, cnt.p);
        FGPU_HIP(hipGetLastError());
        {
            static const char* const names[] = {"kp_count_kernel L1", "kp_count_kernel L2", "kp_count_kernel L3", "kp_count_kernel L4"};
            ProfScope ps(ctx, names[l < 4 ? l : 3], (first ? 4 : 8) * n + 4 * ncnt);
            if (first)
                hipLaunchKernelGGL((kp_count_kernel<true, false>), dim3((u32)nb_max), dim3(256), 0, st, key, val, (const uint2*)nullptr, lv,
                                   (const uint4*)desc.p, cnt.p, (const u32*)nullptr, (u32*)nullptr);
            else
                hipLaunchKernelGGL((kp_count_kernel<false, false>), dim3((u32)nb_max), dim3(256), 0, st, (const u32*)nullptr, (const u32*)nullptr, in, lv,
                                   (const uint4*)desc.p, cnt.p, (const u32*)nullptr, (u32*)nullptr);
            FGPU_HIP(hipGetLastError());
        }
        FGPU_TRY(scan_u32(ctx, cnt.p, pos.p, ncnt, nullptr));
        uint2* outp = last ? nullptr : (l == 0 ? bufA.p : (in == bufA.p ? bufB.p : bufA.p));
        const size_t lds = ((size_t)2 * KP_EB + (size_t)6 * lv.D) * sizeof(u32);
        {
            static const char* const names[] = {"kp_scatter_kernel L1", "kp_scatter_kernel L2", "kp_scatter_kernel L3", "kp_scatter_kernel L4"};
            ProfScope ps(ctx, names[l < 4 ? l : 3], (first ? (implicit ? 4 : 8) : 8) * n + (last ? 4 : 8) * n + 4 * ncnt);
#define KP_SCATTER(F, I, LA)                                                                                                              \
            do {                                                                                                                          \
                if (lds > 48 * 1024)                                                                                                      \
                    FGPU_HIP(hipFuncSetAttribute((const void*)kp_scatter_kernel<F, I, LA, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                hipLaunchKernelGGL((kp_scatter_kernel<F, I, LA, false>), dim3((u32)nb_max), dim3(256), lds, st, key, val, rowptr, nrows, in, lv,  \
                                   (const uint4*)desc.p, (const uint2*)brows.p, (const u32*)pos.p, outp, out_val, (const u32*)nullptr);    \
            } while (0)
            if (first) {
                if (implicit) { if (last) KP_SCATTER(true, true, true); else KP_SCATTER(true, true, false); }
                else { if (last) KP_SCATTER(true, false, true); else KP_SCATTER(true, false, false); }
            } else {
                if (last) KP_SCATTER(false, false, true); else KP_SCATTER(false, false, false);
            }
#undef KP_SCATTER
            FGPU_HIP(hipGetLastError());
        }
        // the next level's segments (or, after the last level, the key pointers)
        const u64 nseg_next = (u64)lv.S * lv.D;
        if (last) {
            hipLaunchKernelGGL(kp_seg_kernel, dim3(cdiv(nkeys + 1, 256)), dim3(256), 0, st, (const u32*)pos.p, (const u32*)blkstart.p, lv, nkeys,
                               keyptr, (u32*)nullptr);
            FGPU_HIP(hipGetLastError());
            if (n_valid_out) {
                if (implicit) *n_valid_out = (u32)n;
                else FGPU_TRY(read_u32(ctx, keyptr + nkeys, n_valid_out));
            }
        } else {
            hipLaunchKernelGGL(kp_seg_kernel, dim3(cdiv(nseg_next + 1, 256)), dim3(256), 0, st, (const u32*)pos.p, (const u32*)blkstart.p, lv,
                               nseg_next, segnext.p, nblk.p);
            FGPU_HIP(hipGetLastError());
            std::swap(segstart, segnext);
            if (nseg_next + 1 <= KP_SMALL_SCAN) {
                hipLaunchKernelGGL(kp_small_scan_kernel, dim3(1), dim3(1024), 0, st, (const u32*)nblk.p, (u32)(nseg_next + 1), blkstart.p);
                FGPU_HIP(hipGetLastError());
            } else {
                FGPU_TRY(scan_u32(ctx, nblk.p, blkstart.p, nseg_next + 1, nullptr));
            }
            in = outp;
        }
        done += w[l];
    }
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
// which form sorts (n pairs, nkeys keys): transpose_mode 2 = the LDS-staged levels always, 3 = the two-level form always,
// 0 = the staged levels where the key space is wide enough for the store shape to matter; false = neither applies
static bool ks_applicable(fgpu_ctx* ctx, u64 n, u64 nkeys, bool* staged) {
    const int mode = ctx->opt.transpose_mode;
    const bool can = nkeys >= 2 && nkeys <= (1ull << 31) && n < 0xFFFFFFFFull - KP_EB;
    *staged = can && (mode == 2 || (mode == 0 && KP_AUTO && nkeys >= (1ull << 17)));
    if (*staged) return true;
    KsGeom g;
    return ks_geometry(n, nkeys, g);
}
static fgpu_info sort_pairs(fgpu_ctx* ctx, const u32* key, const u32* val, const u32* rowptr, u32 nrows, u64 n, u64 nkeys,
                            u32* out_val, u32* keyptr, u32* n_valid_out) {
    bool staged = false;
    if (!ks_applicable(ctx, n, nkeys, &staged)) return FGPU_NO_VALUE;
    return staged ? sort_pairs_by_key_staged(ctx, key, val, rowptr, nrows, n, nkeys, out_val, keyptr, n_valid_out)
                  : sort_pairs_by_key(ctx, key, val, rowptr, nrows, n, nkeys, out_val, keyptr, n_valid_out);
}

// Stable partition of the entries of a CSR into `nparts` streams by slot / div, slot = slot_of[column] (or the column): the
// (slot, row) pairs of part 0 in entry order, then part 1, ...; pstart[nparts + 1] = where every part begins.  One level of
// the partition above with the digit slot / div and implicit rows (the partitioned k-hop plan, bitpart.hip).
fgpu_info partition_csr_entries(fgpu_ctx* ctx, const u32* colidx, const u32* rowptr, u32 nrows, u64 nnz, const u32* slot_of, u32 div,
                                u32 nparts, uint2* out_pairs, u32* pstart_dev) {
    FGPU_REQUIRE(nnz > 0 && nnz < 0xFFFFFFFFull - KP_EB && div > 0 && nparts >= 1 && nparts <= KP_MAX_D, FGPU_INVALID,
                 "partition_csr_entries: size out of range");
    KpLevel lv;
    lv.dbits = 0;
    while ((1u << lv.dbits) < nparts) ++lv.dbits;
    lv.D = 1u << lv.dbits;
    lv.shift = 0;
    lv.S = 1;
    lv.div = div;
    hipStream_t st = ctx->stream();
    const u64 nb_max = (nnz / KP_EB + 2 + 7) & ~7ull;
    const size_t ncnt = (size_t)lv.D * nb_max + 1;
    DevBuf<u32> segstart, blkstart, cnt, pos, slots;
    DevBuf<uint4> desc;
    DevBuf<uint2> brows;
    if (slot_of) FGPU_TRY(slots.alloc(ctx, nnz));
    FGPU_TRY(segstart.alloc(ctx, 4));
    FGPU_TRY(blkstart.alloc(ctx, 4));
    FGPU_TRY(desc.alloc(ctx, nb_max + 1));
    FGPU_TRY(brows.alloc(ctx, nb_max + 1));
    FGPU_TRY(cnt.alloc(ctx, ncnt + 1));
    FGPU_TRY(pos.alloc(ctx, ncnt + 1));
    hipLaunchKernelGGL(kp_first_kernel, dim3(1), dim3(1), 0, st, (u32)nnz, segstart.p, blkstart.p);
    hipLaunchKernelGGL(kp_blk_table_kernel, dim3(cdiv(nb_max, 256)), dim3(256), 0, st, (const u32*)segstart.p, (const u32*)blkstart.p, lv,
                       (u32)nb_max, desc.p, rowptr, nrows, brows.p, cnt.p);
    FGPU_HIP(hipGetLastError());
    {
        ProfScope ps(ctx, "kp_count_kernel part", 4 * nnz + 4 * ncnt);
        hipLaunchKernelGGL((kp_count_kernel<true, true>), dim3((u32)nb_max), dim3(256), 0, st, colidx, (const u32*)nullptr, (const uint2*)nullptr, lv,
                           (const uint4*)desc.p, cnt.p, slot_of, slot_of ? slots.p : (u32*)nullptr);
        FGPU_HIP(hipGetLastError());
    }
    FGPU_TRY(scan_u32(ctx, cnt.p, pos.p, ncnt, nullptr));
    {
        ProfScope ps(ctx, "kp_scatter_kernel part", 4 * nnz + 8 * nnz + 4 * ncnt);
        const size_t lds = ((size_t)2 * KP_EB + (size_t)6 * lv.D) * sizeof(u32);
        hipLaunchKernelGGL((kp_scatter_kernel<true, true, false, true>), dim3((u32)nb_max), dim3(256), lds, st,
                           slot_of ? (const u32*)slots.p : colidx, (const u32*)nullptr, rowptr, nrows, (const uint2*)nullptr, lv,
                           (const uint4*)desc.p, (const uint2*)brows.p, (const u32*)pos.p, out_pairs, (u32*)nullptr, (const u32*)nullptr);
        FGPU_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(kp_seg_kernel, dim3(cdiv((u64)nparts + 1, 256)), dim3(256), 0, st, (const u32*)pos.p, (const u32*)blkstart.p, lv,
                       (u64)nparts, pstart_dev, (u32*)nullptr);
    FGPU_HIP(hipGetLastError());
    return FGPU_OK;
}

// the stable sort by itself (ranking rows by degree for the partitioned k-hop plan, bitpart.hip): explicit values, any key space
fgpu_info sort_u32_pairs_by_key(fgpu_ctx* ctx, const u32* key, const u32* val, u64 n, u64 nkeys, u32* out_val, u32* keyptr) {
    fgpu_info i = sort_pairs_by_key_staged(ctx, key, val, nullptr, 0, n, nkeys, out_val, keyptr, nullptr);
    FGPU_REQUIRE(i != FGPU_NO_VALUE, FGPU_INVALID, "sort_u32_pairs_by_key: size out of range");
    return i;
}

fgpu_info mat_transpose_counting(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a) {
    if (a->is_hyper() || a->nnz < 4096 || a->nrows == 0) return FGPU_NO_VALUE;
    bool staged = false;
    if (!ks_applicable(ctx, a->nnz, a->ncols, &staged)) return FGPU_NO_VALUE;
    fgpu_mat* t = nullptr;
    FGPU_TRY(mat_alloc(ctx, &t, a->ncols, a->nrows, a->nnz, false, 0, false));
    fgpu_info i = sort_pairs(ctx, a->colidx, nullptr, a->rowptr, (u32)a->nrows, a->nnz, a->ncols, t->colidx,
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
    bool staged = false;
    if (!ks_applicable(ctx, n, ncols, &staged) || !ks_applicable(ctx, n, nrows, &staged)) return FGPU_NO_VALUE;
    DevBuf<u32> byc_row, colptr, byr_col, rowptr;
    FGPU_TRY(byc_row.alloc(ctx, n));
    FGPU_TRY(colptr.alloc(ctx, ncols + 1));
    u32 nv = 0;
    FGPU_TRY(sort_pairs(ctx, cols, rows, nullptr, 0, n, ncols, byc_row.p, colptr.p, &nv));
    if (nv == 0) return FGPU_NO_VALUE;   // nothing survived: let the generic path build the empty matrix
    FGPU_TRY(byr_col.alloc(ctx, nv));
    FGPU_TRY(rowptr.alloc(ctx, nrows + 1));
    // the column-major form is a CSR over `ncols` rows whose "column ids" are the original rows
    fgpu_info i = sort_pairs(ctx, byc_row.p, nullptr, colptr.p, (u32)ncols, nv, nrows, byr_col.p, rowptr.p, nullptr);
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
