// bitexpand.hip — bit-parallel k-hop expansion for the CondTraverse core when products get dense.
//
// The reference evaluates F·A₁·A₂… with GrB_mxm(ANY_PAIR) per hop (Matrix::delta_lmxm,
// graph/src/graph/graphblas/matrix.rs:1317-1402, driven by expand_batch,
// graph/src/runtime/ops/cond_traverse.rs:602-605).  On R-MAT graphs the second / third hop of a
// 1024-row batch touches billions of entries that collapse onto a few hundred million distinct
// (row, dest) pairs: gathering and sorting them (spgemm.hip) is bound by that "flops" volume.
// The same set algebra has a transposed form with one BIT per source row:
//
//     X[v] = { i : (i, v) in F }                 W = ceil(nsrc / 64) words per vertex
//     Y[v] = OR_{u in A'[v,:]} X[u]              one pull over the in-edges per hop
//     Y[v] &= ~X[u] for (u, v) in dm ;  Y[v] |= X[u] for (u, v) in dp
//
// which is exactly (F·m)<¬(F·dm)> ∪ (F·dp) read column-wise, including the reference's row-level
// mask quirk (bit i of (X·dm)[v] is set iff ANY source of row i has a tombstoned edge to v).  A hop
// streams the transposed matrix once (4 B / entry) and gathers one X row (8·W bytes, a full cache
// line at W = 16) per entry: HBM-bound, independent of how many duplicates the product contains.
// The result is converted back to the ascending (row_i, dest) CSR the operator emits
// (cond_traverse.rs:644) by a 64×64 bit transposition per (64 vertices, word): ballot per source bit.
#include "bitexpand.hpp"

namespace fgpu {

constexpr u32 BP_ITEM = 256;    // entries of A' per work item (the sparse pull walks an item as 4 trips of 64)
static_assert(BP_ITEM == 256, "bp_pull_kernel<.., SPARSE> unrolls an item into four 64-entry trips");
constexpr u32 BP_VCHUNK = 4096; // vertices per emission chunk (64 blocks of 64)

static fgpu_info bp_acc_alloc(fgpu_ctx* ctx, DevBuf<u64>& acc) {
    FGPU_TRY(acc.alloc(ctx, BP_ACC_WORDS));
    FGPU_HIP(hipMemsetAsync(acc.p, 0, BP_ACC_WORDS * sizeof(u64), ctx->stream()));
    return FGPU_OK;
}
// the slots summed by one workgroup; thread 0 then publishes both sums into the lane's mapped line itself (ctx.hip pub_begin /
// pub_wait) — ONE dispatch and one scalar round trip per read-back; without a mapped line the sums go to slot 0's spare words
// and read_words fetches them
__global__ __launch_bounds__(256) void bp_acc_fold_kernel(unsigned long long* __restrict__ acc, u32* __restrict__ pub, u32 seq) {
    __shared__ u64 s_a[4], s_b[4];
    u64 a = acc[(size_t)threadIdx.x * BP_ACC_STRIDE], b = acc[(size_t)threadIdx.x * BP_ACC_STRIDE + 1];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if (lane_id() == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 sa = s_a[0] + s_a[1] + s_a[2] + s_a[3], sb = s_b[0] + s_b[1] + s_b[2] + s_b[3];
        if (pub) {
            __hip_atomic_store(pub + 0, (u32)sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub + 1, (u32)(sa >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub + 2, (u32)sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub + 3, (u32)(sb >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // after the words: release
        } else {
            acc[2] = sa; acc[3] = sb;
        }
    }
}
static_assert(BP_ACC_SLOTS == 256 && BP_ACC_STRIDE >= 4, "bp_acc_fold_kernel: one thread per slot, sums in words 2 and 3 of slot 0");
static fgpu_info bp_acc_read(fgpu_ctx* ctx, u64* acc, u64* a, u64* b) {
    u32* pub = nullptr;
    u32 seq = 0;
    const bool mapped = pub_begin(ctx, &pub, &seq);
    hipLaunchKernelGGL(bp_acc_fold_kernel, dim3(1), dim3(256), 0, ctx->stream(), (unsigned long long*)acc, mapped ? pub : (u32*)nullptr, seq);
    FGPU_HIP(hipGetLastError());
    u32 w[4] = {0, 0, 0, 0};
    if (mapped) FGPU_TRY(pub_wait(ctx, seq, 4, w));
    else FGPU_TRY(read_words(ctx, (const u32*)(acc + 2), 4, w));
    if (a) *a = (u64)w[0] | ((u64)w[1] << 32);
    if (b) *b = (u64)w[2] | ((u64)w[3] << 32);
    return FGPU_OK;
}

// ---------------------------------------------------------------------------------
// transpose cache + item list
// ---------------------------------------------------------------------------------
__global__ void bp_item_count_kernel(const u32* __restrict__ rowptr, u32 nrows, u32* __restrict__ cnt) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    cnt[r] = (r < nrows) ? (rowptr[r + 1] - rowptr[r] + BP_ITEM - 1) / BP_ITEM : 0u;
}

// bit v <=> row v has more than BP_ITEM entries (its items OR into one row: they need a zeroed, shared target)
__global__ __launch_bounds__(256) void bp_split_bits_kernel(const u32* __restrict__ rowptr, u32 nrows, u64* __restrict__ bits) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (nrows + 63) >> 6;
    for (u32 w = wave; w < nwords; w += nwaves) {
        const u32 r = (w << 6) + lane;
        const u64 m = __ballot(r < nrows && rowptr[r + 1] - rowptr[r] > BP_ITEM);
        if (lane == 0) bits[w] = m;
    }
}

__global__ void bp_sitem_count_kernel(const u32* __restrict__ rowptr, u32 nrows, u32* __restrict__ cnt) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    const u32 d = (r < nrows) ? rowptr[r + 1] - rowptr[r] : 0u;
    cnt[r] = d > BP_ITEM ? (d + BP_ITEM - 1) / BP_ITEM : 0u;
}

// SPLIT_ONLY: only the rows cut into several items (off = scan of bp_sitem_count_kernel's counts)
template <bool SPLIT_ONLY>
__global__ void bp_item_fill_kernel(const u32* __restrict__ rowptr, u32 nrows, const u32* __restrict__ off,
                                    u32* __restrict__ items) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const u32 b = rowptr[r], e = rowptr[r + 1];
    const u32 n = (e - b + BP_ITEM - 1) / BP_ITEM;
    if (SPLIT_ONLY && n <= 1) return;
    const u32 split = n > 1 ? 0x80000000u : 0u;
    u32 o = off[r];
    for (u32 k = 0; k < n; ++k, ++o) {
        const u32 ib = b + k * BP_ITEM;
        const u32 ie = ib + BP_ITEM < e ? ib + BP_ITEM : e;
        items[3 * o] = r;
        items[3 * o + 1] = ib;
        items[3 * o + 2] = ie | split;  // nnz < 2^31 is checked by the caller
    }
}

// pattern transpose of `m`, built once per snapshot and owned by it (the reference keeps the same
// thing per relationship type: Tensor::matrix_t, tensor.rs:886-888)
static fgpu_info transposed_with_items(fgpu_ctx* ctx, const fgpu_mat* m, const fgpu_mat** out) {
    std::lock_guard<std::mutex> idx_guard(m->idx_mu);   // one builder; the build is synchronised before it is published
    if (!m->tcache) {
        fgpu_mat* t = nullptr;
        FGPU_TRY(mat_transpose_pattern(ctx, &t, m));
        FGPU_HIP(hipStreamSynchronize(ctx->stream()));
        m->tcache = t;
    }
    const fgpu_mat* t = m->tcache;   // reachable only through m: m's mutex covers its item list too
    if (!t->bp_items && t->nnz) {
        FGPU_REQUIRE(!t->is_hyper(), FGPU_INVALID, "bit-parallel expansion needs a non-hypersparse transpose");
        const u32 nrows = (u32)t->nrows;
        DevBuf<u32> cnt, off;
        FGPU_TRY(cnt.alloc(ctx, (size_t)nrows + 1));
        FGPU_TRY(off.alloc(ctx, (size_t)nrows + 1));
        hipLaunchKernelGGL(bp_item_count_kernel, dim3(cdiv((u64)nrows + 1, 256)), dim3(256), 0, ctx->stream(),
                           (const u32*)t->rowptr, nrows, cnt.p);
        FGPU_HIP(hipGetLastError());
        FGPU_TRY(scan_u32(ctx, cnt.p, off.p, (u64)nrows + 1, nullptr));
        u32 n = 0;
        FGPU_TRY(read_u32(ctx, off.p + nrows, &n));
        u32* items = nullptr;
        FGPU_TRY(ctx->dev_alloc((void**)&items, (size_t)(n ? n : 1) * 3 * sizeof(u32)));
        hipLaunchKernelGGL(bp_item_fill_kernel<false>, dim3(cdiv(nrows, 256)), dim3(256), 0, ctx->stream(),
                           (const u32*)t->rowptr, nrows, (const u32*)off.p, items);
        // the split rows' items once more, on their own: the row-group form of the sparse pull leaves exactly these
        // to the item kernel
        hipLaunchKernelGGL(bp_sitem_count_kernel, dim3(cdiv((u64)nrows + 1, 256)), dim3(256), 0, ctx->stream(),
                           (const u32*)t->rowptr, nrows, cnt.p);
        FGPU_HIP(hipGetLastError());
        FGPU_TRY(scan_u32(ctx, cnt.p, off.p, (u64)nrows + 1, nullptr));
        u32 ns = 0;
        FGPU_TRY(read_u32(ctx, off.p + nrows, &ns));
        u32* sitems = nullptr;
        fgpu_info ssi = ctx->dev_alloc((void**)&sitems, (size_t)(ns ? ns : 1) * 3 * sizeof(u32));
        if (ssi != FGPU_OK) { ctx->dev_free(items); return ssi; }
        if (ns)
            hipLaunchKernelGGL(bp_item_fill_kernel<true>, dim3(cdiv(nrows, 256)), dim3(256), 0, ctx->stream(),
                               (const u32*)t->rowptr, nrows, (const u32*)off.p, sitems);
        u64* sbits = nullptr;
        fgpu_info si = ctx->dev_alloc((void**)&sbits, ((size_t)nrows / 64 + 2) * sizeof(u64));
        if (si != FGPU_OK) { ctx->dev_free(items); ctx->dev_free(sitems); return si; }
        hipLaunchKernelGGL(bp_split_bits_kernel, dim3(ctx->cus * 4), dim3(256), 0, ctx->stream(), (const u32*)t->rowptr,
                           nrows, sbits);
        t->bp_split_bits = sbits;
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream());
        if (e != hipSuccess) { ctx->dev_free(items); ctx->dev_free(sitems); set_error("item list build failed: %s", hipGetErrorString(e)); return FGPU_DEVICE; }
        t->n_bp_sitems = ns;
        t->bp_sitems = sitems;
        t->n_bp_items = n;
        t->bp_items = items;
    }
    *out = t;
    return FGPU_OK;
}

// ---------------------------------------------------------------------------------
// F -> X
// ---------------------------------------------------------------------------------
// Entry-parallel: F has few rows (a batch of <= a few thousand sources) but after a hop or two its rows hold
// up to ~10^6 entries each, so one wavefront per row leaves most of the chip idle behind the hub rows.  Every
// lane takes one entry and finds its row in the (L1-resident) row-pointer array.
__global__ __launch_bounds__(256) void bp_scatter_csr_kernel(CsrView f, u32 nrows, u32 nnz, u32 ws,
                                                            u64* __restrict__ x, uint8_t* __restrict__ xflag) {
    for (u32 q = blockIdx.x * 256 + threadIdx.x; q < nnz; q += gridDim.x * 256) {
        u32 lo = 0, hi = nrows - 1;   // largest i with rowptr[i] <= q
        while (lo < hi) {
            u32 mid = (lo + hi + 1) >> 1;
            if (f.rowptr[mid] <= q) lo = mid; else hi = mid - 1;
        }
        const u32 col = f.colidx[q];
        atomicOr((unsigned long long*)&x[(size_t)col * ws + (lo >> 6)], 1ull << (lo & 63));
        xflag[col] = 1;
    }
}

// lazy state: zero exactly the rows the scatter is about to touch (duplicates write the same zeros)
__global__ __launch_bounds__(256) void bp_zero_rows_kernel(const u32* __restrict__ col, u32 nnz, u32 ws, u32 lnsh,
                                                          u64* __restrict__ x) {
    const u32 ln = 1u << lnsh;
    const u32 t = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    const u32 sub = t & (ln - 1u);
    for (u32 q = t >> lnsh; q < nnz; q += nth >> lnsh) {
        const u32 c = col[q];
        for (u32 k = sub; k < ws; k += ln) x[(size_t)c * ws + k] = 0ull;
    }
}

// (eight flags a load; one atomic per WORKGROUP: a byte per lane and an atomic per wavefront on the one counter took 105 us
// for the 4 M flags of RMAT-22 — 8192 same-address atomics at ~5.6 ns each behind a 64-byte-per-instruction read)
__global__ __launch_bounds__(256) void bp_flag_count_kernel(const uint8_t* __restrict__ flag, u32 n, unsigned long long* __restrict__ out) {
    __shared__ u32 s_c[4];
    u32 c = 0;
    const u32 n8 = n >> 3;
    const u64* f8 = reinterpret_cast<const u64*>(flag);     // (device blocks are 256-byte aligned)
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n8; i += gridDim.x * 256) {
        u64 v = f8[i];
        v |= v >> 4; v |= v >> 2; v |= v >> 1;               // bit 0 of every byte = the byte is non-zero
        c += (u32)__popcll(v & 0x0101010101010101ull);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7u)) c += flag[(n8 << 3) + threadIdx.x] != 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if (lane_id() == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 t = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        if (t) atomicAdd(out, (unsigned long long)t);
    }
}

// nnz and the order-independent checksum of the result (sum of row_hash(row) * dest_hash(dest), common.hpp) straight
// from the bit state — fgpu_expand_count needs no CSR.  One lane per (vertex, word).
// tab[k][j][x] = sum of row_hash(64 k + 4 j + b) over the set bits b of the nibble x: 16 look-ups per word replace a
// loop over its set bits (divergent: a wavefront ran as long as its fullest word) with two 64-bit multiplies each.
// (rowmap, nullable: bit i of a row stands for source row rowmap[i], i < nsrc — a chain over compacted source rows)
__global__ void bp_cs_table_kernel(u32 w, u64* __restrict__ tab, const u32* __restrict__ rowmap, u32 nsrc) {
    const u32 t = blockIdx.x * 256 + threadIdx.x;     // (k, j, x)
    if (t >= w * 256) return;
    const u32 k = t >> 8, j = (t >> 4) & 15, x = t & 15;
    u64 s = 0;
    for (u32 b = 0; b < 4; ++b)
        if ((x >> b) & 1u) {
            const u32 bit = k * 64 + 4 * j + b;
            s += cs_row_hash(rowmap ? (bit < nsrc ? (u64)rowmap[bit] : 0ull) : (u64)bit);   // (bits >= nsrc are never set)
        }
    tab[t] = s;
}

template <bool WITH_SUM>
__global__ __launch_bounds__(256) void bp_count_kernel(const u64* __restrict__ y, u32 n, u32 w, u32 ws,
                                                      const u64* __restrict__ label, const u64* __restrict__ tab,
                                                      unsigned long long* __restrict__ acc,
                                                      const u32* __restrict__ vmap /* nullable: row r holds vertex vmap[r] */) {
    extern __shared__ u64 s_tab[];                    // w x 16 x 16 sums (2 KiB per word index)
    if (WITH_SUM) {
        for (u32 i = threadIdx.x; i < w * 256; i += 256) s_tab[i] = tab[i];
        __syncthreads();
    }
    u64 cnt = 0, sum = 0;
    const u64 total = (u64)n * w;
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t < total; t += (u64)gridDim.x * 256) {
        const u32 r = (u32)(t / w), k = (u32)(t % w);
        const u32 v = vmap ? vmap[r] : r;
        if (label && !((label[v >> 6] >> (v & 63)) & 1ull)) continue;
        const u64 bits = y[(size_t)r * ws + k];
        if (bits == 0ull) continue;
        cnt += (u64)__popcll(bits);
        if (WITH_SUM) {
            const u64* tk = s_tab + (size_t)k * 256;
            u64 rs = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) rs += tk[j * 16 + (u32)((bits >> (4 * j)) & 15ull)];
            sum += rs * cs_dest_hash(v);
        }
    }
    bp_block_add2(cnt, WITH_SUM ? sum : 0ull, acc);
}

// ---------------------------------------------------------------------------------
// one hop: Y[v] = OR over in-neighbours u of X[u]
//   ws  = words per vertex row (power of two <= 64, or a multiple of 64)
//   LN  = lanes per neighbour = min(ws, 64); 64 / LN neighbours are gathered per load
// ---------------------------------------------------------------------------------
// SPARSE: the frontier rows X[u] are mostly zero (the hop right after the switch from the sorted-CSR products:
// a few 10^4 non-zero rows among 10^7 vertices).  A byte flag per vertex ("X[u] has a bit set", written by
// whoever wrote X) is probed first — 64 neighbours per wavefront step, one byte each — and only flagged
// neighbours pay the 8 W-byte row gather.  Every variant writes the flags of Y for the next hop.
// flag bytes -> flag bits: the sparse pull probes one flag per matrix entry, and a 1-bit-per-vertex map (2 MiB at
// scale 24) stays in every XCD's L2 where the byte map (16 MiB) does not — measured: the byte probes alone made the
// sparse hop fetch 25 GB through the fabric (profiles/r02d_kernel_counters.json: 198 M fabric reads per launch)
__global__ __launch_bounds__(256) void bp_flag_bits_kernel(const uint8_t* __restrict__ flag, u32 n, u64* __restrict__ bits) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (n + 63) >> 6;
    for (u32 w = wave; w < nwords; w += nwaves) {
        const u32 v = (w << 6) + lane;
        const u64 m = __ballot(v < n && flag[v] != 0);
        if (lane == 0) bits[w] = m;
    }
}

// coarse filter over the flag bits for the sparse pull: bit j <=> any flag among the 64 << g vertices of block j.  It is
// small enough for LDS (<= 16 KiB), and after a hop from ~10^4 frontier vertices it is ~90 % zeros: most of the 263 M
// probes of an RMAT-24 pull are answered from LDS instead of from L2 (where random 4-byte loads run at ~110 G/s: the
// probes, not bytes, were what the sparse hop cost — 2.4 ms for 3 GB of traffic)
__global__ __launch_bounds__(256) void bp_coarse_bits_kernel(const u64* __restrict__ bits, u32 nwords, u32 g,
                                                             u64* __restrict__ coarse, u32 ncoarse_words) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 cw = wave; cw < ncoarse_words; cw += nwaves) {
        const u64 first = ((u64)cw * 64 + lane) << g;   // first fine word of this lane's block
        u64 any = 0ull;
        for (u32 k = 0; k < (1u << g); ++k)
            if (first + k < nwords) any |= bits[first + k];
        const u64 m = __ballot(any != 0ull);
        if (lane == 0) coarse[cw] = m;
    }
}

struct BpProbe {
    const u64* bits;    // flag bit per vertex
    const u64* coarse;  // flag bit per block of 64 << g vertices (staged in LDS)
    u32 cshift;         // vertex >> cshift = coarse bit (6 + g)
    u32 cwords;         // 64-bit words of the coarse map
};

// What happens to a finished row (MODE): 0 = it is stored into Y (split rows OR into it) and flagged — a hop in the
// middle of a chain; 1 / 2 = the LAST hop of a count-only chain: the row is counted here (2: and its checksum terms
// summed through the LDS nibble tables) and never written, except for "touched" rows — rows cut into several items or
// named by a delta layer — which go to their slot of the side buffer `y` (BpFinal, bitexpand.hpp).
template <int LN, bool SPARSE, int MODE>
__global__ __launch_bounds__(MODE == 2 ? 1024 : 256) void bp_pull_kernel(CsrView at, const u32* __restrict__ items, u32 nitems, u32 ws,
                                                     const u64* __restrict__ x, BpProbe pr,
                                                     u64* __restrict__ y, uint8_t* __restrict__ yflag, BpFinal fin,
                                                     const u32* __restrict__ yperm /* MODE 0, nullable: row v of Y is stored at slot yperm[v] */) {
    // MODE 2 runs 1024-thread workgroups: the 2 KiB-per-word tables are shared by 16 wavefronts, so the LDS they take
    // does not cost resident wavefronts (the gathers are latency-bound: 20 instead of 32 waves per CU made the dense
    // hop 1.6 x slower when every 256-thread workgroup carried its own copy)
    extern __shared__ u64 s_tab[];
    const u64* __restrict__ xbits = pr.bits;
    const u32* s_co = reinterpret_cast<const u32*>(s_tab + (MODE == 2 ? fin.w * 256 : 0));   // coarse flag map (SPARSE)
    if (MODE == 2)
        for (u32 i = threadIdx.x; i < fin.w * 256; i += blockDim.x) s_tab[i] = fin.tab[i];
    if (SPARSE) {
        u64* co = s_tab + (MODE == 2 ? fin.w * 256 : 0);
        for (u32 i = threadIdx.x; i < pr.cwords; i += blockDim.x) co[i] = pr.coarse[i];
    }
    if (MODE == 2 || SPARSE) __syncthreads();
    // per-wavefront list of an item's live neighbours (SPARSE): BP_ITEM ids behind the tables and the coarse map
    u32* s_live = reinterpret_cast<u32*>(s_tab + (MODE == 2 ? fin.w * 256 : 0) + (SPARSE ? pr.cwords : 0)) +
                  (threadIdx.x >> 6) * BP_ITEM;
    u64 f_cnt = 0, f_sum = 0;
    constexpr int SLOTS = 64 / LN;
    const u32 lane = lane_id();
    const u32 wl = lane % LN, slot = lane / LN;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const u32 nwaves = (gridDim.x * blockDim.x) >> 6;
    const u32 nwb = ws / LN;  // word blocks per row (1 unless ws > 64)
    for (u32 it = wave; it < nitems; it += nwaves) {
        // an item belongs to one wavefront: its fields are wave-uniform — kept in scalar registers, so the row
        // addresses and (counting hop) the vertex hash are scalar work
        const u32 v = (u32)__builtin_amdgcn_readfirstlane((int)items[3 * it]);
        const u32 b = (u32)__builtin_amdgcn_readfirstlane((int)items[3 * it + 1]);
        const u32 e3 = (u32)__builtin_amdgcn_readfirstlane((int)items[3 * it + 2]);
        const u32 e = e3 & 0x7FFFFFFFu;
        const bool split = (e3 >> 31) != 0;
        const u32 yv = (MODE == 0 && yperm) ? yperm[v] : v;   // (requested here, with the column ids — not under the `if` that stores the row)
        bool any = false;
        u32 n_live = 0;
        if (SPARSE) {
            // all four 64-entry trips of an item issue their column-id loads and their flag probes before any of them
            // is consumed (a trip is a chain of two dependent round trips: column id -> flag word); the flagged
            // neighbours are then compacted, in entry order, into the wavefront's list in LDS
            u32 un[4];
            u64 live[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 q = b + 64 * k + lane;
                const u32 c = at.colidx[q < e ? q : e - 1u];     // (clamped, not conditional: see bp_pull_groups_kernel)
                un[k] = (q < e) ? c : 0xFFFFFFFFu;
            }
            u64 fw[4];
            bool sv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 cb = (un[k] != 0xFFFFFFFFu ? un[k] : 0u) >> pr.cshift;
                sv[k] = un[k] != 0xFFFFFFFFu && ((s_co[cb >> 5] >> (cb & 31)) & 1u);
                fw[k] = xbits[sv[k] ? (un[k] >> 6) : 0u];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) live[k] = __ballot(sv[k] && ((fw[k] >> (un[k] & 63)) & 1ull));
            const u64 below = (1ull << lane) - 1ull;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((live[k] >> lane) & 1ull) s_live[n_live + (u32)__popcll(live[k] & below)] = un[k];
                n_live += (u32)__popcll(live[k]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the list is read by other lanes of this wavefront
            __builtin_amdgcn_wave_barrier();
        }
        for (u32 wb = 0; wb < nwb; ++wb) {
            const u32 wo = wb * LN + wl;
            u64 acc = 0ull;
            if (SPARSE) {
                // the item's live neighbours were compacted into the wavefront's LDS list: 4 gathers in flight per lane,
                // as in the dense form (a loop that took SLOTS flagged neighbours per trip and waited for their rows
                // before taking the next ones ran at one memory round trip per SLOTS rows: 2.6 ms a hop at RMAT-24)
                for (u32 i0 = 0; i0 < n_live; i0 += 4 * SLOTS) {
                    u32 u[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32 i = i0 + k * SLOTS + slot;
                        u[k] = (i < n_live) ? s_live[i] : 0xFFFFFFFFu;
                    }
                    u64 xv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u64 r = x[(size_t)(u[k] != 0xFFFFFFFFu ? u[k] : 0u) * ws + wo];
                        xv[k] = (u[k] != 0xFFFFFFFFu) ? r : 0ull;
                    }
                    acc |= (xv[0] | xv[1]) | (xv[2] | xv[3]);
                }
            } else {
                // G gathers in flight per lane: 8 when a row has 16+ words (SLOTS <= 4) — an item of A' holds ~30 entries at
                // RMAT-24, so most items then finish in ONE round of gathers instead of two dependent ones
                constexpr int G = SLOTS <= 4 ? 8 : 4;
                for (u32 q0 = b; q0 < e; q0 += G * SLOTS) {
                    u32 u[G];
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        const u32 q = q0 + k * SLOTS + slot;
                        u[k] = (q < e) ? at.colidx[q] : 0xFFFFFFFFu;
                    }
                    u64 xv[G];
#pragma unroll
                    for (int k = 0; k < G; ++k) xv[k] = (u[k] != 0xFFFFFFFFu) ? x[(size_t)u[k] * ws + wo] : 0ull;
#pragma unroll
                    for (int k = 0; k < G; ++k) acc |= xv[k];
                }
            }
#pragma unroll
            for (int d = LN; d < 64; d <<= 1) acc |= __shfl_xor(acc, d, 64);
            if (MODE == 0) {
                if (slot == 0 && acc) {
                    u64* dst = &y[(size_t)yv * ws + wo];
                    if (split) atomicOr((unsigned long long*)dst, (unsigned long long)acc);
                    else *dst = acc;
                }
                any |= __ballot(acc != 0ull) != 0ull;   // any word of the row, whichever lane holds it
            } else if (acc) {                          // after the butterfly every slot holds the row's word `wo`
                const u64 tw = fin.tbits[v >> 6];
                if ((tw >> (v & 63)) & 1ull) {         // touched: several items and / or delta fix-ups meet in the side buffer
                    if (slot == 0) {
                        const u32 sl = fin.tpref[v >> 6] + (u32)__popcll(tw & ((1ull << (v & 63)) - 1ull));
                        atomicOr((unsigned long long*)&y[(size_t)sl * ws + wo], (unsigned long long)acc);
                    }
                } else if (!fin.label || ((fin.label[v >> 6] >> (v & 63)) & 1ull)) {
                    if (slot == 0) f_cnt += (u64)__popcll(acc);
                    if (MODE == 2) {
                        // the 16 nibble look-ups of the word are shared out over the SLOTS lanes that hold it
                        const u64* tk = s_tab + (size_t)wo * 256;
                        u64 rs = 0;
#pragma unroll
                        for (int j = (int)slot; j < 16; j += SLOTS) rs += tk[j * 16 + (u32)((acc >> (4 * j)) & 15ull)];
                        f_sum += rs * cs_dest_hash(v);
                    }
                }
            }
        }
        if (MODE == 0 && any && lane == 0) yflag[v] = 1;   // "maybe non-zero": benign races, never cleared within a hop
    }
    if (MODE != 0) bp_block_add2(f_cnt, MODE == 2 ? f_sum : 0ull, fin.acc);
}

// ---------------------------------------------------------------------------------
// compact records of a sparse bit state, for the pull that reads it
// ---------------------------------------------------------------------------------
// The state a chain holds right after it left the CSR form is the union of ~10^3 out-neighbourhoods: 10^4 - 10^5 non-zero
// rows whose rows hold ONE or two bits each (bit i of X[u] <=> source i has an edge to u; a vertex needs an in-degree of
// ~N / 256 from the sources of the pass to collect more than four).  The next pull gathers such a row once per LIVE entry of
// A' — half of the entries at 1024 live sources, the hubs being in every frontier — and with 128-byte rows that is 16 lanes,
// a 128-byte line through the vector cache and a 16-word OR per entry to convey one bit (hop 2 of a 1024-live-row pass:
// 635 us at RMAT-22 against 354 us for the same hop at half the row width).  rec[u] = the bits of X[u] as up to four 16-bit
// source indices (0xFFFF = none), or BP_REC_ESC when the row holds more: a lane per live entry then loads 8 bytes and sets
// the bits in the group's LDS accumulator itself; only the entries that meet an ESC row go through the row gathers.
constexpr u64 BP_REC_ESC = 0xFFFFFFFFFFFFFFFEull;
__global__ __launch_bounds__(256) void bp_records_kernel(const uint8_t* __restrict__ flag, u32 n, u32 ws, const u64* __restrict__ x,
                                                        u64* __restrict__ rec) {
    __shared__ u32 s_list[2048];
    __shared__ u32 s_cnt;
    const u32 tid = threadIdx.x;
    const u32 tiles = (n + 2047u) >> 11;
    for (u32 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const u32 base = (tile << 11) + tid * 8u;
        u64 f = 0;
        if (base + 8u <= n) {
            f = *reinterpret_cast<const u64*>(flag + base);
        } else {
            for (u32 j = 0; j < 8u; ++j)
                if (base + j < n && flag[base + j]) f |= 0xffull << (8u * j);
        }
        if (f) {
            for (u32 j = 0; j < 8u; ++j)
                if ((f >> (8u * j)) & 0xffull) s_list[atomicAdd(&s_cnt, 1u)] = base + j;
        }
        __syncthreads();
        const u32 cnt = s_cnt;
        for (u32 i = tid; i < cnt; i += 256u) {
            const u32 u = s_list[i];
            const u64* row = x + (size_t)u * ws;
            u64 r = ~0ull;
            u32 c = 0;
            for (u32 k = 0; k < ws && c <= 4u; ++k) {
                u64 w = row[k];
                while (w && c <= 4u) {
                    const u32 b = (u32)__builtin_ctzll(w);
                    w &= w - 1ull;
                    if (c < 4u) r = (r & ~(0xFFFFull << (16u * c))) | ((u64)(k * 64u + b) << (16u * c));
                    ++c;
                }
            }
            rec[u] = c > 4u ? BP_REC_ESC : r;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// sparse pull, row-group form (a mid-chain hop whose X has few non-zero rows)
// ---------------------------------------------------------------------------------
// The item form gives a wavefront one row at a time: at RMAT-24 that is 8.5 M items of ~30 entries, each a chain of five
// dependent round trips (item -> column ids -> flag words -> rows of X -> Y) run at 12 % lane occupancy — 2.3 ms a hop
// whatever is done to the probes.  Here a wavefront takes BP_GROUP consecutive rows: their entries are one contiguous
// range of A', walked a lane per entry (the row of an entry by a 5-step search of the group's offsets in LDS), flagged
// neighbours are compacted into a list of (row, id) pairs, their rows of X are OR-ed into the group's accumulator in
// LDS (ds_or_b64), and the non-zero rows are written to Y once, coalesced.  Rows of more than BP_ITEM entries are left
// to the item kernel (bp_sitems): they OR into Y with atomics and would stall a group.
constexpr u32 BP_GROUP = 32;
constexpr u32 BP_GROUP_WAVES = 8;   // wavefronts per workgroup
template <int LN>
__global__ __launch_bounds__(BP_GROUP_WAVES * 64) void bp_pull_groups_kernel(CsrView at, u32 nrows, const u64* __restrict__ x,
                                                                             BpProbe pr, u64* __restrict__ y,
                                                                             uint8_t* __restrict__ yflag,
                                                                             const u32* __restrict__ next_rowptr,
                                                                             unsigned long long* __restrict__ stats,
                                                                             const u64* __restrict__ later_bits,
                                                                             const u32* __restrict__ yperm /* nullable: row v of Y at slot yperm[v] */,
                                                                             const u64* __restrict__ rec /* nullable: bp_records_kernel's form of X */) {
    // stats (nullable, with next_rowptr): [0] += popcount(Y[v]) * out-degree of v in the next hop's matrix, [1] += rows
    // written — what bp_flops / bp_count_flags would find in a pass of their own.  Rows flagged in `later_bits`
    // (nullable: destinations of a delta layer, whose rows change after this kernel) are left to bp_split_stats_kernel.
    u64 st_flops = 0;
    u32 st_rows = 0;
    constexpr int SLOTS = 64 / LN;
    constexpr u32 R = BP_GROUP;
    extern __shared__ u64 s_mem[];
    const u32* s_co = reinterpret_cast<const u32*>(s_mem);
    for (u32 i = threadIdx.x; i < pr.cwords; i += blockDim.x) s_mem[i] = pr.coarse[i];
    const u32 lane = lane_id();
    const u32 wv = threadIdx.x >> 6;
    u64* acc = s_mem + pr.cwords + (size_t)wv * ((size_t)R * LN + 256 + 32);   // R rows of LN words
    u64* list = acc + (size_t)R * LN;                                           // (row << 32 | id) of up to 256 live entries
    u32* pref = reinterpret_cast<u32*>(list + 256);                             // [0, R): entry offset of a row in the group
    u32* base = pref + R;                                                       // [0, R): rowptr - pref
    for (u32 i = lane; i < R * LN; i += 64) acc[i] = 0ull;
    __syncthreads();
    const u32 wl = lane % LN, slot = lane / LN;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const u32 nwaves = (gridDim.x * blockDim.x) >> 6;
    const u32 ngroups = (nrows + R - 1) / R;
    const u64 below = (1ull << lane) - 1ull;
    for (u32 g = wave; g < ngroups; g += nwaves) {
        const u32 v0 = g * R;
        const u32 ri = v0 + lane < nrows ? v0 + lane : nrows;
        const u32 rp = at.rowptr[ri];
        const u32 nrp = stats ? next_rowptr[ri] : 0u;   // next hop's out-degrees of the group's rows, loaded with the rest
        // ... and the slots of the group's rows in Y: one coalesced load here instead of a gather under `if (a)` in each of the
        // flush's steps (hipcc sinks a conditional load into its branch and waits for it there: eight dependent round trips a group)
        const u32 yp = yperm ? yperm[ri < nrows ? ri : nrows - 1u] : ri;
        const u32 later = later_bits ? (u32)(later_bits[v0 >> 6] >> (v0 & 63)) : 0u;   // (R = 32 rows: half a word)
        const u32 rp1 = (u32)__shfl_down((int)rp, 1, 64);
        const u32 ndeg = (u32)__shfl_down((int)nrp, 1, 64) - nrp;
        u32 deg = lane < R ? rp1 - rp : 0u;
        if (deg > BP_ITEM) deg = 0u;                     // split rows: the item kernel's
        u32 incl = deg;
#pragma unroll
        for (int d = 1; d < (int)R; d <<= 1) {
            const u32 t = (u32)__shfl_up((int)incl, d, 64);
            if ((int)lane >= d) incl += t;
        }
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, R - 1);
        if (total == 0) continue;
        if (lane < R) {
            pref[lane] = incl - deg;
            base[lane] = rp - (incl - deg);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (u32 t0 = 0; t0 < total; t0 += 256) {
            u32 un[4], rw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 t = t0 + 64 * k + lane;
                u32 lo = 0;                                  // largest row with pref[row] <= t
#pragma unroll
                for (u32 step = R / 2; step >= 1; step >>= 1)
                    if (pref[lo + step] <= t) lo += step;
                rw[k] = lo;
                // (loads go to a clamped address, never under a branch: hipcc sinks a conditional load into its branch and waits
                // for it there — the four flag probes below were four dependent round trips in the ISA, not one)
                const u32 tc = t < total ? t : total - 1u;
                const u32 c = at.colidx[tc + base[t < total ? lo : R - 1u]];
                un[k] = (t < total) ? c : 0xFFFFFFFFu;
            }
            u64 live[4];
            u64 fw[4];
            bool sv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 cb = (un[k] != 0xFFFFFFFFu ? un[k] : 0u) >> pr.cshift;
                sv[k] = un[k] != 0xFFFFFFFFu && ((s_co[cb >> 5] >> (cb & 31)) & 1u);
                fw[k] = pr.bits[sv[k] ? (un[k] >> 6) : 0u];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) live[k] = __ballot(sv[k] && ((fw[k] >> (un[k] & 63)) & 1ull));
            u32 n_live = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((live[k] >> lane) & 1ull) list[n_live + (u32)__popcll(live[k] & below)] = ((u64)rw[k] << 32) | un[k];
                n_live += (u32)__popcll(live[k]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (rec) {
                // a lane per live entry (<= 256 of them: four per lane, their records in flight together): the record's source
                // indices are set in the row's accumulator; entries whose neighbour holds more than four bits are compacted to
                // the front of the list and take the row gathers below
                u64 pe[4], rc[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32 i = 64u * k + lane;
                    pe[k] = (i < n_live) ? list[i] : ~0ull;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u64 r = rec[pe[k] != ~0ull ? (u32)pe[k] : 0u];
                    rc[k] = (pe[k] != ~0ull) ? r : ~0ull;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (every lane has read its list entries before any is rewritten)
                __builtin_amdgcn_wave_barrier();
                u32 n_esc = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool esc = pe[k] != ~0ull && rc[k] == BP_REC_ESC;
                    const u64 em = __ballot(esc);
                    if (esc) list[n_esc + (u32)__popcll(em & below)] = pe[k];
                    n_esc += (u32)__popcll(em);
                    if (!esc) {
                        u64* arow = acc + (size_t)(u32)(pe[k] >> 32) * LN;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const u32 sid = (u32)(rc[k] >> (16 * j)) & 0xFFFFu;
                            if (sid < 0xFFFEu) atomicOr((unsigned long long*)&arow[sid >> 6], 1ull << (sid & 63u));
                        }
                    }
                }
                n_live = n_esc;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            for (u32 i0 = 0; i0 < n_live; i0 += 4 * SLOTS) {
                u64 pu[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32 i = i0 + k * SLOTS + slot;
                    pu[k] = (i < n_live) ? list[i] : ~0ull;
                }
                u64 xv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u64 r = x[(size_t)(pu[k] != ~0ull ? (u32)pu[k] : 0u) * LN + wl];
                    xv[k] = (pu[k] != ~0ull) ? r : 0ull;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (xv[k]) atomicOr((unsigned long long*)&acc[(u32)(pu[k] >> 32) * LN + wl], (unsigned long long)xv[k]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // flush: SLOTS rows per step, a row's LN words on LN consecutive lanes
        for (u32 r0 = 0; r0 < R; r0 += SLOTS) {
            const u32 row = r0 + slot;
            const u64 a = row < R ? acc[row * LN + wl] : 0ull;   // (LN == 1: 64 slots, 32 rows)
            const u32 yrow = (u32)__shfl((int)yp, (int)(row < R ? row : 0u), 64);
            if (a) {
                y[(size_t)yrow * LN + wl] = a;
                acc[row * LN + wl] = 0ull;
            }
            const u64 nzm = __ballot(a != 0ull);
            const u64 mine = LN == 64 ? nzm : (nzm >> (slot * LN)) & ((1ull << (LN % 64)) - 1ull);
            if (wl == 0 && mine) yflag[v0 + row] = 1;
            if (stats) {   // (wave-uniform)
                const u32 dg = (u32)__shfl((int)ndeg, (int)(row < R ? row : 0u), 64);
                const bool now = row < R && !((later >> row) & 1u);
                if (now && mine && wl == 0) st_rows += 1;
                if (now) st_flops += (u64)__popcll(a) * dg;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (stats) bp_block_add2(st_flops, (u64)st_rows, stats);   // (wave-uniform condition: every wavefront of the workgroup arrives)
}

// the same two sums over the rows the item kernel finished with atomics (rows of more than BP_ITEM entries: `bits`).
// LN = min(ws, 64) lanes read a row, 64 / LN rows per step (a wavefront per row with a 64-lane reduction took 180 us for
// the ~10^5 split rows of RMAT-24: 48 of 64 lanes idle on 16-word rows, one dependent load per row).
__global__ __launch_bounds__(256) void bp_split_stats_kernel(const u64* __restrict__ bits, u32 n, u32 ws, u32 lnsh,
                                                            const u64* __restrict__ y, const u32* __restrict__ next_rowptr,
                                                            unsigned long long* __restrict__ stats, const u32* __restrict__ yperm) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (n + 63) >> 6;
    const u32 ln = 1u << lnsh, rpw = 64u >> lnsh;
    const u32 wl = lane & (ln - 1u), slot = lane >> lnsh;
    const u64 field = ln == 64 ? ~0ull : (((1ull << ln) - 1ull) << (slot * ln));
    u64 fl = 0, rows = 0;
    // 64 bitmap words per step, a lane each (one dependent load per 4096 vertices instead of one per 64: the ~10^5 split rows
    // of a hop sit in a bitmap that is almost all zeros, and a wavefront walking it word by word spent 40 us on 32 round trips).
    // The set bits of the 64 words are packed into an LDS list first, so that every step reads 64 / LN rows whatever words
    // they came from (a dirty layer names ~2 vertices in every word: a step per word read 2 rows with 8 slots, 93 us).
    __shared__ u32 s_list[4][256];
    u32* list = s_list[threadIdx.x >> 6];
    for (u32 w0 = wave * 64; w0 < nwords; w0 += nwaves * 64) {
      u64 mine = w0 + lane < nwords ? bits[w0 + lane] : 0ull;
      while (__ballot(mine != 0ull)) {
        // up to 256 vertices into the list: lane order, bit order (lanes whose bits do not fit keep them for the next round)
        const u32 pc = (u32)__popcll(mine);
        u32 inc = pc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const u32 t = (u32)__shfl_up((int)inc, o, 64); if ((int)lane >= o) inc += t; }
        u32 at = inc - pc;
        const u32 total = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        while (mine && at < 256u) {
            list[at++] = ((w0 + lane) << 6) + (u32)__builtin_ctzll(mine);
            mine &= mine - 1ull;
        }
        const u32 cnt = total < 256u ? total : 256u;
        for (u32 b0 = 0; b0 < cnt; b0 += rpw) {
            const u32 v = b0 + slot < cnt ? list[b0 + slot] : 0xFFFFFFFFu;
            u32 pcr = 0;
            if (v != 0xFFFFFFFFu)
                for (u32 k = wl; k < ws; k += ln) pcr += (u32)__popcll(y[(size_t)(yperm ? yperm[v] : v) * ws + k]);
            const u64 nz = __ballot(pcr != 0);
            if (pcr) fl += (u64)pcr * (next_rowptr[v + 1] - next_rowptr[v]);
            if (wl == 0 && (nz & field)) rows += 1;
        }
      }
    }
    bp_block_add2(fl, rows, stats);
}

// side-buffer bookkeeping of the counting hop: mark the destinations a delta layer names, popcount the touched words
__global__ void bp_mark_cols_kernel(const u32* __restrict__ col, u32 nnz, u64* __restrict__ bits) {
    for (u32 q = blockIdx.x * 256 + threadIdx.x; q < nnz; q += gridDim.x * 256)
        atomicOr((unsigned long long*)&bits[col[q] >> 6], 1ull << (col[q] & 63));
}
__global__ void bp_word_popc_kernel(const u64* __restrict__ bits, u32 nwords, u32* __restrict__ pc) {
    const u32 w = blockIdx.x * 256 + threadIdx.x;
    if (w <= nwords) pc[w] = w < nwords ? (u32)__popcll(bits[w]) : 0u;
}
// vertex of every slot of the side buffer (slot = rank of v in the touched bitmap)
__global__ __launch_bounds__(256) void bp_slot_vertex_kernel(const u64* __restrict__ tbits, const u32* __restrict__ tpref,
                                                            u32 n, u32* __restrict__ vmap) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 nwords = (n + 63) >> 6;
    for (u32 wd = wave; wd < nwords; wd += nwaves) {
        const u64 tw = tbits[wd];
        if ((tw >> lane) & 1ull) vmap[tpref[wd] + (u32)__popcll(tw & ((1ull << lane) - 1ull))] = (wd << 6) + lane;
    }
}

// delta layers: Y[v] &= ~X[u] for (u, v) in dm, Y[v] |= X[u] for (u, v) in dp.  Entry-parallel — a delta layer
// built on the device has a dense row-pointer array (one slot per vertex, almost all empty), and a wavefront
// per stored row spent its time walking 16 M empty rows: LN lanes take one entry, its row comes from a binary
// search over the row pointers.
template <bool IS_DM>
__global__ __launch_bounds__(256) void bp_delta_kernel(CsrView d, u32 nnz, u32 w, u32 ws, u32 ln,
                                                      const u64* __restrict__ x, u64* __restrict__ y,
                                                      uint8_t* __restrict__ yflag, const u64* __restrict__ tbits,
                                                      const u32* __restrict__ tpref,
                                                      const uint8_t* __restrict__ xflag /* lazy X: rows without a flag are undefined (= empty) */,
                                                      const u32* __restrict__ wordrow /* stored row of every 64th entry */,
                                                      const u32* __restrict__ xperm /* nullable: row u of X at slot xperm[u] */,
                                                      const u32* __restrict__ yperm /* nullable (no side buffer): row v of Y at slot yperm[v] */) {
    const u32 t = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    const u32 per = nth / ln;             // entries in flight per sweep
    const u32 sub = t % ln;
    for (u32 q = t / ln; q < nnz; q += per) {
        // largest stored row i with rowptr[i] <= q, searched between the rows of the 64-entry word's ends (a delta
        // layer holds ~10^5 rows: the full search was 18 dependent loads per entry)
        u32 lo = wordrow[q >> 6], hi = wordrow[(q >> 6) + 1];
        while (lo < hi) {
            const u32 mid = (lo + hi + 1) >> 1;
            if (d.rowptr[mid] <= q) lo = mid; else hi = mid - 1;
        }
        const u32 u = d.hrows ? d.hrows[lo] : lo;
        if (xflag && !xflag[u]) continue;
        const u32 v = d.colidx[q];
        // the row of v: in Y, or (counting hop) in its slot of the side buffer — every delta destination is "touched"
        size_t yrow = (size_t)(yperm ? yperm[v] : v) * ws;
        if (tbits) {
            const u64 tw = tbits[v >> 6];
            yrow = (size_t)(tpref[v >> 6] + (u32)__popcll(tw & ((1ull << (v & 63)) - 1ull))) * ws;
        }
        for (u32 k = sub; k < w; k += ln) {
            const u64 xv = x[(size_t)(xperm ? xperm[u] : u) * ws + k];
            if (xv == 0ull) continue;
            if (IS_DM) atomicAnd((unsigned long long*)&y[yrow + k], (unsigned long long)~xv);
            else {
                atomicOr((unsigned long long*)&y[yrow + k], (unsigned long long)xv);
                if (yflag) yflag[v] = 1;
            }
        }
    }
}

// traversed-edge count of a hop: sum_v popcount(X[v]) * deg(v)
__global__ __launch_bounds__(256) void bp_flops_kernel(CsrView a, u32 w, u32 ws, const u64* __restrict__ x,
                                                      const uint8_t* __restrict__ xflag,
                                                      unsigned long long* __restrict__ out, const u32* __restrict__ xperm) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    u64 sum = 0;
    for (u32 r0 = wave * 64; r0 < a.nvec; r0 += nwaves * 64) {
        const u32 r = r0 + lane;
        if (r >= a.nvec) continue;
        const u32 deg = a.rowptr[r + 1] - a.rowptr[r];
        if (deg == 0) continue;
        const u32 v = a.hrows ? a.hrows[r] : r;
        if (xflag && !xflag[v]) continue;   // a byte instead of the 8 W-byte row: most rows of a sparse state are empty
        u32 pc = 0;
        for (u32 k = 0; k < w; ++k) pc += (u32)__popcll(x[(size_t)(xperm ? xperm[v] : v) * ws + k]);
        sum += (u64)pc * deg;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0 && sum) atomicAdd(out, (unsigned long long)sum);
}

// ---------------------------------------------------------------------------------
// Y -> CSR over the source rows (the north_star's "ballot / prefix-scan output compaction").  A workgroup owns
// (chunk of 4096 vertices, block of <= 64 words): 64 vertices at a time, their rows are read COALESCED into an LDS tile
// (row stride padded by one word: the transposed reads below are then bank-conflict free), then wavefront k takes words
// k, k + 4, ...: lane l holds word wi of vertex v0 + l, and a ballot over bit b is the 64-vertex mask of source row
// 64 wi + b — only the bit columns that occur in the block are visited (a wave-wide OR of the words names them; a 2-hop
// state holds ~7 set bits per 64 x 64 block).  Pass 1 counts per (row, chunk); a flat exclusive scan of cnt[row][chunk]
// IS the CSR position of every (row, chunk) run, so pass 2 writes ascending dest ids with no sort.
// (The first version gave a wavefront one word COLUMN of a chunk: lane l read 8 bytes of row v0 + l, i.e. one 128-byte
// line per lane for 8 useful bytes — PMC showed 30.8 GB fetched per launch for a 2.1 GB state, 4.5 ms per pass.)
// ---------------------------------------------------------------------------------
constexpr u32 BP_ROWS_WB = 64;   // words per word block (a row of more than 64 words is emitted block by block)
constexpr u32 BP_ROWS_NB = 2;    // 64-vertex blocks per LDS tile (one barrier pair and one round of loads per 128 rows)
template <bool EMIT>
__global__ __launch_bounds__(256) void bp_rows_kernel(const u64* __restrict__ y, u32 n, u32 w, u32 ws, u32 nchunks,
                                                     const u64* __restrict__ label, u32* __restrict__ cnt,
                                                     const u64* __restrict__ off, u32* __restrict__ col,
                                                     const uint8_t* __restrict__ flag /* nullable: 0 = row v is empty */) {
    extern __shared__ u64 s_rows[];
    const u32 nwb = (w + BP_ROWS_WB - 1) / BP_ROWS_WB;
    const u32 c = blockIdx.x / nwb, wb = blockIdx.x % nwb;
    const u32 w0 = wb * BP_ROWS_WB;
    const u32 W = (w - w0 < BP_ROWS_WB) ? (w - w0) : BP_ROWS_WB;   // words of this block
    const u32 stride = W + 1;                                       // LDS row stride in words
    u64* tile = s_rows;                                             // BP_ROWS_NB x 64 rows x stride
    u32* acc = reinterpret_cast<u32*>(s_rows + BP_ROWS_NB * 64 * stride);   // W x 64: count / write position of row 64 (w0 + wi) + bit
    const u32 lane = lane_id();
    const u32 wave = threadIdx.x >> 6;
    for (u32 i = threadIdx.x; i < W * 64; i += 256) {
        const u32 row = (w0 + (i >> 6)) * 64 + (i & 63);
        acc[i] = EMIT ? (u32)off[(size_t)row * nchunks + c] : 0u;
    }
    const u64 below = (1ull << lane) - 1ull;
    // the byte flags of the state ("row may hold a bit") for the whole chunk, once: after a hop from a light frontier four
    // rows in five are empty (3.7 M of 16.7 M at RMAT-24) and their 128 bytes are never read; having them in LDS also takes
    // one dependent memory round trip (flags -> rows) out of every tile
    uint8_t* s_flag = reinterpret_cast<uint8_t*>(acc + W * 64);     // BP_VCHUNK bytes
    for (u32 i = threadIdx.x; i < BP_VCHUNK; i += 256) {
        const u32 v = c * BP_VCHUNK + i;
        s_flag[i] = (v < n && (!flag || flag[v])) ? 1 : 0;
    }
    constexpr u32 NB = BP_ROWS_NB;                                  // 64-vertex blocks staged per tile
    for (u32 vb = 0; vb < BP_VCHUNK / 64; vb += NB) {
        const u32 v0 = c * BP_VCHUNK + vb * 64;
        if (v0 >= n) break;                                         // (block-uniform)
        __syncthreads();                                            // previous tile consumed (acc, flags initialised)
        u64 lw[NB];
#pragma unroll
        for (u32 q = 0; q < NB; ++q) {
            const u32 vq = v0 + q * 64;
            lw[q] = __ballot(s_flag[(vb + q) * 64 + lane] != 0);
            if (label && vq < n) lw[q] &= label[vq >> 6];
        }
        u64 any = 0ull;
        for (u32 i = threadIdx.x; i < NB * 64 * W; i += 256) {
            const u32 r = i / W, k = i - r * W;                     // r < NB * 64
            u64 word = 0ull;
            if ((lw[(r >> 6) % NB] >> (r & 63u)) & 1ull) word = y[(size_t)(v0 + r) * ws + w0 + k];
            tile[r * stride + k] = word;
            any |= word;
        }
        if (!__syncthreads_or(any != 0ull)) continue;              // the rows are empty (a barrier: the tile is complete)
        for (u32 wi = wave; wi < W; wi += 4) {
            u32 pos = acc[wi * 64 + lane];                          // lane b: count / position of row 64 (w0 + wi) + b
#pragma unroll
            for (u32 q = 0; q < NB; ++q) {                          // sub-blocks in vertex order: positions stay ascending
                const u64 word = tile[(q * 64 + lane) * stride + wi];
                u64 cols = word;                                    // bit columns that occur among the 64 vertices
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) cols |= __shfl_xor(cols, d, 64);
                const u32 lo = (u32)word, hi = (u32)(word >> 32);
                while (cols) {
                    const u32 bbit = (u32)__builtin_ctzll(cols);    // (wave-uniform)
                    cols &= cols - 1ull;
                    const u32 half = bbit < 32 ? lo : hi;
                    const u64 m = __ballot((half >> (bbit & 31)) & 1u);
                    const u32 pc = (u32)__popcll(m);
                    if (EMIT) {
                        const u32 p = (u32)__builtin_amdgcn_readlane((int)pos, (int)bbit);
                        if ((m >> lane) & 1ull) col[(size_t)p + (u32)__popcll(m & below)] = v0 + q * 64 + lane;
                    }
                    if (lane == bbit) pos += pc;
                }
            }
            acc[wi * 64 + lane] = pos;
        }
    }
    if (!EMIT) {
        __syncthreads();
        for (u32 i = threadIdx.x; i < W * 64; i += 256) {
            const u32 row = (w0 + (i >> 6)) * 64 + (i & 63);
            cnt[(size_t)row * nchunks + c] = acc[i];
        }
    }
}

__global__ void bp_rowptr_full_kernel(const u32* __restrict__ rp_live, const u32* __restrict__ rank, u32 nfull, u32* __restrict__ rowptr) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nfull) rowptr[i] = rp_live[rank[i]];
}
__global__ void bp_rowptr_kernel(const u64* __restrict__ off, u32 k, u32 nchunks, u32* __restrict__ rowptr) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= k) rowptr[i] = (u32)off[(size_t)i * nchunks];
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
static u32 pow2_ceil(u32 x) {
    u32 p = 1;
    while (p < x) p <<= 1;
    return p;
}

static void bp_layout(BitState& s, u32 n, u32 nsrc) {
    s.n = n; s.nsrc = nsrc;
    s.w = (nsrc + 63) / 64;
    if (s.w == 0) s.w = 1;
    s.ws = s.w <= 64 ? pow2_ceil(s.w) : ((s.w + 63) / 64) * 64;
}

static fgpu_info bp_alloc_zero(fgpu_ctx* ctx, DevBuf<u64>& buf, const BitState& s) {
    const size_t words = (size_t)s.n * s.ws;
    // a state the previous batch handed back already zeroed (bp_recycle_state) costs nothing; otherwise a memset
    // (2 GiB = 0.31 ms at RMAT-24, 8.6 GB = 1.4 ms at RMAT-26, per batch)
    buf.release();
    void* q = nullptr;
    bool was_zero = false;
    FGPU_TRY(ctx->dev_alloc_zeroed(&q, (words ? words : 1) * sizeof(u64), &was_zero));
    buf.ctx = ctx; buf.p = (u64*)q; buf.n = words;
    if (was_zero) return FGPU_OK;
    ProfScope ps(ctx, "bit-state memset", words * sizeof(u64));
    FGPU_HIP(hipMemsetAsync(buf.p, 0, words * sizeof(u64), ctx->stream()));
    return FGPU_OK;
}

// zero the rows the flags name (every non-zero row of a non-lazy state is flagged), LN lanes per row
__global__ __launch_bounds__(256) void bp_rezero_rows_kernel(const uint8_t* __restrict__ flag, u32 n, u32 ws2, u32 lsh,
                                                            uint4* __restrict__ x, const u32* __restrict__ xperm) {
    // A block compacts the flagged rows of a 2048-row tile into LDS, then clears them with (1 << lsh) lanes of 16 B per row.
    __shared__ u32 s_list[2048];
    __shared__ u32 s_cnt;
    const u32 tid = threadIdx.x, L = 1u << lsh;
    const u32 tiles = (n + 2047u) >> 11;
    for (u32 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const u32 base = (tile << 11) + tid * 8u;
        u64 f = 0;
        if (base + 8u <= n) {
            f = *reinterpret_cast<const u64*>(flag + base);
        } else {
            for (u32 j = 0; j < 8u; ++j)
                if (base + j < n && flag[base + j]) f |= 0xffull << (8u * j);
        }
        if (f) {
            for (u32 j = 0; j < 8u; ++j)
                if ((f >> (8u * j)) & 0xffull) s_list[atomicAdd(&s_cnt, 1u)] = base + j;
        }
        __syncthreads();
        const u32 cnt = s_cnt;
        for (u32 i = tid >> lsh; i < cnt; i += 256u >> lsh) {
            uint4* row = x + (size_t)(xperm ? xperm[s_list[i]] : s_list[i]) * ws2;
            for (u32 k = tid & (L - 1u); k < ws2; k += L) row[k] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
    }
}

// Move the flagged rows of a (non-lazy) state from one layout to another — row v from slot src[v] to slot dst[v] (nullptr =
// vertex order) of a zeroed block — and leave zeros behind, so the old block goes back to the pool as a zeroed one.  Used
// when the hop that produced a state could not write it in the layout its reader wants (a state scattered from a CSR
// frontier feeding a partitioned count hop; a permuted state meeting the plain or sparse pull after all).
__global__ __launch_bounds__(256) void bp_move_rows_kernel(const uint8_t* __restrict__ flag, u32 n, u32 ws2, u32 lsh,
                                                          uint4* __restrict__ from, uint4* __restrict__ to,
                                                          const u32* __restrict__ src, const u32* __restrict__ dst) {
    __shared__ u32 s_list[2048];
    __shared__ u32 s_cnt;
    const u32 tid = threadIdx.x, L = 1u << lsh;
    const u32 tiles = (n + 2047u) >> 11;
    for (u32 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const u32 base = (tile << 11) + tid * 8u;
        for (u32 j = 0; j < 8u; ++j)
            if (base + j < n && flag[base + j]) s_list[atomicAdd(&s_cnt, 1u)] = base + j;
        __syncthreads();
        const u32 cnt = s_cnt;
        for (u32 i = tid >> lsh; i < cnt; i += 256u >> lsh) {
            const u32 v = s_list[i];
            uint4* a = from + (size_t)(src ? src[v] : v) * ws2;
            uint4* b = to + (size_t)(dst ? dst[v] : v) * ws2;
            for (u32 k = tid & (L - 1u); k < ws2; k += L) {
                b[k] = a[k];
                a[k] = make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads();
    }
}
static fgpu_info bp_relayout(fgpu_ctx* ctx, BitState& s, const u32* want) {
    if (s.perm == want) return FGPU_OK;
    FGPU_REQUIRE(!s.lazy && s.flag.p && (s.ws & 1u) == 0, FGPU_INVALID, "bit state: cannot change the layout of a lazy / odd-width state");
    const size_t words = (size_t)s.n * s.ws;
    void* q = nullptr;
    bool was_zero = false;
    FGPU_TRY(ctx->dev_alloc_zeroed(&q, (words ? words : 1) * sizeof(u64), &was_zero));
    if (!was_zero) FGPU_HIP(hipMemsetAsync(q, 0, words * sizeof(u64), ctx->stream()));
    const u32 ws2 = s.ws / 2;
    u32 lsh = 0;
    while ((2u << lsh) <= ws2 && lsh < 4) ++lsh;
    const u32 tiles = (s.n + 2047u) >> 11;
    const u32 grid = tiles < (u32)ctx->cus * 8u ? tiles : (u32)ctx->cus * 8u;
    {
        ProfScope ps(ctx, "bp_move_rows_kernel", (u64)s.n + 2 * s.nz_rows * s.ws * 8);
        hipLaunchKernelGGL(bp_move_rows_kernel, dim3(grid ? grid : 1), dim3(256), 0, ctx->stream(), (const uint8_t*)s.flag.p, s.n, ws2, lsh,
                           (uint4*)s.x.p, (uint4*)q, s.perm, want);
        FGPU_HIP(hipGetLastError());
    }
    ctx->dev_free_zeroed(s.x.take(), words * sizeof(u64));   // every row that held bits was cleared on the way
    s.x.ctx = ctx; s.x.p = (u64*)q; s.x.n = words;
    s.perm = want;
    return FGPU_OK;
}

// The chain is done with state `s` (the input of its last hop).  When few of its rows hold bits, clearing those rows and
// handing the block back marked "zero" is cheaper than the memset the next batch would pay for a state of the same size.
static void bp_recycle_state(fgpu_ctx* ctx, BitState& s) {
    const size_t words = (size_t)s.n * s.ws;
    if (s.x.p && s.flag.p && !s.lazy && (s.ws & 1u) == 0 && words >= (1u << 20) && s.nz_rows * 8 < (u64)s.n * 7) {   // (a pass of 1024 live sources leaves a third of the rows non-zero: still cheaper than the memset)
        const u32 ws2 = s.ws / 2;
        u32 lsh = 0;
        while ((2u << lsh) <= ws2 && lsh < 4) ++lsh;
        const u32 tiles = (s.n + 2047u) >> 11;
        const u32 grid = tiles < (u32)ctx->cus * 8u ? tiles : (u32)ctx->cus * 8u;
        ProfScope ps(ctx, "bp_rezero_rows_kernel", (u64)s.n + s.nz_rows * s.ws * 8);
        hipLaunchKernelGGL(bp_rezero_rows_kernel, dim3(grid ? grid : 1), dim3(256), 0, ctx->stream(), (const uint8_t*)s.flag.p, s.n,
                           ws2, lsh, (uint4*)s.x.p, s.perm);
        if (hipGetLastError() == hipSuccess) {
            ctx->dev_free_zeroed(s.x.take(), words * sizeof(u64));
            s.flag.release();
            return;
        }
    }
    s.x.release();
    s.flag.release();
}

// for the chain drivers (spgemm.hip): the state has been read for the last time
void bp_finish(fgpu_ctx* ctx, BitState& s) { bp_recycle_state(ctx, s); }

static fgpu_info bp_alloc_flags(fgpu_ctx* ctx, BitState& s) {
    FGPU_TRY(s.flag.alloc(ctx, (size_t)s.n + 1));
    FGPU_HIP(hipMemsetAsync(s.flag.p, 0, (size_t)s.n + 1, ctx->stream()));
    return FGPU_OK;
}

// LDS the sparse pull needs beside the checksum tables of a counting hop: the coarse flag map and the live lists
constexpr size_t BP_SPARSE_LDS = 16384 + 16 * BP_ITEM * sizeof(u32);
static bool bp_sparse_fits(const fgpu_ctx* ctx, size_t table_bytes) { return table_bytes + BP_SPARSE_LDS <= (size_t)ctx->opt.lds_limit; }

static fgpu_info bp_count_flags(fgpu_ctx* ctx, BitState& s) {
    DevBuf<u64> acc;
    FGPU_TRY(acc.alloc(ctx, 1));
    FGPU_HIP(hipMemsetAsync(acc.p, 0, sizeof(u64), ctx->stream()));
    if (s.n) {
        ProfScope ps(ctx, "bp_flag_count_kernel", (u64)s.n);
        u32 grid = cdiv(s.n, 256 * 8);
        if (grid > (u32)ctx->cus * 2) grid = ctx->cus * 2;
        hipLaunchKernelGGL(bp_flag_count_kernel, dim3(grid), dim3(256), 0, ctx->stream(), (const uint8_t*)s.flag.p, s.n,
                           (unsigned long long*)acc.p);
        FGPU_HIP(hipGetLastError());
    }
    return read_u64(ctx, acc.p, &s.nz_rows);
}

fgpu_info bp_from_csr(fgpu_ctx* ctx, BitState& s, const fgpu_mat* f) {
    FGPU_REQUIRE(!f->is_hyper(), FGPU_INVALID, "bit-parallel expansion: F must not be hypersparse");
    bp_layout(s, (u32)f->ncols, (u32)f->nrows);
    // a light frontier (the next pull is certainly the sparse one, whatever its counting mode): zero only the rows the
    // scatter touches — the whole-state memset is 2 GiB = 0.32 ms at RMAT-24 for ~10^4 rows in use
    s.lazy = f->nnz && f->nnz * 8 < (u64)s.n && bp_sparse_fits(ctx, (size_t)s.w * 256 * sizeof(u64));
    if (s.lazy) FGPU_TRY(s.x.alloc(ctx, (size_t)s.n * s.ws));
    else FGPU_TRY(bp_alloc_zero(ctx, s.x, s));
    FGPU_TRY(bp_alloc_flags(ctx, s));
    if (s.lazy) {
        u32 lnsh = 0;
        while ((2u << lnsh) <= s.ws && lnsh < 6) ++lnsh;
        u32 grid = cdiv((u64)f->nnz << lnsh, 256);
        if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
        hipLaunchKernelGGL(bp_zero_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream(), (const u32*)f->colidx, (u32)f->nnz, s.ws,
                           lnsh, s.x.p);
        FGPU_HIP(hipGetLastError());
    }
    if (f->nnz) {
        ProfScope ps(ctx, "bp_scatter_csr_kernel", 4 * (u64)f->nnz + 4 * ((u64)f->nrows + 1) + 16 * (u64)f->nnz);
        u32 grid = cdiv(f->nnz, 256);
        if (grid > (u32)ctx->cus * 32) grid = ctx->cus * 32;
        hipLaunchKernelGGL(bp_scatter_csr_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(f), (u32)f->nrows,
                           (u32)f->nnz, s.ws, s.x.p, s.flag.p);
        FGPU_HIP(hipGetLastError());
    }
    s.nz_rows = f->nnz < f->ncols ? f->nnz : f->ncols;   // an upper bound is all the next hop needs
    return FGPU_OK;
}

// ---------------------------------------------------------------------------------
// F (sorted CSR, few entries) -> one hop -> bit state, by PUSHING: bit i of Y[b] for every (i, u) in F and b in A[u, :].
// A chain that goes to bit form with a LIGHT frontier (the first hop of expand_mode 2, small batches) would pay a whole
// pull — a probe per entry of A', 263 M at RMAT-24 — for a few thousand traversed edges; pushing them costs one 8-byte
// atomic each and needs no dense X (memset, F -> X scatter, flops pass).  The atomics run at ~8 G/s on random words of a
// 2 GiB state (measured: 33 M edges in 4.06 ms against 2.64 ms for the sparse pull), so the caller pushes only below
// nnz / 32 traversed edges.  OP 0: Y |= bit (m, dp), OP 1: Y &= ~bit (dm: the row-level mask
// of Matrix::delta_lmxm, matrix.rs:1343-1361 — any source of row i with a tombstoned edge to b clears (i, b)).
// ---------------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(256) void bp_push_csr_kernel(CsrView f, u32 nnzf, CsrView a, u32 ws, u64* __restrict__ y,
                                                         uint8_t* __restrict__ yflag) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 e = wave; e < nnzf; e += nwaves) {
        u32 lo = 0, hi = f.nrows - 1;                 // row i of entry e (wave-uniform)
        while (lo < hi) {
            const u32 mid = (lo + hi + 1) >> 1;
            if (f.rowptr[mid] <= e) lo = mid; else hi = mid - 1;
        }
        const u32 i = lo, u = f.colidx[e];
        u32 rb, re;
        row_range(a, u, rb, re);
        const size_t wi = i >> 6;
        const unsigned long long bit = 1ull << (i & 63);
        for (u32 q = rb + lane; q < re; q += 64) {
            const u32 b = a.colidx[q];
            unsigned long long* dst = (unsigned long long*)&y[(size_t)b * ws + wi];
            if (OP == 0) { atomicOr(dst, bit); yflag[b] = 1; }
            else atomicAnd(dst, ~bit);
        }
    }
}

// s <- (F·m) &~ (F·dm) | (F·dp) in bit form, straight from the CSR frontier (no dense X is ever built)
fgpu_info bp_push_from_csr(fgpu_ctx* ctx, BitState& s, const fgpu_mat* f, const fgpu_mat* m, const fgpu_mat* dp,
                           const fgpu_mat* dm) {
    FGPU_REQUIRE(!f->is_hyper(), FGPU_INVALID, "bit-parallel expansion: F must not be hypersparse");
    bp_layout(s, (u32)m->ncols, (u32)f->nrows);
    FGPU_TRY(bp_alloc_zero(ctx, s.x, s));
    FGPU_TRY(bp_alloc_flags(ctx, s));
    if (f->nnz) {
        u32 grid = cdiv(f->nnz, 4);
        if (grid > (u32)ctx->cus * 32) grid = ctx->cus * 32;
        struct { const fgpu_mat* a; int op; const char* name; } steps[3] = {
            {m, 0, "bp_push_csr_kernel<m>"}, {dm, 1, "bp_push_csr_kernel<dm>"}, {dp, 0, "bp_push_csr_kernel<dp>"}};
        for (auto& st : steps) {
            if (!st.a || st.a->nnz == 0) continue;
            ProfScope ps(ctx, st.name, 12 * (u64)f->nnz);
            if (st.op == 0)
                hipLaunchKernelGGL(bp_push_csr_kernel<0>, dim3(grid), dim3(256), 0, ctx->stream(), view_of(f), (u32)f->nnz,
                                   view_of(st.a), s.ws, s.x.p, s.flag.p);
            else
                hipLaunchKernelGGL(bp_push_csr_kernel<1>, dim3(grid), dim3(256), 0, ctx->stream(), view_of(f), (u32)f->nnz,
                                   view_of(st.a), s.ws, s.x.p, s.flag.p);
            FGPU_HIP(hipGetLastError());
        }
    }
    return bp_count_flags(ctx, s);
}

// U |= X (same layout): the DISTINCT union over the hops of a variable-length pattern
__global__ void bp_or_kernel(u64* __restrict__ u, const u64* __restrict__ x, u64 words) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < words; i += (u64)gridDim.x * 256) {
        const u64 v = x[i];
        if (v) u[i] |= v;
    }
}

fgpu_info bp_accumulate(fgpu_ctx* ctx, BitState& u, const BitState& x) {
    if (u.x.p == nullptr) {
        bp_layout(u, x.n, x.nsrc);
        FGPU_TRY(bp_alloc_zero(ctx, u.x, u));
    }
    FGPU_REQUIRE(u.n == x.n && u.ws == x.ws, FGPU_DIM_MISMATCH, "bit-state union: layouts differ");
    const u64 words = (u64)x.n * x.ws;
    if (words) {
        ProfScope ps(ctx, "bp_or_kernel", 3 * words * sizeof(u64));
        u32 grid = cdiv(words, 256);
        if (grid > (u32)ctx->cus * 32) grid = ctx->cus * 32;
        hipLaunchKernelGGL(bp_or_kernel, dim3(grid), dim3(256), 0, ctx->stream(), u.x.p, (const u64*)x.x.p, words);
        FGPU_HIP(hipGetLastError());
    }
    return FGPU_OK;
}

fgpu_info bp_count(fgpu_ctx* ctx, const BitState& s, const u64* label_dev, u64* nnz, u64* checksum) {
    DevBuf<u64> acc, tab;
    FGPU_TRY(bp_acc_alloc(ctx, acc));
    const u64 total = (u64)s.n * s.w;
    if (total) {
        // the look-up tables live in LDS: 2 KiB per word of the row, i.e. batches of up to 4096 source rows per pass
        const size_t lds = checksum ? (size_t)s.w * 256 * sizeof(u64) : 0;
        FGPU_REQUIRE(lds <= (size_t)ctx->opt.lds_limit, FGPU_INVALID,
                     "expand checksum: %u source rows need %zu B of LDS tables (limit %d); batch the sources", s.nsrc, lds,
                     ctx->opt.lds_limit);
        ProfScope ps(ctx, checksum ? "bp_count_kernel<checksum>" : "bp_count_kernel<count>", total * sizeof(u64));
        u32 grid = cdiv(total, 256);
        const u32 cap = checksum ? (u32)ctx->cus * 8 : (u32)ctx->cus * 32;   // every workgroup copies the tables once
        if (grid > cap) grid = cap;
        if (checksum) {
            FGPU_TRY(tab.alloc(ctx, (size_t)s.w * 256));
            hipLaunchKernelGGL(bp_cs_table_kernel, dim3(s.w), dim3(256), 0, ctx->stream(), s.w, tab.p, (const u32*)s.rowmap.p, s.nsrc);
            if (lds > 48 * 1024)
                FGPU_HIP(hipFuncSetAttribute((const void*)bp_count_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(bp_count_kernel<true>, dim3(grid), dim3(256), lds, ctx->stream(), (const u64*)s.x.p, s.n,
                               s.w, s.ws, label_dev, (const u64*)tab.p, (unsigned long long*)acc.p, (const u32*)nullptr);
        } else {
            hipLaunchKernelGGL(bp_count_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream(), (const u64*)s.x.p, s.n,
                               s.w, s.ws, label_dev, (const u64*)nullptr, (unsigned long long*)acc.p, (const u32*)nullptr);
        }
        FGPU_HIP(hipGetLastError());
    }
    return bp_acc_read(ctx, acc.p, nnz, checksum);
}

static fgpu_info bp_flops(fgpu_ctx* ctx, const BitState& s, const fgpu_mat* a, u64* flops) {
    if (!a || a->nnz == 0) return FGPU_OK;
    DevBuf<u64> acc;
    FGPU_TRY(acc.alloc(ctx, 1));
    FGPU_HIP(hipMemsetAsync(acc.p, 0, sizeof(u64), ctx->stream()));
    ProfScope ps(ctx, "bp_flops_kernel", 4 * ((u64)a->nvec + 1) + (u64)s.n + (s.nz_rows < s.n ? s.nz_rows : (u64)s.n) * s.w * 8);
    u32 grid = cdiv(a->nvec ? a->nvec : 1, 256);
    if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
    hipLaunchKernelGGL(bp_flops_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(a), s.w, s.ws,
                       (const u64*)s.x.p, (const uint8_t*)s.flag.p, (unsigned long long*)acc.p, s.perm);
    FGPU_HIP(hipGetLastError());
    u64 v = 0;
    FGPU_TRY(read_u64(ctx, acc.p, &v));
    *flops += v;
    return FGPU_OK;
}

// one delta_lmxm in bit form: s.x <- ((X·m) & ~(X·dm)) | (X·dp)
// what the counting form of a hop needs beside the hop itself
struct CountArgs {
    const u64* label;   // destination-label bitmap on the device (nullable)
    u64* nnz;
    u64* checksum;      // nullable
};

static fgpu_info bp_hop_impl(fgpu_ctx* ctx, BitState& s, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm,
                             u64* flops, const CountArgs* ca, const fgpu_mat* next_m = nullptr, const fgpu_mat* count_next = nullptr) {
    FGPU_REQUIRE(m->nrows == s.n, FGPU_DIM_MISMATCH, "bit-parallel hop: matrix has %llu rows, frontier %u",
                 (unsigned long long)m->nrows, s.n);
    FGPU_REQUIRE(m->nnz < 0x7FFFFFFFull, FGPU_INVALID, "bit-parallel hop: nnz must be < 2^31");
    FGPU_REQUIRE(ca || !s.perm, FGPU_INVALID, "bit-parallel hop: a mid-chain hop met a state laid out for a count hop");
    if (flops) {
        if (s.pre_for == m) *flops += s.pre_flops;     // summed by the hop that produced X
        else FGPU_TRY(bp_flops(ctx, s, m, flops));
        if (dp && dp->nnz) FGPU_TRY(bp_flops(ctx, s, dp, flops));
    }
    s.pre_for = nullptr;
    const bool has_dm = dm && dm->nnz, has_dp = dp && dp->nnz;
    const u32 n_out = (u32)m->ncols;
    const fgpu_mat* t = nullptr;
    if (m->nnz) FGPU_TRY(transposed_with_items(ctx, m, &t));
    BitState o;
    bp_layout(o, n_out, s.nsrc);
    // ---- counting hop: touched bitmap (split rows + delta destinations), its prefix, the side buffer, the tables
    DevBuf<u64> tbits, side, tab, acc;
    DevBuf<u32> tpc, tpref, ttot;
    BpFinal fin = {nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    u32 ntouched = 0;
    const int mode = !ca ? 0 : (ca->checksum ? 2 : 1);
    size_t lds = 0;
    // the dense count hop runs in its XCD-partitioned form (bitpart.hip) when the state is well past one L2: the sparse
    // form, rows wider than 128 bytes, 8-byte rows and mid-chain hops keep the plain pull
    const bool will_sparse = m->nnz && s.flag.p != nullptr && s.nz_rows * 8 < (u64)s.n &&
                             bp_sparse_fits(ctx, mode == 2 ? (size_t)s.w * 256 * sizeof(u64) : 0);
    const BpXPlan* xp = nullptr;
    if (ca && t && !will_sparse && !s.lazy && s.ws >= 2 && s.ws <= 16 && (u64)s.n * s.ws * 8 >= (u64)ctx->opt.expand_xcd_min_mb << 20)
        FGPU_TRY(bp_xplan(ctx, m, t, &xp));
    // the layout of the state: the count hop reads what its plan gathers from (hot-first per partition when the partitioned
    // form runs, vertex order otherwise); a mid-chain hop about to feed such a count hop WRITES its rows in that layout
    if (ca) FGPU_TRY(bp_relayout(ctx, s, xp ? bp_xplan_perm(xp) : nullptr));
    const u32* operm = nullptr;
    if (!ca && count_next && count_next->nnz && !count_next->is_hyper() && count_next->nrows == m->ncols && o.ws >= 2 && o.ws <= 16 &&
        (u64)o.n * o.ws * 8 >= (u64)ctx->opt.expand_xcd_min_mb << 20) {
        const fgpu_mat* tn = nullptr;
        const BpXPlan* nxp = nullptr;
        FGPU_TRY(transposed_with_items(ctx, count_next, &tn));
        FGPU_TRY(bp_xplan(ctx, count_next, tn, &nxp));
        operm = bp_xplan_perm(nxp);
    }
    // clean layers in the partitioned form: the fold completes every row, nothing is "touched" — no bitmap, no prefix, no side
    // buffer, no read-back (six launches and a host round trip per batch)
    const bool no_touched = ca && xp && !has_dm && !has_dp;
    if (ca && !no_touched) {
        const u32 nwords = (n_out + 63) / 64;
        FGPU_TRY(tbits.alloc(ctx, (size_t)nwords + 2));
        if (xp)   // every row is completed by the fold: only the destinations of a delta layer are "touched"
            FGPU_HIP(hipMemsetAsync(tbits.p, 0, (size_t)nwords * sizeof(u64), ctx->stream()));
        else if (t && t->bp_split_bits)
            FGPU_HIP(hipMemcpyAsync(tbits.p, t->bp_split_bits, (size_t)nwords * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream()));
        else
            FGPU_HIP(hipMemsetAsync(tbits.p, 0, (size_t)nwords * sizeof(u64), ctx->stream()));
        for (const fgpu_mat* d : {has_dm ? dm : nullptr, has_dp ? dp : nullptr}) {
            if (!d) continue;
            u32 grid = cdiv(d->nnz, 256);
            if (grid > (u32)ctx->cus * 8) grid = ctx->cus * 8;
            hipLaunchKernelGGL(bp_mark_cols_kernel, dim3(grid), dim3(256), 0, ctx->stream(), (const u32*)d->colidx, (u32)d->nnz, tbits.p);
        }
        FGPU_TRY(tpc.alloc(ctx, (size_t)nwords + 2));
        FGPU_TRY(tpref.alloc(ctx, (size_t)nwords + 2));
        FGPU_TRY(ttot.alloc(ctx, 1));
        hipLaunchKernelGGL(bp_word_popc_kernel, dim3(cdiv((u64)nwords + 1, 256)), dim3(256), 0, ctx->stream(), (const u64*)tbits.p,
                           nwords, tpc.p);
        FGPU_HIP(hipGetLastError());
        FGPU_TRY(scan_u32(ctx, tpc.p, tpref.p, (u64)nwords + 1, ttot.p));
        FGPU_TRY(read_u32(ctx, ttot.p, &ntouched));
        FGPU_TRY(side.alloc(ctx, (size_t)(ntouched ? ntouched : 1) * s.ws));
        FGPU_HIP(hipMemsetAsync(side.p, 0, (size_t)(ntouched ? ntouched : 1) * s.ws * sizeof(u64), ctx->stream()));
    }
    if (ca) {
        FGPU_TRY(bp_acc_alloc(ctx, acc));
        if (mode == 2) {
            lds = (size_t)s.w * 256 * sizeof(u64);
            FGPU_REQUIRE(lds <= (size_t)ctx->opt.lds_limit, FGPU_INVALID,
                         "expand checksum: %u source rows need %zu B of LDS tables (limit %d); batch the sources", s.nsrc, lds,
                         ctx->opt.lds_limit);
            FGPU_TRY(tab.alloc(ctx, (size_t)s.w * 256));
            hipLaunchKernelGGL(bp_cs_table_kernel, dim3(s.w), dim3(256), 0, ctx->stream(), s.w, tab.p, (const u32*)s.rowmap.p, s.nsrc);
        }
        fin = BpFinal{tbits.p, tpref.p, ca->label, tab.p, (unsigned long long*)acc.p, s.w};
    } else {
        FGPU_TRY(bp_alloc_zero(ctx, o.x, o));
        FGPU_TRY(bp_alloc_flags(ctx, o));
    }
    bool fuse_stats = false;
    DevBuf<u64> gstats, later;
    u64* ydst = ca ? side.p : o.x.p;          // rows of Y, or the slots of the side buffer
    uint8_t* yflag = ca ? nullptr : o.flag.p;
    int pull_idx = -1;
    if (m->nnz && xp) {
        const u64 xrows = s.nz_rows < (u64)s.n ? s.nz_rows : (u64)s.n;
        FGPU_TRY(bp_xpull_count(ctx, xp, t, (const u64*)s.x.p, s.ws, mode, fin, side.p, lds, xrows));
    } else if (m->nnz) {
        // (the sparse form stages a <= 16 KiB coarse flag map in LDS next to the checksum tables of MODE 2)
        const bool sparse = s.flag.p != nullptr && s.nz_rows * 8 < (u64)s.n && bp_sparse_fits(ctx, lds);
        FGPU_REQUIRE(sparse || !s.lazy, FGPU_INVALID, "bit-parallel hop: a lazily zeroed state needs the sparse pull");
        // row-group form: rows of <= BP_ITEM entries by bp_pull_groups_kernel, the split rows' items by the item kernel
        const bool groups = sparse && mode == 0 && s.ws <= 16 && ctx->opt.expand_row_groups;
        // ... and it can sum the next hop's traversed-edge count and the flagged rows on its way (the next matrix must be
        // plain CSR over the same vertices); rows a delta fix-up changes after the pull are summed after it
        fuse_stats = groups && next_m && !next_m->is_hyper() && next_m->nrows == m->ncols;
        const u32 nitems = groups ? t->n_bp_sitems : t->n_bp_items;
        const u32* item_list = groups ? t->bp_sitems : t->bp_items;
        const CsrView tv = view_of(t);
        u32 grid = cdiv(nitems ? nitems : 1, 4);
        if (grid > (u32)ctx->cus * 32) grid = ctx->cus * 32;
        const u32 threads = mode == 2 ? 1024 : 256;
        if (mode == 2) {                              // 16-wavefront workgroups: same wavefront count, a quarter of the grid
            grid = cdiv(nitems ? nitems : 1, 16);
            if (grid > (u32)ctx->cus * 8) grid = ctx->cus * 8;
        }
        const u32 ln = s.ws < 64 ? s.ws : 64;
        // fewer than 1 row in 8 flagged: probing a flag bit per neighbour first beats gathering 8 W-byte rows
        DevBuf<u64> xbits, xcoarse;
        BpProbe pr = {nullptr, nullptr, 0, 0};
        size_t lds_co = 0;
        if (sparse) {
            const u32 nw = (u32)(((size_t)s.n + 63) / 64);
            FGPU_TRY(xbits.alloc(ctx, (size_t)nw + 1));
            hipLaunchKernelGGL(bp_flag_bits_kernel, dim3(ctx->cus * 4), dim3(256), 0, ctx->stream(), (const uint8_t*)s.flag.p,
                               s.n, xbits.p);
            FGPU_HIP(hipGetLastError());
            u32 g = 0;                                    // coarse map <= 16 KiB = 2048 words of 64 blocks
            while ((((u64)nw + (1ull << g) - 1) >> g) > 2048ull * 64ull) ++g;
            const u32 nblocks = (u32)(((u64)nw + (1ull << g) - 1) >> g);
            const u32 cwords = (nblocks + 63) / 64;
            FGPU_TRY(xcoarse.alloc(ctx, (size_t)cwords + 1));
            hipLaunchKernelGGL(bp_coarse_bits_kernel, dim3(cdiv(cwords, 4)), dim3(256), 0, ctx->stream(), (const u64*)xbits.p, nw, g,
                               xcoarse.p, cwords);
            FGPU_HIP(hipGetLastError());
            pr = BpProbe{xbits.p, xcoarse.p, 6 + g, cwords};
            lds_co = (size_t)cwords * sizeof(u64);
        }
        // algorithmic bytes of the launch: the column ids of A' and the item list once, every non-zero X row once
        // (the per-entry row gathers beyond that are cache traffic), the flag bitmap in the sparse form; the
        // non-zero Y rows a mid-chain hop writes are added once they are counted (bp_count_flags below)
        const u64 xrows = s.nz_rows < (u64)s.n ? s.nz_rows : (u64)s.n;
        const char* nm = ca ? (sparse ? "bp_pull_kernel<sparse, count>" : "bp_pull_kernel<dense, count>")
                            : (sparse ? "bp_pull_kernel<sparse>" : "bp_pull_kernel<dense>");
        ProfScope ps(ctx, nm, 4 * (u64)t->nnz + 12 * (u64)nitems + xrows * 8 * s.w +
                                  (sparse ? (u64)s.n / 8 : 0));
        ps.idx_out = &pull_idx;
#define BP_LAUNCH3(LN, SP, MD)                                                                                          \
    do {                                                                                                                \
        const size_t lds_all = (MD == 2 ? lds : 0) + (SP ? lds_co + (threads / 64) * BP_ITEM * sizeof(u32) : 0);        \
        if (lds_all > 48 * 1024)                                                                                        \
            FGPU_HIP(hipFuncSetAttribute((const void*)bp_pull_kernel<LN, SP, MD>,                                       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_all));                    \
        hipLaunchKernelGGL((bp_pull_kernel<LN, SP, MD>), dim3(grid), dim3(threads), lds_all, ctx->stream(),             \
                           tv, item_list, nitems, s.ws, (const u64*)s.x.p, pr, ydst, yflag, fin, MD == 0 ? operm : (const u32*)nullptr); \
    } while (0)
#define BP_LAUNCH(LN)                                                                                                   \
    do {                                                                                                                \
        if (mode == 0) { if (sparse) BP_LAUNCH3(LN, true, 0); else BP_LAUNCH3(LN, false, 0); }                          \
        else if (mode == 1) { if (sparse) BP_LAUNCH3(LN, true, 1); else BP_LAUNCH3(LN, false, 1); }                     \
        else { if (sparse) BP_LAUNCH3(LN, true, 2); else BP_LAUNCH3(LN, false, 2); }                                    \
    } while (0)
        if (fuse_stats) {
            FGPU_TRY(bp_acc_alloc(ctx, gstats));
            if (has_dm || has_dp) {   // split rows + delta destinations: the rows summed after the fix-ups
                const u32 nwords = (n_out + 63) / 64;
                FGPU_TRY(later.alloc(ctx, (size_t)nwords + 2));
                if (t->bp_split_bits)
                    FGPU_HIP(hipMemcpyAsync(later.p, t->bp_split_bits, (size_t)nwords * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream()));
                else
                    FGPU_HIP(hipMemsetAsync(later.p, 0, (size_t)nwords * sizeof(u64), ctx->stream()));
                for (const fgpu_mat* d : {has_dm ? dm : nullptr, has_dp ? dp : nullptr}) {
                    if (!d) continue;
                    u32 grid = cdiv(d->nnz, 256);
                    if (grid > (u32)ctx->cus * 8) grid = ctx->cus * 8;
                    hipLaunchKernelGGL(bp_mark_cols_kernel, dim3(grid), dim3(256), 0, ctx->stream(), (const u32*)d->colidx, (u32)d->nnz, later.p);
                }
                FGPU_HIP(hipGetLastError());
            }
        }
        DevBuf<u64> recs;
        if (groups && ctx->opt.expand_records && s.nsrc < 0xFFFEu && s.nz_rows * 8 < (u64)s.n) {
            // the state's non-zero rows as records of <= 4 source indices (bp_records_kernel): what the row-group pull reads per
            // live entry instead of the whole row
            FGPU_TRY(recs.alloc(ctx, (size_t)s.n + 1));
            ProfScope psr(ctx, "bp_records_kernel", (u64)s.n + s.nz_rows * (s.ws * 8 + 8));
            const u32 tiles = (s.n + 2047u) >> 11;
            const u32 rgrid = tiles < (u32)ctx->cus * 8u ? tiles : (u32)ctx->cus * 8u;
            hipLaunchKernelGGL(bp_records_kernel, dim3(rgrid ? rgrid : 1), dim3(256), 0, ctx->stream(), (const uint8_t*)s.flag.p, s.n, s.ws,
                               (const u64*)s.x.p, recs.p);
            FGPU_HIP(hipGetLastError());
        }
        if (groups) {
            ProfScope pg(ctx, "sparse pull: row groups", 0);   // (nested in the hop's record: the split between the two launches)
            const size_t per_wave = ((size_t)BP_GROUP * s.ws + 256 + 32) * sizeof(u64);
            const size_t lds_g = lds_co + BP_GROUP_WAVES * per_wave;
            u32 wgs = (u32)((size_t)ctx->opt.lds_limit / lds_g);
            if (wgs < 1) wgs = 1;
            if (wgs > 4) wgs = 4;
            const u32 ggrid = (u32)ctx->cus * wgs;
#define BP_GROUPS(LN)                                                                                                   \
    do {                                                                                                                \
        if (lds_g > 48 * 1024)                                                                                          \
            FGPU_HIP(hipFuncSetAttribute((const void*)bp_pull_groups_kernel<LN>,                                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g));                      \
        hipLaunchKernelGGL(bp_pull_groups_kernel<LN>, dim3(ggrid), dim3(BP_GROUP_WAVES * 64), lds_g, ctx->stream(),     \
                           view_of(t), (u32)t->nrows, (const u64*)s.x.p, pr, ydst, yflag,                               \
                           fuse_stats ? (const u32*)next_m->rowptr : (const u32*)nullptr,                               \
                           fuse_stats ? (unsigned long long*)gstats.p : (unsigned long long*)nullptr,                   \
                           (const u64*)later.p, operm, (const u64*)recs.p);                                             \
    } while (0)
            switch (s.ws) {
                case 1: BP_GROUPS(1); break;
                case 2: BP_GROUPS(2); break;
                case 4: BP_GROUPS(4); break;
                case 8: BP_GROUPS(8); break;
                default: BP_GROUPS(16); break;
            }
#undef BP_GROUPS
            FGPU_HIP(hipGetLastError());
        }
        if (nitems) switch (ln) {
            case 1: BP_LAUNCH(1); break;
            case 2: BP_LAUNCH(2); break;
            case 4: BP_LAUNCH(4); break;
            case 8: BP_LAUNCH(8); break;
            case 16: BP_LAUNCH(16); break;
            case 32: BP_LAUNCH(32); break;
            default: BP_LAUNCH(64); break;
        }
#undef BP_LAUNCH
#undef BP_LAUNCH3
        FGPU_HIP(hipGetLastError());
    }
    {
        u32 ln = 1;                               // lanes per delta entry: a power of two covering the row words
        while (ln < s.w && ln < 64) ln <<= 1;
        const u32 *wr_dm = nullptr, *wr_dp = nullptr;
        if (has_dm) FGPU_TRY(mat_wordrow(ctx, dm, &wr_dm));
        if (has_dp) FGPU_TRY(mat_wordrow(ctx, dp, &wr_dp));
        if (has_dm) {
            ProfScope ps(ctx, "bp_delta_kernel<dm>", (u64)dm->nnz * (4 + 16 * s.w));
            u32 grid = cdiv((u64)dm->nnz * ln, 256);
            if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
            hipLaunchKernelGGL(bp_delta_kernel<true>, dim3(grid), dim3(256), 0, ctx->stream(), view_of(dm), (u32)dm->nnz,
                               s.w, s.ws, ln, (const u64*)s.x.p, ydst, yflag, fin.tbits, fin.tpref,
                               s.lazy ? (const uint8_t*)s.flag.p : (const uint8_t*)nullptr, wr_dm, s.perm, ca ? (const u32*)nullptr : operm);
            FGPU_HIP(hipGetLastError());
        }
        if (has_dp) {
            ProfScope ps(ctx, "bp_delta_kernel<dp>", (u64)dp->nnz * (4 + 16 * s.w));
            u32 grid = cdiv((u64)dp->nnz * ln, 256);
            if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
            hipLaunchKernelGGL(bp_delta_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream(), view_of(dp), (u32)dp->nnz,
                               s.w, s.ws, ln, (const u64*)s.x.p, ydst, yflag, fin.tbits, fin.tpref,
                               s.lazy ? (const uint8_t*)s.flag.p : (const uint8_t*)nullptr, wr_dp, s.perm, ca ? (const u32*)nullptr : operm);
            FGPU_HIP(hipGetLastError());
        }
    }
    if (ca) {
        if (ntouched) {
            ProfScope ps(ctx, "bp_count_kernel<side rows>", (u64)ntouched * s.w * 8);
            DevBuf<u32> vmap;
            FGPU_TRY(vmap.alloc(ctx, (size_t)ntouched + 1));
            hipLaunchKernelGGL(bp_slot_vertex_kernel, dim3(ctx->cus * 4), dim3(256), 0, ctx->stream(), (const u64*)tbits.p,
                               (const u32*)tpref.p, n_out, vmap.p);
            const u64 total = (u64)ntouched * s.w;
            u32 grid = cdiv(total, 256);
            const u32 cap = mode == 2 ? (u32)ctx->cus * 4 : (u32)ctx->cus * 16;
            if (grid > cap) grid = cap;
            if (mode == 2) {
                if (lds > 48 * 1024)
                    FGPU_HIP(hipFuncSetAttribute((const void*)bp_count_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(bp_count_kernel<true>, dim3(grid), dim3(256), lds, ctx->stream(), (const u64*)side.p, ntouched,
                                   s.w, s.ws, ca->label, (const u64*)tab.p, (unsigned long long*)acc.p, (const u32*)vmap.p);
            } else {
                hipLaunchKernelGGL(bp_count_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream(), (const u64*)side.p, ntouched,
                                   s.w, s.ws, ca->label, (const u64*)nullptr, (unsigned long long*)acc.p, (const u32*)vmap.p);
            }
            FGPU_HIP(hipGetLastError());
        }
        FGPU_TRY(bp_acc_read(ctx, acc.p, ca->nnz, ca->checksum));   // ONE read-back for both sums
        bp_recycle_state(ctx, s);   // the chain ends here: no state is left behind (the block goes back zeroed when that is cheap)
        s.n = n_out;
        return FGPU_OK;
    }
    if (fuse_stats) {
        if (t->n_bp_sitems || later.p) {
            u32 lnsh = 0;
            while ((2u << lnsh) <= s.ws && lnsh < 6) ++lnsh;
            hipLaunchKernelGGL(bp_split_stats_kernel, dim3(ctx->cus * 2), dim3(256), 0, ctx->stream(),
                               later.p ? (const u64*)later.p : (const u64*)t->bp_split_bits,
                               n_out, s.ws, lnsh, (const u64*)o.x.p, (const u32*)next_m->rowptr, (unsigned long long*)gstats.p, operm);
            FGPU_HIP(hipGetLastError());
        }
        u64 st[2] = {0, 0};
        FGPU_TRY(bp_acc_read(ctx, gstats.p, &st[0], &st[1]));
        o.nz_rows = st[1];
        o.pre_for = next_m;
        o.pre_flops = st[0];
    } else {
        FGPU_TRY(bp_count_flags(ctx, o));
    }
    prof_add_bytes(ctx, pull_idx, o.nz_rows * 8 * s.w);
    s.x = std::move(o.x);
    s.flag = std::move(o.flag);
    s.perm = operm;
    s.nz_rows = o.nz_rows;
    s.lazy = false;            // o.x was zeroed as a whole
    s.pre_for = o.pre_for;
    s.pre_flops = o.pre_flops;
    s.n = o.n;
    return FGPU_OK;
}

fgpu_info bp_hop(fgpu_ctx* ctx, BitState& s, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm, u64* flops,
                 const fgpu_mat* next_m, const fgpu_mat* count_next) {
    return bp_hop_impl(ctx, s, m, dp, dm, flops, nullptr, next_m, count_next);
}

fgpu_info bp_hop_count(fgpu_ctx* ctx, BitState& s, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm, u64* flops,
                       const u64* label_dev, u64* nnz, u64* checksum) {
    CountArgs ca{label_dev, nnz, checksum};
    return bp_hop_impl(ctx, s, m, dp, dm, flops, &ca);
}

// ---------------------------------------------------------------------------------
// the LAST hop of a chain whose every row has a pre-bound destination (CondTraverse with `to` bound on every row of the batch,
// the multi-hop ExpandInto shape of tests/flow/test_expand_into.py:63-95; cond_traverse.rs:657-661): row i only asks whether
// dst[i] is in (X·m)<not (X·dm)> U (X·dp) — ONE bit of one row of Y.  A wavefront per row walks the in-neighbours of dst[i]
// in A' looking for bit[i]; the delta layers are walked entry by entry (a lane each), the rows whose destination an entry
// names found by a search of the sorted (dst, row) list.  No Y, no emission, nothing to copy back but a byte per row.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bp_probe_rows_kernel(CsrView at, const u32* __restrict__ dst, const u32* __restrict__ bit, u32 k,
                                                            const u64* __restrict__ x, u32 ws, uint8_t* __restrict__ hit) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    for (u32 i = wave; i < k; i += nwaves) {
        const u32 v = dst[i], j = bit[i];
        if (v == 0xFFFFFFFFu || j == 0xFFFFFFFFu) continue;   // (no such vertex / the source row is empty: hit stays 0)
        u32 b, e;
        row_range(at, v, b, e);
        bool found = false;
        for (u32 q0 = b; q0 < e && !found; q0 += 64) {
            const u32 q = q0 + lane;
            bool h = false;
            if (q < e) h = (x[(size_t)at.colidx[q] * ws + (j >> 6)] >> (j & 63)) & 1ull;
            found = __ballot(h) != 0ull;
        }
        if (found && lane == 0) hit[i] = 1;
    }
}
// delta layer: a lane per entry (u, v); every row whose destination is v tests its bit of X[u]
__global__ __launch_bounds__(256) void bp_probe_delta_kernel(CsrView d, u32 nnz, const u32* __restrict__ wordrow,
                                                             const u32* __restrict__ sdst, const u32* __restrict__ srow, u32 k,
                                                             const u32* __restrict__ bit, const u64* __restrict__ x, u32 ws,
                                                             uint8_t* __restrict__ hit) {
    for (u32 q = blockIdx.x * 256 + threadIdx.x; q < nnz; q += gridDim.x * 256) {
        const u32 v = d.colidx[q];
        u32 lo = 0, hi = k;                                  // first entry of the sorted list with sdst >= v
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (sdst[mid] < v) lo = mid + 1; else hi = mid;
        }
        if (lo >= k || sdst[lo] != v) continue;
        u32 rl = wordrow[q >> 6], rh = wordrow[(q >> 6) + 1];   // stored row of entry q (merge.hip mat_wordrow)
        while (rl < rh) {
            const u32 mid = (rl + rh + 1) >> 1;
            if (d.rowptr[mid] <= q) rl = mid; else rh = mid - 1;
        }
        const u32 u = d.hrows ? d.hrows[rl] : rl;
        for (u32 p = lo; p < k && sdst[p] == v; ++p) {
            const u32 i = srow[p], j = bit[i];
            if (j != 0xFFFFFFFFu && ((x[(size_t)u * ws + (j >> 6)] >> (j & 63)) & 1ull)) hit[i] = 1;
        }
    }
}

// hit_m / hit_dm / hit_dp (k bytes each, zeroed by the caller): does row i reach dst[i] through m / dm / dp from the state `s`
fgpu_info bp_probe_rows(fgpu_ctx* ctx, const BitState& s, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm,
                        const u32* dst_dev, const u32* bit_dev, const u32* sdst_dev, const u32* srow_dev, u32 k,
                        uint8_t* hit_m, uint8_t* hit_dm, uint8_t* hit_dp) {
    FGPU_REQUIRE(!s.lazy || s.flag.p, FGPU_INVALID, "bit-parallel probe: bad state");
    FGPU_REQUIRE(m->nrows == s.n, FGPU_DIM_MISMATCH, "bit-parallel probe: matrix has %llu rows, frontier %u",
                 (unsigned long long)m->nrows, s.n);
    if (k == 0) return FGPU_OK;
    if (m->nnz) {
        FGPU_REQUIRE(!s.lazy, FGPU_INVALID, "bit-parallel probe: a lazily zeroed state cannot be probed row by row");
        const fgpu_mat* t = nullptr;
        FGPU_TRY(transposed_with_items(ctx, m, &t));
        ProfScope ps(ctx, "bp_probe_rows_kernel", (u64)k * 16);
        u32 grid = cdiv(k, 4);
        if (grid > (u32)ctx->cus * 16) grid = ctx->cus * 16;
        hipLaunchKernelGGL(bp_probe_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(t), dst_dev, bit_dev, k,
                           (const u64*)s.x.p, s.ws, hit_m);
        FGPU_HIP(hipGetLastError());
    }
    struct { const fgpu_mat* d; uint8_t* hit; } layers[2] = {{dm, hit_dm}, {dp, hit_dp}};
    for (auto& l : layers) {
        if (!l.d || l.d->nnz == 0) continue;
        const u32* wr = nullptr;
        FGPU_TRY(mat_wordrow(ctx, l.d, &wr));
        u32 grid = cdiv(l.d->nnz, 256);
        if (grid > (u32)ctx->cus * 8) grid = ctx->cus * 8;
        hipLaunchKernelGGL(bp_probe_delta_kernel, dim3(grid), dim3(256), 0, ctx->stream(), view_of(l.d), (u32)l.d->nnz, wr, sdst_dev,
                           srow_dev, k, bit_dev, (const u64*)s.x.p, s.ws, l.hit);
        FGPU_HIP(hipGetLastError());
    }
    return FGPU_OK;
}

// X -> CSR snapshot with nsrc rows (dest ascending per row); `label_dev` (nullable) = destination
// label bitmap applied on the way out (cond_traverse.rs:647-651)
// ---------------------------------------------------------------------------------
// emission as a stable sort (round 6): bit state -> (row, vertex) pairs in vertex order -> the LDS-staged counting sort
// ---------------------------------------------------------------------------------
// bp_rows_kernel turns bit columns into rows with one ballot per (64-vertex block, word, bit column that occurs): after two
// hops a block holds ~7 set bits per word, so almost every ballot serves ONE bit — eight wave-wide instructions per emitted
// entry, twice (count, then emit), and every entry leaves as a lone 4-byte store into its row's segment (0.04 of the HBM
// roofline; 55 % of the device time of the operator's 2-hop batch at RMAT-24).  Turning vertex-major bit rows into row-major
// id lists is a stable sort of (row, vertex) pairs by row with the vertices already ascending — what the LDS-staged counting
// sort of transpose.hip does in whole-line runs.  So: a lane per WORD of a non-zero row writes that word's bits as pairs (a
// few instructions per entry, on one lane), the pairs of a 2048-vertex tile landing in one contiguous piece whose start comes
// from a scan of the tiles' popcounts; sort_u32_pairs_by_key then delivers the column ids in row order and the row pointers.
constexpr u32 BP_PT = 2048;   // vertices per tile
// The rows that count (flagged, labelled) are compacted IN ORDER into a list first: one round of flag loads for the whole tile
// instead of a flag -> row dependency per 16 rows (four rows in five are empty after a hop from a light frontier: the first
// version walked all 2048 rows, 128 dependent steps a tile, 280 us for the count pass at RMAT-24), and the passes below run
// four list rows per lane group in flight.  Rows are handled LN = 2^lsh lanes a row (a lane per word, words strided by LN).
template <bool FILL>
__global__ __launch_bounds__(256) void bp_pairs_kernel(const u64* __restrict__ y, u32 n, u32 w, u32 ws, u32 lsh, const uint8_t* __restrict__ flag,
                                                      const u64* __restrict__ label, const u32* __restrict__ perm,
                                                      u32* __restrict__ tile_cnt, const u64* __restrict__ tile_off,
                                                      u32* __restrict__ key, u32* __restrict__ val) {
    __shared__ u32 s_list[BP_PT];             // the tile's rows that count, ascending
    __shared__ u32 s_slot[BP_PT];             // ... and where each lies in the state (perm[v]): gathered ONCE per tile, in one round
                                              // trip — under `if (i < nlist)` in the passes it was a dependent load per row and step
    __shared__ u32 s_row[BP_PT + 1];          // FILL: exclusive prefix of their popcounts
    __shared__ u32 s_wave[4];
    const u32 tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const u32 LN = 1u << lsh, wl = tid & (LN - 1u), grp = tid >> lsh, RPS = 256u >> lsh;      // rows per step of the workgroup
    const u32 v0 = blockIdx.x * BP_PT;
    // ---- the list: thread t owns vertices v0 + 8 t .. + 7
    u32 onbits = 0;
    {
        const u32 base = v0 + tid * 8u;
        u64 f = 0;
        if (!flag) f = ~0ull;
        else if (base + 8u <= n) f = *reinterpret_cast<const u64*>(flag + base);
        else
            for (u32 j = 0; j < 8u; ++j)
                if (base + j < n && flag[base + j]) f |= 0xffull << (8u * j);
        for (u32 j = 0; j < 8u; ++j)
            if (base + j < n && ((f >> (8u * j)) & 0xffull)) onbits |= 1u << j;
        if (label && onbits) onbits &= (u32)(label[base >> 6] >> (base & 63u)) & 0xffu;     // (8 | 64: the byte never straddles a word)
    }
    u32 nlist;
    {
        const u32 c = (u32)__popc(onbits);
        u32 inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const u32 t = (u32)__shfl_up((int)inc, o, 64); if ((int)lane >= o) inc += t; }
        if (lane == 63) s_wave[wv] = inc;
        __syncthreads();
        u32 base = 0;
        for (u32 q = 0; q < wv; ++q) base += s_wave[q];
        nlist = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        u32 at = base + inc - c;
        for (u32 j = 0; j < 8u; ++j)
            if ((onbits >> j) & 1u) s_list[at++] = v0 + tid * 8u + j;
        __syncthreads();
        for (u32 i = tid; i < nlist; i += 256u) s_slot[i] = perm ? perm[s_list[i]] : s_list[i];
        __syncthreads();
    }
    // ---- pass A: popcount of every listed row, four rows per lane group in flight
    u32 total = 0;
    for (u32 i0 = 0; i0 < nlist; i0 += 4u * RPS) {
        u32 pc[4] = {0, 0, 0, 0};
        const u64* row[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 i = i0 + q * RPS + grp;
            row[q] = y + (size_t)s_slot[i < nlist ? i : 0u] * ws;             // (clamped: the loads of the four rows go out together)
        }
        for (u32 k0 = 0; k0 < w; k0 += LN) {
            const u32 k = k0 + wl < w ? k0 + wl : 0u;
            u64 wd[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wd[q] = row[q][k];
#pragma unroll
            for (int q = 0; q < 4; ++q) pc[q] += (k0 + wl < w && i0 + q * RPS + grp < nlist) ? (u32)__popcll(wd[q]) : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            for (u32 d = 1; d < LN; d <<= 1) pc[q] += (u32)__shfl_xor((int)pc[q], (int)d, 64);
            const u32 i = i0 + q * RPS + grp;
            if (FILL) { if (i < nlist && wl == 0) s_row[i] = pc[q]; }
            else if (wl == 0) total += pc[q];
        }
    }
    if (!FILL) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) total += (u32)__shfl_xor((int)total, d, 64);
        __syncthreads();
        if (lane == 0) s_wave[wv] = total;
        __syncthreads();
        if (tid == 0) tile_cnt[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        return;
    }
    __syncthreads();
    // exclusive scan of s_row[0 .. nlist): 8 rows a thread, then the wavefront / workgroup prefix
    {
        u32 loc[8], sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const u32 i = tid * 8 + j; loc[j] = i < nlist ? s_row[i] : 0u; sum += loc[j]; }
        u32 inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const u32 t = (u32)__shfl_up((int)inc, o, 64); if ((int)lane >= o) inc += t; }
        if (lane == 63) s_wave[wv] = inc;
        __syncthreads();
        u32 base = 0;
        for (u32 q = 0; q < wv; ++q) base += s_wave[q];
        u32 run = base + inc - sum;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const u32 i = tid * 8 + j; if (i < nlist) s_row[i] = run; run += loc[j]; }
    }
    __syncthreads();
    // ---- pass B: a lane per word writes the word's bits (rows re-read: the tile's non-zero rows are still in the L2).  The
    // pairs of ONE row may leave in any order of their keys — the sort is by key, and only the order of the VERTICES inside a
    // key matters; a row is one vertex.
    const u32 out0 = (u32)tile_off[blockIdx.x];
    for (u32 i0 = 0; i0 < nlist; i0 += 2u * RPS) {
        for (u32 k0 = 0; k0 < w; k0 += LN) {
            const u32 k = k0 + wl;
            u64 word[2];
            u32 vv[2];
            u32 sl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32 i = i0 + q * RPS + grp;
                vv[q] = i < nlist ? s_list[i] : 0xFFFFFFFFu;
                sl[q] = s_slot[i < nlist ? i : 0u];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) word[q] = y[(size_t)sl[q] * ws + (k < w ? k : 0u)];
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (vv[q] == 0xFFFFFFFFu || k >= w) word[q] = 0ull;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32 i = i0 + q * RPS + grp;
                const u32 pc = (u32)__popcll(word[q]);
                u32 inc = pc;                                  // prefix over the LN lanes of the row
                for (u32 d = 1; d < LN; d <<= 1) { const u32 t = (u32)__shfl_up((int)inc, (int)d, 64); if (wl >= d) inc += t; }
                const u32 rowtot = (u32)__shfl((int)inc, (int)(lane | (LN - 1u)), 64);
                u32 at = 0;
                if (i < nlist) {
                    at = s_row[i];
                    if (wl == 0 && k0 + LN < w) s_row[i] = at + rowtot;      // (the row's next round of words goes on from here)
                }
                u32 o = out0 + at + inc - pc;
                u64 wd = word[q];
                while (wd) {
                    const u32 b = (u32)__builtin_ctzll(wd);
                    wd &= wd - 1ull;
                    key[o] = k * 64u + b;
                    val[o] = vv[q];
                    ++o;
                }
            }
        }
    }
}

// `dense` (nullable): the count pass found more than BP_DENSE_OUT entries per vertex — a result that dense is cheaper through
// the ballot transpose (one ballot serves many bits there, and it writes each row's ids as runs instead of sorting 28 B per
// entry); the caller takes that path, nothing but the 50-80 us count pass is spent.  Crossover from the two forms' measured
// costs (ballot ~99 us per M vertices + 4.9 us per M entries, pairs + sort ~7 + 17: equal at 7.6 entries per vertex; 2-hop
// batches of 1024 rows sit at 1.8 - 3.8, the 3-hop batch of the headline at 115: 11.8 -> 2.8 ms; profiles/NOTES_r06.md section 12).
constexpr u64 BP_DENSE_OUT = 8;
static fgpu_info bp_to_csr_sorted(fgpu_ctx* ctx, const BitState& s, const u64* label_dev, fgpu_mat** out, bool* dense) {
    const u32 ntiles = cdiv(s.n ? s.n : 1, BP_PT);
    u32 lsh = 0;
    while ((2u << lsh) <= s.w && lsh < 4) ++lsh;          // lanes per row: the largest power of two <= min(w, 16)
    DevBuf<u32> tcnt, key, val, rp_live;
    FGPU_TRY(tcnt.alloc(ctx, (size_t)ntiles + 1));
    FGPU_HIP(hipMemsetAsync(tcnt.p + ntiles, 0, sizeof(u32), ctx->stream()));
    const u64 nzr = (s.flag.p && s.nz_rows < (u64)s.n) ? s.nz_rows : (u64)s.n;
    {
        ProfScope ps(ctx, "bp_pairs_kernel<count>", nzr * s.w * 8 + (u64)s.n);
        hipLaunchKernelGGL(bp_pairs_kernel<false>, dim3(ntiles), dim3(256), 0, ctx->stream(), (const u64*)s.x.p, s.n, s.w, s.ws, lsh,
                           (const uint8_t*)s.flag.p, label_dev, s.perm, tcnt.p, (const u64*)nullptr, (u32*)nullptr, (u32*)nullptr);
        FGPU_HIP(hipGetLastError());
    }
    DevBuf<u64> toff;
    FGPU_TRY(toff.alloc(ctx, (size_t)ntiles + 1));
    FGPU_TRY(scan_u32_to_u64(ctx, tcnt.p, toff.p, (u64)ntiles + 1, nullptr));
    u64 nnz = 0;
    FGPU_TRY(read_u64(ctx, toff.p + ntiles, &nnz));
    if (dense && nnz > BP_DENSE_OUT * (u64)s.n) { *dense = true; return FGPU_OK; }
    FGPU_REQUIRE(nnz < 0xFFFFFFFFull - 4096, FGPU_OOM,
                 "expand: %llu result entries exceed the 32-bit row-pointer space; batch the source rows", (unsigned long long)nnz);
    fgpu_mat* o = nullptr;
    const u32 out_rows = s.nsrc_full ? s.nsrc_full : s.nsrc;
    FGPU_TRY(mat_alloc(ctx, &o, out_rows, s.n, nnz, false, 0, false));
    fgpu_info i = FGPU_OK;
    u32* rp = o->rowptr;
    if (s.nsrc_full) { i = rp_live.alloc(ctx, (size_t)s.nsrc + 2); rp = rp_live.p; }
    if (i == FGPU_OK && nnz) {
        i = key.alloc(ctx, nnz);
        if (i == FGPU_OK) i = val.alloc(ctx, nnz);
        if (i == FGPU_OK) {
            ProfScope ps(ctx, "bp_pairs_kernel<fill>", nzr * s.w * 8 + (u64)s.n + 8 * nnz);
            hipLaunchKernelGGL(bp_pairs_kernel<true>, dim3(ntiles), dim3(256), 0, ctx->stream(), (const u64*)s.x.p, s.n, s.w, s.ws, lsh,
                               (const uint8_t*)s.flag.p, label_dev, s.perm, (u32*)nullptr, (const u64*)toff.p, key.p, val.p);
            if (hipGetLastError() != hipSuccess) { set_error("bit-parallel emission failed"); i = FGPU_DEVICE; }
        }
        if (i == FGPU_OK) {
            ProfScope ps(ctx, "emission sort (pairs by row)", 16 * nnz + 4 * nnz);
            i = sort_u32_pairs_by_key(ctx, key.p, val.p, nnz, s.nsrc, o->colidx, rp);
        }
    } else if (i == FGPU_OK) {
        if (hipMemsetAsync(rp, 0, ((size_t)s.nsrc + 1) * sizeof(u32), ctx->stream()) != hipSuccess) i = FGPU_DEVICE;
    }
    if (i == FGPU_OK && s.nsrc_full) {
        // compacted source rows: row i of the result is live row rowrank[i] (an empty source row starts and ends where the next
        // live one starts)
        hipLaunchKernelGGL(bp_rowptr_full_kernel, dim3(cdiv((u64)out_rows + 1, 256)), dim3(256), 0, ctx->stream(),
                           (const u32*)rp_live.p, (const u32*)s.rowrank.p, out_rows, o->rowptr);
        if (hipGetLastError() != hipSuccess) i = FGPU_DEVICE;
    }
    if (i == FGPU_OK && hipStreamSynchronize(ctx->stream()) != hipSuccess) { set_error("bit-parallel emission failed"); i = FGPU_DEVICE; }
    if (i != FGPU_OK) { mat_release(o); return i; }
    *out = o;
    return FGPU_OK;
}

fgpu_info bp_to_csr(fgpu_ctx* ctx, const BitState& s, const u64* label_dev, fgpu_mat** out) {
    // (the sort's key space needs >= 2 rows; a result of a few entries is not worth its launches)
    // expand_emit_sort: 1 = by the density the count pass finds, 2 = always pairs + sort, 0 = always the ballot transpose
    if (ctx->opt.expand_emit_sort && s.nsrc >= 2 && s.n >= 4096) {
        bool dense = false;
        FGPU_TRY(bp_to_csr_sorted(ctx, s, label_dev, out, ctx->opt.expand_emit_sort == 1 ? &dense : nullptr));
        if (!dense) return FGPU_OK;
    }
    const u32 nchunks = cdiv(s.n ? s.n : 1, BP_VCHUNK);
    const u32 krows = s.w * 64;  // counted rows (>= nsrc; the tail rows are empty)
    const size_t ncnt = (size_t)krows * nchunks;
    DevBuf<u32> cnt;
    DevBuf<u64> off;
    FGPU_TRY(cnt.alloc(ctx, ncnt + 1));
    FGPU_TRY(off.alloc(ctx, ncnt + 1));
    FGPU_HIP(hipMemsetAsync(cnt.p + ncnt, 0, sizeof(u32), ctx->stream()));
    const u32 nwb = (s.w + BP_ROWS_WB - 1) / BP_ROWS_WB;
    const u32 grid = nchunks * nwb;
    const u32 wmax = s.w < BP_ROWS_WB ? s.w : BP_ROWS_WB;
    const size_t lds_rows = (size_t)BP_ROWS_NB * 64 * (wmax + 1) * sizeof(u64) + (size_t)wmax * 64 * sizeof(u32) + BP_VCHUNK;
    FGPU_REQUIRE(lds_rows <= (size_t)ctx->opt.lds_limit, FGPU_INVALID,
                 "emission tile of a %u-word row needs %zu B of LDS, the device / lds_limit allows %d", s.w, lds_rows, ctx->opt.lds_limit);
    if (lds_rows > 48 * 1024) {
        FGPU_HIP(hipFuncSetAttribute((const void*)bp_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows));
        FGPU_HIP(hipFuncSetAttribute((const void*)bp_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows));
    }
    // algorithmic bytes of the emission (the north_star's "ballot / prefix-scan output compaction"): per pass the flags of
    // every row and the words of the non-zero rows once, the counts / offsets table once, and 4 B per emitted destination.
    // (Tried and dropped: staging the rows of block i + 1 in registers while block i is balloted — 16 staged words per
    // thread cost occupancy: count 1.03 -> 1.50 ms, emit 1.36 -> 1.93 ms at RMAT-24.)
    const u64 nzr = (s.flag.p && s.nz_rows < (u64)s.n) ? s.nz_rows : (u64)s.n;
    {
        ProfScope ps(ctx, "bp_rows_kernel<count>", nzr * s.w * 8 + (u64)s.n + 4 * (u64)ncnt);
        hipLaunchKernelGGL(bp_rows_kernel<false>, dim3(grid), dim3(256), lds_rows, ctx->stream(), (const u64*)s.x.p, s.n, s.w, s.ws,
                           nchunks, label_dev, cnt.p, (const u64*)nullptr, (u32*)nullptr, (const uint8_t*)s.flag.p);
        FGPU_HIP(hipGetLastError());
    }
    {
        ProfScope ps(ctx, "scan (row x chunk counts)", 12 * (u64)ncnt);
        FGPU_TRY(scan_u32_to_u64(ctx, cnt.p, off.p, ncnt + 1, nullptr));
    }
    u64 nnz = 0;
    FGPU_TRY(read_u64(ctx, off.p + ncnt, &nnz));
    FGPU_REQUIRE(nnz < 0xFFFFFFFFull, FGPU_OOM,
                 "expand: %llu result entries exceed the 32-bit row-pointer space; batch the source rows",
                 (unsigned long long)nnz);
    fgpu_mat* o = nullptr;
    const u32 out_rows = s.nsrc_full ? s.nsrc_full : s.nsrc;
    FGPU_TRY(mat_alloc(ctx, &o, out_rows, s.n, nnz, false, 0, false));
    // rows >= nsrc are empty, so off[nsrc * nchunks] == nnz already
    if (s.nsrc_full) {
        // compacted source rows: row i of the result is live row rowrank[i] (an empty source row starts — and ends — where
        // the next live one starts); the entries themselves are emitted in live-row order, which IS the order of the rows
        DevBuf<u32> rp_live;
        fgpu_info ai = rp_live.alloc(ctx, (size_t)s.nsrc + 1);
        if (ai != FGPU_OK) { mat_release(o); return ai; }
        hipLaunchKernelGGL(bp_rowptr_kernel, dim3(cdiv((u64)s.nsrc + 1, 256)), dim3(256), 0, ctx->stream(),
                           (const u64*)off.p, s.nsrc, nchunks, rp_live.p);
        hipLaunchKernelGGL(bp_rowptr_full_kernel, dim3(cdiv((u64)out_rows + 1, 256)), dim3(256), 0, ctx->stream(),
                           (const u32*)rp_live.p, (const u32*)s.rowrank.p, out_rows, o->rowptr);
        if (hipStreamSynchronize(ctx->stream()) != hipSuccess) { mat_release(o); set_error("bit-parallel emission failed"); return FGPU_DEVICE; }
    } else
    hipLaunchKernelGGL(bp_rowptr_kernel, dim3(cdiv((u64)s.nsrc + 1, 256)), dim3(256), 0, ctx->stream(),
                       (const u64*)off.p, s.nsrc, nchunks, o->rowptr);
    if (nnz) {
        ProfScope ps(ctx, "bp_rows_kernel<emit>", nzr * s.w * 8 + (u64)s.n + 8 * (u64)ncnt + 4 * nnz);
        hipLaunchKernelGGL(bp_rows_kernel<true>, dim3(grid), dim3(256), lds_rows, ctx->stream(), (const u64*)s.x.p, s.n, s.w,
                           s.ws, nchunks, label_dev, (u32*)nullptr, (const u64*)off.p, o->colidx, (const uint8_t*)s.flag.p);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream());
    if (e != hipSuccess) {
        mat_release(o);
        set_error("bit-parallel emission failed: %s", hipGetErrorString(e));
        return FGPU_DEVICE;
    }
    *out = o;
    return FGPU_OK;
}

}  // namespace fgpu
