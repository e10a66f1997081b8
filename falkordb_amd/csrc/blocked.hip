// blocked.hip — the full-matrix boolean pull y = M (x) x for graphs past the reach of the LDS-tile layout of tiled.hip.
//
// tiled.hip stages a 2^20-column tile of x in LDS and publishes one global atomicOr per (tile, 64-row group) slot.  That
// slot holds 252 entries at RMAT-22, 63 at RMAT-24 and 16 at RMAT-26: the wavefront that walks an item has 75 % / 94 % of
// its lanes idle and the pass issues one global atomic per 16 entries — measured 0.67 / 0.35 / 0.13 of the HBM peak.  The
// fix is on the OUTPUT side: a workgroup also keeps a WINDOW of y in LDS, so that every (row, tile) contribution is an LDS
// `ds_or_b32` and y leaves the CU once per window, not once per slot.
//
//   block (w, c)  = the entries of rows [w << wbits, (w + 1) << wbits) with columns in tile c of 2^18 ids, stored
//                   contiguously, blocks of one window one after the other, each padded to a multiple of 256 entries
//                   (= one wavefront trip of 64 lanes x 16 bytes: a trip never straddles two tiles)
//   32-bit entry  =  bits  0..4   bit of the x word              (v_bfe_u32 offset)
//                    bits  5..17  x word in the tile              (2^18 columns = 8192 words = 32 KiB of LDS)
//                    bits 18..29  output word in the window       (<= 12 bits: windows of <= 2^17 rows = 16 KiB of LDS)
//                    bits 30..31  a slice of the QUAD's output bit: the four entries of an aligned 16-byte quad always
//                                 belong to rows with the same (row & 31) — the layout sorts a block by that value and pads
//                                 each of the 32 runs to a multiple of 4 by repeating an entry (OR is idempotent) — and
//                                 entry 0 / 1 / 2 of the quad carry bits 0-1 / 2-3 / 4 of it
// so the kernel needs no per-segment table at all: a lane loads one quad, rebuilds its output bit from the three slices,
// and every hit is one ds_or_b32.  A workgroup owns a unit = (window, range of tiles); per tile its waves issue the loads
// of their trips and of the x tile BEFORE the barrier that retires the previous tile, so a block costs one memory round
// trip.  No cross-lane reduction, no global atomic in the loop; the window is OR-ed into y once at the end.
// Algorithmic bytes are the pass's own: 4 B per entry (+ ~1 % pads), N / 8 of x per window, N / 8 of y.
#include "common.hpp"

namespace fgpu {

constexpr u32 BK_TILE_BITS = 18;                    // columns per x tile
constexpr u32 BK_TILE_WORDS = 1u << (BK_TILE_BITS - 5);
constexpr u32 BK_CHUNK = 256;                       // entries per wavefront trip (64 lanes x 16 B)
constexpr u32 BK_MAX_WBITS = 17;

struct BlockedView {
    const u32* blk_off;    // nblocks + 1 entry offsets (multiples of BK_CHUNK)
    const u32* entries;
    u32 wbits, nwindows, ntiles, nsplit;
};

__device__ __forceinline__ u32 bk_slice(u32 seg, u32 pos_in_quad) {   // the two bits entry `pos_in_quad` of a quad carries
    return ((seg >> (2u * pos_in_quad)) & 3u) << 30;                   // (position 3: seg >> 6 = 0)
}

// ---- build ------------------------------------------------------------------------------------------------------
// a wavefront per row: its entries are sorted by column, so equal tiles are runs of lanes — one atomic per run.
// bucket = (window, tile, row & 31); FILL writes the entry at its final place with the slice of that place.
template <bool FILL>
__global__ __launch_bounds__(256) void bk_scatter_kernel(CsrView a, u32 nrows, u32 wbits, u32 ntiles, u32* __restrict__ cnt,
                                                        const u32* __restrict__ seg_off, u32* __restrict__ entries) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 v = wave; v < nrows; v += nwaves) {
        u32 b, e;
        row_range(a, v, b, e);
        const u32 seg = v & 31u;
        const u32 base_bucket = ((v >> wbits) * ntiles) * 32u + seg;
        const u32 ow = (v & ((1u << wbits) - 1u)) >> 5;
        for (u32 q0 = b; q0 < e; q0 += 64) {
            const u32 q = q0 + lane;
            const bool valid = q < e;
            const u32 u = valid ? a.colidx[q] : 0xFFFFFFFFu;
            const u32 c = u >> BK_TILE_BITS;
            const u32 pc = (u32)__shfl_up((int)c, 1, 64);
            const bool leader = valid && (lane == 0 || pc != c);
            const u64 lm = __ballot(leader);
            // run of this lane: from the last leader at or below it to the next leader above it
            const u64 at_or_below = lm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
            const u32 lead = 63u - (u32)__builtin_clzll(at_or_below | 1ull);
            const u64 above = lm >> 1 >> lane;   // (two shifts: lane 63 must not shift by 64)
            const u32 nvalid = (u32)__popcll(__ballot(valid));
            const u32 run_end = above ? lane + 1u + (u32)__builtin_ctzll(above) : nvalid;
            u32 pos = 0;
            if (leader) pos = atomicAdd(&cnt[base_bucket + c * 32u], run_end - lane);
            pos = (u32)__shfl((int)pos, (int)lead, 64);      // (all lanes active: a shuffle from a masked-off lane reads 0)
            if (FILL && valid) {
                const u32 at = seg_off[base_bucket + c * 32u] + pos + (lane - lead);
                entries[at] = (u & ((1u << BK_TILE_BITS) - 1u)) | (ow << 18) | bk_slice(seg, at & 3u);
            }
        }
    }
}

// padded size of every bucket: a multiple of 4; the last bucket of a block also takes the block up to a multiple of 256
__global__ void bk_pad_count_kernel(const u32* __restrict__ cnt, u32 nblocks, u32* __restrict__ padded) {
    const u32 blk = blockIdx.x * 256 + threadIdx.x;
    if (blk > nblocks) return;
    if (blk == nblocks) { padded[(size_t)nblocks * 32] = 0; return; }
    u32 tot = 0;
    for (u32 s = 0; s < 32; ++s) {
        const u32 p = (cnt[(size_t)blk * 32 + s] + 3u) & ~3u;
        padded[(size_t)blk * 32 + s] = p;
        tot += p;
    }
    if (tot) padded[(size_t)blk * 32 + 31] += ((tot + BK_CHUNK - 1) & ~(BK_CHUNK - 1)) - tot;
}

// pads repeat a real entry of the block (the bucket's first one; for an empty last bucket the block's first one) with the
// segment of THAT entry and the slice of the pad's own place
__global__ void bk_pad_fill_kernel(const u32* __restrict__ cnt, const u32* __restrict__ seg_off, u32 nbuckets,
                                   u32* __restrict__ entries) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nbuckets) return;
    const u32 n = cnt[i], b = seg_off[i], e = seg_off[i + 1];
    if (b + n == e) return;
    u32 src = b, seg = i & 31u;
    if (n == 0) {                                  // (only the last bucket of a block is padded while empty)
        const u32 first = i & ~31u;
        u32 s = 0;
        while (s < 32 && cnt[first + s] == 0) ++s;
        if (s == 32) return;
        src = seg_off[first + s];
        seg = s;
    }
    const u32 body = entries[src] & 0x3FFFFFFFu;
    for (u32 j = b + n; j < e; ++j) entries[j] = body | bk_slice(seg, j & 3u);
}

__global__ void bk_block_off_kernel(const u32* __restrict__ seg_off, u32 nblocks, u32* __restrict__ blk_off) {
    const u32 blk = blockIdx.x * 256 + threadIdx.x;
    if (blk <= nblocks) blk_off[blk] = seg_off[(size_t)blk * 32];
}

// ---- y = M (x) x ------------------------------------------------------------------------------------------------------
typedef u32 bk_u32x4 __attribute__((ext_vector_type(4)));

// One trip of a wavefront: a quad (four entries, one output bit) per lane.  R-MAT rows are heavy-tailed: a hub row puts
// hundreds of consecutive entries of a block on ONE window word, and 64 lanes OR-ing one LDS address in one instruction are
// served one after the other (PMC: 87 % of the LDS cycles of the first version were bank-conflict cycles, 15 per LDS
// instruction).  A row's entries are contiguous in its segment, so lanes whose whole quad names the same word as their
// left neighbour's form a run: the run's hits are combined with two ballots and only its first lane touches the LDS.
__device__ __forceinline__ void bk_probe(const bk_u32x4& d, const u32* __restrict__ xs, u32* __restrict__ os, u32 lane) {
    const u32 seg = (d.x >> 30) | ((d.y >> 30) << 2) | (((d.z >> 30) & 1u) << 4);
    const u32 bit = 1u << seg;
    const u32 ee[4] = {d.x, d.y, d.z, d.w};
    u32 hit[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hit[i] = __builtin_amdgcn_ubfe(xs[(ee[i] >> 5) & (BK_TILE_WORDS - 1u)], ee[i], 1u);
    const u32 w0 = (d.x >> 18) & 0xFFFu, w1 = (d.y >> 18) & 0xFFFu, w2 = (d.z >> 18) & 0xFFFu, w3 = (d.w >> 18) & 0xFFFu;
    const bool uni = w0 == w1 && w1 == w2 && w2 == w3;          // the quad is one row's
    const u32 key = uni ? ((seg << 12) | w0) : (0x80000000u | lane);  // (a mixed quad never continues a run)
    const u32 pkey = (u32)__shfl_up((int)key, 1, 64);           // (all lanes active: see the chunk-prefix note above)
    const bool head = lane == 0 || pkey != key;
    const u64 heads = __ballot(head);
    const u64 hq = __ballot(uni && (hit[0] | hit[1] | hit[2] | hit[3]));
    if (uni) {
        if (head) {
            const u64 above = heads >> 1 >> lane;               // next head above this lane
            const u32 len = above ? 1u + (u32)__builtin_ctzll(above) : 64u - lane;
            const u64 run = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << lane;
            if (hq & run) atomicOr(&os[w0], bit);               // ds_or_b32, once per run
        }
    } else {
        if (hit[0]) atomicOr(&os[w0], bit);
        if (hit[1]) atomicOr(&os[w1], bit);
        if (hit[2]) atomicOr(&os[w2], bit);
        if (hit[3]) atomicOr(&os[w3], bit);
    }
}

// BK_PF = trips a wavefront has in flight per block (x 16 B per lane, twice: current + next block); WPE = wavefronts per
// SIMD the register allocation must allow (8 = two 1024-thread workgroups per CU)
template <int BK_PF, int WPE>
__global__ __launch_bounds__(1024, WPE) void blocked_mxv_kernel(BlockedView t, const u32* __restrict__ x32, u32 x_words32,
                                                          const u64* __restrict__ mask, u64* __restrict__ out, u32 out_words64) {
    extern __shared__ u32 lds[];
    u32* xs = lds;                                   // BK_TILE_WORDS words of x
    u32* os = lds + BK_TILE_WORDS;                   // the window of y: 2^(wbits - 5) words
    const u32 ow = 1u << (t.wbits - 5);
    const u32 lane = lane_id();
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 nwv = blockDim.x >> 6;
    const u32 nunits = t.nwindows * t.nsplit;
    const u32 tiles_per = (t.ntiles + t.nsplit - 1) / t.nsplit;
    for (u32 unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        const u32 w = unit / t.nsplit, part = unit % t.nsplit;
        const u32 c0 = part * tiles_per, c1 = (c0 + tiles_per < t.ntiles) ? c0 + tiles_per : t.ntiles;
        __syncthreads();                             // the previous unit's flush is done
        for (u32 i = threadIdx.x; i < ow; i += blockDim.x) os[i] = 0u;
        // software pipeline over the unit's non-empty blocks: the loads of block i + 1 (its x tile and every wavefront's
        // first BK_PF trips) are issued BEFORE block i is probed, so the memory system works while the LDS does
        const u32* bo = t.blk_off + (size_t)w * t.ntiles;
        auto next_block = [&](u32 c) { while (c < c1 && bo[c] == bo[c + 1]) ++c; return c; };
        auto load_block = [&](u32 c, uint4 (&xv)[2], bk_u32x4 (&d)[BK_PF]) {
            const u32 bb = bo[c], ntrips = (bo[c + 1] - bb) / BK_CHUNK;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const u32 gi = c * BK_TILE_WORDS + (threadIdx.x + r * 1024u) * 4u;
                xv[r] = make_uint4(0, 0, 0, 0);
                if (gi + 4 <= x_words32) xv[r] = *(const uint4*)(x32 + gi);
                else if (gi < x_words32) {           // ragged end of x
                    xv[r].x = x32[gi];
                    if (gi + 1 < x_words32) xv[r].y = x32[gi + 1];
                    if (gi + 2 < x_words32) xv[r].z = x32[gi + 2];
                }
            }
#pragma unroll
            for (int j = 0; j < BK_PF; ++j) {
                const u32 k = wave + j * nwv;        // (wave-uniform)
                d[j] = *(const bk_u32x4*)(t.entries + bb + (k < ntrips ? k : 0u) * BK_CHUNK + lane * 4);
            }
        };
        u32 c = next_block(c0);
        const bool any = c < c1;
        uint4 xv_n[2];
        bk_u32x4 d_c[BK_PF], d_n[BK_PF];
        if (any) {
            load_block(c, xv_n, d_c);
            __syncthreads();                         // the window is zeroed
#pragma unroll
            for (int r = 0; r < 2; ++r) *(uint4*)(xs + (threadIdx.x + r * 1024u) * 4u) = xv_n[r];
            __syncthreads();
        }
        while (c < c1) {
            const u32 bb = bo[c], ntrips = (bo[c + 1] - bb) / BK_CHUNK;
            const u32 cn = next_block(c + 1);
            if (cn < c1) load_block(cn, xv_n, d_n);  // in flight while block c is probed
#pragma unroll
            for (int j = 0; j < BK_PF; ++j)
                if (wave + j * nwv < ntrips) bk_probe(d_c[j], xs, os, lane);
            for (u32 k0 = wave + BK_PF * nwv; k0 < ntrips; k0 += BK_PF * nwv) {   // big blocks: the rest, BK_PF trips at a time
                bk_u32x4 d[BK_PF];
#pragma unroll
                for (int j = 0; j < BK_PF; ++j) {
                    const u32 k = k0 + j * nwv;
                    d[j] = *(const bk_u32x4*)(t.entries + bb + (k < ntrips ? k : 0u) * BK_CHUNK + lane * 4);
                }
#pragma unroll
                for (int j = 0; j < BK_PF; ++j) asm volatile("" : "+v"(d[j]));   // all loads issued before the first probe
#pragma unroll
                for (int j = 0; j < BK_PF; ++j)
                    if (k0 + j * nwv < ntrips) bk_probe(d[j], xs, os, lane);
            }
            if (cn >= c1) break;
            __syncthreads();                         // every wavefront is done with tile c
#pragma unroll
            for (int r = 0; r < 2; ++r) *(uint4*)(xs + (threadIdx.x + r * 1024u) * 4u) = xv_n[r];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < BK_PF; ++j) d_c[j] = d_n[j];
            c = cn;
        }
        __syncthreads();
        if (any) {
            // flush the window: 64-bit words, AND-NOT the mask, one atomic per non-zero word (a window may be shared by
            // nsplit workgroups; `out` was zeroed by the caller)
            const u32 ow64 = ow >> 1;
            const u64* os64 = reinterpret_cast<const u64*>(os);
            const size_t gbase = (size_t)w * ow64;
            for (u32 i = threadIdx.x; i < ow64; i += blockDim.x) {
                u64 v = os64[i];
                if (gbase + i >= out_words64) v = 0ull;
                if (v && mask) v &= ~mask[gbase + i];
                if (v) atomicOr((unsigned long long*)(out + gbase + i), (unsigned long long)v);
            }
        }
    }
}

void blocked_release(fgpu_ctx* ctx, fgpu_tiles* t) {
    if (!ctx) return;
    ctx->dev_free(t->bk_seg_off);
    ctx->dev_free(t->bk_entries);
}

fgpu_info blocked_mxv(fgpu_ctx* ctx, const fgpu_tiles* t, const u64* x_dev, u32 x_words64, const u64* mask_dev, u64* out_dev) {
    if (t->nentries == 0) return FGPU_OK;
    const u32 ow = 1u << (t->bk_wbits - 5);
    const size_t lds = ((size_t)BK_TILE_WORDS + ow) * sizeof(u32);
    FGPU_REQUIRE((int)lds <= ctx->opt.lds_limit, FGPU_INVALID, "blocked kernel needs %zu B of LDS", lds);
    typedef void (*bk_fn)(BlockedView, const u32*, u32, const u64*, u64*, u32);
    // option "blocked_variant".  Measured (cold pass, RMAT-22 / 24 / 26, fraction of the 8 TB/s peak): 0 (default) = 3 trips in
    // flight, two workgroups per CU 0.49 / 0.56 / 0.57; 1 = 2 trips 0.50 / 0.58 / 0.54; 2 = 4 trips with registers
    // unconstrained (one workgroup per CU) 0.44 / 0.47 / 0.41; 3 = 4 trips squeezed into 64 VGPRs (13 spilled) 0.31 / 0.35 / 0.23
    bk_fn fn = blocked_mxv_kernel<3, 8>;
    u32 max_per_cu = 2;
    switch (ctx->opt.blocked_variant) {
        case 1: fn = blocked_mxv_kernel<2, 8>; break;
        case 2: fn = blocked_mxv_kernel<4, 4>; max_per_cu = 1; break;
        case 3: fn = blocked_mxv_kernel<4, 8>; break;
        default: break;
    }
    if (lds > 48 * 1024)
        FGPU_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BlockedView v{t->bk_seg_off, t->bk_entries, t->bk_wbits, t->bk_nwindows, t->ntiles, t->bk_nsplit};
    u32 per_cu = (u32)(ctx->opt.lds_limit / lds);
    if (per_cu > max_per_cu) per_cu = max_per_cu;    // 2 x 1024 threads fill a CU
    if (per_cu < 1) per_cu = 1;
    u32 grid = ctx->opt.tiled_wgs ? (u32)ctx->opt.tiled_wgs : (u32)ctx->cus * per_cu;
    const u32 nunits = t->bk_nwindows * t->bk_nsplit;
    if (grid > nunits) grid = nunits;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(1024), lds, ctx->stream(), v, (const u32*)x_dev, x_words64 * 2, mask_dev,
                       out_dev, t->ngroups);
    FGPU_HIP(hipGetLastError());
    return FGPU_OK;
}

// fills the blocked members of `t` (t->ngroups is set by the caller)
fgpu_info blocked_build(fgpu_ctx* ctx, const fgpu_mat* m, CsrView mv, fgpu_tiles* t) {
    const u64 nrows = m->nrows, ncols = m->ncols;
    // windows of N / 256 rows, between 2^10 and 2^17 (the entry has 12 bits for the output word)
    u32 wbits = 10;
    while (wbits < BK_MAX_WBITS && (nrows >> wbits) > 256) ++wbits;
    const u32 nwindows = (u32)((nrows + (1ull << wbits) - 1) >> wbits);
    const u32 ntiles = (u32)((ncols + (1ull << BK_TILE_BITS) - 1) >> BK_TILE_BITS);
    const u64 nblocks64 = (u64)nwindows * ntiles;
    FGPU_REQUIRE(nblocks64 * 32 < 0x7FFFFFFFull, FGPU_INVALID, "blocked layout: too many blocks");
    const u32 nblocks = (u32)nblocks64, nbuckets = nblocks * 32u;
    DevBuf<u32> cnt, padded, seg_off;
    FGPU_TRY(cnt.alloc(ctx, (size_t)nbuckets + 1));
    FGPU_TRY(padded.alloc(ctx, (size_t)nbuckets + 1));
    FGPU_TRY(seg_off.alloc(ctx, (size_t)nbuckets + 1));
    FGPU_HIP(hipMemsetAsync(cnt.p, 0, ((size_t)nbuckets + 1) * sizeof(u32), ctx->stream()));
    u32 grid = cdiv(nrows, 4);
    if (grid > (u32)ctx->cus * 32) grid = ctx->cus * 32;
    hipLaunchKernelGGL(bk_scatter_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream(), mv, (u32)nrows, wbits, ntiles, cnt.p,
                       (const u32*)nullptr, (u32*)nullptr);
    hipLaunchKernelGGL(bk_pad_count_kernel, dim3(cdiv((u64)nblocks + 1, 256)), dim3(256), 0, ctx->stream(), (const u32*)cnt.p,
                       nblocks, padded.p);
    FGPU_HIP(hipGetLastError());
    DevBuf<u64> off64;
    FGPU_TRY(off64.alloc(ctx, (size_t)nbuckets + 1));
    FGPU_TRY(scan_u32_to_u64(ctx, padded.p, off64.p, (u64)nbuckets + 1, nullptr));
    u64 total = 0;
    FGPU_TRY(read_u64(ctx, off64.p + nbuckets, &total));
    FGPU_REQUIRE(total < 0xFFFFFFF0ull, FGPU_INVALID, "blocked layout: %llu padded entries exceed the 32-bit offset space",
                 (unsigned long long)total);
    FGPU_TRY(ctx->dev_alloc((void**)&t->bk_seg_off, ((size_t)nblocks + 1) * sizeof(u32)));     // block offsets
    FGPU_TRY(ctx->dev_alloc((void**)&t->bk_entries, (size_t)(total ? total : 4) * sizeof(u32)));
    FGPU_TRY(scan_u32(ctx, padded.p, seg_off.p, (u64)nbuckets + 1, nullptr));                  // the same prefix in 32 bits
    FGPU_HIP(hipMemsetAsync(cnt.p, 0, ((size_t)nbuckets + 1) * sizeof(u32), ctx->stream()));
    hipLaunchKernelGGL(bk_scatter_kernel<true>, dim3(grid), dim3(256), 0, ctx->stream(), mv, (u32)nrows, wbits, ntiles, cnt.p,
                       (const u32*)seg_off.p, t->bk_entries);
    hipLaunchKernelGGL(bk_pad_fill_kernel, dim3(cdiv(nbuckets, 256)), dim3(256), 0, ctx->stream(), (const u32*)cnt.p,
                       (const u32*)seg_off.p, nbuckets, t->bk_entries);
    hipLaunchKernelGGL(bk_block_off_kernel, dim3(cdiv((u64)nblocks + 1, 256)), dim3(256), 0, ctx->stream(), (const u32*)seg_off.p,
                       nblocks, t->bk_seg_off);
    FGPU_HIP(hipGetLastError());
    FGPU_HIP(hipStreamSynchronize(ctx->stream()));
    t->kind = 1;
    t->bk_wbits = wbits;
    t->bk_nwindows = nwindows;
    t->ntiles = ntiles;
    t->tile_bits = BK_TILE_BITS;
    t->nentries = total;
    t->nitems = nblocks;
    t->vec = 4; t->k = 1;
    // units = (window, range of tiles): at least two per CU
    u32 nsplit = 1;
    while (nwindows * nsplit < 2u * (u32)ctx->cus && nsplit < ntiles) nsplit <<= 1;
    if (nsplit > ntiles) nsplit = ntiles;
    t->bk_nsplit = nsplit;
    return FGPU_OK;
}

}  // namespace fgpu
