// tiled.hip — boolean y = M (x) x with the x (frontier) tile staged in LDS.
//
// This is the full-matrix form of the GrB_vxm / GrB_mxv the reference's BFS issues
// (graph/src/graph/graphblas/mod.rs:11173-11193; LAGr_BreadthFirstSearch_Extended called from
// graph/src/runtime/functions/algo_procedures.rs:1079-1088): for M = A' it is the pull step
// q<!visited> = A' x q, every stored entry examined once.
//
// Why a second layout next to CSR.  A CSR pull gathers one frontier bit per entry from a
// 512 KiB..8 MiB bitmap: each gather moves a whole cache line L2 -> L1, so the pass is bound by
// L2 line traffic (measured 852 GB/s of algorithmic bytes on RMAT-22), not by HBM.  Here the
// entries are regrouped by COLUMN TILE of 2^tile_bits ids (default 2^20 = a 128 KiB bitmap tile,
// which fits the 160 KiB LDS of a CDNA4 CU).  A workgroup stages its tile of x in LDS once and
// then only streams packed 32-bit entries from HBM (coalesced 16 B per lane); every frontier
// probe is an LDS read.  Inside a tile the entries are grouped by 64 consecutive rows (= one
// 64-bit output word) and cut into items of <= 64*vec*k entries, so R-MAT hub rows spread over
// wavefronts; a wavefront owns whole items, ORs hit bits into a per-lane 64-bit accumulator,
// reduces it across the wave with DPP and publishes one atomicOr per item.
//
// Entry packing: see pack_entry below (32 bit, the same 4 B/entry as a CSR column index).  Inside a
// (tile, 64-row group) slot the entries of rows 0..31 come first, padded to a multiple of `vec`, then
// those of rows 32..63, padded likewise: the `vec` entries a lane loads at once always belong to the same
// half of the output word, so the half is selected once per load instead of once per entry.
#include "common.hpp"

namespace fgpu {

struct TilesView {
    const u32* item_off;
    const u32* item_group;
    const u32* entries;
    const u32* tile_item;
    const u64* row_has;
    u32 tile_bits, ntiles, ngroups, nitems;
};

static TilesView view_of(const fgpu_tiles* t) {
    TilesView v;
    v.item_off = t->item_off; v.item_group = t->item_group; v.entries = t->entries;
    v.tile_item = t->tile_item; v.row_has = t->row_has;
    v.tile_bits = t->tile_bits; v.ntiles = t->ntiles; v.ngroups = t->ngroups; v.nitems = t->nitems;
    return v;
}

void tiles_release(fgpu_tiles* t) {
    if (!t) return;
    if (t->ctx) {
        t->ctx->dev_free(t->item_off);
        t->ctx->dev_free(t->item_group);
        t->ctx->dev_free(t->entries);
        t->ctx->dev_free(t->tile_item);
        t->ctx->dev_free(t->row_has);
        blocked_release(t->ctx, t);
    }
    delete t;
}

// ---------------------------------------------------------------------------------
// build: CSR (rows sorted, columns ascending) -> tiled items
// ---------------------------------------------------------------------------------
// entries[0 .. PAD_HEAD) are padding entries: lanes beyond the end of an item load them instead of
// branching around the load, which keeps every load of a trip in flight together
constexpr u32 PAD_HEAD = 4;

// Packed entry (one per stored element, 4 B as in CSR): the three fields sit where the probe needs them so that
// each costs ONE instruction to extract —
//   bits  2..17  byte offset of the frontier word inside the LDS tile   (e & 0x3FFFC  -> ds_read_b32 address)
//   bits 18..22  bit inside that word                                   (e >> 18      -> v_bfe_u32 offset, low 5 bits used)
//   bits 26..31  row inside the 64-row group                            (e >> 26      -> shift amount, bit 31 = upper half)
// The kernel is VALU-issue bound (a wave64 instruction holds a 16-lane SIMD for 4 cycles), so instructions per
// entry, not bytes, set its speed.  A padding entry addresses the zero word kept past the tile.
__host__ __device__ __forceinline__ u32 pack_entry(u32 col_in_tile, u32 row_in_group) {
    return ((col_in_tile >> 5) << 2) | ((col_in_tile & 31u) << 18) | (row_in_group << 26);
}

__device__ __forceinline__ u32 lower_bound_col(const u32* __restrict__ col, u32 lo, u32 hi, u64 key) {
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if ((u64)col[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(256) void tiles_count_kernel(CsrView a, u32 nrows, u32 ngroups, u32 tile_bits,
                                                         u32 ntiles, u32 vec, u32 cap, u32* __restrict__ cnt_e,
                                                         u32* __restrict__ cnt_i, u64* __restrict__ row_has) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    for (u32 g = wave; g < ngroups; g += nwaves) {
        const u32 r = (g << 6) + lane;
        u32 rb = 0, re = 0;
        if (r < nrows) { rb = a.rowptr[r]; re = a.rowptr[r + 1]; }
        const u64 has = __ballot(re > rb);
        if (lane == 0) row_has[g] = has;
        u32 lo = rb;
        for (u32 c = 0; c < ntiles; ++c) {
            const u32 hi = (c + 1 == ntiles) ? re : lower_bound_col(a.colidx, lo, re, (u64)(c + 1) << tile_bits);
            const u32 total_lo = wave_sum_u32(lane < 32 ? hi - lo : 0u);
            const u32 total_hi = wave_sum_u32(lane < 32 ? 0u : hi - lo);
            if (lane == 0) {
                const u32 padded = (total_lo + vec - 1) / vec * vec + (total_hi + vec - 1) / vec * vec;
                cnt_e[(size_t)c * ngroups + g] = padded;
                cnt_i[(size_t)c * ngroups + g] = (padded + cap - 1) / cap;
            }
            lo = hi;
        }
    }
}

__global__ __launch_bounds__(256) void tiles_fill_kernel(CsrView a, u32 nrows, u32 ngroups, u32 tile_bits, u32 ntiles,
                                                        u32 vec, u32 cap, const u64* __restrict__ eoff,
                                                        const u32* __restrict__ ioff, u32* __restrict__ entries,
                                                        u32* __restrict__ item_off, u32* __restrict__ item_group) {
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256) >> 6;
    const u32 pad_lo = pack_entry(1u << tile_bits, 0);    // probes the zero word past the tile: never a hit
    const u32 pad_hi = pack_entry(1u << tile_bits, 32);   // the same, carrying the upper-half flag
    for (u32 g = wave; g < ngroups; g += nwaves) {
        const u32 r = (g << 6) + lane;
        u32 rb = 0, re = 0;
        if (r < nrows) { rb = a.rowptr[r]; re = a.rowptr[r + 1]; }
        u32 lo = rb;
        for (u32 c = 0; c < ntiles; ++c) {
            const u32 hi = (c + 1 == ntiles) ? re : lower_bound_col(a.colidx, lo, re, (u64)(c + 1) << tile_bits);
            const u32 cnt = hi - lo;
            u32 inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u32 y = __shfl_up(inc, d, 64);
                if (lane >= (u32)d) inc += y;
            }
            const u32 total = __shfl(inc, 63, 64);
            const u32 total_lo = __shfl(inc, 31, 64);                 // entries of rows 0..31
            const u32 lo_padded = (total_lo + vec - 1) / vec * vec;   // rows 32..63 start on a `vec` boundary
            const size_t slot = (size_t)c * ngroups + g;
            const u32 base = (u32)eoff[slot] + PAD_HEAD;
            const u32 padded = (u32)(eoff[slot + 1] - eoff[slot]);
            const u32 pos = base + inc - cnt + (lane < 32 ? 0u : lo_padded - total_lo);
            const u32 cbase = c << tile_bits;
            for (u32 j = 0; j < cnt; ++j) entries[pos + j] = pack_entry(a.colidx[lo + j] - cbase, lane);
            for (u32 j = total_lo + lane; j < lo_padded; j += 64) entries[base + j] = pad_lo;
            for (u32 j = lo_padded + (total - total_lo) + lane; j < padded; j += 64) entries[base + j] = pad_hi;
            const u32 ib = ioff[slot], ni = ioff[slot + 1] - ib;
            for (u32 kk = lane; kk < ni; kk += 64) {
                item_off[ib + kk] = base + kk * cap;
                item_group[ib + kk] = g;
            }
            lo = hi;
        }
    }
}

__global__ void tiles_finish_kernel(const u32* __restrict__ ioff, const u64* __restrict__ eoff, u32 ngroups,
                                    u32 ntiles, u32 tile_bits, u32* __restrict__ tile_item,
                                    u32* __restrict__ item_off, u32* __restrict__ entries) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c <= ntiles) tile_item[c] = ioff[(size_t)c * ngroups];
    if (c == 0) {
        const size_t n = (size_t)ntiles * ngroups;
        item_off[ioff[n]] = (u32)eoff[n] + PAD_HEAD;
    }
    if (c < PAD_HEAD) entries[c] = pack_entry(1u << tile_bits, 0);
}

// ---------------------------------------------------------------------------------
// y = M (x) x, x tile in LDS
// ---------------------------------------------------------------------------------
__device__ __forceinline__ u32 dpp_or_row(u32 v) {
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    return v;
}

// OR over the 64 lanes of a fully active wavefront; the result is wave-uniform (SGPR).
__device__ __forceinline__ u32 wave_or_u32(u32 v) {
    v = dpp_or_row(v);
    return (u32)__builtin_amdgcn_readlane((int)v, 0) | (u32)__builtin_amdgcn_readlane((int)v, 16) |
           (u32)__builtin_amdgcn_readlane((int)v, 32) | (u32)__builtin_amdgcn_readlane((int)v, 48);
}

// OR-reduction of N (power of two, <= 16) per-lane values over a fully active wavefront, all at once.
// Reduce-scatter over the low log2(N) lane bits — at the step for bit b a lane keeps the values whose index
// has bit b equal to its own lane bit and ORs in its partner's copies of them — then a plain butterfly over the
// remaining lane bits.  Afterwards every lane L holds the complete OR of value (L mod N).
// Cost ~ 3 N + 2 (6 - log2 N) + ... instructions against ~15 N for N independent reductions.
template <int N>
__device__ __forceinline__ u32 wave_or_many(u32 (&v)[N], u32 lane) {
    int n = N;
#pragma unroll
    for (int b = 0; (1 << b) < N; ++b) {
        const bool up = (lane >> b) & 1u;
        n >>= 1;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            if (i < n) {
                // values 2i (bit b of its index = 0 after renumbering) and 2i+1 (bit = 1)
                const u32 keep = up ? v[2 * i + 1] : v[2 * i];
                const u32 send = up ? v[2 * i] : v[2 * i + 1];
                v[i] = keep | (u32)__shfl_xor((int)send, 1 << b, 64);
            }
        }
    }
    u32 r = v[0];
#pragma unroll
    for (int b = 0; b < 6; ++b)
        if ((1 << b) >= N) r |= (u32)__shfl_xor((int)r, 1 << b, 64);
    return r;
}

typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
template <int V> struct EntryVec;
template <> struct EntryVec<1> { typedef u32 type; };
template <> struct EntryVec<2> { typedef u32x2 type; };
template <> struct EntryVec<4> { typedef u32x4 type; };

template <int V>
__device__ __forceinline__ void unpack(const typename EntryVec<V>::type& d, u32 (&e)[V]);
template <> __device__ __forceinline__ void unpack<1>(const u32& d, u32 (&e)[1]) { e[0] = d; }
template <> __device__ __forceinline__ void unpack<2>(const u32x2& d, u32 (&e)[2]) { e[0] = d.x; e[1] = d.y; }
template <> __device__ __forceinline__ void unpack<4>(const u32x4& d, u32 (&e)[4]) {
    e[0] = d.x; e[1] = d.y; e[2] = d.z; e[3] = d.w;
}

// V entries per lane per load, K loads per lane per item (item <= 64*V*K entries), U items in
// flight per wavefront.  `mask` (nullable): output is AND-NOTed with it and groups whose rows are
// all masked or empty are skipped without touching their entries.
template <int V, int K, int U, bool NT>
__global__ __launch_bounds__(1024) void tiled_mxv_kernel(TilesView t, const u64* __restrict__ x, u32 x_words32,
                                                        const u64* __restrict__ mask, u64* __restrict__ out,
                                                        u32 wgs_per_tile) {
    extern __shared__ u32 xs[];
    typedef typename EntryVec<V>::type vec_t;
    const u32 lane = lane_id();
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 nwv = blockDim.x >> 6;
    const u32 tw32 = 1u << (t.tile_bits - 5);  // 32-bit words in one x tile
    constexpr u32 CH = 16;  // items per chunk (multiple of every U)
    const u32* __restrict__ x32 = (const u32*)x;
    const u32 nvirt = t.ntiles * wgs_per_tile;
    for (u32 vb = blockIdx.x; vb < nvirt; vb += gridDim.x) {
        const u32 c = vb % t.ntiles;
        const u32 wg = vb / t.ntiles;
        const u32 it0 = t.tile_item[c], it1 = t.tile_item[c + 1];
        if (it0 == it1) continue;  // block-uniform
        __syncthreads();           // previous tile fully consumed
        for (u32 i = threadIdx.x * 4; i < tw32; i += blockDim.x * 4) {
            const u32 gi = c * tw32 + i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (gi + 4 <= x_words32) v = *(const uint4*)(x32 + gi);
            *(uint4*)(xs + i) = v;
        }
        if (threadIdx.x < 4) xs[tw32 + threadIdx.x] = 0;  // the word padding entries probe
        __syncthreads();
        // A wavefront takes chunks of CH consecutive items.  One coalesced load brings the CH+1
        // offsets, CH group ids (and mask words) into lanes 0..CH; the items are walked with
        // v_readlane, U at a time (U*K vector loads in flight); item j's result word is parked in
        // lane j and the whole chunk is published by ONE vector atomicOr.  Headers of the next
        // chunk are fetched while the current one streams.
        const u32 nchunks = (it1 - it0 + CH - 1) / CH;
        const u32 chstride = wgs_per_tile * nwv;
        u32 ch = wg * nwv + wave;
        u32 off_l = 0, grp_l = 0, cnt = 0;
        u64 m_l = 0ull;
        u32 skip_l = 0;
        if (ch < nchunks) {
            const u32 base = it0 + ch * CH;
            cnt = (it1 - base < CH) ? (it1 - base) : CH;
            off_l = (lane <= cnt) ? t.item_off[base + lane] : 0u;
            grp_l = (lane < cnt) ? t.item_group[base + lane] : 0u;
            if (mask != nullptr && lane < cnt) {
                m_l = mask[grp_l];
                skip_l = ((m_l | ~t.row_has[grp_l]) == ~0ull) ? 1u : 0u;
            }
        }
        while (ch < nchunks) {
            const u32 chn = ch + chstride;
            u32 off_n = 0, grp_n = 0, cnt_n = 0, skip_n = 0;
            u64 m_n = 0ull;
            if (chn < nchunks) {
                const u32 base = it0 + chn * CH;
                cnt_n = (it1 - base < CH) ? (it1 - base) : CH;
                off_n = (lane <= cnt_n) ? t.item_off[base + lane] : 0u;
                grp_n = (lane < cnt_n) ? t.item_group[base + lane] : 0u;
                if (mask != nullptr && lane < cnt_n) {
                    m_n = mask[grp_n];
                    skip_n = ((m_n | ~t.row_has[grp_n]) == ~0ull) ? 1u : 0u;
                }
            }
            u32 res_lo = 0, res_hi = 0;
            for (u32 j0 = 0; j0 < cnt; j0 += U) {
                vec_t d[U][K];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const u32 j = j0 + u;  // <= CH - 1 (CH is a multiple of U)
                    u32 b = (u32)__builtin_amdgcn_readlane((int)off_l, (int)j);
                    u32 e = (u32)__builtin_amdgcn_readlane((int)off_l, (int)j + 1);
                    const u32 sk = (u32)__builtin_amdgcn_readlane((int)skip_l, (int)j);
                    if (j >= cnt) { b = 0; e = 0; }
                    if (sk) e = b;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const u32 q = b + (u32)(k * 64 + lane) * V;
                        // lanes past the item's end read the padding entries at the head of the array:
                        // no branch, so all U*K loads of the trip are in flight together
                        const vec_t* src = (const vec_t*)(t.entries + ((q < e) ? q : 0u));
                        d[u][k] = NT ? __builtin_nontemporal_load(src) : *src;
                    }
                }
                // every load of the trip is issued before the first probe: the empty asm makes the
                // loaded registers opaque here, so no consumer is scheduled above it
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int k = 0; k < K; ++k) asm volatile("" : "+v"(d[u][k]));
                }
                // two 32-bit accumulators per item, 32-bit shifts only: on gfx950 a v_lshlrev_b64 result
                // read by a DPP instruction two wait states later came back wrong under multi-wave
                // occupancy (measured; see DESIGN.md "gfx950 findings").  Per entry: v_and (LDS address),
                // ds_read_b32, v_lshrrev, v_bfe_u32 (hit), v_lshrrev (row), v_lshl_or_b32; per load of V
                // entries: one compare + two selects + two ORs for the half of the output word.
                u32 acc[2 * U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    u32 acc_lo = 0, acc_hi = 0;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        u32 ee[V];
                        unpack<V>(d[u][k], ee);
                        u32 part = 0;
#pragma unroll
                        for (int jj = 0; jj < V; ++jj) {
                            const u32 e = ee[jj];
                            const u32 w = *(const u32*)((const char*)xs + (e & 0x3FFFCu));
                            const u32 hit = __builtin_amdgcn_ubfe(w, e >> 18, 1u);   // offset = low 5 bits
                            part = (hit << ((e >> 26) & 31u)) | part;                // one v_lshl_or_b32
                        }
                        // the V entries of one load share their half of the output word (layout guarantee)
                        const bool upper = (i32)ee[0] < 0;
                        acc_lo |= upper ? 0u : part;
                        acc_hi |= upper ? part : 0u;
                    }
                    acc[2 * u] = acc_lo;       // empty / skipped items carry only padding entries: 0
                    acc[2 * u + 1] = acc_hi;
                }
                // OR-reduce the 2U accumulators over the wavefront TOGETHER: a reduce-scatter (each step
                // halves the values a lane carries while doubling the lanes they cover) instead of 2U
                // independent 6-step reductions; lane L ends up with accumulator (L mod 2U) complete.
                const u32 full = wave_or_many<2 * U>(acc, lane);
                {   // item j0+u's halves are accumulators 2u and 2u+1: park them in lane j0+u
                    const u32 u_of_lane = lane - j0;
                    const bool mine = u_of_lane < (u32)U;
                    const u32 lo = (u32)__shfl((int)full, (int)((2 * u_of_lane) & (2 * U - 1)), 64);
                    const u32 hi = (u32)__shfl((int)full, (int)((2 * u_of_lane + 1) & (2 * U - 1)), 64);
                    res_lo = mine ? lo : res_lo;
                    res_hi = mine ? hi : res_hi;
                }
            }
            const u64 word = (((u64)res_hi << 32) | res_lo) & ~m_l;
            if (word) atomicOr((unsigned long long*)(out + grp_l), (unsigned long long)word);
            ch = chn; off_l = off_n; grp_l = grp_n; cnt = cnt_n; m_l = m_n; skip_l = skip_n;
        }
    }
}

typedef void (*tiled_fn)(TilesView, const u64*, u32, const u64*, u64*, u32);

template <int V, int K, bool NT>
static tiled_fn pick_u(int U) {
    switch (U) {
        case 1: return tiled_mxv_kernel<V, K, 1, NT>;
        case 2: return tiled_mxv_kernel<V, K, 2, NT>;
        case 8: return tiled_mxv_kernel<V, K, 8, NT>;
        default: return tiled_mxv_kernel<V, K, 4, NT>;
    }
}

template <bool NT>
static tiled_fn pick_vk(u32 vec, u32 k, int U) {
    if (vec == 4) return k == 2 ? pick_u<4, 2, NT>(U) : pick_u<4, 1, NT>(U);
    if (vec == 2) return k == 2 ? pick_u<2, 2, NT>(U) : pick_u<2, 1, NT>(U);
    return k == 2 ? pick_u<1, 2, NT>(U) : pick_u<1, 1, NT>(U);
}

static tiled_fn pick_kernel(u32 vec, u32 k, int U, bool nt) {
    return nt ? pick_vk<true>(vec, k, U) : pick_vk<false>(vec, k, U);
}

// out (ngroups words, device) = M (x) x, AND-NOT mask.  `out` is zeroed here.
fgpu_info tiles_mxv(fgpu_ctx* ctx, const fgpu_tiles* t, const u64* x_dev, u32 x_words64, const u64* mask_dev,
                    u64* out_dev, bool zero_out) {
    if (zero_out) FGPU_HIP(hipMemsetAsync(out_dev, 0, (size_t)t->ngroups * sizeof(u64), ctx->stream()));
    if (t->kind == 1) return blocked_mxv(ctx, t, x_dev, x_words64, mask_dev, out_dev);
    if (t->nitems == 0) return FGPU_OK;
    tiled_fn fn = pick_kernel(t->vec, t->k, ctx->opt.tiled_u, ctx->opt.tiled_nt != 0);
    const size_t lds = ((size_t)1 << t->tile_bits) / 8 + 16;
    FGPU_REQUIRE((int)lds <= ctx->opt.lds_limit, FGPU_INVALID,
                 "tiled kernel needs %zu B of LDS per workgroup but the device offers %d", lds, ctx->opt.lds_limit);
    if (lds > 48 * 1024)
        FGPU_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const u32 threads = (u32)ctx->opt.tiled_threads;
    u32 per_cu = (u32)(ctx->opt.lds_limit / lds);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2048 / threads) per_cu = 2048 / threads;
    u32 grid = ctx->opt.tiled_wgs ? (u32)ctx->opt.tiled_wgs : (u32)ctx->cus * per_cu;
    u32 wpt = (grid + t->ntiles - 1) / t->ntiles;  // workgroups sharing one tile
    if (wpt < 1) wpt = 1;
    // no more workgroups per tile than it has item trips
    const u32 items_per_tile = (t->nitems + t->ntiles - 1) / t->ntiles;
    const u32 trip = (threads / 64) * 16u;  // one chunk of items per wavefront
    const u32 max_wpt = (items_per_tile + trip - 1) / trip;
    if (wpt > max_wpt) wpt = max_wpt ? max_wpt : 1;
    const u32 nvirt = wpt * t->ntiles;
    if (grid > nvirt) grid = nvirt;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds, ctx->stream(), view_of(t), x_dev, x_words64 * 2, mask_dev,
                       out_dev, wpt);
    FGPU_HIP(hipGetLastError());
    return FGPU_OK;
}

// Built once per snapshot under its index mutex (first dense-frontier vxm on it); `rebuild` = the measurement hook
// fgpu_mat_build_tiles replacing the layout with other parameters, which needs the snapshot to itself.
fgpu_info tiles_build(fgpu_ctx* ctx, const fgpu_mat* m, int tile_bits, int vec, int k, bool rebuild) {
    std::lock_guard<std::mutex> idx_guard(m->idx_mu);
    if (m->tiles && !rebuild) return FGPU_OK;
    FGPU_REQUIRE(m->nrows >= 1 && m->ncols >= 1, FGPU_INVALID, "tiles: empty matrix");
    // layout choice: explicit parameters ask for the tiled one (the measurement hook's sweeps); otherwise the option, and
    // by default the blocked layout once a (tile, 64-row group) slot of the tiled one would hold fewer than ~128 entries
    {
        const bool explicit_tiled = tile_bits > 0 || vec > 0 || k > 0;
        const double slot = (double)m->nnz / ((double)((m->nrows + 63) / 64) * (double)((m->ncols + (1u << 20) - 1) >> 20));
        const int lay = ctx->opt.tiled_layout;
        if (!explicit_tiled && (lay == 2 || (lay == 0 && slot < 128.0))) {
            CsrView bv = view_of(m);
            DevBuf<u32> drp;
            if (m->is_hyper()) {
                FGPU_TRY(dense_rowptr(ctx, m, drp));
                bv.rowptr = drp.p; bv.hrows = nullptr; bv.nvec = (u32)m->nrows;
            }
            fgpu_tiles* bt = new (std::nothrow) fgpu_tiles();
            FGPU_REQUIRE(bt, FGPU_OOM, "out of host memory");
            bt->ctx = ctx;
            bt->ngroups = (u32)((m->nrows + 63) / 64);
            fgpu_info bi = blocked_build(ctx, m, bv, bt);
            if (bi != FGPU_OK) { tiles_release(bt); return bi; }
            tiles_release(m->tiles);
            m->tiles = bt;
            return FGPU_OK;
        }
    }
    if (tile_bits <= 0) {
        tile_bits = 7;
        while (tile_bits < 20 && ((u64)1 << tile_bits) < m->ncols) ++tile_bits;
        while (tile_bits > 7 && (((size_t)1 << tile_bits) / 8 + 16) > (size_t)ctx->opt.lds_limit) --tile_bits;
    }
    FGPU_REQUIRE(tile_bits >= 7 && tile_bits <= 25, FGPU_INVALID, "tiles: tile_bits %d outside [7,25]", tile_bits);
    FGPU_REQUIRE((((size_t)1 << tile_bits) / 8 + 16) <= (size_t)ctx->opt.lds_limit, FGPU_INVALID,
                 "tiles: a 2^%d-column tile does not fit the %d B of LDS", tile_bits, ctx->opt.lds_limit);
    const u32 ntiles = (u32)((m->ncols + ((u64)1 << tile_bits) - 1) >> tile_bits);
    const u32 ngroups = (u32)((m->nrows + 63) / 64);
    if (vec <= 0) {
        const double span = (double)m->nnz / ((double)ngroups * ntiles);
        vec = span >= 128 ? 4 : (span >= 48 ? 2 : 1);
    }
    if (k <= 0) k = 1;
    FGPU_REQUIRE(vec == 1 || vec == 2 || vec == 4, FGPU_INVALID, "tiles: vec must be 1, 2 or 4");
    FGPU_REQUIRE(k == 1 || k == 2, FGPU_INVALID, "tiles: k must be 1 or 2");
    const u32 cap = 64u * (u32)vec * (u32)k;
    // hypersparse snapshots (delta layers) are walked through a dense row-pointer array
    CsrView mv = view_of(m);
    DevBuf<u32> dense_rp;
    if (m->is_hyper()) {
        FGPU_TRY(dense_rowptr(ctx, m, dense_rp));
        mv.rowptr = dense_rp.p; mv.hrows = nullptr; mv.nvec = (u32)m->nrows;
    }
    const size_t nslots = (size_t)ntiles * ngroups;
    DevBuf<u32> cnt_e, cnt_i, ioff;
    DevBuf<u64> eoff;
    FGPU_TRY(cnt_e.alloc(ctx, nslots + 1));
    FGPU_TRY(cnt_i.alloc(ctx, nslots + 1));
    FGPU_TRY(ioff.alloc(ctx, nslots + 1));
    FGPU_TRY(eoff.alloc(ctx, nslots + 1));
    FGPU_HIP(hipMemsetAsync(cnt_e.p + nslots, 0, sizeof(u32), ctx->stream()));
    FGPU_HIP(hipMemsetAsync(cnt_i.p + nslots, 0, sizeof(u32), ctx->stream()));
    fgpu_tiles* t = new (std::nothrow) fgpu_tiles();
    FGPU_REQUIRE(t, FGPU_OOM, "out of host memory");
    t->ctx = ctx; t->tile_bits = (u32)tile_bits; t->ntiles = ntiles; t->ngroups = ngroups;
    t->vec = (u32)vec; t->k = (u32)k;
    fgpu_info info = FGPU_OK;
    do {
        if ((info = ctx->dev_alloc((void**)&t->row_has, (size_t)ngroups * sizeof(u64))) != FGPU_OK) break;
        if ((info = ctx->dev_alloc((void**)&t->tile_item, ((size_t)ntiles + 1) * sizeof(u32))) != FGPU_OK) break;
        u32 grid = cdiv(ngroups, 4);
        if (grid > (u32)ctx->cus * 32) grid = ctx->cus * 32;
        hipLaunchKernelGGL(tiles_count_kernel, dim3(grid), dim3(256), 0, ctx->stream(), mv, (u32)m->nrows,
                           ngroups, (u32)tile_bits, ntiles, (u32)vec, cap, cnt_e.p, cnt_i.p, t->row_has);
        if (hipGetLastError() != hipSuccess) { set_error("tiles_count launch failed"); info = FGPU_DEVICE; break; }
        if ((info = scan_u32_to_u64(ctx, cnt_e.p, eoff.p, nslots + 1, nullptr)) != FGPU_OK) break;
        if ((info = scan_u32(ctx, cnt_i.p, ioff.p, nslots + 1, nullptr)) != FGPU_OK) break;
        u64 total_e = 0;
        u32 total_i = 0;
        if ((info = read_u64(ctx, eoff.p + nslots, &total_e)) != FGPU_OK) break;
        if ((info = read_u32(ctx, ioff.p + nslots, &total_i)) != FGPU_OK) break;
        if (total_e + PAD_HEAD >= 0xFFFFFFFFull) {
            set_error("tiles: %llu padded entries exceed the 32-bit offset space", (unsigned long long)total_e);
            info = FGPU_INVALID;
            break;
        }
        total_e += PAD_HEAD;
        t->nentries = total_e;
        t->nitems = total_i;
        if ((info = ctx->dev_alloc((void**)&t->entries, (size_t)total_e * sizeof(u32))) != FGPU_OK) break;
        if ((info = ctx->dev_alloc((void**)&t->item_off, ((size_t)total_i + 1) * sizeof(u32))) != FGPU_OK) break;
        if ((info = ctx->dev_alloc((void**)&t->item_group, ((size_t)total_i + 1) * sizeof(u32))) != FGPU_OK) break;
        hipLaunchKernelGGL(tiles_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream(), mv, (u32)m->nrows, ngroups,
                           (u32)tile_bits, ntiles, (u32)vec, cap, (const u64*)eoff.p, (const u32*)ioff.p, t->entries, t->item_off,
                           t->item_group);
        if (hipGetLastError() != hipSuccess) { set_error("tiles_fill launch failed"); info = FGPU_DEVICE; break; }
        hipLaunchKernelGGL(tiles_finish_kernel, dim3(cdiv((u64)ntiles + 1, 64)), dim3(64), 0, ctx->stream(),
                           (const u32*)ioff.p, (const u64*)eoff.p, ngroups, ntiles, (u32)tile_bits, t->tile_item,
                           t->item_off, t->entries);
        if (hipGetLastError() != hipSuccess) { set_error("tiles_finish launch failed"); info = FGPU_DEVICE; break; }
        if (hipStreamSynchronize(ctx->stream()) != hipSuccess) { set_error("tiles build failed"); info = FGPU_DEVICE; break; }
    } while (0);
    if (info != FGPU_OK) { tiles_release(t); return info; }
    tiles_release(m->tiles);
    m->tiles = t;
    return FGPU_OK;
}

}  // namespace fgpu

using namespace fgpu;

extern "C" {

fgpu_info fgpu_mat_build_tiles(fgpu_ctx* ctx, fgpu_mat* m, int tile_bits, int vec, int k) {
    FGPU_REQUIRE(ctx && m, FGPU_NULL_POINTER, "fgpu_mat_build_tiles: NULL argument");
    return tiles_build(ctx, m, tile_bits, vec, k, true);
}

fgpu_info fgpu_mat_tiles_info(const fgpu_mat* m, uint64_t info[8]) {
    FGPU_REQUIRE(m && info, FGPU_NULL_POINTER, "fgpu_mat_tiles_info: NULL argument");
    FGPU_REQUIRE(m->tiles, FGPU_NO_VALUE, "matrix has no tiles");
    const fgpu_tiles* t = m->tiles;
    info[0] = t->tile_bits; info[1] = t->ntiles; info[2] = t->ngroups; info[3] = t->nitems;
    info[4] = t->nentries; info[5] = t->vec; info[6] = t->k;
    info[7] = (uint64_t)t->nentries * 4 + (uint64_t)t->nitems * 8 + (uint64_t)t->ngroups * 8;
    if (t->kind == 1) info[7] = (uint64_t)t->nentries * 4 + (uint64_t)t->nitems * 4;   // entries + block offsets
    return FGPU_OK;
}

}  // extern "C"
