// common.hpp — context, device pool, error plumbing and the device-side CSR view
// shared by every translation unit of libfgpu.  gfx950 (MI355X) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fgpu.h"

namespace fgpu {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

constexpr int WAVE = 64;  // CDNA4 wavefront

// ---- error plumbing ---------------------------------------------------------
void set_error(const char* fmt, ...);
const char* get_error();

#define FGPU_HIP(expr)                                                                  \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ::fgpu::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                              __FILE__, __LINE__);                                      \
            return (_e == hipErrorOutOfMemory) ? FGPU_OOM : FGPU_DEVICE;                \
        }                                                                               \
    } while (0)

#define FGPU_TRY(expr)                      \
    do {                                    \
        fgpu_info _i = (expr);              \
        if (_i != FGPU_OK) return _i;       \
    } while (0)

#define FGPU_REQUIRE(cond, code, ...)       \
    do {                                    \
        if (!(cond)) {                      \
            ::fgpu::set_error(__VA_ARGS__); \
            return (code);                  \
        }                                   \
    } while (0)

}  // namespace fgpu

// ---- context -----------------------------------------------------------------
// One per process+device, shared by every host thread of the process (the reference calls GraphBLAS from a
// worker pool on shared handles: threadpool.rs:89-128, matrix.rs:781-796).  Each host thread that calls into
// the library is bound to a LANE of the context: its own HIP stream, its own pinned staging block and its own
// free-list of device blocks (hipMalloc is ~100 us; traversal calls must not pay it per hop), so calls of
// different threads never share a stream-ordered resource.  The context owns the host allocator hooks the
// caller handed to fgpu_init, the engine options and the kernel profiler.
struct fgpu_options {  // fgpu_set_option
    int tiled_u = 8;           // items in flight per wavefront of the tiled kernel (8 KiB of entries per wave)
    int tiled_nt = 0;          // nontemporal entry loads
    int tiled_threads = 1024;  // its workgroup size
    int tiled_wgs = 0;         // its grid (0 = one workgroup per CU)
    int expand_mode = 0;       // 0 auto, 1 sorted-CSR products only, 2 bit-parallel from the first hop
    int expand_row_groups = 1; // sparse mid-chain pull: a wavefront per 32-row group (0 = a wavefront per row item)
    int expand_fuse_count = 1; // fgpu_expand_count: the last bit-parallel hop counts its rows in place (0 = separate count pass)
    int expand_bits_ratio = 28; // fgpu_expand, expand_mode 0: a hop goes to bit form when its traversed edges T exceed nnz / ratio
    int blocked_variant = 0;   // blocked.hip kernel variant (trips in flight / workgroups per CU), see blocked_mxv
    int tiled_layout = 0;      // full-pass pull layout: 0 = pick by size, 1 = LDS x tiles + global atomics (tiled.hip), 2 = x tile
                               // and output window both in LDS (blocked.hip)
    int bfs_wgs_per_cu = 6;    // grid of the fused BFS level kernel, workgroups per CU
    int bfs_tiny = 2;          // consecutive tiny BFS levels in one single-workgroup launch (bfs_tiny_kernel): 0 off, 1 on,
                               // 2 = when the plan's previous search took more than 12 levels
    int bfs_hub_first = 1;     // pull levels read A' rows reordered hub-first (bfs.hip ensure_pull_order)
    int bfs_alive_rule = 1;    // push <-> pull rule of the fused BFS: the unvisited share is taken over the vertices that have an in-edge
                               // (0 = over all vertices, rounds 1-5; A/B)
    int bfs_pb = 1;            // heavy push levels by propagation blocking (bfs.hip bfs_pb_*): the frontier's edges are binned by
                               // destination window, a workgroup per window marks its discoveries in LDS — no global atomic per
                               // edge.  0 off, 1 for plans of at least 2^24 vertices (a heavy push level of a smaller graph is a
                               // few tens of microseconds: the four extra launches cost more), 2 for every single-rank plan
    long long bfs_pb_min_edges = 2ll << 20;   // ... a push level with at least this many edges to examine goes that way
    int bfs_prof_split = 0;    // profiled BFS pass launches the <.., 1|2> twins that name a level push / pull (PMC passes)
    int merge_items = 1;       // Delta merge scatter: 1 = shifted copy by 2048-entry items with the dp insertion positions as events
                               // (merge.hip), 0 = the per-word / per-entry scatter (A/B)
    int merge_mode = 0;        // Delta merge: 0 entry-parallel (merge.hip), 1 one wavefront per row (pattern only)
    int dist_timing = 0;       // fgpu_bfs_dist_run records HIP events around every level kernel and exchange (fgpu_bfs_dist_times)
    int dist_collective = 0;   // frontier exchange of the in-library multi-GPU BFS: 0 grouped ncclSend/ncclRecv
                               // (all-gather-v, direct peer-to-peer over xGMI), 1 one ncclBroadcast per rank in a group
    int dist_force_self = 0;   // TEST ONLY: a communicator of one rank still issues the grouped self send / recv, broadcast and
                               // all-reduce of a multi-rank exchange (dist.hip) — the code path on the real librccl of a 1-GPU box
    int dist_test_delay_us = 0; // TEST ONLY: fgpu_bfs_dist_run puts a kernel spinning this many microseconds behind every level kernel of
                               // THIS context's rank (a peer that finishes its levels late; tests/test_gpu_dist.py)
    int transpose_mode = 0;    // pattern transpose / COO build: 0 counting sort (form picked by key space), 1 COO rebuild through the sorter (A/B), 2 LDS-staged levels, 3 two levels
    int lds_limit = 0;         // usable LDS bytes per workgroup (filled by fgpu_init)
    int expand_compact = 1;    // fgpu_expand*: source rows that are empty after the CSR hops (sources without out-edges: half of an
                               // R-MAT batch) are dropped before the chain goes to bits when that halves the row width (0 = keep; A/B)
    int pagerank_parts = 1;    // PageRank SpMV: 1 = A' in 8 column ranges, range k gathered by XCD k out of its own L2 (when the score
                               // vector exceeds one L2), 2 = always, 0 = the one-pass pull over the whole vector (A/B)
    int expand_first_hop = 1;  // fgpu_expand*: a clean first hop from one-entry rows copies the source rows (0 = the general product; A/B)
    int expand_xcd = 1;        // dense count hop of the bit-parallel chain: 1 = the rows of X are gathered by the XCD that owns their
                               // partition, partial rows folded per vertex (bitpart.hip), 0 = every workgroup gathers from all of X (A/B)
    int expand_xcd_relabel = 1; // ... and the state it reads is laid out hot-first per partition by the hop that produces it (0 = vertex order; A/B)
    int expand_xcd_min_mb = 32; // ... when the bit state holds at least this many MiB (8 L2s of 4 MiB; below that the plain pull)
    int expand_scan_min = 2048; // fgpu_expand_count: a call with more source rows than this is a WHOLE-FRONTIER call (spgemm.hip
                               // expand_count_scan): live rows filtered and compacted on the device, cut into passes (0 = never)
    int expand_scan_rows = 1024; // ... live rows per pass: 1024 = 16 words = one 128-byte line per vertex of the bit state
    int expand_scan_lanes = 3;  // ... lanes (calling thread + workers, a stream and pool each) the passes are dealt to
    int expand_records = 1;     // sparse mid-chain pull: rows of X with <= 4 bits are read as 8-byte records of source indices, a lane
                               // per live entry (bitexpand.hip bp_records_kernel; 0 = every live entry gathers the whole row; A/B)
    int expand_nt = 1;          // XCD-partitioned count hop, streaming hints (bit mask): 1 = the partial rows leave the stream kernel with
                               // non-temporal stores (1 GB per pass that would otherwise displace the partition's hot rows of X from its L2:
                               // stream kernel 824 -> 771 us at RMAT-22, no change at RMAT-26), 2 = its column-id stream is read non-temporal,
                               // 4 = the fold reads the partial rows non-temporal (2, 4: no effect, off; profiles/NOTES_r06.md section 7)
    int expand_emit_sort = 1;   // bit state -> CSR: 2 = (row, vertex) pairs in vertex order + the LDS-staged stable sort by row, 0 = the
                               // ballot transpose of rounds 3-5 (bp_rows_kernel), 1 = pairs + sort unless the count pass finds more
                               // than 8 entries per vertex (a dense result: the ballot transpose is 4 x cheaper there)
    int pinned_results = 1;    // result arrays >= 256 KiB come from the context's pinned-host pool and are filled by DMA (0 = the
                               // caller's allocator / malloc + staged copies, the round-3 path; A/B)
    int pinned_pool_mb = 4096; // pinned blocks kept for reuse after fgpu_free (beyond it they go back to the OS)
};

struct fgpu_lane {  // one per host thread using the context
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;        // own_stream, or the stream handed to fgpu_set_stream by this thread
    void* pinned = nullptr;              // small pinned staging block for control read-backs
    size_t pinned_bytes = 0;
    // scalar read-backs (read_u32 / read_u64): a one-thread kernel stores the value and then a sequence number into this
    // mapped pinned line with system-scope stores; the host spins on the sequence word — ~8 us instead of the ~22 us of a
    // runtime D2H copy + hipStreamSynchronize, and a k-hop batch makes a dozen of them
    uint32_t* pub_host = nullptr;
    uint32_t* pub_dev = nullptr;
    uint32_t pub_seq = 0;
    std::multimap<size_t, void*> pool;   // free device blocks by capacity, recycled in this lane's stream order (ctx->mu)
    void* zero_block = nullptr;          // a block of `pool` whose first zero_bytes are known to be zero (dev_free_zeroed); the
    size_t zero_bytes = 0;               // mark is dropped as soon as the block leaves the pool for anything else
    hipEvent_t fence = nullptr;          // recorded on `stream` by a thread that frees a shared object (fence_mu)
    std::mutex fence_mu;
    bool bound = false;                  // a live thread holds it (ctx->mu)
    // transfer staging (h2d / d2h to pageable caller memory): a ring of XFER_SLOTS pinned chunks, allocated on the lane's
    // first bulk transfer; up to XFER_SLOTS - 1 device copies are on the link while the host drains the oldest
    static constexpr int XFER_SLOTS = 4;
    void* xfer = nullptr;
    size_t xfer_half = 0;                // bytes per slot
    hipEvent_t xfer_ev[XFER_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    bool xfer_busy[XFER_SLOTS] = {false, false, false, false};
    int xfer_next = 0;
};

struct fgpu_prof_entry {   // fgpu_prof_enable / fgpu_prof_read: HIP-event pairs around launches of named kernels
    const char* name;
    hipEvent_t e0, e1;
    uint64_t alg_bytes;
};

struct fgpu_ctx {
    int device = 0;
    uint64_t id = 0;           // unique over the process lifetime (thread-local lane caches key on it)
    fgpu_options opt;
    std::atomic<uint64_t> opt_epoch{0};   // bumped by fgpu_set_option: cached BFS plans (fgpu_bfs) are rebuilt when it moved
    const fgpu_mat* bfs_cache_owner = nullptr;   // the adjacency holding this context's one cached fgpu_bfs plan (under bfs_link_mu())
    void* (*mal)(size_t) = nullptr;
    void (*fre)(void*) = nullptr;
    int cus = 0;
    std::mutex mu;                      // lanes, pools, live map, byte counters
    std::vector<fgpu_lane*> lanes;
    std::map<void*, size_t> live;       // capacity of live blocks
    uint64_t bytes_in_use = 0, bytes_pooled = 0;
    // pinned-host result blocks ("outputs are pinned-host buffers owned by the caller until fgpu_free"):
    // hipHostMalloc costs ~0.1 ms per MiB, so freed blocks are kept by capacity and handed out again
    std::multimap<size_t, void*> pin_pool;
    std::map<const void*, size_t> pin_live;   // block start -> capacity, for every block handed out
    uint64_t pin_pooled = 0;
    std::vector<void*> flag_free_list;        // 32 KiB pinned blocks for control words (flag_alloc), kept until fgpu_finalize
    std::vector<void*> flag_all;
    // multi-GPU: this context's rank in an RCCL communicator (dist.hip); nullptr = not part of one
    void* comm = nullptr;               // ncclComm_t
    int comm_rank = 0, comm_nranks = 1;
    std::atomic<uint64_t> dist_self_calls{0};   // forced self collectives issued (dist_force_self)
    std::atomic<int> scan_active{0};            // lanes of whole-frontier calls running now (expand_count_scan)
    std::atomic<uint32_t> bfs_cp_last{0};   // ... and the fused launches of that search that ran behind bfs_pb_list_kernel ("bfs_cp_last_mask")
    std::atomic<uint32_t> bfs_pb_last{0};   // levels the search fgpu_bfs_stats last read ran by propagation blocking ("bfs_pb_last_levels")
    std::atomic<uint32_t> scan_last_live{0}, scan_last_passes{0};   // the last such call: live source rows, passes ("expand_scan_*")
    std::atomic<uint64_t> expand_launches{0};   // kernels launched by fgpu_expand* (fgpu_get_option "expand_kernel_launches")
    // kernel profiler (measurement hook): off unless fgpu_prof_enable(ctx, 1)
    bool prof_on = false;
    std::mutex prof_mu;
    std::vector<fgpu_prof_entry> prof;
    std::vector<hipEvent_t> prof_free;  // recycled events

    fgpu_lane* lane();                  // the calling thread's lane (bound on first use; makes ctx->device current)
    hipStream_t stream() { return lane()->stream; }
    void* pinned() { return lane()->pinned; }
    bool multi_lane();                  // more than one lane has ever been bound
    // Order every LATER operation of the calling thread's stream after all work queued so far on every other
    // lane: called before a shared object's (snapshot, plan) blocks return to the calling lane's free-list.
    void fence_lanes();
    // A handle about to be returned to the caller may be handed to another thread: when several lanes exist,
    // wait for the work that produced it (single-lane contexts stay asynchronous).
    fgpu_info publish();
    fgpu_info dev_alloc(void** p, size_t bytes);
    // a block whose first `bytes` are zero: the lane's marked block when it fits exactly (no memset), else a fresh block
    // that *was_zero = false tells the caller to clear.  dev_free_zeroed returns a block the caller has re-zeroed (on this
    // lane's stream) and marks it — a k-hop batch hands its 2 GiB bit state to the next batch this way instead of a memset.
    fgpu_info dev_alloc_zeroed(void** p, size_t bytes, bool* was_zero);
    void dev_free_zeroed(void* p, size_t bytes);
    void dev_free(void* p);
    // Bulk transfers between CALLER memory and the device always go through the lane's pinned halves: a
    // hipMemcpyAsync on pageable memory makes the runtime pin / unpin the caller's pages, and once a process had
    // freed and re-used large result buffers a 4 KiB upload was seen to take 27-35 ms that way.
    // h2d returns once `host` has been consumed (the device copy is ordered on the lane's stream);
    // d2h / d2h_widen return with the data in `host` (they wait for the stream).
    fgpu_info h2d(void* dev, const void* host, size_t bytes);
    fgpu_info d2h(void* host, const void* dev, size_t bytes);
    fgpu_info d2h_widen(uint64_t* host, const uint32_t* dev, size_t count);   // u32 on the device, u64 for the caller
    // d2h / d2h_widen take the direct route — one DMA into `host`, the widening done by a kernel — when `host` is pinned
    // (a block of the pool below, or memory the caller registered with HIP); pageable memory goes through the staging ring.
    bool dma_able(const void* host, size_t bytes);
    void* host_alloc(size_t bytes);     // the caller's allocator (small results, control data)
    void* result_alloc(size_t bytes);   // result arrays: pinned pool from 256 KiB up, host_alloc below
    void* pinned_alloc(size_t bytes);   // a block of the pinned pool (nullptr: out of memory)
    void host_free(void* p);            // releases either kind
    // a 32 KiB block of pinned host memory for control words a device writes and a host thread polls (a search plan's done
    // flag, its control block copy).  Pooled for the life of the context: hipHostMalloc / hipHostFree synchronise the whole
    // device inside the runtime, and a thread sitting in one while another thread polls its stream (hipStreamQuery) is how
    // three query threads driving BFS plans stopped (tools/experiments/bfs_threads_hang.py) — so no plan frees host memory
    void* flag_alloc();
    void flag_release(void* p);
    void trim();
};

namespace fgpu {
// RAII: HIP events around one kernel launch (or a short sequence) when the ctx profiler is on.
struct ProfScope {
    fgpu_ctx* ctx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const char* name;      // static string
    uint64_t bytes;        // algorithmic bytes of the launch (SURVEY.md §8d accounting), 0 when not modelled
    int* idx_out = nullptr;  // receives the record's index (or -1) for prof_add_bytes
    ProfScope(fgpu_ctx* c, const char* n, uint64_t alg_bytes);
    ~ProfScope();
};
// add bytes to a record whose size is only known after a later read-back (e.g. the non-zero output rows of a hop)
void prof_add_bytes(fgpu_ctx* ctx, int idx, uint64_t extra);
// out_dev[i] = in_dev[i] for i < n on the calling lane's stream (u32 ids of the device -> the caller's u64 GrB_Index)
fgpu_info widen_on_device(fgpu_ctx* ctx, uint64_t* out_dev, const uint32_t* in_dev, size_t n);
}  // namespace fgpu

namespace fgpu {

// RAII device buffer from the ctx pool.
template <typename T>
struct DevBuf {
    fgpu_ctx* ctx = nullptr;
    T* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : ctx(o.ctx), p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            release();
            ctx = o.ctx; p = o.p; n = o.n;
            o.p = nullptr; o.n = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    fgpu_info alloc(fgpu_ctx* c, size_t count) {
        release();
        ctx = c;
        n = count;
        void* q = nullptr;
        fgpu_info i = c->dev_alloc(&q, (count ? count : 1) * sizeof(T));
        if (i != FGPU_OK) { n = 0; return i; }
        p = (T*)q;
        return FGPU_OK;
    }
    void release() {
        if (p && ctx) ctx->dev_free(p);
        p = nullptr; n = 0;
    }
    T* take() { T* q = p; p = nullptr; n = 0; return q; }
};

}  // namespace fgpu

// ---- matrix snapshot -----------------------------------------------------------
// Immutable CSR on device.  `hrows == nullptr`: rowptr has nrows+1 entries.
// Hypersparse (delta layers, me): hrows = sorted ids of the nvec non-empty rows and
// rowptr has nvec+1 entries; row lookups binary-search hrows.
struct fgpu_tiles;  // tiled.hip: LDS-staged frontier-tile edge layout (acceleration index)

// The stored matrix (dims, rowptr, colidx, vals, hrows) never changes after creation.  The acceleration indexes
// below are built lazily on first use; every such build happens under `idx_mu`, is complete on the device
// (stream synchronised) before its pointer is published, and is never replaced afterwards — concurrent readers
// on other lanes either see no index (and take the lock) or a finished one.
namespace fgpu { struct BpXPlan; struct PrParts; }   // bitexpand.hip, pagerank.hip
struct fgpu_mat {
    fgpu_ctx* ctx = nullptr;
    mutable std::mutex idx_mu;
    uint64_t nrows = 0, ncols = 0, nnz = 0;
    uint32_t nvec = 0;          // number of stored rows (== nrows when not hyper)
    uint32_t* rowptr = nullptr; // device
    uint32_t* colidx = nullptr; // device
    uint64_t* vals = nullptr;   // device, nullable (BOOL: pattern only)
    uint32_t* hrows = nullptr;  // device, nullable
    // static hub list for the push kernels (rows with degree >= HUB_DEG), device
    mutable uint32_t* hub_chunks = nullptr;  // triples (row, begin, end)
    mutable uint32_t n_hub_chunks = 0;
    // finer list for the fused push levels (rows >= PUSH_HUB_DEG in PUSH_HUB_CHUNK-edge items): a frontier of a
    // few hundred near-hub rows must spread over the whole chip, not over frontier/4 workgroups
    mutable uint32_t* push_chunks = nullptr;
    mutable uint32_t n_push_chunks = 0;
    mutable uint32_t max_deg = 0;
    mutable std::atomic<bool> finalized{false};       // hub list / max_deg computed (mat_finalize); merges leave it to the first BFS plan
    mutable uint32_t* pull_col = nullptr; // bfs.hip: column ids with every row reordered hub-first, for the pull levels (lazy, owned)
    mutable uint32_t* wordrow = nullptr;  // merge.hip: stored-row index of entry 64 w, for w in [0, ceil(nnz/64)] (lazy, owned)
    mutable fgpu_tiles* tiles = nullptr;  // built on demand by fgpu_mat_build_tiles; owned by the matrix
    // fgpu_bfs (the one-call entry): the plan of the last (this, At) search is kept on the adjacency so that repeated calls
    // do not pay plan creation (pinned allocations, events, head array: 1.3 ms of a 2.2 ms call at RMAT-22).  `bfs_mu`
    // serialises its users; the transpose remembers which adjacency holds a plan over it (one at a time) so that releasing
    // either matrix drops the plan first.  The links are only changed under bfs_link_mu(); a context keeps ONE cached plan
    // (fgpu_ctx::bfs_cache_owner): caching one on another adjacency drops the previous one.
    mutable std::mutex bfs_mu;
    mutable struct fgpu_bfs_plan* bfs_plan = nullptr;
    mutable const fgpu_mat* bfs_plan_at = nullptr;
    mutable uint64_t bfs_plan_epoch = 0;
    mutable struct fgpu_ctx* bfs_plan_ctx = nullptr;   // the context whose one cached plan this is
    mutable const fgpu_mat* bfs_cached_in = nullptr;
    // bit-parallel expansion (bitexpand.hip): cached pattern transpose of this matrix, and (on that
    // transpose) its rows cut into items of <= 256 entries
    mutable fgpu_mat* tcache = nullptr;
    mutable uint32_t* bp_items = nullptr;  // triples (row, begin, end | split << 31)
    mutable uint32_t n_bp_items = 0;
    mutable uint32_t* bp_sitems = nullptr; // the items of split rows only (rows of more than BP_ITEM entries)
    mutable uint32_t n_bp_sitems = 0;
    mutable uint64_t* bp_split_bits = nullptr;  // on the cached transpose: bit v set <=> row v is cut into several items
    mutable fgpu::PrParts* pr_parts = nullptr;  // pagerank.hip: this matrix split into 8 column ranges (one per XCD), lazily, owned
    mutable fgpu::BpXPlan* bp_xplan = nullptr; // on the cached transpose: the XCD-partitioned layout of the dense count hop
                                               // (bitpart.hip, built on the first such hop; released by bp_xplan_release)
    bool is_hyper() const { return hrows != nullptr; }
};

// Column-tiled edge layout of a matrix M for y = M (x) x over the boolean semiring with the x tile
// staged in LDS (tiled.hip).  Tile c holds the entries whose column lies in
// [c << tile_bits, (c+1) << tile_bits); inside a tile entries are grouped by 64 consecutive rows
// (one output word) and cut into items of at most 64*vec*k entries.
struct fgpu_tiles {
    fgpu_ctx* ctx = nullptr;
    uint32_t tile_bits = 0, ntiles = 0, ngroups = 0, nitems = 0;
    uint32_t vec = 4, k = 1;         // entries per lane per load, loads per lane per item
    uint64_t nentries = 0;           // padded
    uint32_t* item_off = nullptr;    // nitems + 1
    uint32_t* item_group = nullptr;  // nitems
    uint32_t* entries = nullptr;     // packed: bits 0..25 column - tile base (bit tile_bits = pad), 26..31 row & 63
    uint32_t* tile_item = nullptr;   // ntiles + 1 (device)
    uint64_t* row_has = nullptr;     // ngroups words: bit set = row has at least one entry
    // blocked layout (blocked.hip: output windows in LDS as well; what scales past RMAT-22): kind == 1
    uint32_t kind = 0;
    uint32_t bk_wbits = 0, bk_nwindows = 0, bk_nsplit = 1;
    uint32_t* bk_seg_off = nullptr;  // nblocks + 1 block offsets (entries, multiples of 256)
    uint32_t* bk_entries = nullptr;
};

namespace fgpu {

// POD view passed to kernels by value.
struct CsrView {
    const u32* rowptr;
    const u32* colidx;
    const u32* hrows;  // nullable
    u32 nvec;
    u32 nrows;
};

inline CsrView view_of(const fgpu_mat* m) {
    CsrView v;
    v.rowptr = m->rowptr;
    v.colidx = m->colidx;
    v.hrows = m->hrows;
    v.nvec = m->nvec;
    v.nrows = (u32)m->nrows;
    return v;
}

// [begin,end) of row r in the view (device).
__device__ __forceinline__ void row_range(const CsrView& a, u32 r, u32& b, u32& e) {
    if (a.hrows == nullptr) {
        if (r < a.nrows) { b = a.rowptr[r]; e = a.rowptr[r + 1]; }
        else { b = 0; e = 0; }
        return;
    }
    u32 lo = 0, hi = a.nvec;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if (a.hrows[mid] < r) lo = mid + 1; else hi = mid;
    }
    if (lo < a.nvec && a.hrows[lo] == r) { b = a.rowptr[lo]; e = a.rowptr[lo + 1]; }
    else { b = 0; e = 0; }
}

__device__ __forceinline__ u32 lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__host__ __device__ __forceinline__ u64 mix64(u64 z) {  // splitmix64 finalizer
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// The order-independent checksum of a k-hop result (fgpu_expand_count / fgpu_expand_levels) is the sum over its
// (row, dest) entries of row_hash(row) * dest_hash(dest) mod 2^64.  The product form (instead of one hash of the packed
// pair) is what lets the bit-parallel state evaluate it without touching single bits: per vertex the row hashes of a
// 64-bit word are summed through 16 nibble look-ups in LDS, then multiplied once by the vertex' hash — the hash of
// every single entry was 27 % of a 3-hop batch (4.2 of 15.6 ms at RMAT-24).  dest_hash is odd, so no row sum is lost.
__host__ __device__ __forceinline__ u64 cs_row_hash(u64 row) { return mix64(row); }
__host__ __device__ __forceinline__ u64 cs_dest_hash(u64 dest) { return mix64(dest ^ 0x9e3779b97f4a7c15ull) | 1ull; }

inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// ---- primitives (prims.hip) ------------------------------------------------------
// Exclusive prefix sums on device; `total` (nullable device pointer) receives the sum.
fgpu_info scan_u32(fgpu_ctx* ctx, const u32* in, u32* out, u64 n, u32* total_dev);
fgpu_info scan_u32_to_u64(fgpu_ctx* ctx, const u32* in, u64* out, u64 n, u64* total_dev);
// Sort every segment [off[s], off[s+1]) of `data` ascending and drop duplicates in
// place; cnt[s] receives the number of unique keys left at the front of the segment.
// `key_bound` = exclusive upper bound of the keys (bitmap path sizing).
// `dirty` (nullable): segments with dirty[s]==0 are skipped (cnt[s] pre-filled by the caller).
fgpu_info segsort_unique(fgpu_ctx* ctx, u32* data, const u64* off, u32 nseg, u32 key_bound,
                         u32* cnt, const uint8_t* dirty);
// Gather the unique prefixes into a dense CSR; rowptr = exclusive scan of cnt (nseg+1 entries).
fgpu_info compact_segments(fgpu_ctx* ctx, const u32* data, const u64* off, const u32* rowptr,
                           u32 nseg, u32* col_out);

// ---- matrix helpers (mat.hip) ---------------------------------------------------
// free a snapshot no other thread has seen (temporaries, failed builds); fgpu_mat_free adds the cross-lane fence
void mat_release(fgpu_mat* m);
void bp_xplan_release(fgpu_ctx* ctx, BpXPlan* p);   // bitpart.hip
void pr_parts_release(fgpu_ctx* ctx, PrParts* p);   // pagerank.hip
void mat_drop_bfs_plan(const fgpu_mat* a);   // caller holds bfs_link_mu() and a->bfs_mu
// One process-wide mutex orders every change of the (adjacency <-> transpose) plan links (bfs_plan / bfs_plan_at /
// bfs_cached_in / fgpu_ctx::bfs_cache_owner): taken BEFORE any matrix' bfs_mu, released before a search runs.
std::mutex& bfs_link_mu();
fgpu_info mat_alloc(fgpu_ctx* ctx, fgpu_mat** out, u64 nrows, u64 ncols, u64 nnz, bool with_vals,
                    u32 nvec_hyper, bool hyper);
// (m \ dm) U dp, pattern only, on device (K3/K6).
fgpu_info mat_merge_device(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp,
                           const fgpu_mat* dm, bool dm_masks_dp);
// The same merge walked entry-parallel, pattern or UINT64 values (dp's value wins), optionally onto
// new dims (entries at or past them are dropped): merge.hip.
fgpu_info mat_merge_entries(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp,
                            const fgpu_mat* dm, bool dm_masks_dp, u64 out_nrows, u64 out_ncols,
                            bool pattern_only);
fgpu_info mat_from_device_coo_vals(fgpu_ctx* ctx, fgpu_mat** out, u64 nrows, u64 ncols, const u32* rows,
                                   const u32* cols, const u64* vals, u64 n);
fgpu_info mat_transpose_vals(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a);
// dense (nrows+1) rowptr of a possibly hypersparse matrix.
fgpu_info dense_rowptr(fgpu_ctx* ctx, const fgpu_mat* a, DevBuf<u32>& rp);
fgpu_info mat_transpose_pattern(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a);
// transpose.hip: the sort-free builders (stable two-level counting sort); FGPU_NO_VALUE = not applicable, fall back
// stored-row index of entry 64 w of a snapshot, w in [0, ceil(nnz / 64)] (merge.hip; lazy, owned by the snapshot)
fgpu_info mat_wordrow(fgpu_ctx* ctx, const fgpu_mat* a, const uint32_t** out);
void ks_set_wb_override(int wb);   // experiment knob: low-digit bits of the counting sort (0 = pick)
fgpu_info mat_transpose_counting(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a);
// stable sort of n (key, value) pairs by key < nkeys: out_val = the values in key order, keyptr[nkeys + 1] = start of every key's run
fgpu_info sort_u32_pairs_by_key(fgpu_ctx* ctx, const u32* key, const u32* val, u64 n, u64 nkeys, u32* out_val, u32* keyptr);
// stable partition of the entries of a CSR by slot / div (slot = slot_of[column], or the column): (slot, row) pairs, part by part
fgpu_info partition_csr_entries(fgpu_ctx* ctx, const u32* colidx, const u32* rowptr, u32 nrows, u64 nnz, const u32* slot_of, u32 div,
                                u32 nparts, uint2* out_pairs, u32* pstart_dev);
fgpu_info mat_from_device_coo_counting(fgpu_ctx* ctx, fgpu_mat** out, u64 nrows, u64 ncols, const u32* rows,
                                       const u32* cols, u64 n);
fgpu_info mat_finalize(const fgpu_mat* m);  // hub list, max degree (after rowptr/colidx are filled; private or under idx_mu)
fgpu_info mat_ensure_finalized(const fgpu_mat* m);  // lazily, for snapshots produced by the merge kernels
// device COO (u32 rows / cols, n entries) -> CSR snapshot, duplicates collapsed.
fgpu_info mat_from_device_coo(fgpu_ctx* ctx, fgpu_mat** out, u64 nrows, u64 ncols, const u32* rows,
                              const u32* cols, u64 n);
// read one u32 / u64 from device on the ctx stream (synchronises).
fgpu_info read_u32(fgpu_ctx* ctx, const u32* dev, u32* host);
fgpu_info read_words(fgpu_ctx* ctx, const u32* dev, int nwords, u32* host);   // up to 14 consecutive 32-bit words, one round trip
// the two halves of read_words for a kernel that publishes its own result (saves the separate one-thread dispatch):
// pub_begin hands out the lane's mapped line and the sequence number the kernel must store — the words first, then
// `seq` into word 15 with a system-scope release — or returns false when the lane has no mapped line (use read_words then);
// pub_wait spins until that sequence number shows and copies the words out
bool pub_begin(fgpu_ctx* ctx, u32** dst_dev, u32* seq);
fgpu_info pub_wait(fgpu_ctx* ctx, u32 seq, int nwords, u32* host);
fgpu_info read_u64(fgpu_ctx* ctx, const u64* dev, u64* host);

// dist.hip: frontier exchange over the context's communicator.  Rank r's `counts[r]` words live at `buf + offs[r]` on
// every rank after the call; this rank's own piece is `send` (copied into place on the stream).  u64 words.
fgpu_info comm_allgatherv_u64(fgpu_ctx* ctx, const u64* send, u64* buf, const u64* offs, const u64* counts);
fgpu_info comm_allreduce_sum_u32(fgpu_ctx* ctx, u32* buf, u64 n);
fgpu_info comm_group_begin();   // ncclGroupStart / End around the calls of several ranks driven by one thread
fgpu_info comm_group_end();

void tiles_release(fgpu_tiles* t);
// tiled.hip: build the LDS-tile layout of `m` (0 = automatic parameter) / run out = m (x) x & ~mask
fgpu_info tiles_build(fgpu_ctx* ctx, const fgpu_mat* m, int tile_bits, int vec, int k, bool rebuild);
fgpu_info blocked_build(fgpu_ctx* ctx, const fgpu_mat* m, CsrView mv, fgpu_tiles* t);
fgpu_info blocked_mxv(fgpu_ctx* ctx, const fgpu_tiles* t, const u64* x_dev, u32 x_words64, const u64* mask_dev, u64* out_dev);
void blocked_release(fgpu_ctx* ctx, fgpu_tiles* t);
fgpu_info tiles_mxv(fgpu_ctx* ctx, const fgpu_tiles* t, const u64* x_dev, u32 x_words64, const u64* mask_dev,
                    u64* out_dev, bool zero_out);

// ---- bit-parallel expansion (bitexpand.hip) ------------------------------------------
struct BitState {
    DevBuf<u64> x;  // n rows of `ws` words: bit i of row v set <=> (i, v) in F
    u32 n = 0;      // vertices (rows of X)
    u32 nsrc = 0;   // source rows of F
    u32 w = 0;      // words in use per vertex
    u32 ws = 0;     // row stride in words (power of two <= 64, or a multiple of 64)
    DevBuf<uint8_t> flag;  // one byte per vertex: != 0 => row v may hold a set bit (lets a hop skip empty rows)
    u64 nz_rows = 0;       // number of flagged rows (what decides the sparse / dense form of the next hop)
    // traversed-edge count of the NEXT hop over `pre_for` (sum popcount(X[v]) * deg(v)), when the hop that produced X
    // summed it on the way (bp_pull_groups_kernel): saves the pass over the non-zero rows bp_flops would make
    const fgpu_mat* pre_for = nullptr;
    u64 pre_flops = 0;
    // rows of x are defined only where flag != 0 (bp_from_csr of a light frontier zeroes just the rows it scatters into
    // instead of the whole 2 GiB state; every reader of such a state goes through the flags)
    bool lazy = false;
    // a chain whose EMPTY source rows were dropped before it went to bits (spgemm.hip compact_source_rows): bit i stands for
    // source row rowmap[i] — the checksum's row hashes and the emitted row pointers need to know (nullptr: bit i = row i)
    // row v of the state lives at slot perm[v] (nullptr: at v): the layout the XCD-partitioned count hop gathers from
    // (bitpart.hip BpXPlan::perm, owned by the next hop's matrix); only the state that FEEDS a count hop is ever permuted
    const u32* perm = nullptr;
    DevBuf<u32> rowmap;
    DevBuf<u32> rowrank;   // nsrc_full + 1 entries: live rows before source row i (the way back for chains that emit rows)
    u32 nsrc_full = 0;     // source rows before the compaction (0: not compacted)
};
fgpu_info bp_from_csr(fgpu_ctx* ctx, BitState& s, const fgpu_mat* f);
// one hop from a CSR frontier into bit form by pushing (the hop at which a chain leaves the sorted-CSR products)
fgpu_info bp_push_from_csr(fgpu_ctx* ctx, BitState& s, const fgpu_mat* f, const fgpu_mat* m, const fgpu_mat* dp,
                           const fgpu_mat* dm);
// `next_m` (nullable): the base matrix of the hop after this one, if it will run in bit form too
// `count_next` (nullable): the base matrix of the NEXT hop when that hop is the counting end of the chain (bp_hop_count) — the
// state is then written in the layout its partitioned form gathers from
fgpu_info bp_hop(fgpu_ctx* ctx, BitState& s, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm, u64* flops,
                 const fgpu_mat* next_m = nullptr, const fgpu_mat* count_next = nullptr);
fgpu_info bp_to_csr(fgpu_ctx* ctx, const BitState& s, const u64* label_dev, fgpu_mat** out);
// the last hop of an all-pinned batch read off the state: one bit per row (bitexpand.hip)
fgpu_info bp_probe_rows(fgpu_ctx* ctx, const BitState& s, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm,
                        const u32* dst_dev, const u32* bit_dev, const u32* sdst_dev, const u32* srow_dev, u32 k,
                        uint8_t* hit_m, uint8_t* hit_dm, uint8_t* hit_dp);
// nnz + order-independent checksum of the result read straight from the bit state (fgpu_expand_count)
fgpu_info bp_count(fgpu_ctx* ctx, const BitState& s, const u64* label_dev, u64* nnz, u64* checksum);
// the LAST hop of a count-only chain: the hop and the count in one pass — the result rows are counted where they are
// produced and never written (only rows that several work items or a delta layer touch go through a side buffer)
fgpu_info bp_hop_count(fgpu_ctx* ctx, BitState& s, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm, u64* flops,
                       const u64* label_dev, u64* nnz, u64* checksum);
// u |= x (u is allocated, zeroed, on first use): DISTINCT union over the hops of a [*1..k] pattern
fgpu_info bp_accumulate(fgpu_ctx* ctx, BitState& u, const BitState& x);
// The chain is done with `s`: its block goes back to the pool, re-zeroed by its flagged rows and marked "zero" when that is
// cheaper than the memset the next batch's state would otherwise pay (bp_recycle_state).
void bp_finish(fgpu_ctx* ctx, BitState& s);

constexpr u32 HUB_DEG = 4096;    // rows at least this long are expanded by the hub kernel
constexpr u32 HUB_CHUNK = 4096;  // edges per hub work item
constexpr u32 PUSH_HUB_DEG = 1024;    // fused push levels: rows at least this long come from push_chunks ...
constexpr u32 PUSH_HUB_CHUNK = 1024;  // ... in items of one workgroup trip (256 threads x 4 edges)

}  // namespace fgpu
