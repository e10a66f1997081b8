// dist.hip — the multi-GPU side of the C ABI: RCCL communicators owned by fgpu contexts, the frontier exchange
// (all-gather-v) of the partitioned BFS, and nnz-balanced slab boundaries.
//
// SURVEY.md §8e / BASELINE.json north_star: "the adjacency matrix row-partitions across the 8 GPUs of one node with an
// RCCL allgatherv of the frontier vector over xGMI each hop".  The reference reaches its BFS through one call
// (LAGr_BreadthFirstSearch_Extended, algo_procedures.rs:1079-1088); a Redis-module process that links libfgpu.so
// must reach the multi-GPU form the same way — so the communicator, the level loop and the collective live in this
// library (fgpu_bfs_dist_run, bfs.hip), not in a Python driver.  Two ways in:
//   * one process per GPU (torchrun, MPI, ...): rank 0 calls fgpu_comm_unique_id, the launcher's own channel carries
//     the 128 bytes to the other ranks, every rank calls fgpu_comm_init_rank;
//   * one process, several GPUs (the Redis module): one context per device, fgpu_comm_init_all.
// xGMI is point-to-point (7 links per GPU): the exchange is a grouped ncclSend / ncclRecv to every peer — each piece
// crosses exactly one link once — not a ring.
#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is bound at first use, see below

#include "common.hpp"

namespace fgpu {

// RCCL is bound with dlopen at the first fgpu_comm_* call instead of a DT_NEEDED entry: a process that also hosts
// PyTorch (bench.py, the tests) already has PyTorch's own librccl.so.1 mapped, and a second copy of RCCL in one
// process aborts at exit (two sets of static state) — dlopen by SONAME hands back the copy that is already there.  A
// process without PyTorch (the Redis module) gets the system's /opt/rocm/lib/librccl.so.1.
namespace {
struct Rccl {
    void* handle = nullptr;
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommInitAll) CommInitAll = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    decltype(&::ncclSend) Send = nullptr;
    decltype(&::ncclRecv) Recv = nullptr;
    decltype(&::ncclBroadcast) Broadcast = nullptr;
    decltype(&::ncclAllReduce) AllReduce = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    bool loopback = false;   // the bound library declares itself an in-process loop-back (tests/stub_rccl): ranks may share a device
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl* rccl() {
    std::call_once(g_rccl_once, [] {
        Rccl& r = g_rccl;
        // FGPU_RCCL_LIB: bind another implementation of the same eleven entry points (a site's own build of RCCL; the
        // tests' in-process loop-back, which lets the multi-rank branch of the exchange run on a one-GPU box)
        const char* forced = getenv("FGPU_RCCL_LIB");
        if (forced && *forced) {
            r.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!r.handle) {
                const char* e = dlerror();                  // (one call: dlerror() clears the message it returns)
                r.error = std::string("FGPU_RCCL_LIB=") + forced + ": " + (e ? e : "dlopen failed");
                return;
            }
            r.loopback = dlsym(r.handle, "fgpu_stub_rccl_loopback") != nullptr;
        }
        for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            if (r.handle) break;
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) {
            const char* e = dlerror();
            r.error = std::string("librccl.so.1 not found: ") + (e ? e : "dlopen failed");
            return;
        }
        bool ok = true;
        auto bind = [&](auto& fn, const char* sym) {
            fn = (std::remove_reference_t<decltype(fn)>)dlsym(g_rccl.handle, sym);
            if (!fn) { ok = false; g_rccl.error = std::string("librccl lacks ") + sym; }
        };
        bind(r.GetUniqueId, "ncclGetUniqueId");
        bind(r.CommInitRank, "ncclCommInitRank");
        bind(r.CommInitAll, "ncclCommInitAll");
        bind(r.CommDestroy, "ncclCommDestroy");
        bind(r.GroupStart, "ncclGroupStart");
        bind(r.GroupEnd, "ncclGroupEnd");
        bind(r.Send, "ncclSend");
        bind(r.Recv, "ncclRecv");
        bind(r.Broadcast, "ncclBroadcast");
        bind(r.AllReduce, "ncclAllReduce");
        bind(r.GetErrorString, "ncclGetErrorString");
        if (!ok) { dlclose(r.handle); r.handle = nullptr; }
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}
}  // namespace

#define FGPU_RCCL_OR_FAIL(R)                                                                  \
    const Rccl* R = rccl();                                                                   \
    if (!R) { ::fgpu::set_error("RCCL unavailable: %s", g_rccl.error.c_str()); return FGPU_DEVICE; }

#define FGPU_NCCL(expr)                                                                                   \
    do {                                                                                                  \
        ncclResult_t _r = (expr);                                                                         \
        if (_r != ncclSuccess) {                                                                          \
            ::fgpu::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?", __FILE__, __LINE__); \
            return FGPU_DEVICE;                                                                           \
        }                                                                                                 \
    } while (0)

// The rank's own words go into its gathered bitmap with a plain kernel: a runtime device-to-device copy between two level
// kernels left the stream idle for ~16 us around a 5 us blit (RMAT-26 trace, one rank), every level.
__global__ __launch_bounds__(256) void own_words_kernel(const u64* __restrict__ send, u64* __restrict__ out, u64 words) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < words; i += (u64)gridDim.x * 256) out[i] = send[i];
}

fgpu_info comm_allgatherv_u64(fgpu_ctx* ctx, const u64* send, u64* buf, const u64* offs, const u64* counts) {
    const int me = ctx->comm_rank, nr = ctx->comm_nranks;
    hipStream_t st = ctx->stream();
    if (counts[me] && send != buf + offs[me]) {   // (in-place plans produce their words where the bitmap keeps them)
        u64 g = (counts[me] + 1023) / 1024;
        if (g > (u64)ctx->cus * 4) g = (u64)ctx->cus * 4;
        hipLaunchKernelGGL(own_words_kernel, dim3((u32)g), dim3(256), 0, st, send, buf + offs[me], counts[me]);
        FGPU_HIP(hipGetLastError());
    }
    // test-only (option dist_force_self): a communicator of ONE rank still issues the grouped calls of a multi-rank
    // exchange, addressed to itself — a self ncclSend / ncclRecv pair inside ncclGroupStart / End, or (dist_collective 1)
    // ncclBroadcast from root 0 — into a scratch buffer that then REPLACES the rank's own words: symbol binding, group
    // nesting and stream ordering run on the real librccl of a 1-GPU box, and the search's result depends on the transfer
    const bool self_test = nr == 1 && ctx->comm && ctx->opt.dist_force_self && counts[me];
    if ((nr == 1 && !self_test) || !ctx->comm) return FGPU_OK;
    FGPU_RCCL_OR_FAIL(R);
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    if (self_test) {
        DevBuf<u64> scratch;
        FGPU_TRY(scratch.alloc(ctx, counts[me]));
        FGPU_HIP(hipMemsetAsync(scratch.p, 0xA5, counts[me] * sizeof(u64), st));
        FGPU_NCCL(R->GroupStart());
        if (ctx->opt.dist_collective == 1) {
            FGPU_NCCL(R->Broadcast((const void*)send, scratch.p, counts[me], ncclUint64, 0, comm, st));
        } else {
            FGPU_NCCL(R->Send(send, counts[me], ncclUint64, 0, comm, st));
            FGPU_NCCL(R->Recv(scratch.p, counts[me], ncclUint64, 0, comm, st));
        }
        FGPU_NCCL(R->GroupEnd());
        u64 g = (counts[me] + 1023) / 1024;
        if (g > (u64)ctx->cus * 4) g = (u64)ctx->cus * 4;
        hipLaunchKernelGGL(own_words_kernel, dim3((u32)g), dim3(256), 0, st, (const u64*)scratch.p, buf + offs[me], counts[me]);
        FGPU_HIP(hipGetLastError());
        ctx->dist_self_calls.fetch_add(1, std::memory_order_relaxed);
        return FGPU_OK;
    }
    FGPU_NCCL(R->GroupStart());
    if (ctx->opt.dist_collective == 1) {
        for (int r = 0; r < nr; ++r)
            if (counts[r])
                FGPU_NCCL(R->Broadcast(r == me ? (const void*)send : (const void*)(buf + offs[r]), buf + offs[r],
                                        counts[r], ncclUint64, r, comm, st));
    } else {
        for (int r = 0; r < nr; ++r) {
            if (r == me) continue;
            if (counts[me]) FGPU_NCCL(R->Send(send, counts[me], ncclUint64, r, comm, st));
            if (counts[r]) FGPU_NCCL(R->Recv(buf + offs[r], counts[r], ncclUint64, r, comm, st));
        }
    }
    FGPU_NCCL(R->GroupEnd());
    return FGPU_OK;
}

fgpu_info comm_allreduce_sum_u32(fgpu_ctx* ctx, u32* buf, u64 n) {
    if ((ctx->comm_nranks == 1 && !ctx->opt.dist_force_self) || !ctx->comm) return FGPU_OK;
    FGPU_RCCL_OR_FAIL(R);
    if (ctx->comm_nranks == 1) ctx->dist_self_calls.fetch_add(1, std::memory_order_relaxed);   // (test-only path: a sum over one rank)
    FGPU_NCCL(R->AllReduce(buf, buf, n, ncclUint32, ncclSum, (ncclComm_t)ctx->comm, ctx->stream()));
    return FGPU_OK;
}

// nested grouping for a single-process gang: all ranks' calls of one exchange are issued by one thread
fgpu_info comm_group_begin() { FGPU_RCCL_OR_FAIL(R); FGPU_NCCL(R->GroupStart()); return FGPU_OK; }
fgpu_info comm_group_end() { FGPU_RCCL_OR_FAIL(R); FGPU_NCCL(R->GroupEnd()); return FGPU_OK; }

// entries per block of 2^shift columns (LDS-privatised: no per-entry global atomic)
__global__ __launch_bounds__(256) void colblock_hist_kernel(const u32* __restrict__ col, u64 nnz, u32 shift, u32 nblocks,
                                                           unsigned long long* __restrict__ hist) {
    extern __shared__ u32 s_h[];
    for (u32 b = threadIdx.x; b < nblocks; b += 256) s_h[b] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (u64)gridDim.x * 256) atomicAdd(&s_h[col[i] >> shift], 1u);
    __syncthreads();
    for (u32 b = threadIdx.x; b < nblocks; b += 256)
        if (s_h[b]) atomicAdd(&hist[b], (unsigned long long)s_h[b]);
}

}  // namespace fgpu

using namespace fgpu;

extern "C" {

fgpu_info fgpu_comm_unique_id(uint8_t* id) {
    FGPU_REQUIRE(id, FGPU_NULL_POINTER, "fgpu_comm_unique_id: NULL id");
    static_assert(sizeof(ncclUniqueId) == FGPU_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    FGPU_RCCL_OR_FAIL(R);
    ncclUniqueId u;
    FGPU_NCCL(R->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return FGPU_OK;
}

fgpu_info fgpu_comm_init_rank(fgpu_ctx* ctx, int nranks, int rank, const uint8_t* id) {
    FGPU_REQUIRE(ctx && id, FGPU_NULL_POINTER, "fgpu_comm_init_rank: NULL argument");
    FGPU_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, FGPU_INVALID, "fgpu_comm_init_rank: bad rank %d / %d", rank, nranks);
    FGPU_REQUIRE(!ctx->comm, FGPU_INVALID, "fgpu_comm_init_rank: the context already has a communicator");
    (void)ctx->lane();   // makes ctx->device current on this thread
    FGPU_RCCL_OR_FAIL(R);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    FGPU_NCCL(R->CommInitRank(&c, nranks, u, rank));
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_nranks = nranks;
    return FGPU_OK;
}

fgpu_info fgpu_comm_init_all(fgpu_ctx* const* ctxs, int n) {
    FGPU_REQUIRE(ctxs && n >= 1, FGPU_INVALID, "fgpu_comm_init_all: need at least one context");
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        FGPU_REQUIRE(ctxs[i] && !ctxs[i]->comm, FGPU_INVALID, "fgpu_comm_init_all: context %d is NULL or already in a communicator", i);
        devs[i] = ctxs[i]->device;
    }
    FGPU_RCCL_OR_FAIL(R);
    if (!R->loopback)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j)
                FGPU_REQUIRE(devs[j] != devs[i], FGPU_INVALID,
                             "fgpu_comm_init_all: contexts %d and %d share device %d (RCCL wants one rank per GPU)", j, i, devs[i]);
    std::vector<ncclComm_t> comms(n, nullptr);
    FGPU_NCCL(R->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        ctxs[i]->comm = comms[i];
        ctxs[i]->comm_rank = i;
        ctxs[i]->comm_nranks = n;
    }
    return FGPU_OK;
}

fgpu_info fgpu_comm_finalize(fgpu_ctx* ctx) {
    FGPU_REQUIRE(ctx, FGPU_NULL_POINTER, "fgpu_comm_finalize: NULL ctx");
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream());
        if (const Rccl* R = rccl()) (void)R->CommDestroy((ncclComm_t)ctx->comm);
    }
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_nranks = 1;
    return FGPU_OK;
}

fgpu_info fgpu_comm_info(fgpu_ctx* ctx, int32_t* rank, int32_t* nranks) {
    FGPU_REQUIRE(ctx, FGPU_NULL_POINTER, "fgpu_comm_info: NULL ctx");
    if (rank) *rank = ctx->comm_rank;
    if (nranks) *nranks = ctx->comm ? ctx->comm_nranks : 1;
    return FGPU_OK;
}

/* ---- the partition arithmetic, host-only and pure: one definition for the library, the launchers and the CPU tests ---- */

uint32_t fgpu_splits_shift(uint64_t ncols) {
    u32 shift = 12;
    while (((ncols + (1ull << shift) - 1) >> shift) > 8192) ++shift;
    return shift;
}

fgpu_info fgpu_balanced_splits_from_hist(const uint64_t* block_counts, uint64_t nblocks, uint32_t shift, uint64_t ncols,
                                         int nparts, uint64_t* splits) {
    FGPU_REQUIRE(splits && (block_counts || nblocks == 0), FGPU_NULL_POINTER, "fgpu_balanced_splits_from_hist: NULL argument");
    FGPU_REQUIRE(nparts >= 1 && shift >= 12 && shift < 40, FGPU_INVALID, "fgpu_balanced_splits_from_hist: bad nparts / shift");
    const u64 top = (((ncols + 4095) >> 12) << 12);   // the padded vertex count every plan uses
    // boundary k sits at the block edge whose entry prefix is nearest to k * nnz / nparts (edges never move backwards)
    std::vector<u64> pre(nblocks + 1, 0);
    for (u64 b = 0; b < nblocks; ++b) pre[b + 1] = pre[b] + block_counts[b];
    const u64 nnz = pre[nblocks];
    splits[0] = 0;
    u64 j = 0;
    for (int k = 1; k < nparts; ++k) {
        const double t = (double)nnz * (double)k / (double)nparts;
        while (j < nblocks && fabs((double)pre[j + 1] - t) <= fabs((double)pre[j] - t)) ++j;
        const u64 edge = j << shift;
        splits[k] = edge < top ? edge : top;
    }
    splits[nparts] = top;
    return FGPU_OK;
}

fgpu_info fgpu_slab_layout(const uint64_t* splits, uint64_t n, int nranks, uint64_t* lo, uint64_t* hi, uint64_t* word_off,
                           uint64_t* word_cnt) {
    FGPU_REQUIRE(nranks >= 1, FGPU_INVALID, "fgpu_slab_layout: nranks must be >= 1");
    // equal slabs (fgpu_bfs_plan_create): ceil(n / nranks) rounded up to 4096 vertices per rank
    const u64 per = (n + (u64)nranks - 1) / (u64)nranks;
    const u64 slab = (per + 4095) & ~4095ull;
    for (int r = 0; r < nranks; ++r) {
        const u64 l = splits ? splits[r] : slab * (u64)r;
        const u64 h = splits ? splits[r + 1] : slab * (u64)(r + 1);
        FGPU_REQUIRE(h >= l && l % 4096 == 0 && h % 4096 == 0, FGPU_INVALID,
                     "fgpu_slab_layout: boundaries must ascend in multiples of 4096");
        if (lo) lo[r] = l;
        if (hi) hi[r] = h;
        if (word_off) word_off[r] = l / 64;
        if (word_cnt) word_cnt[r] = (h - l) / 64;
    }
    return FGPU_OK;
}

fgpu_info fgpu_mat_balanced_splits(fgpu_ctx* ctx, const fgpu_mat* a, int nparts, uint64_t* splits) {
    FGPU_REQUIRE(ctx && a && splits, FGPU_NULL_POINTER, "fgpu_mat_balanced_splits: NULL argument");
    FGPU_REQUIRE(nparts >= 1, FGPU_INVALID, "fgpu_mat_balanced_splits: nparts must be >= 1");
    // in-degree per block of 2^shift columns; boundaries are block boundaries (>= 4096 so that they stay word-aligned
    // for every bitmap and level kernel), chosen where the running entry count crosses k * nnz / nparts
    const u32 shift = fgpu_splits_shift(a->ncols);
    const u32 nblocks = (u32)((a->ncols + (1ull << shift) - 1) >> shift);
    std::vector<unsigned long long> h(nblocks, 0);
    if (a->nnz) {
        DevBuf<unsigned long long> dh;
        FGPU_TRY(dh.alloc(ctx, nblocks));
        FGPU_HIP(hipMemsetAsync(dh.p, 0, (size_t)nblocks * sizeof(unsigned long long), ctx->stream()));
        u32 grid = cdiv(a->nnz, 256 * 64);
        if (grid > (u32)ctx->cus * 8) grid = ctx->cus * 8;
        hipLaunchKernelGGL(colblock_hist_kernel, dim3(grid ? grid : 1), dim3(256), (size_t)nblocks * sizeof(u32), ctx->stream(),
                           (const u32*)a->colidx, (u64)a->nnz, shift, nblocks, dh.p);
        FGPU_HIP(hipGetLastError());
        FGPU_TRY(ctx->d2h(h.data(), dh.p, (size_t)nblocks * sizeof(unsigned long long)));
        FGPU_HIP(hipStreamSynchronize(ctx->stream()));
    }
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "histogram words are 64-bit");
    return fgpu_balanced_splits_from_hist((const uint64_t*)h.data(), nblocks, shift, a->ncols, nparts, splits);
}

}  // extern "C"
