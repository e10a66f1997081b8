// stub_rccl.hip — TEST INFRASTRUCTURE ONLY: an in-process loop-back implementation of the eleven RCCL entry points
// libfgpu.so binds (falkordb_amd/csrc/dist.hip), so that the MULTI-RANK branch of the frontier exchange —
// fgpu_comm_init_all, grouped ncclSend / ncclRecv per peer pair, ncclBroadcast per rank, the ncclAllReduce of the degree
// vectors, group nesting for a single-process gang — runs on a ONE-GPU box against the oracle (tests/test_gpu_dist.py).
// "Ranks" are communicators of one process, possibly on the same device; data moves with hipMemcpyAsync between their
// streams, ordered by events exactly where RCCL's semantics order it (a receive completes after the matching send's
// stream reached the send; a sender may reuse its buffer once its stream passed the send).  Selected with
// FGPU_RCCL_LIB=<path of this .so>; nothing under falkordb_amd/ links or loads it otherwise.
//
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tests/stub_rccl/stub_rccl.hip -o tests/stub_rccl/libstub_rccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct World {
    int n = 0;
    std::vector<struct StubComm*> comms;
};
struct StubComm {
    World* w = nullptr;
    int rank = 0, dev = 0;
};
struct Op {
    enum Kind { SEND, RECV, ALLREDUCE, BCAST } kind;
    StubComm* c;
    const void* send;
    void* recv;
    size_t count;
    ncclDataType_t type;
    int peer;   // SEND / RECV: the other rank; BCAST: the root
    hipStream_t st;
    bool done = false;
};

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
std::mutex g_mu;
long g_counters[4] = {0, 0, 0, 0};   // executed sends, broadcasts, all-reduces, group flushes (read by the test)

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

__global__ void add_u32_kernel(unsigned* acc, const unsigned* x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc[i] += x[i];
}

// src's stream has reached the producing point -> dst's stream copies -> src's stream may go on once the copy is done
bool ordered_copy(StubComm* sc, hipStream_t sst, const void* sbuf, StubComm* dc, hipStream_t dst, void* dbuf, size_t bytes) {
    if (bytes == 0) return true;
    hipEvent_t e1 = nullptr, e2 = nullptr;
    if (hipSetDevice(sc->dev) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) return false;
    bool ok = hipEventRecord(e1, sst) == hipSuccess;
    ok = ok && hipSetDevice(dc->dev) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamWaitEvent(dst, e1, 0) == hipSuccess;
    ok = ok && hipMemcpyAsync(dbuf, sbuf, bytes, hipMemcpyDefault, dst) == hipSuccess;
    ok = ok && hipEventRecord(e2, dst) == hipSuccess;
    ok = ok && hipSetDevice(sc->dev) == hipSuccess;
    ok = ok && hipStreamWaitEvent(sst, e2, 0) == hipSuccess;
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    return ok;
}

ncclResult_t flush() {
    std::vector<Op> ops;
    ops.swap(t_ops);
    std::lock_guard<std::mutex> g(g_mu);
    g_counters[3] += 1;
    // point-to-point: every send meets the receive posted by its peer for it (in posting order per pair)
    for (Op& s : ops) {
        if (s.kind != Op::SEND || s.done) continue;
        Op* r = nullptr;
        for (Op& c : ops)
            if (c.kind == Op::RECV && !c.done && c.c->w == s.c->w && c.c->rank == s.peer && c.peer == s.c->rank) { r = &c; break; }
        if (!r || r->count != s.count || r->type != s.type) return ncclInvalidUsage;   // (a real RCCL would hang here)
        if (!ordered_copy(s.c, s.st, s.send, r->c, r->st, r->recv, s.count * type_bytes(s.type))) return ncclUnhandledCudaError;
        s.done = r->done = true;
        g_counters[0] += 1;
    }
    for (Op& c : ops)
        if (c.kind == Op::RECV && !c.done) return ncclInvalidUsage;
    // broadcasts: per (world, root) the root's send buffer goes to every caller's receive buffer
    for (Op& root : ops) {
        if (root.kind != Op::BCAST || root.done || root.c->rank != root.peer) continue;
        for (Op& o : ops) {
            if (o.kind != Op::BCAST || o.done || o.c->w != root.c->w || o.peer != root.peer || &o == &root) continue;
            if (!ordered_copy(root.c, root.st, root.send, o.c, o.st, o.recv, root.count * type_bytes(root.type))) return ncclUnhandledCudaError;
            o.done = true;
        }
        if (root.recv != root.send &&
            hipMemcpyAsync(root.recv, root.send, root.count * type_bytes(root.type), hipMemcpyDefault, root.st) != hipSuccess)
            return ncclUnhandledCudaError;
        root.done = true;
        g_counters[1] += 1;
    }
    for (Op& c : ops)
        if (c.kind == Op::BCAST && !c.done) return ncclInvalidUsage;   // a rank called without its root in the group
    // all-reduce (sum of uint32: the slab-local out-degree vectors): all ranks of the world must be in the group
    for (Op& first : ops) {
        if (first.kind != Op::ALLREDUCE || first.done) continue;
        std::vector<Op*> parts;
        for (Op& o : ops)
            if (o.kind == Op::ALLREDUCE && !o.done && o.c->w == first.c->w) parts.push_back(&o);
        if ((int)parts.size() != first.c->w->n || first.type != ncclUint32) return ncclInvalidUsage;
        const size_t bytes = first.count * 4;
        unsigned* tmp = nullptr;
        if (hipSetDevice(first.c->dev) != hipSuccess || hipMalloc((void**)&tmp, bytes ? bytes : 4) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpyAsync(tmp, first.send, bytes, hipMemcpyDefault, first.st) != hipSuccess) return ncclUnhandledCudaError;
        for (Op* o : parts) {
            if (o == &first) continue;
            hipEvent_t e = nullptr;
            (void)hipSetDevice(o->c->dev);
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, o->st) != hipSuccess) return ncclUnhandledCudaError;
            (void)hipSetDevice(first.c->dev);
            if (hipStreamWaitEvent(first.st, e, 0) != hipSuccess) return ncclUnhandledCudaError;
            (void)hipEventDestroy(e);
            hipLaunchKernelGGL(add_u32_kernel, dim3(256), dim3(256), 0, first.st, tmp, (const unsigned*)o->send, first.count);
        }
        for (Op* o : parts) {
            if (!ordered_copy(first.c, first.st, tmp, o->c, o->st, o->recv, bytes)) return ncclUnhandledCudaError;
            o->done = true;
        }
        for (Op* o : parts) (void)hipStreamSynchronize(o->st);   // (test-only library: tmp is freed right away)
        (void)hipFree(tmp);
        g_counters[2] += 1;
    }
    return ncclSuccess;
}

ncclResult_t post(Op op) {
    t_ops.push_back(op);
    return t_depth == 0 ? flush() : ncclSuccess;
}

}  // namespace

extern "C" {

int fgpu_stub_rccl_loopback() { return 1; }   // dist.hip: ranks of this "RCCL" may share a device
void fgpu_stub_rccl_counters(long out[4]) {
    std::lock_guard<std::mutex> g(g_mu);
    memcpy(out, g_counters, sizeof(g_counters));
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    memcpy(id->internal, "fgpu-stub-rccl", 15);
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    World* w = new World();
    w->n = ndev;
    for (int i = 0; i < ndev; ++i) {
        StubComm* c = new StubComm();
        c->w = w; c->rank = i; c->dev = devlist ? devlist[i] : i;
        w->comms.push_back(c);
        comms[i] = reinterpret_cast<ncclComm_t>(c);
    }
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId, int rank) {
    if (!comm || nranks != 1 || rank != 0) return ncclInvalidUsage;   // one process = one world: use ncclCommInitAll for more
    int dev = 0;
    (void)hipGetDevice(&dev);
    return ncclCommInitAll(comm, 1, &dev);
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    StubComm* c = reinterpret_cast<StubComm*>(comm);
    if (!c) return ncclSuccess;
    World* w = c->w;
    for (auto& x : w->comms)
        if (x == c) x = nullptr;
    bool any = false;
    for (auto* x : w->comms) any = any || x != nullptr;
    if (!any) delete w;
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    return --t_depth == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    StubComm* c = reinterpret_cast<StubComm*>(comm);
    if (!c || peer < 0 || peer >= c->w->n || peer == c->rank || !type_bytes(datatype)) return ncclInvalidArgument;
    return post(Op{Op::SEND, c, sendbuff, nullptr, count, datatype, peer, stream});
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    StubComm* c = reinterpret_cast<StubComm*>(comm);
    if (!c || peer < 0 || peer >= c->w->n || peer == c->rank || !type_bytes(datatype)) return ncclInvalidArgument;
    return post(Op{Op::RECV, c, nullptr, recvbuff, count, datatype, peer, stream});
}
ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream) {
    StubComm* c = reinterpret_cast<StubComm*>(comm);
    if (!c || root < 0 || root >= c->w->n || !type_bytes(datatype)) return ncclInvalidArgument;
    return post(Op{Op::BCAST, c, sendbuff, recvbuff, count, datatype, root, stream});
}
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    StubComm* c = reinterpret_cast<StubComm*>(comm);
    if (!c || op != ncclSum) return ncclInvalidArgument;
    return post(Op{Op::ALLREDUCE, c, sendbuff, recvbuff, count, datatype, 0, stream});
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "success";
        case ncclInvalidUsage: return "stub RCCL: unmatched or unsupported call pattern";
        case ncclInvalidArgument: return "stub RCCL: invalid argument";
        case ncclUnhandledCudaError: return "stub RCCL: HIP call failed";
        default: return "stub RCCL: error";
    }
}

}  // extern "C"
