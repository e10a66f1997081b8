// ctx.hip — lifecycle, error strings, device pool.
// Replaces matrix::init / shutdown (reference graph/src/graph/graphblas/matrix.rs:116-221).
#include <stdarg.h>
#include <stdio.h>

#include <iterator>

#include <chrono>

#include "common.hpp"

namespace fgpu {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

// ---- scalar read-backs ----------------------------------------------------------------------------------------------
__global__ void publish_words_kernel(const u32* __restrict__ src, int nwords, u32* __restrict__ dst, u32 seq) {
    if (threadIdx.x == 0) {
        for (int w = 0; w < nwords; ++w) __hip_atomic_store(dst + w, src[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // after the words: release
    }
}

fgpu_info read_words(fgpu_ctx* ctx, const u32* dev, int nwords, u32* host) {   // nwords <= 14
    fgpu_lane* l = ctx->lane();
    if (!l->pub_host) {                                       // (mapping failed at lane creation: the runtime copy)
        FGPU_HIP(hipMemcpyAsync(l->pinned, dev, (size_t)nwords * sizeof(u32), hipMemcpyDeviceToHost, l->stream));
        FGPU_HIP(hipStreamSynchronize(l->stream));
        memcpy(host, l->pinned, (size_t)nwords * sizeof(u32));
        return FGPU_OK;
    }
    u32* dst = nullptr;
    u32 seq = 0;
    (void)pub_begin(ctx, &dst, &seq);
    hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, l->stream, dev, nwords, dst, seq);
    FGPU_HIP(hipGetLastError());
    return pub_wait(ctx, seq, nwords, host);
}

bool pub_begin(fgpu_ctx* ctx, u32** dst_dev, u32* seq) {
    fgpu_lane* l = ctx->lane();
    if (!l->pub_host) return false;
    *seq = ++l->pub_seq ? l->pub_seq : ++l->pub_seq;   // never 0 (the line's initial content)
    *dst_dev = l->pub_dev;
    return true;
}

fgpu_info pub_wait(fgpu_ctx* ctx, u32 seq, int nwords, u32* host) {
    fgpu_lane* l = ctx->lane();
    volatile u32* flag = (volatile u32*)(l->pub_host + 15);
    // spin without touching the stream (a hipStreamQuery costs the next dispatch a system-scope fence); past
    // 2 ms look at the stream now and then so that a failed launch is reported instead of waited for
    const auto t0 = std::chrono::steady_clock::now();
    for (u32 spin = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spin) {
        if ((spin & 0xFFFu) == 0xFFFu &&
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
            const hipError_t q = hipStreamQuery(l->stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
                FGPU_HIP(hipStreamSynchronize(l->stream));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { set_error("scalar read-back: the publish kernel did not run"); return FGPU_DEVICE; }
                break;
            }
            if (q != hipErrorNotReady) { set_error("scalar read-back: stream failed: %s", hipGetErrorString(q)); return FGPU_DEVICE; }
        }
    }
    for (int w = 0; w < nwords; ++w) host[w] = ((volatile u32*)l->pub_host)[w];
    return FGPU_OK;
}

fgpu_info read_u32(fgpu_ctx* ctx, const u32* dev, u32* host) { return read_words(ctx, dev, 1, host); }
fgpu_info read_u64(fgpu_ctx* ctx, const u64* dev, u64* host) {
    u32 w[2] = {0, 0};
    FGPU_TRY(read_words(ctx, (const u32*)dev, 2, w));
    *host = (u64)w[0] | ((u64)w[1] << 32);
    return FGPU_OK;
}

}  // namespace fgpu

using namespace fgpu;

// ---- lanes: one stream + staging block + free-list per host thread -------------------------------------
namespace {

std::mutex g_reg_mu;                        // lock order: g_reg_mu -> ctx->mu -> lane->fence_mu
std::map<uint64_t, fgpu_ctx*> g_live_ctx;   // contexts between fgpu_init and fgpu_finalize
std::atomic<uint64_t> g_next_id{1};

struct LaneRef { uint64_t id; fgpu_ctx* ctx; fgpu_lane* lane; };
struct ThreadLanes {
    std::vector<LaneRef> refs;
    uint64_t last_id = 0;        // the context this thread called last (its device is current)
    fgpu_lane* last_lane = nullptr;
    ~ThreadLanes() {             // the thread exits: its lanes go back to their contexts for the next new thread
        std::lock_guard<std::mutex> g(g_reg_mu);
        for (auto& r : refs) {
            auto it = g_live_ctx.find(r.id);
            if (it == g_live_ctx.end() || it->second != r.ctx) continue;
            std::lock_guard<std::mutex> g2(r.ctx->mu);
            r.lane->bound = false;
        }
    }
};
thread_local ThreadLanes tl_lanes;

fgpu_lane* lane_create() {
    fgpu_lane* l = new (std::nothrow) fgpu_lane();
    if (!l) return nullptr;
    if (hipStreamCreateWithFlags(&l->own_stream, hipStreamNonBlocking) != hipSuccess) { delete l; return nullptr; }
    l->stream = l->own_stream;
    l->pinned_bytes = 1 << 16;
    if (hipHostMalloc(&l->pinned, l->pinned_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipStreamDestroy(l->own_stream);
        delete l;
        return nullptr;
    }
    if (hipEventCreateWithFlags(&l->fence, hipEventDisableTiming) != hipSuccess) {
        (void)hipHostFree(l->pinned);
        (void)hipStreamDestroy(l->own_stream);
        delete l;
        return nullptr;
    }
    // the read-back line (optional: read_words falls back to the runtime copy without it)
    void* pub = nullptr;
    if (hipHostMalloc(&pub, 128, hipHostMallocMapped) == hipSuccess) {
        void* dv = nullptr;
        if (hipHostGetDevicePointer(&dv, pub, 0) == hipSuccess) {
            memset(pub, 0, 128);
            l->pub_host = (uint32_t*)pub;
            l->pub_dev = (uint32_t*)dv;
        } else {
            (void)hipHostFree(pub);
        }
    }
    (void)hipGetLastError();
    return l;
}

void lane_destroy(fgpu_lane* l) {
    if (!l) return;
    if (l->fence) (void)hipEventDestroy(l->fence);
    for (int k = 0; k < fgpu_lane::XFER_SLOTS; ++k)
        if (l->xfer_ev[k]) (void)hipEventDestroy(l->xfer_ev[k]);
    if (l->xfer) (void)hipHostFree(l->xfer);
    if (l->pinned) (void)hipHostFree(l->pinned);
    if (l->pub_host) (void)hipHostFree(l->pub_host);
    if (l->own_stream) (void)hipStreamDestroy(l->own_stream);
    delete l;
}

}  // namespace

fgpu_lane* fgpu_ctx::lane() {
    ThreadLanes& t = tl_lanes;
    if (t.last_id == id) {
        // the process may host other HIP users (PyTorch in bench.py and the tests, a second context of a gang): whoever
        // changed this thread's current device since our last call must not decide where hipMalloc / hipEventCreate land
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device) (void)hipSetDevice(device);
        return t.last_lane;
    }
    (void)hipSetDevice(device);   // HIP's current device is per thread; this thread last called another context (or none)
    for (auto& r : t.refs)
        if (r.id == id) { t.last_id = id; t.last_lane = r.lane; return r.lane; }
    // first call of this thread on this context: adopt a lane an exited thread left behind, or make one
    fgpu_lane* l = nullptr;
    bool others = false;
    {
        std::lock_guard<std::mutex> g(mu);
        for (fgpu_lane* c : lanes)
            if (!c->bound) { l = c; break; }
        if (l) l->bound = true;
        others = !lanes.empty();
    }
    if (!l) {
        l = lane_create();
        if (!l) {
            // out of streams / pinned memory (never seen): degrade to sharing the first lane rather than crash;
            // the sharing threads then need the caller-side serialisation the single-stream design asked for
            std::lock_guard<std::mutex> g(mu);
            l = lanes.empty() ? nullptr : lanes[0];
        } else {
            l->bound = true;
            std::lock_guard<std::mutex> g(mu);
            lanes.push_back(l);
        }
    }
    if (l && others) {
        // joining a context that is already in use: whatever earlier threads queued (snapshots they created and
        // have since handed to this thread) must be complete before this lane's stream reads it
        std::vector<fgpu_lane*> ls;
        { std::lock_guard<std::mutex> g(mu); ls = lanes; }
        for (fgpu_lane* o : ls)
            if (o != l) (void)hipStreamSynchronize(o->stream);
    }
    t.refs.push_back({id, this, l});
    t.last_id = id;
    t.last_lane = l;
    return l;
}

bool fgpu_ctx::multi_lane() {
    std::lock_guard<std::mutex> g(mu);
    return lanes.size() > 1;
}

fgpu_info fgpu_ctx::publish() {
    if (!multi_lane()) return FGPU_OK;
    FGPU_HIP(hipStreamSynchronize(stream()));
    return FGPU_OK;
}

void fgpu_ctx::fence_lanes() {
    fgpu_lane* me = lane();
    std::vector<fgpu_lane*> ls;
    { std::lock_guard<std::mutex> g(mu); if (lanes.size() <= 1) return; ls = lanes; }
    for (fgpu_lane* o : ls) {
        if (o == me) continue;
        std::lock_guard<std::mutex> g(o->fence_mu);
        if (hipEventRecord(o->fence, o->stream) == hipSuccess) (void)hipStreamWaitEvent(me->stream, o->fence, 0);
        else { (void)hipGetLastError(); (void)hipStreamSynchronize(o->stream); }
    }
}

static size_t size_class(size_t bytes) {
    // 256 B granularity below 1 MiB, then 1/8-octave steps: bounded waste, high reuse.
    if (bytes <= 256) return 256;
    if (bytes <= (1u << 20)) return (bytes + 255) & ~(size_t)255;
    size_t p = 1;
    while ((p << 1) < bytes) p <<= 1;  // p < bytes <= 2p
    size_t step = p >> 3;
    return ((bytes + step - 1) / step) * step;
}

fgpu_info fgpu_ctx::dev_alloc(void** p, size_t bytes) {
    size_t cap = size_class(bytes);
    fgpu_lane* ln = lane();
    if (ln) {
        std::lock_guard<std::mutex> g(mu);
        auto it = ln->pool.lower_bound(cap);
        // (a block marked "zero" is left for dev_alloc_zeroed while another one of the same class is free)
        if (it != ln->pool.end() && it->second == ln->zero_block) {
            auto nx = std::next(it);
            if (nx != ln->pool.end() && nx->first <= cap + (cap >> 2)) it = nx;
        }
        if (it != ln->pool.end() && it->first <= cap + (cap >> 2)) {
            *p = it->second;
            size_t c = it->first;
            if (*p == ln->zero_block) { ln->zero_block = nullptr; ln->zero_bytes = 0; }   // about to be overwritten
            ln->pool.erase(it);
            live[*p] = c;
            bytes_pooled -= c;
            bytes_in_use += c;
            return FGPU_OK;
        }
    }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, cap);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        trim();
        e = hipMalloc(&q, cap);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? FGPU_OOM : FGPU_DEVICE;
    }
    std::lock_guard<std::mutex> g(mu);
    live[q] = cap;
    bytes_in_use += cap;
    *p = q;
    return FGPU_OK;
}

fgpu_info fgpu_ctx::dev_alloc_zeroed(void** p, size_t bytes, bool* was_zero) {
    *was_zero = false;
    fgpu_lane* ln = lane();
    if (ln) {
        std::lock_guard<std::mutex> g(mu);
        if (ln->zero_block && ln->zero_bytes >= bytes) {
            const size_t cap = size_class(bytes);
            for (auto it = ln->pool.lower_bound(cap); it != ln->pool.end() && it->first <= cap + (cap >> 2); ++it) {
                if (it->second != ln->zero_block) continue;
                *p = it->second;
                const size_t c = it->first;
                ln->pool.erase(it);
                ln->zero_block = nullptr; ln->zero_bytes = 0;
                live[*p] = c;
                bytes_pooled -= c;
                bytes_in_use += c;
                *was_zero = true;
                return FGPU_OK;
            }
        }
    }
    return dev_alloc(p, bytes);
}

void fgpu_ctx::dev_free_zeroed(void* p, size_t bytes) {
    if (!p) return;
    dev_free(p);
    fgpu_lane* ln = lane();
    if (!ln) return;
    std::lock_guard<std::mutex> g(mu);
    for (auto it = ln->pool.begin(); it != ln->pool.end(); ++it)
        if (it->second == p) { ln->zero_block = p; ln->zero_bytes = bytes; return; }   // (not pooled: freed outright)
}

void fgpu_ctx::dev_free(void* p) {
    if (!p) return;
    fgpu_lane* ln = lane();
    std::lock_guard<std::mutex> g(mu);
    auto it = live.find(p);
    if (it == live.end()) return;
    size_t c = it->second;
    live.erase(it);
    bytes_in_use -= c;
    if (!ln) { (void)hipFree(p); return; }
    // Blocks are recycled stream-ordered inside ONE lane: a block handed out again is only touched by this lane's
    // stream, after its previous users on that stream finished.  Objects other lanes may have touched (snapshots,
    // plans) call fence_lanes() before their blocks come here.
    ln->pool.emplace(c, p);
    bytes_pooled += c;
}

void fgpu_ctx::trim() {
    std::vector<void*> old;
    std::vector<fgpu_lane*> ls;
    {
        std::lock_guard<std::mutex> g(mu);
        for (fgpu_lane* l : lanes) {
            for (auto& kv : l->pool) old.push_back(kv.second);
            l->pool.clear();
            l->zero_block = nullptr; l->zero_bytes = 0;
        }
        bytes_pooled = 0;
        ls = lanes;
    }
    if (old.empty()) return;
    for (fgpu_lane* l : ls) (void)hipStreamSynchronize(l->stream);
    for (void* p : old) (void)hipFree(p);
}

// ---- kernel profiler ----------------------------------------------------------------------------------------
namespace fgpu {
ProfScope::ProfScope(fgpu_ctx* c, const char* n, uint64_t alg_bytes) : ctx(c), name(n), bytes(alg_bytes) {
    if (!c || !c->prof_on) return;
    {
        std::lock_guard<std::mutex> g(c->prof_mu);
        if (c->prof_free.size() >= 2) {
            e0 = c->prof_free.back(); c->prof_free.pop_back();
            e1 = c->prof_free.back(); c->prof_free.pop_back();
        }
    }
    if (!e0) {
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
    }
    (void)hipEventRecord(e0, c->stream());
}
ProfScope::~ProfScope() {
    if (idx_out) *idx_out = -1;
    if (!e0) return;
    (void)hipEventRecord(e1, ctx->stream());
    std::lock_guard<std::mutex> g(ctx->prof_mu);
    if (idx_out) *idx_out = (int)ctx->prof.size();
    ctx->prof.push_back({name, e0, e1, bytes});
}
void prof_add_bytes(fgpu_ctx* ctx, int idx, uint64_t extra) {
    if (idx < 0) return;
    std::lock_guard<std::mutex> g(ctx->prof_mu);
    if ((size_t)idx < ctx->prof.size()) ctx->prof[idx].alg_bytes += extra;
}
}  // namespace fgpu

// ---- staged transfers ---------------------------------------------------------------------------
namespace {
constexpr size_t XFER_HALF = 2u << 20;   // 2 MiB per slot (8 MiB pinned per lane): ~40 us on the link per chunk; a host thread
                                         // copies 2 MiB out in ~200 us, so with three chunks in flight the link is never waited for
constexpr int XS = fgpu_lane::XFER_SLOTS;
constexpr size_t DMA_MIN = 64u << 10;    // below this the staging ring is as fast as a direct DMA and needs no pointer query
constexpr size_t PIN_MIN = 256u << 10;   // result arrays from this size up come from the pinned pool

fgpu_info xfer_ready(fgpu_lane* l) {
    if (l->xfer) return FGPU_OK;
    void* p = nullptr;
    FGPU_HIP(hipHostMalloc(&p, XS * XFER_HALF, hipHostMallocDefault));
    for (int k = 0; k < XS; ++k) {
        if (hipEventCreateWithFlags(&l->xfer_ev[k], hipEventDisableTiming) != hipSuccess) {
            (void)hipHostFree(p);
            set_error("transfer staging: hipEventCreate failed");
            return FGPU_DEVICE;
        }
    }
    l->xfer = p;
    l->xfer_half = XFER_HALF;
    return FGPU_OK;
}

// wait until the device copy that last used slot k has finished
fgpu_info xfer_wait(fgpu_lane* l, int k) {
    if (l->xfer_busy[k]) {
        FGPU_HIP(hipEventSynchronize(l->xfer_ev[k]));
        l->xfer_busy[k] = false;
    }
    return FGPU_OK;
}

// u32 ids of the device -> the u64 ids of the caller (GrB_Index), 4 per thread: one 16-byte load, two 16-byte stores
// `in` is a window of a column array at an arbitrary entry (4-byte aligned only): up to three leading elements go one by one
// so that the 16-byte loads of the body are aligned; the 8-byte stores are aligned wherever the window starts.
__global__ __launch_bounds__(256) void widen_u32_u64_kernel(const u32* __restrict__ in, u64* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t head = ((16u - (u32)(reinterpret_cast<uintptr_t>(in) & 15u)) & 15u) / 4u;
    if (head > n) head = n;
    const size_t n4 = (n - head) / 4;
    const uint4* body = reinterpret_cast<const uint4*>(in + head);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint4 v = body[i];
        u64* o = out + head + 4 * i;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) out[threadIdx.x] = in[threadIdx.x];
        const size_t done = head + 4 * n4;
        if (threadIdx.x < n - done) out[done + threadIdx.x] = in[done + threadIdx.x];
    }
}
}  // namespace

namespace fgpu {
fgpu_info widen_on_device(fgpu_ctx* ctx, u64* out_dev, const u32* in_dev, size_t n) {
    if (!n) return FGPU_OK;
    size_t grid = (n / 4 + 255) / 256;
    if (grid > (size_t)ctx->cus * 16) grid = (size_t)ctx->cus * 16;
    hipLaunchKernelGGL(widen_u32_u64_kernel, dim3(grid ? (u32)grid : 1), dim3(256), 0, ctx->stream(), in_dev, out_dev, n);
    FGPU_HIP(hipGetLastError());
    return FGPU_OK;
}
}  // namespace fgpu

fgpu_info fgpu_ctx::h2d(void* dev, const void* host, size_t bytes) {
    if (bytes == 0) return FGPU_OK;
    fgpu_lane* l = lane();
    FGPU_TRY(xfer_ready(l));
    const char* src = (const char*)host;
    char* dst = (char*)dev;
    for (size_t off = 0; off < bytes; off += l->xfer_half) {
        const size_t n = bytes - off < l->xfer_half ? bytes - off : l->xfer_half;
        const int k = l->xfer_next;
        l->xfer_next = (k + 1) % XS;
        FGPU_TRY(xfer_wait(l, k));
        char* half = (char*)l->xfer + (size_t)k * l->xfer_half;
        memcpy(half, src + off, n);
        FGPU_HIP(hipMemcpyAsync(dst + off, half, n, hipMemcpyHostToDevice, l->stream));
        FGPU_HIP(hipEventRecord(l->xfer_ev[k], l->stream));
        l->xfer_busy[k] = true;
    }
    return FGPU_OK;
}

namespace {
// the common loop of the staged d2h / d2h_widen: `unit` device bytes per element, `emit(slot, first_element, count)`
// moves a finished chunk into the caller's buffer while the next XS - 1 chunks are on the link
template <class Emit>
fgpu_info d2h_loop(fgpu_lane* l, const char* dev, size_t count, size_t unit, Emit emit) {
    if (count == 0) return FGPU_OK;
    FGPU_TRY(xfer_ready(l));
    const size_t per = l->xfer_half / unit;
    struct Pend { int k; size_t first, n; } pend[XS];
    int np = 0, head = 0;
    auto drain_one = [&]() -> fgpu_info {
        Pend& q = pend[head];
        FGPU_TRY(xfer_wait(l, q.k));
        emit((const char*)l->xfer + (size_t)q.k * l->xfer_half, q.first, q.n);
        head = (head + 1) % XS;
        --np;
        return FGPU_OK;
    };
    for (size_t first = 0; first < count; first += per) {
        const size_t n = count - first < per ? count - first : per;
        if (np == XS - 1) FGPU_TRY(drain_one());          // the slot about to be reused is the oldest pending one's successor
        const int k = l->xfer_next;
        l->xfer_next = (k + 1) % XS;
        FGPU_TRY(xfer_wait(l, k));   // (an earlier h2d may still be reading this slot)
        char* half = (char*)l->xfer + (size_t)k * l->xfer_half;
        FGPU_HIP(hipMemcpyAsync(half, dev + first * unit, n * unit, hipMemcpyDeviceToHost, l->stream));
        FGPU_HIP(hipEventRecord(l->xfer_ev[k], l->stream));
        l->xfer_busy[k] = true;
        pend[(head + np) % XS] = {k, first, n};
        ++np;
    }
    while (np) FGPU_TRY(drain_one());
    return FGPU_OK;
}
}  // namespace

bool fgpu_ctx::dma_able(const void* host, size_t bytes) {
    if (!host || bytes < DMA_MIN || !opt.pinned_results) return false;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = pin_live.upper_bound(host);
        if (it != pin_live.begin()) {
            --it;
            const char* b = (const char*)it->first;
            if ((const char*)host >= b && (const char*)host + bytes <= b + it->second) return true;
        }
    }
    // memory the caller pinned itself (hipHostMalloc / hipHostRegister, a torch pinned tensor)
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

fgpu_info fgpu_ctx::d2h(void* host, const void* dev, size_t bytes) {
    if (dma_able(host, bytes)) {
        FGPU_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream()));
        FGPU_HIP(hipStreamSynchronize(stream()));
        return FGPU_OK;
    }
    char* out = (char*)host;
    return d2h_loop(lane(), (const char*)dev, bytes, 1, [&](const char* half, size_t first, size_t n) { memcpy(out + first, half, n); });
}

fgpu_info fgpu_ctx::d2h_widen(uint64_t* host, const uint32_t* dev, size_t count) {
    if (dma_able(host, count * sizeof(u64))) {
        // widened by a kernel, then ONE DMA into the caller's pinned array: no host thread touches the entries
        void* tmp = nullptr;
        FGPU_TRY(dev_alloc(&tmp, count * sizeof(u64)));
        fgpu_info i = fgpu::widen_on_device(this, (u64*)tmp, dev, count);
        if (i == FGPU_OK && hipMemcpyAsync(host, tmp, count * sizeof(u64), hipMemcpyDeviceToHost, stream()) != hipSuccess) {
            set_error("d2h_widen: hipMemcpyAsync failed");
            i = FGPU_DEVICE;
        }
        if (i == FGPU_OK && hipStreamSynchronize(stream()) != hipSuccess) { set_error("d2h_widen: stream failed"); i = FGPU_DEVICE; }
        dev_free(tmp);
        return i;
    }
    return d2h_loop(lane(), (const char*)dev, count, sizeof(u32), [&](const char* half, size_t first, size_t n) {
        const u32* s = (const u32*)half;
        u64* d = host + first;
        for (size_t i = 0; i < n; ++i) d[i] = s[i];
    });
}

void* fgpu_ctx::host_alloc(size_t bytes) {
    if (bytes == 0) bytes = 8;
    return mal ? mal(bytes) : malloc(bytes);
}

constexpr size_t FLAG_BLOCK = 32768;
void* fgpu_ctx::flag_alloc() {
    {
        std::lock_guard<std::mutex> g(mu);
        if (!flag_free_list.empty()) {
            void* p = flag_free_list.back();
            flag_free_list.pop_back();
            memset(p, 0, FLAG_BLOCK);
            return p;
        }
    }
    (void)lane();                                            // (device current)
    void* p = nullptr;
    if (hipHostMalloc(&p, FLAG_BLOCK, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    memset(p, 0, FLAG_BLOCK);
    std::lock_guard<std::mutex> g(mu);
    flag_all.push_back(p);
    return p;
}
void fgpu_ctx::flag_release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu);
    flag_free_list.push_back(p);
}

void* fgpu_ctx::pinned_alloc(size_t bytes) {
    if (bytes == 0) bytes = 8;
    // capacity classes: eighths of a power of two (<= 12.5 % slack), so that results of similar size share blocks
    size_t cap = PIN_MIN;
    while (cap < bytes) cap <<= 1;
    if (cap > PIN_MIN) {
        const size_t step = cap >> 4;                       // sixteenths of cap = eighths of cap / 2
        cap = ((bytes + step - 1) / step) * step;
    }
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = pin_pool.lower_bound(cap);
        // (k-hop results of one query shape vary by tens of per cent; from 32 MiB up ANY pooled block that is large enough is
        // taken — pinning costs ~40 us per MiB, as much as the staged copy it is meant to replace)
        if (it != pin_pool.end() && (it->first <= 2 * cap || cap >= ((size_t)32 << 20))) {
            void* p = it->second;
            pin_pooled -= it->first;
            pin_live[p] = it->first;
            pin_pool.erase(it);
            return p;
        }
    }
    (void)lane();                                            // (device current)
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        // make room: give the pooled blocks back and try once more
        std::vector<void*> old;
        {
            std::lock_guard<std::mutex> g(mu);
            for (auto& kv : pin_pool) old.push_back(kv.second);
            pin_pool.clear();
            pin_pooled = 0;
        }
        for (void* q : old) (void)hipHostFree(q);
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    std::lock_guard<std::mutex> g(mu);
    pin_live[p] = cap;
    return p;
}

void* fgpu_ctx::result_alloc(size_t bytes) {
    // (arrays the pool could not keep for the next call — beyond pinned_pool_mb, 4 GiB by default: an exported RMAT-26 adjacency
    // — stay pageable.  Up to round 5 the bound was 1 GiB "because pinning costs what staging saves"; measured in round 6 on the
    // 2 GB result of an emitting 3-hop batch: staged copy into pageable memory 317 ms, pinned ~40 us per MiB once, DMA after)
    if (opt.pinned_results && bytes >= PIN_MIN && bytes <= ((size_t)opt.pinned_pool_mb << 20)) {
        void* p = pinned_alloc(bytes);
        if (p) return p;
    }
    return host_alloc(bytes);
}

void fgpu_ctx::host_free(void* p) {
    if (!p) return;
    size_t cap = 0;
    bool keep = false;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = pin_live.find(p);
        if (it != pin_live.end()) {
            cap = it->second;
            pin_live.erase(it);
            if (pin_pooled + cap <= (uint64_t)opt.pinned_pool_mb << 20) {
                pin_pool.emplace(cap, p);
                pin_pooled += cap;
                keep = true;
            }
        }
    }
    if (cap) {
        if (!keep) (void)hipHostFree(p);
        return;
    }
    if (fre) fre(p); else free(p);
}

extern "C" {

const char* fgpu_last_error(void) { return get_error(); }

fgpu_info fgpu_init(fgpu_ctx** out, int device, void* (*mal)(size_t), void (*fre)(void*)) {
    FGPU_REQUIRE(out != nullptr, FGPU_NULL_POINTER, "fgpu_init: ctx out pointer is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        set_error("fgpu_init: no HIP device visible (%s); this engine has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return FGPU_DEVICE;
    }
    FGPU_REQUIRE(device >= 0 && device < ndev, FGPU_INVALID, "fgpu_init: device %d out of range [0,%d)",
                 device, ndev);
    FGPU_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    FGPU_HIP(hipGetDeviceProperties(&prop, device));
    FGPU_REQUIRE(prop.warpSize == 64, FGPU_DEVICE,
                 "fgpu_init: device wavefront is %d, kernels are written for 64 (gfx950)", prop.warpSize);
    fgpu_ctx* c = new (std::nothrow) fgpu_ctx();
    FGPU_REQUIRE(c != nullptr, FGPU_OOM, "fgpu_init: out of host memory");
    c->device = device;
    c->id = g_next_id.fetch_add(1);
    c->mal = mal;
    c->fre = fre;
    c->cus = prop.multiProcessorCount;
    c->opt.lds_limit = (int)prop.sharedMemPerBlock;
    if (c->lane() == nullptr) {   // the calling thread's lane: stream + pinned staging block + fence event
        set_error("fgpu_init: could not create a stream / pinned staging block on device %d", device);
        delete c;
        return FGPU_DEVICE;
    }
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        g_live_ctx[c->id] = c;
    }
    *out = c;
    return FGPU_OK;
}

fgpu_info fgpu_finalize(fgpu_ctx* ctx) {
    if (!ctx) return FGPU_OK;
    {
        std::lock_guard<std::mutex> g(g_reg_mu);   // exiting threads stop looking at this context
        g_live_ctx.erase(ctx->id);
    }
    (void)hipSetDevice(ctx->device);
    for (fgpu_lane* l : ctx->lanes) (void)hipStreamSynchronize(l->stream);
    ctx->trim();
    // live blocks still owned by un-freed matrices are released here too
    for (auto& kv : ctx->live) (void)hipFree(kv.first);
    ctx->live.clear();
    for (void* q : ctx->flag_all) (void)hipHostFree(q);
    ctx->flag_all.clear(); ctx->flag_free_list.clear();
    for (auto& kv : ctx->pin_pool) (void)hipHostFree(kv.second);
    for (auto& kv : ctx->pin_live) (void)hipHostFree(const_cast<void*>(kv.first));   // result arrays the caller never freed
    ctx->pin_pool.clear(); ctx->pin_live.clear();
    for (auto& e : ctx->prof) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
    for (hipEvent_t e : ctx->prof_free) (void)hipEventDestroy(e);
    for (fgpu_lane* l : ctx->lanes) lane_destroy(l);
    // threads that still cache a lane of this context key their cache on ctx->id, which is never reused
    if (tl_lanes.last_id == ctx->id) { tl_lanes.last_id = 0; tl_lanes.last_lane = nullptr; }
    for (size_t i = 0; i < tl_lanes.refs.size();)
        if (tl_lanes.refs[i].id == ctx->id) tl_lanes.refs.erase(tl_lanes.refs.begin() + i); else ++i;
    delete ctx;
    return FGPU_OK;
}

void fgpu_free(fgpu_ctx* ctx, void* p) {
    if (!p) return;
    if (ctx) ctx->host_free(p); else free(p);
}

fgpu_info fgpu_host_alloc(fgpu_ctx* ctx, uint64_t bytes, void** out) {
    FGPU_REQUIRE(ctx && out, FGPU_NULL_POINTER, "fgpu_host_alloc: NULL argument");
    *out = ctx->pinned_alloc((size_t)bytes);
    FGPU_REQUIRE(*out, FGPU_OOM, "fgpu_host_alloc: %llu B of pinned host memory are not available", (unsigned long long)bytes);
    return FGPU_OK;
}

fgpu_info fgpu_set_stream(fgpu_ctx* ctx, void* hip_stream) {
    FGPU_REQUIRE(ctx != nullptr, FGPU_NULL_POINTER, "fgpu_set_stream: NULL ctx");
    fgpu_lane* l = ctx->lane();
    // drain work queued on the previous stream so pooled blocks stay stream-ordered
    FGPU_HIP(hipStreamSynchronize(l->stream));
    std::lock_guard<std::mutex> g(l->fence_mu);
    l->stream = hip_stream ? (hipStream_t)hip_stream : l->own_stream;
    return FGPU_OK;
}

fgpu_info fgpu_prof_enable(fgpu_ctx* ctx, int enable) {
    FGPU_REQUIRE(ctx != nullptr, FGPU_NULL_POINTER, "fgpu_prof_enable: NULL ctx");
    std::lock_guard<std::mutex> g(ctx->prof_mu);
    if (enable) {
        for (auto& e : ctx->prof) { ctx->prof_free.push_back(e.e0); ctx->prof_free.push_back(e.e1); }
        ctx->prof.clear();
    }
    ctx->prof_on = enable != 0;
    return FGPU_OK;
}

fgpu_info fgpu_prof_read(fgpu_ctx* ctx, const char** names, double* ms, uint64_t* launches, uint64_t* alg_bytes,
                         int cap, int* n) {
    FGPU_REQUIRE(ctx && names && ms && launches && alg_bytes && n, FGPU_NULL_POINTER, "fgpu_prof_read: NULL argument");
    std::vector<fgpu_lane*> ls;
    { std::lock_guard<std::mutex> g(ctx->mu); ls = ctx->lanes; }
    for (fgpu_lane* l : ls) FGPU_HIP(hipStreamSynchronize(l->stream));
    std::lock_guard<std::mutex> g(ctx->prof_mu);
    int k = 0;
    for (auto& e : ctx->prof) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.e0, e.e1) != hipSuccess) { (void)hipGetLastError(); t = 0.f; }
        int j = 0;
        for (; j < k; ++j)
            if (names[j] == e.name || !strcmp(names[j], e.name)) break;
        if (j == k) {
            if (k == cap) continue;
            names[k] = e.name; ms[k] = 0; launches[k] = 0; alg_bytes[k] = 0;
            ++k;
        }
        ms[j] += t; launches[j] += 1; alg_bytes[j] += e.alg_bytes;
        ctx->prof_free.push_back(e.e0);
        ctx->prof_free.push_back(e.e1);
    }
    ctx->prof.clear();
    *n = k;
    return FGPU_OK;
}

fgpu_info fgpu_get_option(fgpu_ctx* ctx, const char* name, int64_t* value) {
    FGPU_REQUIRE(ctx && name && value, FGPU_NULL_POINTER, "fgpu_get_option: NULL argument");
    if (!strcmp(name, "dist_force_self")) *value = ctx->opt.dist_force_self;
    else if (!strcmp(name, "dist_self_calls")) *value = (int64_t)ctx->dist_self_calls.load(std::memory_order_relaxed);
    else if (!strcmp(name, "dist_collective")) *value = ctx->opt.dist_collective;
    else if (!strcmp(name, "expand_kernel_launches")) *value = (int64_t)ctx->expand_launches.load(std::memory_order_relaxed);
    else if (!strcmp(name, "expand_mode")) *value = ctx->opt.expand_mode;
    else if (!strcmp(name, "expand_scan_min")) *value = ctx->opt.expand_scan_min;
    else if (!strcmp(name, "expand_scan_rows")) *value = ctx->opt.expand_scan_rows;
    else if (!strcmp(name, "expand_scan_lanes")) *value = ctx->opt.expand_scan_lanes;
    else if (!strcmp(name, "expand_nt")) *value = ctx->opt.expand_nt;
    else if (!strcmp(name, "bfs_pb")) *value = ctx->opt.bfs_pb;
    else if (!strcmp(name, "bfs_pb_min_edges")) *value = ctx->opt.bfs_pb_min_edges;
    else if (!strcmp(name, "bfs_cp_last_mask")) *value = ctx->bfs_cp_last.load(std::memory_order_relaxed);
    else if (!strcmp(name, "bfs_pb_last_levels")) *value = ctx->bfs_pb_last.load(std::memory_order_relaxed);
    else if (!strcmp(name, "expand_scan_last_live")) *value = ctx->scan_last_live.load(std::memory_order_relaxed);
    else if (!strcmp(name, "expand_scan_last_passes")) *value = ctx->scan_last_passes.load(std::memory_order_relaxed);
    else { set_error("fgpu_get_option: unknown name '%s'", name); return FGPU_INVALID; }
    return FGPU_OK;
}

fgpu_info fgpu_set_option(fgpu_ctx* ctx, const char* name, int64_t value) {
    FGPU_REQUIRE(ctx && name, FGPU_NULL_POINTER, "fgpu_set_option: NULL argument");
    ctx->opt_epoch.fetch_add(1, std::memory_order_relaxed);
    if (!strcmp(name, "tiled_u")) {
        FGPU_REQUIRE(value == 1 || value == 2 || value == 4 || value == 8, FGPU_INVALID,
                     "tiled_u must be 1, 2, 4 or 8");
        ctx->opt.tiled_u = (int)value;
    } else if (!strcmp(name, "bfs_wgs_per_cu")) {
        FGPU_REQUIRE(value >= 1 && value <= 64, FGPU_INVALID, "bfs_wgs_per_cu out of range");
        ctx->opt.bfs_wgs_per_cu = (int)value;
    } else if (!strcmp(name, "expand_mode")) {
        FGPU_REQUIRE(value >= 0 && value <= 2, FGPU_INVALID, "expand_mode must be 0 (auto), 1 (sorted CSR) or 2 (bit-parallel)");
        ctx->opt.expand_mode = (int)value;
    } else if (!strcmp(name, "transpose_wb")) {
        ks_set_wb_override((int)value);

    } else if (!strcmp(name, "expand_row_groups")) {
        ctx->opt.expand_row_groups = value != 0;
    } else if (!strcmp(name, "expand_fuse_count")) {
        ctx->opt.expand_fuse_count = value != 0;
    } else if (!strcmp(name, "expand_compact")) {
        ctx->opt.expand_compact = value != 0;
    } else if (!strcmp(name, "pagerank_parts")) {
        FGPU_REQUIRE(value >= 0 && value <= 2, FGPU_INVALID, "pagerank_parts must be 0 (off), 1 (by size) or 2 (always)");
        ctx->opt.pagerank_parts = (int)value;
    } else if (!strcmp(name, "expand_first_hop")) {
        ctx->opt.expand_first_hop = value != 0;
    } else if (!strcmp(name, "expand_xcd")) {
        ctx->opt.expand_xcd = value != 0;
    } else if (!strcmp(name, "expand_xcd_relabel")) {
        ctx->opt.expand_xcd_relabel = value != 0;
    } else if (!strcmp(name, "expand_xcd_min_mb")) {
        FGPU_REQUIRE(value >= 0 && value <= (1 << 20), FGPU_INVALID, "expand_xcd_min_mb out of range");
        ctx->opt.expand_xcd_min_mb = (int)value;
    } else if (!strcmp(name, "expand_scan_min")) {
        FGPU_REQUIRE(value >= 0 && value <= (1ll << 31), FGPU_INVALID, "expand_scan_min out of range");
        ctx->opt.expand_scan_min = (int)value;
    } else if (!strcmp(name, "expand_scan_rows")) {
        FGPU_REQUIRE(value >= 64 && value <= 4096 && (value & (value - 1)) == 0, FGPU_INVALID,
                     "expand_scan_rows must be a power of two in 64 .. 4096");
        ctx->opt.expand_scan_rows = (int)value;
    } else if (!strcmp(name, "expand_scan_lanes")) {
        FGPU_REQUIRE(value >= 1 && value <= 16, FGPU_INVALID, "expand_scan_lanes must be 1 .. 16");
        ctx->opt.expand_scan_lanes = (int)value;
    } else if (!strcmp(name, "expand_emit_sort")) {
        FGPU_REQUIRE(value >= 0 && value <= 2, FGPU_INVALID, "expand_emit_sort is 0 (ballot transpose), 1 (by density) or 2 (pairs + sort)");
        ctx->opt.expand_emit_sort = (int)value;
    } else if (!strcmp(name, "expand_records")) {
        ctx->opt.expand_records = value != 0;
    } else if (!strcmp(name, "expand_nt")) {
        FGPU_REQUIRE(value >= 0 && value <= 7, FGPU_INVALID, "expand_nt is a mask of 1 | 2 | 4");
        ctx->opt.expand_nt = (int)value;
    } else if (!strcmp(name, "expand_bits_ratio")) {
        FGPU_REQUIRE(value >= 1 && value <= 1024, FGPU_INVALID, "expand_bits_ratio out of range");
        ctx->opt.expand_bits_ratio = (int)value;
    } else if (!strcmp(name, "blocked_variant")) {
        FGPU_REQUIRE(value >= 0 && value <= 3, FGPU_INVALID, "blocked_variant must be 0..3");
        ctx->opt.blocked_variant = (int)value;
    } else if (!strcmp(name, "tiled_layout")) {
        FGPU_REQUIRE(value >= 0 && value <= 2, FGPU_INVALID, "tiled_layout must be 0 (auto), 1 (tiled) or 2 (blocked)");
        ctx->opt.tiled_layout = (int)value;
    } else if (!strcmp(name, "bfs_tiny")) {
        FGPU_REQUIRE(value >= 0 && value <= 2, FGPU_INVALID, "bfs_tiny must be 0, 1 or 2");
        ctx->opt.bfs_tiny = (int)value;
    } else if (!strcmp(name, "bfs_alive_rule")) {
        ctx->opt.bfs_alive_rule = value != 0;
    } else if (!strcmp(name, "bfs_pb")) {
        FGPU_REQUIRE(value >= 0 && value <= 2, FGPU_INVALID, "bfs_pb must be 0, 1 or 2");
        ctx->opt.bfs_pb = (int)value;
    } else if (!strcmp(name, "bfs_pb_min_edges")) {
        FGPU_REQUIRE(value >= 1, FGPU_INVALID, "bfs_pb_min_edges must be positive");
        ctx->opt.bfs_pb_min_edges = (long long)value;
    } else if (!strcmp(name, "bfs_prof_split")) {
        ctx->opt.bfs_prof_split = value != 0;
    } else if (!strcmp(name, "bfs_hub_first")) {
        ctx->opt.bfs_hub_first = value != 0;
    } else if (!strcmp(name, "dist_timing")) {
        ctx->opt.dist_timing = value != 0;
    } else if (!strcmp(name, "pinned_results")) {
        ctx->opt.pinned_results = value != 0;
    } else if (!strcmp(name, "pinned_pool_mb")) {
        FGPU_REQUIRE(value >= 0 && value <= (1 << 20), FGPU_INVALID, "pinned_pool_mb out of range");
        ctx->opt.pinned_pool_mb = (int)value;
    } else if (!strcmp(name, "dist_test_delay_us")) {
        FGPU_REQUIRE(value >= 0 && value <= 100000, FGPU_INVALID, "dist_test_delay_us out of range");
        ctx->opt.dist_test_delay_us = (int)value;
    } else if (!strcmp(name, "dist_force_self")) {
        ctx->opt.dist_force_self = value != 0;
    } else if (!strcmp(name, "dist_collective")) {
        FGPU_REQUIRE(value == 0 || value == 1, FGPU_INVALID, "dist_collective must be 0 (send/recv) or 1 (broadcasts)");
        ctx->opt.dist_collective = (int)value;
    } else if (!strcmp(name, "transpose_mode")) {
        FGPU_REQUIRE(value >= 0 && value <= 3, FGPU_INVALID,
                     "transpose_mode must be 0 (counting sort, form picked), 1 (COO rebuild), 2 (LDS-staged levels) or 3 (two levels)");
        ctx->opt.transpose_mode = (int)value;
    } else if (!strcmp(name, "merge_items")) {
        ctx->opt.merge_items = value != 0;
    } else if (!strcmp(name, "merge_mode")) {
        FGPU_REQUIRE(value >= 0 && value <= 2, FGPU_INVALID,
                     "merge_mode must be 0 (entry-parallel), 1 (row-wave) or 2 (entry-parallel, base layer marked entry by entry)");
        ctx->opt.merge_mode = (int)value;
    } else if (!strcmp(name, "tiled_nt")) {
        ctx->opt.tiled_nt = value != 0;
    } else if (!strcmp(name, "tiled_threads")) {
        FGPU_REQUIRE(value == 256 || value == 512 || value == 1024, FGPU_INVALID,
                     "tiled_threads must be 256, 512 or 1024");
        ctx->opt.tiled_threads = (int)value;
    } else if (!strcmp(name, "tiled_wgs")) {
        FGPU_REQUIRE(value >= 0 && value <= 65536, FGPU_INVALID, "tiled_wgs out of range");
        ctx->opt.tiled_wgs = (int)value;
    } else {
        set_error("fgpu_set_option: unknown option '%s'", name);
        return FGPU_INVALID;
    }
    return FGPU_OK;
}

fgpu_info fgpu_sync(fgpu_ctx* ctx) {
    FGPU_REQUIRE(ctx != nullptr, FGPU_NULL_POINTER, "fgpu_sync: NULL ctx");
    FGPU_HIP(hipStreamSynchronize(ctx->stream()));   // the calling thread's lane
    return FGPU_OK;
}

fgpu_info fgpu_device_info(fgpu_ctx* ctx, char* name, int32_t* cus, int32_t* wave, int64_t* lds_bytes,
                           int64_t* hbm_bytes) {
    FGPU_REQUIRE(ctx != nullptr, FGPU_NULL_POINTER, "fgpu_device_info: NULL ctx");
    hipDeviceProp_t prop;
    FGPU_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name) {
        snprintf(name, 256, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (wave) *wave = prop.warpSize;
    if (lds_bytes) *lds_bytes = (int64_t)prop.sharedMemPerBlock;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return FGPU_OK;
}

fgpu_info fgpu_device_bytes(fgpu_ctx* ctx, uint64_t* in_use, uint64_t* pooled) {
    FGPU_REQUIRE(ctx != nullptr, FGPU_NULL_POINTER, "fgpu_device_bytes: NULL ctx");
    std::lock_guard<std::mutex> g(ctx->mu);
    if (in_use) *in_use = ctx->bytes_in_use;
    if (pooled) *pooled = ctx->bytes_pooled;
    return FGPU_OK;
}

}  // extern "C"
