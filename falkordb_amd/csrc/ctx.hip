// ctx.hip — lifecycle, error strings, device pool.
// Replaces matrix::init / shutdown (reference graph/src/graph/graphblas/matrix.rs:116-221).
#include <stdarg.h>
#include <stdio.h>

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

fgpu_info read_u32(fgpu_ctx* ctx, const u32* dev, u32* host) {
    FGPU_HIP(hipMemcpyAsync(ctx->pinned, dev, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(hipStreamSynchronize(ctx->stream));
    *host = *(u32*)ctx->pinned;
    return FGPU_OK;
}
fgpu_info read_u64(fgpu_ctx* ctx, const u64* dev, u64* host) {
    FGPU_HIP(hipMemcpyAsync(ctx->pinned, dev, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(hipStreamSynchronize(ctx->stream));
    *host = *(u64*)ctx->pinned;
    return FGPU_OK;
}

}  // namespace fgpu

using namespace fgpu;

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
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = pool.lower_bound(cap);
        if (it != pool.end() && it->first <= cap + (cap >> 2)) {
            *p = it->second;
            size_t c = it->first;
            pool.erase(it);
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

void fgpu_ctx::dev_free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu);
    auto it = live.find(p);
    if (it == live.end()) return;
    size_t c = it->second;
    live.erase(it);
    bytes_in_use -= c;
    // Blocks are recycled stream-ordered: every kernel of this ctx runs on ctx->stream,
    // so a block handed out again is only touched after its previous users finished.
    pool.emplace(c, p);
    bytes_pooled += c;
}

void fgpu_ctx::trim() {
    std::multimap<size_t, void*> old;
    {
        std::lock_guard<std::mutex> g(mu);
        old.swap(pool);
        bytes_pooled = 0;
    }
    if (!old.empty()) (void)hipStreamSynchronize(stream);
    for (auto& kv : old) (void)hipFree(kv.second);
}

void* fgpu_ctx::host_alloc(size_t bytes) {
    if (bytes == 0) bytes = 8;
    return mal ? mal(bytes) : malloc(bytes);
}
void fgpu_ctx::host_free(void* p) {
    if (!p) return;
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
    c->mal = mal;
    c->fre = fre;
    c->cus = prop.multiProcessorCount;
    c->opt.lds_limit = (int)prop.sharedMemPerBlock;
    hipError_t se = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
        delete c;
        return FGPU_DEVICE;
    }
    c->stream = c->own_stream;
    c->pinned_bytes = 1 << 16;
    se = hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault);
    if (se != hipSuccess) {
        set_error("hipHostMalloc failed: %s", hipGetErrorString(se));
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return FGPU_DEVICE;
    }
    *out = c;
    return FGPU_OK;
}

fgpu_info fgpu_finalize(fgpu_ctx* ctx) {
    if (!ctx) return FGPU_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->trim();
    // live blocks still owned by un-freed matrices are released here too
    for (auto& kv : ctx->live) (void)hipFree(kv.first);
    ctx->live.clear();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return FGPU_OK;
}

void fgpu_free(fgpu_ctx* ctx, void* p) {
    if (!p) return;
    if (ctx) ctx->host_free(p); else free(p);
}

fgpu_info fgpu_set_stream(fgpu_ctx* ctx, void* hip_stream) {
    FGPU_REQUIRE(ctx != nullptr, FGPU_NULL_POINTER, "fgpu_set_stream: NULL ctx");
    // drain work queued on the previous stream so pooled blocks stay stream-ordered
    FGPU_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return FGPU_OK;
}

fgpu_info fgpu_set_option(fgpu_ctx* ctx, const char* name, int64_t value) {
    FGPU_REQUIRE(ctx && name, FGPU_NULL_POINTER, "fgpu_set_option: NULL argument");
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
    } else if (!strcmp(name, "bfs_tiny")) {
        FGPU_REQUIRE(value >= 0 && value <= 2, FGPU_INVALID, "bfs_tiny must be 0, 1 or 2");
        ctx->opt.bfs_tiny = (int)value;
    } else if (!strcmp(name, "bfs_hub_first")) {
        ctx->opt.bfs_hub_first = value != 0;
    } else if (!strcmp(name, "merge_mode")) {
        FGPU_REQUIRE(value == 0 || value == 1, FGPU_INVALID, "merge_mode must be 0 (entry-parallel) or 1 (row-wave)");
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
    FGPU_HIP(hipStreamSynchronize(ctx->stream));
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
