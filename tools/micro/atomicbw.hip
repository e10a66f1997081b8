// Random-access op rates on a 512 KiB bitmap (the BFS visited / next-frontier words at RMAT-22): what one
// discovery's memory ops cost chip-wide.  MODE 0 = plain 4 B load, 1 = plain 4 B store, 2 = non-returning agent-scope
// atomicOr, 3 = returning atomicOr (result consumed), 4 = returning atomicOr + dependent non-returning one (a push
// discovery: visited then next), 5 = plain load hint + (rarely) atomic, i.e. already-visited edges.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/atomicbw.hip -o tools/micro/atomicbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* __restrict__ a, unsigned* __restrict__ b, unsigned nbits,
                                         unsigned per_thread, unsigned* out) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (unsigned i = 0; i < per_thread; i += 4) {
        unsigned u[4], r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = mix(tid * per_thread + i + j) % nbits;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned w = u[j] >> 5, bit = 1u << (u[j] & 31);
            if (MODE == 0) r[j] = a[w];
            else if (MODE == 1) { a[w] = bit; r[j] = 0; }
            else if (MODE == 2) { atomicOr(&a[w], bit); r[j] = 0; }
            else if (MODE == 3) r[j] = atomicOr(&a[w], bit);
            else if (MODE == 4) { r[j] = atomicOr(&a[w], bit); if (!(r[j] & bit)) atomicOr(&b[w], bit); }
            else { r[j] = a[w]; if (!(r[j] & bit)) r[j] = atomicOr(&a[w], bit); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += r[j];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static void run(const char* name, unsigned* a, unsigned* b, unsigned nbits, unsigned* out, int grid, unsigned per) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        hipMemset(a, 0, nbits / 8); hipMemset(b, 0, nbits / 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, a, b, nbits, per, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ops = (double)grid * 256 * per;
    printf("%-58s %9.1f us  %7.2f G ops/s\n", name, best * 1e3, ops / (best * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const unsigned nbits = 1u << 22;
    unsigned *a, *b, *out;
    hipMalloc(&a, nbits / 8); hipMalloc(&b, nbits / 8); hipMalloc(&out, 4);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    for (unsigned total : {1u << 20, 1u << 22}) {   // ~ one heavy push level's discoveries / edge checks
        const int grid = cus * 8;
        unsigned per = (total / (grid * 256) + 3) & ~3u; if (!per) per = 4;
        printf("-- %u ops over %d x 256 threads (%u per thread), bitmap %u bits\n", grid * 256 * per, grid, per, nbits);
        run<0>("plain 4 B load", a, b, nbits, out, grid, per);
        run<1>("plain 4 B store", a, b, nbits, out, grid, per);
        run<2>("atomicOr, non-returning", a, b, nbits, out, grid, per);
        run<3>("atomicOr, returning", a, b, nbits, out, grid, per);
        run<4>("atomicOr returning + dependent atomicOr (discovery)", a, b, nbits, out, grid, per);
        run<5>("load hint, atomicOr only if bit clear", a, b, nbits, out, grid, per);
    }
    return 0;
}
