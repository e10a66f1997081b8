// Read-only HBM streaming ceiling on this part: what a kernel that ONLY streams 16 B/lane loads reaches, for
// the shapes the tiled vxm uses (one 1024-thread workgroup per CU) and for a plain 256-thread grid.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/readbw.hip -o tools/micro/readbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int U>
__global__ void read_kernel(const uint4* __restrict__ p, size_t n, unsigned* out) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    for (; i < n; i += stride) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;   // never true in practice: keeps the loads alive
}

template <int U>
static void run(const char* name, const uint4* d, size_t n, unsigned* out, int grid, int block) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(read_kernel<U>, dim3(grid), dim3(block), 0, 0, d, n, out);
    hipEventRecord(a);
    const int it = 20;
    for (int k = 0; k < it; ++k) hipLaunchKernelGGL(read_kernel<U>, dim3(grid), dim3(block), 0, 0, d, n, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    double gb = (double)n * 16 / 1e9;
    printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms / it * 1e3, gb / (ms / it * 1e-3));
}

int main(int argc, char** argv) {
    size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : 266169064ull;   // the RMAT-22 tile layout
    size_t n = bytes / 16;
    uint4* d; unsigned* out;
    hipMalloc(&d, n * 16); hipMalloc(&out, 4);
    hipMemset(d, 1, n * 16);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    int cus = pr.multiProcessorCount;
    printf("%s: %d CUs, streaming %zu bytes\n", pr.gcnArchName, cus, n * 16);
    run<4>("1 WG/CU x 1024 thr, 4 loads in flight", d, n, out, cus, 1024);
    run<8>("1 WG/CU x 1024 thr, 8 loads in flight", d, n, out, cus, 1024);
    run<4>("2 WG/CU x 1024 thr, 4 loads in flight", d, n, out, cus * 2, 1024);
    run<4>("8 WG/CU x 256 thr, 4 loads in flight", d, n, out, cus * 8, 256);
    run<8>("8 WG/CU x 256 thr, 8 loads in flight", d, n, out, cus * 8, 256);
    run<4>("32 WG/CU x 256 thr, 4 loads in flight", d, n, out, cus * 32, 256);
    run<1>("grid = n/256 x 256 thr, 1 load per thread", d, n, out, (int)((n + 255) / 256), 256);
    return 0;
}
