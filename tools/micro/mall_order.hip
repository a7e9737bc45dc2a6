// Does the Infinity Cache keep the tail of what the previous kernel streamed?  Kernel A reads (or writes) a buffer front
// to back; kernel B then reads it front to back or back to front.  If B is faster backwards, consumers should start where
// their producer ended.   hipcc --offload-arch=gfx950 -O3 mall_order.hip -o mall_order && ./mall_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ p, long long n16, int reverse, unsigned* sink) {
    unsigned acc = 0;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n16; i += stride) {
        const long long j = reverse ? n16 - 1 - i : i;
        const uint4 v = p[j];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void wr(uint4* __restrict__ p, long long n16, unsigned seed) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(seed + (unsigned)i, 1, 2, 3);
}

int main() {
    const size_t MB = 1 << 20;
    unsigned* sink; CK(hipMalloc(&sink, 4));
    uint4 *buf, *trash;
    CK(hipMalloc(&buf, 1024 * MB)); CK(hipMalloc(&trash, 1024 * MB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    for (size_t mb : {50, 100, 200, 400, 800}) {
        const long long n16 = (long long)(mb * MB / 16);
        for (int producer = 0; producer < 2; ++producer)
            for (int reverse = 0; reverse < 2; ++reverse) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, trash, (long long)(1024 * MB / 16), rep);   // flush the caches
                    if (producer == 0) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, buf, n16, 0, sink);
                    else hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, buf, n16, rep);
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, buf, n16, reverse, sink);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("%4zu MB  producer %-5s consumer %-8s %8.1f us  %6.2f TB/s\n", mb, producer ? "write" : "read",
                       reverse ? "backward" : "forward", best * 1e3, mb * MB / best / 1e9);
            }
    }
    return 0;
}
