// Achievable HBM bandwidth of plain streaming kernels on this device: copy (read + write), read-only and write-only, with
// 16-byte accesses, U accesses in flight per lane, temporal or non-temporal, over a range of grid sizes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bw_copy tools/micro/bw_copy.hip && tools/micro/bw_copy [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, unsigned* __restrict__ out, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void write_kernel(u32x4* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    const u32x4 v = {1u, 2u, 3u, 4u};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

template <typename F>
static double timed(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e-3;
}

int main(int argc, char** argv) {
    const long long mib = argc > 1 ? atoll(argv[1]) : 384;
    const long long bytes = mib << 20, n = bytes / 16;
    u32x4 *a, *b;
    unsigned* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    const int grids[] = {512, 1024, 2048, 4096, 8192, 16384};
    printf("%lld MiB per buffer; TB/s\n", mib);
    for (int g : grids) {
        printf("grid %5d:", g);
#define RUN(K, MUL) printf(" %5.2f", MUL * bytes / timed([&] { hipLaunchKernelGGL(K, dim3(g), dim3(256), 0, 0, a, b, n); }) / 1e12)
        printf("  copy U1/2/4/8"); RUN((copy_kernel<1, false>), 2.0); RUN((copy_kernel<2, false>), 2.0); RUN((copy_kernel<4, false>), 2.0); RUN((copy_kernel<8, false>), 2.0);
        printf("  nt"); RUN((copy_kernel<1, true>), 2.0); RUN((copy_kernel<2, true>), 2.0); RUN((copy_kernel<4, true>), 2.0); RUN((copy_kernel<8, true>), 2.0);
#undef RUN
#define RUNR(K) printf(" %5.2f", 1.0 * bytes / timed([&] { hipLaunchKernelGGL(K, dim3(g), dim3(256), 0, 0, a, out, n); }) / 1e12)
        printf("  read U4/8"); RUNR((read_kernel<4, false>)); RUNR((read_kernel<8, false>)); printf(" nt"); RUNR((read_kernel<4, true>)); RUNR((read_kernel<8, true>));
#undef RUNR
        printf("  write %5.2f nt %5.2f\n", 1.0 * bytes / timed([&] { hipLaunchKernelGGL((write_kernel<false>), dim3(g), dim3(256), 0, 0, b, n); }) / 1e12,
               1.0 * bytes / timed([&] { hipLaunchKernelGGL((write_kernel<true>), dim3(g), dim3(256), 0, 0, b, n); }) / 1e12);
        fflush(stdout);
    }
    return 0;
}
