// Which bf16 MFMA shape runs the 3x3 tile kernel's inner loop faster on random data?  Both variants keep the kernel's
// per-wave output tile (4 rows x 32 pixels x 64 channels = 128 accumulator registers), read every operand from LDS with
// ds_read_b128 in the kernel's ratio (72 reads per 32 reduced channels per wave), run two workgroups of four waves per CU
// and meet at a barrier per slice.  No global traffic inside the loop.  Results are summed into a sink so nothing is dropped.
//   hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape && ./mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int R = 4, LDS_BYTES = 76 * 1024;

template <int OFF> __device__ __forceinline__ bf16x8_t rd(unsigned a) {
    bf16x8_t d;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF) : "memory");
    return d;
}

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void loop(const uint4* __restrict__ init, float* __restrict__ sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = init[(blockIdx.x & 7) * (LDS_BYTES / 16) + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(size_t)lds + lane * 16 + wave * 4352;      // conflict-free: consecutive lanes, 16 B apart
    float total = 0.f;
    if constexpr (SHAPE == 0) {                        // 32x32x16: acc[R][2] of 16 registers
        f32x16_t acc[R][2];
        for (int a = 0; a < R; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const unsigned so = base + (it & 1) * 38016;                        // alternate the two 16-channel stages
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                bf16x8_t br[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) br[k] = rd<0>(so + 19584 + (g * 6 + k) * 1024);
#pragma unroll
                for (int i = 0; i < R + 2; ++i) {
                    const bf16x8_t ar = rd<0>(so + i * 1088 + g * 32);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int a = i - dy;
                        if (a >= 0 && a < R) {
#pragma unroll
                            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar, br[dy * 2 + b], acc[a][b], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_s_barrier();
        }
        for (int a = 0; a < R; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) total += acc[a][b][r];
    } else {                                           // 16x16x32: acc[R][2 pixel halves][4 channel blocks] of 4 registers
        f32x4_t acc[R][2][4];
        for (int a = 0; a < R; ++a) for (int h = 0; h < 2; ++h) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) acc[a][h][n][r] = 0.f;
        for (int it = 0; it < iters; it += 2) {        // one 32-channel slice = two of the other variant's iterations
            const unsigned so = base;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                bf16x8_t br[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) br[k] = rd<0>(so + 39168 + (g * 12 + k) * 1024);
#pragma unroll
                for (int i = 0; i < R + 2; ++i) {
                    bf16x8_t ar[2];
                    ar[0] = rd<0>(so + i * 2176 + g * 64);
                    ar[1] = rd<0>(so + i * 2176 + g * 64 + 1024);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int a = i - dy;
                        if (a >= 0 && a < R) {
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int n = 0; n < 4; ++n)
                                    acc[a][h][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[h], br[dy * 4 + n], acc[a][h][n], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_s_barrier();          // the weight ring advances by thirds: three barriers per slice
            }
        }
        for (int a = 0; a < R; ++a) for (int h = 0; h < 2; ++h) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) total += acc[a][h][n][r];
    }
    if (total == 12345.678f) sink[0] = total;
}

int main() {
    const size_t n16 = 8 * (LDS_BYTES / 16);
    uint4* init; float* sink;
    CK(hipMalloc(&init, n16 * 16)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int zero = 0; zero < 2; ++zero) {
        unsigned short* h = (unsigned short*)malloc(n16 * 16);
        srand(1);
        for (size_t i = 0; i < n16 * 8; ++i) {         // bf16 values in (-2, 2): random sign, exponent 126..127, random mantissa
            h[i] = zero ? 0 : (unsigned short)(((rand() & 1) << 15) | ((126 + (rand() & 1)) << 7) | (rand() & 127));
        }
        CK(hipMemcpy(init, h, n16 * 16, hipMemcpyHostToDevice));
        free(h);
        const int iters = 2000, grid = 512;
        // FLOP per iteration (16 channels) per workgroup: 4 waves x 72 MFMA x 32768
        const double flop = (double)grid * iters * 4 * 72 * 32768.0;
        for (int rep = 0; rep < 3; ++rep)
            for (int shape = 0; shape < 2; ++shape) {
                for (int w = 0; w < 2; ++w) {
                    if (shape == 0) hipLaunchKernelGGL(loop<0>, dim3(grid), dim3(256), 0, 0, init, sink, iters);
                    else hipLaunchKernelGGL(loop<1>, dim3(grid), dim3(256), 0, 0, init, sink, iters);
                }
                CK(hipEventRecord(e0));
                for (int w = 0; w < 5; ++w) {
                    if (shape == 0) hipLaunchKernelGGL(loop<0>, dim3(grid), dim3(256), 0, 0, init, sink, iters);
                    else hipLaunchKernelGGL(loop<1>, dim3(grid), dim3(256), 0, 0, init, sink, iters);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("%s operands  %-9s  %8.3f ms per launch  %7.1f TFLOP/s\n", zero ? "zero  " : "random", shape ? "16x16x32" : "32x32x16",
                       ms / 5, flop / (ms / 5) / 1e9);
            }
    }
    return 0;
}
