// Shared device/host helpers for the U-ResNet HIP kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/uresnet_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// ---- host-side error plumbing (never throws across the C ABI) -----------------------------
void ur_set_error(const char* fmt, ...);
int ur_check_launch(const char* what);
// fragment-order image of a 3x3 layer's weights for the tile kernel (wide = 0) or the 128-channel kernel (wide = 1)
int ur_pack3x3_image(const float* w, void* packed, int cin, int cout, int dgrad, int wide, hipStream_t s);

// ConvTranspose2d(2, 2) forward / backward-data as a DMA-staged row GEMM (convt_gemm.hip); `supported` says whether a launch
// can go there (x: the low-resolution tensor, y: the 2x tensor), else the callers use the gather kernel
int ur_convt_gemm_supported(const ur_conv_desc* d, int dgrad, const void* x, const void* y);
int ur_convt_gemm_fwd(const void* x, const void* wp, const float* bias, void* y, const ur_conv_desc* d, hipStream_t st);
int ur_convt_gemm_dgrad(const void* dy, const void* wp, void* dx, const ur_conv_desc* d, hipStream_t st);

#define UR_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ur_set_error(__VA_ARGS__);        \
            return UR_ERR_INVALID;            \
        }                                     \
    } while (0)

static inline int ur_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int ur_grid_1d(long long work_items, int block, int cap = 256 * 16) {
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---- element access templated on the storage type -----------------------------------------
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int dtype = UR_F32; };
template <> struct ElemTraits<bf16_t> { static constexpr int dtype = UR_BF16; };

__device__ __forceinline__ float ur_to_float(float v) { return v; }
__device__ __forceinline__ float ur_to_float(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T ur_from_float(float v);
template <> __device__ __forceinline__ float ur_from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t ur_from_float<bf16_t>(float v) { return (bf16_t)v; }

// 4 consecutive elements -> float4 (pointer must be 4-element aligned)
__device__ __forceinline__ float4 ur_load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ur_load4(const bf16_t* p) {
    bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void ur_store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void ur_store4(bf16_t* p, float4 v) {
    bf16x4_t o = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
    *reinterpret_cast<bf16x4_t*>(p) = o;
}

// wave64 butterfly sum
__device__ __forceinline__ float ur_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum of `nsplit` fp32 slabs of `per` values each (per % 4 == 0, slabs 16-byte aligned).  A 256-thread
// workgroup covers 1024 / SL consecutive values with SL split lanes per value: 16-byte loads, four loads in
// flight per lane, fixed summation order.  Returns true in the threads that end up holding a finished value
// (`idx`, `total`); those are consecutive threads with consecutive `idx`.
template <int SL>
__device__ __forceinline__ bool ur_slab_sum(const float* __restrict__ ws, long long per, int nsplit, long long& idx,
                                            float& total) {
    constexpr int OL = 256 / SL;
    __shared__ float4 red[SL][OL];
    const int ol = threadIdx.x % OL, sl = threadIdx.x / OL;
    const long long base = ((long long)blockIdx.x * OL + ol) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (base < per) {
        const float* p = ws + base;
        int k = sl;
        for (; k + 3 * SL < nsplit; k += 4 * SL) {
            const float4 a = ur_load4(p + (long long)k * per), b = ur_load4(p + (long long)(k + SL) * per);
            const float4 c = ur_load4(p + (long long)(k + 2 * SL) * per), d = ur_load4(p + (long long)(k + 3 * SL) * per);
            s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
            s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; k < nsplit; k += SL) {
            const float4 a = ur_load4(p + (long long)k * per);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    red[sl][ol] = s;
    __syncthreads();
    if (threadIdx.x >= OL * 4) return false;
    const float* r = reinterpret_cast<const float*>(&red[0][0]) + threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < SL; ++k) t += r[k * OL * 4];
    idx = (long long)blockIdx.x * OL * 4 + threadIdx.x;
    total = t;
    return idx < per;
}
static inline int ur_slab_lanes(int nsplit) { return nsplit >= 64 ? 32 : 8; }
static inline unsigned ur_slab_grid(long long per, int lanes) { return (unsigned)((per + 1024 / lanes - 1) / (1024 / lanes)); }

#define UR_DISPATCH_DTYPE(dtype, ...)                                  \
    do {                                                               \
        if ((dtype) == UR_F32) { typedef float T; __VA_ARGS__; }       \
        else if ((dtype) == UR_BF16) { typedef bf16_t T; __VA_ARGS__; } \
        else { ur_set_error("unsupported dtype %d", (int)(dtype)); return UR_ERR_INVALID; } \
    } while (0)
