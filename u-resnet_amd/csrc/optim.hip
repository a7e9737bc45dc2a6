// Multi-tensor optimiser steps: one launch updates every parameter tensor of the model.
// blockIdx.y selects the tensor (pointer table in device memory), blockIdx.x grid-strides over
// its elements (Adam: float4 accesses); fp32 master weights, fp32 moments.
#include "ur_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ void adam_one(float& pi, float gi, float& mi, float& vi, float beta1, float beta2, float eps,
                                         float wd, float bc2_sqrt, float step_size, float grad_scale) {
    gi *= grad_scale;
    if (wd != 0.f) gi += wd * pi;
    mi = mi + (gi - mi) * (1.f - beta1);                        // lerp form used by torch.optim.Adam
    vi = vi * beta2 + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
}

__global__ __launch_bounds__(NT) void adam_multi_kernel(const void* const* __restrict__ ptrs,
                                                        const long long* __restrict__ sizes, float lr, float beta1,
                                                        float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                        float grad_scale) {
    const int t = blockIdx.y;
    float* __restrict__ p = (float*)ptrs[t * 4 + 0];
    const float* __restrict__ g = (const float*)ptrs[t * 4 + 1];
    float* __restrict__ m = (float*)ptrs[t * 4 + 2];
    float* __restrict__ v = (float*)ptrs[t * 4 + 3];
    const long long n = sizes[t];
    const float step_size = lr / bc1;
    // 16-byte accesses over the aligned body (workgroups beyond a small tensor's length leave at once), scalar tail
    const bool al16 = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    const long long n4 = al16 ? n / 4 : 0;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        float4 p4 = reinterpret_cast<float4*>(p)[i], m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
        const float4 g4 = reinterpret_cast<const float4*>(g)[i];
        adam_one(p4.x, g4.x, m4.x, v4.x, beta1, beta2, eps, wd, bc2_sqrt, step_size, grad_scale);
        adam_one(p4.y, g4.y, m4.y, v4.y, beta1, beta2, eps, wd, bc2_sqrt, step_size, grad_scale);
        adam_one(p4.z, g4.z, m4.z, v4.z, beta1, beta2, eps, wd, bc2_sqrt, step_size, grad_scale);
        adam_one(p4.w, g4.w, m4.w, v4.w, beta1, beta2, eps, wd, bc2_sqrt, step_size, grad_scale);
        reinterpret_cast<float4*>(p)[i] = p4; reinterpret_cast<float4*>(m)[i] = m4; reinterpret_cast<float4*>(v)[i] = v4;
    }
    for (long long i = n4 * 4 + blockIdx.x * (long long)NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_one(pi, g[i], mi, vi, beta1, beta2, eps, wd, bc2_sqrt, step_size, grad_scale);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// Device-resident step counts: every tensor owns an int32 counter in device memory (5th pointer of its table row).
// adam_advance_steps_kernel increments the counters, adam_multi_dev_kernel (next launch on the same stream) derives the
// two bias corrections from its tensor's counter.  Nothing about the step comes from the host, so the pair can be
// captured in a graph and replayed; tensors whose state began at different times keep their own counts.
__global__ __launch_bounds__(NT) void adam_advance_steps_kernel(const void* const* __restrict__ ptrs, int ntensors) {
    for (int t = blockIdx.x * NT + threadIdx.x; t < ntensors; t += gridDim.x * NT) {
        int* c = (int*)ptrs[t * 5 + 4];
        *c = *c + 1;
    }
}

__global__ __launch_bounds__(NT) void adam_multi_dev_kernel(const void* const* __restrict__ ptrs,
                                                            const long long* __restrict__ sizes, float lr, float beta1,
                                                            float beta2, float eps, float wd, float grad_scale) {
    const int t = blockIdx.y;
    const long long n = sizes[t];
    if ((long long)blockIdx.x * NT >= n) return;                             // surplus workgroup of a short tensor
    float* __restrict__ p = (float*)ptrs[t * 5 + 0];
    const float* __restrict__ g = (const float*)ptrs[t * 5 + 1];
    float* __restrict__ m = (float*)ptrs[t * 5 + 2];
    float* __restrict__ v = (float*)ptrs[t * 5 + 3];
    const int step = *(const int*)ptrs[t * 5 + 4];
    // same arithmetic as the host path: corrections in double, rounded to float once
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    const float step_size = lr / bc1;
    const bool al16 = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    const long long n4 = al16 ? n / 4 : 0;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        float4 p4 = reinterpret_cast<float4*>(p)[i], m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
        const float4 g4 = reinterpret_cast<const float4*>(g)[i];
        adam_one(p4.x, g4.x, m4.x, v4.x, beta1, beta2, eps, wd, bc2_sqrt, step_size, grad_scale);
        adam_one(p4.y, g4.y, m4.y, v4.y, beta1, beta2, eps, wd, bc2_sqrt, step_size, grad_scale);
        adam_one(p4.z, g4.z, m4.z, v4.z, beta1, beta2, eps, wd, bc2_sqrt, step_size, grad_scale);
        adam_one(p4.w, g4.w, m4.w, v4.w, beta1, beta2, eps, wd, bc2_sqrt, step_size, grad_scale);
        reinterpret_cast<float4*>(p)[i] = p4; reinterpret_cast<float4*>(m)[i] = m4; reinterpret_cast<float4*>(v)[i] = v4;
    }
    for (long long i = n4 * 4 + blockIdx.x * (long long)NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_one(pi, g[i], mi, vi, beta1, beta2, eps, wd, bc2_sqrt, step_size, grad_scale);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

__global__ __launch_bounds__(NT) void sgd_multi_kernel(const void* const* __restrict__ ptrs,
                                                       const long long* __restrict__ sizes, float lr, float wd,
                                                       float grad_scale) {
    const int t = blockIdx.y;
    float* __restrict__ p = (float*)ptrs[t * 2 + 0];
    const float* __restrict__ g = (const float*)ptrs[t * 2 + 1];
    const long long n = sizes[t];
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        float gi = g[i] * grad_scale;
        float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        p[i] = pi - lr * gi;
    }
}

// workgroups per tensor, sized for the largest one (16 elements per thread); the others' surplus workgroups exit
int blocks_for(long long max_size) {
    long long b = (max_size + NT * 16 - 1) / (NT * 16);
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (int)b;
}

}  // namespace

extern "C" {

int ur_adam_multi(const void* const* ptrs, const long long* sizes, int ntensors, long long max_size, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, ur_stream_t s) {
    UR_REQUIRE(ptrs && sizes && ntensors > 0 && ntensors <= 65535 && max_size > 0 && step >= 1, "adam_multi: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks_for(max_size), ntensors), dim3(NT), 0, (hipStream_t)s, ptrs, sizes, lr,
                       beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    return ur_check_launch("adam_multi");
}

int ur_adam_multi_dev(const void* const* ptrs, const long long* sizes, int ntensors, long long max_size, float lr,
                      float beta1, float beta2, float eps, float weight_decay, float grad_scale, ur_stream_t s) {
    UR_REQUIRE(ptrs && sizes && ntensors > 0 && ntensors <= 65535 && max_size > 0, "adam_multi_dev: bad arguments");
    hipLaunchKernelGGL(adam_advance_steps_kernel, dim3((ntensors + NT - 1) / NT), dim3(NT), 0, (hipStream_t)s, ptrs, ntensors);
    hipLaunchKernelGGL(adam_multi_dev_kernel, dim3(blocks_for(max_size), ntensors), dim3(NT), 0, (hipStream_t)s, ptrs, sizes,
                       lr, beta1, beta2, eps, weight_decay, grad_scale);
    return ur_check_launch("adam_multi_dev");
}

int ur_sgd_multi(const void* const* ptrs, const long long* sizes, int ntensors, long long max_size, float lr,
                 float weight_decay, float grad_scale, ur_stream_t s) {
    UR_REQUIRE(ptrs && sizes && ntensors > 0 && ntensors <= 65535 && max_size > 0, "sgd_multi: bad arguments");
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(blocks_for(max_size), ntensors), dim3(NT), 0, (hipStream_t)s, ptrs, sizes, lr,
                       weight_decay, grad_scale);
    return ur_check_launch("sgd_multi");
}

}  // extern "C"
