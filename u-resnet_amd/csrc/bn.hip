// BatchNorm2d (+ReLU, +residual add) forward / backward and per-channel column reductions.
// All tensors are dense NHWC viewed as [rows][C]; every kernel is HBM-bound: one coalesced
// pass, 16-byte (bf16x8 / float4) accesses where C allows, fp32 arithmetic, and per-channel
// reductions done as thread-private sums -> LDS tree over the row slots of a workgroup ->
// one partial row per workgroup (summed later in a fixed order: reproducible, no atomics).
#include "ur_common.h"
#include <stdlib.h>

#ifndef UR_RED_GRID_CAP
#define UR_RED_GRID_CAP 2048
#endif
#ifndef UR_BN_RED_UNROLL
#define UR_BN_RED_UNROLL 1
#endif
#ifndef UR_BN_RED_VEC8
#define UR_BN_RED_VEC8 0
#endif

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------ column reduction skeleton
// Thread (slot, colgroup) walks rows slot, slot+S, ... of its VEC channels and accumulates
// NV values per channel; workgroup writes partial[blockIdx.x][C][NV].
template <int VEC, int NV, typename Op>
__device__ __forceinline__ void colreduce_body(const Op& op, long long rows, int C, float* __restrict__ partial) {
    __shared__ float red[NT * VEC * NV];
    const int cq = (C + VEC - 1) / VEC;                 // column groups in total
    const int cqb = cq < NT ? cq : NT;                  // column groups handled per pass
    const int slots = NT / cqb;
    const int tid = threadIdx.x;
    const int slot = tid / cqb, cg = tid - slot * cqb;
    for (int cbase = 0; cbase < cq; cbase += cqb) {
        const int col = (cbase + cg) * VEC;
        float acc[VEC * NV];
#pragma unroll
        for (int i = 0; i < VEC * NV; ++i) acc[i] = 0.f;
        if (slot < slots && col < C) {
            // independent rows: unrolled so that the loads of several rows are in flight together
#pragma unroll UR_BN_RED_UNROLL
            for (long long r = (long long)blockIdx.x * slots + slot; r < rows; r += (long long)gridDim.x * slots)
                op(r, col, acc);
        }
#pragma unroll
        for (int i = 0; i < VEC * NV; ++i) red[i * NT + tid] = acc[i];
        __syncthreads();
        if (tid < cqb && col < C) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if (col + v < C) {
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        float s = 0.f;
                        for (int sl = 0; sl < slots; ++sl) s += red[(v * NV + k) * NT + sl * cqb + tid];
                        partial[((size_t)blockIdx.x * C + col + v) * NV + k] = s;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <typename T, int VEC> struct RowVec {
    float v[VEC];
    __device__ __forceinline__ void load(const T* p) {
        if constexpr (VEC == 8) {
#pragma unroll
            for (int h = 0; h < 2; ++h) { float4 f = ur_load4(p + 4 * h); v[4 * h] = f.x; v[4 * h + 1] = f.y; v[4 * h + 2] = f.z; v[4 * h + 3] = f.w; }
        } else if constexpr (VEC == 4) { float4 f = ur_load4(p); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
        else v[0] = ur_to_float(*p);
    }
};

template <> struct RowVec<bf16_t, 8> {
    float v[8];
    __device__ __forceinline__ void load(const bf16_t* p) {
        const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
};

template <typename T, int VEC>
__global__ __launch_bounds__(NT) void bn_stats_kernel(const T* __restrict__ y, long long rows, int C,
                                                      float* __restrict__ partial) {
    auto op = [&](long long r, int col, float* acc) {
        RowVec<T, VEC> a; a.load(y + r * C + col);
#pragma unroll
        for (int v = 0; v < VEC; ++v) { acc[v * 2] += a.v[v]; acc[v * 2 + 1] += a.v[v] * a.v[v]; }
    };
    colreduce_body<VEC, 2>(op, rows, C, partial);
}

template <typename T, int VEC>
__global__ __launch_bounds__(NT) void colsum_kernel(const T* __restrict__ x, long long rows, int C, int ld,
                                                    float* __restrict__ partial) {
    auto op = [&](long long r, int col, float* acc) {
        RowVec<T, VEC> a; a.load(x + r * ld + col);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += a.v[v];
    };
    colreduce_body<VEC, 1>(op, rows, C, partial);
}

// g = dz masked by the ReLU: from the saved output z when given, else recomputed from y
// ((y-mean)*invstd*gamma + beta > 0 has the sign of the stored activation).  A thread keeps one channel
// group for its whole row walk, so the per-channel constants live in registers.
// PITCHED: dz is a channel window with its own row pitch (a second address chain; the dense case shares y's)
template <typename T, int VEC, bool PITCHED>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                           const T* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int act, long long rows,
                                                           int C, int dz_ld, float* __restrict__ partial) {
    const bool have_z = act && z != nullptr;
    const bool recompute = act && z == nullptr;
    int cached_col = -1;
    float m[VEC], is[VEC], ga[VEC], be[VEC];
    auto op = [&](long long r, int col, float* acc) {
        if (col != cached_col) {
            cached_col = col;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                m[v] = mean[col + v]; is[v] = invstd[col + v];
                ga[v] = recompute ? gamma[col + v] : 0.f; be[v] = recompute ? beta[col + v] : 1.f;
            }
        }
        RowVec<T, VEC> g, yy, zz;
        g.load(dz + r * (PITCHED ? dz_ld : C) + col); yy.load(y + r * C + col);
        if (have_z) zz.load(z + r * C + col);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float xh = (yy.v[v] - m[v]) * is[v];
            float pre = xh * ga[v] + be[v];                    // no activation: 0 * xh + 1 > 0
            if (have_z) pre = zz.v[v];
            const float gv = pre > 0.f ? g.v[v] : 0.f;
            acc[v * 2] += gv; acc[v * 2 + 1] += gv * xh;
        }
    };
    colreduce_body<VEC, 2>(op, rows, C, partial);
}

// Fold partial[nparts][C][NV] down to its first `chunks` rows in place: workgroup (b, cgroup) owns rows
// b, b+chunks, b+2*chunks, ... of 32 channels (8 row slots x 32 channels, coalesced 32*NV-float reads),
// sums them in double in a fixed order and overwrites row b -- the only row of its set any workgroup writes.
template <int NV>
__global__ __launch_bounds__(256) void fold_partials_kernel(float* __restrict__ partial, int nparts, int C, int chunks) {
    __shared__ double red[8][32][NV];
    const int b = blockIdx.x, c = blockIdx.y * 32 + (threadIdx.x & 31), slot = threadIdx.x >> 5;
    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
    if (c < C) {
        for (int p = b + slot * chunks; p < nparts; p += 8 * chunks) {
#pragma unroll
            for (int k = 0; k < NV; ++k) acc[k] += (double)partial[((size_t)p * C + c) * NV + k];
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) red[slot][threadIdx.x & 31][k] = acc[k];
    __syncthreads();
    if (slot == 0 && c < C) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double s = 0.0;
            for (int sl = 0; sl < 8; ++sl) s += red[sl][threadIdx.x][k];
            partial[((size_t)b * C + c) * NV + k] = (float)s;
        }
    }
}

constexpr int FOLD_CHUNKS = 32;

// returns the number of partial rows left after folding (launches nothing when already small)
template <int NV>
int fold_partials(float* partial, int nparts, int C, hipStream_t st) {
    if (nparts <= FOLD_CHUNKS * 2) return nparts;
    hipLaunchKernelGGL((fold_partials_kernel<NV>), dim3(FOLD_CHUNKS, (C + 31) / 32), dim3(256), 0, st, partial, nparts, C,
                       FOLD_CHUNKS);
    return FOLD_CHUNKS;
}

// out[c][k] = sum_p partial[p][c][k] in double; optional mean/invstd/running-stat epilogue
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int nparts, long long count, int C, float eps,
                                   float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ aff) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    int p = 0;
    for (; p + 8 <= nparts; p += 8) {          // 8 independent loads in flight, summed in row order
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float2*>(partial + ((size_t)(p + u) * C + c) * 2);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += v[u].x; s2 += v[u].y; }
    }
    for (; p < nparts; ++p) {
        s1 += partial[((size_t)p * C + c) * 2];
        s2 += partial[((size_t)p * C + c) * 2 + 1];
    }
    double m = s1 / (double)count;
    double var = s2 / (double)count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (aff) {                                     // the normalisation folded to one multiply-add per element: [2][C]
        const float k = gamma[c] * invstd[c];
        aff[c] = k;
        aff[C + c] = beta[c] - mean[c] * k;
    }
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    if (rvar) {
        double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

// The sums over MANY partial rows [nparts][C][2] and the epilogue of bn_finalize_kernel (FINALIZE) or sum_partials_kernel
// (!FINALIZE) in one launch: a 1024-thread workgroup owns 8 channels (64 contiguous bytes per row), thread (slot, channel)
// walks rows slot, slot + 128, ... with 8 loads in flight and sums in double; the 128 slot sums are added in a fixed order
// through LDS (16 at a time, then 8).  Replaces fold_partials_kernel + the one-thread-per-channel kernel where the row
// count is large: one launch and no intermediate rounding to float.
template <bool FINALIZE>
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ partial, int nparts, long long count, int C,
                                                           float eps, float momentum, float* __restrict__ o0, float* __restrict__ o1,
                                                           float* __restrict__ rmean, float* __restrict__ rvar,
                                                           long long* __restrict__ nbt, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ aff,
                                                           float* __restrict__ sums) {
    __shared__ double red[128][16];
    __shared__ double red2[8][16];
    const int t = threadIdx.x, ch = t & 7, slot = t >> 3;
    const int c = blockIdx.x * 8 + ch;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        const float2* src = reinterpret_cast<const float2*>(partial) + c;
        int p = slot;
        for (; p + 7 * 128 < nparts; p += 8 * 128) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(p + u * 128) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0 += v[u].x; a1 += v[u].y; }
        }
        for (; p < nparts; p += 128) { const float2 v = src[(size_t)p * C]; a0 += v.x; a1 += v.y; }
    }
    red[slot][ch * 2] = a0;
    red[slot][ch * 2 + 1] = a1;
    __syncthreads();
    if (t < 128) {
        const int o = t & 15, j = t >> 4;
        double q = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) q += red[j * 16 + k][o];
        red2[j][o] = q;
    }
    __syncthreads();
    if (FINALIZE && blockIdx.x == 0 && t == 0 && nbt) *nbt += 1;
    if (t >= 8 || c >= C) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1 += red2[j][ch * 2]; s2 += red2[j][ch * 2 + 1]; }
    if (!FINALIZE) {
        if (sums) { sums[(size_t)c * 2] = (float)s1; sums[(size_t)c * 2 + 1] = (float)s2; }
        if (o0) o0[c] = (float)s1;
        if (o1) o1[c] = (float)s2;
        return;
    }
    const double m = s1 / (double)count;                     // from here on: bn_finalize_kernel's epilogue
    double var = s2 / (double)count - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)eps));
    o0[c] = mf;
    o1[c] = isf;
    if (aff) {
        const float k = gamma[c] * isf;
        aff[c] = k;
        aff[C + c] = beta[c] - mf * k;
    }
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mf;
    if (rvar) {
        const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

#ifndef UR_REDUCE_ROWS_MIN
#define UR_REDUCE_ROWS_MIN (FOLD_CHUNKS * 2 + 1)     // the row counts that used to be folded first
#endif
bool reduce_rows_enabled() {
    const char* e = getenv("URESNET_REDUCE_ROWS");
    return !(e && e[0] == '0');
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, int C, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float k = gamma[c] * (1.f / sqrtf(rvar[c] + eps));
    scale[c] = k;
    shift[c] = beta[c] - rmean[c] * k;
}

__global__ void bn_eval_prepare_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int C,
                                       float eps, float* __restrict__ mean, float* __restrict__ invstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    invstd[c] = 1.f / sqrtf(rvar[c] + eps);
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, int nparts, int C, int NV,
                                    float* __restrict__ sums, float* __restrict__ o0, float* __restrict__ o1) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    for (int k = 0; k < NV; ++k) {
        double s = 0.0;
        int p = 0;
        for (; p + 8 <= nparts; p += 8) {      // 8 independent loads in flight, summed in row order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[((size_t)(p + u) * C + c) * NV + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; p < nparts; ++p) s += partial[((size_t)p * C + c) * NV + k];
        if (sums) sums[(size_t)c * NV + k] = (float)s;
        if (k == 0 && o0) o0[c] = (float)s;
        if (k == 1 && o1) o1[c] = (float)s;
    }
}

// ------------------------------------------------------------------ elementwise passes
template <typename T> struct EW { static constexpr int V = 4; };

template <typename T>
__global__ __launch_bounds__(NT) void bn_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const T* __restrict__ res, int act,
                                                        T* __restrict__ z, long long total, int C, int z_ld) {
    // C % 4 == 0 (checked on the host): a float4 / bf16x4 never straddles a row
    const long long stride = (long long)gridDim.x * NT * 4;
    const int cstep = (int)(stride % C);          // channel offset advances mod C: no 64-bit division in the loop
    const long long zstep = (stride / C) * z_ld + cstep;           // z may be a channel window: rows are z_ld apart
    const long long i0 = (blockIdx.x * (long long)NT + threadIdx.x) * 4;
    int cnext = (int)(i0 % C);
    long long zoff = (i0 / C) * z_ld + cnext;
    for (long long i = i0; i < total; i += stride) {
        const int c = cnext;
        const long long zi = zoff;
        cnext += cstep; zoff += zstep;
        if (cnext >= C) { cnext -= C; zoff += z_ld - C; }
        float4 v = ur_load4(y + i);
        float4 m = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
        float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = (v.x - m.x) * is.x * g.x + b.x; o.y = (v.y - m.y) * is.y * g.y + b.y;
        o.z = (v.z - m.z) * is.z * g.z + b.z; o.w = (v.w - m.w) * is.w * g.w + b.w;
        if (res) { float4 r = ur_load4(res + i); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        if (act) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        ur_store4(z + zi, o);
    }
}

// Per-channel constants of 8 consecutive channels, kept in registers.
struct Chan8 {
    float m[8], is[8], ga[8], be[8];
    __device__ __forceinline__ void load(const float* mean, const float* invstd, const float* gamma, const float* beta, int c) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 a = *reinterpret_cast<const float4*>(mean + c + 4 * h), b = *reinterpret_cast<const float4*>(invstd + c + 4 * h);
            const float4 d = *reinterpret_cast<const float4*>(gamma + c + 4 * h), e = *reinterpret_cast<const float4*>(beta + c + 4 * h);
            m[4 * h] = a.x; m[4 * h + 1] = a.y; m[4 * h + 2] = a.z; m[4 * h + 3] = a.w;
            is[4 * h] = b.x; is[4 * h + 1] = b.y; is[4 * h + 2] = b.z; is[4 * h + 3] = b.w;
            ga[4 * h] = d.x; ga[4 * h + 1] = d.y; ga[4 * h + 2] = d.z; ga[4 * h + 3] = d.w;
            be[4 * h] = e.x; be[4 * h + 1] = e.y; be[4 * h + 2] = e.z; be[4 * h + 3] = e.w;
        }
    }
};

// bf16, 8 channels per thread (16-byte accesses).  The grid stride is a multiple of C for the power-of-two
// channel counts of the network, so a thread keeps its 8 channels for the whole walk (FIXED): constants are
// loaded once and UNROLL independent 16-byte loads are in flight per tensor before the first use.
// 16-byte accesses of the streaming passes, non-temporal: every byte of these tensors is touched once per pass and the
// tensors are many times the size of the L2.  Measured on a plain copy of 384 MiB: 5.9 TB/s with ordinary accesses, 6.4 TB/s
// non-temporal; on the training step of the default workload 31.15 -> 30.53 ms (loads 1.2 %, loads and stores 2.0 %).
// -DUR_NT_LOAD=0 / -DUR_NT_STORE=0 build the ordinary forms for A/B runs.
#ifndef UR_NT_LOAD
#define UR_NT_LOAD 1
#endif
#ifndef UR_NT_STORE
#define UR_NT_STORE 1
#endif
__device__ __forceinline__ bf16x8_t ld_stream(const bf16_t* p) {
    const bf16x8_t* q = reinterpret_cast<const bf16x8_t*>(p);
    return UR_NT_LOAD ? __builtin_nontemporal_load(q) : *q;
}
__device__ __forceinline__ void st_stream(bf16_t* p, const bf16x8_t& v) {
    bf16x8_t* q = reinterpret_cast<bf16x8_t*>(p);
    if (UR_NT_STORE) __builtin_nontemporal_store(v, q); else *q = v;
}

#ifndef UR_EW_UNROLL
#define UR_EW_UNROLL 4
#endif
#ifndef UR_EW_BLOCKS_PER_CU
#define UR_EW_BLOCKS_PER_CU 4
#endif
constexpr int EW_UNROLL = UR_EW_UNROLL;

template <bool FIXED, bool RES>
__global__ __launch_bounds__(NT) void bn_act_fwd_bf16x8_kernel(const bf16_t* __restrict__ y, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const bf16_t* __restrict__ res,
                                                               int act, bf16_t* __restrict__ z, long long total, int C,
                                                               int z_ld) {
    const long long stride = (long long)gridDim.x * NT * 8;
    const int cstep = (int)(stride % C);
    const long long zstep = (stride / C) * z_ld + cstep;           // z may be a channel window: rows are z_ld apart
    long long i = (blockIdx.x * (long long)NT + threadIdx.x) * 8;
    int cnext = (int)(i % C);
    long long zoff = (i / C) * z_ld + cnext;
    Chan8 k;
    if (FIXED) k.load(mean, invstd, gamma, beta, cnext);
    auto one = [&](const bf16x8_t& y8, const bf16x8_t& r8, long long zi) {
        bf16x8_t o8;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            float o = ((float)y8[v] - k.m[v]) * k.is[v] * k.ga[v] + k.be[v];      // same operation order as the 4-wide kernel
            if (RES) o += (float)r8[v];
            if (act) o = fmaxf(o, 0.f);
            o8[v] = (bf16_t)o;
        }
        st_stream(z + zi, o8);
    };
    if (FIXED) {
        for (; i + (EW_UNROLL - 1) * stride < total; i += EW_UNROLL * stride, zoff += EW_UNROLL * zstep) {
            bf16x8_t y8[EW_UNROLL], r8[EW_UNROLL];
#pragma unroll
            for (int u = 0; u < EW_UNROLL; ++u) {
                y8[u] = ld_stream(y + i + u * stride);
                if (RES) r8[u] = ld_stream(res + i + u * stride); else r8[u] = y8[u];
            }
#pragma unroll
            for (int u = 0; u < EW_UNROLL; ++u) one(y8[u], r8[u], zoff + u * zstep);
        }
    }
    for (; i < total; i += stride) {
        const long long zi = zoff;
        zoff += zstep;
        if (!FIXED) {
            k.load(mean, invstd, gamma, beta, cnext);
            cnext += cstep;
            if (cnext >= C) { cnext -= C; zoff += z_ld - C; }
        }
        const bf16x8_t y8 = ld_stream(y + i);
        bf16x8_t r8 = y8;
        if (RES) r8 = ld_stream(res + i);
        one(y8, r8, zi);
    }
}

// bn_act_fwd_bf16x8_kernel (dense result, no residual) that also sums its result per channel (sum z, sum z^2 of the
// ROUNDED values, i.e. what bn_stats_kernel would read back) for a BatchNorm that follows: one pass over z less.
// Needs NT * 8 % C == 0: a thread then stays on one channel group (tid % (C / 8)) for its whole walk.  The partial row of
// a workgroup has bn_stats_kernel's layout, partial[blockIdx.x][C][2], and sums the row slots in a fixed order.
__global__ __launch_bounds__(NT) void bn_act_fwd_stats_bf16x8_kernel(const bf16_t* __restrict__ y, const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, int act, bf16_t* __restrict__ z,
                                                                     long long total, int C, float* __restrict__ partial) {
    __shared__ float red[NT * 16];
    const long long stride = (long long)gridDim.x * NT * 8;
    long long i = (blockIdx.x * (long long)NT + threadIdx.x) * 8;
    const int tid = threadIdx.x, cq = C >> 3, slots = NT / cq;
    Chan8 k;
    k.load(mean, invstd, gamma, beta, (tid % cq) * 8);
    float s0[8], s1[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) { s0[v] = 0.f; s1[v] = 0.f; }
    auto one = [&](const bf16x8_t& y8, long long zi) {
        bf16x8_t o8;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            float o = ((float)y8[v] - k.m[v]) * k.is[v] * k.ga[v] + k.be[v];      // bn_act_fwd_bf16x8_kernel's expression
            if (act) o = fmaxf(o, 0.f);
            o8[v] = (bf16_t)o;
            const float r = (float)o8[v];
            s0[v] += r; s1[v] += r * r;
        }
        st_stream(z + zi, o8);
    };
    for (; i + (EW_UNROLL - 1) * stride < total; i += EW_UNROLL * stride) {
        bf16x8_t y8[EW_UNROLL];
#pragma unroll
        for (int u = 0; u < EW_UNROLL; ++u) y8[u] = ld_stream(y + i + u * stride);
#pragma unroll
        for (int u = 0; u < EW_UNROLL; ++u) one(y8[u], i + u * stride);
    }
    for (; i < total; i += stride) one(ld_stream(y + i), i);
#pragma unroll
    for (int v = 0; v < 8; ++v) { red[(v * 2) * NT + tid] = s0[v]; red[(v * 2 + 1) * NT + tid] = s1[v]; }
    __syncthreads();
    for (int o = tid; o < cq * 16; o += NT) {
        const int q = o / cq, g = o - q * cq;
        float t = 0.f;
        for (int sl = 0; sl < slots; ++sl) t += red[q * NT + sl * cq + g];
        partial[((size_t)blockIdx.x * C + g * 8 + (q >> 1)) * 2 + (q & 1)] = t;
    }
}

// bn_act_fwd_bf16x8_kernel and MaxPool2d(2, 2) of its result in one pass: a thread owns 8 channels of one 2x2 window --
// four 16-byte loads of y, four stores of z (possibly into a channel window), one store of the maximum and of the
// 8 position bytes.  z is bit for bit what the plain kernel writes (same expression) and the maximum is taken over the
// rounded values (first maximum in (ky, kx) order wins, NaN propagates: maxpool_fwd_kernel's rule), so the pooled tensor
// and the positions equal those of the separate pooling pass, which would have to read z back.
// C / 8 divides the grid stride (a power of two up to NT): a thread keeps its channels.
__global__ __launch_bounds__(NT) void bn_act_fwd_pool_bf16x8_kernel(const bf16_t* __restrict__ y, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    int act, bf16_t* __restrict__ z, int z_ld,
                                                                    bf16_t* __restrict__ pooled, uint8_t* __restrict__ idx, int N,
                                                                    int H, int W, int C) {
    const int OH = H / 2, OW = W / 2, CV = C / 8;
    const unsigned total = (unsigned)N * OH * OW * CV;
    unsigned i = blockIdx.x * (unsigned)NT + threadIdx.x;
    const int cv = (int)(i % CV);
    Chan8 k;
    k.load(mean, invstd, gamma, beta, cv * 8);
    for (; i < total; i += gridDim.x * (unsigned)NT) {
        const unsigned pix = i / CV;
        const int ox = (int)(pix % OW);
        const unsigned t = pix / OW;
        const int oy = (int)(t % OH), n = (int)(t / OH);
        const size_t p00 = ((size_t)n * H + 2 * oy) * W + 2 * ox;                 // pixel (2 oy, 2 ox)
        const size_t poff[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
        bf16x8_t y8[4], o8[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) y8[q] = ld_stream(y + poff[q] * C + cv * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                float o = ((float)y8[q][v] - k.m[v]) * k.is[v] * k.ga[v] + k.be[v];
                if (act) o = fmaxf(o, 0.f);
                o8[q][v] = (bf16_t)o;
            }
            st_stream(z + poff[q] * z_ld + cv * 8, o8[q]);
        }
        bf16x8_t m8;
        unsigned long long sel = 0;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            float m = (float)o8[0][v];
            int sl = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const float c = (float)o8[q][v];
                if (c > m || c != c) { m = c; sl = q; }
            }
            m8[v] = (bf16_t)m;
            sel |= (unsigned long long)sl << (8 * v);
        }
        st_stream(pooled + (size_t)pix * C + cv * 8, m8);
        *reinterpret_cast<unsigned long long*>(idx + (size_t)pix * C + cv * 8) = sel;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void bn_act_fwd_scalar_kernel(const T* __restrict__ y, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const T* __restrict__ res, int act, T* __restrict__ z,
                                                               long long total, int C, int z_ld) {
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int c = (int)(i % C);
        const long long zi = (i / C) * z_ld + c;
        float o = (ur_to_float(y[i]) - mean[c]) * invstd[c] * gamma[c] + beta[c];
        if (res) o += ur_to_float(res[i]);
        if (act) o = fmaxf(o, 0.f);
        z[zi] = ur_from_float<T>(o);
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(NT) void bn_act_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                              const T* __restrict__ z, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ sums,
                                                              int act, int eval, float inv_count,
                                                              T* __restrict__ dy, T* __restrict__ dres, long long total, int C,
                                                              int dz_ld) {
    const long long stride = (long long)gridDim.x * NT * VEC;
    const int cstep = (int)(stride % C);          // channel offset advances mod C: no 64-bit division in the loop
    int cnext = (int)(((blockIdx.x * (long long)NT + threadIdx.x) * VEC) % C);
    for (long long i = (blockIdx.x * (long long)NT + threadIdx.x) * VEC; i < total; i += stride) {
        const int c = cnext;
        cnext += cstep; if (cnext >= C) cnext -= C;
        RowVec<T, VEC> g, yy, zz;
        g.load(dz + (dz_ld == C ? i : (i / C) * dz_ld + c)); yy.load(y + i);
        const bool have_z = act && z != nullptr;
        if (have_z) zz.load(z + i);
        float o[VEC], gm[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float is = invstd[c + v], ga = gamma[c + v];
            const float xh = (yy.v[v] - mean[c + v]) * is;
            float pre = 1.f;
            if (have_z) pre = zz.v[v];
            else if (act) pre = xh * ga + beta[c + v];
            const float gv = pre > 0.f ? g.v[v] : 0.f;
            gm[v] = gv;
            const float k = ga * is;
            if (eval) o[v] = k * gv;
            else o[v] = k * (gv - sums[(c + v) * 2] * inv_count - xh * sums[(c + v) * 2 + 1] * inv_count);
        }
        if constexpr (VEC == 4) {
            ur_store4(dy + i, make_float4(o[0], o[1], o[2], o[3]));
            if (dres) ur_store4(dres + i, make_float4(gm[0], gm[1], gm[2], gm[3]));
        } else {
            dy[i] = ur_from_float<T>(o[0]);
            if (dres) dres[i] = ur_from_float<T>(gm[0]);
        }
    }
}

// bf16, 8 channels per thread: 16-byte tensor accesses, constants in registers, EW_UNROLL loads per tensor in flight
template <bool FIXED, bool HAVE_Z, bool DRES>
__global__ __launch_bounds__(NT) void bn_act_bwd_apply_bf16x8_kernel(
    const bf16_t* __restrict__ dz, const bf16_t* __restrict__ y, const bf16_t* __restrict__ z, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ sums, int act, int eval, float inv_count, bf16_t* __restrict__ dy, bf16_t* __restrict__ dres,
    long long total, int C, int dz_ld) {
    const long long stride = (long long)gridDim.x * NT * 8;
    const int cstep = (int)(stride % C);          // channel offset advances mod C: no 64-bit division in the loop
    const long long gstep = (stride / C) * dz_ld + cstep;         // dz may be a channel window: rows are dz_ld apart
    int cnext = (int)(((blockIdx.x * (long long)NT + threadIdx.x) * 8) % C);
    long long goff = (((blockIdx.x * (long long)NT + threadIdx.x) * 8) / C) * dz_ld + cnext;
    Chan8 k;
    float s0[8], s1[8];
    auto load_consts = [&](int c) {
        k.load(mean, invstd, gamma, beta, c);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p;
            if (!eval) {
                p = *reinterpret_cast<const float4*>(sums + (c + 4 * h) * 2);
                q = *reinterpret_cast<const float4*>(sums + (c + 4 * h) * 2 + 4);
            }
            s0[4 * h] = p.x * inv_count; s1[4 * h] = p.y * inv_count; s0[4 * h + 1] = p.z * inv_count; s1[4 * h + 1] = p.w * inv_count;
            s0[4 * h + 2] = q.x * inv_count; s1[4 * h + 2] = q.y * inv_count; s0[4 * h + 3] = q.z * inv_count; s1[4 * h + 3] = q.w * inv_count;
        }
    };
    if (FIXED) load_consts(cnext);
    auto one = [&](const bf16x8_t& g8, const bf16x8_t& y8, const bf16x8_t& z8, long long i) {
        bf16x8_t o8, r8;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const float xh = ((float)y8[v] - k.m[v]) * k.is[v];
            float pre = 1.f;
            if (HAVE_Z) pre = (float)z8[v];
            else if (act) pre = xh * k.ga[v] + k.be[v];
            const float gv = pre > 0.f ? (float)g8[v] : 0.f;
            const float kk = k.ga[v] * k.is[v];
            const float o = eval ? kk * gv : kk * (gv - s0[v] - xh * s1[v]);
            o8[v] = (bf16_t)o; r8[v] = (bf16_t)gv;
        }
        st_stream(dy + i, o8);
        if (DRES) st_stream(dres + i, r8);
    };
    long long i = (blockIdx.x * (long long)NT + threadIdx.x) * 8;
    if (FIXED) {
        constexpr int U = HAVE_Z ? 2 : EW_UNROLL;
        for (; i + (U - 1) * stride < total; i += U * stride, goff += U * gstep) {
            bf16x8_t g8[U], y8[U], z8[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                g8[u] = ld_stream(dz + goff + u * gstep);
                y8[u] = ld_stream(y + i + u * stride);
                if (HAVE_Z) z8[u] = ld_stream(z + i + u * stride); else z8[u] = g8[u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) one(g8[u], y8[u], z8[u], i + u * stride);
        }
    }
    for (; i < total; i += stride) {
        const long long gi = goff;
        goff += gstep;
        if (!FIXED) {
            load_consts(cnext);
            cnext += cstep;
            if (cnext >= C) { cnext -= C; goff += dz_ld - C; }
        }
        const bf16x8_t g8 = ld_stream(dz + gi);
        const bf16x8_t y8 = ld_stream(y + i);
        bf16x8_t z8 = g8;
        if (HAVE_Z) z8 = ld_stream(z + i);
        one(g8, y8, z8, i);
    }
}

// Backward reduction for bf16 with 8 channels per thread: thread (row slot, channel group) keeps its constants in
// registers and has U rows (16-byte loads of dz, y and optionally z) in flight before the first use.
// Requires C % 8 == 0 and C / 8 <= NT.
#ifndef UR_BN_RED8_UNROLL
#define UR_BN_RED8_UNROLL 2
#endif
template <bool HAVE_Z, bool PITCHED>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_bf16x8_kernel(
    const bf16_t* __restrict__ dz, const bf16_t* __restrict__ y, const bf16_t* __restrict__ z, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta, int act, long long rows,
    int C, int dz_ld, float* __restrict__ partial) {
    constexpr int U = HAVE_Z ? 2 : UR_BN_RED8_UNROLL;
    __shared__ float red[NT * 16];
    const int cq = C >> 3, slots = NT / cq;
    const int tid = threadIdx.x, slot = tid / cq, cg = tid - slot * cq, col = cg * 8;
    Chan8 k;
    k.load(mean, invstd, gamma, beta, col);
    if (!act || HAVE_Z) {
#pragma unroll
        for (int v = 0; v < 8; ++v) { k.ga[v] = 0.f; k.be[v] = 1.f; }          // mask always open (or taken from z)
    }
    float s0[8], s1[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) { s0[v] = 0.f; s1[v] = 0.f; }
    auto one = [&](const bf16x8_t& g8, const bf16x8_t& y8, const bf16x8_t& z8) {
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const float xh = ((float)y8[v] - k.m[v]) * k.is[v];
            float pre = xh * k.ga[v] + k.be[v];
            if (HAVE_Z) pre = (float)z8[v];
            const float gv = pre > 0.f ? (float)g8[v] : 0.f;
            s0[v] += gv; s1[v] += gv * xh;
        }
    };
    const long long step = (long long)gridDim.x * slots;
    long long r = (long long)blockIdx.x * slots + slot;
    if (slot >= slots) r = rows;                    // NT is not a multiple of C / 8: the surplus threads add nothing
    const bf16_t* py = y + r * C + col;
    const bf16_t* pz = z + r * C + col;
    const bf16_t* pg = dz + r * (PITCHED ? dz_ld : C) + col;
    const long long ystep = step * C, gstep = step * (PITCHED ? dz_ld : C);
    for (; r + (U - 1) * step < rows; r += U * step, py += U * ystep, pz += U * ystep, pg += U * gstep) {
        bf16x8_t g8[U], y8[U], z8[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            g8[u] = ld_stream(pg + u * gstep);
            y8[u] = ld_stream(py + u * ystep);
            if (HAVE_Z) z8[u] = ld_stream(pz + u * ystep); else z8[u] = g8[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(g8[u], y8[u], z8[u]);
    }
    for (; r < rows; r += step, py += ystep, pz += ystep, pg += gstep) {
        const bf16x8_t g8 = ld_stream(pg);
        const bf16x8_t y8 = ld_stream(py);
        bf16x8_t z8 = g8;
        if (HAVE_Z) z8 = ld_stream(pz);
        one(g8, y8, z8);
    }
#pragma unroll
    for (int v = 0; v < 8; ++v) { red[(v * 2) * NT + tid] = s0[v]; red[(v * 2 + 1) * NT + tid] = s1[v]; }
    __syncthreads();
    // output o = (value index i, channel group): fixed-order sum over the row slots
    for (int o = tid; o < cq * 16; o += NT) {
        const int i = o / cq, g = o - i * cq;
        float t = 0.f;
        for (int sl = 0; sl < slots; ++sl) t += red[i * NT + sl * cq + g];
        partial[((size_t)blockIdx.x * C + g * 8 + (i >> 1)) * 2 + (i & 1)] = t;
    }
}

// The same reduction for a BatchNorm(+ReLU) whose output went through a 2x2 max-pool (and, optionally, a skip
// connection): the gradient dz of that output is the max-pool backward of the pooled gradient plus the skip
// connection's gradient.  This kernel forms dz row by row exactly as the max-pool backward kernel does (pooled value where
// the saved index selects this pixel, else zero; plus the addend), stores it, and feeds it to the sums -- same walk,
// same summation order and same partial rows as bn_bwd_reduce_bf16x8_kernel on the stored dz, one pass over dz less.
__global__ __launch_bounds__(NT) void bn_bwd_reduce_pool_bf16x8_kernel(
    const bf16_t* __restrict__ dpool, const uint8_t* __restrict__ idx, const bf16_t* __restrict__ addend, int add_ld,
    bf16_t* __restrict__ dz, const bf16_t* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, unsigned rows, int H, int W, int C,
    float* __restrict__ partial) {
    constexpr int U = UR_BN_RED8_UNROLL;
    __shared__ float red[NT * 16];
    const int cq = C >> 3, slots = NT / cq;
    const int tid = threadIdx.x, slot = tid / cq, cg = tid - slot * cq, col = cg * 8;
    const int OH = H >> 1, OW = W >> 1;
    Chan8 k;
    k.load(mean, invstd, gamma, beta, col);
    float s0[8], s1[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) { s0[v] = 0.f; s1[v] = 0.f; }
    struct Src { bf16x8_t g, a, y; uint2 sel; int me; bool pooled; };
    auto fetch = [&](unsigned r) {
        Src t;
        const unsigned ix = r % (unsigned)W, q = r / (unsigned)W;
        const unsigned iy = q % (unsigned)H, n = q / (unsigned)H;
        const unsigned oy = iy >> 1, ox = ix >> 1;
        t.me = (int)((iy & 1) * 2 + (ix & 1));
        t.pooled = oy < (unsigned)OH && ox < (unsigned)OW;
        const size_t op = (((size_t)n * OH + oy) * OW + ox) * C + col;
        t.g = t.pooled ? ld_stream(dpool + op) : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        t.sel = t.pooled ? *reinterpret_cast<const uint2*>(idx + op) : make_uint2(0u, 0u);
        if (addend) t.a = ld_stream(addend + (size_t)r * add_ld + col);
        t.y = ld_stream(y + (size_t)r * C + col);
        return t;
    };
    auto one = [&](const Src& t, unsigned r) {
        bf16x8_t d8;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int sel = (int)(((v < 4 ? t.sel.x : t.sel.y) >> (8 * (v & 3))) & 0xffu);
            float o = (t.pooled && sel == t.me) ? (float)t.g[v] : 0.f;
            if (addend) o += (float)t.a[v];
            d8[v] = (bf16_t)o;                                      // what the max-pool backward kernel stores
        }
        st_stream(dz + (size_t)r * C + col, d8);
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const float xh = ((float)t.y[v] - k.m[v]) * k.is[v];
            const float pre = xh * k.ga[v] + k.be[v];
            const float gv = pre > 0.f ? (float)d8[v] : 0.f;
            s0[v] += gv; s1[v] += gv * xh;
        }
    };
    const unsigned step = gridDim.x * (unsigned)slots;
    unsigned long long r = (unsigned long long)blockIdx.x * slots + slot;
    if (slot >= slots) r = rows;                    // NT is not a multiple of C / 8: the surplus threads add nothing
    for (; r + (unsigned long long)(U - 1) * step < rows; r += (unsigned long long)U * step) {
        Src t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = fetch((unsigned)(r + (unsigned long long)u * step));
#pragma unroll
        for (int u = 0; u < U; ++u) one(t[u], (unsigned)(r + (unsigned long long)u * step));
    }
    for (; r < rows; r += step) one(fetch((unsigned)r), (unsigned)r);
#pragma unroll
    for (int v = 0; v < 8; ++v) { red[(v * 2) * NT + tid] = s0[v]; red[(v * 2 + 1) * NT + tid] = s1[v]; }
    __syncthreads();
    for (int o = tid; o < cq * 16; o += NT) {
        const int i = o / cq, g = o - i * cq;
        float t = 0.f;
        for (int sl = 0; sl < slots; ++sl) t += red[i * NT + sl * cq + g];
        partial[((size_t)blockIdx.x * C + g * 8 + (i >> 1)) * 2 + (i & 1)] = t;
    }
}

// Elementwise grid: enough workgroups to fill the chip several times over, each thread walking EW_UNROLL+ pieces.
// The count is rounded down to a multiple of C / gcd(C, NT * 8) when possible, which makes the grid stride a
// multiple of C (a thread then keeps its channels for the whole walk) for channel counts such as 80 too.
int ew_grid(long long pieces, int C) {
    long long g = (pieces + (long long)NT * EW_UNROLL - 1) / ((long long)NT * EW_UNROLL);
    if (g < 1) g = 1;
    if (g > 256 * UR_EW_BLOCKS_PER_CU) g = 256 * UR_EW_BLOCKS_PER_CU;
    int a = C, b = NT * 8;
    while (b) { const int t = a % b; a = b; b = t; }
    const int q = C / a;
    if (g >= q) g -= g % q;
    return (int)g;
}

int reduce_grid(long long rows, int C, int vec) {
    int cq = (C + vec - 1) / vec;
    int cqb = cq < NT ? cq : NT;
    int slots = NT / cqb;
    long long g = (rows + (long long)slots * 8 - 1) / ((long long)slots * 8);     // >= 8 rows per thread
    if (g < 1) g = 1;
    if (g > UR_RED_GRID_CAP) g = UR_RED_GRID_CAP;
    return (int)g;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

size_t ur_bn_stats_rows(long long rows, int c) { return (size_t)reduce_grid(rows, c, c % 4 == 0 ? 4 : 1); }

int ur_bn_stats(const void* y, int dtype, long long rows, int c, float* partial, ur_stream_t s) {
    UR_REQUIRE(y && partial && rows > 0 && c > 0, "bn_stats: bad arguments");
    const bool vec = (c % 4 == 0) && aligned16(y);
    const int grid = reduce_grid(rows, c, c % 4 == 0 ? 4 : 1);
    if (UR_BN_RED_VEC8 && dtype == UR_BF16 && vec && c % 8 == 0) {
        hipLaunchKernelGGL((bn_stats_kernel<bf16_t, 8>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)y, rows, c, partial);
        return ur_check_launch("bn_stats");
    }
    UR_DISPATCH_DTYPE(dtype, {
        if (vec) hipLaunchKernelGGL((bn_stats_kernel<T, 4>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const T*)y, rows, c, partial);
        else hipLaunchKernelGGL((bn_stats_kernel<T, 1>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const T*)y, rows, c, partial);
    });
    return ur_check_launch("bn_stats");
}

size_t ur_colsum_workspace(long long rows, int c) {
    return (size_t)reduce_grid(rows, c, c % 4 == 0 ? 4 : 1) * c * sizeof(float);
}

int ur_colsum(const void* x, int dtype, long long rows, int c, int ld, float* out, void* workspace, size_t ws_bytes,
              ur_stream_t s) {
    UR_REQUIRE(x && out && workspace && rows > 0 && c > 0 && ld >= c, "colsum: bad arguments");
    UR_REQUIRE(ws_bytes >= ur_colsum_workspace(rows, c), "colsum: workspace too small");
    const int esz = dtype == UR_BF16 ? 2 : 4;
    const bool vec = (c % 4 == 0) && (ld % 4 == 0) && (((uintptr_t)x) % (4 * esz) == 0);
    const int grid = reduce_grid(rows, c, c % 4 == 0 ? 4 : 1);
    float* partial = (float*)workspace;
    UR_DISPATCH_DTYPE(dtype, {
        if (vec) hipLaunchKernelGGL((colsum_kernel<T, 4>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const T*)x, rows, c, ld, partial);
        else hipLaunchKernelGGL((colsum_kernel<T, 1>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const T*)x, rows, c, ld, partial);
    });
    if (int e = ur_check_launch("colsum")) return e;
    const int nparts = fold_partials<1>(partial, grid, c, (hipStream_t)s);
    if (int e = ur_check_launch("colsum_fold_partials")) return e;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(ur_cdiv(c, 64)), dim3(64), 0, (hipStream_t)s, partial, nparts, c, 1,
                       (float*)nullptr, out, (float*)nullptr);
    return ur_check_launch("colsum_finalize");
}

int ur_bn_finalize(float* partial, int nparts, long long count, int c, float eps, float momentum, float* mean,
                   float* invstd, float* running_mean, float* running_var, long long* num_batches_tracked,
                   ur_stream_t s) {
    UR_REQUIRE(partial && mean && invstd && nparts > 0 && count > 0 && c > 0, "bn_finalize: bad arguments");
    if (nparts >= UR_REDUCE_ROWS_MIN && reduce_rows_enabled()) {
        hipLaunchKernelGGL(reduce_rows_kernel<true>, dim3(ur_cdiv(c, 8)), dim3(1024), 0, (hipStream_t)s, partial, nparts, count, c,
                           eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked, (const float*)nullptr,
                           (const float*)nullptr, (float*)nullptr, (float*)nullptr);
        return ur_check_launch("bn_finalize");
    }
    nparts = fold_partials<2>(partial, nparts, c, (hipStream_t)s);
    if (int e = ur_check_launch("bn_fold_partials")) return e;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ur_cdiv(c, 64)), dim3(64), 0, (hipStream_t)s, partial, nparts, count, c,
                       eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr);
    return ur_check_launch("bn_finalize");
}

int ur_bn_finalize_affine(float* partial, int nparts, long long count, int c, float eps, float momentum, float* mean,
                          float* invstd, float* running_mean, float* running_var, long long* num_batches_tracked,
                          const float* gamma, const float* beta, float* aff, ur_stream_t s) {
    UR_REQUIRE(partial && mean && invstd && gamma && beta && aff && nparts > 0 && count > 0 && c > 0, "bn_finalize_affine: bad arguments");
    if (nparts >= UR_REDUCE_ROWS_MIN && reduce_rows_enabled()) {
        hipLaunchKernelGGL(reduce_rows_kernel<true>, dim3(ur_cdiv(c, 8)), dim3(1024), 0, (hipStream_t)s, partial, nparts, count, c,
                           eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked, gamma, beta, aff,
                           (float*)nullptr);
        return ur_check_launch("bn_finalize_affine");
    }
    nparts = fold_partials<2>(partial, nparts, c, (hipStream_t)s);
    if (int e = ur_check_launch("bn_fold_partials")) return e;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ur_cdiv(c, 64)), dim3(64), 0, (hipStream_t)s, partial, nparts, count, c,
                       eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked, gamma, beta, aff);
    return ur_check_launch("bn_finalize_affine");
}

int ur_bn_eval_prepare(const float* running_mean, const float* running_var, int c, float eps, float* mean,
                       float* invstd, ur_stream_t s) {
    UR_REQUIRE(running_mean && running_var && mean && invstd && c > 0, "bn_eval_prepare: bad arguments");
    hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(ur_cdiv(c, 64)), dim3(64), 0, (hipStream_t)s, running_mean,
                       running_var, c, eps, mean, invstd);
    return ur_check_launch("bn_eval_prepare");
}

int ur_bn_eval_affine(const float* running_mean, const float* running_var, const float* gamma, const float* beta, int c,
                      float eps, float* scale, float* shift, ur_stream_t s) {
    UR_REQUIRE(running_mean && running_var && gamma && beta && scale && shift && c > 0, "bn_eval_affine: bad arguments");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(ur_cdiv(c, 64)), dim3(64), 0, (hipStream_t)s, running_mean, running_var,
                       gamma, beta, c, eps, scale, shift);
    return ur_check_launch("bn_eval_affine");
}

int ur_bn_act_fwd(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  const void* residual, int act, void* z, int dtype, long long rows, int c, int z_ld, ur_stream_t s) {
    UR_REQUIRE(y && mean && invstd && gamma && beta && z && rows > 0 && c > 0, "bn_act_fwd: bad arguments");
    if (z_ld == 0) z_ld = c;
    UR_REQUIRE(z_ld >= c, "bn_act_fwd: output pitch %d smaller than the channel count %d", z_ld, c);
    const long long total = rows * c;
    const bool vec = (c % 4 == 0) && (z_ld % (dtype == UR_BF16 ? 8 : 4) == 0) && aligned16(y) && aligned16(z) && aligned16(residual) && aligned16(mean) &&
                     aligned16(invstd) && aligned16(gamma) && aligned16(beta);
    if (dtype == UR_BF16 && vec && c % 8 == 0) {
        const int grid = ew_grid(total / 8, c);
        const bool fixed = ((long long)grid * NT * 8) % c == 0;
#define UR_BN_FWD(FIXED, RES)                                                                                             \
    hipLaunchKernelGGL((bn_act_fwd_bf16x8_kernel<FIXED, RES>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)y,  \
                       mean, invstd, gamma, beta, (const bf16_t*)residual, act, (bf16_t*)z, total, c, z_ld)
        if (fixed) { if (residual) UR_BN_FWD(true, true); else UR_BN_FWD(true, false); }
        else { if (residual) UR_BN_FWD(false, true); else UR_BN_FWD(false, false); }
#undef UR_BN_FWD
        return ur_check_launch("bn_act_fwd");
    }
    UR_DISPATCH_DTYPE(dtype, {
        if (vec) hipLaunchKernelGGL((bn_act_fwd_kernel<T>), dim3(ur_grid_1d(total / 4, NT)), dim3(NT), 0, (hipStream_t)s,
                                    (const T*)y, mean, invstd, gamma, beta, (const T*)residual, act, (T*)z, total, c, z_ld);
        else hipLaunchKernelGGL((bn_act_fwd_scalar_kernel<T>), dim3(ur_grid_1d(total, NT)), dim3(NT), 0, (hipStream_t)s,
                                (const T*)y, mean, invstd, gamma, beta, (const T*)residual, act, (T*)z, total, c, z_ld);
    });
    return ur_check_launch("bn_act_fwd");
}

int ur_bn_act_fwd_stats_supported(int dtype, long long rows, int c) {
    return dtype == UR_BF16 && rows > 0 && c >= 8 && c % 8 == 0 && (NT * 8) % c == 0;
}

size_t ur_bn_act_fwd_stats_rows(long long rows, int c) { return (size_t)ew_grid(rows * c / 8, c); }

int ur_bn_act_fwd_stats(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                        void* z, int dtype, long long rows, int c, float* partial, ur_stream_t s) {
    UR_REQUIRE(y && mean && invstd && gamma && beta && z && partial, "bn_act_fwd_stats: null pointer");
    UR_REQUIRE(ur_bn_act_fwd_stats_supported(dtype, rows, c),
               "bn_act_fwd_stats: needs bf16 and a channel count that divides %d (multiple of 8)", NT * 8);
    UR_REQUIRE(aligned16(y) && aligned16(z) && aligned16(mean) && aligned16(invstd) && aligned16(gamma) && aligned16(beta),
               "bn_act_fwd_stats: pointers must be 16-byte aligned");
    const long long total = rows * c;
    const int grid = ew_grid(total / 8, c);
    hipLaunchKernelGGL(bn_act_fwd_stats_bf16x8_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)y, mean, invstd,
                       gamma, beta, act, (bf16_t*)z, total, c, partial);
    return ur_check_launch("bn_act_fwd_stats");
}

int ur_bn_act_fwd_pool_supported(int dtype, int n, int h, int w, int c) {
    const int cv = c / 8;
    return dtype == UR_BF16 && n > 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0 && c >= 8 && c % 8 == 0 && cv <= NT &&
           (cv & (cv - 1)) == 0 && (long long)n * h * w * c < (1LL << 31);
}

int ur_bn_act_fwd_pool(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                       void* z, int z_ld, void* pooled, uint8_t* idx, int dtype, int n, int h, int w, int c, ur_stream_t s) {
    UR_REQUIRE(y && mean && invstd && gamma && beta && z && pooled && idx, "bn_act_fwd_pool: null pointer");
    UR_REQUIRE(ur_bn_act_fwd_pool_supported(dtype, n, h, w, c),
               "bn_act_fwd_pool: needs bf16, even height and width, a power-of-two channel count in [8, %d]", NT * 8);
    if (z_ld == 0) z_ld = c;
    UR_REQUIRE(z_ld >= c && z_ld % 8 == 0, "bn_act_fwd_pool: output pitch %d must be a multiple of 8 >= %d", z_ld, c);
    UR_REQUIRE((long long)n * h * w * z_ld < (1LL << 40), "bn_act_fwd_pool: output too large");
    UR_REQUIRE(aligned16(y) && aligned16(z) && aligned16(pooled) && aligned16(mean) && aligned16(invstd) && aligned16(gamma) &&
               aligned16(beta) && ((uintptr_t)idx % 8) == 0, "bn_act_fwd_pool: pointers must be 16-byte aligned (idx: 8)");
    const long long items = (long long)n * (h / 2) * (w / 2) * (c / 8);
    long long grid = (items + NT - 1) / NT;
    if (grid > 256 * UR_EW_BLOCKS_PER_CU) grid = 256 * UR_EW_BLOCKS_PER_CU;
    hipLaunchKernelGGL(bn_act_fwd_pool_bf16x8_kernel, dim3((unsigned)grid), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)y, mean,
                       invstd, gamma, beta, act, (bf16_t*)z, z_ld, (bf16_t*)pooled, idx, n, h, w, c);
    return ur_check_launch("bn_act_fwd_pool");
}

int ur_bn_act_bwd_reduce(const void* dz, const void* y, const void* z, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, int act, int dtype, long long rows, int c, int dz_ld,
                         float* partial, ur_stream_t s) {
    UR_REQUIRE(dz && y && mean && invstd && partial && rows > 0 && c > 0, "bn_act_bwd_reduce: bad arguments");
    if (dz_ld == 0) dz_ld = c;
    UR_REQUIRE(dz_ld >= c, "bn_act_bwd_reduce: gradient pitch %d is smaller than the channel count %d", dz_ld, c);
    UR_REQUIRE(!act || z || (gamma && beta), "bn_act_bwd_reduce: the activation mask needs z or (gamma, beta)");
    const bool vec = (c % 4 == 0) && aligned16(dz) && aligned16(y) && aligned16(z) &&
                     (dz_ld * (dtype == UR_BF16 ? 2 : 4)) % 16 == 0;
    const int grid = reduce_grid(rows, c, c % 4 == 0 ? 4 : 1);
#define UR_BN_RED(TT, V)                                                                                             \
    do {                                                                                                             \
        if (dz_ld == c) hipLaunchKernelGGL((bn_bwd_reduce_kernel<TT, V, false>), dim3(grid), dim3(NT), 0, (hipStream_t)s, \
                                           (const TT*)dz, (const TT*)y, (const TT*)z, mean, invstd, gamma, beta, act, rows, c, dz_ld, partial); \
        else hipLaunchKernelGGL((bn_bwd_reduce_kernel<TT, V, true>), dim3(grid), dim3(NT), 0, (hipStream_t)s,            \
                                (const TT*)dz, (const TT*)y, (const TT*)z, mean, invstd, gamma, beta, act, rows, c, dz_ld, partial); \
    } while (0)
    if (dtype == UR_BF16 && vec && c % 8 == 0 && c / 8 <= NT && (dz_ld * 2) % 16 == 0 && aligned16(mean) &&
        aligned16(invstd) && (!act || z || (aligned16(gamma) && aligned16(beta)))) {
        const bool have_z = act && z;
        const float* ga = gamma ? gamma : mean;        // constants are loaded unconditionally, then overridden
        const float* be = beta ? beta : mean;
#define UR_BN_RED8(HZ, PI)                                                                                            \
    hipLaunchKernelGGL((bn_bwd_reduce_bf16x8_kernel<HZ, PI>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)dz, \
                       (const bf16_t*)y, (const bf16_t*)z, mean, invstd, ga, be, act, rows, c, dz_ld, partial)
        if (have_z) { if (dz_ld == c) UR_BN_RED8(true, false); else UR_BN_RED8(true, true); }
        else { if (dz_ld == c) UR_BN_RED8(false, false); else UR_BN_RED8(false, true); }
#undef UR_BN_RED8
        return ur_check_launch("bn_act_bwd_reduce");
    }
    if (UR_BN_RED_VEC8 && dtype == UR_BF16 && vec && c % 8 == 0) {
        UR_BN_RED(bf16_t, 8);
        return ur_check_launch("bn_act_bwd_reduce");
    }
    UR_DISPATCH_DTYPE(dtype, {
        if (vec) UR_BN_RED(T, 4);
        else UR_BN_RED(T, 1);
    });
#undef UR_BN_RED
    return ur_check_launch("bn_act_bwd_reduce");
}

int ur_maxpool2x2_bwd_bnsums_supported(int dtype, int n, int h, int w, int c) {
    return dtype == UR_BF16 && n > 0 && h > 1 && w > 1 && c > 0 && c % 8 == 0 && c / 8 <= NT &&
           (long long)n * h * w < (1LL << 31);
}

int ur_maxpool2x2_bwd_bnsums(const void* dpool, const uint8_t* idx, void* dz, const void* addend, int add_ld, const void* bn_y,
                             const float* mean, const float* invstd, const float* gamma, const float* beta, float* partial,
                             int dtype, int n, int h, int w, int c, ur_stream_t s) {
    UR_REQUIRE(dpool && idx && dz && bn_y && mean && invstd && gamma && beta && partial, "maxpool2x2_bwd_bnsums: null pointer");
    UR_REQUIRE(ur_maxpool2x2_bwd_bnsums_supported(dtype, n, h, w, c),
               "maxpool2x2_bwd_bnsums: bf16, 8 | channels <= %d and fewer than 2^31 pixels only", NT * 8);
    if (addend && add_ld == 0) add_ld = c;
    UR_REQUIRE(!addend || (add_ld >= c && add_ld % 8 == 0), "maxpool2x2_bwd_bnsums: addend pitch %d (channels %d)", add_ld, c);
    UR_REQUIRE(aligned16(dpool) && aligned16(dz) && aligned16(bn_y) && aligned16(addend) && ((uintptr_t)idx & 7) == 0 &&
               aligned16(mean) && aligned16(invstd) && aligned16(gamma) && aligned16(beta),
               "maxpool2x2_bwd_bnsums: tensors must be 16-byte aligned (index bytes 8)");
    const long long rows = (long long)n * h * w;
    const int grid = reduce_grid(rows, c, 4);                 // the grid of ur_bn_act_bwd_reduce: ur_bn_stats_rows(rows, c) partial rows
    hipLaunchKernelGGL(bn_bwd_reduce_pool_bf16x8_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)dpool, idx,
                       (const bf16_t*)addend, add_ld, (bf16_t*)dz, (const bf16_t*)bn_y, mean, invstd, gamma, beta, (unsigned)rows,
                       h, w, c, partial);
    return ur_check_launch("maxpool2x2_bwd_bnsums");
}

int ur_bn_bwd_finalize(float* partial, int nparts, int c, float* sums, float* dgamma, float* dbeta,
                       ur_stream_t s) {
    UR_REQUIRE(partial && sums && nparts > 0 && c > 0, "bn_bwd_finalize: bad arguments");
    if (nparts >= UR_REDUCE_ROWS_MIN && reduce_rows_enabled()) {
        hipLaunchKernelGGL(reduce_rows_kernel<false>, dim3(ur_cdiv(c, 8)), dim3(1024), 0, (hipStream_t)s, partial, nparts, 0LL, c,
                           0.f, 0.f, dbeta, dgamma, (float*)nullptr, (float*)nullptr, (long long*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (float*)nullptr, sums);
        return ur_check_launch("bn_bwd_finalize");
    }
    nparts = fold_partials<2>(partial, nparts, c, (hipStream_t)s);
    if (int e = ur_check_launch("bn_fold_partials")) return e;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(ur_cdiv(c, 64)), dim3(64), 0, (hipStream_t)s, partial, nparts, c, 2,
                       sums, dbeta, dgamma);
    return ur_check_launch("bn_bwd_finalize");
}

int ur_bn_act_bwd_apply(const void* dz, const void* y, const void* z, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, const float* sums, int act, int eval, void* dy, void* dres,
                        int dtype, long long rows, int c, int dz_ld, ur_stream_t s) {
    UR_REQUIRE(dz && y && mean && invstd && gamma && dy && rows > 0 && c > 0, "bn_act_bwd_apply: bad arguments");
    if (dz_ld == 0) dz_ld = c;
    UR_REQUIRE(dz_ld >= c, "bn_act_bwd_apply: gradient pitch %d is smaller than the channel count %d", dz_ld, c);
    UR_REQUIRE(!act || z || beta, "bn_act_bwd_apply: the activation mask needs z or beta");
    UR_REQUIRE(eval || sums, "bn_act_bwd_apply: training mode needs the reduced sums");
    const long long total = rows * c;
    const float inv_count = 1.f / (float)rows;
    const bool al = aligned16(dz) && aligned16(y) && aligned16(z) && aligned16(dy) && aligned16(dres) &&
                    (dz_ld * (dtype == UR_BF16 ? 2 : 4)) % 16 == 0;
    if (dtype == UR_BF16 && c % 8 == 0 && al && beta && aligned16(mean) && aligned16(invstd) && aligned16(gamma) &&
        aligned16(beta) && aligned16(sums)) {
        const int grid = ew_grid(total / 8, c);
        const bool fixed = ((long long)grid * NT * 8) % c == 0;
        const bool have_z = act && z;
#define UR_BN_BWD(FIXED, HZ, DR)                                                                                          \
    hipLaunchKernelGGL((bn_act_bwd_apply_bf16x8_kernel<FIXED, HZ, DR>), dim3(grid), dim3(NT), 0, (hipStream_t)s,           \
                       (const bf16_t*)dz, (const bf16_t*)y, (const bf16_t*)z, mean, invstd, gamma, beta, sums, act, eval,  \
                       inv_count, (bf16_t*)dy, (bf16_t*)dres, total, c, dz_ld)
        if (fixed) {
            if (have_z) { if (dres) UR_BN_BWD(true, true, true); else UR_BN_BWD(true, true, false); }
            else { if (dres) UR_BN_BWD(true, false, true); else UR_BN_BWD(true, false, false); }
        } else {
            if (have_z) { if (dres) UR_BN_BWD(false, true, true); else UR_BN_BWD(false, true, false); }
            else { if (dres) UR_BN_BWD(false, false, true); else UR_BN_BWD(false, false, false); }
        }
#undef UR_BN_BWD
        return ur_check_launch("bn_act_bwd_apply");
    }
    const bool vec = (c % 4 == 0) && al;
    UR_DISPATCH_DTYPE(dtype, {
        if (vec) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, 4>), dim3(ur_grid_1d(total / 4, NT)), dim3(NT), 0,
                                    (hipStream_t)s, (const T*)dz, (const T*)y, (const T*)z, mean, invstd, gamma, beta, sums,
                                    act, eval, inv_count, (T*)dy, (T*)dres, total, c, dz_ld);
        else hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, 1>), dim3(ur_grid_1d(total, NT)), dim3(NT), 0, (hipStream_t)s,
                                (const T*)dz, (const T*)y, (const T*)z, mean, invstd, gamma, beta, sums, act, eval,
                                inv_count, (T*)dy, (T*)dres, total, c, dz_ld);
    });
    return ur_check_launch("bn_act_bwd_apply");
}

}  // extern "C"
