// Bandwidth-bound spatial / layout kernels on NHWC tensors: 2x2 max-pool, bilinear upsample,
// average pool, channel-window copy (pad + concat), channel shuffle, elementwise add and
// activations, and the NCHW <-> NHWC boundary transposes.  One pass each, channel-fastest
// thread mapping so a wave touches contiguous memory, 4-channel (8/16-byte) accesses when the
// channel counts and pitches allow.
#include "ur_common.h"
#include <type_traits>

namespace {

constexpr int NT = 256;

template <typename T, int V> struct Pack {
    float v[V];
    __device__ __forceinline__ void load(const T* p) {
        if constexpr (V == 8) {                       // 16 bytes of bf16 (two 16-byte halves for fp32)
#pragma unroll
            for (int h = 0; h < 2; ++h) { float4 f = ur_load4(p + 4 * h); v[4 * h] = f.x; v[4 * h + 1] = f.y; v[4 * h + 2] = f.z; v[4 * h + 3] = f.w; }
        } else if constexpr (V == 4) { float4 f = ur_load4(p); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
        else v[0] = ur_to_float(*p);
    }
    __device__ __forceinline__ void store(T* p) const {
        if constexpr (V == 8) { ur_store4(p, make_float4(v[0], v[1], v[2], v[3])); ur_store4(p + 4, make_float4(v[4], v[5], v[6], v[7])); }
        else if constexpr (V == 4) ur_store4(p, make_float4(v[0], v[1], v[2], v[3]));
        else *p = ur_from_float<T>(v[0]);
    }
    __device__ __forceinline__ void fill(float x) {
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = x;
    }
};

template <> struct Pack<bf16_t, 8> {
    float v[8];
    __device__ __forceinline__ void load(const bf16_t* p) {
        const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
    __device__ __forceinline__ void store(bf16_t* p) const {
        bf16x8_t t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
        *reinterpret_cast<bf16x8_t*>(p) = t;
    }
    __device__ __forceinline__ void fill(float x) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = x;
    }
};

// V window-position bytes of the max-pool index, as one 4- or 8-byte access when V allows
template <int V> __device__ __forceinline__ void store_sel(uint8_t* p, const uint8_t* sel) {
    if constexpr (V == 8) {
        unsigned long long w = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) w |= (unsigned long long)sel[k] << (8 * k);
        *reinterpret_cast<unsigned long long*>(p) = w;
    } else if constexpr (V == 4) {
        *reinterpret_cast<unsigned*>(p) = sel[0] | (sel[1] << 8) | (sel[2] << 16) | (sel[3] << 24);
    } else p[0] = sel[0];
}
template <int V> __device__ __forceinline__ void load_sel(const uint8_t* p, uint8_t* sel) {
    if constexpr (V == 8) {
        const unsigned long long w = *reinterpret_cast<const unsigned long long*>(p);
#pragma unroll
        for (int k = 0; k < 8; ++k) sel[k] = (uint8_t)(w >> (8 * k));
    } else if constexpr (V == 4) {
        const unsigned w = *reinterpret_cast<const unsigned*>(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) sel[k] = (uint8_t)(w >> (8 * k));
    } else sel[0] = p[0];
}

// 64-bit division costs ~100 instructions on the GPU: index arithmetic is 32-bit (hosts check totals < 2^31)
#define GRID_STRIDE(i, total) \
    for (unsigned i = blockIdx.x * (unsigned)NT + threadIdx.x; i < (unsigned)(total); i += gridDim.x * (unsigned)NT)

// ------------------------------------------------------------------ max pool 2x2 stride 2
template <typename T, int V>
__global__ __launch_bounds__(NT) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         uint8_t* __restrict__ idx, int N, int H, int W, int C, int LD) {
    // LD: channel pitch of x (x may be a channel window of a wider tensor); y and idx are dense
    const int OH = H / 2, OW = W / 2, CV = C / V;
    const unsigned total = (unsigned)N * OH * OW * CV;
    GRID_STRIDE(i, total) {
        int cv = (int)(i % CV);
        unsigned pix = i / CV;
        int ox = (int)(pix % OW);
        unsigned t = pix / OW;
        int oy = (int)(t % OH), n = (int)(t / OH);
        const T* base = x + (((long long)n * H + 2 * oy) * W + 2 * ox) * LD + cv * V;
        Pack<T, V> a, b, c, d, o;
        a.load(base); b.load(base + LD); c.load(base + (long long)W * LD); d.load(base + (long long)W * LD + LD);
        uint8_t sel[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            // first maximum in (ky,kx) scan order wins; NaN propagates like PyTorch (val > max || isnan(val))
            float m = a.v[k]; int s = 0;
            if (b.v[k] > m || b.v[k] != b.v[k]) { m = b.v[k]; s = 1; }
            if (c.v[k] > m || c.v[k] != c.v[k]) { m = c.v[k]; s = 2; }
            if (d.v[k] > m || d.v[k] != d.v[k]) { m = d.v[k]; s = 3; }
            o.v[k] = m; sel[k] = (uint8_t)s;
        }
        o.store(y + (size_t)pix * C + cv * V);
        store_sel<V>(idx + (size_t)pix * C + cv * V, sel);
    }
}

template <typename T, int V>
__global__ __launch_bounds__(NT) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                         T* __restrict__ dx, const T* __restrict__ addend, int add_ld, int N,
                                                         int H, int W, int C) {
    const int OH = H / 2, OW = W / 2, CV = C / V;
    const unsigned total = (unsigned)N * H * W * CV;
    GRID_STRIDE(i, total) {
        int cv = (int)(i % CV);
        unsigned pix = i / CV;
        int ix = (int)(pix % W);
        unsigned t = pix / W;
        int iy = (int)(t % H), n = (int)(t / H);
        int oy = iy >> 1, ox = ix >> 1;
        Pack<T, V> o; o.fill(0.f);
        if (oy < OH && ox < OW) {
            long long op = (((long long)n * OH + oy) * OW + ox) * C + cv * V;
            Pack<T, V> g; g.load(dy + op);
            const int me = (iy & 1) * 2 + (ix & 1);
            uint8_t sel[V];
            load_sel<V>(idx + op, sel);
#pragma unroll
            for (int k = 0; k < V; ++k) o.v[k] = sel[k] == me ? g.v[k] : 0.f;
        }
        if (addend) {                                   // a second gradient of the pooled tensor (its skip connection)
            Pack<T, V> a; a.load(addend + (size_t)pix * add_ld + cv * V);
#pragma unroll
            for (int k = 0; k < V; ++k) o.v[k] += a.v[k];
        }
        o.store(dx + (size_t)pix * C + cv * V);
    }
}

// ------------------------------------------------------------------ bilinear upsample, align_corners = False
__device__ __forceinline__ void bilin_src(int o, float rscale, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float src = rscale * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

template <typename T, int V>
__global__ __launch_bounds__(NT) void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                          int C, int scale, float rscale) {
    const int OH = H * scale, OW = W * scale, CV = C / V;
    const unsigned total = (unsigned)N * OH * OW * CV;
    GRID_STRIDE(i, total) {
        int cv = (int)(i % CV);
        unsigned pix = i / CV;
        int ox = (int)(pix % OW);
        unsigned t = pix / OW;
        int oy = (int)(t % OH), n = (int)(t / OH);
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        bilin_src(oy, rscale, H, y0, y1, ly0, ly1);
        bilin_src(ox, rscale, W, x0, x1, lx0, lx1);
        const T* b = x + (long long)n * H * W * C + cv * V;
        Pack<T, V> v00, v01, v10, v11, o;
        v00.load(b + ((long long)y0 * W + x0) * C); v01.load(b + ((long long)y0 * W + x1) * C);
        v10.load(b + ((long long)y1 * W + x0) * C); v11.load(b + ((long long)y1 * W + x1) * C);
#pragma unroll
        for (int k = 0; k < V; ++k)
            o.v[k] = ly0 * (lx0 * v00.v[k] + lx1 * v01.v[k]) + ly1 * (lx0 * v10.v[k] + lx1 * v11.v[k]);
        o.store(y + (size_t)pix * C + cv * V);
    }
}

// One output row per blockIdx.y: the row's vertical taps are workgroup-uniform and the only per-thread division left
// is by the channel-vector count (the flat kernel above spends more time in three 32-bit divisions per 16-byte store
// than on memory).
template <typename T, int V>
__global__ __launch_bounds__(NT) void upsample_fwd_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C,
                                                               int scale, float rscale) {
    const int OH = H * scale, OW = W * scale, CV = C / V;
    const int e = blockIdx.x * NT + threadIdx.x;
    if (e >= OW * CV) return;
    const int row = blockIdx.y, n = row / OH, oy = row - n * OH;
    const int ox = e / CV, cv = e - ox * CV;
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    bilin_src(oy, rscale, H, y0, y1, ly0, ly1);
    bilin_src(ox, rscale, W, x0, x1, lx0, lx1);
    const T* b = x + (long long)n * H * W * C + cv * V;
    Pack<T, V> v00, v01, v10, v11, o;
    v00.load(b + ((long long)y0 * W + x0) * C); v01.load(b + ((long long)y0 * W + x1) * C);
    v10.load(b + ((long long)y1 * W + x0) * C); v11.load(b + ((long long)y1 * W + x1) * C);
#pragma unroll
    for (int k = 0; k < V; ++k)
        o.v[k] = ly0 * (lx0 * v00.v[k] + lx1 * v01.v[k]) + ly1 * (lx0 * v10.v[k] + lx1 * v11.v[k]);
    o.store(y + ((size_t)row * OW + ox) * C + cv * V);
}

// gather form of the adjoint: each input pixel sums the output pixels that read it (fixed order, no atomics)
template <typename T, int V>
__global__ __launch_bounds__(NT) void upsample_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W,
                                                          int C, int scale, float rscale) {
    const int OH = H * scale, OW = W * scale, CV = C / V;
    const unsigned total = (unsigned)N * H * W * CV;
    GRID_STRIDE(i, total) {
        int cv = (int)(i % CV);
        unsigned pix = i / CV;
        int ix = (int)(pix % W);
        unsigned t = pix / W;
        int iy = (int)(t % H), n = (int)(t / H);
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        if (scale == 2) {
            // the 4 x 4 output pixels that can read this input pixel: all 16 loads issued before the first use (the
            // generic loops below wait for each load in turn); same summation order, zero weights where they skip
            float wy[4], wx[4]; int oys[4], oxs[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int oy = 2 * iy - 1 + a, ox = 2 * ix - 1 + a;
                const bool vy = (unsigned)oy < (unsigned)OH, vx = (unsigned)ox < (unsigned)OW;
                oys[a] = vy ? oy : 0; oxs[a] = vx ? ox : 0;
                int i0, i1; float l0, l1;
                bilin_src(oys[a], rscale, H, i0, i1, l0, l1);
                wy[a] = vy ? (i0 == iy ? l0 : 0.f) + (i1 == iy ? l1 : 0.f) : 0.f;
                bilin_src(oxs[a], rscale, W, i0, i1, l0, l1);
                wx[a] = vx ? (i0 == ix ? l0 : 0.f) + (i1 == ix ? l1 : 0.f) : 0.f;
            }
            Pack<T, V> g[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) g[a][b].load(dy + (((long long)n * OH + oys[a]) * OW + oxs[b]) * C + cv * V);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float wgt = wy[a] * wx[b];
                    if (wgt != 0.f) {
#pragma unroll
                        for (int k = 0; k < V; ++k) acc[k] += wy[a] * wx[b] * g[a][b].v[k];
                    }
                }
            Pack<T, V> o2;
#pragma unroll
            for (int k = 0; k < V; ++k) o2.v[k] = acc[k];
            o2.store(dx + (size_t)pix * C + cv * V);
            continue;
        }
        const int oy_lo = max(0, scale * (iy - 1)), oy_hi = min(OH, scale * (iy + 2));
        const int ox_lo = max(0, scale * (ix - 1)), ox_hi = min(OW, scale * (ix + 2));
        for (int oy = oy_lo; oy < oy_hi; ++oy) {
            int y0, y1; float l0, l1;
            bilin_src(oy, rscale, H, y0, y1, l0, l1);
            float wy = (y0 == iy ? l0 : 0.f) + (y1 == iy ? l1 : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox < ox_hi; ++ox) {
                int x0, x1; float m0, m1;
                bilin_src(ox, rscale, W, x0, x1, m0, m1);
                float wx = (x0 == ix ? m0 : 0.f) + (x1 == ix ? m1 : 0.f);
                if (wx == 0.f) continue;
                Pack<T, V> g; g.load(dy + (((long long)n * OH + oy) * OW + ox) * C + cv * V);
#pragma unroll
                for (int k = 0; k < V; ++k) acc[k] += wy * wx * g.v[k];
            }
        }
        Pack<T, V> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.v[k] = acc[k];
        o.store(dx + (size_t)pix * C + cv * V);
    }
}

// ------------------------------------------------------------------ average pool k x k stride k (floor)
template <typename T>
__global__ __launch_bounds__(NT) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                         int C, int k) {
    const int OH = H / k, OW = W / k;
    const unsigned total = (unsigned)N * OH * OW * C;
    const float inv = 1.f / (float)(k * k);
    GRID_STRIDE(i, total) {
        int c = (int)(i % C);
        unsigned pix = i / C;
        int ox = (int)(pix % OW);
        unsigned t = pix / OW;
        int oy = (int)(t % OH), n = (int)(t / OH);
        float s = 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx)
                s += ur_to_float(x[(((long long)n * H + oy * k + dy) * W + ox * k + dx) * C + c]);
        y[i] = ur_from_float<T>(s * inv);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W,
                                                         int C, int k) {
    const int OH = H / k, OW = W / k;
    const unsigned total = (unsigned)N * H * W * C;
    const float inv = 1.f / (float)(k * k);
    GRID_STRIDE(i, total) {
        int c = (int)(i % C);
        unsigned pix = i / C;
        int ix = (int)(pix % W);
        unsigned t = pix / W;
        int iy = (int)(t % H), n = (int)(t / H);
        int oy = iy / k, ox = ix / k;
        float v = 0.f;
        if (oy < OH && ox < OW) v = ur_to_float(dy[(((long long)n * OH + oy) * OW + ox) * C + c]) * inv;
        dx[i] = ur_from_float<T>(v);
    }
}

// ------------------------------------------------------------------ channel-window copy with spatial pad / crop
template <typename T, int V>
__global__ __launch_bounds__(NT) void copy_channels_kernel(const T* __restrict__ src, int src_ld, int sh, int sw,
                                                           T* __restrict__ dst, int dst_ld, int dh, int dw, int pad_t,
                                                           int pad_l, int N, int C) {
    const int CV = C / V;
    const unsigned total = (unsigned)N * dh * dw * CV;
    GRID_STRIDE(i, total) {
        int cv = (int)(i % CV);
        unsigned pix = i / CV;
        int x = (int)(pix % dw);
        unsigned t = pix / dw;
        int y = (int)(t % dh), n = (int)(t / dh);
        int sy = y - pad_t, sx = x - pad_l;
        Pack<T, V> o; o.fill(0.f);
        if ((unsigned)sy < (unsigned)sh && (unsigned)sx < (unsigned)sw)
            o.load(src + (((long long)n * sh + sy) * sw + sx) * src_ld + cv * V);
        o.store(dst + (size_t)pix * dst_ld + cv * V);
    }
}

// Rows of C channels -> rows of OC >= C channels, zero filled (the generic kernels gather 8 channels per load, so a
// 3-channel image or a 19-class gradient is widened to a multiple of 8 first).  A workgroup moves 256 rows: the
// contiguous source span goes through LDS with 16-byte accesses, then every thread writes one widened row in 16-byte
// pieces.  OC * sizeof(T) must be a multiple of 16, both tensors 16-byte aligned, C * sizeof(T) <= 128.
template <typename T>
__global__ __launch_bounds__(NT) void pad_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, long long rows, int C, int OC) {
    __shared__ uint4 stage[NT * 128 / 16];
    T* tile = reinterpret_cast<T*>(stage);
    constexpr int EV = 16 / (int)sizeof(T);
    const long long nblocks = (rows + NT - 1) / NT;
    for (long long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const long long r0 = b * NT;
        const int nrows = (int)((rows - r0) < NT ? (rows - r0) : NT);
        if (nrows == NT) {
            const uint4* sp = reinterpret_cast<const uint4*>(src + r0 * C);
            for (int i = threadIdx.x; i < NT * C / EV; i += NT) stage[i] = sp[i];
        } else {
            for (int i = threadIdx.x; i < nrows * C; i += NT) tile[i] = src[r0 * C + i];
        }
        __syncthreads();
        if ((int)threadIdx.x < nrows) {
            const T* row = tile + threadIdx.x * C;
            uint4* out = reinterpret_cast<uint4*>(dst + (r0 + threadIdx.x) * OC);
            for (int q = 0; q < OC / EV; ++q) {
                union { T e[EV]; uint4 v; } o;
#pragma unroll
                for (int k = 0; k < EV; ++k) o.e[k] = q * EV + k < C ? row[q * EV + k] : ur_from_float<T>(0.f);
                out[q] = o.v;
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void channel_shuffle_kernel(const T* __restrict__ x, T* __restrict__ y, long long rows,
                                                             int C, int groups) {
    const int cpg = C / groups;
    const long long total = rows * C;
    GRID_STRIDE(i, total) {
        int co = (int)(i % C);
        unsigned r = i / C;
        int j = co / groups, g = co - j * groups;      // out channel j*groups + g  <-  in channel g*cpg + j
        y[i] = x[(size_t)r * C + g * cpg + j];
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void add_inplace_kernel(T* __restrict__ a, const T* __restrict__ b, long long count) {
    const long long n4 = count / 4;
    GRID_STRIDE(i, n4) {
        float4 u = ur_load4(a + (size_t)i * 4), v = ur_load4(b + (size_t)i * 4);
        ur_store4(a + (size_t)i * 4, make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        long long i = n4 * 4 + threadIdx.x;
        a[i] = ur_from_float<T>(ur_to_float(a[i]) + ur_to_float(b[i]));
    }
}

// ------------------------------------------------------------------ activations (S/C:17-34)
__device__ __forceinline__ float act_apply(float x, int act, float slope) {
    switch (act) {
    case UR_ACT_RELU: return fmaxf(x, 0.f);
    case UR_ACT_RELU6: return fminf(fmaxf(x, 0.f), 6.f);
    case UR_ACT_LEAKYRELU: return x > 0.f ? x : x * slope;
    case UR_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    case UR_ACT_HSWISH: return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
    }
    return x;
}
__device__ __forceinline__ float act_grad(float x, int act, float slope) {
    switch (act) {
    case UR_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case UR_ACT_RELU6: return (x > 0.f && x < 6.f) ? 1.f : 0.f;
    case UR_ACT_LEAKYRELU: return x > 0.f ? 1.f : slope;
    case UR_ACT_GELU: {
        float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
        float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
        return cdf + x * pdf;
    }
    // torch's hardswish_backward: the sloped piece is the open interval (-3, 3); at -3 the gradient is 0, at 3 it is 1
    case UR_ACT_HSWISH: return x <= -3.f ? 0.f : (x < 3.f ? (2.f * x + 3.f) * (1.f / 6.f) : 1.f);
    }
    return 1.f;
}

template <typename T>
__global__ __launch_bounds__(NT) void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long count, int act,
                                                     float slope) {
    GRID_STRIDE(i, count) y[i] = ur_from_float<T>(act_apply(ur_to_float(x[i]), act, slope));
}
template <typename T>
__global__ __launch_bounds__(NT) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx,
                                                     long long count, int act, float slope) {
    GRID_STRIDE(i, count) dx[i] = ur_from_float<T>(ur_to_float(dy[i]) * act_grad(ur_to_float(x[i]), act, slope));
}

// PReLU (nn.PReLU(num_parameters = nw, init), S/C:26-27): y = x > 0 ? x : a[c] * x with a learnable slope per channel
// (nw == C) or one for all (nw == 1); channels are the fastest dimension (NHWC, or [rows][C]).  Backward writes dx and the
// per-element slope-gradient terms t = dy * x * [x <= 0] in fp32; their column sums (ur_colsum) are the slope's gradient.
template <typename T>
__global__ __launch_bounds__(NT) void prelu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ a, int nw, T* __restrict__ y,
                                                       long long count, int C) {
    GRID_STRIDE(i, count) {
        const float v = ur_to_float(x[i]);
        y[i] = ur_from_float<T>(v > 0.f ? v : a[nw == 1 ? 0 : (int)(i % C)] * v);
    }
}
template <typename T>
__global__ __launch_bounds__(NT) void prelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ a,
                                                       int nw, T* __restrict__ dx, float* __restrict__ t, long long count, int C) {
    GRID_STRIDE(i, count) {
        const float v = ur_to_float(x[i]), g = ur_to_float(dy[i]);
        dx[i] = ur_from_float<T>(v > 0.f ? g : a[nw == 1 ? 0 : (int)(i % C)] * g);
        t[i] = v > 0.f ? 0.f : g * v;
    }
}

// ------------------------------------------------------------------ [B][R][S] -> [B][S][R] with dtype conversion
template <typename TS, typename TD>
__global__ void transpose_cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int R, int S) {
    __shared__ float tile[32][33];
    const long long b = blockIdx.y;
    const int tiles_s = (S + 31) / 32;
    const int r0 = (blockIdx.x / tiles_s) * 32, s0 = (blockIdx.x % tiles_s) * 32;
    const TS* sp = src + b * (long long)R * S;
    TD* dp = dst + b * (long long)R * S;
    for (int j = threadIdx.y; j < 32; j += 8) {
        int r = r0 + j, s = s0 + threadIdx.x;
        if (r < R && s < S) tile[j][threadIdx.x] = ur_to_float(sp[(long long)r * S + s]);
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        int s = s0 + j, r = r0 + threadIdx.x;
        if (r < R && s < S) dp[(long long)s * R + r] = ur_from_float<TD>(tile[threadIdx.x][j]);
    }
}

// Few planes (an image: R = 1..4 channel planes of S pixels) -> pixel-major.  The 32 x 32 tile above would be nine
// tenths empty; here a thread takes 8 consecutive pixels: two 16-byte loads per plane, 8 R contiguous outputs
// written as 16-byte pieces.  Needs S % 8 == 0 and 16-byte aligned tensors.
template <typename TD, int R>
__global__ __launch_bounds__(NT) void planes_to_pixels_kernel(const float* __restrict__ src, TD* __restrict__ dst, long long S,
                                                              long long groups) {
    constexpr int PIECES = 8 * R * (int)sizeof(TD) / 16;
    for (long long gidx = blockIdx.x * (long long)NT + threadIdx.x; gidx < groups; gidx += (long long)gridDim.x * NT) {
        const long long per_img = S / 8;
        const long long b = gidx / per_img, s0 = (gidx - b * per_img) * 8;
        const float* sp = src + b * R * S + s0;
        float v[R][8];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 lo = *reinterpret_cast<const float4*>(sp + r * S), hi = *reinterpret_cast<const float4*>(sp + r * S + 4);
            v[r][0] = lo.x; v[r][1] = lo.y; v[r][2] = lo.z; v[r][3] = lo.w;
            v[r][4] = hi.x; v[r][5] = hi.y; v[r][6] = hi.z; v[r][7] = hi.w;
        }
        union { TD e[8 * R]; uint4 q[PIECES]; } o;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) o.e[i * R + r] = ur_from_float<TD>(v[r][i]);
        uint4* dp = reinterpret_cast<uint4*>(dst + (b * S + s0) * R);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) dp[i] = o.q[i];
    }
}

template <typename TD>
bool launch_planes(const void* src, void* dst, int B, int R, int S, hipStream_t st) {
    if (R < 1 || R > 4 || S % 8 != 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return false;
    const long long groups = (long long)B * (S / 8);
    const int grid = ur_grid_1d(groups, NT);
    const float* s = (const float*)src;
    TD* d = (TD*)dst;
    switch (R) {
        case 1: hipLaunchKernelGGL((planes_to_pixels_kernel<TD, 1>), dim3(grid), dim3(NT), 0, st, s, d, (long long)S, groups); break;
        case 2: hipLaunchKernelGGL((planes_to_pixels_kernel<TD, 2>), dim3(grid), dim3(NT), 0, st, s, d, (long long)S, groups); break;
        case 3: hipLaunchKernelGGL((planes_to_pixels_kernel<TD, 3>), dim3(grid), dim3(NT), 0, st, s, d, (long long)S, groups); break;
        default: hipLaunchKernelGGL((planes_to_pixels_kernel<TD, 4>), dim3(grid), dim3(NT), 0, st, s, d, (long long)S, groups); break;
    }
    return true;
}

template <typename TS, typename TD>
int launch_transpose(const void* src, void* dst, int B, int R, int S, hipStream_t st) {
    if constexpr (std::is_same<TS, float>::value) {
        if (launch_planes<TD>(src, dst, B, R, S, st)) return ur_check_launch("planes_to_pixels");
    }
    const long long tiles = (long long)ur_cdiv(S, 32) * ur_cdiv(R, 32);
    if (tiles >= (1LL << 31) || B > 65535) { ur_set_error("transpose: dimension too large"); return UR_ERR_INVALID; }
    dim3 grid((unsigned)tiles, B, 1);
    hipLaunchKernelGGL((transpose_cast_kernel<TS, TD>), grid, dim3(32, 8), 0, st, (const TS*)src, (TD*)dst, R, S);
    return ur_check_launch("transpose_cast");
}

int transpose_dispatch(const void* src, int sdt, void* dst, int ddt, int B, int R, int S, hipStream_t st) {
    if (sdt == UR_F32 && ddt == UR_F32) return launch_transpose<float, float>(src, dst, B, R, S, st);
    if (sdt == UR_F32 && ddt == UR_BF16) return launch_transpose<float, bf16_t>(src, dst, B, R, S, st);
    if (sdt == UR_BF16 && ddt == UR_F32) return launch_transpose<bf16_t, float>(src, dst, B, R, S, st);
    if (sdt == UR_BF16 && ddt == UR_BF16) return launch_transpose<bf16_t, bf16_t>(src, dst, B, R, S, st);
    ur_set_error("transpose: unsupported dtypes %d -> %d", sdt, ddt);
    return UR_ERR_INVALID;
}

bool al(const void* p, int bytes) { return ((uintptr_t)p % bytes) == 0; }
#define UR_REQUIRE_32BIT(total, what) UR_REQUIRE((long long)(total) < (1LL << 31), what ": more than 2^31 elements")
int esize(int dtype) { return dtype == UR_BF16 ? 2 : 4; }

// ------------------------------------------------------------------ input packing (the step before the hot path)
// HWC uint8 image -> NHWC activation: out[c] = (float(px[c]) - sub[c]) * mul[c] for c < C, 0 for C <= c < OC
// (two separately rounded fp32 operations: albumentations' Normalize computes img -= mean*255; img *= 1/(std*255));
// with `divide` the second operation is a correctly rounded division by mul[c] (image / 255.0).
struct PackConsts { float sub[4], mul[4]; int divide; };
template <typename T>
__global__ __launch_bounds__(NT) void pack_image_kernel(const uint8_t* __restrict__ img, T* __restrict__ out, long long pixels,
                                                        int C, int OC, PackConsts k) {
    for (long long p = blockIdx.x * (long long)NT + threadIdx.x; p < pixels; p += (long long)gridDim.x * NT) {
        const uint8_t* px = img + p * C;
        T* o = out + p * OC;
        for (int c = 0; c < OC; ++c) {
            float v = 0.f;
            if (c < C) {
                const float d = __fsub_rn((float)px[c], k.sub[c]);
                v = k.divide ? __fdiv_rn(d, k.mul[c]) : __fmul_rn(d, k.mul[c]);
            }
            o[c] = ur_from_float<T>(v);
        }
    }
}

// label ids -> training classes through a 256-entry table (Cityscapes: 34 ids -> 19 classes, everything else 255)
struct LabelLut { uint8_t v[256]; };
__global__ __launch_bounds__(NT) void remap_labels_kernel(const uint8_t* __restrict__ ids, int64_t* __restrict__ out,
                                                          long long count, LabelLut lut) {
    __shared__ uint8_t tab[256];
    tab[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < count; i += (long long)gridDim.x * NT) out[i] = tab[ids[i]];
}

}  // namespace

#define SPATIAL_LAUNCH(kernel, total, ...)                                                                     \
    UR_DISPATCH_DTYPE(dtype, {                                                                                 \
        if (vec8) hipLaunchKernelGGL((kernel<T, 8>), dim3(ur_grid_1d((total) / 8, NT)), dim3(NT), 0, (hipStream_t)s, __VA_ARGS__); \
        else if (vec) hipLaunchKernelGGL((kernel<T, 4>), dim3(ur_grid_1d((total) / 4, NT)), dim3(NT), 0, (hipStream_t)s, __VA_ARGS__); \
        else hipLaunchKernelGGL((kernel<T, 1>), dim3(ur_grid_1d((total), NT)), dim3(NT), 0, (hipStream_t)s, __VA_ARGS__);        \
    })

namespace {
// One wave that does nothing until `ticks` of the constant 100 MHz counter have passed (time-bounded: it always ends).
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace

extern "C" {

int ur_maxpool2x2_fwd(const void* x, void* y, uint8_t* idx, int dtype, int n, int h, int w, int c, int x_ld, ur_stream_t s) {
    UR_REQUIRE(x && y && idx && n > 0 && h >= 2 && w >= 2 && c > 0, "maxpool_fwd: bad arguments");
    if (x_ld == 0) x_ld = c;
    UR_REQUIRE(x_ld >= c, "maxpool_fwd: input pitch %d smaller than the channel count %d", x_ld, c);
    const bool vec = c % 4 == 0 && x_ld % 4 == 0 && al(x, 4 * esize(dtype)) && al(y, 4 * esize(dtype));
    const bool vec8 = dtype == UR_BF16 && c % 8 == 0 && x_ld % 8 == 0 && al(x, 16) && al(y, 16) && al(idx, 8);
    const long long total = (long long)n * (h / 2) * (w / 2) * c;
    UR_REQUIRE_32BIT(total, "maxpool_fwd");
    SPATIAL_LAUNCH(maxpool_fwd_kernel, total, (const T*)x, (T*)y, idx, n, h, w, c, x_ld);
    return ur_check_launch("maxpool_fwd");
}

int ur_maxpool2x2_bwd(const void* dy, const uint8_t* idx, void* dx, const void* addend, int add_ld, int dtype, int n, int h,
                      int w, int c, ur_stream_t s) {
    UR_REQUIRE(dy && dx && idx && n > 0 && h >= 2 && w >= 2 && c > 0, "maxpool_bwd: bad arguments");
    if (add_ld == 0) add_ld = c;
    UR_REQUIRE(!addend || add_ld >= c, "maxpool_bwd: addend pitch %d is smaller than the channel count %d", add_ld, c);
    const bool vec = c % 4 == 0 && al(dy, 4 * esize(dtype)) && al(dx, 4 * esize(dtype)) &&
                     (!addend || (al(addend, 4 * esize(dtype)) && add_ld % 4 == 0));
    const bool vec8 = dtype == UR_BF16 && c % 8 == 0 && al(dy, 16) && al(dx, 16) && al(idx, 8) &&
                      (!addend || (al(addend, 16) && add_ld % 8 == 0));
    const long long total = (long long)n * h * w * c;
    UR_REQUIRE_32BIT(total, "maxpool_bwd");
    SPATIAL_LAUNCH(maxpool_bwd_kernel, total, (const T*)dy, idx, (T*)dx, (const T*)addend, add_ld, n, h, w, c);
    return ur_check_launch("maxpool_bwd");
}

int ur_upsample_bilinear_fwd(const void* x, void* y, int dtype, int n, int h, int w, int c, int scale, ur_stream_t s) {
    UR_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && scale >= 1, "upsample_fwd: bad arguments");
    const bool vec = c % 4 == 0 && al(x, 4 * esize(dtype)) && al(y, 4 * esize(dtype));
    const bool vec8 = dtype == UR_BF16 && c % 8 == 0 && al(x, 16) && al(y, 16);
    const long long total = (long long)n * h * scale * w * scale * c;
    UR_REQUIRE_32BIT(total, "upsample_fwd");
    const float rscale = (float)(1.0 / (double)scale);
    if ((long long)n * h * scale <= 65535) {
        const unsigned rows = (unsigned)(n * h * scale);
        const int ow = w * scale;
        UR_DISPATCH_DTYPE(dtype, {
            if (vec8) hipLaunchKernelGGL((upsample_fwd_rows_kernel<T, 8>), dim3(ur_cdiv((long long)ow * (c / 8), NT), rows), dim3(NT), 0, (hipStream_t)s, (const T*)x, (T*)y, h, w, c, scale, rscale);
            else if (vec) hipLaunchKernelGGL((upsample_fwd_rows_kernel<T, 4>), dim3(ur_cdiv((long long)ow * (c / 4), NT), rows), dim3(NT), 0, (hipStream_t)s, (const T*)x, (T*)y, h, w, c, scale, rscale);
            else hipLaunchKernelGGL((upsample_fwd_rows_kernel<T, 1>), dim3(ur_cdiv((long long)ow * c, NT), rows), dim3(NT), 0, (hipStream_t)s, (const T*)x, (T*)y, h, w, c, scale, rscale);
        });
        return ur_check_launch("upsample_fwd");
    }
    SPATIAL_LAUNCH(upsample_fwd_kernel, total, (const T*)x, (T*)y, n, h, w, c, scale, rscale);
    return ur_check_launch("upsample_fwd");
}

int ur_upsample_bilinear_bwd(const void* dy, void* dx, int dtype, int n, int h, int w, int c, int scale, ur_stream_t s) {
    UR_REQUIRE(dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && scale >= 1, "upsample_bwd: bad arguments");
    const bool vec = c % 4 == 0 && al(dy, 4 * esize(dtype)) && al(dx, 4 * esize(dtype));
    const bool vec8 = dtype == UR_BF16 && c % 8 == 0 && al(dy, 16) && al(dx, 16);
    const long long total = (long long)n * h * w * c;
    UR_REQUIRE_32BIT(total, "upsample_bwd");
    const float rscale = (float)(1.0 / (double)scale);
    SPATIAL_LAUNCH(upsample_bwd_kernel, total, (const T*)dy, (T*)dx, n, h, w, c, scale, rscale);
    return ur_check_launch("upsample_bwd");
}

int ur_avgpool_fwd(const void* x, void* y, int dtype, int n, int h, int w, int c, int k, ur_stream_t s) {
    UR_REQUIRE(x && y && n > 0 && k > 0 && h >= k && w >= k && c > 0, "avgpool_fwd: bad arguments");
    const long long total = (long long)n * (h / k) * (w / k) * c;
    UR_REQUIRE_32BIT(total, "avgpool_fwd");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((avgpool_fwd_kernel<T>), dim3(ur_grid_1d(total, NT)), dim3(NT), 0,
                                                (hipStream_t)s, (const T*)x, (T*)y, n, h, w, c, k));
    return ur_check_launch("avgpool_fwd");
}

int ur_avgpool_bwd(const void* dy, void* dx, int dtype, int n, int h, int w, int c, int k, ur_stream_t s) {
    UR_REQUIRE(dy && dx && n > 0 && k > 0 && h >= k && w >= k && c > 0, "avgpool_bwd: bad arguments");
    const long long total = (long long)n * h * w * c;
    UR_REQUIRE_32BIT(total, "avgpool_bwd");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((avgpool_bwd_kernel<T>), dim3(ur_grid_1d(total, NT)), dim3(NT), 0,
                                                (hipStream_t)s, (const T*)dy, (T*)dx, n, h, w, c, k));
    return ur_check_launch("avgpool_bwd");
}

int ur_copy_channels(const void* src, int src_ld, int src_coff, int sh, int sw, void* dst, int dst_ld, int dst_coff,
                     int dh, int dw, int pad_t, int pad_l, int n, int c, int dtype, ur_stream_t s) {
    UR_REQUIRE(src && dst && n > 0 && c > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0, "copy_channels: bad arguments");
    UR_REQUIRE(src_coff >= 0 && dst_coff >= 0 && src_coff + c <= src_ld && dst_coff + c <= dst_ld,
               "copy_channels: channel window outside the buffer pitch");
    const int es = esize(dtype);
    const char* sp = (const char*)src + (size_t)src_coff * es;
    char* dp = (char*)dst + (size_t)dst_coff * es;
    const bool vec = c % 4 == 0 && src_ld % 4 == 0 && dst_ld % 4 == 0 && al(sp, 4 * es) && al(dp, 4 * es);
    const bool vec8 = dtype == UR_BF16 && c % 8 == 0 && src_ld % 8 == 0 && dst_ld % 8 == 0 && al(sp, 16) && al(dp, 16);
    const long long total = (long long)n * dh * dw * c;
    UR_REQUIRE_32BIT(total, "copy_channels");
    SPATIAL_LAUNCH(copy_channels_kernel, total, (const T*)sp, src_ld, sh, sw, (T*)dp, dst_ld, dh, dw, pad_t, pad_l, n, c);
    return ur_check_launch("copy_channels");
}

int ur_pad_channels(const void* src, int c, void* dst, int dst_c, long long rows, int dtype, ur_stream_t s) {
    UR_REQUIRE(src && dst && rows > 0 && c > 0 && dst_c >= c, "pad_channels: bad arguments");
    const int es = esize(dtype);
    UR_REQUIRE((dst_c * es) % 16 == 0 && c * es <= 128 && al(src, 16) && al(dst, 16),
               "pad_channels: needs 16-byte aligned tensors, output rows of whole 16-byte pieces and input rows of at most 128 bytes");
    const int grid = ur_grid_1d(rows, NT, 256 * 8);
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pad_rows_kernel<T>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const T*)src,
                                                (T*)dst, rows, c, dst_c));
    return ur_check_launch("pad_channels");
}

int ur_channel_shuffle(const void* x, void* y, int dtype, long long rows, int c, int groups, ur_stream_t s) {
    UR_REQUIRE(x && y && rows > 0 && c > 0 && groups > 0 && c % groups == 0, "channel_shuffle: bad arguments");
    const long long total = rows * c;
    UR_REQUIRE_32BIT(total, "channel_shuffle");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((channel_shuffle_kernel<T>), dim3(ur_grid_1d(total, NT)), dim3(NT), 0,
                                                (hipStream_t)s, (const T*)x, (T*)y, rows, c, groups));
    return ur_check_launch("channel_shuffle");
}

int ur_spin(int microseconds, ur_stream_t s) {
    UR_REQUIRE(microseconds > 0 && microseconds <= 100000, "spin: 1..100000 microseconds");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, (long long)microseconds * 100);
    return ur_check_launch("spin");
}

int ur_add_inplace(void* a, const void* b, int dtype, long long count, ur_stream_t s) {
    UR_REQUIRE(a && b && count > 0, "add_inplace: bad arguments");
    UR_REQUIRE(al(a, 4 * esize(dtype)) && al(b, 4 * esize(dtype)), "add_inplace: buffers must be 4-element aligned");
    UR_REQUIRE_32BIT(count, "add_inplace");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((add_inplace_kernel<T>), dim3(ur_grid_1d(count / 4 + 1, NT)), dim3(NT), 0,
                                                (hipStream_t)s, (T*)a, (const T*)b, count));
    return ur_check_launch("add_inplace");
}

int ur_act_fwd(const void* x, void* y, int dtype, long long count, int act, float slope, ur_stream_t s) {
    UR_REQUIRE(x && y && count > 0 && act >= UR_ACT_NONE && act <= UR_ACT_HSWISH, "act_fwd: bad arguments");
    UR_REQUIRE_32BIT(count, "act_fwd");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((act_fwd_kernel<T>), dim3(ur_grid_1d(count, NT)), dim3(NT), 0,
                                                (hipStream_t)s, (const T*)x, (T*)y, count, act, slope));
    return ur_check_launch("act_fwd");
}

int ur_act_bwd(const void* dy, const void* x, void* dx, int dtype, long long count, int act, float slope, ur_stream_t s) {
    UR_REQUIRE(dy && x && dx && count > 0 && act >= UR_ACT_NONE && act <= UR_ACT_HSWISH, "act_bwd: bad arguments");
    UR_REQUIRE_32BIT(count, "act_bwd");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(ur_grid_1d(count, NT)), dim3(NT), 0,
                                                (hipStream_t)s, (const T*)dy, (const T*)x, (T*)dx, count, act, slope));
    return ur_check_launch("act_bwd");
}

int ur_prelu_fwd(const void* x, const float* slope, int nslope, void* y, int dtype, long long count, int c, ur_stream_t s) {
    UR_REQUIRE(x && slope && y && count > 0 && c > 0 && count % c == 0 && (nslope == 1 || nslope == c), "prelu_fwd: bad arguments");
    UR_REQUIRE_32BIT(count, "prelu_fwd");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((prelu_fwd_kernel<T>), dim3(ur_grid_1d(count, NT)), dim3(NT), 0, (hipStream_t)s,
                                                (const T*)x, slope, nslope, (T*)y, count, c));
    return ur_check_launch("prelu_fwd");
}

int ur_prelu_bwd(const void* dy, const void* x, const float* slope, int nslope, void* dx, float* slope_terms, int dtype,
                 long long count, int c, ur_stream_t s) {
    UR_REQUIRE(dy && x && slope && dx && slope_terms && count > 0 && c > 0 && count % c == 0 && (nslope == 1 || nslope == c),
               "prelu_bwd: bad arguments");
    UR_REQUIRE_32BIT(count, "prelu_bwd");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((prelu_bwd_kernel<T>), dim3(ur_grid_1d(count, NT)), dim3(NT), 0, (hipStream_t)s,
                                                (const T*)dy, (const T*)x, slope, nslope, (T*)dx, slope_terms, count, c));
    return ur_check_launch("prelu_bwd");
}

int ur_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int n, int c, int h, int w, ur_stream_t s) {
    UR_REQUIRE(src && dst && n > 0 && c > 0 && h > 0 && w > 0, "nchw_to_nhwc: bad arguments");
    return transpose_dispatch(src, src_dtype, dst, dst_dtype, n, c, h * w, (hipStream_t)s);
}

int ur_nhwc_to_nchw(const void* src, int src_dtype, void* dst, int dst_dtype, int n, int c, int h, int w, ur_stream_t s) {
    UR_REQUIRE(src && dst && n > 0 && c > 0 && h > 0 && w > 0, "nhwc_to_nchw: bad arguments");
    return transpose_dispatch(src, src_dtype, dst, dst_dtype, n, h * w, c, (hipStream_t)s);
}

int ur_pack_image_u8(const uint8_t* img, void* out, int dtype, long long pixels, int c, int out_c, const float* sub,
                     const float* mul, int divide, ur_stream_t s) {
    UR_REQUIRE(img && out && sub && mul && pixels > 0 && c > 0 && c <= 4 && out_c >= c, "pack_image_u8: bad arguments");
    PackConsts k{};
    k.divide = divide != 0;
    for (int i = 0; i < c; ++i) { k.sub[i] = sub[i]; k.mul[i] = mul[i]; }
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pack_image_kernel<T>), dim3(ur_grid_1d(pixels, NT)), dim3(NT), 0, (hipStream_t)s,
                                                img, (T*)out, pixels, c, out_c, k));
    return ur_check_launch("pack_image_u8");
}

int ur_remap_labels_u8(const uint8_t* ids, int64_t* out, long long count, const uint8_t* lut256, ur_stream_t s) {
    UR_REQUIRE(ids && out && lut256 && count > 0, "remap_labels_u8: bad arguments");
    static_assert(NT == 256, "the table is loaded one entry per thread");
    LabelLut lut;
    for (int i = 0; i < 256; ++i) lut.v[i] = lut256[i];
    hipLaunchKernelGGL(remap_labels_kernel, dim3(ur_grid_1d(count, NT)), dim3(NT), 0, (hipStream_t)s, ids, out, count, lut);
    return ur_check_launch("remap_labels_u8");
}

}  // extern "C"
