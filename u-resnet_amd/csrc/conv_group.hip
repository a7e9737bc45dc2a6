// Grouped 3x3 convolutions with 1 or 4 channels per group (EUCB depthwise, EUCB* groups=C/4),
// stride 1, pad 1, NHWC.  These are bandwidth-bound (<= 36 MAC per loaded element): a direct
// kernel, one thread per (pixel, 4-channel quad), 8/16-byte coalesced accesses, fp32 math.
// CPG == 4: the quad is one group (4 in x 4 out).  CPG == 1: the quad is 4 independent channels.
// Weights stay in PyTorch layout w[c_out][cpg][3][3] fp32 and are read through the cache.
#include "ur_common.h"
#include "wgrad32_tile.h"

namespace {

constexpr int NT = 256;

// 4 consecutive channels as loaded (8 bytes of bf16 / 16 bytes of fp32); converted to fp32 only when consumed,
// so a software-prefetched pixel costs 2 (bf16) or 4 (fp32) registers per tap
template <typename T> struct Raw4;
template <> struct Raw4<bf16_t> {
    bf16x4_t v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x4_t*>(p); }
    __device__ __forceinline__ void zero() { v = bf16x4_t{0, 0, 0, 0}; }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
};
template <> struct Raw4<float> {
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ float get(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
};

// out[a] = sum_{tap,b} in[p + off(tap)][b] * Wt(a, b, tap)
//   forward       : off = (ky-1, kx-1), Wt = w[q*4+a][b][tap]
//   backward-data : off = (1-ky, 1-kx), Wt = w[q*4+b][a][tap]
template <typename T, int CPG, bool DGRAD>
__global__ __launch_bounds__(NT) void gconv3x3_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                      T* __restrict__ out, int N, int H, int W, int C) {
    const int Q = C / 4;
    const long long total = (long long)N * H * W * Q;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int q = (int)(i % Q);
        const long long pix = i / Q;
        const int x = (int)(pix % W);
        const long long t = pix / W;
        const int y = (int)(t % H), n = (int)(t / H);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* wq = w + (size_t)q * 4 * CPG * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = DGRAD ? y + 1 - ky : y + ky - 1;
            if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = DGRAD ? x + 1 - kx : x + kx - 1;
                if ((unsigned)xx >= (unsigned)W) continue;
                const float4 v = ur_load4(in + (((long long)n * H + yy) * W + xx) * C + q * 4);
                const float vi[4] = {v.x, v.y, v.z, v.w};
                const int tap = ky * 3 + kx;
                if constexpr (CPG == 1) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a] += vi[a] * wq[a * 9 + tap];
                } else {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[a] += vi[b] * (DGRAD ? wq[(b * 4 + a) * 9 + tap] : wq[(a * 4 + b) * 9 + tap]);
                }
            }
        }
        ur_store4(out + pix * C + q * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// Same arithmetic with the quad's weights held in registers: when the number of quads divides the
// workgroup size every thread keeps ONE quad for its whole grid-stride walk over pixels, so the
// 36 / 144 weights are loaded once per thread instead of once per pixel.
template <typename T, int CPG, bool DGRAD>
__global__ __launch_bounds__(NT) void gconv3x3_regw_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                           T* __restrict__ out, int N, int H, int W, int C) {
    constexpr int NV = 4 * CPG;
    const int Q = C / 4;                         // Q divides NT (checked on the host)
    const int q = threadIdx.x % Q, pslot = threadIdx.x / Q, PS = NT / Q;
    float wr[9][NV];
    const float* wq = w + (size_t)q * 4 * CPG * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if constexpr (CPG == 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a) wr[tap][a] = wq[a * 9 + tap];
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) wr[tap][a * 4 + b] = DGRAD ? wq[(b * 4 + a) * 9 + tap] : wq[(a * 4 + b) * 9 + tap];
        }
    }
    const int P = N * H * W;                     // < 2^31 (checked on the host): 32-bit index arithmetic
    const int pstep = gridDim.x * PS;
    // the 9 taps of the NEXT pixel are in flight while the current pixel is multiplied
    auto load_taps = [&](int pix, Raw4<T>* taps) {
        const int t = pix / W;
        const int x = pix - t * W;
        const int n = t / H;
        const int y = t - n * H;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = DGRAD ? y + 1 - ky : y + ky - 1;
            const bool yok = (unsigned)yy < (unsigned)H;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = DGRAD ? x + 1 - kx : x + kx - 1;
                if (yok && (unsigned)xx < (unsigned)W) taps[ky * 3 + kx].load(in + ((size_t)((n * H + yy) * W + xx)) * C + q * 4);
                else taps[ky * 3 + kx].zero();
            }
        }
    };
    Raw4<T> cur[9], nxt[9];
    int pix = blockIdx.x * PS + pslot;
    if (pix < P) load_taps(pix, cur);
    for (; pix < P; pix += pstep) {
        const bool more = pix + pstep < P;
        if (more) load_taps(pix + pstep, nxt);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float vi[4] = {cur[tap].get(0), cur[tap].get(1), cur[tap].get(2), cur[tap].get(3)};
            if constexpr (CPG == 1) {
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] += vi[a] * wr[tap][a];
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a] += vi[b] * wr[tap][a * 4 + b];
            }
        }
        ur_store4(out + (size_t)pix * C + q * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
        if (more) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) cur[tap] = nxt[tap];
        }
    }
}

// bf16, 4 channels per group, on the matrix cores.  v_mfma_f32_4x4x4_16b_bf16 multiplies 16 independent
// 4x4 blocks per wave: a block is (channel quad q, 4 consecutive pixels of a row).  Lane (block, j) supplies
// the 8-byte quad it loaded for pixel j as the B column and receives the 4 output channels of pixel j, so
// loads and stores keep the (pixel, quad) layout of the direct kernel; lane (block, i) supplies row i of the
// tap's 4x4 weight block (bf16, 18 registers for all 9 taps).  Every lane walks down a column strip of RS
// rows with a 3x3 register window: 3 loads per output instead of 9.
//   grid.x = image * row strips * column strips, grid.y = quad blocks of QB = min(Q, 64) quads
// STATS: also emit per-channel partial sums (sum, sum of squares of the fp32 results) for a following BatchNorm, one
// row per blockIdx.x: stats[gridDim.x][C][2].
// NQ = channel quads per lane: 2 where the channel count is a multiple of 8 (16-byte loads and stores; a lane then feeds
// two matrix instructions per tap, one per quad), else 1.  QB counts such lane units per workgroup.
template <int NQ> struct GcVec { typedef s16x4_t type; };
template <> struct GcVec<2> { typedef __attribute__((ext_vector_type(8))) short type; };

template <bool DGRAD, bool STATS, int NQ>
__global__ __launch_bounds__(NT) void gconv3x3_mfma_kernel(const bf16_t* __restrict__ in, const float* __restrict__ w,
                                                           bf16_t* __restrict__ out, int H, int W, int C, int QB, int RS,
                                                           int xstrips, int ystrips, float* __restrict__ stats) {
    typedef typename GcVec<NQ>::type vec_t;
    const int lane4 = threadIdx.x & 3;
    const int gb = threadIdx.x >> 2;                         // 0..63: block index within the workgroup
    const int u = blockIdx.y * QB + gb % QB;                 // lane unit: quads u * NQ .. u * NQ + NQ - 1
    const int pg = gb / QB;                                  // pixel group within the workgroup
    const int PG = 64 / QB;
    int bid = blockIdx.x;
    const int xs = bid % xstrips; bid /= xstrips;
    const int ys = bid % ystrips;
    const int n = bid / ystrips;
    const int x = (xs * PG + pg) * 4 + lane4;
    const int y0 = ys * RS;
    const int y1 = min(y0 + RS, H);

    // weight rows: forward A[i][k] = w[q*4+i][k][tap]; backward-data A[i][k] = w[q*4+k][i][tap]
    s16x4_t wt[NQ][9];
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
        const int q = u * NQ + h;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            bf16x4_t v;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = (bf16_t)(DGRAD ? w[((size_t)(q * 4 + k) * 4 + lane4) * 9 + tap] : w[((size_t)(q * 4 + lane4) * 4 + k) * 9 + tap]);
            wt[h][tap] = __builtin_bit_cast(s16x4_t, v);
        }
    }

    const bool xin = x < W;
    const bool xl = xin && x >= 1, xr = x + 1 < W;
    const bf16_t* img = in + (size_t)n * H * W * C + u * (NQ * 4);
    vec_t zero;
#pragma unroll
    for (int k = 0; k < NQ * 4; ++k) zero[k] = 0;
    auto load_row = [&](int yy, vec_t* row) {
        const bool yok = (unsigned)yy < (unsigned)H;
        const bf16_t* p = img + ((size_t)yy * W + x) * C;
        row[0] = (yok && xl) ? *reinterpret_cast<const vec_t*>(p - C) : zero;
        row[1] = (yok && xin) ? *reinterpret_cast<const vec_t*>(p) : zero;
        row[2] = (yok && xr) ? *reinterpret_cast<const vec_t*>(p + C) : zero;
    };
    auto quad = [&](const vec_t& v, int h) -> s16x4_t {
        if constexpr (NQ == 1) return v;
        else return h == 0 ? __builtin_shufflevector(v, v, 0, 1, 2, 3) : __builtin_shufflevector(v, v, 4, 5, 6, 7);
    };
    vec_t win[3][3];
    float st[NQ * 8];
#pragma unroll
    for (int k = 0; k < NQ * 8; ++k) st[k] = 0.f;
    load_row(y0 - 1, win[0]);
    load_row(y0, win[1]);
    for (int y = y0; y < y1; ++y) {
        load_row(y + 1, win[2]);
        f32x4_t acc[NQ];
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            acc[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc[h] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wt[h][ky * 3 + kx], quad(DGRAD ? win[2 - ky][2 - kx] : win[ky][kx], h),
                                                                    acc[h], 0, 0, 0);
        }
        if (xin) {
            bf16_t* o = out + (((size_t)n * H + y) * W + x) * C + u * (NQ * 4);
            if constexpr (NQ == 1) {
                ur_store4(o, make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]));
            } else {
                const bf16x8_t v = {(bf16_t)acc[0][0], (bf16_t)acc[0][1], (bf16_t)acc[0][2], (bf16_t)acc[0][3],
                                    (bf16_t)acc[1][0], (bf16_t)acc[1][1], (bf16_t)acc[1][2], (bf16_t)acc[1][3]};
                *reinterpret_cast<bf16x8_t*>(o) = v;
            }
        }
        if constexpr (STATS) {
            if (xin) {
#pragma unroll
                for (int h = 0; h < NQ; ++h)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { st[h * 8 + k] += acc[h][k]; st[h * 8 + 4 + k] += acc[h][k] * acc[h][k]; }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { win[0][k] = win[1][k]; win[1][k] = win[2][k]; }
    }
    if constexpr (STATS) {
        // thread t = ((pixel group * QB + local unit) << 2) + pixel lane; fixed summation order over the 4 * PG threads
        // that hold the same unit
        __shared__ float red[NT][NQ * 8];
#pragma unroll
        for (int k = 0; k < NQ * 8; ++k) red[threadIdx.x][k] = st[k];
        __syncthreads();
        for (int o = threadIdx.x; o < QB * NQ * 8; o += NT) {
            const int ul = o / (NQ * 8), k = o % (NQ * 8);
            float t = 0.f;
            for (int g = 0; g < PG; ++g)
#pragma unroll
                for (int l = 0; l < 4; ++l) t += red[((g * QB + ul) << 2) + l][k];
            const int ch = ((blockIdx.y * QB + ul) * NQ + (k >> 3)) * 4 + (k & 3);
            stats[((size_t)blockIdx.x * C + ch) * 2 + ((k >> 2) & 1)] = t;
        }
    }
}

// geometry the matrix-core kernel covers: lane units (one quad, or two where 8 divides the channel count) per workgroup
// divide 64, or 64 divides the unit count
static int gconv_nq(int c) { return c % 8 == 0 ? 2 : 1; }
static bool mfma_gconv_ok(int c, int cpg) {
    if (cpg != 4 || c % 4) return false;
    const int U = c / (4 * gconv_nq(c));
    return U >= 64 ? U % 64 == 0 : 64 % U == 0;
}

struct GconvPlan { int NQ, U, QB, xstrips, ystrips, RS; };
GconvPlan gconv_mfma_plan(int n, int h, int w_, int c) {
    GconvPlan p;
    p.NQ = gconv_nq(c);
    p.U = c / (4 * p.NQ);
    p.QB = p.U < 64 ? p.U : 64;
    const int PG = 64 / p.QB;
    p.xstrips = ur_cdiv(w_, 4 * PG);
    p.RS = 32;
    while (p.RS > 4 && (long long)n * ur_cdiv(h, p.RS) * p.xstrips * (p.U / p.QB) < 2048) p.RS >>= 1;
    p.ystrips = ur_cdiv(h, p.RS);
    return p;
}

template <bool DGRAD>
void launch_gconv_mfma(const bf16_t* in, const float* w, bf16_t* out, int n, int h, int w_, int c, hipStream_t st,
                       float* stats = nullptr) {
    const GconvPlan p = gconv_mfma_plan(n, h, w_, c);
    const dim3 grid((unsigned)(n * p.ystrips * p.xstrips), p.U / p.QB);
#define GC_LAUNCH(ST, NQ_)                                                                                                  \
    hipLaunchKernelGGL((gconv3x3_mfma_kernel<DGRAD, ST, NQ_>), grid, dim3(NT), 0, st, in, w, out, h, w_, c, p.QB, p.RS, p.xstrips, \
                       p.ystrips, stats)
    if (p.NQ == 2) { if (stats) GC_LAUNCH(true, 2); else GC_LAUNCH(false, 2); }
    else { if (stats) GC_LAUNCH(true, 1); else GC_LAUNCH(false, 1); }
#undef GC_LAUNCH
}

bool gconv_mfma_used(int dtype, int n, int h, int w_, int c, int cpg) {
    return dtype == UR_BF16 && mfma_gconv_ok(c, cpg) && (long long)n * ur_cdiv(h, 4) * ur_cdiv(w_, 4) < (1LL << 31);
}

template <typename T, bool DGRAD>
void launch_gconv(const T* in, const float* w, T* out, int n, int h, int w_, int c, int cpg, hipStream_t st) {
    const int Q = c / 4;
    const long long P = (long long)n * h * w_;
    if constexpr (ElemTraits<T>::dtype == UR_BF16) {
        if (gconv_mfma_used(UR_BF16, n, h, w_, c, cpg)) {
            launch_gconv_mfma<DGRAD>(in, w, out, n, h, w_, c, st);
            return;
        }
    }
    if (NT % Q == 0) {
        const int PS = NT / Q;
        const int grid = ur_grid_1d(P, PS, 256 * 8);
        if (cpg == 1) hipLaunchKernelGGL((gconv3x3_regw_kernel<T, 1, DGRAD>), dim3(grid), dim3(NT), 0, st, in, w, out, n, h, w_, c);
        else hipLaunchKernelGGL((gconv3x3_regw_kernel<T, 4, DGRAD>), dim3(grid), dim3(NT), 0, st, in, w, out, n, h, w_, c);
    } else {
        const int grid = ur_grid_1d(P * Q, NT);
        if (cpg == 1) hipLaunchKernelGGL((gconv3x3_kernel<T, 1, DGRAD>), dim3(grid), dim3(NT), 0, st, in, w, out, n, h, w_, c);
        else hipLaunchKernelGGL((gconv3x3_kernel<T, 4, DGRAD>), dim3(grid), dim3(NT), 0, st, in, w, out, n, h, w_, c);
    }
}

// Backward-weight: thread (slot, quad) accumulates its quad's CPG*4*9 products over the pixels
// slot, slot+S, ...; the slots of a workgroup are summed through LDS in a fixed order and each
// workgroup writes one partial row partial[block][C*CPG*9] (layout = PyTorch weight layout).
template <typename T, int CPG>
__global__ __launch_bounds__(NT) void gconv3x3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            float* __restrict__ partial, int N, int H, int W, int C) {
    // Thread = (pixel slot, kernel row ky, channel quad q): it accumulates the 3 taps of one kernel row for its
    // 4 output channels over a strided pixel walk (3 * 4 * CPG sums instead of 9 * 4 * CPG: 5 waves per SIMD
    // instead of 1, and the next pixel's 4 loads are in flight while the current one is multiplied).
    constexpr int NV = 4 * CPG;                   // values per tap held by a thread
    __shared__ float red[NT * NV];
    const int Q = C / 4;
    const int units = 3 * Q;                      // (ky, q) pairs
    const int ub = units < NT ? units : NT;
    const int slots = NT / ub;
    const int tid = threadIdx.x;
    const int slot = tid / ub, ul = tid - slot * ub;
    const int P = N * H * W;                      // < 2^31 (checked on the host)
    const size_t wsz = (size_t)C * CPG * 9;
    for (int ubase = 0; ubase < units; ubase += ub) {
        const int u = ubase + ul;
        const int ky = u / Q, q = u - ky * Q;
        float acc[3][NV];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[t][v] = 0.f;
        if (slot < slots && u < units) {
            const int pstep = gridDim.x * slots;
            auto load_pix = [&](int pix, Raw4<T>& g, Raw4<T>* taps) {
                const int t = pix / W;
                const int xx0 = pix - t * W;
                const int n = t / H;
                const int yy = t - n * H + ky - 1;
                g.load(dy + (size_t)pix * C + q * 4);
                const bool yok = (unsigned)yy < (unsigned)H;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = xx0 + kx - 1;
                    if (yok && (unsigned)xx < (unsigned)W) taps[kx].load(x + ((size_t)((n * H + yy) * W + xx)) * C + q * 4);
                    else taps[kx].zero();
                }
            };
            Raw4<T> gc, gn, cur[3], nxt[3];
            int pix = blockIdx.x * slots + slot;
            if (pix < P) load_pix(pix, gc, cur);
            for (; pix < P; pix += pstep) {
                const bool more = pix + pstep < P;
                if (more) load_pix(pix + pstep, gn, nxt);
                const float g[4] = {gc.get(0), gc.get(1), gc.get(2), gc.get(3)};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float vi[4] = {cur[kx].get(0), cur[kx].get(1), cur[kx].get(2), cur[kx].get(3)};
                    if constexpr (CPG == 1) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) acc[kx][a] += g[a] * vi[a];
                    } else {
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 4; ++b) acc[kx][a * 4 + b] += g[a] * vi[b];
                    }
                }
                if (more) {
                    gc = gn;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) cur[kx] = nxt[kx];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int v = 0; v < NV; ++v) red[v * NT + tid] = acc[t][v];
            __syncthreads();
            if (tid < ub && u < units) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float s = 0.f;
                    for (int sl = 0; sl < slots; ++sl) s += red[v * NT + sl * ub + tid];
                    // v = a (CPG==1) or a*4+b: weight index ((q*4+a)*CPG + b)*9 + ky*3 + t
                    const int a = CPG == 1 ? v : v >> 2, b = CPG == 1 ? 0 : v & 3;
                    partial[(size_t)blockIdx.x * wsz + ((size_t)(q * 4 + a) * CPG + b) * 9 + ky * 3 + t] = s;
                }
            }
            __syncthreads();
        }
    }
}

static bool wgrad_mfma_ok(int dtype, int c, int cpg, const void* x, const void* dy) {
    return dtype == UR_BF16 && cpg == 4 && c % 32 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0;
}
static int wgrad_mfma_splits(int n, int h, int w, int c) { return wg32::splits(n, h, w, c / 32); }

// dw[i] = sum_p partial[p][i], fixed summation order
template <int SL>
__global__ __launch_bounds__(256) void gconv_wgrad_reduce_kernel(const float* __restrict__ partial, int nparts, long long wsz,
                                                                 float* __restrict__ dw) {
    long long i;
    float t;
    if (ur_slab_sum<SL>(partial, wsz, nparts, i, t)) dw[i] = t;
}

int wgrad_blocks(long long P, int C) {
    int units = 3 * (C / 4);
    int ub = units < NT ? units : NT;
    int slots = NT / ub;
    long long g = (P + (long long)slots * 16 - 1) / ((long long)slots * 16);
    if (g < 1) g = 1;
    if (g > 1024) g = 1024;
    return (int)g;
}

int check_args(const void* a, const void* b, const void* c, int dtype, int n, int h, int w, int ch, int cpg) {
    UR_REQUIRE(a && b && c, "gconv3x3: null pointer");
    UR_REQUIRE(n > 0 && h > 0 && w > 0 && ch > 0, "gconv3x3: non-positive dimension");
    UR_REQUIRE((long long)n * h * w < (1LL << 31), "gconv3x3: more than 2^31 pixels");
    UR_REQUIRE(cpg == 1 || cpg == 4, "gconv3x3: channels per group must be 1 or 4 (got %d)", cpg);
    UR_REQUIRE(ch % 4 == 0, "gconv3x3: channel count %d is not a multiple of 4", ch);
    const int es = dtype == UR_BF16 ? 2 : 4;
    // 4-element accesses; 8-element (16-byte) ones in the bf16 matrix-core kernel where 8 divides the channel count
    const int al = (dtype == UR_BF16 && cpg == 4 && ch % 8 == 0) ? 8 * es : 4 * es;
    UR_REQUIRE(((uintptr_t)a % al) == 0 && ((uintptr_t)c % al) == 0, "gconv3x3: tensors must be %d-byte aligned", al);
    return UR_OK;
}

}  // namespace

extern "C" {

int ur_gconv3x3_fwd(const void* x, const float* w, void* y, int dtype, int n, int h, int w_, int c, int cpg, ur_stream_t s) {
    if (int e = check_args(x, w, y, dtype, n, h, w_, c, cpg)) return e;
    UR_DISPATCH_DTYPE(dtype, (launch_gconv<T, false>((const T*)x, w, (T*)y, n, h, w_, c, cpg, (hipStream_t)s)));
    return ur_check_launch("gconv3x3_fwd");
}

int ur_gconv3x3_stats_rows(int dtype, int n, int h, int w_, int c, int cpg) {
    if (n <= 0 || h <= 0 || w_ <= 0 || c <= 0 || !gconv_mfma_used(dtype, n, h, w_, c, cpg)) return 0;
    const GconvPlan p = gconv_mfma_plan(n, h, w_, c);
    return n * p.ystrips * p.xstrips;
}

int ur_gconv3x3_fwd_stats(const void* x, const float* w, void* y, float* stats, int dtype, int n, int h, int w_, int c, int cpg,
                          ur_stream_t s) {
    if (int e = check_args(x, w, y, dtype, n, h, w_, c, cpg)) return e;
    UR_REQUIRE(stats && gconv_mfma_used(dtype, n, h, w_, c, cpg),
               "gconv3x3_fwd_stats: statistics come from the bf16 4-channels-per-group kernel only (ur_gconv3x3_stats_rows == 0 otherwise)");
    launch_gconv_mfma<false>((const bf16_t*)x, w, (bf16_t*)y, n, h, w_, c, (hipStream_t)s, stats);
    return ur_check_launch("gconv3x3_fwd_stats");
}

int ur_gconv3x3_dgrad(const void* dy, const float* w, void* dx, int dtype, int n, int h, int w_, int c, int cpg, ur_stream_t s) {
    if (int e = check_args(dy, w, dx, dtype, n, h, w_, c, cpg)) return e;
    UR_DISPATCH_DTYPE(dtype, (launch_gconv<T, true>((const T*)dy, w, (T*)dx, n, h, w_, c, cpg, (hipStream_t)s)));
    return ur_check_launch("gconv3x3_dgrad");
}

size_t ur_gconv3x3_wgrad_workspace(int n, int h, int w_, int c, int cpg) {
    // enough for either kernel (the choice also depends on dtype and alignment)
    size_t rows = (size_t)wgrad_blocks((long long)n * h * w_, c);
    if (cpg == 4 && c % 32 == 0 && (size_t)wgrad_mfma_splits(n, h, w_, c) > rows) rows = wgrad_mfma_splits(n, h, w_, c);
    return rows * c * cpg * 9 * sizeof(float);
}

int ur_gconv3x3_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes, int dtype, int n, int h,
                      int w_, int c, int cpg, ur_stream_t s) {
    if (int e = check_args(x, dw, dy, dtype, n, h, w_, c, cpg)) return e;
    UR_REQUIRE(workspace && ws_bytes >= ur_gconv3x3_wgrad_workspace(n, h, w_, c, cpg), "gconv3x3_wgrad: workspace too small");
    const int grid = wgrad_blocks((long long)n * h * w_, c);
    float* partial = (float*)workspace;
    if (wgrad_mfma_ok(dtype, c, cpg, x, dy))      // 32-channel blocks on the matrix cores, diagonal 4x4 blocks kept
        return wg32::run<true>((const bf16_t*)x, (const bf16_t*)dy, dw, partial, n, h, w_, c, c, c, c, (hipStream_t)s);
    UR_DISPATCH_DTYPE(dtype, {
        if (cpg == 1) hipLaunchKernelGGL((gconv3x3_wgrad_kernel<T, 1>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const T*)x, (const T*)dy, partial, n, h, w_, c);
        else hipLaunchKernelGGL((gconv3x3_wgrad_kernel<T, 4>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const T*)x, (const T*)dy, partial, n, h, w_, c);
    });
    if (int e = ur_check_launch("gconv3x3_wgrad")) return e;
    const long long wsz = (long long)c * cpg * 9;
    if (ur_slab_lanes(grid) == 32)
        hipLaunchKernelGGL(gconv_wgrad_reduce_kernel<32>, dim3(ur_slab_grid(wsz, 32)), dim3(256), 0, (hipStream_t)s, partial, grid, wsz, dw);
    else
        hipLaunchKernelGGL(gconv_wgrad_reduce_kernel<8>, dim3(ur_slab_grid(wsz, 8)), dim3(256), 0, (hipStream_t)s, partial, grid, wsz, dw);
    return ur_check_launch("gconv3x3_wgrad_reduce");
}

}  // extern "C"
