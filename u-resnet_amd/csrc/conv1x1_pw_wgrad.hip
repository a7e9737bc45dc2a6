// Backward-weight of the pointwise (1x1, stride 1) bf16 convolutions with few channels on full-resolution activations
// (segmentation head 80 -> 19, EUCB projection 32 -> 16; reference
// Segmentation/Cityscapes/modules/U_ResNet.py:53-55,127-133):  dW[co][ci] = sum over pixels of dy[p][co] * x[p][ci].
//
// A few FLOP per byte: the kernel is a stream over the pixels.  The generic backward-weight GEMM pads these 16..32 x 32..96
// channel products to 128 x 128 tiles, eight times the matrix work and a third below the memory rate.  Here
//   * a wave owns 32-pixel chunks and never meets another wave in the loop: it fetches the chunk's x and dy rows as
//     consecutive 16-byte pieces (the next chunk's loads are in flight while the current one is multiplied), parks them in
//     its own LDS rows (odd pitch) and reads both back transposed (ds_read_b64_tr_b16: the reduction index is the pixel,
//     memory is channel-contiguous) as operands of v_mfma_f32_16x16x32_bf16, K = the chunk's 32 pixels;
//   * the accumulators (16 x 16 blocks of dW) stay in registers for the whole kernel; at the end the four waves of a
//     workgroup are summed in wave order through LDS and the workgroup writes one fp32 slab; a second kernel sums the
//     slabs in a fixed order (bitwise reproducible) and writes dW in PyTorch's [cout][cin] layout.
#include "ur_common.h"
#include <stdlib.h>

namespace {

constexpr int NT = 256;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

struct PwWgParams {
    const bf16_t* x;     // [P][ld_x], K channels used
    const bf16_t* dy;    // [P][ld_dy], M channels used
    float* ws;           // [workgroups][16 MB][16 KB]
    long long P;
    int K, M, ld_x, ld_dy, pitch_x, pitch_y;
    const float* x_aff;  // AFF: [2][K] per-channel scale then shift applied to x on its way into LDS (a folded BatchNorm)
    int x_relu;          // AFF: ReLU after the multiply-add
    int aff_off;         // AFF: byte offset of the constants inside the dynamic LDS (behind the staging rows and the sum buffer)
    // x as two dense tensors (a channel concatenation that was never materialised): channels [K1, K) come from x2 (pitch
    // ld_x2, constants x_aff2 = [2][K - K1]) when x2 is set; K1 is a multiple of 8
    const bf16_t* x2; int K1, ld_x2; const float* x_aff2;
    // BNS: [K] batch mean and inverse standard deviation of the BatchNorm folded into x_aff (both tensors' channels in order)
    const float* bn_mean; const float* bn_invstd;
};

// x -> bf16(max(x * scale + shift, 0)) on 8 consecutive channels; cs = the channels' scales in LDS, their shifts K floats on
__device__ __forceinline__ bf16x8_t pw_affine8(const bf16x8_t& v, const float* cs, int K, int relu) {
    const float4 s0 = *reinterpret_cast<const float4*>(cs), s1 = *reinterpret_cast<const float4*>(cs + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(cs + K), h1 = *reinterpret_cast<const float4*>(cs + K + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = fmaf((float)v[e], sc[e], sh[e]);
        if (relu) f = fmaxf(f, 0.f);
        o[e] = (bf16_t)f;
    }
    return o;
}

int odd_pitch(int bytes) {
    int q = bytes / 16 + 1;
    if (!(q & 1)) ++q;
    return q * 16;
}

// BNS: x -> the two operands from which both the weight gradient and the BatchNorm's backward sums follow.  With
// a = relu(gamma * xhat + beta) the input of the convolution and open = (a > 0):
//   S1[co][c] = sum_p dy[p][co] * open[p][c],   S2[co][c] = sum_p dy[p][co] * open[p][c] * xhat[p][c]
//   dW[co][c] = gamma[c] * S2 + beta[c] * S1;   sum_p g[p][c] = sum_co W[co][c] * S1,   sum_p g * xhat = sum_co W[co][c] * S2
// (g = the convolution's input gradient masked by the ReLU) -- the reduction pass of the BatchNorm backward, which would read
// the input gradient and x again, is not needed.  cs: scale, shift (the forward's folded constants: same mask), then
// invstd and -mean * invstd, K floats apart.
__device__ __forceinline__ void pw_bns8(const bf16x8_t& v, const float* cs, int K, bf16x8_t& open, bf16x8_t& openxh) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 sc = *reinterpret_cast<const float4*>(cs + 4 * h), sh = *reinterpret_cast<const float4*>(cs + K + 4 * h);
        const float4 xs = *reinterpret_cast<const float4*>(cs + 2 * K + 4 * h), xo = *reinterpret_cast<const float4*>(cs + 3 * K + 4 * h);
        const float a[4] = {sc.x, sc.y, sc.z, sc.w}, b[4] = {sh.x, sh.y, sh.z, sh.w};
        const float c[4] = {xs.x, xs.y, xs.z, xs.w}, d[4] = {xo.x, xo.y, xo.z, xo.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = (float)v[4 * h + e];
            const bool on = fmaf(x, a[e], b[e]) > 0.f;
            open[4 * h + e] = (bf16_t)(on ? 1.f : 0.f);
            openxh[4 * h + e] = (bf16_t)(on ? fmaf(x, c[e], d[e]) : 0.f);
        }
    }
}

// MB / KB = 16-channel blocks of dy / x that the instantiation covers (M <= 16 MB, K <= 16 KB; rows beyond M or K are zero)
template <int MB, int KB, bool AFF, bool BNS = false>
__global__ __launch_bounds__(NT) void pw_wgrad_kernel(PwWgParams p) {
    static_assert(!BNS || AFF, "the BatchNorm sums need the folded constants");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NX = BNS ? 2 : 1;                        // x tiles per wave: BNS stages `open` and `open * xhat`
    unsigned char* xt = lds + (size_t)wave * 32 * (NX * p.pitch_x + p.pitch_y);
    unsigned char* yt = xt + NX * 32 * p.pitch_x;
    const float* aff = reinterpret_cast<const float*>(lds + p.aff_off);
    if constexpr (AFF) {
        float* a = const_cast<float*>(aff);
        const int ka = p.x_aff2 ? p.K1 : p.K;
        for (int i = threadIdx.x; i < 2 * p.K; i += NT) {
            const int which = i >= p.K ? 1 : 0, c = i - which * p.K;
            a[i] = c < ka ? p.x_aff[which * ka + c] : p.x_aff2[which * (p.K - ka) + c - ka];
        }
        if constexpr (BNS) {
            for (int c = threadIdx.x; c < p.K; c += NT) {
                const float is = p.bn_invstd[c];
                a[2 * p.K + c] = is;
                a[3 * p.K + c] = -p.bn_mean[c] * is;
            }
        }
        __syncthreads();
    }

    // pieces of a chunk: id = i * 64 + lane -> (pixel, 8-channel group); the same for every chunk
    const int gx = p.K >> 3, gy = p.M >> 3;
    constexpr int XL = KB, YL = MB;                       // loads per lane: 32 * gx <= 64 KB pieces, 32 * gy <= 64 MB
    int xpix[XL], xsrc[XL], xlds[XL], ypix[YL], ysrc[YL], ylds[YL], xch[AFF ? XL : 1];
    bool xsecond[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int id = i * 64 + lane, pix = id / gx, grp = id - pix * gx;
        xpix[i] = id < 32 * gx ? pix : 1 << 20;
        xsecond[i] = p.x2 != nullptr && grp * 8 >= p.K1;
        xsrc[i] = xsecond[i] ? pix * p.ld_x2 + grp * 8 - p.K1 : pix * p.ld_x + grp * 8;
        xlds[i] = pix * p.pitch_x + grp * 16;
        if constexpr (AFF) xch[i] = grp * 8;
    }
#pragma unroll
    for (int i = 0; i < YL; ++i) {
        const int id = i * 64 + lane, pix = id / gy, grp = id - pix * gy;
        ypix[i] = id < 32 * gy ? pix : 1 << 20;
        ysrc[i] = pix * p.ld_dy + grp * 8;
        ylds[i] = pix * p.pitch_y + grp * 16;
    }
    // transposing reads: 16-lane group g covers pixels 8g + q (+4 for the second read), lane = 4q + c4 addresses channels
    // 4 c4 .. 4 c4 + 3 of a 16-channel block; afterwards lane j of the group holds channel j, pixels 8g .. 8g+7
    const int g = lane >> 4, li = lane & 15;
    const int tpix = 8 * g + (li >> 2), tch = 4 * (li & 3);
    const unsigned xr = (unsigned)(size_t)xt + tpix * p.pitch_x + tch * 2, yr = (unsigned)(size_t)yt + tpix * p.pitch_y + tch * 2;
    auto frag = [](unsigned a, int second) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(size_t)a);
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(size_t)(a + second));
        const s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, both);
    };

    f32x4_t acc[MB][KB], acc2[BNS ? MB : 1][BNS ? KB : 1];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (BNS) acc2[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }

    const long long chunks = (p.P + 31) / 32;
    const long long step = (long long)gridDim.x * (NT / 64);
    long long c = (long long)blockIdx.x * (NT / 64) + wave;
    bf16x8_t xv[XL], yv[YL];
    auto fetch = [&](long long chunk) {
        const long long first = chunk * 32;
        const int live = (int)(p.P - first < 32 ? p.P - first : 32);
        const bf16_t* xb = p.x + first * p.ld_x;
        const bf16_t* xb2 = p.x2 ? p.x2 + first * p.ld_x2 : xb;
        const bf16_t* yb = p.dy + first * p.ld_dy;
#pragma unroll
        for (int i = 0; i < XL; ++i)
            xv[i] = xpix[i] < live ? *reinterpret_cast<const bf16x8_t*>((xsecond[i] ? xb2 : xb) + xsrc[i]) : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < YL; ++i)
            yv[i] = ypix[i] < live ? *reinterpret_cast<const bf16x8_t*>(yb + ysrc[i]) : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    };
    // channel groups beyond K / M inside the last 16-channel block are never written: zero the tiles once
    for (int i = lane; i < 32 * (NX * p.pitch_x + p.pitch_y) / 16; i += 64)
        *reinterpret_cast<bf16x8_t*>(xt + i * 16) = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    if (c < chunks) fetch(c);
    for (; c < chunks; c += step) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // the previous chunk's operand reads are done
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < XL; ++i)
            if (xpix[i] < 32) {
                if constexpr (BNS) {
                    bf16x8_t o1, o2;
                    pw_bns8(xv[i], aff + xch[i], p.K, o1, o2);
                    *reinterpret_cast<bf16x8_t*>(xt + xlds[i]) = o1;
                    *reinterpret_cast<bf16x8_t*>(xt + 32 * p.pitch_x + xlds[i]) = o2;
                } else if constexpr (AFF) *reinterpret_cast<bf16x8_t*>(xt + xlds[i]) = pw_affine8(xv[i], aff + xch[i], p.K, p.x_relu);
                else *reinterpret_cast<bf16x8_t*>(xt + xlds[i]) = xv[i];
            }
#pragma unroll
        for (int i = 0; i < YL; ++i)
            if (ypix[i] < 32) *reinterpret_cast<bf16x8_t*>(yt + ylds[i]) = yv[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (c + step < chunks) fetch(c + step);
        bf16x8_t af[MB];
#pragma unroll
        for (int a = 0; a < MB; ++a) af[a] = frag(yr + a * 32, 4 * p.pitch_y);
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            const bf16x8_t bf = frag(xr + b * 32, 4 * p.pitch_x);
#pragma unroll
            for (int a = 0; a < MB; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf, acc[a][b], 0, 0, 0);
            if constexpr (BNS) {
                const bf16x8_t bf2 = frag(xr + 32 * p.pitch_x + b * 32, 4 * p.pitch_x);
#pragma unroll
                for (int a = 0; a < MB; ++a) acc2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf2, acc2[a][b], 0, 0, 0);
            }
        }
    }

    // ---- the four waves' blocks, summed in wave order, -> one slab per workgroup ------------------------------------------
    float* red = reinterpret_cast<float*>(lds);
    constexpr int NOUT = MB * 16 * KB * 16;
    float* slab = p.ws + (size_t)blockIdx.x * NOUT * NX;       // BNS: S1 then S2
#pragma unroll
    for (int which = 0; which < NX; ++which) {
        for (int w = 0; w < NT / 64; ++w) {
            __syncthreads();
            if (wave == w) {
#pragma unroll
                for (int a = 0; a < MB; ++a)
#pragma unroll
                    for (int b = 0; b < KB; ++b)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int idx = (a * 16 + 4 * g + i) * (KB * 16) + b * 16 + li;
                            float v = acc[a][b][i];
                            if constexpr (BNS) { if (which == 1) v = acc2[a][b][i]; }
                            red[idx] = (w == 0 ? 0.f : red[idx]) + v;
                        }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < NOUT; i += NT) slab[which * NOUT + i] = red[i];
    }
}

// raw[i] = sum_s ws[s][i] over the slabs (fixed order)
template <int SL>
__global__ __launch_bounds__(256) void pw_slab_sum_kernel(const float* __restrict__ ws, float* __restrict__ raw, long long per, int nslab) {
    long long idx;
    float t;
    if (ur_slab_sum<SL>(ws, per, nslab, idx, t)) raw[idx] = t;
}

// From S1, S2 ([rows][cols] each, see pw_bns8): the weight gradient dw[co][ci] (co < w_cout, ci < w_cin) and, per input
// channel, the BatchNorm's backward sums (sum g, sum g * xhat) = (dbeta, dgamma).  The weight enters rounded to bf16, as the
// backward-data kernel that produced g used it.
__global__ __launch_bounds__(256) void pw_wgrad_bn_finalize_kernel(const float* __restrict__ raw, int rows, int cols,
                                                                   const float* __restrict__ w, int w_cout, int w_cin,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ dw, float* __restrict__ sums,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= w_cin) return;
    const float* s1 = raw;
    const float* s2 = raw + (size_t)rows * cols;
    const float ga = gamma[c], be = beta[c];
    float g0 = 0.f, g1 = 0.f;
    for (int o = 0; o < w_cout; ++o) {
        const float a = s1[(size_t)o * cols + c], b = s2[(size_t)o * cols + c];
        if (dw) dw[(size_t)o * w_cin + c] = ga * b + be * a;
        const float wq = (float)(bf16_t)w[(size_t)o * w_cin + c];
        g0 = fmaf(wq, a, g0);
        g1 = fmaf(wq, b, g1);
    }
    sums[(size_t)c * 2] = g0; sums[(size_t)c * 2 + 1] = g1;
    if (dbeta) dbeta[c] = g0;
    if (dgamma) dgamma[c] = g1;
}

// dw[co][ci] = sum_s ws[s][co][ci] over the [16 MB][16 KB] slabs (co < w_cout, ci < w_cin are written)
template <int SL>
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int rows, int cols,
                                                              int w_cout, int w_cin, int nslab) {
    long long idx;
    float t;
    if (!ur_slab_sum<SL>(ws, (long long)rows * cols, nslab, idx, t)) return;
    const int co = (int)(idx / cols), ci = (int)(idx - (long long)co * cols);
    if (co < w_cout && ci < w_cin) dw[(size_t)co * w_cin + ci] = t;
}

bool shape_ok(const ur_conv_desc* d) {
    return d && d->dtype == UR_BF16 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->oh == d->h && d->ow == d->w &&
           d->cin % 8 == 0 && d->cout % 8 == 0 && d->in_ld % 8 == 0 && d->out_ld % 8 == 0;
}

// the instantiations: (cout blocks, cin blocks) of 16 channels
struct Cfg { int mb, kb; };
bool pick(const ur_conv_desc* d, Cfg& c) {
    // (a 64 x 128 instantiation was measured too: 91 us against the generic kernel's 61 on 128 -> 64 at 6 x 256 x 512)
    static const Cfg all[] = {{1, 2}, {2, 2}, {2, 5}, {2, 6}};
    for (const Cfg& k : all)
        if (d->cout <= 16 * k.mb && d->cin <= 16 * k.kb) { c = k; return true; }
    return false;
}

// LDS rows hold whole 16-channel blocks (the transposing reads fetch them whole; channels beyond K / M stay zero)
size_t lds_bytes(const Cfg& c, int nx = 1) {
    const size_t stage = (size_t)(NT / 64) * 32 * (nx * odd_pitch(c.kb * 32) + odd_pitch(c.mb * 32));
    const size_t red = (size_t)c.mb * 16 * c.kb * 16 * sizeof(float);
    return stage > red ? stage : red;
}

// workgroups = slabs; URESNET_PW_WGRAD_WGS overrides for A/B runs
int pw_wg_grid() {
    static const int g = [] { const char* e = getenv("URESNET_PW_WGRAD_WGS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
    return g;
}
constexpr int PW_WG_GRID_MAX = 4096;

}  // namespace

extern "C" {

int ur_conv1x1_pw_wgrad_supported(const ur_conv_desc* d) {
    Cfg c;
    // (worth it where the generic GEMM's 128 x 128 tile is mostly padding and the tensors are large enough to stream)
    return shape_ok(d) && pick(d, c) && (long long)d->n * d->h * d->w >= 1 << 16;
}

size_t ur_conv1x1_pw_wgrad_workspace(const ur_conv_desc* d) {
    Cfg c;
    if (!shape_ok(d) || !pick(d, c)) return 0;
    return (size_t)PW_WG_GRID_MAX * c.mb * 16 * c.kb * 16 * sizeof(float);
}

// ur_conv1x1_pw_wgrad_bn_pair: two slabs per workgroup and the summed pair behind them
size_t ur_conv1x1_pw_wgrad_bn_workspace(const ur_conv_desc* d) {
    Cfg c;
    if (!shape_ok(d) || !pick(d, c)) return 0;
    return (size_t)(PW_WG_GRID_MAX + 1) * 2 * c.mb * 16 * c.kb * 16 * sizeof(float);
}

}  // extern "C"

namespace {
struct PwWgSecond { const void* x2; int k1, ld_x2; const float* aff2; };
struct PwWgBn { const float *mean, *invstd, *gamma, *beta, *w; int w_cout, w_cin; float *sums, *dgamma, *dbeta; };
int pw_wgrad_launch(const void* x, const float* x_aff, int x_relu, const void* dy, float* dw, void* workspace, size_t ws_bytes,
                    const ur_conv_desc* d, ur_stream_t s, const PwWgSecond* second = nullptr, const PwWgBn* bn = nullptr) {
    Cfg c;
    UR_REQUIRE(shape_ok(d) && pick(d, c), "conv1x1_pw_wgrad: needs bf16, 1x1 stride 1, 8 | channels and pitches, cout <= 32, cin <= 96");
    UR_REQUIRE(x && dy && (dw || bn) && workspace, "conv1x1_pw_wgrad: null pointer");
    UR_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "conv1x1_pw_wgrad: tensors must be 16-byte aligned");
    UR_REQUIRE(ws_bytes >= (bn ? ur_conv1x1_pw_wgrad_bn_workspace(d) : ur_conv1x1_pw_wgrad_workspace(d)), "conv1x1_pw_wgrad: workspace too small");
    UR_REQUIRE(!bn || x_aff, "conv1x1_pw_wgrad: the BatchNorm sums need the folded constants");
    PwWgParams p{};
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.ws = (float*)workspace;
    p.P = (long long)d->n * d->h * d->w;
    p.K = d->cin; p.M = d->cout; p.ld_x = d->in_ld; p.ld_dy = d->out_ld;
    p.pitch_x = odd_pitch(c.kb * 32); p.pitch_y = odd_pitch(c.mb * 32);
    UR_REQUIRE(p.P * (p.ld_x > p.ld_dy ? p.ld_x : p.ld_dy) < (1LL << 31), "conv1x1_pw_wgrad: tensor exceeds 2^31 elements");
    if (second) {
        UR_REQUIRE(second->x2 && second->k1 > 0 && second->k1 < p.K && second->k1 % 8 == 0 && second->ld_x2 % 8 == 0 &&
                       second->ld_x2 >= p.K - second->k1 && ((uintptr_t)second->x2 % 16) == 0 && (!x_aff || second->aff2) &&
                       p.P * second->ld_x2 < (1LL << 31),
                   "conv1x1_pw_wgrad: the two-tensor input needs 8 | channel split and pitch, 16-byte alignment and constants for both");
        p.x2 = (const bf16_t*)second->x2; p.K1 = second->k1; p.ld_x2 = second->ld_x2; p.x_aff2 = second->aff2;
    }
    size_t lds = lds_bytes(c, bn ? 2 : 1);
    if (x_aff) {                                   // the input's constants sit behind the staging rows / sum buffer
        lds = (lds + 15) / 16 * 16;
        p.x_aff = x_aff; p.x_relu = x_relu; p.aff_off = (int)lds;
        lds += (size_t)(bn ? 4 : 2) * p.K * sizeof(float);
    }
    if (bn) { p.bn_mean = bn->mean; p.bn_invstd = bn->invstd; }
    const long long chunks = (p.P + 31) / 32;
    int grid = pw_wg_grid() < PW_WG_GRID_MAX ? pw_wg_grid() : PW_WG_GRID_MAX;
    if (chunks < (long long)grid * (NT / 64)) grid = (int)((chunks + NT / 64 - 1) / (NT / 64));
    hipStream_t st = (hipStream_t)s;
#define UR_PW_WG(MB, KB)                                                                                  \
    do {                                                                                                  \
        if (bn) hipLaunchKernelGGL((pw_wgrad_kernel<MB, KB, true, true>), dim3(grid), dim3(NT), lds, st, p); \
        else if (x_aff) hipLaunchKernelGGL((pw_wgrad_kernel<MB, KB, true>), dim3(grid), dim3(NT), lds, st, p); \
        else hipLaunchKernelGGL((pw_wgrad_kernel<MB, KB, false>), dim3(grid), dim3(NT), lds, st, p);      \
    } while (0)
    if (c.mb == 1) UR_PW_WG(1, 2);
    else if (c.mb == 2 && c.kb == 2) UR_PW_WG(2, 2);
    else if (c.mb == 2 && c.kb == 5) UR_PW_WG(2, 5);
    else UR_PW_WG(2, 6);
#undef UR_PW_WG
    if (int e = ur_check_launch("conv1x1_pw_wgrad")) return e;
    const int rows = c.mb * 16, cols = c.kb * 16;
    const long long per = (long long)rows * cols;
    if (bn) {
        float* raw = p.ws + (size_t)PW_WG_GRID_MAX * 2 * per;
        if (ur_slab_lanes(grid) == 32)
            hipLaunchKernelGGL(pw_slab_sum_kernel<32>, dim3(ur_slab_grid(2 * per, 32)), dim3(256), 0, st, p.ws, raw, 2 * per, grid);
        else
            hipLaunchKernelGGL(pw_slab_sum_kernel<8>, dim3(ur_slab_grid(2 * per, 8)), dim3(256), 0, st, p.ws, raw, 2 * per, grid);
        if (int e = ur_check_launch("conv1x1_pw_wgrad_slab_sum")) return e;
        hipLaunchKernelGGL(pw_wgrad_bn_finalize_kernel, dim3(ur_cdiv(bn->w_cin, 256)), dim3(256), 0, st, raw, rows, cols, bn->w,
                           bn->w_cout, bn->w_cin, bn->gamma, bn->beta, dw, bn->sums, bn->dgamma, bn->dbeta);
        return ur_check_launch("conv1x1_pw_wgrad_bn_finalize");
    }
    if (ur_slab_lanes(grid) == 32)
        hipLaunchKernelGGL(pw_wgrad_reduce_kernel<32>, dim3(ur_slab_grid(per, 32)), dim3(256), 0, st, p.ws, dw, rows, cols, d->cout,
                           d->cin, grid);
    else
        hipLaunchKernelGGL(pw_wgrad_reduce_kernel<8>, dim3(ur_slab_grid(per, 8)), dim3(256), 0, st, p.ws, dw, rows, cols, d->cout,
                           d->cin, grid);
    return ur_check_launch("conv1x1_pw_wgrad_reduce");
}
}  // namespace

extern "C" {

int ur_conv1x1_pw_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes, const ur_conv_desc* d,
                        ur_stream_t s) {
    return pw_wgrad_launch(x, nullptr, 0, dy, dw, workspace, ws_bytes, d, s);
}

int ur_conv1x1_pw_wgrad_affine(const void* x, const float* x_aff, int x_relu, const void* dy, float* dw, void* workspace,
                               size_t ws_bytes, const ur_conv_desc* d, ur_stream_t s) {
    UR_REQUIRE(x_aff, "conv1x1_pw_wgrad_affine: null constants");
    return pw_wgrad_launch(x, x_aff, x_relu, dy, dw, workspace, ws_bytes, d, s);
}

int ur_conv1x1_pw_wgrad_affine_pair(const void* xa, int ca, int ld_a, const void* xb, int cb, int ld_b, const float* aff_a,
                                    const float* aff_b, int x_relu, const void* dy, float* dw, void* workspace, size_t ws_bytes,
                                    const ur_conv_desc* d, ur_stream_t s) {
    UR_REQUIRE(xa && xb && aff_a && aff_b && d, "conv1x1_pw_wgrad_affine_pair: null argument");
    UR_REQUIRE(ca + cb == d->cin, "conv1x1_pw_wgrad_affine_pair: %d + %d channels against a descriptor of %d", ca, cb, d->cin);
    ur_conv_desc da = *d;
    da.in_ld = ld_a ? ld_a : ca;
    UR_REQUIRE(da.in_ld % 8 == 0 && da.in_ld >= ca, "conv1x1_pw_wgrad_affine_pair: bad pitch of the first tensor");
    const PwWgSecond second{xb, ca, ld_b ? ld_b : cb, aff_b};
    return pw_wgrad_launch(xa, aff_a, x_relu, dy, dw, workspace, ws_bytes, &da, s, &second);
}


int ur_conv1x1_pw_wgrad_bn(const void* x, const float* x_aff, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, const void* dy, const float* w, int w_cout, float* dw, float* sums, float* dgamma,
                           float* dbeta, void* workspace, size_t ws_bytes, const ur_conv_desc* d, ur_stream_t s) {
    UR_REQUIRE(x && x_aff && mean && invstd && gamma && beta && w && sums && d, "conv1x1_pw_wgrad_bn: null argument");
    UR_REQUIRE(w_cout > 0 && w_cout <= d->cout, "conv1x1_pw_wgrad_bn: %d weight rows against a descriptor of %d", w_cout, d->cout);
    const PwWgBn bn{mean, invstd, gamma, beta, w, w_cout, d->cin, sums, dgamma, dbeta};
    return pw_wgrad_launch(x, x_aff, 1, dy, dw, workspace, ws_bytes, d, s, nullptr, &bn);      // (dw may be null)
}

int ur_conv1x1_pw_wgrad_bn_pair(const void* xa, int ca, int ld_a, const void* xb, int cb, int ld_b, const float* aff_a,
                                const float* aff_b, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                const void* dy, const float* w, int w_cout, float* dw, float* sums, float* dgamma, float* dbeta,
                                void* workspace, size_t ws_bytes, const ur_conv_desc* d, ur_stream_t s) {
    UR_REQUIRE(xa && xb && aff_a && aff_b && mean && invstd && gamma && beta && w && sums && d, "conv1x1_pw_wgrad_bn_pair: null argument");
    UR_REQUIRE(ca + cb == d->cin, "conv1x1_pw_wgrad_bn_pair: %d + %d channels against a descriptor of %d", ca, cb, d->cin);
    UR_REQUIRE(w_cout > 0 && w_cout <= d->cout, "conv1x1_pw_wgrad_bn_pair: %d weight rows against a descriptor of %d", w_cout, d->cout);
    ur_conv_desc da = *d;
    da.in_ld = ld_a ? ld_a : ca;
    UR_REQUIRE(da.in_ld % 8 == 0 && da.in_ld >= ca, "conv1x1_pw_wgrad_bn_pair: bad pitch of the first tensor");
    const PwWgSecond second{xb, ca, ld_b ? ld_b : cb, aff_b};
    const PwWgBn bn{mean, invstd, gamma, beta, w, w_cout, d->cin, sums, dgamma, dbeta};
    return pw_wgrad_launch(xa, aff_a, 1, dy, dw, workspace, ws_bytes, &da, s, &second, &bn);      // (dw may be null)
}

}  // extern "C"
