// Backward-weight convolution as an MFMA GEMM whose reduction runs over pixels.
//
//   dW[o][(tap,i)] = sum_pix dV[pix][o] * U[pix (+) tap][i]
// rows  = channels of the output-gradient tensor dV (tile 128),
// cols  = (tap, channel of the input tensor U), channel fastest (tile 128),
// K     = pixels of dV, processed in slices of BK pixels; the pixel range is split over
//         gridDim.y workgroups whose fp32 partial tiles go to a workspace slab each, and a
//         second kernel sums the slabs in a fixed order (bitwise reproducible, no atomics)
//         while transposing (tap,i) -> PyTorch's [o][i][ky][kx].
// Both operands sit in LDS as [pixel][channel] (the NHWC order they are loaded in).  The
// MFMA wants 8 consecutive k (= pixels) per lane, so bf16 fragments are fetched with the
// transposing LDS read ds_read_b64_tr_b16; fp32 uses v_mfma_f32_32x32x2_f32, whose one-k-per
// -lane operands are plain row reads of that layout.
#include "ur_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int TM = 128, TN = 128, NTHREADS = 256;

struct WgradParams {
    const void* u;       // input tensor   (N, H, W, I), pitch u_ld
    const void* dv;      // output gradient (N, OH, OW, O), pitch dv_ld
    float* ws;           // [nsplit][O][Kred]
    int N, H, W, I, OH, OW, O;
    int kh, kw, stride, pad;
    int Kred;            // kh*kw*I
    int P;               // N*OH*OW
    int u_ld, dv_ld;
    int stages_per_split;
    int ntiles;
    int xcd_order;       // splits are a multiple of 8: see the kernel's id decoding
};

template <typename T> struct WgCfg;
template <> struct WgCfg<bf16_t> { static constexpr int BK = 64, PITCH = 160, VW = 8; };   // 256 B row + 64 B pad
template <> struct WgCfg<float> { static constexpr int BK = 32, PITCH = 132, VW = 4; };    // 512 B row + 16 B pad

template <typename T, int V> struct Chunk;          // V consecutive channels held in registers
template <> struct Chunk<bf16_t, 8> {
    bf16x8_t v;
    __device__ __forceinline__ void zero() { v = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; }
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8_t*>(p); }
    __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<bf16x8_t*>(p) = v; }
};
template <> struct Chunk<float, 4> {      // fp32 moves 16 bytes per chunk as well (4 channels)
    float4 a;
    __device__ __forceinline__ void zero() { a = make_float4(0, 0, 0, 0); }
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = a; }
};
template <typename T> struct Chunk<T, 1> {
    T v;
    __device__ __forceinline__ void zero() { v = ur_from_float<T>(0.f); }
    __device__ __forceinline__ void load(const T* p) { v = *p; }
    __device__ __forceinline__ void store(T* p) const { *p = v; }
};

// TMT = rows (channels of dV) of the output tile: 128 (2 x 2 waves of 64 x 64) or 64 (1 x 4 waves of 64 x 32) for layers
// with at most 64 such channels, which would leave half of every 128-row tile's MFMAs multiplying zeros.
template <typename T, int VA, int VB, int TMT>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_kernel(WgradParams p) {
    constexpr int WN = TMT == 128 ? 2 : 4, WCOLS = TN / WN, NBL = WCOLS / 32;      // waves along n, columns and 32-blocks per wave
    constexpr int BK = WgCfg<T>::BK, PITCH = WgCfg<T>::PITCH;
    __shared__ __attribute__((aligned(16))) T As[BK * PITCH];
    __shared__ __attribute__((aligned(16))) T Bs[BK * PITCH];

    // Workgroup ids congruent mod 8 run on one XCD (one L2 each).  With xcd_order all output tiles of a pixel range (split)
    // go to one XCD, next to each other in dispatch order, so the operand rows that those tiles share are fetched from memory
    // once and not once per tile (a transposed convolution 256 -> 128 has 8 tiles: 800 MB of operand reads for 300 MB).
    int tile = blockIdx.x, split = blockIdx.y;
    if (p.xcd_order) {
        const int id = blockIdx.x + gridDim.x * blockIdx.y, j = id >> 3;
        tile = j % (int)gridDim.x;
        split = (j / (int)gridDim.x) * 8 + (id & 7);
    }
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int o0 = mt * TMT, n0 = nt * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const T* __restrict__ U = static_cast<const T*>(p.u);
    const T* __restrict__ DV = static_cast<const T*>(p.dv);

    // loader geometry: chunks of V channels; (TMT/V) chunk-columns per pixel row
    constexpr int ACOLS = TMT / VA, AROWS_PER_PASS = NTHREADS / ACOLS, APASS = BK / AROWS_PER_PASS;
    constexpr int BCOLS = TN / VB, BROWS_PER_PASS = NTHREADS / BCOLS, BPASS = BK / BROWS_PER_PASS;
    const int a_col = (tid % ACOLS) * VA, a_row = tid / ACOLS;
    const int b_col = (tid % BCOLS) * VB, b_row = tid / BCOLS;
    const bool a_ok = (o0 + a_col) < p.O;              // O % VA == 0 whenever VA > 1
    // this thread's B column is a fixed (tap, channel)
    const int bn = n0 + b_col;
    const bool b_ok = bn < p.Kred;
    int b_ky = 0, b_kx = 0, b_ci = 0;
    if (b_ok) { int tap = bn / p.I; b_ci = bn - tap * p.I; b_ky = tap / p.kw; b_kx = tap - b_ky * p.kw; }

    Chunk<T, VA> pa[APASS];
    Chunk<T, VB> pb[BPASS];
    const int stage0 = split * p.stages_per_split;
    int nstages = (p.P + BK - 1) / BK - stage0;
    if (nstages > p.stages_per_split) nstages = p.stages_per_split;

    // The gather runs once per 16 MFMAs, so the common case avoids integer division and 32-bit multiplies (quarter-rate
    // VALU): when a row of dV is at least BK pixels wide and element offsets fit 32 bits, every loader slot keeps its
    // pixel coordinates and advances them by BK pixels per stage with at most one carry per dimension.
    const bool fast = p.OW >= BK && (long long)p.N * p.H * p.W * p.u_ld < (1LL << 31) && (long long)p.P * p.dv_ld < (1LL << 31) &&
                      (long long)p.W * p.u_ld < (1LL << 24) && p.H < (1 << 22);
    int fa_off[APASS], fb_img[BPASS], fb_oy[BPASS], fb_ox[BPASS], fa_pix[APASS], fb_pix[BPASS];
    const int u_row = p.W * p.u_ld, u_img = p.H * p.W * p.u_ld;
    if (fast) {
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            fa_pix[i] = stage0 * BK + a_row + i * AROWS_PER_PASS;
            fa_off[i] = fa_pix[i] * p.dv_ld + o0 + a_col;
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const int pix = stage0 * BK + b_row + i * BROWS_PER_PASS;
            fb_pix[i] = pix;
            const int img = pix / (p.OH * p.OW), rem = pix - img * (p.OH * p.OW);
            fb_img[i] = img * u_img; fb_oy[i] = rem / p.OW; fb_ox[i] = rem - fb_oy[i] * p.OW;
        }
    }
    auto gather = [&](int st) {
        if (fast) {
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                if (a_ok && fa_pix[i] < p.P) pa[i].load(DV + fa_off[i]);
                else pa[i].zero();
                fa_pix[i] += BK; fa_off[i] += BK * p.dv_ld;
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const int iy = fb_oy[i] * p.stride + b_ky - p.pad, ix = fb_ox[i] * p.stride + b_kx - p.pad;
                const bool ok = b_ok && fb_pix[i] < p.P && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                if (ok) pb[i].load(U + (fb_img[i] + (int)__umul24(iy, u_row) + (int)__umul24(ix, p.u_ld) + b_ci));
                else pb[i].zero();
                fb_pix[i] += BK; fb_ox[i] += BK;
                if (fb_ox[i] >= p.OW) { fb_ox[i] -= p.OW; if (++fb_oy[i] == p.OH) { fb_oy[i] = 0; fb_img[i] += u_img; } }
            }
            return;
        }
        const int pix0 = (stage0 + st) * BK;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            int pix = pix0 + a_row + i * AROWS_PER_PASS;
            if (a_ok && pix < p.P) pa[i].load(DV + (size_t)pix * p.dv_ld + o0 + a_col);
            else pa[i].zero();
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            int pix = pix0 + b_row + i * BROWS_PER_PASS;
            int pc = pix < p.P ? pix : 0;
            int img = pc / (p.OH * p.OW);
            int rem = pc - img * (p.OH * p.OW);
            int oy = rem / p.OW, ox = rem - oy * p.OW;
            int iy = oy * p.stride + b_ky - p.pad, ix = ox * p.stride + b_kx - p.pad;
            bool ok = b_ok && pix < p.P && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            if (ok) pb[i].load(U + ((size_t)(img * p.H + iy) * p.W + ix) * p.u_ld + b_ci);
            else pb[i].zero();
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < APASS; ++i) pa[i].store(&As[(a_row + i * AROWS_PER_PASS) * PITCH + a_col]);
#pragma unroll
        for (int i = 0; i < BPASS; ++i) pb[i].store(&Bs[(b_row + i * BROWS_PER_PASS) * PITCH + b_col]);
    };

    f32x16_t acc[2][NBL];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (nstages > 0) gather(0);
    for (int st = 0; st < nstages; ++st) {
        __syncthreads();
        commit();
        __syncthreads();
        if (st + 1 < nstages) gather(st + 1);
        if constexpr (std::is_same<T, bf16_t>::value) {
            typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
            // transposing read: 16-lane group g covers pixels [8*(g>>1) .. +3] x channels [16*(g&1) .. +15]
            const int g = lane >> 4, li = lane & 15;
            const int t_pix = 8 * (g >> 1) + (li >> 2), t_ch = 16 * (g & 1) + 4 * (li & 3);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8_t af[2], bfr[NBL];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const T* q = &As[(ks * 16 + t_pix) * PITCH + wm * 64 + a * 32 + t_ch];
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)q);
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q + 4 * PITCH));
                    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
                    s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    af[a] = __builtin_bit_cast(bf16x8_t, both);
                }
#pragma unroll
                for (int b = 0; b < NBL; ++b) {
                    const T* q = &Bs[(ks * 16 + t_pix) * PITCH + wn * WCOLS + b * 32 + t_ch];
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)q);
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q + 4 * PITCH));
                    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
                    s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bfr[b] = __builtin_bit_cast(bf16x8_t, both);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < NBL; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
            }
        } else {
            const int lrow = lane & 31, lhalf = lane >> 5;
#pragma unroll 4
            for (int kk = 0; kk < BK; kk += 2) {
                float af[2], bfr[NBL];
#pragma unroll
                for (int a = 0; a < 2; ++a) af[a] = As[(kk + lhalf) * PITCH + wm * 64 + a * 32 + lrow];
#pragma unroll
                for (int b = 0; b < NBL; ++b) bfr[b] = Bs[(kk + lhalf) * PITCH + wn * WCOLS + b * 32 + lrow];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < NBL; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bfr[b], acc[a][b], 0, 0, 0);
            }
        }
    }

    // partial tile -> this split's slab
    const int lrow = lane & 31, lhalf = lane >> 5;
    float* slab = p.ws + (size_t)split * p.O * p.Kred;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
            const int n = n0 + wn * WCOLS + b * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                if (o < p.O && n < p.Kred) slab[(size_t)o * p.Kred + n] = acc[a][b][r];
            }
        }
}

// dw[o][i][tap] = sum_s ws[s][o][tap*I + i].  32 outputs x 8 split lanes per workgroup: the slab reads of a
// wave are 128-byte rows and the split loop is 8-way parallel (long split lists come with small outputs).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int O, int I,
                                                           int taps, int nsplit) {
    __shared__ float red[8][32];
    const long long per = (long long)O * I * taps;
    const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long long idx = blockIdx.x * 32LL + ol;
    float s = 0.f;
    if (idx < per)
        for (int k = sl; k < nsplit; k += 8) s += ws[(long long)k * per + idx];
    red[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && idx < per) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][ol];
        int o = (int)(idx / ((long long)I * taps));
        int rem = (int)(idx - (long long)o * I * taps);
        int tap = rem / I, i = rem - tap * I;           // slab order: (tap, i) contiguous
        dw[((long long)o * I + i) * taps + tap] = t;
    }
}

// the same sum with 16-byte slab reads, for slabs whose length is a multiple of 4
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ ws, float* __restrict__ dw, int O, int I,
                                                            int taps, int nsplit) {
    long long idx;
    float t;
    if (!ur_slab_sum<SL>(ws, (long long)O * I * taps, nsplit, idx, t)) return;
    const int o = (int)(idx / ((long long)I * taps));
    const int rem = (int)(idx - (long long)o * I * taps);
    const int tap = rem / I, i = rem - tap * I;
    dw[((long long)o * I + i) * taps + tap] = t;
}

struct WgradPlan { int mtiles, ntiles, nsplit, stages_per_split, tm; };

template <typename T>
WgradPlan plan_for(int O, int Kred, long long P) {
    constexpr int BK = WgCfg<T>::BK;
    WgradPlan pl;
    static const bool tm64 = [] { const char* e = getenv("URESNET_WGRAD_TM64"); return !e || e[0] != '0'; }();   // 0: A/B runs
    pl.tm = (O <= 64 && tm64) ? 64 : TM;
    pl.mtiles = ur_cdiv(O, pl.tm);
    pl.ntiles = ur_cdiv(Kred, TN);
    long long stages = (P + BK - 1) / BK;
    long long tiles = (long long)pl.mtiles * pl.ntiles;
    // One resident round: the vector-loader instantiations hold three workgroups per CU (156 registers, 40 KiB of
    // LDS), so 768 equal shares finish together; 1024 left a quarter-filled second round (1.54 -> 1.32 ms per step
    // over the network's eleven layers of this kind).  URESNET_WGRAD_WGS overrides for A/B runs.
    static const int total = [] { const char* e = getenv("URESNET_WGRAD_WGS"); return e && atoi(e) > 0 ? atoi(e) : 768; }();
    long long want = (total + tiles - 1) / tiles;
    if (want > stages) want = stages;
    if (want < 1) want = 1;
    pl.stages_per_split = (int)((stages + want - 1) / want);
    pl.nsplit = (int)((stages + pl.stages_per_split - 1) / pl.stages_per_split);
    return pl;
}

WgradPlan plan_dtype(int dtype, int O, int Kred, long long P) {
    return dtype == UR_BF16 ? plan_for<bf16_t>(O, Kred, P) : plan_for<float>(O, Kred, P);
}

template <typename T>
int launch_wgrad(WgradParams& p, float* dw, hipStream_t st) {
    WgradPlan pl = plan_for<T>(p.O, p.Kred, p.P);
    p.stages_per_split = pl.stages_per_split;
    p.ntiles = pl.ntiles;
    static const bool xcd_ok = [] { const char* e = getenv("URESNET_WGRAD_XCD"); return !e || e[0] != '0'; }();
    p.xcd_order = xcd_ok && pl.nsplit % 8 == 0 && pl.mtiles * pl.ntiles > 1;
    dim3 grid(pl.mtiles * pl.ntiles, pl.nsplit);
    constexpr int VW = WgCfg<T>::VW;
    const bool va = (p.O % VW == 0) && (p.dv_ld % VW == 0) && ((uintptr_t)p.dv % 16 == 0);
    const bool vb = (p.I % VW == 0) && (p.u_ld % VW == 0) && ((uintptr_t)p.u % 16 == 0);
#define UR_WGRAD_LAUNCH(A_, B_)                                                                                         \
    do {                                                                                                                \
        if (pl.tm == 64) hipLaunchKernelGGL((conv_wgrad_kernel<T, A_, B_, 64>), grid, dim3(NTHREADS), 0, st, p);        \
        else hipLaunchKernelGGL((conv_wgrad_kernel<T, A_, B_, 128>), grid, dim3(NTHREADS), 0, st, p);                   \
    } while (0)
    if (va && vb) UR_WGRAD_LAUNCH(VW, VW);
    else if (va) UR_WGRAD_LAUNCH(VW, 1);
    else if (vb) UR_WGRAD_LAUNCH(1, VW);
    else UR_WGRAD_LAUNCH(1, 1);
#undef UR_WGRAD_LAUNCH
    if (int e = ur_check_launch("conv_wgrad")) return e;
    const long long per = (long long)p.O * p.Kred;
    if (per % 4 != 0)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per + 31) / 32)), dim3(256), 0, st, p.ws, dw, p.O, p.I,
                           p.kh * p.kw, pl.nsplit);
    else if (ur_slab_lanes(pl.nsplit) == 32)
        hipLaunchKernelGGL(wgrad_reduce4_kernel<32>, dim3(ur_slab_grid(per, 32)), dim3(256), 0, st, p.ws, dw, p.O, p.I,
                           p.kh * p.kw, pl.nsplit);
    else
        hipLaunchKernelGGL(wgrad_reduce4_kernel<8>, dim3(ur_slab_grid(per, 8)), dim3(256), 0, st, p.ws, dw, p.O, p.I,
                           p.kh * p.kw, pl.nsplit);
    return ur_check_launch("wgrad_reduce");
}

int check_common(const ur_conv_desc* d, const void* a, const void* b, const float* dw, const void* ws) {
    UR_REQUIRE(d != nullptr, "wgrad: null descriptor");
    UR_REQUIRE(d->dtype == UR_F32 || d->dtype == UR_BF16, "wgrad: bad dtype %d", d->dtype);
    UR_REQUIRE(a && b && dw && ws, "wgrad: null pointer");
    UR_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0 && d->oh > 0 && d->ow > 0,
               "wgrad: non-positive dimension");
    UR_REQUIRE(d->in_ld >= d->cin && d->out_ld >= d->cout, "wgrad: pitch smaller than channel count");
    UR_REQUIRE((long long)d->n * d->h * d->w < (1LL << 31) && (long long)d->n * d->oh * d->ow < (1LL << 31),
               "wgrad: more than 2^31 pixels");
    return UR_OK;
}

}  // namespace

extern "C" {

size_t ur_conv2d_wgrad_workspace(const ur_conv_desc* d) {
    if (!d) return 0;
    const int Kred = d->kh * d->kw * d->cin;
    WgradPlan pl = plan_dtype(d->dtype, d->cout, Kred, (long long)d->n * d->oh * d->ow);
    return (size_t)pl.nsplit * d->cout * Kred * sizeof(float);
}

int ur_conv2d_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes, const ur_conv_desc* d,
                    ur_stream_t s) {
    if (int e = check_common(d, x, dy, dw, workspace)) return e;
    UR_REQUIRE(ws_bytes >= ur_conv2d_wgrad_workspace(d), "conv2d_wgrad: workspace too small (%zu < %zu)", ws_bytes,
               ur_conv2d_wgrad_workspace(d));
    WgradParams p{};
    p.u = x; p.dv = dy; p.ws = (float*)workspace;
    p.N = d->n; p.H = d->h; p.W = d->w; p.I = d->cin; p.OH = d->oh; p.OW = d->ow; p.O = d->cout;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.Kred = d->kh * d->kw * d->cin; p.P = d->n * d->oh * d->ow;
    p.u_ld = d->in_ld; p.dv_ld = d->out_ld;
    UR_DISPATCH_DTYPE(d->dtype, return launch_wgrad<T>(p, dw, (hipStream_t)s));
}

// ConvTranspose2d(k=2,s=2): dW[ci][co][dy][dx] = sum x[n,y,x,ci] * dout[n,2y+dy,2x+dx,co]
// = backward-weight of the 2x2 stride-2 convolution that maps dout-shaped tensors to x-shaped ones.
size_t ur_convT2x2_wgrad_workspace(const ur_conv_desc* d) {
    if (!d) return 0;
    const int Kred = 4 * d->cout;
    WgradPlan pl = plan_dtype(d->dtype, d->cin, Kred, (long long)d->n * d->h * d->w);
    return (size_t)pl.nsplit * d->cin * Kred * sizeof(float);
}

int ur_convT2x2_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes,
                      const ur_conv_desc* d, ur_stream_t s) {
    if (int e = check_common(d, x, dy, dw, workspace)) return e;
    UR_REQUIRE(d->kh == 2 && d->kw == 2 && d->stride == 2 && d->pad == 0 && d->oh == 2 * d->h && d->ow == 2 * d->w,
               "convT2x2_wgrad: geometry must be 2x2 stride 2 pad 0 with a 2x output");
    UR_REQUIRE(ws_bytes >= ur_convT2x2_wgrad_workspace(d), "convT2x2_wgrad: workspace too small");
    WgradParams p{};
    p.u = dy; p.dv = x; p.ws = (float*)workspace;
    p.N = d->n; p.H = d->oh; p.W = d->ow; p.I = d->cout; p.OH = d->h; p.OW = d->w; p.O = d->cin;
    p.kh = 2; p.kw = 2; p.stride = 2; p.pad = 0;
    p.Kred = 4 * d->cout; p.P = d->n * d->h * d->w;
    p.u_ld = d->out_ld; p.dv_ld = d->in_ld;
    UR_DISPATCH_DTYPE(d->dtype, return launch_wgrad<T>(p, dw, (hipStream_t)s));
}

}  // extern "C"
