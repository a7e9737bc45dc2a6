// Implicit-GEMM convolution on the gfx950 matrix cores (forward and backward-data).
//
// GEMM view:  D[m][n] = sum_k A[m][k] * B[n][k]
//   rows  m = (image, y, x) of the tensor being produced (forward: output pixels,
//             backward-data: input pixels), flattened,
//   cols  n = channel of the tensor being produced,
//   k       = (ky, kx, c) with the channel of the gathered tensor fastest, so 8 consecutive
//             k are 8 consecutive NHWC channels of one tap (a single 16-byte bf16 load).
// A is never materialised: each workgroup gathers its 256 x 32 slice straight from the
// NHWC source into LDS (zero for padding / stride holes).  B is the weight matrix, pre-packed
// to [n][Kpad] by ur_conv2d_pack_weight.  bf16 storage uses v_mfma_f32_32x32x16_bf16, fp32
// storage uses v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain).
//
// Workgroup = 4 waves; wave w owns rows [64w, 64w+64) x all BN columns of the tile
// (2 x BN/32 accumulator tiles of 32x32).  One LDS buffer, register prefetch of the next
// k-slice while the current one is multiplied.
#include "ur_common.h"
#include <stdlib.h>
#include <type_traits>

// tensors streamed once per launch go through non-temporal accesses (0.2 % of the training step); -DUR_IGEMM_NT=0: ordinary
#ifndef UR_IGEMM_NT
#define UR_IGEMM_NT 1
#endif
namespace {

constexpr int BM = 256;
constexpr int BK = 32;
constexpr int NTHREADS = 256;

struct IgemmParams {
    const void* src;
    const void* wgt;
    const float* bias;
    void* dst;
    float* stats;
    int N, H, W, C;      // gathered tensor: images, height, width, channels per tap
    int OH, OW;          // row-space height / width
    int Ncols;           // columns = channels produced
    int kh, kw, stride, pad;
    int Kred, Kpad;
    int M;
    int src_ld, dst_ld;
    int mode;            // 0: forward gather, 1: backward-data gather
    int scatter;         // 1: ConvTranspose 2x2 s2 scatter epilogue (n = (dy,dx,co))
    int cout_t;          // scatter: channels of the transposed conv's output
    int mtiles, ntiles;
    int vec_store;       // 1: 16-byte stores are legal (pitch, column count, base address and scatter width all aligned)
};

template <typename T> struct LdsPitch;
template <> struct LdsPitch<bf16_t> { static constexpr int elems = 40; };   // 64 B row + 16 B pad
template <> struct LdsPitch<float> { static constexpr int elems = 36; };    // 128 B row + 16 B pad

template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
    bf16x8_t v;
    __device__ __forceinline__ void zero() { v = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; }
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8_t*>(p); }
    __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<bf16x8_t*>(p) = v; }
};
template <> struct Vec8<float> {
    float4 a, b;
    __device__ __forceinline__ void zero() { a = make_float4(0, 0, 0, 0); b = a; }
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = a;
        *reinterpret_cast<float4*>(p + 4) = b;
    }
};

// Source pixel of row coordinate (oy, ox) for tap (ky, kx); returns false for padding / holes.
__device__ __forceinline__ bool tap_source(const IgemmParams& p, int oy, int ox, int ky, int kx, int& iy, int& ix) {
    if (p.mode == 0) {
        iy = oy * p.stride + ky - p.pad;
        ix = ox * p.stride + kx - p.pad;
        return (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    }
    int ty = oy + p.pad - ky, tx = ox + p.pad - kx;
    if (ty < 0 || tx < 0) return false;
    if (p.stride == 1) {
        iy = ty; ix = tx;
    } else {
        iy = ty / p.stride; ix = tx / p.stride;
        if (iy * p.stride != ty || ix * p.stride != tx) return false;
    }
    return iy < p.H && ix < p.W;
}

// Three workgroups per CU for the 32/64-column tiles (168 registers): the layers that come here have 1-8 reduction
// slices, so what hides the gather latency is the number of resident workgroups (measured: +18 % over two).
template <typename T, int BN, bool VEC8, bool FAST>
__global__ __launch_bounds__(NTHREADS, (BN <= 64 && sizeof(T) == 2) ? 3 : (sizeof(T) == 2 ? 2 : 1)) void conv_igemm_kernel(IgemmParams p) {
    constexpr int PITCH = LdsPitch<T>::elems;
    constexpr int NT = BN / 32;
    __shared__ __attribute__((aligned(16))) T As[BM * PITCH];
    __shared__ __attribute__((aligned(16))) T Bs[BN * PITCH];
    __shared__ float red[4][BN][2];

    // XCD-aware tile order: workgroups id, id+8, id+16, ... share an XCD (and its L2); give
    // them the column tiles of one row tile so the gathered A slice is fetched from HBM once.
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int mt = (slot / p.ntiles) * 8 + xcd;
    const int nt = slot % p.ntiles;
    if (mt >= p.mtiles) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* __restrict__ src = static_cast<const T*>(p.src);
    const T* __restrict__ wgt = static_cast<const T*>(p.wgt);

    // ---- per-thread gather bookkeeping --------------------------------------------------
    // VEC8: thread owns chunk (tid&3) of rows (tid>>2) + 64*i.   scalar: thread owns row tid.
    constexpr int AROWS = VEC8 ? 4 : 1;
    int r_img[AROWS], r_y[AROWS], r_x[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        int row = VEC8 ? ((tid >> 2) + 64 * i) : tid;
        int m = m0 + row;
        if (m < p.M) {
            int img = m / (p.OH * p.OW);
            int rem = m - img * (p.OH * p.OW);
            r_img[i] = img; r_y[i] = rem / p.OW; r_x[i] = rem - r_y[i] * p.OW;
        } else {
            r_img[i] = -1; r_y[i] = 0; r_x[i] = 0;
        }
    }
    // FAST (vector gather, 32-bit element offsets, stride 1 or 2): the gather runs once per 8-32 MFMAs, so it must not
    // divide or use 32-bit multiplies (quarter-rate VALU).  Per row: the coordinates and element offset of tap (0, 0);
    // per thread: the tap (ky, kx) and channel of its 8-wide k chunk, advanced by 32 k per slice with carries.
    int f_y0[AROWS], f_x0[AROWS], f_off0[AROWS];
    int f_k = (tid & 3) * 8, f_ky = 0, f_kx = 0, f_ci = 0;
    unsigned f_ok = 0;
    const int f_row = p.W * p.src_ld;                    // < 2^24 (checked on the host)
    if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            if (p.mode == 0) { f_y0[i] = r_y[i] * p.stride - p.pad; f_x0[i] = r_x[i] * p.stride - p.pad; }
            else { f_y0[i] = r_y[i] + p.pad; f_x0[i] = r_x[i] + p.pad; }
            f_off0[i] = r_img[i] * p.H * p.W * p.src_ld;     // image base; the tap term is added per slice
        }
        const int tap = f_k / p.C;
        f_ci = f_k - tap * p.C;
        f_ky = tap / p.kw; f_kx = tap - f_ky * p.kw;
    }
    constexpr int BCH = (BN * 4 + NTHREADS - 1) / NTHREADS;   // 16-byte-equivalent B chunks per thread

    Vec8<T> pa[AROWS];
    float pa_s[VEC8 ? 1 : BK];
    Vec8<T> pb[BCH];

    auto gather = [&](int kt) {
        if constexpr (VEC8 && FAST) {
            const bool kin = f_k < p.Kred;
#pragma unroll
            for (int i = 0; i < AROWS; ++i) {
                int iy, ix;
                bool ok;
                if (p.mode == 0) {
                    iy = f_y0[i] + f_ky; ix = f_x0[i] + f_kx;
                    ok = true;
                } else {
                    const int ty = f_y0[i] - f_ky, tx = f_x0[i] - f_kx;
                    if (p.stride == 1) { iy = ty; ix = tx; ok = true; }
                    else { iy = ty >> 1; ix = tx >> 1; ok = ((ty | tx) & 1) == 0; }      // stride 2
                }
                ok = ok && kin && r_img[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const int off = f_off0[i] + (int)__umul24(iy, f_row) + (int)__umul24(ix, p.src_ld) + f_ci;
                pa[i].load(src + (ok ? off : 0));                 // element 0 is always a valid address;
                f_ok = ok ? (f_ok | (1u << i)) : (f_ok & ~(1u << i));   // zeroed when written to LDS, not while in flight
            }
            // advance this thread's k chunk by one slice
            f_k += BK; f_ci += BK;
#pragma unroll
            for (int w = 0; w < BK / 8; ++w) {
                if (f_ci >= p.C) { f_ci -= p.C; if (++f_kx == p.kw) { f_kx = 0; ++f_ky; } }
            }
        } else if constexpr (VEC8) {
            const int k0 = kt * BK + (tid & 3) * 8;
            const bool kin = k0 < p.Kred;
            int tap = 0, ci = 0, ky = 0, kx = 0;
            if (kin) { tap = k0 / p.C; ci = k0 - tap * p.C; ky = tap / p.kw; kx = tap - ky * p.kw; }
#pragma unroll
            for (int i = 0; i < AROWS; ++i) {
                int iy, ix;
                if (kin && r_img[i] >= 0 && tap_source(p, r_y[i], r_x[i], ky, kx, iy, ix)) {
                    size_t off = ((size_t)(r_img[i] * p.H + iy) * p.W + ix) * p.src_ld + ci;
                    pa[i].load(src + off);
                } else {
                    pa[i].zero();
                }
            }
        } else {
            int k = kt * BK;
            int tap = k / p.C, ci = k - tap * p.C;
            int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll 4
            for (int j = 0; j < BK; ++j) {
                float v = 0.f;
                int iy, ix;
                if (k + j < p.Kred && r_img[0] >= 0 && tap_source(p, r_y[0], r_x[0], ky, kx, iy, ix)) {
                    size_t off = ((size_t)(r_img[0] * p.H + iy) * p.W + ix) * p.src_ld + ci;
                    v = ur_to_float(src[off]);
                }
                pa_s[j] = v;
                if (++ci == p.C) { ci = 0; if (++kx == p.kw) { kx = 0; ++ky; } }
            }
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            int chunk = tid + i * NTHREADS;
            int n = chunk >> 2, kc = (chunk & 3) * 8;
            if (chunk < BN * 4 && n0 + n < p.Ncols) pb[i].load(wgt + (size_t)(n0 + n) * p.Kpad + kt * BK + kc);
            else pb[i].zero();
        }
    };
    auto commit = [&]() {
        if constexpr (VEC8) {
#pragma unroll
            for (int i = 0; i < AROWS; ++i) {
                if constexpr (FAST) { if (!((f_ok >> i) & 1)) pa[i].zero(); }
                pa[i].store(&As[((tid >> 2) + 64 * i) * PITCH + (tid & 3) * 8]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < BK; ++j) As[tid * PITCH + j] = ur_from_float<T>(pa_s[j]);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            int chunk = tid + i * NTHREADS;
            if (chunk < BN * 4) pb[i].store(&Bs[(chunk >> 2) * PITCH + (chunk & 3) * 8]);
        }
    };

    f32x16_t acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int ktiles = p.Kpad / BK;
    const int lrow = lane & 31, lhalf = lane >> 5;
    gather(0);
    for (int kt = 0; kt < ktiles; ++kt) {
        __syncthreads();                 // previous slice fully consumed
        commit();
        __syncthreads();
        if (kt + 1 < ktiles) gather(kt + 1);
        if constexpr (std::is_same<T, bf16_t>::value) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t af[2], bfr[NT];
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    af[a] = *reinterpret_cast<const bf16x8_t*>(&As[(wave * 64 + a * 32 + lrow) * PITCH + ks * 16 + lhalf * 8]);
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    bfr[b] = *reinterpret_cast<const bf16x8_t*>(&Bs[(b * 32 + lrow) * PITCH + ks * 16 + lhalf * 8]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
            }
        } else {
            // 32x32x2 fp32: lane half h supplies k = 16h + s at step s (same permutation on A and B)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 af[2], bfr[NT];
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    af[a] = *reinterpret_cast<const float4*>(&As[(wave * 64 + a * 32 + lrow) * PITCH + lhalf * 16 + q * 4]);
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    bfr[b] = *reinterpret_cast<const float4*>(&Bs[(b * 32 + lrow) * PITCH + lhalf * 16 + q * 4]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bfr[b].x, acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bfr[b].y, acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bfr[b].z, acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bfr[b].w, acc[a][b], 0, 0, 0);
                    }
            }
        }
    }

    // ---- epilogue: bias, store, per-channel partial statistics ---------------------------
    T* __restrict__ dst = static_cast<T*>(p.dst);
    float s1[NT], s2[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) { s1[b] = 0.f; s2[b] = 0.f; }
    if (p.vec_store) {
        // The accumulator layout gives a lane one column and 32 rows; stores of one element per lane would be
        // 2-4 bytes each.  Stage each 64 x 32 sub-tile in the wave's own rows of As (no other wave reads them)
        // and write it back out as 16-byte row pieces.  Layers with short reductions spend more time here than in
        // the slice loop, so the common case (full tile, 32 | channels of a transposed-convolution tap) takes a
        // path without per-element bounds tests, divisions or statistics it does not need.
        constexpr int VE = 16 / (int)sizeof(T);      // elements per 16-byte piece
        constexpr int CPR = 32 / VE;                 // pieces per 32-column row
        T* stage = As + wave * 64 * PITCH;
        const bool full = m0 + BM <= p.M && n0 + BN <= p.Ncols;
        float bv[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int n = n0 + b * 32 + lrow;
            bv[b] = (p.bias && n < p.Ncols) ? p.bias[p.scatter ? n % p.cout_t : n] : 0.f;
        }
        if (p.stats) {
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const bool nok = n0 + b * 32 + lrow < p.Ncols;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        float v = acc[a][b][r] + bv[b];
                        if (!full) v = (m0 + wave * 64 + row < p.M && nok) ? v : 0.f;
                        s1[b] += v; s2[b] += v * v;
                    }
            }
        }
        size_t rowoff[CPR];
        bool rowok[CPR];
#pragma unroll
        for (int i = 0; i < CPR; ++i) {
            const int row = (lane + 64 * i) / CPR;
            const int m = m0 + wave * 64 + row;
            rowok[i] = m < p.M;
            if (!p.scatter) {
                rowoff[i] = (size_t)m * p.dst_ld;
            } else {
                const int img = m / (p.OH * p.OW);
                const int rem = m - img * (p.OH * p.OW);
                const int y = rem / p.OW, x = rem - y * p.OW;
                rowoff[i] = ((size_t)(img * 2 * p.OH + 2 * y) * (2 * p.OW) + 2 * x) * p.dst_ld;
            }
        }
        const bool tap_uniform = p.scatter && p.cout_t % 32 == 0;      // a 32-column sub-tile lies inside one tap
#pragma unroll
        for (int b = 0; b < NT; ++b) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    stage[row * PITCH + lrow] = ur_from_float<T>(acc[a][b][r] + bv[b]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            size_t coloff = (size_t)(n0 + b * 32);                      // column offset of the sub-tile, plain layout
            if (tap_uniform) {
                const int tap = (n0 + b * 32) / p.cout_t, co = (n0 + b * 32) - tap * p.cout_t;
                coloff = ((size_t)(tap >> 1) * (2 * p.OW) + (tap & 1)) * p.dst_ld + co;
            }
#pragma unroll
            for (int i = 0; i < CPR; ++i) {
                const int idx = lane + 64 * i;
                const int row = idx / CPR, piece = idx % CPR;
                const int nn = n0 + b * 32 + piece * VE;
                if (full || (rowok[i] && nn < p.Ncols)) {
                    size_t off;
                    if (!p.scatter || tap_uniform) {
                        off = rowoff[i] + coloff + piece * VE;
                    } else {
                        const int tap = nn / p.cout_t, co = nn - tap * p.cout_t;
                        off = rowoff[i] + ((size_t)(tap >> 1) * (2 * p.OW) + (tap & 1)) * p.dst_ld + co;
                    }
                    typedef __attribute__((ext_vector_type(4))) float raw16_t;
                    const raw16_t piece16 = *reinterpret_cast<const raw16_t*>(stage + row * PITCH + piece * VE);
                    if (UR_IGEMM_NT) __builtin_nontemporal_store(piece16, reinterpret_cast<raw16_t*>(dst + off));     // written once, read by the next kernel
                    else *reinterpret_cast<raw16_t*>(dst + off) = piece16;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int n = n0 + b * 32 + lrow;
        const bool nok = n < p.Ncols;
        int sc_tap = 0, sc_co = n;
        if (p.scatter) { sc_tap = n / p.cout_t; sc_co = n - sc_tap * p.cout_t; }
        const float bv = (p.bias && nok) ? p.bias[p.scatter ? sc_co : n] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                const int m = m0 + row;
                if (m < p.M && nok) {
                    float v = acc[a][b][r] + bv;
                    size_t off;
                    if (!p.scatter) {
                        off = (size_t)m * p.dst_ld + n;
                    } else {
                        int img = m / (p.OH * p.OW);
                        int rem = m - img * (p.OH * p.OW);
                        int y = rem / p.OW, x = rem - y * p.OW;
                        int dy = sc_tap >> 1, dx = sc_tap & 1;
                        off = ((size_t)(img * 2 * p.OH + 2 * y + dy) * (2 * p.OW) + 2 * x + dx) * p.dst_ld + sc_co;
                    }
                    dst[off] = ur_from_float<T>(v);
                    s1[b] += v; s2[b] += v * v;
                }
            }
        }
    }
    }
    if (p.stats) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            float t1 = s1[b] + __shfl_xor(s1[b], 32, 64);
            float t2 = s2[b] + __shfl_xor(s2[b], 32, 64);
            if (lhalf == 0) { red[wave][b * 32 + lrow][0] = t1; red[wave][b * 32 + lrow][1] = t2; }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.Ncols) {
            float t1 = red[0][tid][0] + red[1][tid][0] + red[2][tid][0] + red[3][tid][0];
            float t2 = red[0][tid][1] + red[1][tid][1] + red[2][tid][1] + red[3][tid][1];
            float* o = p.stats + ((size_t)mt * p.Ncols + n0 + tid) * 2;
            o[0] = t1; o[1] = t2;
        }
    }
}

// ---- weight packing ---------------------------------------------------------------------
// dst[n][k] (row length Kpad, zero padded), k = (tap, c) with c fastest;
// element = w[n*sn + c*sc + tap*st]  (strides select OIHW forward / backward-data / ConvT views)
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int nrows, int taps, int cdim,
                                   int Kpad, long long sn, long long sc, long long st, int nsplit, long long sn2) {
    long long total = (long long)nrows * Kpad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int n = (int)(i / Kpad), k = (int)(i - (long long)n * Kpad);
        float v = 0.f;
        if (k < taps * cdim) {
            int tap = k / cdim, c = k - tap * cdim;
            // rows may themselves be (tap2, n2) pairs (ConvT forward): n = tap2*nsplit + n2
            long long base;
            if (nsplit > 0) { int tap2 = n / nsplit, n2 = n - tap2 * nsplit; base = (long long)n2 * sn + (long long)tap2 * sn2; }
            else base = (long long)n * sn;
            v = w[base + (long long)c * sc + (long long)tap * st];
        }
        dst[i] = ur_from_float<T>(v);
    }
}

struct PackView { int nrows, taps, cdim, nsplit; long long sn, sc, st, sn2; };

int pack_view(const ur_conv_desc* d, int mode, PackView* v) {
    const long long khw = (long long)d->kh * d->kw;
    switch (mode) {
    case UR_PACK_FWD:       // w[co][ci][ky][kx]; rows co, k=(tap,ci)
        *v = {d->cout, (int)khw, d->cin, 0, (long long)d->cin * khw, khw, 1, 0}; return 0;
    case UR_PACK_DGRAD:     // rows ci, k=(tap,co)
        *v = {d->cin, (int)khw, d->cout, 0, khw, (long long)d->cin * khw, 1, 0}; return 0;
    case UR_PACK_CONVT_FWD: // w[ci][co][dy][dx]; rows (tap,co), k = ci
        *v = {4 * d->cout, 1, d->cin, d->cout, 4, (long long)d->cout * 4, 0, 1}; return 0;
    case UR_PACK_CONVT_DGRAD: // rows ci, k=(tap,co)
        *v = {d->cin, 4, d->cout, 0, (long long)d->cout * 4, 4, 1, 0}; return 0;
    }
    return -1;
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

template <typename T, int BN>
int launch_igemm(IgemmParams& p, hipStream_t st) {
    p.mtiles = ur_cdiv(p.M, BM);
    p.ntiles = ur_cdiv(p.Ncols, BN);
    const int grid = round_up(p.mtiles, 8) * p.ntiles;
    const uintptr_t sa = (uintptr_t)p.src;
    const bool vec = (p.C % 8 == 0) && (p.src_ld % 8 == 0) && (sa % 16 == 0);
    constexpr int VE = 16 / (int)sizeof(T);
    p.vec_store = (p.dst_ld % VE == 0) && (p.Ncols % VE == 0) && ((uintptr_t)p.dst % 16 == 0) &&
                  (!p.scatter || p.cout_t % VE == 0);
    const bool fast = vec && (p.stride == 1 || p.stride == 2) && (long long)p.N * p.H * p.W * p.src_ld < (1LL << 31) &&
                      (long long)p.W * p.src_ld < (1LL << 24) && p.H < (1 << 22) && p.kh * p.kw < 4096;
    if (vec && fast) hipLaunchKernelGGL((conv_igemm_kernel<T, BN, true, true>), dim3(grid), dim3(NTHREADS), 0, st, p);
    else if (vec) hipLaunchKernelGGL((conv_igemm_kernel<T, BN, true, false>), dim3(grid), dim3(NTHREADS), 0, st, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, BN, false, false>), dim3(grid), dim3(NTHREADS), 0, st, p);
    return ur_check_launch("conv_igemm");
}

template <typename T>
int dispatch_igemm(IgemmParams& p, hipStream_t st) {
    // 128-column tiles (246 registers, two workgroups per CU) halve the number of times a row tile's prologue, A gather
    // and epilogue run, at the price of occupancy.  Measured per layer against 64-column tiles (three per CU): they win
    // on GEMM-like layers with >= 4 reduction slices and enough tiles for >= 2 full rounds (transposed convolutions
    // 128->64 ... 512->256: 207 -> 175, 139 -> 124, 104 -> 91 us; their backward-data 129 -> 111 us) and lose on short
    // reductions (1x1 64->128: 82 -> 93 us) and small grids (1024->512 at 32x64: 95 -> 103 us).
    // URESNET_IGEMM_BN128=1 / 0 forces / forbids them for A/B runs.
    static const int bn128 = [] { const char* e = getenv("URESNET_IGEMM_BN128"); return !e ? -1 : (e[0] == '1' ? 1 : 0); }();
    const bool wide = sizeof(T) == 2 && p.Ncols >= 128 && p.Kred >= 128 &&
                      (long long)ur_cdiv(p.M, BM) * ur_cdiv(p.Ncols, 128) >= 1024;
    if (p.Ncols > 64 && (bn128 == 1 || (bn128 < 0 && wide))) return launch_igemm<T, 128>(p, st);
    if (p.Ncols > 32) return launch_igemm<T, 64>(p, st);
    return launch_igemm<T, 32>(p, st);
}

int check_desc(const ur_conv_desc* d) {
    UR_REQUIRE(d != nullptr, "null conv descriptor");
    UR_REQUIRE(d->dtype == UR_F32 || d->dtype == UR_BF16, "conv: bad dtype %d", d->dtype);
    UR_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "conv: non-positive dimension");
    UR_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0, "conv: bad kernel geometry");
    UR_REQUIRE(d->in_ld >= d->cin && d->out_ld >= d->cout, "conv: pitch smaller than channel count");
    UR_REQUIRE((long long)d->n * d->h * d->w < (1LL << 31) && (long long)d->n * d->oh * d->ow < (1LL << 31),
               "conv: more than 2^31 pixels");
    return UR_OK;
}

int check_conv_out(const ur_conv_desc* d) {
    int oh = (d->h + 2 * d->pad - d->kh) / d->stride + 1, ow = (d->w + 2 * d->pad - d->kw) / d->stride + 1;
    UR_REQUIRE(oh == d->oh && ow == d->ow, "conv: output %dx%d does not match geometry (expected %dx%d)", d->oh,
               d->ow, oh, ow);
    return UR_OK;
}

}  // namespace

extern "C" {

size_t ur_conv2d_packed_elems(const ur_conv_desc* d, int mode) {
    PackView v;
    if (!d || pack_view(d, mode, &v)) return 0;
    return (size_t)v.nrows * round_up(v.taps * v.cdim, BK);
}

int ur_conv2d_pack_weight(const float* w, void* packed, const ur_conv_desc* d, int mode, ur_stream_t s) {
    if (int e = check_desc(d)) return e;
    PackView v;
    UR_REQUIRE(pack_view(d, mode, &v) == 0, "pack_weight: bad mode %d", mode);
    UR_REQUIRE(w && packed, "pack_weight: null pointer");
    if (mode >= UR_PACK_CONVT_FWD) UR_REQUIRE(d->kh == 2 && d->kw == 2, "pack_weight: ConvT modes need a 2x2 kernel");
    const int Kpad = round_up(v.taps * v.cdim, BK);
    const long long total = (long long)v.nrows * Kpad;
    const int grid = ur_grid_1d(total, 256);
    UR_DISPATCH_DTYPE(d->dtype, hipLaunchKernelGGL((pack_weight_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)s, w,
                                                   (T*)packed, v.nrows, v.taps, v.cdim, Kpad, v.sn, v.sc, v.st,
                                                   v.nsplit, v.sn2));
    return ur_check_launch("pack_weight");
}

int ur_conv2d_stats_rows(const ur_conv_desc* d) { return d ? ur_cdiv((long long)d->n * d->oh * d->ow, BM) : 0; }

int ur_conv2d_fwd(const void* x, const void* wp, const float* bias, void* y, float* stats, const ur_conv_desc* d,
                  ur_stream_t s) {
    if (int e = check_desc(d)) return e;
    if (int e = check_conv_out(d)) return e;
    UR_REQUIRE(x && wp && y, "conv2d_fwd: null pointer");
    IgemmParams p{};
    p.src = x; p.wgt = wp; p.bias = bias; p.dst = y; p.stats = stats;
    p.N = d->n; p.H = d->h; p.W = d->w; p.C = d->cin;
    p.OH = d->oh; p.OW = d->ow; p.Ncols = d->cout;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.Kred = d->kh * d->kw * d->cin; p.Kpad = round_up(p.Kred, BK);
    p.M = d->n * d->oh * d->ow;
    p.src_ld = d->in_ld; p.dst_ld = d->out_ld; p.mode = 0;
    UR_DISPATCH_DTYPE(d->dtype, return dispatch_igemm<T>(p, (hipStream_t)s));
}

int ur_conv2d_dgrad(const void* dy, const void* wp, void* dx, const ur_conv_desc* d, ur_stream_t s) {
    if (int e = check_desc(d)) return e;
    if (int e = check_conv_out(d)) return e;
    UR_REQUIRE(dy && wp && dx, "conv2d_dgrad: null pointer");
    IgemmParams p{};
    p.src = dy; p.wgt = wp; p.dst = dx;
    p.N = d->n; p.H = d->oh; p.W = d->ow; p.C = d->cout;      // gather from dy
    p.OH = d->h; p.OW = d->w; p.Ncols = d->cin;               // rows are input pixels
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.Kred = d->kh * d->kw * d->cout; p.Kpad = round_up(p.Kred, BK);
    p.M = d->n * d->h * d->w;
    p.src_ld = d->out_ld; p.dst_ld = d->in_ld; p.mode = 1;
    UR_DISPATCH_DTYPE(d->dtype, return dispatch_igemm<T>(p, (hipStream_t)s));
}

// URESNET_CONVT_GEMM=0: the transposed convolutions stay on the gather kernel (A/B runs)
static bool convt_gemm_enabled() {
    const char* e = getenv("URESNET_CONVT_GEMM");          // read per call: tests switch it inside one process
    return !e || e[0] != '0';
}

static int check_convT(const ur_conv_desc* d) {
    if (int e = check_desc(d)) return e;
    UR_REQUIRE(d->kh == 2 && d->kw == 2 && d->stride == 2 && d->pad == 0, "convT2x2: kernel must be 2x2 stride 2 pad 0");
    UR_REQUIRE(d->oh == 2 * d->h && d->ow == 2 * d->w, "convT2x2: output must be exactly 2x the input");
    return UR_OK;
}

int ur_convT2x2_fwd(const void* x, const void* wp, const float* bias, void* y, const ur_conv_desc* d, ur_stream_t s) {
    if (int e = check_convT(d)) return e;
    UR_REQUIRE(x && wp && y, "convT2x2_fwd: null pointer");
    if (convt_gemm_enabled() && ur_convt_gemm_supported(d, 0, x, y)) return ur_convt_gemm_fwd(x, wp, bias, y, d, (hipStream_t)s);
    IgemmParams p{};
    p.src = x; p.wgt = wp; p.bias = bias; p.dst = y;
    p.N = d->n; p.H = d->h; p.W = d->w; p.C = d->cin;
    p.OH = d->h; p.OW = d->w; p.Ncols = 4 * d->cout;          // a 1x1 GEMM with 4*cout columns
    p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0;
    p.Kred = d->cin; p.Kpad = round_up(p.Kred, BK);
    p.M = d->n * d->h * d->w;
    p.src_ld = d->in_ld; p.dst_ld = d->out_ld; p.mode = 0; p.scatter = 1; p.cout_t = d->cout;
    UR_DISPATCH_DTYPE(d->dtype, return dispatch_igemm<T>(p, (hipStream_t)s));
}

int ur_convT2x2_dgrad(const void* dy, const void* wp, void* dx, const ur_conv_desc* d, ur_stream_t s) {
    if (int e = check_convT(d)) return e;
    UR_REQUIRE(dy && wp && dx, "convT2x2_dgrad: null pointer");
    if (convt_gemm_enabled() && ur_convt_gemm_supported(d, 1, dx, dy)) return ur_convt_gemm_dgrad(dy, wp, dx, d, (hipStream_t)s);
    IgemmParams p{};                                          // a forward 2x2 stride-2 conv over dy
    p.src = dy; p.wgt = wp; p.dst = dx;
    p.N = d->n; p.H = d->oh; p.W = d->ow; p.C = d->cout;
    p.OH = d->h; p.OW = d->w; p.Ncols = d->cin;
    p.kh = 2; p.kw = 2; p.stride = 2; p.pad = 0;
    p.Kred = 4 * d->cout; p.Kpad = round_up(p.Kred, BK);
    p.M = d->n * d->h * d->w;
    p.src_ld = d->out_ld; p.dst_ld = d->in_ld; p.mode = 0;
    UR_DISPATCH_DTYPE(d->dtype, return dispatch_igemm<T>(p, (hipStream_t)s));
}

}  // extern "C"
