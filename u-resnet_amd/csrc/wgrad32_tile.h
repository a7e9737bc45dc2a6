// 3x3 / stride 1 / pad 1 backward-weight in blocks of 32 x 32 channels on the bf16 MFMA (included by the dense and the
// grouped convolution files).
//
//   D[co][ci] = sum_pixels dy[p][co] * x[p + tap][ci]   for one block of 32 dy channels and one block of 32 x channels
// costs one v_mfma_f32_32x32x16_bf16 per tap and 16 pixels.  Same scheme as the 64 x 64 tile kernel: a workgroup stages
// a 4 x 32-pixel dy tile and its 6 x 34 x halo in LDS (64 bytes per pixel), the reduction index is the pixel, so both
// operands come from the transposing LDS read ds_read_b64_tr_b16; wave w owns tile row w.  After the tile walk the four
// waves' accumulators are summed through LDS and written as one partial row per split:
//   dense (DIAG = false): row[co][ci][tap] in OIHW order -- the layers with 32 | channels that the 64-block kernel
//                         cannot take (ResLayer4: 64 -> 32, 32 -> 32);
//   DIAG:                 4 channels per group: only the eight diagonal 4 x 4 blocks are the grouped gradient,
//                         row[c][b][tap]; the matrix cores make the other seven eighths free next to the two tensor
//                         reads (a direct kernel is bound by its 144 multiply-adds per loaded quad).
#pragma once
#include "ur_common.h"

namespace wg32 {
constexpr int NT = 256, TW = 32, TH = 4, HP = 36, HCOLS = 34;
constexpr int DY_BYTES = TH * TW * 64, HALO_BYTES = (TH + 2) * HP * 64;
constexpr int DY_ITERS = TH * TW * 4 / NT, HALO_PIECES = (TH + 2) * HCOLS * 4, HALO_ITERS = (HALO_PIECES + NT - 1) / NT;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* base0, const unsigned char* base1, int imm) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base0 + imm));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base1 + imm));
    s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, both);
}

struct Params {
    const bf16_t* x;      // [N][H][W][ld_x]
    const bf16_t* dy;     // [N][H][W][ld_dy]
    float* partial;       // [splits][row_elems]
    int N, H, W, Cin, ld_x, ld_dy;
    int ci_blocks;        // dense: x channel blocks (blockIdx.y = cob * ci_blocks + cib); DIAG: unused (cob == cib)
    int tiles_x, tiles_y, tiles;
    long long row_elems;
};

template <bool DIAG>
__global__ __launch_bounds__(NT, 2) void wgrad32_kernel(Params p) {
    constexpr int RED_BYTES = 4 * 16 * 64 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[DY_BYTES + HALO_BYTES > RED_BYTES ? DY_BYTES + HALO_BYTES : RED_BYTES];
    unsigned char* dyl = lds;
    unsigned char* xl = lds + DY_BYTES;
    const int split = blockIdx.x, nsplit = gridDim.x;
    const int cob = DIAG ? blockIdx.y : blockIdx.y / p.ci_blocks;
    const int cib = DIAG ? blockIdx.y : blockIdx.y - cob * p.ci_blocks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int piece = tid & 3;
    const int H = p.H, W = p.W;

    bf16x8_t pd[DY_ITERS], ph[HALO_ITERS];
    auto prefetch = [&](int tile) {
        const int img = tile / (p.tiles_y * p.tiles_x);
        const int trem = tile - img * (p.tiles_y * p.tiles_x);
        const int ty0 = (trem / p.tiles_x) * TH, tx0 = (trem % p.tiles_x) * TW;
        const bf16_t* dyb = p.dy + (size_t)img * H * W * p.ld_dy + cob * 32 + piece * 8;
        const bf16_t* xb = p.x + (size_t)img * H * W * p.ld_x + cib * 32 + piece * 8;
#pragma unroll
        for (int j = 0; j < DY_ITERS; ++j) {
            const int pix = (tid >> 2) + j * (NT / 4);
            const int oy = ty0 + (pix >> 5), ox = tx0 + (pix & 31);
            if (oy < H && ox < W) pd[j] = *reinterpret_cast<const bf16x8_t*>(dyb + ((size_t)oy * W + ox) * p.ld_dy);
            else pd[j] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < HALO_ITERS; ++j) {
            const int hp = (tid >> 2) + j * (NT / 4);
            const int hy = hp / HCOLS, hx = hp - hy * HCOLS;
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
            if (hp < (TH + 2) * HCOLS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                ph[j] = *reinterpret_cast<const bf16x8_t*>(xb + ((size_t)gy * W + gx) * p.ld_x);
            else ph[j] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < DY_ITERS; ++j) {
            const int pix = (tid >> 2) + j * (NT / 4);
            *reinterpret_cast<bf16x8_t*>(dyl + pix * 64 + piece * 16) = pd[j];
        }
#pragma unroll
        for (int j = 0; j < HALO_ITERS; ++j) {
            const int hp = (tid >> 2) + j * (NT / 4);
            const int hy = hp / HCOLS, hx = hp - hy * HCOLS;
            if (hp < (TH + 2) * HCOLS) *reinterpret_cast<bf16x8_t*>(xl + (hy * HP + hx) * 64 + piece * 16) = ph[j];
        }
    };

    // transposing-read lane constants: 16-lane group g reads pixels 8 (g >> 1) + (li >> 2) (+4 for the second read),
    // channels 16 (g & 1) + 4 (li & 3)
    const int g = lane >> 4, li = lane & 15;
    const int lpix = 8 * (g >> 1) + (li >> 2), lch = 16 * (g & 1) + 4 * (li & 3);
    const unsigned char* a_base[2];
    const unsigned char* b_base[3][2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        a_base[h2] = dyl + (wave * TW + lpix + 4 * h2) * 64 + lch * 2;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) b_base[dx][h2] = xl + (wave * HP + lpix + 4 * h2 + dx) * 64 + lch * 2;
    }

    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    int tile = split;
    if (tile < p.tiles) prefetch(tile);
    for (; tile < p.tiles; tile += nsplit) {
        __syncthreads();
        commit();
        __syncthreads();
        if (tile + nsplit < p.tiles) prefetch(tile + nsplit);
#pragma unroll
        for (int kx = 0; kx < 2; ++kx) {
            const bf16x8_t af = tr_frag(a_base[0], a_base[1], kx * 16 * 64);
#pragma unroll
            for (int dyk = 0; dyk < 3; ++dyk)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const bf16x8_t bfr = tr_frag(b_base[dx][0], b_base[dx][1], (dyk * HP + kx * 16) * 64);
                    acc[dyk * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[dyk * 3 + dx], 0, 0, 0);
                }
        }
    }

    // sum the four waves' blocks tap by tap through LDS (fixed order)
    float* red = reinterpret_cast<float*>(lds);
    float* row = p.partial + (size_t)split * p.row_elems;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = tid + NT * k, r = o >> 6, ln = o & 63;
            const int co = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), ci = ln & 31;
            if (!DIAG || (co >> 2) == (ci >> 2)) {
                const float v = (red[(0 * 16 + r) * 64 + ln] + red[(1 * 16 + r) * 64 + ln]) +
                                (red[(2 * 16 + r) * 64 + ln] + red[(3 * 16 + r) * 64 + ln]);
                if (DIAG) row[((size_t)(cob * 32 + co) * 4 + (ci & 3)) * 9 + t] = v;
                else row[((size_t)(cob * 32 + co) * p.Cin + cib * 32 + ci) * 9 + t] = v;
            }
        }
    }
}

// out[i] = sum_s partial[s][i], fixed summation order
template <int SL>
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ partial, int nparts, long long row_elems,
                                                       float* __restrict__ out) {
    long long i;
    float t;
    if (ur_slab_sum<SL>(partial, row_elems, nparts, i, t)) out[i] = t;
}

static inline int splits(int n, int h, int w, int blocks) {
    const int tiles = n * ur_cdiv(h, TH) * ur_cdiv(w, TW);
    int want = 512 / blocks;
    if (want < 1) want = 1;
    return want < tiles ? want : tiles;
}

// launches the block kernel and the slab sum; `out` receives row_elems floats
template <bool DIAG>
static inline int run(const bf16_t* x, const bf16_t* dy, float* out, float* partial, int n, int h, int w, int cin, int cout,
                      int ld_x, int ld_dy, hipStream_t st) {
    Params p{};
    p.x = x; p.dy = dy; p.partial = partial;
    p.N = n; p.H = h; p.W = w; p.Cin = cin; p.ld_x = ld_x; p.ld_dy = ld_dy;
    p.ci_blocks = cin / 32;
    p.tiles_x = ur_cdiv(w, TW); p.tiles_y = ur_cdiv(h, TH); p.tiles = n * p.tiles_x * p.tiles_y;
    const int blocks = DIAG ? cout / 32 : (cout / 32) * (cin / 32);
    p.row_elems = DIAG ? (long long)cout * 4 * 9 : (long long)cout * cin * 9;
    const int ns = splits(n, h, w, blocks);
    hipLaunchKernelGGL((wgrad32_kernel<DIAG>), dim3(ns, blocks), dim3(NT), 0, st, p);
    if (int e = ur_check_launch("wgrad32")) return e;
    if (ur_slab_lanes(ns) == 32)
        hipLaunchKernelGGL(slab_sum_kernel<32>, dim3(ur_slab_grid(p.row_elems, 32)), dim3(256), 0, st, partial, ns, p.row_elems, out);
    else
        hipLaunchKernelGGL(slab_sum_kernel<8>, dim3(ur_slab_grid(p.row_elems, 8)), dim3(256), 0, st, partial, ns, p.row_elems, out);
    return ur_check_launch("wgrad32_sum");
}
}  // namespace wg32
