// Backward-weight of the 3x3 / stride 1 / pad 1 convolutions, bf16 NHWC, LDS-tiled on the bf16 MFMA.
//
//   dW[co][ci][tap] = sum_pixels dy[p][co] * x[p + tap - 1][ci]
// Workgroup (4 waves) owns a 64(co) x 64(ci) x 9(tap) block of dW and a strided subset of the
// 4 x 32-pixel spatial tiles (split over gridDim.y).  Per tile it stages the dy tile [128 px][64 co]
// and the x halo [6 x 34 px][64 ci] in LDS once; the halo then serves all nine taps, so every
// staged element feeds 9 x 64 MACs.  The reduction index of the GEMM is the pixel, while memory is
// channel-contiguous, so both MFMA operands are fetched with the transposing LDS read
// ds_read_b64_tr_b16.  LDS rows are 128 B per pixel with the two 64-byte halves swapped on every
// second pixel pair (conflict-free for the 4-pixel x 64-byte blocks a transposing read touches); the halo
// row pitch is 36 pixels so every read address is a per-lane constant plus a compile-time immediate.
// Wave (a, b) accumulates co-half a x ci-half b for all 9 taps (9 x 16 accumulator registers);
// 72 MFMAs per tile per wave.  Partial blocks go to one fp32 slab per split and are summed in a
// fixed order by a second kernel that also transposes to PyTorch's [co][ci][ky][kx].
#include "ur_common.h"
#include "wgrad32_tile.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int TW = 32, BO = 64, BI = 64, NTHREADS = 256;
constexpr int HPITCH = 36;                                 // halo row pitch in pixels (34 used)
constexpr int HALO_COLS = TW + 2;

struct WgTileParams {
    const bf16_t* x;     // [N][H][W][ld_x]
    const bf16_t* dy;    // [N][H][W][ld_dy]
    float* ws;           // [nsplit][Cout][9][Cin]
    int N, H, W, Cin, Cout, ld_x, ld_dy;
    int tiles_x, tiles_y, tiles;
    int ci_blocks;
    int pairs, nsplit, xcd_order;
};

// Workgroup id -> (channel-block pair, split).  Ids congruent mod 8 run on the same XCD (one L2 each): with
// xcd_order every pair of a split lands on one XCD, so the x and dy tiles that the pairs of a split share are
// fetched from memory once instead of once per XCD.
__device__ __forceinline__ void wg_decode(const WgTileParams& p, int& pair, int& split) {
    const int id = blockIdx.x;
    if (p.xcd_order) {
        const int xcd = id & 7, j = id >> 3;
        pair = j % p.pairs;
        split = (j / p.pairs) * 8 + xcd;
    } else {
        pair = id % p.pairs;
        split = id / p.pairs;
    }
}

// byte offset of channel c (multiple of 4) of pixel `pix` inside an image with 128-byte pixel rows
__device__ __forceinline__ int swz(int pix, int c) {
    return pix * 128 + ((((c >> 5) ^ (pix >> 1)) & 1) << 6) + (c & 31) * 2;
}

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* base0, const unsigned char* base1, int imm) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base0 + imm));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base1 + imm));
    s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, both);
}

// the same with 32-bit LDS byte addresses (half the registers of generic pointers)
__device__ __forceinline__ bf16x8_t tr_frag_u(unsigned a0, unsigned a1, int imm) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(size_t)(a0 + imm));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(size_t)(a1 + imm));
    s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, both);
}

template <int TH, int MINW>
__global__ __launch_bounds__(NTHREADS, MINW) void wgrad3x3_tile_kernel(WgTileParams p) {
    constexpr int DY_PIX = TH * TW, HALO_ROWS = TH + 2;
    constexpr int DY_BYTES = DY_PIX * 128, HALO_BYTES = HALO_ROWS * HPITCH * 128;
    constexpr int DY_ITERS = DY_PIX * 8 / NTHREADS;
    constexpr int HALO_ITERS = (HALO_ROWS * HALO_COLS * 8 + NTHREADS - 1) / NTHREADS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[DY_BYTES + HALO_BYTES];
    unsigned char* dyl = lds;
    unsigned char* xl = lds + DY_BYTES;

    int pair, split;
    wg_decode(p, pair, split);
    const int nsplit = p.nsplit;
    const int cob = pair / p.ci_blocks, cib = pair - cob * p.ci_blocks;
    const int co0 = cob * BO, ci0 = cib * BI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;

    // ---- staging: piece i = tid + 256 j covers channels (tid & 7) * 8 .. +8 of pixel i >> 3; the pixel
    // coordinates are recomputed per tile (a few integer ops) rather than held in registers ------------
    const int piece = tid & 7, piece_ch = piece * 8;

    bf16x8_t pd[DY_ITERS], ph[HALO_ITERS];
    auto prefetch = [&](int tile) {
        const int img = tile / (p.tiles_y * p.tiles_x);
        const int trem = tile - img * (p.tiles_y * p.tiles_x);
        const int ty0 = (trem / p.tiles_x) * TH, tx0 = (trem % p.tiles_x) * TW;
        const bf16_t* dyb = p.dy + (size_t)img * p.H * p.W * p.ld_dy + co0 + piece_ch;
        const bf16_t* xb = p.x + (size_t)img * p.H * p.W * p.ld_x + ci0 + piece_ch;
#pragma unroll
        for (int j = 0; j < DY_ITERS; ++j) {
            const int pix = (tid >> 3) + j * (NTHREADS / 8);
            const int oy = ty0 + (pix >> 5), ox = tx0 + (pix & 31);
            if (oy < p.H && ox < p.W) pd[j] = *reinterpret_cast<const bf16x8_t*>(dyb + ((size_t)oy * p.W + ox) * p.ld_dy);
            else pd[j] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < HALO_ITERS; ++j) {
            const int hp = (tid >> 3) + j * (NTHREADS / 8);
            const int hy = hp / HALO_COLS, hx = hp - hy * HALO_COLS;
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
            if (hp < HALO_ROWS * HALO_COLS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                ph[j] = *reinterpret_cast<const bf16x8_t*>(xb + ((size_t)gy * p.W + gx) * p.ld_x);
            else ph[j] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < DY_ITERS; ++j) {
            const int pix = (tid >> 3) + j * (NTHREADS / 8);
            *reinterpret_cast<bf16x8_t*>(dyl + swz(pix, piece_ch)) = pd[j];
        }
#pragma unroll
        for (int j = 0; j < HALO_ITERS; ++j) {
            const int hp = (tid >> 3) + j * (NTHREADS / 8);
            const int hy = hp / HALO_COLS, hx = hp - hy * HALO_COLS;
            if (hp < HALO_ROWS * HALO_COLS) *reinterpret_cast<bf16x8_t*>(xl + swz(hy * HPITCH + hx, piece_ch)) = ph[j];
        }
    };

    // ---- transposing-read lane constants ---------------------------------------------------------------
    // 16-lane group g: pixels 8*(g>>1) + q (+4 for the second read), channels 16*(g&1) + 4*p', lane = 4q + p'
    const int g = lane >> 4, li = lane & 15;
    const int lpix = 8 * (g >> 1) + (li >> 2), lch = 16 * (g & 1) + 4 * (li & 3);
    const unsigned char* a_base[2];
    const unsigned char* b_base[3][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        a_base[s] = dyl + swz(lpix + 4 * s, wa * 32 + lch);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) b_base[dx][s] = xl + swz(lpix + 4 * s + dx, wb * 32 + lch);
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
        for (int r = 0; r < TH; ++r) {
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) {
                // pixel bases are multiples of 4, so they do not disturb the lane-constant swizzle bits
                const bf16x8_t af = tr_frag(a_base[0], a_base[1], (r * TW + kx * 16) * 128);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const bf16x8_t bfr = tr_frag(b_base[dx][0], b_base[dx][1], ((r + dy) * HPITCH + kx * 16) * 128);
                        acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[dy * 3 + dx], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- partial block -> slab [split][co][tap][ci] ------------------------------------------------------
    const int lx = lane & 31, lh = lane >> 5;
    float* slab = p.ws + (size_t)split * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            slab[((size_t)co * 9 + t) * p.Cin + ci0 + wb * 32 + lx] = acc[t][r];
        }
}

// ---- version 2: vertical strips with a rolling halo ---------------------------------------------------------------------
// A workgroup walks a contiguous run of tiles DOWN a 32-pixel column strip (tile order: image, column strip, row).  The
// x halo of a 2-row tile is 4 rows, two of which the previous tile of the strip already staged: the LDS halo is a ring of
// 4 row slots (global row y lives in slot (y + 1) & 3 relative to the strip start), and a tile loads only its 2 new rows.
// Against version 1 (every tile stages all 4 rows, consecutive tiles go to different workgroups) this halves the x bytes
// fetched and staged; dy is read once either way.  M16 selects v_mfma_f32_16x16x32_bf16 (K = the 32 pixels of a tile row,
// 16 x 16 output blocks) instead of 32x32x16: same cycles per FLOP, a different operand footprint.
template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int HPITCH2 = 48;                                // halo row pitch (pixels): a multiple of 16 keeps every swizzle bit lane-constant

// 128-byte pixel rows in four 32-byte (16-channel) slots, slot index XORed with pixel bits 1 and 3: the 8 pixels x 32 bytes
// that one 32-lane group of a transposing read touches (pixels w..w+3 and w+8..w+11 of one row) land on 8 distinct
// 32-byte bank groups for every w.
__device__ __forceinline__ int swz16(int pix, int c) {
    return pix * 128 + ((((c >> 4) ^ ((pix >> 1) & 1) ^ (((pix >> 3) & 1) << 1)) & 3) << 5) + (c & 15) * 2;
}

// SHIFT (with M16): the x fragments of the three horizontal taps are the same pixels shifted by 0, 1 and 2, and a lane's
// fragment is 8 CONSECUTIVE pixels of one channel: one 12-pixel read (three transposing reads instead of six) serves all
// three taps -- registers 0..3, registers 1..4, and four v_alignbit in between.  Backward-weight has the fewest MFMAs per LDS
// read of the three convolution kernels; this takes 43 % of its reads away.
template <bool M16, bool SHIFT = false>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad3x3_strip_kernel(WgTileParams p) {
    static_assert(M16 || !SHIFT, "the shifted-fragment form is written for the 16x16x32 operand layout");
    constexpr int TH = 2;
    constexpr int HP = M16 ? HPITCH2 : HPITCH;
    constexpr int DY_PIX = TH * TW, DY_BYTES = DY_PIX * 128, HALO_BYTES = 4 * HP * 128;
    static_assert(NTHREADS == TW * 8, "one thread per 16-byte piece of a 32-pixel row of 64 channels");
    __shared__ __attribute__((aligned(16))) unsigned char lds[DY_BYTES + HALO_BYTES];
    unsigned char* dyl = lds;
    unsigned char* xl = lds + DY_BYTES;
    auto sw = [](int pix, int c) { return M16 ? swz16(pix, c) : swz(pix, c); };

    int pair, split;
    wg_decode(p, pair, split);
    const int cob = pair / p.ci_blocks, cib = pair - cob * p.ci_blocks;
    const int co0 = cob * BO, ci0 = cib * BI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int piece_ch = (tid & 7) * 8;

    // tiles [l0, l1) of the order (image, column strip, row); ty runs fastest
    const int per = (p.tiles + p.nsplit - 1) / p.nsplit;
    const int l0 = split * per, l1 = (l0 + per < p.tiles) ? l0 + per : p.tiles;

    // Staging pieces (16 bytes = 8 channels of one pixel).  Main pieces: thread -> pixel column tid >> 3 of the tile, one
    // piece per tile row / halo row, so a row is one load per thread at a per-thread constant offset from a uniform row
    // address, and its LDS address is a per-thread constant plus a uniform row offset (row pitches of the LDS images are
    // multiples of the swizzle period).  Tail pieces: halo columns 32 and 33, 16 pieces per row, threads 0..63.
    const int col = tid >> 3;
    const unsigned lds_main = (unsigned)sw(col, piece_ch);
    const unsigned dy_voff = (unsigned)(col * p.ld_dy + piece_ch) * 2u, x_voff = (unsigned)(col * p.ld_x + piece_ch) * 2u;
    const int trow = tid >> 4, tcol = TW + ((tid >> 3) & 1);
    const bf16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    bf16x8_t pd[TH], ph[5];
    // rows [row0, 4) of the halo of tile (img, tx, ty): row0 = 0 at the start of a strip run, else 2 (ph[j] = row row0 + j)
    auto prefetch = [&](int l, bool full) {
        const int ty = l % p.tiles_y, rest = l / p.tiles_y;
        const int tx = rest % p.tiles_x, img = rest / p.tiles_x;
        const int ty0 = ty * TH, tx0 = tx * TW;
        const int row0 = full ? 0 : 2;
        const long long dy_row = ((long long)(img * p.H + ty0) * p.W + tx0) * p.ld_dy + co0;
        const char* dyb = reinterpret_cast<const char*>(p.dy) + dy_row * 2;
        const bool dcol_ok = tx0 + col < p.W;
#pragma unroll
        for (int j = 0; j < TH; ++j) {
            if (ty0 + j < p.H && dcol_ok)
                pd[j] = *reinterpret_cast<const bf16x8_t*>(dyb + (size_t)j * p.W * p.ld_dy * 2 + dy_voff);
            else pd[j] = zero8;
        }
        // address of halo row row0, halo column 0; only dereferenced where the pixel exists
        const long long x_row = ((long long)(img * p.H + ty0 - 1 + row0) * p.W + tx0 - 1) * p.ld_x + ci0;
        const char* xb = reinterpret_cast<const char*>(p.x) + x_row * 2;
        const bool xcol_ok = (unsigned)(tx0 - 1 + col) < (unsigned)p.W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 2 || full) {
                if ((unsigned)(ty0 - 1 + row0 + j) < (unsigned)p.H && xcol_ok)
                    ph[j] = *reinterpret_cast<const bf16x8_t*>(xb + (size_t)j * p.W * p.ld_x * 2 + x_voff);
                else ph[j] = zero8;
            }
        }
        if (trow < 4 - row0 && (unsigned)(ty0 - 1 + row0 + trow) < (unsigned)p.H && (unsigned)(tx0 - 1 + tcol) < (unsigned)p.W)
            ph[4] = *reinterpret_cast<const bf16x8_t*>(xb + (((size_t)trow * p.W + tcol) * p.ld_x + piece_ch) * 2);
        else ph[4] = zero8;
    };
    // k = parity of the tile's position in its strip run: halo row hr of the tile lives in ring slot (hr + 2k) & 3
    auto commit = [&](int k, bool full) {
#pragma unroll
        for (int j = 0; j < TH; ++j) *reinterpret_cast<bf16x8_t*>(dyl + j * TW * 128 + lds_main) = pd[j];
        const int row0 = full ? 0 : 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 2 || full) {
                const int slot = (row0 + j + 2 * k) & 3;
                *reinterpret_cast<bf16x8_t*>(xl + slot * HP * 128 + lds_main) = ph[j];
            }
        }
        if (trow < 4 - row0) {
            const int slot = (row0 + trow + 2 * k) & 3;
            *reinterpret_cast<bf16x8_t*>(xl + sw(slot * HP + tcol, piece_ch)) = ph[4];
        }
    };

    const int g = lane >> 4, li = lane & 15;
    constexpr int NACC = M16 ? 36 : 9;
    typedef typename std::conditional<M16, f32x4_t, f32x16_t>::type acc_t;
    acc_t acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < (M16 ? 4 : 16); ++r) acc[t][r] = 0.f;

    // transposing-read lane constants (see the version-1 kernel for the 32x32x16 pattern).  16x16x32: 16-lane group g
    // covers pixels 8g + q (+4 for the second read) of a 32-pixel row, lane = 4q + p' addresses channels 4p'..4p'+3 of a
    // 16-channel block; after the transpose lane j of the group holds channel j, pixels 8g..8g+7.
    const int lpix = M16 ? 8 * g + (li >> 2) : 8 * (g >> 1) + (li >> 2);
    const int lch = M16 ? 4 * (li & 3) : 16 * (g & 1) + 4 * (li & 3);
    constexpr int NH = M16 ? 2 : 1;                      // 16-channel halves of the wave's 32 co / 32 ci
    const unsigned dyl0 = (unsigned)(size_t)dyl, xl0 = (unsigned)(size_t)xl;
    unsigned a_base[NH][2], b_base[SHIFT ? 1 : 3][NH][SHIFT ? 3 : 2];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int s = 0; s < 2; ++s) a_base[h][s] = dyl0 + sw(lpix + 4 * s, wa * 32 + 16 * h + lch);
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int s = 0; s < (SHIFT ? 3 : 2); ++s)
#pragma unroll
            for (int dx = 0; dx < (SHIFT ? 1 : 3); ++dx) b_base[dx][h][s] = xl0 + sw(lpix + 4 * s + dx, wb * 32 + 16 * h + lch);

    // Ring rotation without a second copy of the loop: halo row hr sits in slot (hr + 2k) & 3, i.e. two rows further on for
    // hr < 2 and two rows back for hr >= 2 when k = 1.  The two byte offsets are added to the lane bases per tile.
    auto compute = [&](int k) {
        const int rot[2] = {k ? 2 * HP * 128 : 0, k ? -2 * HP * 128 : 0};
        if constexpr (M16) {
            bf16x8_t af[TH][2];
            sfor<TH>([&](auto R_) { sfor<2>([&](auto H_) {
                constexpr int r = R_, h = H_;
                af[r][h] = tr_frag_u(a_base[h][0], a_base[h][1], (r * TW) * 128);
            }); });
            sfor<4>([&](auto HR_) {
                constexpr int hr = HR_;
                bf16x8_t bf[3][2];
                if constexpr (SHIFT) {
                    sfor<2>([&](auto H_) {
                        constexpr int h = H_;
                        unsigned w[6];                       // pixels 8g .. 8g+11 of this lane's channel, two per register
                        sfor<3>([&](auto S_) {
                            constexpr int sx = S_;
                            const s16x4_t q = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (lds_s16x4_ptr)(size_t)(b_base[0][h][sx] + rot[hr >> 1] + (hr * HP) * 128));
                            const uint2 u = __builtin_bit_cast(uint2, q);
                            w[2 * sx] = u.x; w[2 * sx + 1] = u.y;
                        });
                        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                        bf[0][h] = __builtin_bit_cast(bf16x8_t, (u32x4_t){w[0], w[1], w[2], w[3]});
                        bf[1][h] = __builtin_bit_cast(bf16x8_t, (u32x4_t){__builtin_amdgcn_alignbit(w[1], w[0], 16),
                                                                          __builtin_amdgcn_alignbit(w[2], w[1], 16),
                                                                          __builtin_amdgcn_alignbit(w[3], w[2], 16),
                                                                          __builtin_amdgcn_alignbit(w[4], w[3], 16)});
                        bf[2][h] = __builtin_bit_cast(bf16x8_t, (u32x4_t){w[1], w[2], w[3], w[4]});
                    });
                } else {
                    sfor<3>([&](auto DX_) { sfor<2>([&](auto H_) {
                        constexpr int dx = DX_, h = H_;
                        bf[dx][h] = tr_frag_u(b_base[dx][h][0] + rot[hr >> 1], b_base[dx][h][1] + rot[hr >> 1], (hr * HP) * 128);
                    }); });
                }
                sfor<TH>([&](auto R_) {
                    constexpr int r = R_, dy = hr - r;
                    if constexpr (dy >= 0 && dy <= 2) {
                        sfor<3>([&](auto DX_) { sfor<2>([&](auto HCO_) { sfor<2>([&](auto HCI_) {
                            constexpr int dx = DX_, hco = HCO_, hci = HCI_;
                            constexpr int a = ((dy * 3 + dx) * 2 + hco) * 2 + hci;
                            acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][hco], bf[dx][hci], acc[a], 0, 0, 0);
                        }); }); });
                    }
                });
            });
        } else {
            sfor<2>([&](auto KX_) {
                constexpr int kx = KX_;
                bf16x8_t af[TH];
                sfor<TH>([&](auto R_) { constexpr int r = R_; af[r] = tr_frag_u(a_base[0][0], a_base[0][1], (r * TW + kx * 16) * 128); });
                sfor<4>([&](auto HR_) {
                    constexpr int hr = HR_;
                    bf16x8_t bf[3];
                    sfor<3>([&](auto DX_) {
                        constexpr int dx = DX_;
                        bf[dx] = tr_frag_u(b_base[dx][0][0] + rot[hr >> 1], b_base[dx][0][1] + rot[hr >> 1], (hr * HP + kx * 16) * 128);
                    });
                    sfor<TH>([&](auto R_) {
                        constexpr int r = R_, dy = hr - r;
                        if constexpr (dy >= 0 && dy <= 2) {
                            sfor<3>([&](auto DX_) {
                                constexpr int dx = DX_;
                                acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[r], bf[dx], acc[dy * 3 + dx], 0, 0, 0);
                            });
                        }
                    });
                });
            });
        }
    };

    int k = 0;
    bool full = true;                                    // the tile about to be committed starts a strip run
    if (l0 < l1) prefetch(l0, true);
    for (int l = l0; l < l1; ++l) {
        __syncthreads();
        commit(k, full);
        __syncthreads();
        const bool next_full = (l + 1) % p.tiles_y == 0;        // the next tile is the first row of another strip
        if (l + 1 < l1) prefetch(l + 1, next_full);
        compute(k);
        k = next_full ? 0 : (k ^ 1);
        full = next_full;
    }

    // ---- partial block -> slab [split][co][tap][ci] ------------------------------------------------------
    float* slab = p.ws + (size_t)split * p.Cout * 9 * p.Cin;
    if constexpr (M16) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int hco = 0; hco < 2; ++hco)
#pragma unroll
                for (int hci = 0; hci < 2; ++hci)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int co = co0 + wa * 32 + hco * 16 + 4 * g + i;
                        slab[((size_t)co * 9 + t) * p.Cin + ci0 + wb * 32 + hci * 16 + li] = acc[(t * 2 + hco) * 2 + hci][i];
                    }
    } else {
        const int lx = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[((size_t)co * 9 + t) * p.Cin + ci0 + wb * 32 + lx] = acc[t][r];
            }
    }
}

// dw[co][ci][tap] = sum_s ws[s][co][tap][ci]
template <int SL>
__global__ __launch_bounds__(256) void wgrad3x3_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout,
                                                              int Cin, int nsplit) {
    long long idx;
    float t;
    if (!ur_slab_sum<SL>(ws, (long long)Cout * 9 * Cin, nsplit, idx, t)) return;
    const int ci = (int)(idx % Cin);
    const long long q = idx / Cin;
    const int tap = (int)(q % 9), co = (int)(q / 9);
    dw[((long long)co * Cin + ci) * 9 + tap] = t;
}

struct Plan { int pairs, nsplit, tiles; };

// Tile height: 2 rows (two workgroups per CU; measured 4 % faster) or 4 rows (one workgroup per CU,
// accumulators in AGPRs) when URESNET_WGRAD_TH=4 is set for A/B runs.
int tile_rows() {
    static const int th = [] { const char* e = getenv("URESNET_WGRAD_TH"); return (e && e[0] == '4') ? 4 : 2; }();
    return th;
}

Plan make_plan(const ur_conv_desc* d) {
    Plan pl;
    pl.pairs = (d->cout / BO) * (d->cin / BI);
    pl.tiles = d->n * ur_cdiv(d->h, tile_rows()) * ur_cdiv(d->w, TW);
    int want = (512 + pl.pairs - 1) / pl.pairs;          // two resident workgroups per CU; fewer, larger slabs
    if (want > pl.tiles) want = pl.tiles;
    if (want < 1) want = 1;
    if (want > 65535) want = 65535;
    pl.nsplit = want;
    return pl;
}

// 1: the 64 x 64-block kernel; 2: the 32 x 32-block kernel (channel counts that are multiples of 32 but not both of 64)
int supported(const ur_conv_desc* d) {
    if (!d || d->dtype != UR_BF16) return 0;
    if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->oh == d->h && d->ow == d->w)) return 0;
    if (d->in_ld % 8 != 0 || d->out_ld % 8 != 0) return 0;
    if ((long long)d->n * d->h * d->w >= (1LL << 31)) return 0;
    if (d->cout % BO == 0 && d->cin % BI == 0) return 1;
    if (d->cout % 32 == 0 && d->cin % 32 == 0) return 2;
    return 0;
}

}  // namespace

extern "C" {

int ur_wgrad3x3_tile_supported(const ur_conv_desc* d) { return supported(d); }

size_t ur_wgrad3x3_tile_workspace(const ur_conv_desc* d) {
    if (!supported(d)) return 0;
    if (supported(d) == 2)
        return (size_t)wg32::splits(d->n, d->h, d->w, (d->cout / 32) * (d->cin / 32)) * d->cout * d->cin * 9 * sizeof(float);
    Plan pl = make_plan(d);
    return (size_t)pl.nsplit * d->cout * 9 * d->cin * sizeof(float);
}

int ur_wgrad3x3_tile(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes, const ur_conv_desc* d,
                     ur_stream_t s) {
    return ur_wgrad3x3_tile_marked(x, dy, dw, workspace, ws_bytes, d, nullptr, s);
}

int ur_wgrad3x3_tile_marked(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes, const ur_conv_desc* d,
                            void* event_after_main, ur_stream_t s) {
    UR_REQUIRE(supported(d), "wgrad3x3_tile: unsupported geometry (needs bf16, 3x3 s1 p1, 32 | cin, 32 | cout)");
    UR_REQUIRE(x && dy && dw && workspace, "wgrad3x3_tile: null pointer");
    UR_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "wgrad3x3_tile: tensors must be 16-byte aligned");
    UR_REQUIRE(ws_bytes >= ur_wgrad3x3_tile_workspace(d), "wgrad3x3_tile: workspace too small");
    if (supported(d) == 2) {
        UR_REQUIRE(!event_after_main, "wgrad3x3_tile_marked: the 32 x 32-block kernel takes no event");
        return wg32::run<false>((const bf16_t*)x, (const bf16_t*)dy, dw, (float*)workspace, d->n, d->h, d->w, d->cin, d->cout,
                                d->in_ld, d->out_ld, (hipStream_t)s);
    }
    Plan pl = make_plan(d);
    WgTileParams p{};
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.ws = (float*)workspace;
    p.N = d->n; p.H = d->h; p.W = d->w; p.Cin = d->cin; p.Cout = d->cout; p.ld_x = d->in_ld; p.ld_dy = d->out_ld;
    p.tiles_x = ur_cdiv(d->w, TW); p.tiles_y = ur_cdiv(d->h, tile_rows()); p.tiles = pl.tiles;
    p.ci_blocks = d->cin / BI;
    p.pairs = pl.pairs; p.nsplit = pl.nsplit;
    p.xcd_order = pl.nsplit % 8 == 0;
    // 4 (default): column strips with a rolling halo on the 16x16x32 MFMA, one x fragment read per three horizontal taps;
    // 3: a read per tap; 2: strips on 32x32x16; 1: the first kernel (every tile stages its whole halo).
    static const int version = [] { const char* e = getenv("URESNET_WGRAD_V"); return e ? atoi(e) : 4; }();
    if (version >= 2 && version <= 4) {
        p.tiles_y = ur_cdiv(d->h, 2); p.tiles = d->n * p.tiles_x * p.tiles_y;
        if (version == 4) hipLaunchKernelGGL((wgrad3x3_strip_kernel<true, true>), dim3(pl.pairs * pl.nsplit), dim3(NTHREADS), 0, (hipStream_t)s, p);
        else if (version == 3) hipLaunchKernelGGL((wgrad3x3_strip_kernel<true>), dim3(pl.pairs * pl.nsplit), dim3(NTHREADS), 0, (hipStream_t)s, p);
        else hipLaunchKernelGGL((wgrad3x3_strip_kernel<false>), dim3(pl.pairs * pl.nsplit), dim3(NTHREADS), 0, (hipStream_t)s, p);
    } else
    if (tile_rows() == 4) hipLaunchKernelGGL((wgrad3x3_tile_kernel<4, 1>), dim3(pl.pairs * pl.nsplit), dim3(NTHREADS), 0, (hipStream_t)s, p);
    else hipLaunchKernelGGL((wgrad3x3_tile_kernel<2, 2>), dim3(pl.pairs * pl.nsplit), dim3(NTHREADS), 0, (hipStream_t)s, p);
    if (int e = ur_check_launch("wgrad3x3_tile")) return e;
    if (event_after_main && hipEventRecord((hipEvent_t)event_after_main, (hipStream_t)s) != hipSuccess) {
        ur_set_error("wgrad3x3_tile_marked: hipEventRecord failed");
        return UR_ERR_LAUNCH;
    }
    const long long per = (long long)d->cout * 9 * d->cin;
    if (ur_slab_lanes(pl.nsplit) == 32)
        hipLaunchKernelGGL(wgrad3x3_reduce_kernel<32>, dim3(ur_slab_grid(per, 32)), dim3(256), 0, (hipStream_t)s, p.ws, dw,
                           d->cout, d->cin, pl.nsplit);
    else
        hipLaunchKernelGGL(wgrad3x3_reduce_kernel<8>, dim3(ur_slab_grid(per, 8)), dim3(256), 0, (hipStream_t)s, p.ws, dw,
                           d->cout, d->cin, pl.nsplit);
    return ur_check_launch("wgrad3x3_reduce");
}

}  // extern "C"
