// 3x3 / stride 1 / pad 1 convolution, bf16 NHWC, as an LDS-tiled implicit GEMM on the bf16 MFMA.
// Used for the forward pass and (with tap-flipped, transposed weights) for backward-data of the
// layers that carry ~94 % of U-ResNet's FLOPs.
//
// Workgroup (4 waves) -> 4R x 32 output pixels x 64 output channels, R = 4 (R = 2 for small images).  A channel
// count of 32 (mod 64) runs its last block half empty (zero weights, no stores): still twice as fast as the generic
// kernel on the 32-channel layers.
// The reduction runs over 16-channel slices of the input; two LDS stages alternate, one barrier per slice.
// A stage holds
//   * the (4R+2) x 34 pixel halo of the slice, 32 bytes per pixel; the two 16-byte pieces of a pixel are
//     swapped in every second group of 8 columns, which makes the 32-consecutive-pixel fragment reads
//     conflict-free and keeps every row at a constant byte distance (fragment address = lane base + immediate),
//   * the 9 x 64 x 16 weight slice, pre-packed in MFMA-fragment order and copied by LDS-DMA.
// Halo loads cover two slices at a time: four consecutive lanes fetch the 64 contiguous bytes (32 channels)
// of a pixel, so one wave-wide load touches 16 cache lines instead of 32 -- with 32-byte pieces the
// texture-address path, not the MFMA, set the pace.  Lanes 0,1 of a quad own the even slice of the pair and
// write it to LDS during the odd slice before it; lanes 2,3 own the odd slice and write it one slice later.
// Wave w owns output rows [wR, wR+R).  For one horizontal tap offset it walks the R+2 halo rows once
// ("phases"); the fragment of halo row i feeds the MFMAs of every (output row, vertical tap) pair with
// row + tap = i, and each of the 6 weight fragments serves R rows: (R+2) + 6 LDS reads per 6R MFMAs.
// LDS reads are issued by hand (inline assembly, counted s_waitcnt) three phases ahead of their MFMAs.
// Left to itself the compiler sinks every read down to its first use and each group of MFMAs then waits
// out a full LDS round trip.
// The epilogue stages the wave's tile in LDS and writes 16-byte pieces (full 128-byte pixel rows).
#include "ur_common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

// Output stores and the BatchNorm-input loads of the epilogue touch their bytes once per launch: non-temporal (step
// 29.41 -> 29.35 ms, the gain of the BatchNorm passes' own non-temporal accesses is ten times that).  -DUR_CONV_NT=0: ordinary.
#ifndef UR_CONV_NT
#define UR_CONV_NT 1
#endif
#if UR_CONV_NT
#define OUT_STORE(q, v) __builtin_nontemporal_store(v, q)
#define Y_LOAD(q) __builtin_nontemporal_load(q)
#else
#define OUT_STORE(q, v) (*(q) = (v))
#define Y_LOAD(q) (*(q))
#endif
namespace {

constexpr int TW = 32, BN = 64, KC = 16, NTHREADS = 256;
constexpr int HALO_W = TW + 2;
constexpr int ROW_BYTES = HALO_W * 32;               // one halo row of one slice
constexpr int W_PIECES = 9 * 2 * 64;                 // 16-byte pieces of one weight slice
constexpr int W_GROUPS = W_PIECES / 64;              // one group = one wave-wide 16-byte LDS-DMA (1 KiB)
constexpr int W_GROUP_ITERS = (W_GROUPS + 3) / 4;
constexpr int W_BYTES = W_PIECES * 16;               // 18432
constexpr int OUT_PITCH = 144;                       // bytes per staged output pixel (128 + 16: the two half-waves land on disjoint banks)

template <int R> struct Geo {
    static constexpr int TH = 4 * R;
    static constexpr int HALO_PIX = (TH + 2) * HALO_W;
    static constexpr int HALO_PIECES = HALO_PIX * 4;               // 16-byte pieces of a 32-channel slice pair
    static constexpr int HALO_ITERS = (HALO_PIECES + NTHREADS - 1) / NTHREADS;
    static constexpr int HALO_BYTES = HALO_PIX * 32;
    static constexpr int STAGE_BYTES = HALO_BYTES + W_BYTES;
    static constexpr int OUT_BYTES = 4 * R * TW * OUT_PITCH;       // epilogue staging, all four waves
    static constexpr int RED_BYTES = 4 * BN * 2 * 4;
    static constexpr int DUMP_BYTES = NTHREADS * 16;               // per-lane slots for the halo writes of non-owner lanes
    static constexpr int LDS_BYTES = (2 * STAGE_BYTES > OUT_BYTES + RED_BYTES ? 2 * STAGE_BYTES : OUT_BYTES + RED_BYTES) + DUMP_BYTES;
};

struct TileParams {
    const bf16_t* x;      // [N][H][W][ld_in]
    const bf16_t* wpk;    // [Cout/64][Cin/16][9][2][64][8]
    bf16_t* y;            // [N][H][W][ld_out]
    float* stats;         // [tiles][Cout][2] or null
    const bf16_t* addend; // [N][H][W][ld_add] or null: added to the rounded result (a second gradient of the same tensor,
    int ld_add;           //   or the residual of an inference-mode block)
    const float* scale;   // [Cout] or null: result = acc * scale + shift (BatchNorm with fixed statistics, folded)
    const float* shift;
    int relu;             // clamp at zero after the affine map and the addend
    // EPI 3 / 4 (backward-data whose result is the gradient dz of a BatchNorm(+ReLU) output): also emit that BatchNorm's
    // backward partial sums per tile, stats[tile][c] = (sum g, sum g * x_hat) with g = dz where the activation passed
    const bf16_t* bn_y;   // [N][H][W][ld_bny] the BatchNorm's input (raw convolution output)
    const bf16_t* bn_z;   // [N][H][W][Cout] its activated output, or null: the ReLU mask is rebuilt from y
    const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta;
    int ld_bny;
    int N, H, W, Cin, Cout;
    int ld_in, ld_out;
    int tiles_x, tiles_y, tiles;     // spatial tiles per image row / column / total (N * ty * tx)
    int nblocks;                     // ceil(Cout / 64): the last block of a 32 (mod 64) channel count is half empty
};

// ---- hand-issued LDS traffic ---------------------------------------------------------------------------
// Asynchronous global -> LDS copy (LDS-DMA): every lane supplies its own 16-byte source, the data land at
// (wave-uniform) LDS byte address dst + lane * 16.  Counted in vmcnt; the issuing wave orders it for readers
// with s_waitcnt vmcnt(0) followed by the workgroup barrier.
__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8_t& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write16(unsigned addr, const bf16x8_t& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
// wait until at most N of this wave's LDS operations are outstanding (they complete in issue order)
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory");
}

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Read schedule of one slice.  Phase P = (horizontal offset g = P / NA, halo row i = P % NA) needs the row
// fragment of halo row i and, in the first three phases of an offset, the two weight fragments of vertical tap i.
template <int R> struct Sched {
    static constexpr int NA = R + 2;
    static constexpr int PH = 3 * NA;
    static constexpr int LOOK = 3;                         // phases between a read and its MFMAs
    static constexpr int WS = 5;                           // staging piece j is handled in phase WS + j
    static constexpr int nreads(int u) { return u >= PH ? 0 : ((u % NA) < 3 ? 3 : 1); }
    // LDS operations issued after the reads of phase t and before the wait of slot t.  Slot u issues the reads of
    // phase u + LOOK, waits for those of phase u, then issues its MFMAs with at most one halo write among them.
    // lgkmcnt holds 4 bits.
    static constexpr int younger(int t, int nwrites) {
        int n = 0;
        for (int u = t + 1; u <= t + LOOK; ++u) n += nreads(u);
        for (int u = t - LOOK; u < t; ++u) n += (u >= WS && u < WS + nwrites) ? 1 : 0;
        return n < 15 ? n : 15;
    }
    static constexpr int nmfma(int u) {                    // MFMAs of phase u
        int n = 0;
        for (int dy = 0; dy < 3; ++dy) n += ((u % NA) - dy >= 0 && (u % NA) - dy < R) ? 2 : 0;
        return n;
    }
};

__device__ __forceinline__ int halo_lds_offset(int hy, int hx, int piece) {
    return hy * ROW_BYTES + hx * 32 + ((piece ^ ((hx >> 3) & 1)) << 4);
}

// EPI selects the epilogue at compile time (run-time options turn into hundreds of selects on the 128 accumulators):
// 0 = rounded result (+ per-channel statistics), 1 = + addend, 2 = folded BatchNorm (scale, shift, optional residual, ReLU),
// 3 = rounded result + BatchNorm(+ReLU)-backward partial sums of the tensor it is the gradient of (ReLU mask rebuilt from
// the BatchNorm's input), 4 = 3 with an addend, 5 / 6 = 3 / 4 with the mask taken from the stored activated tensor
template <int R, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void conv3x3_tile_kernel(TileParams p) {
    using G = Geo<R>;
    using S = Sched<R>;
    constexpr int NA = S::NA, NW = G::HALO_ITERS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS_BYTES];

    // XCD-aware order: ids congruent mod 8 share an L2; they walk the channel blocks of one spatial tile
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int tile = (slot / p.nblocks) * 8 + xcd;
    const int nb = slot % p.nblocks;
    if (tile >= p.tiles) return;
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int trem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty0 = (trem / p.tiles_x) * G::TH, tx0 = (trem % p.tiles_x) * TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lx = lane & 31, lh = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)lds;

    // ---- staging bookkeeping: halo pieces this thread owns (fixed for the whole channel loop) -------
    // Piece i = tid + 256 j covers 8 channels of pixel i >> 2: channels [8 (i & 3), +8) of a 32-channel slice
    // pair, i.e. piece (i & 1) of slice (i >> 1) & 1 of the pair; i & 3 = tid & 3 for every j.
    // Branch-free: surplus threads of the last round repeat the last piece of their kind (same value to the same
    // address).
    // Nothing per piece is kept in registers (they are needed for accumulators and fragments): offsets are
    // recomputed from the thread id at every use, and the id is laundered through an empty asm so that the
    // compiler does not hoist the loop-invariant results back into 20 registers.
    const int chunks = p.Cin / KC;                 // even: Cin is a multiple of 32
    // weights: 18 groups of 64 pieces (1 KiB); wave w copies groups w, w+4, ... straight into LDS
    const bf16x8_t* wsrc = reinterpret_cast<const bf16x8_t*>(p.wpk) + (size_t)nb * chunks * W_PIECES + lane;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int img_pix0 = img * p.H * p.W;

    bf16x8_t ph[NW];
    // Halo coordinates of the thread's pieces, walked without integer division or 32-bit multiplies (both run at
    // a quarter of the VALU rate, and this code sits between MFMA phases): piece j+1 is 64 pixels = one row and
    // 30 columns further.  Surplus pieces of the last round fall back to the last halo pixel.
    struct Walk {
        int hy, hx;
        __device__ __forceinline__ void start(int t) { const int pix = t >> 2; hy = pix >= HALO_W ? 1 : 0; hx = pix - hy * HALO_W; }
        __device__ __forceinline__ void next() {
            hx += 64 - HALO_W; hy += 1;
            if (hx >= HALO_W) { hx -= HALO_W; hy += 1; }
            if (hy >= G::TH + 2) { hy = G::TH + 1; hx = HALO_W - 1; }
        }
    };
    const int row_elems = p.W * p.ld_in;                                   // < 2^24 (checked on the host)
    const int off00 = (img_pix0 + (ty0 - 1) * p.W + tx0 - 1) * p.ld_in;    // element offset of halo pixel (0, 0); may be negative
    // one piece of the pair of slices starting at `chunk` (even); wk = halo pixel of the piece
    auto load_piece = [&](bf16x8_t& dst, const Walk& wk, int t, int chunk) {
        const bool ok = (unsigned)(ty0 - 1 + wk.hy) < (unsigned)p.H && (unsigned)(tx0 - 1 + wk.hx) < (unsigned)p.W;
        // padding pixels load the first element of the tensor (always a valid address) and are zeroed when written
        // to LDS: no branch, and no arithmetic on the loaded registers while the load is in flight
        const int off = (int)__umul24(wk.hy, row_elems) + (int)__umul24(wk.hx, p.ld_in) + off00 + (t & 3) * 8 + chunk * KC;   // < 2^31
        dst = *reinterpret_cast<const bf16x8_t*>(p.x + (ok ? off : 0));
    };
    // The lanes that own slices of parity `half` write their piece; the other lanes write theirs to a per-lane
    // dump slot behind the stages (a select on the address instead of a branch keeps the whole phase one basic
    // block, so that the scheduler can interleave this arithmetic with the MFMAs).
    auto write_piece = [&](const bf16x8_t& src, const Walk& wk, int t, int half) {
        const bool ok = (unsigned)(ty0 - 1 + wk.hy) < (unsigned)p.H && (unsigned)(tx0 - 1 + wk.hx) < (unsigned)p.W;
        const bf16x8_t v = ok ? src : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned real = lds0 + half * G::STAGE_BYTES + __umul24(wk.hy, ROW_BYTES) + wk.hx * 32 + ((((wk.hx >> 3) ^ t) & 1) << 4);
        const unsigned dump = lds0 + G::LDS_BYTES - NTHREADS * 16 + t * 16;
        lds_write16<0>(((t >> 1) & 1) == half ? real : dump, v);
    };
    // wave w copies weight groups w, w+4, ... (1 KiB each) of slice `chunk` straight into LDS; the two surplus
    // slots of the last round repeat group 17 (same bytes, same place)
    auto dma_group = [&](int chunk, int j) {
        int g = wave_u + 4 * j;
        g = g < W_GROUPS ? g : W_GROUPS - 1;
        lds_dma16(wsrc + (size_t)chunk * W_PIECES + g * 64,
                  __builtin_amdgcn_readfirstlane(lds0 + (chunk & 1) * G::STAGE_BYTES) + G::HALO_BYTES + g * 1024);
    };

    f32x16_t acc[R][2];
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment addresses inside stage 0: row fragments per horizontal offset (rows are + i * ROW_BYTES),
    // weight fragments (fragment f is + f * 1024)
    unsigned a_base[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) a_base[dx] = lds0 + halo_lds_offset(wave * R, lx + dx, lh);
    const unsigned b_base = lds0 + G::HALO_BYTES + lane * 16;

    {
        int t = tid;
        asm volatile("" : "+v"(t));
        Walk wk; wk.start(t);
#pragma unroll
        for (int j = 0; j < NW; ++j) { load_piece(ph[j], wk, t, 0); wk.next(); }
#pragma unroll
        for (int j = 0; j < W_GROUP_ITERS; ++j) dma_group(0, j);
        wk.start(t);
#pragma unroll
        for (int j = 0; j < NW; ++j) { write_piece(ph[j], wk, t, 0); wk.next(); }   // slice 1 stays in the registers of lanes 2,3
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // One reduction slice.  ODD = parity of the slice; NEXT = another slice follows;
    // PAIR = (even slices) another slice pair follows this one, (odd slices) same as NEXT.
    auto slice = [&](int chunk, auto odd_tag, auto next_tag, auto pair_tag) {
        constexpr bool ODD = decltype(odd_tag)::value, NEXT = decltype(next_tag)::value, PAIR = decltype(pair_tag)::value;
        constexpr int NWR = NEXT ? NW : 0;
        const unsigned so = ODD ? G::STAGE_BYTES : 0;              // stage read by this slice; the other one was last
                                                                   // read before the previous barrier and is refilled now
        int t = tid;
        asm volatile("" : "+v"(t));
        Walk wk; wk.start(t);
        // Last slice of a launch that also sums a BatchNorm's backward terms: the halo registers are free from here on, so
        // the first NW of the lane's 4R pieces of that BatchNorm's input are fetched into them now, under this slice's MFMAs,
        // instead of after the main loop where nothing overlaps the read.  Tiles that the epilogue handles piece by piece
        // (image border, half-empty channel block) fetch element 0 instead and load their pieces there as before.
        constexpr bool YPRE = !NEXT && EPI >= 3 && EPI <= 6;
        int ybase = 0, yrow = 0, ycol = 0;
        if constexpr (YPRE) {
            const bool whole = ty0 + G::TH <= p.H && tx0 + TW <= p.W && (nb + 1) * BN <= p.Cout;
            const int l = t & 63, wv = t >> 6;
            ybase = whole ? ((img * p.H + ty0 + wv * R) * p.W + tx0 + (l >> 3)) * p.ld_bny + nb * BN + (l & 7) * 8 : 0;
            yrow = whole ? p.W * p.ld_bny : 0;
            ycol = whole ? 8 * p.ld_bny : 0;
        }
        const unsigned ab[3] = {a_base[0] + so, a_base[1] + so, a_base[2] + so};
        const unsigned bb = b_base + so;
        bf16x8_t ar[4], br[8];
        auto issue = [&](auto PP) {                                // reads of phase PP
            constexpr int P = PP, g = P / NA, i = P % NA;
            if constexpr (P < S::PH) {
                lds_read16<i * ROW_BYTES>(ar[P % 4], ab[g]);
                if constexpr (i < 3) {
                    lds_read16<((i * 3 + g) * 2 + 0) * 1024>(br[(g * 6 + i * 2 + 0) % 8], bb);
                    lds_read16<((i * 3 + g) * 2 + 1) * 1024>(br[(g * 6 + i * 2 + 1) % 8], bb);
                }
            }
        };
        static_for<S::LOOK>([&](auto P) { issue(P); });
        static_for<S::PH>([&](auto PP) {
            constexpr int P = PP, g = P / NA, i = P % NA;
            issue(std::integral_constant<int, P + S::LOOK>{});
            lds_wait<S::younger(P, NWR)>();
            __builtin_amdgcn_sched_barrier(0);
            // staging for the next slice, one piece per phase: weight groups by LDS-DMA, then halo pieces to the
            // other stage; even slices refill the freed registers
            constexpr bool STAGE = NEXT && P >= S::WS && P < S::WS + NW;
            if constexpr (NEXT && P < W_GROUP_ITERS) dma_group(chunk + 1, P);
            if constexpr (STAGE) {
                write_piece(ph[P - S::WS], wk, t, ODD ? 0 : 1);
                if constexpr (!ODD && PAIR) load_piece(ph[P - S::WS], wk, t, chunk + 2);
                wk.next();
            }
            if constexpr (YPRE && P >= S::WS && P < S::WS + NW && P - S::WS < 4 * R) {
                constexpr int j = P - S::WS;
                ph[j] = Y_LOAD(reinterpret_cast<const bf16x8_t*>(p.bn_y + ybase + (j >> 2) * yrow + (j & 3) * ycol));
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int a = i - dy;
                if (a >= 0 && a < R) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[P % 4], br[(g * 6 + dy * 2 + b) % 8], acc[a][b], 0, 0, 0);
                }
            }
            if constexpr (STAGE) {
                // interleave: the piece's address arithmetic (about 30 VALU instructions) issues under the MFMAs
                static_for<S::nmfma(P)>([&](auto) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // this wave's LDS-DMA pieces and halo writes have landed (the halo loads just issued stay in flight)
        if constexpr (!ODD && PAIR) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(NW) : "memory");
        else if constexpr (YPRE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fetches above stay in flight
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    int chunk = 0;
    for (; chunk + 2 < chunks; chunk += 2) {
        slice(chunk, F_{}, T_{}, T_{});
        slice(chunk + 1, T_{}, T_{}, T_{});
    }
    slice(chunk, F_{}, T_{}, F_{});
    slice(chunk + 1, T_{}, F_{}, F_{});

    // ---- epilogue: per-channel partial statistics of the fp32 results, bf16 tile via LDS -> 16-byte stores ----
    unsigned char* ostage = lds + wave * (R * TW * OUT_PITCH);
    float* red = reinterpret_cast<float*>(lds + G::OUT_BYTES);
    const bool interior = ty0 + G::TH <= p.H && tx0 + TW <= p.W;        // no pixel of the tile lies outside the image
    if constexpr (EPI == 2) {
        float sc[2], sh[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = nb * BN + b * 32 + lx;
            sc[b] = c < p.Cout ? p.scale[c] : 0.f; sh[b] = c < p.Cout ? p.shift[c] : 0.f;
        }
        const bool relu_now = p.relu && !p.addend;
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[a][b][r] * sc[b] + sh[b];
                    acc[a][b][r] = relu_now ? fmaxf(v, 0.f) : v;
                }
    }
    if (EPI == 0 && p.stats) {
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        if (interior) {
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float v = acc[a][b][r]; s1[b] += v; s2[b] += v * v; }
        } else {
#pragma unroll
            for (int a = 0; a < R; ++a) {
                const int oy = ty0 + wave * R + a;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int px = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float v = (oy < p.H && tx0 + px < p.W) ? acc[a][b][r] : 0.f;
                        s1[b] += v; s2[b] += v * v;
                    }
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float t1 = s1[b] + __shfl_xor(s1[b], 32, 64);
            const float t2 = s2[b] + __shfl_xor(s2[b], 32, 64);
            if (lh == 0) { red[(wave * BN + b * 32 + lx) * 2] = t1; red[(wave * BN + b * 32 + lx) * 2 + 1] = t2; }
        }
    }
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = (r & 3) + 8 * (r >> 2) + 4 * lh;
                *reinterpret_cast<bf16_t*>(ostage + (a * TW + px) * OUT_PITCH + (b * 32 + lx) * 2) = (bf16_t)acc[a][b][r];
            }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // EPI 3 / 4: a lane stores the same 8 channels (piece = lane & 7) of R * 4 pixels; it keeps their BatchNorm constants
    // and the two running sums per channel in registers (the accumulators are dead once the tile is staged)
    constexpr bool BNSUMS = EPI >= 3 && EPI <= 6, BN_ADD = EPI == 4 || EPI == 6, BN_FROM_Z = EPI == 5 || EPI == 6;
    float bm[8], bis[8], bga[8], bbe[8], bs1[8], bs2[8];
    if constexpr (BNSUMS) {
        // the loads stay below this point: issued before the accumulators are staged they would need 32 more registers
        // in a kernel that has none to spare
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int c0 = nb * BN + (lane & 7) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool okc = c0 + k < p.Cout;
            bm[k] = okc ? p.bn_mean[c0 + k] : 0.f; bis[k] = okc ? p.bn_invstd[c0 + k] : 0.f;
            bga[k] = okc ? p.bn_gamma[c0 + k] : 0.f; bbe[k] = okc ? p.bn_beta[c0 + k] : 0.f;
            bs1[k] = 0.f; bs2[k] = 0.f;
        }
    }
    // Piece i of a lane (i = 0 .. 4R-1) is pixel (row i >> 2, column 8 (i & 3) + (lane >> 3)) of the wave's rows, channels
    // 8 (lane & 7) .. +8: its element offset in a tensor of pitch ld is a per-lane base plus (i >> 2) W ld + (i & 3) 8 ld,
    // two uniform terms with compile-time factors (all tensors hold fewer than 2^31 elements: checked on the host).  Written
    // out per piece as (pixel index) * pitch, the epilogue spent ~190 quarter-rate multiplies per lane on addresses.
    const int pix_lane = (img * p.H + ty0 + wave * R) * p.W + tx0 + (lane >> 3);
    const int ch_lane = nb * BN + (lane & 7) * 8;
    struct Base { int at, ld; };        // element offset of the lane's piece 0 in one tensor, and the tensor's pitch
    auto base_of = [&](int ld) { return Base{pix_lane * ld + ch_lane, ld}; };
    const Base b_out = base_of(p.ld_out), b_bny = base_of(EPI >= 3 ? p.ld_bny : 0), b_z = base_of(EPI >= 5 ? p.Cout : 0),
               b_add = base_of(p.ld_add);
    auto elem_off = [&](int i, const Base& b) { return b.at + ((i >> 2) * p.W + (i & 3) * 8) * b.ld; };
    auto store_piece = [&](int i, bool check) {
        const int idx = lane + 64 * i;
        const int pxl = idx >> 3, piece = idx & 7;
        const int oy = ty0 + wave * R + (pxl >> 5), ox = tx0 + (pxl & 31);
        if (!check || (oy < p.H && ox < p.W && nb * BN + piece * 8 < p.Cout)) {
            bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(ostage + pxl * OUT_PITCH + piece * 16);
            if constexpr (BNSUMS) {
                const bf16x8_t y8 = Y_LOAD(reinterpret_cast<const bf16x8_t*>(p.bn_y + elem_off(i, b_bny)));
                bf16x8_t z8 = y8;
                if constexpr (BN_FROM_Z) z8 = *reinterpret_cast<const bf16x8_t*>(p.bn_z + elem_off(i, b_z));
                if constexpr (BN_ADD) {
                    const bf16x8_t ad = *reinterpret_cast<const bf16x8_t*>(p.addend + elem_off(i, b_add));
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (bf16_t)((float)v[k] + (float)ad[k]);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float xh = ((float)y8[k] - bm[k]) * bis[k];
                    const float pre = BN_FROM_Z ? (float)z8[k] : xh * bga[k] + bbe[k];
                    const float gv = pre > 0.f ? (float)v[k] : 0.f;
                    bs1[k] += gv; bs2[k] += gv * xh;
                }
            }
            if constexpr (EPI == 1 || EPI == 2) {
                if (EPI == 1 || p.addend) {
                    const bf16x8_t ad = *reinterpret_cast<const bf16x8_t*>(p.addend + elem_off(i, b_add));
                    const bool relu = EPI == 2 && p.relu;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float t = (float)v[k] + (float)ad[k];
                        if (relu) t = fmaxf(t, 0.f);
                        v[k] = (bf16_t)t;
                    }
                }
            }
            OUT_STORE(reinterpret_cast<bf16x8_t*>(p.y + elem_off(i, b_out)), v);
        }
    };
    if (interior && (nb + 1) * BN <= p.Cout) {
        if constexpr (BNSUMS) {
            // Batches: all global loads of a batch (BatchNorm input, activated tensor, addend) are issued before the first
            // is used, so a batch pays one memory latency; the scheduling barriers keep the compiler from merging the
            // batches (it then hoists every conversion to the top and spills a hundred registers).
            constexpr int NP = R * 4, BT = (EPI == 3 && NP >= 16) ? 16 : (BN_ADD && R == 4 ? 4 : 8);
#pragma unroll
            for (int b0 = 0; b0 < NP; b0 += BT) {
                bf16x8_t yv[BT], zv[BT], av[BT];
#pragma unroll
                for (int j = 0; j < BT; ++j) {
                    if (b0 + j < NW) yv[j] = ph[b0 + j];          // fetched during the last slice
                    else yv[j] = Y_LOAD(reinterpret_cast<const bf16x8_t*>(p.bn_y + elem_off(b0 + j, b_bny)));
                    if constexpr (BN_FROM_Z) zv[j] = *reinterpret_cast<const bf16x8_t*>(p.bn_z + elem_off(b0 + j, b_z));
                    if constexpr (BN_ADD) av[j] = *reinterpret_cast<const bf16x8_t*>(p.addend + elem_off(b0 + j, b_add));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < BT; ++j) {
                    const int idx = lane + 64 * (b0 + j), pxl = idx >> 3, piece = idx & 7;
                    bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(ostage + pxl * OUT_PITCH + piece * 16);
                    if constexpr (BN_ADD) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = (bf16_t)((float)v[k] + (float)av[j][k]);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float xh = ((float)yv[j][k] - bm[k]) * bis[k];
                        const float pre = BN_FROM_Z ? (float)zv[j][k] : xh * bga[k] + bbe[k];
                        const float gv = pre > 0.f ? (float)v[k] : 0.f;
                        bs1[k] += gv; bs2[k] += gv * xh;
                    }
                    OUT_STORE(reinterpret_cast<bf16x8_t*>(p.y + elem_off(b0 + j, b_out)), v);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < R * 4; ++i) store_piece(i, false);
        }
    } else {
#pragma unroll
        for (int i = 0; i < R * 4; ++i) {
            store_piece(i, true);
            if (BNSUMS && (i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (BNSUMS) {
        // lanes with equal lane & 7 hold the same 8 channels: fold lane bits 3, 4, 5, then one row per wave as in EPI 0
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { bs1[k] += __shfl_xor(bs1[k], o, 64); bs2[k] += __shfl_xor(bs2[k], o, 64); }
        }
        if (lane < 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { red[(wave * BN + lane * 8 + k) * 2] = bs1[k]; red[(wave * BN + lane * 8 + k) * 2 + 1] = bs2[k]; }
        }
    }
    if ((EPI == 0 || BNSUMS) && p.stats) {
        __syncthreads();
        if (tid < BN && nb * BN + tid < p.Cout) {
            float* o = p.stats + ((size_t)tile * p.Cout + nb * BN + tid) * 2;
            o[0] = red[tid * 2] + red[(BN + tid) * 2] + red[(2 * BN + tid) * 2] + red[(3 * BN + tid) * 2];
            o[1] = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1] + red[(2 * BN + tid) * 2 + 1] + red[(3 * BN + tid) * 2 + 1];
        }
    }
}

// Fragment-order weight image.  piece = ((nb*chunks + chunk)*9 + tap)*2 + b, then lane, then 8 k-values.
//   forward : value = w[co = nb*64 + b*32 + (lane&31)][ci = chunk*16 + 8*(lane>>5) + j][tap]
//   dgrad   : the kernel convolves dy (channels = cout of the layer) into dx (channels = cin):
//             value = w[co = chunk*16 + 8*(lane>>5) + j][ci = nb*64 + b*32 + (lane&31)][8 - tap]
// One (64 produced channels) x (16 reduced channels) x 9 taps block of the image = 18 KiB of contiguous output.  Its
// source is 64 runs of 144 floats (forward) or 16 runs of 576 floats (dgrad): read coalesced into LDS, written
// back as 16-byte pieces in fragment order.
constexpr int PACK_BLOCK = BN * KC * 9;
constexpr int PACK_ROW = KC * 9 + 1;              // forward staging rows are 144 floats: one pad word makes the lane stride odd
constexpr int PACK_TILE = BN * PACK_ROW;          // (an even stride of 144 words put 32 lanes on two LDS banks)
// `wide` selects the piece order of the 128-channel kernel (conv3x3_wide.hip): the same 64 x 16 x 9 source block, its
// pieces scattered in runs of 256 bytes: piece (row r, k-half kh, tap) of block (nb, chunk) goes to
// [nb / 2][chunk / 2][dx][dy][nb & 1][r / 16][lane = (r & 15) + 16 * (2 * (chunk & 1) + kh)].
__device__ __forceinline__ void pack_tile_weights_body(const float* __restrict__ w, bf16_t* __restrict__ dst, int cin, int cout,
                                                        int dgrad, int wide, int first, int step, float* tile) {
    const int rows = dgrad ? cin : cout;      // channels produced by the kernel
    const int red = dgrad ? cout : cin;       // channels reduced over
    const int chunks = red / KC;
    const int nblocks = ((rows + BN - 1) / BN) * chunks;       // a last block of 32 channels is padded with zero weights
    for (int blk = first; blk < nblocks; blk += step) {
        const int nb = blk / chunks, chunk = blk - nb * chunks;
        // 36 elements per thread, 12 loads in flight at a time (one load per loop trip left the copy latency-bound)
        constexpr int U = 12;
        static_assert(PACK_BLOCK % (256 * U) == 0, "the staging loop assumes 256 threads");
        for (int base = threadIdx.x; base < PACK_BLOCK; base += 256 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * 256;
                if (!dgrad) {
                    const int n = i / (KC * 9), rem = i - n * (KC * 9);
                    v[u] = nb * BN + n < rows ? w[((size_t)(nb * BN + n) * cin + chunk * KC) * 9 + rem] : 0.f;
                } else {
                    const int k = i / (BN * 9), rem = i - k * (BN * 9);
                    v[u] = nb * BN + rem / 9 < rows ? w[((size_t)(chunk * KC + k) * cin + nb * BN) * 9 + rem] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * 256;
                if (!dgrad) { const int n = i / (KC * 9); tile[n * PACK_ROW + (i - n * (KC * 9))] = v[u]; }
                else tile[i] = v[u];
            }
        }
        __syncthreads();
        for (int pi = threadIdx.x; pi < PACK_BLOCK / 8; pi += blockDim.x) {
            const int lane = pi & 63, t = pi >> 6, b = t & 1, tap = t >> 1;
            const int n = b * 32 + (lane & 31), k0 = 8 * (lane >> 5);
            bf16x8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = (bf16_t)(dgrad ? tile[(k0 + j) * (BN * 9) + n * 9 + (8 - tap)] : tile[n * PACK_ROW + (k0 + j) * 9 + tap]);
            size_t piece = (size_t)blk * (PACK_BLOCK / 8) + pi;
            if (wide) {
                const int dyk = tap / 3, dxk = tap - dyk * 3;
                piece = ((((((size_t)(nb >> 1) * (chunks >> 1) + (chunk >> 1)) * 3 + dxk) * 3 + dyk) * 2 + (nb & 1)) * 4 + (n >> 4)) * 64 +
                        (n & 15) + 16 * (2 * (chunk & 1) + (lane >> 5));
            }
            *reinterpret_cast<bf16x8_t*>(dst + piece * 8) = o;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void pack_tile_weights_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst, int cin,
                                                                int cout, int dgrad, int wide) {
    __shared__ float tile[PACK_TILE];
    pack_tile_weights_body(w, dst, cin, cout, dgrad, wide, blockIdx.x, gridDim.x, tile);
}

// Many layers in one launch: table[e] = {w, dst, cin, cout, dgrad, first block} (64-bit words).  One workgroup per
// 64 x 16 x 9 block of any layer: the blocks of all layers form one list (a 1024 x 1024 layer has 1024 of them, a
// 64 x 64 layer 4), so the launch is balanced whatever the mix; the workgroup finds its layer by bisection.
__global__ __launch_bounds__(256) void pack_tile_weights_multi_kernel(const long long* __restrict__ table, int entries) {
    __shared__ float tile[PACK_TILE];
    const long long b = blockIdx.x;
    int lo = 0, hi = entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[6 * (long long)mid + 5] <= b) lo = mid; else hi = mid - 1;
    }
    const long long* row = table + 6 * (long long)lo;
    pack_tile_weights_body(reinterpret_cast<const float*>(row[0]), reinterpret_cast<bf16_t*>(row[1]), (int)row[2], (int)row[3],
                           (int)row[4] & 1, ((int)row[4] >> 1) & 1, (int)(b - row[5]), 1 << 30, tile);
}

int check(const ur_conv_desc* d, bool dgrad) {
    UR_REQUIRE(d != nullptr, "conv3x3_tile: null descriptor");
    UR_REQUIRE(d->dtype == UR_BF16, "conv3x3_tile: bf16 only");
    UR_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->oh == d->h && d->ow == d->w,
               "conv3x3_tile: 3x3 stride 1 pad 1 only");
    const int rows = dgrad ? d->cin : d->cout, red = dgrad ? d->cout : d->cin;
    UR_REQUIRE(rows % 32 == 0 && red % (2 * KC) == 0, "conv3x3_tile: needs 32 | produced channels and %d | reduced channels",
               2 * KC);
    UR_REQUIRE(d->in_ld % 8 == 0 && d->out_ld % 8 == 0, "conv3x3_tile: pitches must be multiples of 8");
    UR_REQUIRE((long long)d->n * d->h * d->w * (d->in_ld > d->out_ld ? d->in_ld : d->out_ld) < (1LL << 31),
               "conv3x3_tile: tensor exceeds 2^31 elements");
    UR_REQUIRE((long long)d->w * (d->in_ld > d->out_ld ? d->in_ld : d->out_ld) < (1LL << 24),
               "conv3x3_tile: an image row exceeds 2^24 elements");
    return UR_OK;
}

// Rows per wave.  16-row tiles halve the LDS traffic per MFMA; 8-row tiles are kept for images too small to
// fill 256 CUs x 2 workgroups with the larger tile.  URESNET_TILE_ROWS=2|4 overrides (testing).
int pick_rows(const ur_conv_desc* d, bool dgrad) {
    static const int forced = [] { const char* e = getenv("URESNET_TILE_ROWS"); return e ? atoi(e) : 0; }();
    if (forced == 2 || forced == 4) return forced;
    const long long wgs4 = (long long)d->n * ur_cdiv(d->h, 16) * ur_cdiv(d->w, TW) * ur_cdiv(dgrad ? d->cin : d->cout, BN);
    // measured per layer: 16-row tiles win as soon as they fill one resident round (512 workgroups), even when the
    // last round is half empty (768 workgroups: 0.37 vs 0.40 ms on 512 channels at 64x128); below that the smaller
    // tile's extra workgroups win (384 vs 768 workgroups: 0.21 vs 0.20 ms on 1024 channels at 32x64)
    return wgs4 >= 512 ? 4 : 2;
}

struct BnSums {          // the BatchNorm whose output gradient a backward-data launch produces (EPI 3 / 4)
    const void* y; int ld_y; const void* z; const float *mean, *invstd, *gamma, *beta;
};

int launch(const void* src, const void* wpk, void* dst, float* stats, const ur_conv_desc* d, bool dgrad, hipStream_t st,
           const void* addend = nullptr, int add_ld = 0, const float* scale = nullptr, const float* shift = nullptr,
           int relu = 0, const BnSums* bn = nullptr) {
    const int R = pick_rows(d, dgrad);
    const int TH = 4 * R;
    TileParams p{};
    p.x = (const bf16_t*)src; p.wpk = (const bf16_t*)wpk; p.y = (bf16_t*)dst; p.stats = stats;
    p.addend = (const bf16_t*)addend; p.ld_add = add_ld;
    p.scale = scale; p.shift = shift; p.relu = relu;
    p.N = d->n; p.H = d->h; p.W = d->w;
    p.Cin = dgrad ? d->cout : d->cin; p.Cout = dgrad ? d->cin : d->cout;
    p.ld_in = dgrad ? d->out_ld : d->in_ld; p.ld_out = dgrad ? d->in_ld : d->out_ld;
    p.tiles_x = ur_cdiv(d->w, TW); p.tiles_y = ur_cdiv(d->h, TH);
    p.tiles = d->n * p.tiles_x * p.tiles_y;
    p.nblocks = ur_cdiv(p.Cout, BN);
    const int grid = ((p.tiles + 7) / 8 * 8) * p.nblocks;
    int epi = scale ? 2 : (addend ? 1 : 0);
    if (bn) {
        p.bn_y = (const bf16_t*)bn->y; p.ld_bny = bn->ld_y; p.bn_z = (const bf16_t*)bn->z;
        p.bn_mean = bn->mean; p.bn_invstd = bn->invstd; p.bn_gamma = bn->gamma; p.bn_beta = bn->beta;
        epi = (bn->z ? 5 : 3) + (addend ? 1 : 0);
    }
#define UR_TILE_LAUNCH(RR, EE) hipLaunchKernelGGL((conv3x3_tile_kernel<RR, EE>), dim3(grid), dim3(NTHREADS), 0, st, p)
#define UR_TILE_PICK(RR) switch (epi) { case 0: UR_TILE_LAUNCH(RR, 0); break; case 1: UR_TILE_LAUNCH(RR, 1); break; \
                                         case 2: UR_TILE_LAUNCH(RR, 2); break; case 3: UR_TILE_LAUNCH(RR, 3); break; \
                                         case 4: UR_TILE_LAUNCH(RR, 4); break; case 5: UR_TILE_LAUNCH(RR, 5); break; \
                                         default: UR_TILE_LAUNCH(RR, 6); }
    if (R == 4) { UR_TILE_PICK(4) } else { UR_TILE_PICK(2) }
#undef UR_TILE_PICK
#undef UR_TILE_LAUNCH
    return ur_check_launch("conv3x3_tile");
}

}  // namespace

// One layer's image in either piece order (shared with conv3x3_wide.hip, whose caller has checked the shape).
int ur_pack3x3_image(const float* w, void* packed, int cin, int cout, int dgrad, int wide, hipStream_t s) {
    const long long total = (long long)cin * cout * 9;
    hipLaunchKernelGGL(pack_tile_weights_kernel, dim3(ur_grid_1d(total, PACK_BLOCK, 2048)), dim3(256), 0, s, w, (bf16_t*)packed, cin,
                       cout, dgrad, wide);
    return ur_check_launch("pack_tile_weights");
}

extern "C" {

int ur_conv3x3_tile_supported(const ur_conv_desc* d, int dgrad) {
    if (!d || d->dtype != UR_BF16) return 0;
    if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->oh == d->h && d->ow == d->w)) return 0;
    const int rows = dgrad ? d->cin : d->cout, red = dgrad ? d->cout : d->cin;
    if (rows % 32 != 0 || red % (2 * KC) != 0 || d->in_ld % 8 != 0 || d->out_ld % 8 != 0) return 0;
    if ((long long)d->n * d->h * d->w * (d->in_ld > d->out_ld ? d->in_ld : d->out_ld) >= (1LL << 31)) return 0;
    if ((long long)d->w * (d->in_ld > d->out_ld ? d->in_ld : d->out_ld) >= (1LL << 24)) return 0;
    return 1;
}

// either direction: the produced channels are padded to a multiple of 64
size_t ur_conv3x3_tile_packed_elems(const ur_conv_desc* d) {
    return d ? (size_t)ur_cdiv(d->cin, BN) * BN * ur_cdiv(d->cout, BN) * BN * 9 : 0;
}

int ur_conv3x3_tile_pack_weight(const float* w, void* packed, const ur_conv_desc* d, int dgrad, ur_stream_t s) {
    if (int e = check(d, dgrad != 0)) return e;
    UR_REQUIRE(w && packed, "conv3x3_tile_pack_weight: null pointer");
    return ur_pack3x3_image(w, packed, d->cin, d->cout, dgrad != 0, 0, (hipStream_t)s);
}

int ur_conv3x3_tile_pack_weights_multi(const long long* table, int entries, long long total_blocks, ur_stream_t s) {
    UR_REQUIRE(table && entries > 0 && total_blocks > 0 && total_blocks < (1LL << 31),
               "conv3x3_tile_pack_weights_multi: bad arguments");
    hipLaunchKernelGGL(pack_tile_weights_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)s, table, entries);
    return ur_check_launch("pack_tile_weights_multi");
}

// blocks of one layer's image in one direction (the unit of the multi-layer launch)
long long ur_conv3x3_tile_pack_blocks(int cin, int cout, int dgrad) {
    const int rows = dgrad ? cin : cout, red = dgrad ? cout : cin;
    return (long long)ur_cdiv(rows, BN) * (red / KC);
}

int ur_conv3x3_tile_stats_rows(const ur_conv_desc* d) {
    return d ? d->n * ur_cdiv(d->h, 4 * pick_rows(d, false)) * ur_cdiv(d->w, TW) : 0;
}

int ur_conv3x3_tile_dgrad_stats_rows(const ur_conv_desc* d) {
    return d ? d->n * ur_cdiv(d->h, 4 * pick_rows(d, true)) * ur_cdiv(d->w, TW) : 0;
}

int ur_conv3x3_tile_dgrad_bnsums(const void* dy, const void* w_packed_dgrad, void* dx, const void* addend, int add_ld,
                                 const void* bn_y, int bn_y_ld, const void* bn_z, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, float* partial, const ur_conv_desc* d, ur_stream_t s) {
    if (int e = check(d, true)) return e;
    UR_REQUIRE(dy && w_packed_dgrad && dx && bn_y && mean && invstd && gamma && beta && partial,
               "conv3x3_tile_dgrad_bnsums: null pointer");
    if (bn_y_ld == 0) bn_y_ld = d->cin;
    UR_REQUIRE(bn_y_ld >= d->cin && bn_y_ld % 8 == 0 && ((uintptr_t)bn_y % 16) == 0 && (!bn_z || ((uintptr_t)bn_z % 16) == 0),
               "conv3x3_tile_dgrad_bnsums: BatchNorm tensors must be 16-byte aligned with a pitch that is a multiple of 8");
    UR_REQUIRE((long long)d->n * d->h * d->w * bn_y_ld < (1LL << 31), "conv3x3_tile_dgrad_bnsums: tensor exceeds 2^31 elements");
    if (addend) {
        if (add_ld == 0) add_ld = d->cin;
        UR_REQUIRE(add_ld >= d->cin && add_ld % 8 == 0 && ((uintptr_t)addend % 16) == 0,
                   "conv3x3_tile_dgrad_bnsums: addend pitch %d must be a multiple of 8 >= %d and the pointer 16-byte aligned", add_ld,
                   d->cin);
        UR_REQUIRE((long long)d->n * d->h * d->w * add_ld < (1LL << 31), "conv3x3_tile_dgrad_bnsums: addend exceeds 2^31 elements");
    }
    const BnSums bn{bn_y, bn_y_ld, bn_z, mean, invstd, gamma, beta};
    return launch(dy, w_packed_dgrad, dx, partial, d, true, (hipStream_t)s, addend, add_ld, nullptr, nullptr, 0, &bn);
}

int ur_conv3x3_tile_fwd(const void* x, const void* w_packed, void* y, float* stats, const ur_conv_desc* d, ur_stream_t s) {
    if (int e = check(d, false)) return e;
    UR_REQUIRE(x && w_packed && y, "conv3x3_tile_fwd: null pointer");
    return launch(x, w_packed, y, stats, d, false, (hipStream_t)s);
}

int ur_conv3x3_tile_dgrad(const void* dy, const void* w_packed_dgrad, void* dx, const ur_conv_desc* d, ur_stream_t s) {
    if (int e = check(d, true)) return e;
    UR_REQUIRE(dy && w_packed_dgrad && dx, "conv3x3_tile_dgrad: null pointer");
    return launch(dy, w_packed_dgrad, dx, nullptr, d, true, (hipStream_t)s);
}

int ur_conv3x3_tile_fwd_affine(const void* x, const void* w_packed, void* y, const float* scale, const float* shift, int relu,
                               const void* residual, int res_ld, const ur_conv_desc* d, ur_stream_t s) {
    if (int e = check(d, false)) return e;
    UR_REQUIRE(x && w_packed && y && scale && shift, "conv3x3_tile_fwd_affine: null pointer");
    if (residual) {
        if (res_ld == 0) res_ld = d->cout;
        UR_REQUIRE(res_ld >= d->cout && res_ld % 8 == 0 && ((uintptr_t)residual % 16) == 0,
                   "conv3x3_tile_fwd_affine: residual pitch %d must be a multiple of 8 >= %d and the pointer 16-byte aligned",
                   res_ld, d->cout);
        UR_REQUIRE((long long)d->n * d->h * d->w * res_ld < (1LL << 31), "conv3x3_tile_fwd_affine: residual exceeds 2^31 elements");
    }
    return launch(x, w_packed, y, nullptr, d, false, (hipStream_t)s, residual, res_ld, scale, shift, relu != 0);
}

int ur_conv3x3_tile_dgrad_add(const void* dy, const void* w_packed_dgrad, void* dx, const void* addend, int add_ld,
                              const ur_conv_desc* d, ur_stream_t s) {
    if (int e = check(d, true)) return e;
    UR_REQUIRE(dy && w_packed_dgrad && dx && addend, "conv3x3_tile_dgrad_add: null pointer");
    if (add_ld == 0) add_ld = d->cin;
    UR_REQUIRE(add_ld >= d->cin && add_ld % 8 == 0 && ((uintptr_t)addend % 16) == 0,
               "conv3x3_tile_dgrad_add: addend pitch %d must be a multiple of 8 >= %d and the pointer 16-byte aligned", add_ld, d->cin);
    UR_REQUIRE((long long)d->n * d->h * d->w * add_ld < (1LL << 31), "conv3x3_tile_dgrad_add: addend exceeds 2^31 elements");
    return launch(dy, w_packed_dgrad, dx, nullptr, d, true, (hipStream_t)s, addend, add_ld);
}

}  // extern "C"
