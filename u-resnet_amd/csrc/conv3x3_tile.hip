// 3x3 / stride 1 / pad 1 convolution, bf16 NHWC, as an LDS-tiled implicit GEMM on the bf16 MFMA.
// Used for the forward pass and (with tap-flipped, transposed weights) for backward-data of the
// layers that carry ~94 % of U-ResNet's FLOPs.
//
// Workgroup (4 waves) -> 4R x 32 output pixels x 64 output channels, R = 4 (R = 2 for small images).
// The reduction runs over 16-channel slices of the input.  Per slice the workgroup stages
//   * the (4R+2) x 34 pixel halo, 32 bytes per pixel; the two 16-byte pieces of a pixel are swapped in every
//     second group of 8 columns, which makes the 32-consecutive-pixel fragment reads conflict-free and keeps
//     every row at a constant byte distance (fragment address = per-lane base + immediate),
//   * the 9 x 64 x 16 weight slice, pre-packed in MFMA-fragment order and copied by LDS-DMA,
// into one of two LDS stages: loads for slice c+1 are issued before the MFMAs of slice c and the halo is
// written to the other stage behind two thirds of them; one barrier per slice.
// Wave w owns output rows [wR, wR+R).  For one horizontal tap offset it reads R+2 halo-row fragments and
// reuses each for the three vertical taps, and each of the 6 weight fragments for R rows:
// (R+2) + 6 LDS reads feed 6R v_mfma_f32_32x32x16_bf16 (0.5 reads per MFMA at R = 4).
// LDS reads are issued by hand (inline assembly, counted s_waitcnt): weight fragments two steps ahead, the row
// fragments of the next horizontal offset five steps ahead.  Left to itself the compiler sinks every read
// down to its first use and each group of four MFMAs then waits out a full LDS round trip.
// The epilogue stages the wave's tile in LDS and writes 16-byte pieces (full 128-byte pixel rows).
#include "ur_common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

constexpr int TW = 32, BN = 64, KC = 16, NTHREADS = 256;
constexpr int HALO_W = TW + 2;
constexpr int ROW_BYTES = HALO_W * 32;               // one halo row of one slice
constexpr int W_PIECES = 9 * 2 * 64;                 // 16-byte pieces of one weight slice
constexpr int W_GROUPS = W_PIECES / 64;              // one group = one wave-wide 16-byte LDS-DMA (1 KiB)
constexpr int W_GROUP_ITERS = (W_GROUPS + 3) / 4;
constexpr int W_BYTES = W_PIECES * 16;               // 18432
constexpr int OUT_PITCH = 144;                       // bytes per staged output pixel (128 + 16: the two half-waves land on disjoint banks)
constexpr int STEPS = 18;
constexpr int COMMIT_STEP = 16;                      // step ahead of which the next slice's halo is written (loads had 16 steps to land)

template <int R> struct Geo {
    static constexpr int TH = 4 * R;
    static constexpr int HALO_PIX = (TH + 2) * HALO_W;
    static constexpr int HALO_PIECES = HALO_PIX * 2;
    static constexpr int HALO_ITERS = (HALO_PIECES + NTHREADS - 1) / NTHREADS;
    static constexpr int HALO_BYTES = HALO_PIX * 32;
    static constexpr int STAGE_BYTES = HALO_BYTES + W_BYTES;
    static constexpr int OUT_BYTES = 4 * R * TW * OUT_PITCH;       // epilogue staging, all four waves
    static constexpr int RED_BYTES = 4 * BN * 2 * 4;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES > OUT_BYTES + RED_BYTES ? 2 * STAGE_BYTES : OUT_BYTES + RED_BYTES;
};

struct TileParams {
    const bf16_t* x;      // [N][H][W][ld_in]
    const bf16_t* wpk;    // [Cout/64][Cin/16][9][2][64][8]
    bf16_t* y;            // [N][H][W][ld_out]
    float* stats;         // [tiles][Cout][2] or null
    int N, H, W, Cin, Cout;
    int ld_in, ld_out;
    int tiles_x, tiles_y, tiles;     // spatial tiles per image row / column / total (N * ty * tx)
    int nblocks;                     // Cout / 64
};

__device__ __attribute__((aligned(16))) const unsigned int tile_zero_piece[4] = {0u, 0u, 0u, 0u};

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

// LDS operations a wave issues in step t of a slice, in issue order: halo writes, weight fragment t+2, row fragments
constexpr int ops_write(int t, bool more, int nw) { return (more && t == COMMIT_STEP) ? nw : 0; }
constexpr int ops_b(int t) { return t + 2 < STEPS ? 1 : 0; }
constexpr int ops_a(int t, int na) { return (t % 6 == 1 && t / 6 < 2) ? na : 0; }
constexpr int ops_all(int t, bool more, int nw, int na) { return ops_write(t, more, nw) + ops_b(t) + ops_a(t, na); }
// operations issued after weight fragment s (issued in step s-2, or ahead of the loop for s < 2) up to the
// MFMAs of step s: the lgkmcnt value at which fragment s is known to have arrived
constexpr int younger_than_b(int s, bool more, int nw, int na) {
    return s == 0 ? 1 + ops_all(0, more, nw, na)
         : s == 1 ? ops_all(0, more, nw, na) + ops_all(1, more, nw, na)
                  : ops_a(s - 2, na) + ops_all(s - 1, more, nw, na) + ops_all(s, more, nw, na);
}

__device__ __forceinline__ int halo_lds_offset(int hy, int hx, int piece) {
    return hy * ROW_BYTES + hx * 32 + ((piece ^ ((hx >> 3) & 1)) << 4);
}

template <int R>
__global__ __launch_bounds__(NTHREADS, 2) void conv3x3_tile_kernel(TileParams p) {
    using G = Geo<R>;
    constexpr int NA = R + 2, NW = G::HALO_ITERS;
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
    // Branch-free: surplus threads of the last round repeat the last piece (same value to the same address);
    // padding pixels read a zero block, so no arithmetic touches the loaded registers before the LDS write.
    int h_goff[NW];                  // element offset of the piece's pixel (+ 8 * piece), -1 = zero block
    unsigned h_loff[NW];             // LDS byte address inside stage 0
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        int i = tid + j * NTHREADS;
        i = i < G::HALO_PIECES ? i : G::HALO_PIECES - 1;
        const int pix = i >> 1, piece = i & 1;
        const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        h_loff[j] = lds0 + halo_lds_offset(hy, hx, piece);
        h_goff[j] = -1;
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
            h_goff[j] = ((img * p.H + gy) * p.W + gx) * p.ld_in + piece * 8;     // < 2^31 (checked on the host)
    }
    const int chunks = p.Cin / KC;
    // weights: 18 groups of 64 pieces (1 KiB); wave w copies groups w, w+4, ... straight into LDS
    const bf16x8_t* wsrc = reinterpret_cast<const bf16x8_t*>(p.wpk) + (size_t)nb * chunks * W_PIECES + lane;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    bf16x8_t ph[NW];
    auto prefetch = [&](int chunk, unsigned stage_off) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const bf16_t* src = h_goff[j] >= 0 ? p.x + (size_t)h_goff[j] + chunk * KC
                                               : reinterpret_cast<const bf16_t*>(tile_zero_piece);
            ph[j] = *reinterpret_cast<const bf16x8_t*>(src);
        }
        const bf16x8_t* ws = wsrc + (size_t)chunk * W_PIECES;
        const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + stage_off) + G::HALO_BYTES;
#pragma unroll
        for (int j = 0; j < W_GROUP_ITERS; ++j) {
            // the two surplus slots of the last round repeat group 17 (same bytes, same place)
            int g = wave_u + 4 * j;
            g = g < W_GROUPS ? g : W_GROUPS - 1;
            lds_dma16(ws + g * 64, wdst + g * 1024);
        }
    };
    auto commit = [&](unsigned stage_off) {
#pragma unroll
        for (int j = 0; j < NW; ++j) lds_write16<0>(h_loff[j] + stage_off, ph[j]);
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

    prefetch(0, 0);
    commit(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // one reduction slice; MORE = another slice follows
    auto slice = [&](int chunk, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const unsigned so = (chunk & 1) * G::STAGE_BYTES;          // stage read by this slice
        const unsigned oo = G::STAGE_BYTES - so;                    // the other one: last read before the previous barrier
        if constexpr (MORE) prefetch(chunk + 1, oo);
        const unsigned ab[3] = {a_base[0] + so, a_base[1] + so, a_base[2] + so};
        const unsigned bb = b_base + so;
        bf16x8_t af[2][NA], bq[3];
        static_for<NA>([&](auto i) { lds_read16<i * ROW_BYTES>(af[0][i], ab[0]); });
        lds_read16<0>(bq[0], bb);
        lds_read16<1024>(bq[1], bb);
        static_for<STEPS>([&](auto S) {
            constexpr int s = S, dx = s / 6, dy = (s % 6) / 2, b = s & 1;
            if constexpr (MORE && s == COMMIT_STEP) commit(oo);
            if constexpr (s + 2 < STEPS) {
                constexpr int s2 = s + 2, tap2 = ((s2 % 6) / 2) * 3 + s2 / 6;
                lds_read16<(tap2 * 2 + (s2 & 1)) * 1024>(bq[s2 % 3], bb);
            }
            if constexpr (s % 6 == 1 && dx < 2)
                static_for<NA>([&](auto i) { lds_read16<i * ROW_BYTES>(af[(dx + 1) & 1][i], ab[dx + 1]); });
            lds_wait<younger_than_b(s, MORE, NW, NA)>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < R; ++a)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[dx & 1][a + dy], bq[s % 3], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        // this wave's LDS-DMA pieces and halo writes have landed; then everybody's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int chunk = 0; chunk + 1 < chunks; ++chunk) slice(chunk, std::true_type{});
    slice(chunks - 1, std::false_type{});

    // ---- epilogue: per-channel partial statistics of the fp32 results, bf16 tile via LDS -> 16-byte stores ----
    unsigned char* ostage = lds + wave * (R * TW * OUT_PITCH);
    float* red = reinterpret_cast<float*>(lds + G::OUT_BYTES);
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < R; ++a) {
        const int oy = ty0 + wave * R + a;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v = acc[a][b][r];
                *reinterpret_cast<bf16_t*>(ostage + (a * TW + px) * OUT_PITCH + (b * 32 + lx) * 2) = (bf16_t)v;
                if (oy < p.H && tx0 + px < p.W) { s1[b] += v; s2[b] += v * v; }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < R * 4; ++i) {
        const int idx = lane + 64 * i;
        const int pxl = idx >> 3, piece = idx & 7;
        const int oy = ty0 + wave * R + (pxl >> 5), ox = tx0 + (pxl & 31);
        if (oy < p.H && ox < p.W)
            *reinterpret_cast<float4*>(p.y + ((size_t)(img * p.H + oy) * p.W + ox) * p.ld_out + nb * BN + piece * 8) =
                *reinterpret_cast<const float4*>(ostage + pxl * OUT_PITCH + piece * 16);
    }
    if (p.stats) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float t1 = s1[b] + __shfl_xor(s1[b], 32, 64);
            const float t2 = s2[b] + __shfl_xor(s2[b], 32, 64);
            if (lh == 0) { red[(wave * BN + b * 32 + lx) * 2] = t1; red[(wave * BN + b * 32 + lx) * 2 + 1] = t2; }
        }
        __syncthreads();
        if (tid < BN) {
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
__global__ void pack_tile_weights_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst, int cin, int cout,
                                         int dgrad) {
    const int rows = dgrad ? cin : cout;      // channels produced by the kernel
    const int red = dgrad ? cout : cin;       // channels reduced over
    const int chunks = red / KC;
    const long long total = (long long)(rows / BN) * chunks * W_PIECES * 8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        long long t = i >> 9;
        const int b = (int)(t & 1); t >>= 1;
        const int tap = (int)(t % 9); t /= 9;
        const int chunk = (int)(t % chunks);
        const int nb = (int)(t / chunks);
        const int n = nb * BN + b * 32 + (lane & 31);
        const int k = chunk * KC + 8 * (lane >> 5) + j;
        float v;
        if (!dgrad) v = w[((size_t)n * cin + k) * 9 + tap];
        else v = w[((size_t)k * cin + n) * 9 + (8 - tap)];
        dst[i] = (bf16_t)v;
    }
}

int check(const ur_conv_desc* d, bool dgrad) {
    UR_REQUIRE(d != nullptr, "conv3x3_tile: null descriptor");
    UR_REQUIRE(d->dtype == UR_BF16, "conv3x3_tile: bf16 only");
    UR_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->oh == d->h && d->ow == d->w,
               "conv3x3_tile: 3x3 stride 1 pad 1 only");
    const int rows = dgrad ? d->cin : d->cout, red = dgrad ? d->cout : d->cin;
    UR_REQUIRE(rows % BN == 0 && red % KC == 0, "conv3x3_tile: needs %d | produced channels and %d | reduced channels",
               BN, KC);
    UR_REQUIRE(d->in_ld % 8 == 0 && d->out_ld % 8 == 0, "conv3x3_tile: pitches must be multiples of 8");
    UR_REQUIRE((long long)d->n * d->h * d->w * (d->in_ld > d->out_ld ? d->in_ld : d->out_ld) < (1LL << 31),
               "conv3x3_tile: tensor exceeds 2^31 elements");
    return UR_OK;
}

// Rows per wave.  16-row tiles halve the LDS traffic per MFMA; 8-row tiles are kept for images too small to
// fill 256 CUs x 2 workgroups with the larger tile.  URESNET_TILE_ROWS=2|4 overrides (testing).
int pick_rows(const ur_conv_desc* d, bool dgrad) {
    static const int forced = [] { const char* e = getenv("URESNET_TILE_ROWS"); return e ? atoi(e) : 0; }();
    if (forced == 2 || forced == 4) return forced;
    const long long wgs4 = (long long)d->n * ur_cdiv(d->h, 16) * ur_cdiv(d->w, TW) * ((dgrad ? d->cin : d->cout) / BN);
    return (wgs4 >= 1024 || wgs4 % 512 == 0) ? 4 : 2;
}

int launch(const void* src, const void* wpk, void* dst, float* stats, const ur_conv_desc* d, bool dgrad, hipStream_t st) {
    const int R = pick_rows(d, dgrad);
    const int TH = 4 * R;
    TileParams p{};
    p.x = (const bf16_t*)src; p.wpk = (const bf16_t*)wpk; p.y = (bf16_t*)dst; p.stats = stats;
    p.N = d->n; p.H = d->h; p.W = d->w;
    p.Cin = dgrad ? d->cout : d->cin; p.Cout = dgrad ? d->cin : d->cout;
    p.ld_in = dgrad ? d->out_ld : d->in_ld; p.ld_out = dgrad ? d->in_ld : d->out_ld;
    p.tiles_x = ur_cdiv(d->w, TW); p.tiles_y = ur_cdiv(d->h, TH);
    p.tiles = d->n * p.tiles_x * p.tiles_y;
    p.nblocks = p.Cout / BN;
    const int grid = ((p.tiles + 7) / 8 * 8) * p.nblocks;
    if (R == 4) hipLaunchKernelGGL(conv3x3_tile_kernel<4>, dim3(grid), dim3(NTHREADS), 0, st, p);
    else hipLaunchKernelGGL(conv3x3_tile_kernel<2>, dim3(grid), dim3(NTHREADS), 0, st, p);
    return ur_check_launch("conv3x3_tile");
}

}  // namespace

extern "C" {

int ur_conv3x3_tile_supported(const ur_conv_desc* d, int dgrad) {
    if (!d || d->dtype != UR_BF16) return 0;
    if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->oh == d->h && d->ow == d->w)) return 0;
    const int rows = dgrad ? d->cin : d->cout, red = dgrad ? d->cout : d->cin;
    if (rows % BN != 0 || red % KC != 0 || d->in_ld % 8 != 0 || d->out_ld % 8 != 0) return 0;
    if ((long long)d->n * d->h * d->w * (d->in_ld > d->out_ld ? d->in_ld : d->out_ld) >= (1LL << 31)) return 0;
    return 1;
}

size_t ur_conv3x3_tile_packed_elems(const ur_conv_desc* d) { return d ? (size_t)d->cin * d->cout * 9 : 0; }

int ur_conv3x3_tile_pack_weight(const float* w, void* packed, const ur_conv_desc* d, int dgrad, ur_stream_t s) {
    if (int e = check(d, dgrad != 0)) return e;
    UR_REQUIRE(w && packed, "conv3x3_tile_pack_weight: null pointer");
    const long long total = (long long)d->cin * d->cout * 9;
    hipLaunchKernelGGL(pack_tile_weights_kernel, dim3(ur_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)s, w,
                       (bf16_t*)packed, d->cin, d->cout, dgrad != 0);
    return ur_check_launch("pack_tile_weights");
}

int ur_conv3x3_tile_stats_rows(const ur_conv_desc* d) {
    return d ? d->n * ur_cdiv(d->h, 4 * pick_rows(d, false)) * ur_cdiv(d->w, TW) : 0;
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

}  // extern "C"
