// 3x3 / stride 1 / pad 1 convolution, bf16 NHWC, as an LDS-tiled implicit GEMM on the bf16 MFMA.
// Used for the forward pass and (with tap-flipped, transposed weights) for backward-data of the
// Cin,Cout >= 64 layers that carry ~94 % of U-ResNet's FLOPs.
//
// Workgroup (4 waves) -> 8 x 32 output pixels x 64 output channels.  Per 32-channel slice of the
// input it stages
//   * the (8+2) x (32+2) pixel halo once in LDS (each input element then feeds 9 taps x 64 channels),
//     16-byte pieces XOR-swizzled by pixel so the 32-consecutive-pixel fragment reads are conflict-free,
//   * the 9 x 64 x 32 weight slice, pre-packed on the host side in MFMA-fragment order so the LDS image
//     is a straight copy and every fragment read is base + lane * 16.
// Wave w owns output rows 2w, 2w+1 (2 x 2 accumulator tiles of 32 pixels x 32 channels); per slice it
// issues 9 taps x 2 k-steps x 4 = 72 v_mfma_f32_32x32x16_bf16.  The next slice is prefetched into
// registers while the current one is multiplied; two workgroups per CU (58 KB LDS each) overlap one
// group's LDS refill with the other's MFMA work.
#include "ur_common.h"

namespace {

constexpr int TH = 8, TW = 32, BN = 64, KC = 32, NTHREADS = 256;
constexpr int HALO_H = TH + 2, HALO_W = TW + 2, HALO_PIX = HALO_H * HALO_W;   // 340
constexpr int HALO_PIECES = HALO_PIX * 4;                                        // 16-byte pieces (8 channels)
constexpr int HALO_ITERS = (HALO_PIECES + NTHREADS - 1) / NTHREADS;              // 6
constexpr int W_PIECES = 9 * 2 * 2 * 64;                                         // 2304 pieces of 16 bytes
constexpr int W_ITERS = W_PIECES / NTHREADS;                                     // 9
constexpr int HALO_BYTES = HALO_PIX * 64;                                        // 21760
constexpr int W_BYTES = W_PIECES * 16;                                           // 36864

struct TileParams {
    const bf16_t* x;      // [N][H][W][ld_in]
    const bf16_t* wpk;    // [Cout/64][Cin/32][9][2][2][64][8]
    bf16_t* y;            // [N][H][W][ld_out]
    float* stats;         // [tiles][Cout][2] or null
    int N, H, W, Cin, Cout;
    int ld_in, ld_out;
    int tiles_x, tiles_y, tiles;     // spatial tiles per image row / column / total (N * ty * tx)
    int nblocks;                     // Cout / 64
};

__device__ __forceinline__ int halo_lds_offset(int pix, int piece) {
    return pix * 64 + ((piece ^ ((pix >> 2) & 3)) << 4);
}

__global__ __launch_bounds__(NTHREADS, 2) void conv3x3_tile_kernel(TileParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[HALO_BYTES + W_BYTES];
    __shared__ float red[4][BN][2];
    unsigned char* halo = lds;
    unsigned char* wl = lds + HALO_BYTES;

    // XCD-aware order: ids congruent mod 8 share an L2; they walk the channel blocks of one spatial tile
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int tile = (slot / p.nblocks) * 8 + xcd;
    const int nb = slot % p.nblocks;
    if (tile >= p.tiles) return;
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int trem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty0 = (trem / p.tiles_x) * TH, tx0 = (trem % p.tiles_x) * TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- staging bookkeeping: halo pieces this thread owns (fixed for the whole channel loop) -------
    int h_goff[HALO_ITERS];          // element offset of the piece's pixel (+ 8 * piece), -1 = zero fill
    int h_loff[HALO_ITERS];          // LDS byte offset, -1 = no piece
#pragma unroll
    for (int j = 0; j < HALO_ITERS; ++j) {
        const int i = tid + j * NTHREADS;
        h_goff[j] = -1; h_loff[j] = -1;
        if (i < HALO_PIECES) {
            const int pix = i >> 2, piece = i & 3;
            const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
            h_loff[j] = halo_lds_offset(pix, piece);
            if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                h_goff[j] = ((img * p.H + gy) * p.W + gx) * p.ld_in + piece * 8;     // < 2^31 (checked on the host)
        }
    }
    const int chunks = p.Cin / KC;
    const bf16x8_t* wsrc = reinterpret_cast<const bf16x8_t*>(p.wpk) + (size_t)nb * chunks * W_PIECES + tid;

    bf16x8_t ph[HALO_ITERS], pw[W_ITERS];
    auto prefetch = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < HALO_ITERS; ++j) {
            if (h_goff[j] >= 0) ph[j] = *reinterpret_cast<const bf16x8_t*>(p.x + (size_t)h_goff[j] + chunk * KC);
            else ph[j] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
        const bf16x8_t* ws = wsrc + (size_t)chunk * W_PIECES;
#pragma unroll
        for (int j = 0; j < W_ITERS; ++j) pw[j] = ws[j * NTHREADS];
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < HALO_ITERS; ++j)
            if (h_loff[j] >= 0) *reinterpret_cast<bf16x8_t*>(halo + h_loff[j]) = ph[j];
#pragma unroll
        for (int j = 0; j < W_ITERS; ++j) *reinterpret_cast<bf16x8_t*>(wl + (tid + j * NTHREADS) * 16) = pw[j];
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lx = lane & 31, lh = lane >> 5;
    prefetch(0);
    for (int chunk = 0; chunk < chunks; ++chunk) {
        __syncthreads();                    // every wave finished reading the previous slice
        commit();
        __syncthreads();
        if (chunk + 1 < chunks) prefetch(chunk + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t af[2], bfr[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int pix = (wave * 2 + a + dy) * HALO_W + lx + dx;
                    af[a] = *reinterpret_cast<const bf16x8_t*>(halo + halo_lds_offset(pix, ks * 2 + lh));
                }
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    bfr[b] = *reinterpret_cast<const bf16x8_t*>(wl + (((tap * 2 + ks) * 2 + b) * 64 + lane) * 16);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: store bf16, per-channel partial statistics of the fp32 results --------------------
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = ty0 + wave * 2 + a;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int co = nb * BN + b * 32 + lx;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = tx0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (oy < p.H && ox < p.W) {
                    const float v = acc[a][b][r];
                    p.y[((size_t)(img * p.H + oy) * p.W + ox) * p.ld_out + co] = (bf16_t)v;
                    s1[b] += v; s2[b] += v * v;
                }
            }
        }
    }
    if (p.stats) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float t1 = s1[b] + __shfl_xor(s1[b], 32, 64);
            const float t2 = s2[b] + __shfl_xor(s2[b], 32, 64);
            if (lh == 0) { red[wave][b * 32 + lx][0] = t1; red[wave][b * 32 + lx][1] = t2; }
        }
        __syncthreads();
        if (tid < BN) {
            float* o = p.stats + ((size_t)tile * p.Cout + nb * BN + tid) * 2;
            o[0] = red[0][tid][0] + red[1][tid][0] + red[2][tid][0] + red[3][tid][0];
            o[1] = red[0][tid][1] + red[1][tid][1] + red[2][tid][1] + red[3][tid][1];
        }
    }
}

// Fragment-order weight image.  piece = (((nb*chunks + chunk)*9 + tap)*2 + ks)*2 + b, then lane, then 8 k-values.
//   forward : value = w[co = nb*64 + b*32 + (lane&31)][ci = chunk*32 + ks*16 + 8*(lane>>5) + j][tap]
//   dgrad   : the kernel convolves dy (channels = cout of the layer) into dx (channels = cin):
//             value = w[co = chunk*32 + ks*16 + 8*(lane>>5) + j][ci = nb*64 + b*32 + (lane&31)][8 - tap]
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
        const int ks = (int)(t & 1); t >>= 1;
        const int tap = (int)(t % 9); t /= 9;
        const int chunk = (int)(t % chunks);
        const int nb = (int)(t / chunks);
        const int n = nb * BN + b * 32 + (lane & 31);
        const int k = chunk * KC + ks * 16 + 8 * (lane >> 5) + j;
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

int launch(const void* src, const void* wpk, void* dst, float* stats, const ur_conv_desc* d, bool dgrad, hipStream_t st) {
    TileParams p{};
    p.x = (const bf16_t*)src; p.wpk = (const bf16_t*)wpk; p.y = (bf16_t*)dst; p.stats = stats;
    p.N = d->n; p.H = d->h; p.W = d->w;
    p.Cin = dgrad ? d->cout : d->cin; p.Cout = dgrad ? d->cin : d->cout;
    p.ld_in = dgrad ? d->out_ld : d->in_ld; p.ld_out = dgrad ? d->in_ld : d->out_ld;
    p.tiles_x = ur_cdiv(d->w, TW); p.tiles_y = ur_cdiv(d->h, TH);
    p.tiles = d->n * p.tiles_x * p.tiles_y;
    p.nblocks = p.Cout / BN;
    const int grid = ((p.tiles + 7) / 8 * 8) * p.nblocks;
    hipLaunchKernelGGL(conv3x3_tile_kernel, dim3(grid), dim3(NTHREADS), 0, st, p);
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

int ur_conv3x3_tile_stats_rows(const ur_conv_desc* d) { return d ? d->n * ur_cdiv(d->h, TH) * ur_cdiv(d->w, TW) : 0; }

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
