// ConvTranspose2d(kernel 2, stride 2) forward and backward-data as a row GEMM whose operands go from memory to LDS by
// LDS-DMA (the generic gather kernel stages them through registers, two barriers per 32-deep slice, and is bound by its
// LDS traffic at about 500 TFLOP/s).
//
//   forward      : D[m][(q, co)] = sum_ci x[m][ci] * w[ci][co][q] + bias[co]      row m = low-resolution pixel (img, y, x),
//                  written to pixel (2y + q / 2, 2x + q % 2) of the output (possibly a channel window of a wider tensor)
//   backward-data: dx[m][ci]     = sum_(q, co) dy[pixel (2y + q / 2, 2x + q % 2)][co] * w[ci][co][q]
//
// Both are D[m][n] = sum_k A[m][k] * B[n][k] with B the weight matrix packed by ur_conv2d_pack_weight (modes
// UR_PACK_CONVT_FWD / UR_PACK_CONVT_DGRAD: rows n, k contiguous).  Because kernel == stride, a 32-deep k slice of the
// backward-data operand (and a 64-wide column block of the forward result) lies inside one tap q, so the tap only adds a
// uniform offset to a row address.
//
// Workgroup = 4 waves, tile 256 rows x 128 columns, wave w owns rows [64w, 64w + 64) x all columns (2 x 4 accumulator
// blocks of v_mfma_f32_32x32x16_bf16).  Three LDS stages of one k slice each (A 256 x 64 bytes, B 128 x 64 bytes); a
// wave copies its own 64 A rows and 32 of the B rows, 6 DMA instructions per slice; slice s + 2 is issued when slice s
// starts, one barrier per slice.  LDS rows are 64 bytes = four 16-byte chunks; chunk j of row r sits at position
// j ^ ((r >> 2) & 3), which spreads the 32 rows of a fragment read over all banks; the permutation is applied on the
// source side of the DMA (lane -> which global chunk it fetches), the destination is always base + lane * 16.
#include "ur_common.h"

namespace {

constexpr int GM = 256, GN = 128, GK = 32, NTH = 256, STAGES = 3;
constexpr int A_BYTES = GM * 64, B_BYTES = GN * 64, STAGE_BYTES = A_BYTES + B_BYTES;
constexpr int OUT_PITCH = GN * 2 + 16;            // staged result row: 128 columns + 16 bytes (the two half-waves on disjoint banks)
static_assert(4 * 64 * OUT_PITCH <= STAGES * STAGE_BYTES, "the staged tile must fit the operand stages");

struct GemmParams {
    const bf16_t* a; const bf16_t* b; const float* bias; bf16_t* d;
    int M, N, K;              // K: multiple of 32; N: multiple of 128
    int lda, ldb, ldd;        // element pitches: A pixel rows, B rows (padded K), D pixel rows
    int W;                    // width of the low-resolution tensor: row m = (img * H + y) * W + x, W >= 64
    int cout_t;               // channels per tap of the transposed convolution
    int mtiles, ntiles;
};

typedef __attribute__((address_space(3))) const bf16x8_t* lds_frag_ptr;      // 32-bit LDS address -> ds_read_b128

__device__ __forceinline__ void dma16(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}

// MODE 1: forward (the result rows scatter over the 2x tensor), MODE 2: backward-data (the operand rows gather from it)
template <int MODE>
__global__ __launch_bounds__(NTH, 2) void convt_gemm_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[STAGES * STAGE_BYTES];
    // XCD-aware order: ids congruent mod 8 share an L2; they take the column tiles of one row tile
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int mt = (slot / p.ntiles) * 8 + xcd;
    const int nt = slot % p.ntiles;
    if (mt >= p.mtiles) return;
    const int m0 = mt * GM, n0 = nt * GN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)lds;
    const int W2 = 2 * p.W;

    // ---- DMA sources: 4 A rows and 2 B rows per lane, 32-bit element offsets (checked on the host) ----------------------
    const int gchunk = (lane & 3) ^ ((lane >> 4) & 3);                 // global chunk that lands at position lane & 3 of its row
    int aoff[4], boff[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + 64 * wave + 16 * i + (lane >> 2);
        m = m < p.M ? m : p.M - 1;                                      // rows past the end repeat the last row (never stored)
        if (MODE == 2) {
            const int r = m / p.W, x = m - r * p.W;
            aoff[i] = (2 * r * W2 + 2 * x) * p.lda + gchunk * 8;        // pixel (2y, 2x) of the 2x tensor: 2 (img H + y) rows down
        } else {
            aoff[i] = m * p.lda + gchunk * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) boff[i] = (n0 + 32 * wave + 16 * i + (lane >> 2)) * p.ldb + gchunk * 8;

    // k offset of the next slice to issue: forward 32 s; backward-data (tap offset) + channel inside the tap
    int iss = 0, iss_q = 0, iss_co = 0;
    auto issue = [&]() {
        const unsigned st = lds0 + (iss % STAGES) * STAGE_BYTES;
        int koff = iss * GK;
        if (MODE == 2) koff = ((iss_q >> 1) * W2 + (iss_q & 1)) * p.lda + iss_co;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(p.a + aoff[i] + koff, st + (64 * wave + 16 * i) * 64);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(p.b + boff[i] + iss * GK, st + A_BYTES + (32 * wave + 16 * i) * 64);
        ++iss;
        if (MODE == 2) { iss_co += GK; if (iss_co == p.cout_t) { iss_co = 0; ++iss_q; } }
    };

    // ---- fragment addresses (stage 0): row r of a 32-row block, k chunk (2 ks + kh) at position chunk ^ ((r >> 2) & 3) --
    const int lrow = lane & 31, kh = lane >> 5, sw = (lane >> 2) & 3;
    const unsigned pos[2] = {(unsigned)(((0 + kh) ^ sw) << 4), (unsigned)(((2 + kh) ^ sw) << 4)};
    const unsigned a_base = lds0 + (64 * wave + lrow) * 64, b_base = lds0 + A_BYTES + lrow * 64;

    f32x16_t acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int ktiles = p.K / GK;
    issue();
    if (ktiles > 1) issue();
    for (int s = 0; s < ktiles; ++s) {
        // this wave's copies of slice s have landed (those of slice s + 1 may still be in flight) ...
        if (s + 1 < ktiles) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();        // ... and so have everyone's; nobody reads the stage of slice s - 1 any more
        asm volatile("" ::: "memory");
        if (s + 2 < ktiles) issue();
        const unsigned so = (s % STAGES) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[2], bf[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *(lds_frag_ptr)(size_t)(a_base + so + a * 2048 + pos[ks]);
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = *(lds_frag_ptr)(size_t)(b_base + so + b * 2048 + pos[ks]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
    __syncthreads();                         // the stages become the staging area of the result

    // ---- epilogue: (+ bias), round, stage the wave's 64 x 128 block, write 16-byte pieces --------------------------------
    unsigned char* ostage = lds + wave * (64 * OUT_PITCH);
    // column halves (64 wide) of the tile: forward -> (tap, first channel) of the half; backward-data -> plain columns
    int half_q[2], half_co[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + 64 * h;
        half_q[h] = MODE == 1 ? n / p.cout_t : 0;
        half_co[h] = MODE == 1 ? n - half_q[h] * p.cout_t : n;
    }
    float bv[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) bv[b] = (MODE == 1 && p.bias) ? p.bias[half_co[b >> 1] + (b & 1) * 32 + lrow] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                *reinterpret_cast<bf16_t*>(ostage + row * OUT_PITCH + (b * 32 + lrow) * 2) = (bf16_t)(acc[a][b][r] + bv[b]);
            }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rows of the wave are consecutive low-resolution pixels mw + j; in the 2x tensor pixel (2y, 2x) of row j is
    // P0 + 2j, plus one more row pair (2W pixels... W2 pixels twice) once x runs past the end of the image row (W >= 64: at most once)
    const int mw = m0 + 64 * wave;
    const int rw = mw / p.W, xw = mw - rw * p.W;
    const int P0 = 2 * rw * W2 + 2 * xw;
    const int piece = lane & 15, h = piece >> 3;
    int coloff;
    if (MODE == 1) coloff = ((half_q[h] >> 1) * W2 + (half_q[h] & 1)) * p.ldd + half_co[h] + (piece & 7) * 8;
    else coloff = half_co[h] + (piece & 7) * 8;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = (lane >> 4) + 4 * i;
        if (mw + j < p.M) {
            int rowoff;
            if (MODE == 1) rowoff = (P0 + 2 * j + (xw + j >= p.W ? W2 : 0)) * p.ldd;
            else rowoff = (mw + j) * p.ldd;
            const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(ostage + j * OUT_PITCH + piece * 16);
            __builtin_nontemporal_store(v, reinterpret_cast<bf16x8_t*>(p.d + rowoff + coloff));
        }
    }
}

bool aligned16(const void* q) { return ((uintptr_t)q % 16) == 0; }

}  // namespace

// 1 when the launch can go to this kernel (the callers in conv_igemm.hip fall back to the gather kernel otherwise)
int ur_convt_gemm_supported(const ur_conv_desc* d, int dgrad, const void* x, const void* y) {
    if (!d || d->dtype != UR_BF16 || d->kh != 2 || d->kw != 2 || d->stride != 2 || d->pad != 0) return 0;
    if (d->oh != 2 * d->h || d->ow != 2 * d->w || d->w < 64) return 0;
    if (d->cin % 32 || d->cout % 64 || d->in_ld % 8 || d->out_ld % 8 || !aligned16(x) || !aligned16(y)) return 0;
    if (dgrad ? d->cin % GN != 0 : (4 * d->cout) % GN != 0) return 0;
    if ((long long)d->n * d->h * d->w * d->in_ld >= (1LL << 31) || (long long)d->n * d->oh * d->ow * d->out_ld >= (1LL << 31)) return 0;
    return 1;
}

int ur_convt_gemm_fwd(const void* x, const void* wp, const float* bias, void* y, const ur_conv_desc* d, hipStream_t st) {
    GemmParams p{};
    p.a = (const bf16_t*)x; p.b = (const bf16_t*)wp; p.bias = bias; p.d = (bf16_t*)y;
    p.M = d->n * d->h * d->w; p.N = 4 * d->cout; p.K = d->cin;
    p.lda = d->in_ld; p.ldb = d->cin; p.ldd = d->out_ld;                  // (cin is a multiple of 32: the packed rows are not padded)
    p.W = d->w; p.cout_t = d->cout;
    p.mtiles = ur_cdiv(p.M, GM); p.ntiles = p.N / GN;
    const int grid = ((p.mtiles + 7) / 8 * 8) * p.ntiles;
    hipLaunchKernelGGL((convt_gemm_kernel<1>), dim3(grid), dim3(NTH), 0, st, p);
    return ur_check_launch("convt_gemm_fwd");
}

int ur_convt_gemm_dgrad(const void* dy, const void* wp, void* dx, const ur_conv_desc* d, hipStream_t st) {
    GemmParams p{};
    p.a = (const bf16_t*)dy; p.b = (const bf16_t*)wp; p.d = (bf16_t*)dx;
    p.M = d->n * d->h * d->w; p.N = d->cin; p.K = 4 * d->cout;
    p.lda = d->out_ld; p.ldb = 4 * d->cout; p.ldd = d->in_ld;
    p.W = d->w; p.cout_t = d->cout;
    p.mtiles = ur_cdiv(p.M, GM); p.ntiles = p.N / GN;
    const int grid = ((p.mtiles + 7) / 8 * 8) * p.ntiles;
    hipLaunchKernelGGL((convt_gemm_kernel<2>), dim3(grid), dim3(NTH), 0, st, p);
    return ur_check_launch("convt_gemm_dgrad");
}
