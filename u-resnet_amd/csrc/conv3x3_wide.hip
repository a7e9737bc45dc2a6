// 3x3 / stride 1 / pad 1 convolution, bf16 NHWC, for layers that produce at least 128 channels: one persistent
// 8-wave workgroup per CU with all of its LDS, operands staged by LDS-DMA, v_mfma_f32_16x16x32_bf16.
//
// Why a second tile kernel.  conv3x3_tile.hip runs two 4-wave workgroups per CU on 16-channel slices and stages the
// halo through registers (40 registers, a ds_write pass and its address arithmetic per slice).  Its inner loop alone --
// the same LDS reads and MFMAs without any staging -- sustains about 1770 TFLOP/s on random data, the kernel 1100-1300;
// and the 16x16x32 shape of the same loop sustains about 2000 (the chip holds a higher clock on it;
// tools/micro/mfma_shape.hip).  That shape reduces over 32 channels per instruction, and a 32-channel stage does not fit
// twice into half a CU's LDS.  Here one workgroup owns the CU:
//   * tile = 16 x 32 output pixels x 128 output channels; wave w owns rows [4 (w & 3), +4) and channels [64 (w >> 2), +64):
//     the per-wave tile of the other kernel, with the halo shared by both channel halves (half the input traffic);
//   * reduction in 32-channel slices; a slice is three "thirds", one per horizontal tap offset dx;
//   * LDS: two halo buffers (18 x 34 pixels x 64 bytes, pixel-major, the four 16-byte pieces of a pixel XOR-swizzled with
//     bits 1-2 of the column: conflict-free ds_read_b128 for the 16 consecutive pixels of a fragment at every dx) and a
//     ring of three weight thirds (3 dy x 128 channels x 32 k, fragment order) = 152,064 bytes;
//   * everything arrives by LDS-DMA (global_load_lds_dwordx4): the weight third two thirds ahead, the next slice's halo
//     spread over the current slice's thirds; padding pixels are fetched from a zero block.  No staging registers, no
//     ds_write, one counted s_waitcnt vmcnt + one barrier per third;
//   * the workgroup is persistent: it walks (tile, channel block) items in an XCD-aware order and the prefetch cursors run
//     across item boundaries, so only the first item pays the fill latency; the epilogue stages one output row per wave
//     at a time in the halo buffer that has just been consumed.
#include "ur_common.h"
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

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

constexpr int WT_TW = 32, WT_TH = 16, WT_BN = 128, WT_KC = 32, WT_THREADS = 512, WT_R = 4;
constexpr int WT_HW = WT_TW + 2, WT_HR = WT_TH + 2;
constexpr int WT_ROW_BYTES = WT_HW * 64;                      // 2176
constexpr int WT_HALO_BYTES = WT_HR * WT_ROW_BYTES;           // 39168
constexpr int WT_HALO_SLOTS = WT_HALO_BYTES / 16;             // 2448 16-byte pieces
constexpr int WT_HALO_INSTR = (WT_HALO_SLOTS + 63) / 64;      // 39 wave-wide DMA instructions
constexpr int WT_HALO_PER_WAVE = (WT_HALO_INSTR + 7) / 8;     // 5
constexpr int WT_THIRD_BYTES = 3 * 2 * 4 * 1024;              // 24576: [dy][channel half][n][lane][16 B]
constexpr int WT_THIRD_PER_WAVE = WT_THIRD_BYTES / 1024 / 8;  // 3 DMA instructions per wave
constexpr int WT_OUT_PITCH = 144;                             // staged output pixel: 128 B + 16 B pad
constexpr int WT_STAGE_BYTES = WT_TW * WT_OUT_PITCH;          // one output row of one wave
constexpr int WT_LDS = 2 * WT_HALO_BYTES + 3 * WT_THIRD_BYTES;
static_assert(8 * WT_STAGE_BYTES <= WT_HALO_BYTES, "the eight waves' row staging lives in one halo buffer");

__device__ __align__(64) const unsigned char wt_zero_block[64] = {0};

struct WideParams {
    const bf16_t* x;      // [N][H][W][ld_in]
    const bf16_t* wpk;    // [Cout/128][Cin/32][dx][dy][half][n][lane][8]
    bf16_t* y;            // [N][H][W][ld_out]
    float* stats;         // [tiles][Cout][2] or null: per-tile sums (EPI 0: of y and y^2; EPI 3..6: of the BatchNorm backward pass)
    const bf16_t* addend; // EPI 1, 4, 6: [N][H][W][ld_add], added to the result before it is stored
    const bf16_t* bn_y;   // EPI 3..6: input of the BatchNorm whose output gradient this launch produces, [N][H][W][ld_bny]
    const bf16_t* bn_z;   // EPI 5, 6: that BatchNorm's activated output, dense [N][H][W][Cout] (the ReLU mask is z > 0)
    const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;
    int N, H, W, Cin, Cout, ld_in, ld_out, ld_add, ld_bny;
    int tiles_x, tiles_y, tiles, tiles8, nblocks, items;       // items = tiles8 * nblocks (tiles rounded up to 8)
};

__device__ __forceinline__ void wt_dma16(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}
// the same with the address as a uniform base + a 32-bit byte offset per lane (no 64-bit vector arithmetic)
__device__ __forceinline__ void wt_dma16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
template <int OFF>
__device__ __forceinline__ void wt_read16(bf16x8_t& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void wt_mfma(f32x4_t& c, const bf16x8_t& a, const bf16x8_t& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int CTRL>
__device__ __forceinline__ float wt_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over each row of 16 lanes, left in every lane of the row (DPP row rotations: VALU only, no LDS permutes)
__device__ __forceinline__ float wt_row_sum(float v) {
    v += wt_dpp<0x128>(v); v += wt_dpp<0x124>(v); v += wt_dpp<0x122>(v); v += wt_dpp<0x121>(v);
    return v;
}
template <int N>
__device__ __forceinline__ void wt_lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wt_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

template <typename F, int... I>
__device__ __forceinline__ void wt_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void wt_for(F&& f) { wt_for_impl(f, std::make_integer_sequence<int, N>{}); }

// position in the workgroup's work list: item (tile, channel block), 32-channel slice, third
struct Cursor {
    int item, nb, img, ty0, tx0, chunk, third;
    bool valid;
};

// EPI selects the epilogue at compile time: 0 = store (+ statistics of the fp32 results when p.stats), 1 = + addend,
// 3 = BatchNorm-backward sums with the ReLU mask rebuilt from bn_y, 4 = 3 + addend, 5 = sums with the mask from bn_z,
// 6 = 5 + addend (the numbering of conv3x3_tile.hip).
template <int EPI>
__global__ __launch_bounds__(WT_THREADS, 1) void conv3x3_wide_kernel(WideParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[WT_LDS];
    __shared__ float red[8][64][2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave & 3, ch = wave >> 2;
    const int m = lane & 15, kq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)lds;
    const unsigned ring0 = lds0 + 2 * WT_HALO_BYTES;
    const int chunks = p.Cin / WT_KC;

    auto decode = [&](Cursor& c) {                 // XCD-aware: ids congruent mod 8 walk the channel blocks of one tile
        for (;;) {
            if (c.item >= p.items) { c.valid = false; return; }
            const int xcd = c.item & 7, s = c.item >> 3;
            const int tile = (s / p.nblocks) * 8 + xcd;
            if (tile < p.tiles) {
                c.nb = s % p.nblocks;
                c.img = tile / (p.tiles_y * p.tiles_x);
                const int trem = tile - c.img * (p.tiles_y * p.tiles_x);
                c.ty0 = (trem / p.tiles_x) * WT_TH; c.tx0 = (trem % p.tiles_x) * WT_TW;
                c.valid = true;
                return;
            }
            c.item += gridDim.x;
        }
    };
    auto start = [&](Cursor& c) { c.item = blockIdx.x; c.chunk = 0; c.third = 0; decode(c); };
    auto next_third = [&](Cursor& c) {
        if (++c.third == 3) {
            c.third = 0;
            if (++c.chunk == chunks) { c.chunk = 0; c.item += gridDim.x; decode(c); }
        }
    };
    auto next_slice = [&](Cursor& c) {
        if (++c.chunk == chunks) { c.chunk = 0; c.item += gridDim.x; decode(c); }
    };

    // ---- DMA issue -------------------------------------------------------------------------------------------------------
    // The address of a DMA is a uniform 64-bit base (scalar arithmetic) plus a per-lane 32-bit byte offset that is computed
    // once: between the MFMA phases every VALU instruction is time the matrix pipe idles, the eight waves being in step.
    const unsigned lane16 = lane * 16;
    // weight third of cursor c into ring slot `slot`: 24 wave-instructions of 1 KiB, three per wave
    auto issue_weights = [&](const Cursor& c, int slot) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wpk) +
                                   ((size_t)(c.nb * chunks + c.chunk) * 3 + c.third) * WT_THIRD_BYTES + wave * (WT_THIRD_PER_WAVE * 1024);
        const unsigned dst = ring0 + slot * WT_THIRD_BYTES + wave * (WT_THIRD_PER_WAVE * 1024);
#pragma unroll
        for (int j = 0; j < WT_THIRD_PER_WAVE; ++j) wt_dma16_s(lane16, src + j * 1024, dst + j * 1024);
    };
    // Halo of the slice of cursor c into halo buffer `buf`.  Wave w owns wave-instructions w, w + 8, ...; lane l of
    // instruction q writes 16-byte slot s = 64 q + l: pixel s >> 2 (row-major over 18 x 34), physical piece s & 3, which
    // holds channel piece (s & 3) ^ ((column >> 1) & 3).  h_off = byte offset from the halo's first pixel; a tile on the image
    // border tests each slot's (row, column) and fetches its padding pixels from a block of zeros.
    auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
    const int halo_count = wave == 7 ? WT_HALO_PER_WAVE - 1 : WT_HALO_PER_WAVE;   // DMA instructions of this wave per halo
    unsigned h_off[WT_HALO_PER_WAVE];
#pragma unroll
    for (int j = 0; j < WT_HALO_PER_WAVE; ++j) {
        const int s = (wave + 8 * j) * 64 + lane;
        const int pix = s >> 2, hy = pix / WT_HW, hx = pix - hy * WT_HW;
        h_off[j] = (unsigned)(((hy * p.W + hx) * p.ld_in + (((s & 3) ^ ((hx >> 1) & 3)) << 3)) * 2);
    }
    auto issue_halo = [&](const Cursor& c, int buf) {
        const unsigned dst0 = lds0 + buf * WT_HALO_BYTES + wave * 1024;
        // (the first halo pixel (ty0 - 1, tx0 - 1) may lie outside the image: only the address is formed, and not used then)
        const unsigned char* org = reinterpret_cast<const unsigned char*>(p.x) +
                                   (((long long)(c.img * p.H + c.ty0 - 1) * p.W + c.tx0 - 1) * p.ld_in + c.chunk * WT_KC) * 2;
        const bool inside = c.ty0 >= 1 && c.tx0 >= 1 && c.ty0 + WT_TH + 1 <= p.H && c.tx0 + WT_TW + 1 <= p.W;
        const int l = opaque(lane);
        wt_for<WT_HALO_PER_WAVE>([&](auto J_) {
            constexpr int j = J_;
            if (wave + 8 * j < WT_HALO_INSTR) {              // uniform: the waits below count the instructions issued
                const int s = (wave + 8 * j) * 64 + l;
                if (s < WT_HALO_SLOTS) {
                    if (inside) {
                        wt_dma16_s(h_off[j], org, dst0 + j * 8192);
                    } else {                                 // tile on the image border: (row, column) of the slot once more
                        const int pix = s >> 2, hy = pix / WT_HW, hx = pix - hy * WT_HW;
                        const int gy = c.ty0 - 1 + hy, gx = c.tx0 - 1 + hx;
                        const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                        // (opaque: the 64-bit form of the offset must not become a second loop-invariant register pair)
                        wt_dma16(ok ? (const void*)(org + (unsigned)opaque((int)h_off[j])) : (const void*)wt_zero_block, dst0 + j * 8192);
                    }
                }
            }
        });
    };
    // s_waitcnt vmcnt(n) for the handful of counts that occur (the instruction takes an immediate)
    auto vm_wait_upto = [&](int n) {
        if (n == 0) wt_vm_wait<0>();
        else if (n == 3) wt_vm_wait<3>();
        else if (n == 7) wt_vm_wait<7>();
        else if (n == 8) wt_vm_wait<8>();
        else if (n == 23) wt_vm_wait<23>();
        else if (n == 24) wt_vm_wait<24>();
        else if (n == 25) wt_vm_wait<25>();
        else wt_vm_wait<0>();
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------------------
    // Pixel operand: lane (m, kq) reads piece kq of pixel column h * 16 + m + dx of halo row rg * 4 + i (row: + i * ROW_BYTES)
    unsigned a0[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int hx = h * 16 + m + g;
            a0[g][h] = lds0 + (rg * WT_R * WT_HW + hx) * 64 + ((kq ^ ((hx >> 1) & 3)) << 4);
        }
    // Weight operand: fragment (dy, n) of this wave's channel half at + (dy * 8 + n) * 1024 inside a third
    const unsigned b0 = ring0 + ch * 4096 + lane * 16;

    f32x4_t acc[WT_R][2][4];
    bf16x8_t br[3][4], ar[3][2];
    auto read_w = [&](auto DY_, unsigned bb) {
        constexpr int dy = DY_;
        wt_for<4>([&](auto N_) { constexpr int n = N_; wt_read16<(dy * 8 + n) * 1024>(br[dy][n], bb); });
    };
    auto read_row = [&](auto I_, const unsigned (&ab)[2]) {         // halo row i of this wave's group -> ar[i % 3]
        constexpr int i = I_;
        wt_read16<i * WT_ROW_BYTES>(ar[i % 3][0], ab[0]);
        wt_read16<i * WT_ROW_BYTES>(ar[i % 3][1], ab[1]);
    };
    auto mma = [&](auto DY_, auto I_) {                               // output row i - dy from halo row i and tap row dy
        constexpr int dy = DY_, i = I_, a = i - dy;
        static_assert(a >= 0 && a < WT_R, "");
        wt_for<2>([&](auto H_) { wt_for<4>([&](auto N_) {
            constexpr int h = H_, n = N_;
            // (asm, accumulating in place: left to the register allocator, 128 accumulators + 72 operand registers that are
            // live around the loops end up copied and spilled -- and a copy of an operand whose read is in flight is wrong)
            wt_mfma(acc[a][h][n], br[dy][n], ar[i % 3][h]);
        }); });
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;

    Cursor cur, pw, ph;
    start(cur);
    if (!cur.valid) return;
    pw = cur; ph = cur;
    // fill: halo of the first slice, weight thirds 0 and 1
    issue_halo(ph, 0);
    next_slice(ph);
    issue_weights(pw, 0); next_third(pw);
    if (pw.valid) { issue_weights(pw, 1); next_third(pw); }
    wt_vm_wait<0>();
    __builtin_amdgcn_s_barrier();

    // Schedule.  The barrier of third s sits after its fourth MFMA phase (of six).  At that point every wave has read all it
    // needs of the third's weight slot and of the halo rows, so right after it the wave issues "stage s+1" -- the weight third
    // s+3 into the slot just released and, at a slice boundary, the halo of the slice after next into the buffer just
    // released -- and the LDS reads of third s+1's first operands, which the two remaining MFMA phases cover.  A stage lands
    // two thirds (weights) or three (halo) after it is issued; the barrier of third s admits in flight only what was issued
    // after the operands third s+1 needs (loads and stores retire in issue order).
    bool w_issued = false, h_issued = false;       // the latest weight stage / the halo stage for the next slice went out
    int stores_in_flight = 0;                      // output stores of the previous item, when their count is known
    if (pw.valid) { issue_weights(pw, 2); next_third(pw); w_issued = true; }
    if (ph.valid) { issue_halo(ph, 1); next_slice(ph); h_issued = true; }
    {
        const unsigned ab[2] = {a0[0][0], a0[0][1]};
        read_w(I0{}, b0); read_row(I0{}, ab); read_w(I1{}, b0); read_w(I2{}, b0);
    }

    int step = 0, slice = 0;                       // ring slot = step % 3, halo buffer = slice & 1
    while (cur.valid) {
#pragma unroll
        for (int a = 0; a < WT_R; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[a][h][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int item_tile_img = cur.img, item_ty0 = cur.ty0, item_tx0 = cur.tx0, item_nb = cur.nb;
        const int item_id = cur.item;
        bool halo_deferred = false;
        // one 32-channel slice = three thirds; the item's last slice is a copy of the code whose last third leaves the reads
        // of the next third's operands to after the epilogue
        auto do_slice = [&](int chunk, auto LAST_) {
            constexpr bool LAST = LAST_;
            const unsigned hoff = (slice & 1) * WT_HALO_BYTES;
            const bool h_this_slice = h_issued;    // the halo stage that went out for this slice's successor
            wt_for<3>([&](auto G_) {
                constexpr int g = G_;
                const int slot = step % 3;
                const unsigned ab[2] = {a0[g][0] + hoff, a0[g][1] + hoff};
                // in flight on entry, in order: weights dy 0 (4 reads), row 0 (2), weights dy 1 (4), weights dy 2 (4)
                wt_lds_wait<8>();
                read_row(I1{}, ab);
                mma(I0{}, I0{});
                read_row(I2{}, ab);
                wt_lds_wait<2>();
                mma(I0{}, I1{}); mma(I1{}, I1{});
                read_row(I3{}, ab);
                wt_lds_wait<2>();
                mma(I0{}, I2{}); mma(I1{}, I2{}); mma(I2{}, I2{});
                read_row(I4{}, ab); read_row(I5{}, ab);
                wt_lds_wait<4>();
                mma(I0{}, I3{}); mma(I1{}, I3{}); mma(I2{}, I3{});
                wt_lds_wait<0>();
                {
                    int admit = 0;
                    if constexpr (g == 2) admit = w_issued ? WT_THIRD_PER_WAVE : 0;
                    else if (w_issued && h_this_slice) admit = WT_THIRD_PER_WAVE + halo_count + stores_in_flight;
                    vm_wait_upto(admit);
                    if constexpr (g == 1) stores_in_flight = 0;
                }
                __builtin_amdgcn_s_barrier();
                // stage for the next third
                w_issued = pw.valid;
                if (w_issued) { issue_weights(pw, slot); next_third(pw); }
                if constexpr (g == 2) {
                    h_issued = ph.valid;
                    if (h_issued) {
                        if constexpr (!LAST) { issue_halo(ph, slice & 1); next_slice(ph); }
                        else halo_deferred = true;             // the epilogue stages its rows in that buffer first
                    }
                }
                // The two remaining MFMA phases, with the LDS reads of the next third's first operands issued in between.  After
                // the item's last third the reads wait until the epilogue is done: it needs the registers.
                if constexpr (g == 2 && LAST) {
                    mma(I1{}, I4{}); mma(I2{}, I4{}); mma(I2{}, I5{});
                } else {
                    const unsigned bbn = b0 + ((step + 1) % 3) * WT_THIRD_BYTES;
                    const unsigned hoffn = g == 2 ? ((slice + 1) & 1) * WT_HALO_BYTES : hoff;
                    const unsigned abn[2] = {a0[(g + 1) % 3][0] + hoffn, a0[(g + 1) % 3][1] + hoffn};
                    read_w(I0{}, bbn); read_row(I0{}, abn);
                    mma(I1{}, I4{}); mma(I2{}, I4{});
                    read_w(I1{}, bbn);
                    mma(I2{}, I5{});
                    read_w(I2{}, bbn);
                }
                ++step;
            });
            next_slice(cur);           // (cur.third stays 0: the thirds are unrolled above)
            ++slice;
        };
        for (int chunk = 0; chunk + 1 < chunks; ++chunk) do_slice(chunk, std::false_type{});
        do_slice(chunks - 1, std::true_type{});

        // ---- epilogue: one output row per wave at a time through the halo buffer the last slice has released ----------
        // The weights are the MFMA's row operand, so a lane holds four consecutive channels (n * 16 + 4 * kq + i) of one
        // pixel (h * 16 + m): one 8-byte LDS write per accumulator, 16-byte reads and stores on the way out.  All eight
        // waves are here at the same time and the matrix pipe idles: the instruction count is what this costs.
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // the last MFMAs' results (the compiler does not see them)
        const int tile = (item_id >> 3) / p.nblocks * 8 + (item_id & 7);
        const bool want_stats = p.stats != nullptr;
        const bool full_tile = item_ty0 + WT_TH <= p.H && item_tx0 + WT_TW <= p.W;
        auto epilogue = [&](auto FULL_, auto STATS_) {
            constexpr bool FULL = FULL_, STATS = STATS_;
            const int el = opaque(lane), em = el & 15, ekq = el >> 4;   // (not hoisted out of the item loop)
            unsigned char* ostage = lds + ((slice - 1) & 1) * WT_HALO_BYTES + wave * WT_STAGE_BYTES;
            unsigned char* wr = ostage + em * WT_OUT_PITCH + ekq * 8;
            const unsigned char* rd = ostage + (el >> 3) * WT_OUT_PITCH + (el & 7) * 16;
            const unsigned lane_out = (unsigned)(((el >> 3) * p.ld_out + (el & 7) * 8) * 2);
            f32x2_t s1[4][2], s2[4][2];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 2; ++i) s1[n][i] = s2[n][i] = f32x2_t{0.f, 0.f};
#pragma unroll
            for (int a = 0; a < WT_R; ++a) {
                const int oy = item_ty0 + rg * WT_R + a;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool inside = FULL || (oy < p.H && item_tx0 + h * 16 + em < p.W);
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const f32x4_t v = acc[a][h][n];
                        if constexpr (STATS) {
                            f32x2_t lo{v[0], v[1]}, hi{v[2], v[3]};
                            if (!inside) { lo = f32x2_t{0.f, 0.f}; hi = lo; }
                            s1[n][0] += lo; s1[n][1] += hi;
                            s2[n][0] = __builtin_elementwise_fma(lo, lo, s2[n][0]);
                            s2[n][1] = __builtin_elementwise_fma(hi, hi, s2[n][1]);
                        }
                        bf16x4_t o;
                        o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
                        *reinterpret_cast<bf16x4_t*>(wr + h * 16 * WT_OUT_PITCH + n * 32) = o;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // uniform part of the address in scalar registers, the lane's part is one 32-bit offset
                unsigned char* yrow = reinterpret_cast<unsigned char*>(p.y) +
                                      (((long long)(item_tile_img * p.H + oy) * p.W + item_tx0) * p.ld_out + item_nb * WT_BN + ch * 64) * 2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (FULL || (oy < p.H && item_tx0 + (el >> 3) + 8 * k < p.W)) {
                        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(rd + 8 * k * WT_OUT_PITCH);
                        OUT_STORE(reinterpret_cast<bf16x8_t*>(yrow + (size_t)(8 * k * p.ld_out * 2) + lane_out), v);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if constexpr (STATS) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float u = wt_row_sum(s1[n][i >> 1][i & 1]), q = wt_row_sum(s2[n][i >> 1][i & 1]);   // over the row's 16 pixels
                        if (em == 0) *reinterpret_cast<float2*>(&red[wave][n * 16 + 4 * ekq + i][0]) = float2{u, q};
                    }
            }
        };
        // Epilogues that read other tensors at the output's positions.  After the row is staged a lane holds 8 consecutive
        // channels (piece = lane & 7) of pixels (lane >> 3) + 8 k: the layout of its 16-byte stores, and of 16-byte loads
        // of the addend and of the BatchNorm's input / output.  The loads of row a + 1 are issued before row a is worked on.
        auto epilogue_x = [&](auto FULL_) {
            constexpr bool FULL = FULL_;
            constexpr bool BNS = EPI >= 3, ADD = EPI == 1 || EPI == 4 || EPI == 6, FROMZ = EPI == 5 || EPI == 6;
            const int el = opaque(lane), em = el & 15, ekq = el >> 4, piece = el & 7, pr = el >> 3;
            unsigned char* ostage = lds + ((slice - 1) & 1) * WT_HALO_BYTES + wave * WT_STAGE_BYTES;
            unsigned char* wr = ostage + em * WT_OUT_PITCH + ekq * 8;
            const unsigned char* rd = ostage + pr * WT_OUT_PITCH + piece * 16;
            const int cb = item_nb * WT_BN + ch * 64;            // first channel of this wave's half
            const unsigned lane_out = (unsigned)((pr * p.ld_out + piece * 8) * 2), lane_add = (unsigned)((pr * p.ld_add + piece * 8) * 2);
            const unsigned lane_y = (unsigned)((pr * p.ld_bny + piece * 8) * 2), lane_z = (unsigned)((pr * p.Cout + piece * 8) * 2);
            float bm[8], bis[8], bga[8], bbe[8], bs1[8], bs2[8];
            if constexpr (BNS) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bm[j] = p.bn_mean[cb + piece * 8 + j]; bis[j] = p.bn_invstd[cb + piece * 8 + j];
                    if constexpr (!FROMZ) { bga[j] = p.bn_gamma[cb + piece * 8 + j]; bbe[j] = p.bn_beta[cb + piece * 8 + j]; }
                    bs1[j] = 0.f; bs2[j] = 0.f;
                }
            }
            auto row_base = [&](const bf16_t* t, int ld, int a) {      // uniform: first pixel of output row a of this wave
                const int oy = item_ty0 + rg * WT_R + a;
                return reinterpret_cast<const unsigned char*>(t) + (((long long)(item_tile_img * p.H + oy) * p.W + item_tx0) * ld + cb) * 2;
            };
            // The BatchNorm tensors of a row are loaded while the row before it is worked on; where there is an addend as
            // well it is loaded at the top of its own row (a third tensor a row ahead does not fit the register file).
            constexpr bool ADD_AHEAD = ADD && !BNS;
            bf16x8_t yv[2][4], zv[2][4], av[2][4];
            auto load_row = [&](auto A_, auto AHEAD_) {
                constexpr int a = A_, b = a & 1;
                constexpr bool AHEAD = AHEAD_;
                const int oy = item_ty0 + rg * WT_R + a;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (FULL || (oy < p.H && item_tx0 + pr + 8 * k < p.W)) {
                        if constexpr (BNS && AHEAD) yv[b][k] = Y_LOAD(reinterpret_cast<const bf16x8_t*>(row_base(p.bn_y, p.ld_bny, a) + (size_t)(8 * k * p.ld_bny * 2) + lane_y));
                        if constexpr (FROMZ && AHEAD) zv[b][k] = *reinterpret_cast<const bf16x8_t*>(row_base(p.bn_z, p.Cout, a) + (size_t)(8 * k * p.Cout * 2) + lane_z);
                        if constexpr (ADD && AHEAD == ADD_AHEAD) av[ADD_AHEAD ? b : 0][k] = *reinterpret_cast<const bf16x8_t*>(row_base(p.addend, p.ld_add, a) + (size_t)(8 * k * p.ld_add * 2) + lane_add);
                    }
                }
            };
            load_row(I0{}, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            wt_for<WT_R>([&](auto A_) {
                constexpr int a = A_, b = a & 1;
                const int oy = item_ty0 + rg * WT_R + a;
                if constexpr (ADD && !ADD_AHEAD) { load_row(A_, std::false_type{}); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const f32x4_t v = acc[a][h][n];
                        bf16x4_t o;
                        o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
                        *reinterpret_cast<bf16x4_t*>(wr + h * 16 * WT_OUT_PITCH + n * 32) = o;
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // (the barriers keep the compiler from issuing every row's loads at the top: 4 rows x 3 tensors of them do not fit)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (a + 1 < WT_R) load_row(std::integral_constant<int, a + 1>{}, std::true_type{});
                __builtin_amdgcn_sched_barrier(0);
                unsigned char* yrow = const_cast<unsigned char*>(row_base(p.y, p.ld_out, a));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (FULL || (oy < p.H && item_tx0 + pr + 8 * k < p.W)) {
                        bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(rd + 8 * k * WT_OUT_PITCH);
                        if constexpr (ADD) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = (bf16_t)((float)v[j] + (float)av[ADD_AHEAD ? b : 0][k][j]);
                        }
                        if constexpr (BNS) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const float xh = ((float)yv[b][k][j] - bm[j]) * bis[j];
                                float pre;
                                if constexpr (FROMZ) pre = (float)zv[b][k][j]; else pre = xh * bga[j] + bbe[j];
                                const float gv = pre > 0.f ? (float)v[j] : 0.f;
                                bs1[j] += gv; bs2[j] += gv * xh;
                            }
                        }
                        OUT_STORE(reinterpret_cast<bf16x8_t*>(yrow + (size_t)(8 * k * p.ld_out * 2) + lane_out), v);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (BNS) {
                // lanes with equal lane & 7 hold the same 8 channels: fold lane bits 3 (row rotation), 4 and 5
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float u = bs1[j], q = bs2[j];
                    u += wt_dpp<0x128>(u); q += wt_dpp<0x128>(q);
                    u += __shfl_xor(u, 16, 64); q += __shfl_xor(q, 16, 64);
                    u += __shfl_xor(u, 32, 64); q += __shfl_xor(q, 32, 64);
                    if (pr == 0) *reinterpret_cast<float2*>(&red[wave][piece * 8 + j][0]) = float2{u, q};
                }
            }
        };
        if constexpr (EPI == 0) {
            if (want_stats) { if (full_tile) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::false_type{}, std::true_type{}); }
            else { if (full_tile) epilogue(std::true_type{}, std::false_type{}); else epilogue(std::false_type{}, std::false_type{}); }
        } else {
            if (full_tile) epilogue_x(std::true_type{}); else epilogue_x(std::false_type{});
        }
        __syncthreads();               // staging area and `red` complete; the next halo DMA may overwrite the buffer
        if (want_stats && tid < WT_BN) {
            const int hh = tid >> 6, c = tid & 63;
            float2 o;
            o.x = red[hh * 4][c][0] + red[hh * 4 + 1][c][0] + red[hh * 4 + 2][c][0] + red[hh * 4 + 3][c][0];
            o.y = red[hh * 4][c][1] + red[hh * 4 + 1][c][1] + red[hh * 4 + 2][c][1] + red[hh * 4 + 3][c][1];
            *reinterpret_cast<float2*>(p.stats + ((size_t)tile * p.Cout + item_nb * WT_BN + tid) * 2) = o;
        }
        // a tile inside the image issues exactly 16 row stores per wave (+ 1 for the statistics in waves 0 and 1): the next
        // two barriers admit them in flight; for an edge tile the count is not known and those barriers wait for them
        stores_in_flight = full_tile ? 4 * WT_R + ((want_stats && wave < 2) ? 1 : 0) : 0;
        if (halo_deferred) { issue_halo(ph, (slice - 1) & 1); next_slice(ph); }
        {   // first operands of the next item's first third (as at the end of every other third)
            const unsigned bbn = b0 + (step % 3) * WT_THIRD_BYTES;
            const unsigned hoffn = (slice & 1) * WT_HALO_BYTES;
            const unsigned abn[2] = {a0[0][0] + hoffn, a0[0][1] + hoffn};
            read_w(I0{}, bbn); read_row(I0{}, abn); read_w(I1{}, bbn); read_w(I2{}, bbn);
        }
    }
}

// Weight image (packed by ur_pack3x3_image, conv3x3_tile.hip): 16-byte piece index
// ((((nb * chunks + chunk) * 3 + dx) * 3 + dy) * 2 + half) * 4 + n, then lane, then 8 k-values:
//   forward : value = w[co = nb*128 + half*64 + n*16 + (lane & 15)][ci = chunk*32 + 8*(lane >> 4) + j][tap = dy*3 + dx]
//   dgrad   : value = w[co = chunk*32 + 8*(lane >> 4) + j][ci = nb*128 + half*64 + n*16 + (lane & 15)][8 - (dy*3 + dx)]

int wide_check(const ur_conv_desc* d, bool dgrad) {
    UR_REQUIRE(d != nullptr, "conv3x3_wide: null descriptor");
    UR_REQUIRE(d->dtype == UR_BF16, "conv3x3_wide: bf16 only");
    UR_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->oh == d->h && d->ow == d->w,
               "conv3x3_wide: 3x3 stride 1 pad 1 only");
    const int rows = dgrad ? d->cin : d->cout, redc = dgrad ? d->cout : d->cin;
    UR_REQUIRE(rows % WT_BN == 0 && redc % WT_KC == 0, "conv3x3_wide: needs %d | produced channels and %d | reduced channels", WT_BN,
               WT_KC);
    UR_REQUIRE(d->in_ld % 8 == 0 && d->out_ld % 8 == 0, "conv3x3_wide: pitches must be multiples of 8");
    UR_REQUIRE((long long)d->n * d->h * d->w * (d->in_ld > d->out_ld ? d->in_ld : d->out_ld) < (1LL << 31),
               "conv3x3_wide: tensor exceeds 2^31 elements");
    return UR_OK;
}

struct WideExtra {        // what the epilogues other than 0 read (see WideParams)
    const void* addend; int ld_add;
    const void* bn_y; int ld_bny; const void* bn_z; const float *mean, *invstd, *gamma, *beta;
};

int wide_launch(const void* src, const void* wpk, void* dst, float* stats, const ur_conv_desc* d, bool dgrad, hipStream_t st,
                int epi = 0, const WideExtra* ex = nullptr) {
    WideParams p{};
    p.x = (const bf16_t*)src; p.wpk = (const bf16_t*)wpk; p.y = (bf16_t*)dst; p.stats = stats;
    p.N = d->n; p.H = d->h; p.W = d->w;
    p.Cin = dgrad ? d->cout : d->cin; p.Cout = dgrad ? d->cin : d->cout;
    p.ld_in = dgrad ? d->out_ld : d->in_ld; p.ld_out = dgrad ? d->in_ld : d->out_ld;
    if (ex) {
        p.addend = (const bf16_t*)ex->addend; p.ld_add = ex->ld_add ? ex->ld_add : p.Cout;
        p.bn_y = (const bf16_t*)ex->bn_y; p.ld_bny = ex->ld_bny ? ex->ld_bny : p.Cout; p.bn_z = (const bf16_t*)ex->bn_z;
        p.bn_mean = ex->mean; p.bn_invstd = ex->invstd; p.bn_gamma = ex->gamma; p.bn_beta = ex->beta;
    }
    p.tiles_x = ur_cdiv(d->w, WT_TW); p.tiles_y = ur_cdiv(d->h, WT_TH);
    p.tiles = d->n * p.tiles_x * p.tiles_y;
    p.tiles8 = (p.tiles + 7) / 8 * 8;
    p.nblocks = p.Cout / WT_BN;
    p.items = p.tiles8 * p.nblocks;
    int grid = 256;                                   // one workgroup per CU
    if (p.items < grid) grid = p.items;
#define UR_WIDE_LAUNCH(E) hipLaunchKernelGGL(conv3x3_wide_kernel<E>, dim3(grid), dim3(WT_THREADS), 0, st, p)
    switch (epi) {           // (4, 5 and 6 compile but leave no registers for the main loop: those stay with the tile kernel)
        case 0: UR_WIDE_LAUNCH(0); break;
        case 1: UR_WIDE_LAUNCH(1); break;
        default: UR_WIDE_LAUNCH(3); break;
    }
#undef UR_WIDE_LAUNCH
    return ur_check_launch("conv3x3_wide");
}

}  // namespace

extern "C" {

int ur_conv3x3_wide_supported(const ur_conv_desc* d, int dgrad) {
    if (!d || d->dtype != UR_BF16) return 0;
    if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->oh == d->h && d->ow == d->w)) return 0;
    const int rows = dgrad ? d->cin : d->cout, redc = dgrad ? d->cout : d->cin;
    if (rows % WT_BN != 0 || redc % WT_KC != 0 || d->in_ld % 8 != 0 || d->out_ld % 8 != 0) return 0;
    if ((long long)d->n * d->h * d->w * (d->in_ld > d->out_ld ? d->in_ld : d->out_ld) >= (1LL << 31)) return 0;
    return 1;
}

size_t ur_conv3x3_wide_packed_elems(const ur_conv_desc* d) { return d ? (size_t)d->cin * d->cout * 9 : 0; }

int ur_conv3x3_wide_pack_weight(const float* w, void* packed, const ur_conv_desc* d, int dgrad, ur_stream_t s) {
    if (int e = wide_check(d, dgrad != 0)) return e;
    UR_REQUIRE(w && packed, "conv3x3_wide_pack_weight: null pointer");
    return ur_pack3x3_image(w, packed, d->cin, d->cout, dgrad != 0, 1, (hipStream_t)s);
}

// Whether this kernel is the faster one for the layer (measured against ur_conv3x3_tile_* on MI355X): it needs enough
// (tile, channel block) items to occupy the 256 CUs without a ragged last round -- 384 items are two rounds of which
// the second is half empty, and there the smaller workgroups of the tile kernel balance better.
int ur_conv3x3_wide_preferred(const ur_conv_desc* d, int dgrad) {
    if (!ur_conv3x3_wide_supported(d, dgrad)) return 0;
    const long long tiles = (long long)d->n * ur_cdiv(d->h, WT_TH) * ur_cdiv(d->w, WT_TW);
    const long long items = tiles * ((dgrad ? d->cin : d->cout) / WT_BN);
    return (items >= 128 && items <= 256) || items >= 512;
}

int ur_conv3x3_wide_stats_rows(const ur_conv_desc* d) { return d ? d->n * ur_cdiv(d->h, WT_TH) * ur_cdiv(d->w, WT_TW) : 0; }

int ur_conv3x3_wide_fwd(const void* x, const void* w_packed, void* y, float* stats, const ur_conv_desc* d, ur_stream_t s) {
    if (int e = wide_check(d, false)) return e;
    UR_REQUIRE(x && w_packed && y, "conv3x3_wide_fwd: null pointer");
    UR_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w_packed % 16) == 0,
               "conv3x3_wide_fwd: tensors must be 16-byte aligned");
    return wide_launch(x, w_packed, y, stats, d, false, (hipStream_t)s);
}

int ur_conv3x3_wide_dgrad(const void* dy, const void* w_packed_dgrad, void* dx, const ur_conv_desc* d, ur_stream_t s) {
    if (int e = wide_check(d, true)) return e;
    UR_REQUIRE(dy && w_packed_dgrad && dx, "conv3x3_wide_dgrad: null pointer");
    UR_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)w_packed_dgrad % 16) == 0,
               "conv3x3_wide_dgrad: tensors must be 16-byte aligned");
    return wide_launch(dy, w_packed_dgrad, dx, nullptr, d, true, (hipStream_t)s);
}

int ur_conv3x3_wide_dgrad_add(const void* dy, const void* w_packed_dgrad, void* dx, const void* addend, int add_ld,
                              const ur_conv_desc* d, ur_stream_t s) {
    if (int e = wide_check(d, true)) return e;
    UR_REQUIRE(dy && w_packed_dgrad && dx && addend, "conv3x3_wide_dgrad_add: null pointer");
    UR_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)w_packed_dgrad % 16) == 0 &&
                   ((uintptr_t)addend % 16) == 0 && add_ld % 8 == 0,
               "conv3x3_wide_dgrad_add: tensors must be 16-byte aligned, the addend's pitch a multiple of 8");
    WideExtra ex{};
    ex.addend = addend; ex.ld_add = add_ld;
    return wide_launch(dy, w_packed_dgrad, dx, nullptr, d, true, (hipStream_t)s, 1, &ex);
}

int ur_conv3x3_wide_dgrad_bnsums(const void* dy, const void* w_packed_dgrad, void* dx, const void* bn_y, int bn_y_ld,
                                 const float* mean, const float* invstd, const float* gamma, const float* beta, float* partial,
                                 const ur_conv_desc* d, ur_stream_t s) {
    if (int e = wide_check(d, true)) return e;
    UR_REQUIRE(dy && w_packed_dgrad && dx && bn_y && mean && invstd && gamma && beta && partial,
               "conv3x3_wide_dgrad_bnsums: null pointer");
    UR_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)w_packed_dgrad % 16) == 0 &&
                   ((uintptr_t)bn_y % 16) == 0 && bn_y_ld % 8 == 0,
               "conv3x3_wide_dgrad_bnsums: tensors must be 16-byte aligned, pitches multiples of 8");
    WideExtra ex{};
    ex.bn_y = bn_y; ex.ld_bny = bn_y_ld; ex.mean = mean; ex.invstd = invstd; ex.gamma = gamma; ex.beta = beta;
    return wide_launch(dy, w_packed_dgrad, dx, partial, d, true, (hipStream_t)s, 3, &ex);
}

}  // extern "C"
