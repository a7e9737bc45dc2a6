// Loss reductions and output masks (NHWC logits, classes contiguous per pixel).
//   cross-entropy with ignore_index: one thread per pixel keeps the <=64-class row in registers
//   (max, log-sum-exp, NLL), wave64 butterfly + LDS for the (sum, valid-count) pair, one partial
//   per workgroup, a finalize kernel adds the partials in double in a fixed order.
//   The backward kernel rebuilds softmax from the saved log-sum-exp and scales by the
//   device-resident valid count, so no host synchronisation sits between forward and backward.
#include "ur_common.h"
#include <pthread.h>

namespace {

constexpr int NT = 256;

__device__ __forceinline__ void block_sum2(float a, float b, float* out2) {
    __shared__ float sh[2][NT / 64];
    a = ur_wave_sum(a); b = ur_wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wave] = a; sh[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s0 = 0.f, s1 = 0.f;
        for (int i = 0; i < NT / 64; ++i) { s0 += sh[0][i]; s1 += sh[1][i]; }
        out2[0] = s0; out2[1] = s1;
    }
}

// exp() of the softmax terms.  bf16 logits: the hardware exponential (v_exp_f32 on x * log2 e, about 1e-6 relative), a
// tenth of the instructions of the library routine -- the cross-entropy kernels were bound by it (19 calls per pixel), and
// the result is rounded to bf16 or summed over 19 terms anyway.  fp32 logits keep expf.
template <typename T> __device__ __forceinline__ float ce_exp(float x) {
    if constexpr (sizeof(T) == 2) return __expf(x);
    else return expf(x);
}

template <typename T>
__global__ __launch_bounds__(NT) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                    long long rows, int classes, int ld, long long ignore_index,
                                                    float* __restrict__ lse, float* __restrict__ partial) {
    float nll = 0.f, cnt = 0.f;
    for (long long r = blockIdx.x * (long long)NT + threadIdx.x; r < rows; r += (long long)gridDim.x * NT) {
        const T* row = logits + r * ld;
        float m = -INFINITY;
        for (int c = 0; c < classes; ++c) m = fmaxf(m, ur_to_float(row[c]));
        float s = 0.f;
        for (int c = 0; c < classes; ++c) s += ce_exp<T>(ur_to_float(row[c]) - m);
        const float l = m + logf(s);
        lse[r] = l;
        const long long t = target[r];
        if (t != ignore_index && t >= 0 && t < classes) { nll += l - ur_to_float(row[t]); cnt += 1.f; }
    }
    block_sum2(nll, cnt, partial + (size_t)blockIdx.x * 2);
}

// out[0] = sum, out[1] = count, out[2] = sum / count (mean; NaN when count == 0, like PyTorch)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partial, int nparts,
                                                            float* __restrict__ out, double fixed_count) {
    // one workgroup; lane t sums partials t, t+256, ... in double, then a fixed-order tree over the lanes
    __shared__ double ss[256], cs[256];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) { s += partial[i * 2]; c += partial[i * 2 + 1]; }
    ss[threadIdx.x] = s; cs[threadIdx.x] = c;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) { ss[threadIdx.x] += ss[threadIdx.x + h]; cs[threadIdx.x] += cs[threadIdx.x + h]; }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    s = ss[0]; c = cs[0];
    if (fixed_count > 0.0) c = fixed_count;
    out[0] = (float)s; out[1] = (float)c; out[2] = (float)(s / c);
}

template <typename T>
__global__ __launch_bounds__(NT) void ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                    long long rows, int classes, int ld, long long ignore_index,
                                                    const float* __restrict__ lse, const float* __restrict__ count,
                                                    const float* __restrict__ gout, T* __restrict__ dlogits, int dld) {
    const float scale = (gout ? gout[0] : 1.f) / count[0];
    for (long long r = blockIdx.x * (long long)NT + threadIdx.x; r < rows; r += (long long)gridDim.x * NT) {
        const T* row = logits + r * ld;
        T* drow = dlogits + r * dld;
        const long long t = target[r];
        const bool valid = t != ignore_index && t >= 0 && t < classes;
        const float l = lse[r];
        for (int c = 0; c < classes; ++c) {
            float g = 0.f;
            if (valid) g = (ce_exp<T>(ur_to_float(row[c]) - l) - (c == t ? 1.f : 0.f)) * scale;
            drow[c] = ur_from_float<T>(g);
        }
    }
}

// Dense logits (ld == classes) with short rows: a thread-per-pixel walk reads and writes 2-byte elements 38 bytes
// apart, so the staged variants move each workgroup's 256 rows (one contiguous, 16-byte aligned span) through LDS
// with 16-byte accesses and do the per-pixel arithmetic on the LDS copy.
constexpr int CE_STAGE_BYTES = 32768;              // rows of at most 128 bytes

template <typename T, bool BWD>
__global__ __launch_bounds__(NT) void ce_staged_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                       long long rows, int classes, long long ignore_index,
                                                       float* __restrict__ lse, float* __restrict__ partial,
                                                       const float* __restrict__ count, const float* __restrict__ gout,
                                                       T* __restrict__ dlogits) {
    __shared__ uint4 stage[CE_STAGE_BYTES / 16];
    T* tile = reinterpret_cast<T*>(stage);
    const int row_bytes = classes * (int)sizeof(T);
    const int span16 = NT * row_bytes / 16;        // 16-byte pieces of a full 256-row span (exact: 256 * anything)
    float nll = 0.f, cnt = 0.f;
    float scale = 0.f;
    if (BWD) scale = (gout ? gout[0] : 1.f) / count[0];
    const long long nblocks = (rows + NT - 1) / NT;
    // A full span's pieces are fetched into registers one span ahead (at most 8 per thread: 32 KiB / 16 / 256), so the
    // memory latency of span b + gridDim.x passes under the arithmetic of span b instead of in front of every span.
    constexpr int PF = CE_STAGE_BYTES / 16 / NT;
    uint4 pf[PF];
#define UR_CE_FETCH(bb)                                                                                    \
    do {                                                                                                   \
        const uint4* src_ = reinterpret_cast<const uint4*>(logits + (bb) * NT * classes);                  \
        _Pragma("unroll") for (int j = 0; j < PF; ++j) {                                                   \
            const int i_ = (int)threadIdx.x + j * NT;                                                      \
            if (j * NT < span16) pf[j] = src_[i_ < span16 ? i_ : 0];   /* uniform test; surplus lanes repeat piece 0 */ \
        }                                                                                                  \
    } while (0)
#pragma unroll
    for (int j = 0; j < PF; ++j) pf[j] = uint4{0u, 0u, 0u, 0u};
    if ((long long)blockIdx.x < nblocks && ((long long)blockIdx.x + 1) * NT <= rows) UR_CE_FETCH((long long)blockIdx.x);
    for (long long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const long long r0 = b * NT;
        const int nrows = (int)((rows - r0) < NT ? (rows - r0) : NT);
        const int pieces = nrows == NT ? span16 : (nrows * row_bytes + 15) / 16;
        if (nrows == NT) {
#pragma unroll
            for (int j = 0; j < PF; ++j)
                if (j * NT < span16 && (int)threadIdx.x + j * NT < span16) stage[threadIdx.x + j * NT] = pf[j];
        } else {                                    // last span: element-wise, it may end inside a 16-byte piece
            for (int i = threadIdx.x; i < nrows * classes; i += NT) tile[i] = logits[r0 * classes + i];
        }
        __syncthreads();
        if (b + gridDim.x < nblocks && (b + gridDim.x + 1) * NT <= rows) UR_CE_FETCH(b + gridDim.x);
        const int rl = threadIdx.x;
        if (rl < nrows) {
            T* row = tile + rl * classes;
            const long long t = target[r0 + rl];
            const bool valid = t != ignore_index && t >= 0 && t < classes;
            if (!BWD) {
                float m = -INFINITY, sum = 0.f;
                if (classes <= 32) {
                    // the row in registers: 32 unrolled slots with a uniform test each, so the LDS reads are issued back to
                    // back (a loop with a run-time bound waits for every 2-byte read in turn) and the second pass reads nothing
                    float v[32];
#pragma unroll
                    for (int c = 0; c < 32; ++c) v[c] = c < classes ? ur_to_float(row[c]) : -INFINITY;
#pragma unroll
                    for (int c = 0; c < 32; ++c) m = fmaxf(m, v[c]);
#pragma unroll
                    for (int c = 0; c < 32; ++c)
                        if (c < classes) sum += ce_exp<T>(v[c] - m);
                } else {
                    for (int c = 0; c < classes; ++c) m = fmaxf(m, ur_to_float(row[c]));
                    for (int c = 0; c < classes; ++c) sum += ce_exp<T>(ur_to_float(row[c]) - m);
                }
                const float l = m + logf(sum);
                lse[r0 + rl] = l;
                if (valid) { nll += l - ur_to_float(row[t]); cnt += 1.f; }
            } else {
                const float l = lse[r0 + rl];
                for (int c = 0; c < classes; ++c) {
                    float g = 0.f;
                    if (valid) g = (ce_exp<T>(ur_to_float(row[c]) - l) - (c == t ? 1.f : 0.f)) * scale;
                    row[c] = ur_from_float<T>(g);
                }
            }
        }
        if (BWD) {
            __syncthreads();
            uint4* dst = reinterpret_cast<uint4*>(dlogits + r0 * classes);
            if (nrows == NT) {
                for (int i = threadIdx.x; i < pieces; i += NT) dst[i] = stage[i];
            } else {
                for (int i = threadIdx.x; i < nrows * classes; i += NT) dlogits[r0 * classes + i] = tile[i];
            }
        }
        __syncthreads();
    }
    if (!BWD) block_sum2(nll, cnt, partial + (size_t)blockIdx.x * 2);
#undef UR_CE_FETCH
}

// ce_staged_kernel<T, true> with the gradient rows widened to `dld` columns (zero fill) and the per-class sums of the
// stored gradient as a by-product: what the 1x1 head behind the loss needs next is exactly this pair -- its streaming
// kernels take channel counts that are multiples of 8 (19 classes -> 24 columns), and its bias gradient is the column
// sum.  Rows of at most 64 bytes in and out.  partial[block][32]: per-class sums of the block's rows.
constexpr int CEP_ROW = 64;                        // bytes per staged row, in and out

template <typename T>
__global__ __launch_bounds__(NT) void ce_bwd_padded_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                           long long rows, int classes, long long ignore_index,
                                                           const float* __restrict__ lse, const float* __restrict__ count,
                                                           const float* __restrict__ gout, T* __restrict__ dlogits, int dld,
                                                           float* __restrict__ partial) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    __shared__ uint4 stage_in[NT * CEP_ROW / 16];
    __shared__ u32x4_t stage_out[NT * CEP_ROW / 16];
    __shared__ float colred[NT / 64][32];
    T* tin = reinterpret_cast<T*>(stage_in);
    const int in16 = NT * classes * (int)sizeof(T) / 16;             // exact: 256 * anything
    const int out_pieces = dld * (int)sizeof(T) / 16;                // 16-byte pieces of a widened row (1 .. 4)
    const float scale = (gout ? gout[0] : 1.f) / count[0];
    // per-class sums of the rows this thread handles, in registers (the class loop is unrolled to the kernel's limit of 32
    // with a uniform test per class, so the arrays are never indexed at run time)
    float cs[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) cs[c] = 0.f;
    const long long nblocks = (rows + NT - 1) / NT;
    for (long long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const long long r0 = b * NT;
        const int nrows = (int)((rows - r0) < NT ? (rows - r0) : NT);
        if (nrows == NT) {
            const uint4* src = reinterpret_cast<const uint4*>(logits + r0 * classes);
            for (int i = threadIdx.x; i < in16; i += NT) stage_in[i] = src[i];
        } else {
            for (int i = threadIdx.x; i < nrows * classes; i += NT) tin[i] = logits[r0 * classes + i];
        }
        __syncthreads();
        const int rl = threadIdx.x;
        if (rl < nrows) {
            const T* row = tin + rl * classes;
            const long long t = target[r0 + rl];
            const bool valid = t != ignore_index && t >= 0 && t < classes;
            const float l = lse[r0 + rl];
            // the widened row is built in registers and goes to LDS as whole 16-byte pieces (rows 48 bytes apart put 16
            // consecutive lanes on all 64 banks), from where the span leaves as one contiguous run; stores of 16 bytes at a
            // 48-byte stride straight from the registers were three times slower
            union { T e[CEP_ROW / (int)sizeof(T)]; u32x4_t q[CEP_ROW / 16]; } o;
#pragma unroll
            for (int c = 0; c < CEP_ROW / (int)sizeof(T); ++c) {
                T gr = ur_from_float<T>(0.f);
                if (c < classes) {
                    float g = 0.f;
                    if (valid) g = (ce_exp<T>(ur_to_float(row[c]) - l) - (c == t ? 1.f : 0.f)) * scale;
                    gr = ur_from_float<T>(g);
                    cs[c] += ur_to_float(gr);
                }
                o.e[c] = gr;
            }
#pragma unroll
            for (int j = 0; j < CEP_ROW / 16; ++j)
                if (j < out_pieces) stage_out[rl * out_pieces + j] = o.q[j];
        }
        __syncthreads();
        if (nrows == NT) {
            u32x4_t* dst = reinterpret_cast<u32x4_t*>(dlogits + r0 * dld);
            for (int i = threadIdx.x; i < NT * out_pieces; i += NT) __builtin_nontemporal_store(stage_out[i], dst + i);
        } else {
            u32x4_t* dst = reinterpret_cast<u32x4_t*>(dlogits + r0 * dld);
            for (int i = threadIdx.x; i < nrows * out_pieces; i += NT) __builtin_nontemporal_store(stage_out[i], dst + i);
        }
    }
    // block sums: wave butterfly per class, then the four waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        if (c < classes) {
            const float t = ur_wave_sum(cs[c]);
            if (lane == 0) colred[wave][c] = t;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
        if ((int)threadIdx.x < classes)
            for (int wv = 0; wv < NT / 64; ++wv) t += colred[wv][threadIdx.x];
        partial[(size_t)blockIdx.x * 32 + threadIdx.x] = t;
    }
}

// colsum[c] = sum over blocks of partial[block][c], in double, fixed order
__global__ __launch_bounds__(256) void ce_colsum_finalize_kernel(const float* __restrict__ partial, int nparts, int classes,
                                                                 float* __restrict__ colsum) {
    __shared__ double red[8][32];
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
    double t = 0.0;
    int b = slice;
    for (; b + 56 < nparts; b += 64) {           // 8 independent loads in flight, added in block order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b + 8 * u) * 32 + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += (double)v[u];
    }
    for (; b < nparts; b += 8) t += (double)partial[(size_t)b * 32 + col];
    red[slice][col] = t;
    __syncthreads();
    if (threadIdx.x < classes) {
        double v = 0.0;
        for (int sl = 0; sl < 8; ++sl) v += red[sl][threadIdx.x];
        colsum[threadIdx.x] = (float)v;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void bce_fwd_kernel(const T* __restrict__ logits, const float* __restrict__ target,
                                                     long long count, float* __restrict__ partial) {
    float s = 0.f;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < count; i += (long long)gridDim.x * NT) {
        const float x = ur_to_float(logits[i]), t = target[i];
        s += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    }
    block_sum2(s, 0.f, partial + (size_t)blockIdx.x * 2);
}

template <typename T>
__global__ __launch_bounds__(NT) void bce_bwd_kernel(const T* __restrict__ logits, const float* __restrict__ target,
                                                     long long count, const float* __restrict__ gout,
                                                     T* __restrict__ dlogits) {
    const float scale = (gout ? gout[0] : 1.f) / (float)count;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < count; i += (long long)gridDim.x * NT) {
        const float x = ur_to_float(logits[i]);
        const float sg = 1.f / (1.f + expf(-x));
        dlogits[i] = ur_from_float<T>((sg - target[i]) * scale);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void argmax_kernel(const T* __restrict__ logits, long long rows, int classes, int ld,
                                                    int64_t* __restrict__ out) {
    for (long long r = blockIdx.x * (long long)NT + threadIdx.x; r < rows; r += (long long)gridDim.x * NT) {
        const T* row = logits + r * ld;
        float m = ur_to_float(row[0]);
        int best = 0;
        for (int c = 1; c < classes; ++c) {
            const float v = ur_to_float(row[c]);
            // strict '>' keeps the first maximum; a NaN beats every number and the first NaN wins (torch.argmax)
            if ((v > m) || (v != v && m == m)) { m = v; best = c; }
        }
        out[r] = best;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void sigmoid_mask_kernel(const T* __restrict__ logits, long long count,
                                                          uint8_t* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < count; i += (long long)gridDim.x * NT) {
        // evaluates sigmoid in fp32 exactly as written in the reference (tiny positive logits round to 0.5 -> false)
        const float sg = 1.f / (1.f + expf(-ur_to_float(logits[i])));
        out[i] = sg > 0.5f ? 1 : 0;
    }
}

// Confusion matrix of argmax(logits) against the targets: cm[t][p] += 1 for every pixel whose target is a class
// (ignore_index and out-of-range targets are skipped).  Workgroup-private counts in LDS (classes <= CM_LDS_CLASSES),
// then one 64-bit integer atomic per non-zero cell: integer sums, so the result does not depend on the order.
constexpr int CM_LDS_CLASSES = 32;
template <typename T>
__global__ __launch_bounds__(NT) void confusion_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                       long long rows, int classes, int ld, long long ignore_index,
                                                       unsigned long long* __restrict__ cm) {
    __shared__ unsigned int local[CM_LDS_CLASSES * CM_LDS_CLASSES];
    const bool use_lds = classes <= CM_LDS_CLASSES;
    if (use_lds) {
        for (int i = threadIdx.x; i < classes * classes; i += NT) local[i] = 0;
        __syncthreads();
    }
    for (long long r = blockIdx.x * (long long)NT + threadIdx.x; r < rows; r += (long long)gridDim.x * NT) {
        const long long t = target[r];
        if (t == ignore_index || t < 0 || t >= classes) continue;
        const T* row = logits + r * ld;
        float m = ur_to_float(row[0]);
        int best = 0;
        for (int c = 1; c < classes; ++c) {
            const float v = ur_to_float(row[c]);
            if ((v > m) || (v != v && m == m)) { m = v; best = c; }        // same tie / NaN rule as argmax_kernel
        }
        if (use_lds) atomicAdd(&local[(int)t * classes + best], 1u);
        else atomicAdd(&cm[(size_t)t * classes + best], 1ull);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < classes * classes; i += NT)
            if (local[i]) atomicAdd(&cm[i], (unsigned long long)local[i]);
    }
}

// Macro-averaged precision / recall / F1 / Jaccard and the accuracy of one confusion matrix, added to
// acc[0..6] = (precision, recall, accuracy, f1, iou, count of matrices, loss sum: += *loss when given).  The average runs over the labels that occur
// in the targets or in the predictions, a label with a zero denominator scores 0 (sklearn's average='macro',
// zero_division=0 as called at train_UResNetAblitionStudy.py:82-89).  One wave, double arithmetic, fixed order.
__global__ __launch_bounds__(64) void seg_metrics_kernel(const unsigned long long* __restrict__ cm, int classes,
                                                         const float* __restrict__ loss, double* __restrict__ acc) {
    double pr = 0.0, rc = 0.0, f1 = 0.0, io = 0.0, tp_all = 0.0, total = 0.0, labels = 0.0;
    for (int k = threadIdx.x; k < classes; k += 64) {
        double tp = (double)cm[(size_t)k * classes + k], row = 0.0, col = 0.0;
        for (int j = 0; j < classes; ++j) { row += (double)cm[(size_t)k * classes + j]; col += (double)cm[(size_t)j * classes + k]; }
        const double fn = row - tp, fp = col - tp;
        if (row + col > 0.0) {
            labels += 1.0;
            pr += (tp + fp) > 0.0 ? tp / (tp + fp) : 0.0;
            rc += (tp + fn) > 0.0 ? tp / (tp + fn) : 0.0;
            f1 += (2.0 * tp + fp + fn) > 0.0 ? 2.0 * tp / (2.0 * tp + fp + fn) : 0.0;
            io += (tp + fp + fn) > 0.0 ? tp / (tp + fp + fn) : 0.0;
        }
        tp_all += tp; total += row;
    }
    __shared__ double red[7][64];
    red[0][threadIdx.x] = pr; red[1][threadIdx.x] = rc; red[2][threadIdx.x] = f1; red[3][threadIdx.x] = io;
    red[4][threadIdx.x] = tp_all; red[5][threadIdx.x] = total; red[6][threadIdx.x] = labels;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double s[7];
    for (int q = 0; q < 7; ++q) { s[q] = 0.0; for (int l = 0; l < 64; ++l) s[q] += red[q][l]; }
    const double nl = s[6] > 0.0 ? s[6] : 1.0;
    acc[0] += s[0] / nl; acc[1] += s[1] / nl; acc[2] += s[5] > 0.0 ? s[4] / s[5] : 0.0;
    acc[3] += s[2] / nl; acc[4] += s[3] / nl; acc[5] += 1.0;
    if (loss) acc[6] += (double)loss[0];
}

int loss_blocks(long long n) { return ur_grid_1d(n, NT, 1024); }

// ---- ur_ce_fwd_noalloc: per-workgroup partial sums in a device array that belongs to the library -------------------
// One (NLL sum, valid count) pair per workgroup of the largest grid loss_blocks() hands out: 8 KiB per GPU, held from the
// moment the code object is loaded on that GPU until the process ends.  In host code `ce_scratch` is only a handle for
// hipGetSymbolAddress: each GPU has its own copy at its own address, resolved once per device and kept in ce_scratch_dev.
constexpr int CE_NOALLOC_MAX_BLOCKS = 1024;
constexpr long long CE_NOALLOC_MAX_ROWS = 1LL << 24;       // 64 rows per thread of the largest grid
constexpr int CE_MAX_DEVICES = 64;
static __device__ __align__(256) float ce_scratch[CE_NOALLOC_MAX_BLOCKS * 2];
static float* ce_scratch_dev[CE_MAX_DEVICES];               // device address of ce_scratch on GPU i (nullptr: not resolved yet)
static pthread_mutex_t ce_scratch_lock = PTHREAD_MUTEX_INITIALIZER;

// Device address of the current GPU's ce_scratch, or nullptr with the error string set.
float* ce_scratch_for_current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CE_MAX_DEVICES) {
        ur_set_error("ce_fwd_noalloc: cannot identify the current device (%d)", dev);
        return nullptr;
    }
    pthread_mutex_lock(&ce_scratch_lock);
    float* p = ce_scratch_dev[dev];
    if (!p) {
        void* addr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&addr, HIP_SYMBOL(ce_scratch));
        if (e != hipSuccess || !addr) {
            ur_set_error("ce_fwd_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
        } else {
            p = ce_scratch_dev[dev] = (float*)addr;
        }
    }
    pthread_mutex_unlock(&ce_scratch_lock);
    return p;
}

}  // namespace

extern "C" {

size_t ur_loss_workspace(long long n) { return (size_t)loss_blocks(n) * 2 * sizeof(float); }

int ur_ce_fwd(const void* logits, int dtype, const int64_t* target, long long rows, int classes, int ld,
              long long ignore_index, float* lse, float* out3, void* workspace, size_t ws_bytes, ur_stream_t s) {
    UR_REQUIRE(logits && target && lse && out3 && workspace && rows > 0 && classes > 0 && ld >= classes,
               "ce_fwd: bad arguments");
    UR_REQUIRE(ws_bytes >= ur_loss_workspace(rows), "ce_fwd: workspace too small");
    const int grid = loss_blocks(rows);
    float* partial = (float*)workspace;
    const int esz = dtype == UR_BF16 ? 2 : 4;
    if (ld == classes && classes * esz <= CE_STAGE_BYTES / NT && ((uintptr_t)logits % 16) == 0) {
        UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ce_staged_kernel<T, false>), dim3(grid), dim3(NT), 0, (hipStream_t)s,
                                                    (const T*)logits, target, rows, classes, ignore_index, lse, partial,
                                                    (const float*)nullptr, (const float*)nullptr, (T*)nullptr));
    } else
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ce_fwd_kernel<T>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const T*)logits,
                                                target, rows, classes, ld, ignore_index, lse, partial));
    if (int e = ur_check_launch("ce_fwd")) return e;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, partial, grid, out3, 0.0);
    return ur_check_launch("ce_finalize");
}

long long ur_ce_fwd_noalloc_max_rows(void) { return CE_NOALLOC_MAX_ROWS; }

int ur_ce_fwd_noalloc_prepare(void) { return ce_scratch_for_current_device() ? UR_OK : UR_ERR_INVALID; }

int ur_ce_fwd_noalloc(const void* logits, int dtype, const int64_t* target, long long rows, int classes, int ld,
                      long long ignore_index, float* lse, float* out3, ur_stream_t s) {
    UR_REQUIRE(logits && target && lse && out3 && rows > 0 && classes > 0 && ld >= classes, "ce_fwd_noalloc: bad arguments");
    UR_REQUIRE(rows <= CE_NOALLOC_MAX_ROWS, "ce_fwd_noalloc: %lld rows exceed the limit of %lld (use ur_ce_fwd with a workspace)",
               rows, CE_NOALLOC_MAX_ROWS);
    const int grid = loss_blocks(rows);
    UR_REQUIRE(grid <= CE_NOALLOC_MAX_BLOCKS, "ce_fwd_noalloc: grid of %d workgroups exceeds the scratch array", grid);
    float* partial = ce_scratch_for_current_device();
    if (!partial) return UR_ERR_INVALID;
    // every workgroup of the grid overwrites its own pair and the finalizer reads exactly `grid` pairs: whatever an
    // earlier call left in the array is never read, so nothing needs clearing
    return ur_ce_fwd(logits, dtype, target, rows, classes, ld, ignore_index, lse, out3, partial,
                     (size_t)grid * 2 * sizeof(float), s);
}

int ur_ce_bwd(const void* logits, int dtype, const int64_t* target, long long rows, int classes, int ld,
              long long ignore_index, const float* lse, const float* count, const float* gout, void* dlogits, int dld,
              ur_stream_t s) {
    UR_REQUIRE(logits && target && lse && count && dlogits && rows > 0 && classes > 0 && ld >= classes && dld >= classes,
               "ce_bwd: bad arguments");
    const int esz = dtype == UR_BF16 ? 2 : 4;
    if (ld == classes && dld == classes && classes * esz <= CE_STAGE_BYTES / NT && ((uintptr_t)logits % 16) == 0 &&
        ((uintptr_t)dlogits % 16) == 0) {
        UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ce_staged_kernel<T, true>), dim3(loss_blocks(rows)), dim3(NT), 0,
                                                    (hipStream_t)s, (const T*)logits, target, rows, classes, ignore_index,
                                                    const_cast<float*>(lse), (float*)nullptr, count, gout, (T*)dlogits));
        return ur_check_launch("ce_bwd");
    }
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ce_bwd_kernel<T>), dim3(loss_blocks(rows)), dim3(NT), 0, (hipStream_t)s,
                                                (const T*)logits, target, rows, classes, ld, ignore_index, lse, count, gout,
                                                (T*)dlogits, dld));
    return ur_check_launch("ce_bwd");
}

size_t ur_ce_bwd_padded_workspace(long long rows) { return (size_t)loss_blocks(rows) * 32 * sizeof(float); }

int ur_ce_bwd_padded_supported(int dtype, int classes, int dld) {
    const int esz = dtype == UR_BF16 ? 2 : (dtype == UR_F32 ? 4 : 0);
    return esz && classes > 0 && classes <= 32 && dld >= classes && classes * esz <= CEP_ROW && dld * esz <= CEP_ROW &&
           (dld * esz) % 16 == 0;
}

int ur_ce_bwd_padded(const void* logits, int dtype, const int64_t* target, long long rows, int classes,
                     long long ignore_index, const float* lse, const float* count, const float* gout, void* dlogits, int dld,
                     float* colsum, void* workspace, size_t ws_bytes, ur_stream_t s) {
    UR_REQUIRE(logits && target && lse && count && dlogits && colsum && workspace && rows > 0, "ce_bwd_padded: bad arguments");
    UR_REQUIRE(ur_ce_bwd_padded_supported(dtype, classes, dld),
               "ce_bwd_padded: needs at most 32 classes and rows of at most %d bytes whose padded length is a multiple of 16", CEP_ROW);
    UR_REQUIRE(((uintptr_t)logits % 16) == 0 && ((uintptr_t)dlogits % 16) == 0, "ce_bwd_padded: tensors must be 16-byte aligned");
    UR_REQUIRE(ws_bytes >= ur_ce_bwd_padded_workspace(rows), "ce_bwd_padded: workspace too small");
    const int grid = loss_blocks(rows);
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ce_bwd_padded_kernel<T>), dim3(grid), dim3(NT), 0, (hipStream_t)s,
                                                (const T*)logits, target, rows, classes, ignore_index, lse, count, gout,
                                                (T*)dlogits, dld, (float*)workspace));
    if (int e = ur_check_launch("ce_bwd_padded")) return e;
    hipLaunchKernelGGL(ce_colsum_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, (const float*)workspace, grid, classes,
                       colsum);
    return ur_check_launch("ce_colsum_finalize");
}

int ur_bce_fwd(const void* logits, int dtype, const float* target, long long count, float* out3, void* workspace,
               size_t ws_bytes, ur_stream_t s) {
    UR_REQUIRE(logits && target && out3 && workspace && count > 0, "bce_fwd: bad arguments");
    UR_REQUIRE(ws_bytes >= ur_loss_workspace(count), "bce_fwd: workspace too small");
    const int grid = loss_blocks(count);
    float* partial = (float*)workspace;
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bce_fwd_kernel<T>), dim3(grid), dim3(NT), 0, (hipStream_t)s, (const T*)logits,
                                                target, count, partial));
    if (int e = ur_check_launch("bce_fwd")) return e;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, partial, grid, out3, (double)count);
    return ur_check_launch("bce_finalize");
}

int ur_bce_bwd(const void* logits, int dtype, const float* target, long long count, const float* gout, void* dlogits,
               ur_stream_t s) {
    UR_REQUIRE(logits && target && dlogits && count > 0, "bce_bwd: bad arguments");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bce_bwd_kernel<T>), dim3(loss_blocks(count)), dim3(NT), 0, (hipStream_t)s,
                                                (const T*)logits, target, count, gout, (T*)dlogits));
    return ur_check_launch("bce_bwd");
}

int ur_argmax_channels(const void* logits, int dtype, long long rows, int classes, int ld, int64_t* out, ur_stream_t s) {
    UR_REQUIRE(logits && out && rows > 0 && classes > 0 && ld >= classes, "argmax: bad arguments");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((argmax_kernel<T>), dim3(ur_grid_1d(rows, NT)), dim3(NT), 0, (hipStream_t)s,
                                                (const T*)logits, rows, classes, ld, out));
    return ur_check_launch("argmax");
}

int ur_confusion_matrix(const void* logits, int dtype, const int64_t* target, long long rows, int classes, int ld,
                        long long ignore_index, unsigned long long* cm, ur_stream_t s) {
    UR_REQUIRE(logits && target && cm && rows > 0 && classes > 0 && ld >= classes, "confusion_matrix: bad arguments");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((confusion_kernel<T>), dim3(ur_grid_1d(rows, NT * 4, 2048)), dim3(NT), 0,
                                                (hipStream_t)s, (const T*)logits, target, rows, classes, ld, ignore_index, cm));
    return ur_check_launch("confusion_matrix");
}

int ur_seg_metrics_accumulate(const unsigned long long* cm, int classes, const float* loss, double* acc7, ur_stream_t s) {
    UR_REQUIRE(cm && acc7 && classes > 0, "seg_metrics: bad arguments");
    hipLaunchKernelGGL(seg_metrics_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, cm, classes, loss, acc7);
    return ur_check_launch("seg_metrics");
}

int ur_sigmoid_mask(const void* logits, int dtype, long long count, uint8_t* out, ur_stream_t s) {
    UR_REQUIRE(logits && out && count > 0, "sigmoid_mask: bad arguments");
    UR_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sigmoid_mask_kernel<T>), dim3(ur_grid_1d(count, NT)), dim3(NT), 0,
                                                (hipStream_t)s, (const T*)logits, count, out));
    return ur_check_launch("sigmoid_mask");
}

}  // extern "C"
