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
        for (int c = 0; c < classes; ++c) s += expf(ur_to_float(row[c]) - m);
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
            if (valid) g = (expf(ur_to_float(row[c]) - l) - (c == t ? 1.f : 0.f)) * scale;
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
    for (long long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const long long r0 = b * NT;
        const int nrows = (int)((rows - r0) < NT ? (rows - r0) : NT);
        const int pieces = nrows == NT ? span16 : (nrows * row_bytes + 15) / 16;
        const uint4* src = reinterpret_cast<const uint4*>(logits + r0 * classes);
        if (nrows == NT) {
            for (int i = threadIdx.x; i < pieces; i += NT) stage[i] = src[i];
        } else {                                    // last span: element-wise, it may end inside a 16-byte piece
            for (int i = threadIdx.x; i < nrows * classes; i += NT) tile[i] = logits[r0 * classes + i];
        }
        __syncthreads();
        const int rl = threadIdx.x;
        if (rl < nrows) {
            T* row = tile + rl * classes;
            const long long t = target[r0 + rl];
            const bool valid = t != ignore_index && t >= 0 && t < classes;
            if (!BWD) {
                float m = -INFINITY;
                for (int c = 0; c < classes; ++c) m = fmaxf(m, ur_to_float(row[c]));
                float sum = 0.f;
                for (int c = 0; c < classes; ++c) sum += expf(ur_to_float(row[c]) - m);
                const float l = m + logf(sum);
                lse[r0 + rl] = l;
                if (valid) { nll += l - ur_to_float(row[t]); cnt += 1.f; }
            } else {
                const float l = lse[r0 + rl];
                for (int c = 0; c < classes; ++c) {
                    float g = 0.f;
                    if (valid) g = (expf(ur_to_float(row[c]) - l) - (c == t ? 1.f : 0.f)) * scale;
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
