/* liburesnet_hip.so -- C ABI of the MI355X (gfx950) U-ResNet hot path.
 *
 * The reference (Nanlingxq/U-ResNet) is pure PyTorch: it has no FFI of its own, every
 * "kernel" is an ATen dispatch issued by a torch.nn layer.  Each entry point below
 * therefore cites the reference line whose ATen op it replaces.  Paths are relative to
 * the reference root; S/C = Segmentation/Cityscapes/modules/U_ResNet.py,
 * C/C = Classification/modules/U_ResNet_CIFAR.py.
 *
 * Conventions
 *  - Every activation tensor is NHWC ("channels last") in device memory.  `ld` arguments
 *    are the pixel pitch in elements (>= channels), so a channel slice of a wider concat
 *    buffer can be read or written in place.
 *  - dtype selects the activation storage type (UR_F32 or UR_BF16).  Parameters, their
 *    gradients, batch-norm statistics and loss values are always fp32.
 *  - The caller owns every buffer including workspaces; the library allocates nothing.  Its only
 *    mutable state is the thread-local error string and, for ur_ce_fwd_noalloc, one 8 KiB device
 *    array per GPU inside the code object with a mutex-guarded table of its per-GPU addresses
 *    (both private to the library).
 *  - All work is enqueued on the given stream (hipStream_t passed as void*); no call
 *    synchronises the device.
 *  - Return value: 0 on success, a negative UR_ERR_* code otherwise; ur_last_error()
 *    returns the message of the calling thread's last failure.
 */
#ifndef URESNET_HIP_H
#define URESNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UR_OK 0
#define UR_ERR_INVALID (-1)
#define UR_ERR_LAUNCH (-2)

#define UR_F32 0
#define UR_BF16 1

typedef void* ur_stream_t;

int ur_version(void);
const char* ur_last_error(void);

/* ---------------------------------------------------------------- dense convolution
 * Geometry of one nn.Conv2d(groups=1) call.  Replaces aten::convolution /
 * convolution_backward issued by S/C:94,97 (DoubleConv), S/C:140,143,151 (BasicBlock),
 * S/C:230 (7x7 stem), S/C:50,79,132 (1x1 heads), C/C:550 (5x5 convert), and nn.Linear
 * (C/C:549, as a 1x1 convolution on a 1x1 map). */
typedef struct ur_conv_desc {
    int dtype;                 /* UR_F32 | UR_BF16 */
    int n, h, w, cin;          /* input  NHWC */
    int oh, ow, cout;          /* output NHWC */
    int kh, kw, stride, pad;
    int in_ld, out_ld;         /* pixel pitch of the input / output buffers (elements) */
} ur_conv_desc;

#define UR_PACK_FWD 0          /* OIHW fp32 -> [cout][Kpad], k = (ky,kx,ci)              */
#define UR_PACK_DGRAD 1        /* OIHW fp32 -> [cin ][Kpad], k = (ky,kx,co)              */
#define UR_PACK_CONVT_FWD 2    /* (cin,cout,2,2) fp32 -> [(dy,dx,co)][Kpad], k = ci      */
#define UR_PACK_CONVT_DGRAD 3  /* (cin,cout,2,2) fp32 -> [ci][Kpad], k = (dy,dx,co)      */

/* number of elements (of the activation dtype) of a packed weight matrix */
size_t ur_conv2d_packed_elems(const ur_conv_desc* d, int mode);
int ur_conv2d_pack_weight(const float* w, void* packed, const ur_conv_desc* d, int mode, ur_stream_t s);

/* y = conv(x) (+bias).  If stats != NULL the kernel also emits per-channel partial sums
 * (sum, sum of squares) of the fp32 results: stats[ur_conv2d_stats_rows(d)][cout][2]. */
int ur_conv2d_stats_rows(const ur_conv_desc* d);
int ur_conv2d_fwd(const void* x, const void* w_packed_fwd, const float* bias, void* y, float* stats,
                  const ur_conv_desc* d, ur_stream_t s);
/* dx = conv_backward_input(dy) */
int ur_conv2d_dgrad(const void* dy, const void* w_packed_dgrad, void* dx, const ur_conv_desc* d, ur_stream_t s);
/* dw (fp32, OIHW) = conv_backward_weight(x, dy); deterministic split-K through `workspace` */
size_t ur_conv2d_wgrad_workspace(const ur_conv_desc* d);
int ur_conv2d_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes,
                    const ur_conv_desc* d, ur_stream_t s);

/* ---------------------------------------------------------------- 1x1 / stride 1, bf16: streaming kernel
 * Same results as ur_conv2d_fwd (without statistics) / ur_conv2d_dgrad for the pointwise layers that are bound by
 * HBM, not by arithmetic (EUCB projection S/C:53-55, segmentation head S/C:127-133 and their gradients).  The
 * weight is the fp32 parameter itself, [w_cout][w_cin] (no packing step); the tensors may carry more channels than
 * the weight (zero channel padding up to a multiple of 8) and may be channel windows (in_ld / out_ld).
 * Needs 8 | reduced channels and pitch; ur_conv1x1_pw_supported tells whether the register budget covers the shape. */
int ur_conv1x1_pw_supported(const ur_conv_desc* d, int dgrad);
int ur_conv1x1_pw_fwd(const void* x, const float* w, int w_cout, int w_cin, const float* bias, void* y,
                      const ur_conv_desc* d, ur_stream_t s);
int ur_conv1x1_pw_dgrad(const void* dy, const float* w, int w_cout, int w_cin, void* dx, const ur_conv_desc* d,
                        ur_stream_t s);
/* backward-weight of the same layers (same contract as ur_conv2d_wgrad: dw fp32 [cout][cin], fixed summation order):
 * a stream over the pixels with the 16..32 x 32..96 channel product on the matrix cores, instead of the generic GEMM's
 * 128 x 128 tiles.  Needs 8 | channels and pitches, cout <= 32, cin <= 96. */
int ur_conv1x1_pw_wgrad_supported(const ur_conv_desc* d);
size_t ur_conv1x1_pw_wgrad_workspace(const ur_conv_desc* d);
int ur_conv1x1_pw_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes, const ur_conv_desc* d,
                        ur_stream_t s);
/* The same two kernels for a 1x1 convolution whose input is BatchNorm(+ReLU) of a stored tensor (the segmentation head
 * S/C:127-133 BN -> ReLU -> conv, EUCB* S/C:80-85): x is the BatchNorm's INPUT and in_aff = [2][cin] per-channel scale then
 * shift (scale = gamma * invstd, shift = beta - mean * scale, as ur_bn_finalize_affine writes them); every 16-byte piece becomes
 * bf16(max(x * scale + shift, 0)) (without max when relu == 0) on its way into the kernel's LDS rows, so the normalised
 * tensor is never written or read. */
int ur_conv1x1_pw_fwd_affine(const void* x, const float* in_aff, int in_relu, const float* w, int w_cout, int w_cin,
                             const float* bias, void* y, const ur_conv_desc* d, ur_stream_t s);
int ur_conv1x1_pw_wgrad_affine(const void* x, const float* x_aff, int x_relu, const void* dy, float* dw, void* workspace,
                               size_t ws_bytes, const ur_conv_desc* d, ur_stream_t s);
/* The same three kernels for a 1x1 convolution over a channel concatenation that is never materialised (the Feature
 * Merge S/C:280 torch.cat([U_out, Res_out], 1) followed by the head S/C:127-133): logical input channels [0, ca) live in the
 * dense tensor xa (pitch ld_a, 0 = ca), channels [ca, ca + cb) in xb (pitch ld_b); ca + cb = d->cin, 8 | ca, cb and pitches;
 * aff_a = [2][ca], aff_b = [2][cb] as in_aff above.  Backward-data writes the two gradients into two dense tensors.  A
 * 64-channel tensor written as a window of an 80-channel one is 128-byte runs at a 160-byte pitch, i.e. partial lines in
 * every pass that touches the window; two dense tensors stream at full rate.  d->in_ld is ignored. */
int ur_conv1x1_pw_fwd_affine_pair(const void* xa, int ca, int ld_a, const void* xb, int cb, int ld_b, const float* aff_a,
                                  const float* aff_b, int in_relu, const float* w, int w_cout, int w_cin, const float* bias, void* y,
                                  const ur_conv_desc* d, ur_stream_t s);
int ur_conv1x1_pw_dgrad_pair(const void* dy, const float* w, int w_cout, int w_cin, void* dxa, int ca, int ld_a, void* dxb, int cb,
                             int ld_b, const ur_conv_desc* d, ur_stream_t s);
int ur_conv1x1_pw_wgrad_affine_pair(const void* xa, int ca, int ld_a, const void* xb, int cb, int ld_b, const float* aff_a,
                                    const float* aff_b, int x_relu, const void* dy, float* dw, void* workspace, size_t ws_bytes,
                                    const ur_conv_desc* d, ur_stream_t s);
/* ur_conv1x1_pw_wgrad_affine_pair for a head whose input is relu(BatchNorm(cat)) in training mode (S/C:127-133 over S/C:280),
 * which also returns that BatchNorm's backward sums, so that ur_bn_act_bwd_reduce over the input gradient is not needed.
 * The kernel multiplies dy with the ReLU mask and with mask * xhat instead of with the activated input:
 *   S1[co][c] = sum_p dy[p][co] open[p][c], S2[co][c] = sum_p dy[p][co] open[p][c] xhat[p][c];
 *   dw[co][c] = gamma[c] S2 + beta[c] S1 ([w_cout][ca + cb], may be NULL);
 *   sums[c] = (sum_co W[co][c] S1, sum_co W[co][c] S2) = (sum g, sum g xhat), also written to dbeta / dgamma (may be NULL),
 * with W rounded to bf16 as ur_conv1x1_pw_dgrad_pair used it.  mean, invstd, gamma, beta: [ca + cb] fp32; aff_a / aff_b: the
 * folded constants of the forward call (the mask is the forward's); w: [w_cout][ca + cb] fp32. */
/* Backward-data of such a head with the BatchNorm's second backward pass in its epilogue: the product (the gradient of the
 * activated tensor) is never stored; what is stored is the gradient of the BatchNorm's INPUT v,
 *   dv = gamma invstd (g' - s0 - xhat s1),  g' = the product where the activation passed, (s0, s1) = sums[c] / (n h w)
 * (ur_bn_act_bwd_apply's value on the bf16-rounded product, to fp32 rounding), read from va / vb and written to dva / dvb
 * (two dense tensors of ca and cb channels, pitches ld_a / ld_b, 0 = dense; vb = dvb = NULL: one tensor).  sums = what
 * ur_conv1x1_pw_wgrad_bn[_pair] returned: call that first.  Saves writing and re-reading the activated tensor's gradient. */
int ur_conv1x1_pw_dgrad_bn_pair(const void* dy, const float* w, int w_cout, int w_cin, const void* va, void* dva, int ca, int ld_a,
                                const void* vb, void* dvb, int cb, int ld_b, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, const float* sums, const ur_conv_desc* d, ur_stream_t s);
size_t ur_conv1x1_pw_wgrad_bn_workspace(const ur_conv_desc* d);
/* the same over one input tensor (x, pitch d->in_ld): the backward of ur_conv1x1_pw_fwd_affine in training mode (the EUCB*
 * projection behind its last BatchNorm+ReLU, S/C:84-86) */
int ur_conv1x1_pw_wgrad_bn(const void* x, const float* x_aff, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, const void* dy, const float* w, int w_cout, float* dw, float* sums, float* dgamma,
                           float* dbeta, void* workspace, size_t ws_bytes, const ur_conv_desc* d, ur_stream_t s);
int ur_conv1x1_pw_wgrad_bn_pair(const void* xa, int ca, int ld_a, const void* xb, int cb, int ld_b, const float* aff_a,
                                const float* aff_b, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                const void* dy, const float* w, int w_cout, float* dw, float* sums, float* dgamma, float* dbeta,
                                void* workspace, size_t ws_bytes, const ur_conv_desc* d, ur_stream_t s);

/* ---------------------------------------------------------------- 3x3 / stride 1 / pad 1, bf16: LDS-tiled kernel
 * Same contract as ur_conv2d_fwd / ur_conv2d_dgrad for the layers that carry ~94 % of the FLOPs
 * (S/C:94,97,140,143 with 32..1024 channels): needs 32 | produced channels (computed in blocks of 64), 32 | reduced channels.
 * Weights are packed in MFMA-fragment order (dgrad != 0: tap-flipped and transposed). */
int ur_conv3x3_tile_supported(const ur_conv_desc* d, int dgrad);
size_t ur_conv3x3_tile_packed_elems(const ur_conv_desc* d);
int ur_conv3x3_tile_pack_weight(const float* w, void* packed, const ur_conv_desc* d, int dgrad, ur_stream_t s);
/* the same for many layers in one launch (after an optimiser step every layer's image is stale at once):
 * table = device array of `entries` rows {w, packed, cin, cout, dgrad (bit 1: piece order of ur_conv3x3_wide_*), first_block},
 * six 64-bit words each, where
 * first_block = the sum of ur_conv3x3_tile_pack_blocks over the rows before it; total_blocks = that sum over all rows */
long long ur_conv3x3_tile_pack_blocks(int cin, int cout, int dgrad);
int ur_conv3x3_tile_pack_weights_multi(const long long* table, int entries, long long total_blocks, ur_stream_t s);
int ur_conv3x3_tile_stats_rows(const ur_conv_desc* d);
int ur_conv3x3_tile_fwd(const void* x, const void* w_packed, void* y, float* stats, const ur_conv_desc* d,
                        ur_stream_t s);
int ur_conv3x3_tile_dgrad(const void* dy, const void* w_packed_dgrad, void* dx, const ur_conv_desc* d,
                          ur_stream_t s);
/* Inference with fixed BatchNorm statistics folded into the convolution's epilogue (model.eval(), S/C:95-99,141-158):
 * y = act(conv(x) * scale[c] + shift[c] (+ residual)), scale = gamma * invstd, shift = beta - mean * scale
 * (ur_bn_eval_affine).  No intermediate tensor, no BatchNorm pass.  relu != 0 clamps at zero; residual may be NULL. */
int ur_conv3x3_tile_fwd_affine(const void* x, const void* w_packed, void* y, const float* scale, const float* shift, int relu,
                               const void* residual, int res_ld, const ur_conv_desc* d, ur_stream_t s);
/* dx = round(conv_backward_input(dy)) + addend: the input of a residual block (S/C:156-159) has a second gradient
 * from the shortcut; it is added in the epilogue (same rounding as a separate add of the two bf16 tensors).
 * add_ld = channel pitch of addend in elements (0 or cin: dense). */
int ur_conv3x3_tile_dgrad_add(const void* dy, const void* w_packed_dgrad, void* dx, const void* addend, int add_ld,
                              const ur_conv_desc* d, ur_stream_t s);
/* Backward-data that also starts the backward pass of the BatchNorm(+ReLU) whose output the input tensor was
 * (S/C:94-99, 140-144: conv -> BN -> ReLU -> conv).  dx (+ addend, as ur_conv3x3_tile_dgrad_add) is the gradient dz of
 * that BatchNorm's output (the activation is ReLU); while the tile is still on chip the kernel reads the BatchNorm's input bn_y (and, for a
 * residual block, its activated output bn_z, else NULL: the ReLU mask is rebuilt from bn_y) and writes per-tile partial
 * sums partial[ur_conv3x3_tile_dgrad_stats_rows(d)][cin][2] = (sum g, sum g * x_hat), g = dz where the activation
 * passed.  Folded by ur_bn_bwd_finalize they replace ur_bn_act_bwd_reduce: one pass over dz and bn_y less. */
int ur_conv3x3_tile_dgrad_stats_rows(const ur_conv_desc* d);
int ur_conv3x3_tile_dgrad_bnsums(const void* dy, const void* w_packed_dgrad, void* dx, const void* addend, int add_ld,
                                 const void* bn_y, int bn_y_ld, const void* bn_z, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, float* partial, const ur_conv_desc* d, ur_stream_t s);

/* ---------------------------------------------------------------- 3x3 / stride 1 / pad 1, bf16: 128-channel kernel
 * Same contract and results (to bf16 rounding of a differently ordered fp32 sum) as ur_conv3x3_tile_fwd / _dgrad for the
 * layers of S/C:94,97,140,143 that produce a multiple of 128 channels and reduce over a multiple of 32: one persistent
 * 8-wave workgroup per CU, operands staged by LDS-DMA, 16x16x32 MFMA.  Its weight image has its own piece order
 * (ur_conv3x3_wide_pack_weight; same size as the tile kernel's; a row of ur_conv3x3_tile_pack_weights_multi builds it
 * when bit 1 of the row's dgrad word is set).  ur_conv3x3_wide_preferred tells whether it is the faster of the two for
 * the shape.  stats: [ur_conv3x3_wide_stats_rows(d)][cout][2] per-tile partial sums as for the tile kernel. */
int ur_conv3x3_wide_supported(const ur_conv_desc* d, int dgrad);
int ur_conv3x3_wide_preferred(const ur_conv_desc* d, int dgrad);
size_t ur_conv3x3_wide_packed_elems(const ur_conv_desc* d);
int ur_conv3x3_wide_pack_weight(const float* w, void* packed, const ur_conv_desc* d, int dgrad, ur_stream_t s);
int ur_conv3x3_wide_stats_rows(const ur_conv_desc* d);
int ur_conv3x3_wide_fwd(const void* x, const void* w_packed, void* y, float* stats, const ur_conv_desc* d, ur_stream_t s);
int ur_conv3x3_wide_dgrad(const void* dy, const void* w_packed_dgrad, void* dx, const ur_conv_desc* d, ur_stream_t s);
/* as ur_conv3x3_tile_dgrad_add, and as ur_conv3x3_tile_dgrad_bnsums without addend and with the ReLU mask rebuilt from bn_y
 * (the residual-block forms, with bn_z or an addend, stay with the tile kernel);
 * partial[ur_conv3x3_wide_stats_rows(d)][cin][2] */
int ur_conv3x3_wide_dgrad_add(const void* dy, const void* w_packed_dgrad, void* dx, const void* addend, int add_ld,
                              const ur_conv_desc* d, ur_stream_t s);
int ur_conv3x3_wide_dgrad_bnsums(const void* dy, const void* w_packed_dgrad, void* dx, const void* bn_y, int bn_y_ld,
                                 const float* mean, const float* invstd, const float* gamma, const float* beta, float* partial,
                                 const ur_conv_desc* d, ur_stream_t s);

/* backward-weight counterpart (same contract as ur_conv2d_wgrad; needs 32 | cin and 32 | cout: 64 x 64-channel
 * blocks when both are multiples of 64, 32 x 32 blocks otherwise) */
int ur_wgrad3x3_tile_supported(const ur_conv_desc* d);
size_t ur_wgrad3x3_tile_workspace(const ur_conv_desc* d);
int ur_wgrad3x3_tile(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes,
                     const ur_conv_desc* d, ur_stream_t s);
/* The same call for measurement harnesses: the library records `event_after_main` (a hipEvent_t, or NULL) on `s` between
 * its two kernels, the matrix-core kernel that writes the partial slabs and the HBM-bound kernel that sums them, so that
 * a caller bracketing the call with two events of its own gets each kernel's duration.  Only for layers where
 * ur_wgrad3x3_tile_supported(d) == 1 (the 32 x 32-block kernel rejects a non-NULL event).  No counterpart in the reference. */
int ur_wgrad3x3_tile_marked(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes,
                            const ur_conv_desc* d, void* event_after_main, ur_stream_t s);

/* ---------------------------------------------------------------- ConvTranspose2d(k=2, s=2) + bias
 * Replaces aten::conv_transpose2d of `Up.up` (S/C:113,117).  d describes the transposed
 * convolution itself: input (n,h,w,cin) -> output (n,2h,2w,cout); kh=kw=stride=2, pad=0.
 * Weight layout is PyTorch's (cin, cout, 2, 2). */
int ur_convT2x2_fwd(const void* x, const void* w_packed, const float* bias, void* y, const ur_conv_desc* d,
                    ur_stream_t s);
int ur_convT2x2_dgrad(const void* dy, const void* w_packed_dgrad, void* dx, const ur_conv_desc* d, ur_stream_t s);
size_t ur_convT2x2_wgrad_workspace(const ur_conv_desc* d);
int ur_convT2x2_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes,
                      const ur_conv_desc* d, ur_stream_t s);

/* per-channel column sum of an NHWC tensor: out[c] = sum_rows x[row][c] (bias gradients of
 * S/C:50,79,113,132).  workspace: ur_colsum_workspace(rows, c) bytes. */
size_t ur_colsum_workspace(long long rows, int c);
int ur_colsum(const void* x, int dtype, long long rows, int c, int ld, float* out, void* workspace,
              size_t ws_bytes, ur_stream_t s);

/* ---------------------------------------------------------------- grouped / depthwise 3x3 (stride 1, pad 1)
 * EUCB depthwise conv (S/C:44, groups=C, cpg=1) and EUCB* grouped conv (S/C:67,73,
 * groups=C/4, cpg=4).  Weights stay in PyTorch layout (C, cpg, 3, 3) fp32. */
int ur_gconv3x3_fwd(const void* x, const float* w, void* y, int dtype, int n, int h, int w_, int c, int cpg,
                    ur_stream_t s);
/* the same with BatchNorm partial sums from the kernel's epilogue (EUCB* stages, S/C:69-81: grouped conv -> BN):
 * stats[ur_gconv3x3_stats_rows(...)][c][2]; rows == 0 means this shape has no such kernel (use ur_bn_stats) */
int ur_gconv3x3_stats_rows(int dtype, int n, int h, int w_, int c, int cpg);
int ur_gconv3x3_fwd_stats(const void* x, const float* w, void* y, float* stats, int dtype, int n, int h, int w_, int c, int cpg,
                          ur_stream_t s);
int ur_gconv3x3_dgrad(const void* dy, const float* w, void* dx, int dtype, int n, int h, int w_, int c, int cpg,
                      ur_stream_t s);
size_t ur_gconv3x3_wgrad_workspace(int n, int h, int w_, int c, int cpg);
int ur_gconv3x3_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes, int dtype,
                      int n, int h, int w_, int c, int cpg, ur_stream_t s);

/* ---------------------------------------------------------------- BatchNorm2d (+ReLU, +residual add)
 * Replaces native_batch_norm / relu_ / add_ of S/C:95-96,98-99,141-142,144,152,157-158,
 * 130-131 and their backward.  Training statistics follow PyTorch: biased variance for
 * normalisation, unbiased for running_var, running = (1-momentum)*running + momentum*batch. */
size_t ur_bn_stats_rows(long long rows, int c);          /* rows of partials ur_bn_stats writes */
int ur_bn_stats(const void* y, int dtype, long long rows, int c, float* partial, ur_stream_t s);
/* reduce partial[nparts][c][2] -> mean, invstd; update running stats (and increment the
 * num_batches_tracked counter) where the pointers are non-NULL.  `partial` is used as scratch
 * (long partial lists are folded in place first). */
int ur_bn_finalize(float* partial, int nparts, long long count, int c, float eps, float momentum,
                   float* mean, float* invstd, float* running_mean, float* running_var,
                   long long* num_batches_tracked, ur_stream_t s);
/* ur_bn_finalize that also writes the normalisation as one multiply-add per element, aff = [2][c]: scale = gamma * invstd,
 * then shift = beta - mean * scale (the constants ur_conv1x1_pw_fwd_affine / _wgrad_affine take) */
int ur_bn_finalize_affine(float* partial, int nparts, long long count, int c, float eps, float momentum, float* mean,
                          float* invstd, float* running_mean, float* running_var, long long* num_batches_tracked,
                          const float* gamma, const float* beta, float* aff, ur_stream_t s);
/* eval mode: mean = running_mean, invstd = rsqrt(running_var + eps) */
int ur_bn_eval_prepare(const float* running_mean, const float* running_var, int c, float eps, float* mean,
                       float* invstd, ur_stream_t s);
/* eval mode folded to one multiply-add per channel: scale = gamma * rsqrt(running_var + eps),
 * shift = beta - running_mean * scale */
int ur_bn_eval_affine(const float* running_mean, const float* running_var, const float* gamma, const float* beta, int c,
                      float eps, float* scale, float* shift, ur_stream_t s);
/* z = act((y-mean)*invstd*gamma + beta (+ residual));  act: 0 none, 1 relu.  z_ld = channel pitch of z in
 * elements (0 or c: dense); a larger pitch writes into a channel window of a wider NHWC tensor (the skip half
 * of a torch.cat, S/C:119-124) */
int ur_bn_act_fwd(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  const void* residual, int act, void* z, int dtype, long long rows, int c, int z_ld,
                  ur_stream_t s);
/* ur_bn_act_fwd (no residual) and ur_maxpool2x2_fwd of its result in one pass, for an encoder stage whose output feeds both
 * the skip connection and the next stage's MaxPool2d(2, 2) (S/C:105, S/C:268-276): z (pitch z_ld, 0 = dense), pooled
 * [n][h/2][w/2][c] and idx (one position byte per pooled element) are bit for bit what the two calls produce; the pooling
 * pass no longer reads z back.  bf16, even h and w, c a power of two in [8, 2048]. */
int ur_bn_act_fwd_pool_supported(int dtype, int n, int h, int w, int c);
int ur_bn_act_fwd_pool(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                       void* z, int z_ld, void* pooled, uint8_t* idx, int dtype, int n, int h, int w, int c, ur_stream_t s);
/* ur_bn_act_fwd (dense result, no residual) that also returns ur_bn_stats of its result, for an activation that a second
 * BatchNorm normalises next (the last stage of the U-Net branch feeding the head's BatchNorm, S/C:283-289 -> S/C:127-133):
 * partial[ur_bn_act_fwd_stats_rows(rows, c)][c][2] = per-workgroup (sum z, sum z^2) over the rounded values, ready for
 * ur_bn_finalize[_affine]; z is bit for bit ur_bn_act_fwd's.  bf16, c a multiple of 8 that divides 2048. */
int ur_bn_act_fwd_stats_supported(int dtype, long long rows, int c);
size_t ur_bn_act_fwd_stats_rows(long long rows, int c);
int ur_bn_act_fwd_stats(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                        void* z, int dtype, long long rows, int c, float* partial, ur_stream_t s);
/* backward pass 1: partial[nparts][c][2] = (sum g, sum g*xhat), g = dz masked by the ReLU (act != 0).
 * The mask comes from the saved output z when z != NULL, else it is recomputed from y (needs gamma, beta;
 * valid only when no residual was added before the activation).  dz_ld = channel pitch of dz in elements
 * (0 or c: dense); a larger pitch reads a channel window of a wider NHWC gradient (the backward of a
 * torch.cat, S/C:119-124, S/C:280, without the split copy). */
int ur_bn_act_bwd_reduce(const void* dz, const void* y, const void* z, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, int act, int dtype, long long rows, int c, int dz_ld,
                         float* partial, ur_stream_t s);
/* Backward of MaxPool2d(2, 2) (S/C:103-107) whose input is a BatchNorm+ReLU output (the encoder stage's result, which also
 * feeds the skip connection, S/C:268-276): ur_maxpool2x2_bwd (dz = scattered pooled gradient + optional addend of pitch
 * add_ld) and ur_bn_act_bwd_reduce (mask rebuilt from bn_y) in one pass.  dz is stored as ur_maxpool2x2_bwd stores it and the
 * partial rows (ur_bn_stats_rows(n*h*w, c) of them) equal bit for bit what ur_bn_act_bwd_reduce computes from that dz. */
int ur_maxpool2x2_bwd_bnsums_supported(int dtype, int n, int h, int w, int c);
int ur_maxpool2x2_bwd_bnsums(const void* dpool, const uint8_t* idx, void* dz, const void* addend, int add_ld, const void* bn_y,
                             const float* mean, const float* invstd, const float* gamma, const float* beta, float* partial,
                             int dtype, int n, int h, int w, int c, ur_stream_t s);
/* sums[c][2] = reduce(partial); dgamma = sums[.,1]; dbeta = sums[.,0]; `partial` is used as scratch */
int ur_bn_bwd_finalize(float* partial, int nparts, int c, float* sums, float* dgamma, float* dbeta,
                       ur_stream_t s);
/* backward pass 2: dy = gamma*invstd*(g - sums0/count - xhat*sums1/count); dres = g (optional).
 * eval != 0 treats the statistics as constants: dy = gamma*invstd*g.  z may be NULL as in pass 1. */
int ur_bn_act_bwd_apply(const void* dz, const void* y, const void* z, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, const float* sums, int act, int eval, void* dy,
                        void* dres, int dtype, long long rows, int c, int dz_ld, ur_stream_t s);

/* ---------------------------------------------------------------- spatial ops
 * MaxPool2d(2,2) (S/C:105), bilinear Upsample(scale, align_corners=False) (S/C:43,66,362;
 * C/C:303), F.avg_pool2d(k) (C/C:572), F.pad + torch.cat (S/C:121-124, S/C:280). */
/* x_ld: channel pitch of x in elements (0 or c: dense) -- x may be a channel window of a wider NHWC tensor */
int ur_maxpool2x2_fwd(const void* x, void* y, uint8_t* idx, int dtype, int n, int h, int w, int c, int x_ld,
                      ur_stream_t s);
/* dx = scatter(dy) (+ addend): `addend` (NULL: none; add_ld = its channel pitch, 0 or c: dense) is the second
 * gradient of the pooled tensor when it also feeds a skip connection (S/C:270-279), summed here instead of by
 * a separate add */
int ur_maxpool2x2_bwd(const void* dy, const uint8_t* idx, void* dx, const void* addend, int add_ld, int dtype, int n,
                      int h, int w, int c, ur_stream_t s);
int ur_upsample_bilinear_fwd(const void* x, void* y, int dtype, int n, int h, int w, int c, int scale,
                             ur_stream_t s);
int ur_upsample_bilinear_bwd(const void* dy, void* dx, int dtype, int n, int h, int w, int c, int scale,
                             ur_stream_t s);
int ur_avgpool_fwd(const void* x, void* y, int dtype, int n, int h, int w, int c, int k, ur_stream_t s);
int ur_avgpool_bwd(const void* dy, void* dx, int dtype, int n, int h, int w, int c, int k, ur_stream_t s);
/* dst[n][y+pad_t][x+pad_l][dst_coff .. dst_coff+c) = src[n][y][x][src_coff .. +c);  the rest of
 * dst's channel window is zero-filled where padding applies (negative pads crop). */
int ur_copy_channels(const void* src, int src_ld, int src_coff, int sh, int sw, void* dst, int dst_ld,
                     int dst_coff, int dh, int dw, int pad_t, int pad_l, int n, int c, int dtype,
                     ur_stream_t s);
/* dst[r][0..c) = src[r][0..c), dst[r][c..dst_c) = 0 for dense rows: widens a 3-channel image or a 19-class gradient
 * (S/C:94, S/C:132 backward) to the multiple of 8 channels the generic kernels gather.  dst_c * element size must be a
 * multiple of 16, c * element size <= 128, both tensors 16-byte aligned. */
int ur_pad_channels(const void* src, int c, void* dst, int dst_c, long long rows, int dtype, ur_stream_t s);
/* general channel shuffle (S/C:6-15): out[.., j*groups + i] = x[.., i*(c/groups) + j] */
int ur_channel_shuffle(const void* x, void* y, int dtype, long long rows, int c, int groups, ur_stream_t s);
/* stand-alone activations of act_layer (S/C:17-34); backward takes the forward INPUT x */
#define UR_ACT_NONE 0
#define UR_ACT_RELU 1
#define UR_ACT_RELU6 2
#define UR_ACT_LEAKYRELU 3
#define UR_ACT_GELU 4
#define UR_ACT_HSWISH 5
int ur_act_fwd(const void* x, void* y, int dtype, long long count, int act, float slope, ur_stream_t s);
int ur_act_bwd(const void* dy, const void* x, void* dx, int dtype, long long count, int act, float slope,
               ur_stream_t s);
/* nn.PReLU(num_parameters = nslope, init) of act_layer('prelu') (S/C:26-27): y = x > 0 ? x : slope[c] * x, channels
 * fastest (count = rows * c), nslope = 1 or c.  Backward writes dx and slope_terms[count] = dy * x * [x <= 0] (fp32); the
 * slope gradient is their per-channel column sum (ur_colsum; over everything when nslope == 1). */
int ur_prelu_fwd(const void* x, const float* slope, int nslope, void* y, int dtype, long long count, int c, ur_stream_t s);
int ur_prelu_bwd(const void* dy, const void* x, const float* slope, int nslope, void* dx, float* slope_terms, int dtype,
                 long long count, int c, ur_stream_t s);
/* One idle wave that occupies stream s for `microseconds` (1..100000) and touches no memory.  No counterpart in the
 * reference: the host side uses it once per device to find a second stream whose launches really run beside the first one's
 * (two streams that the runtime maps onto one hardware queue execute in order), before it puts the ResBlock branch of a
 * two-branch network (S/C:262-280: the branches meet only at the Feature Merge) on that stream. */
int ur_spin(int microseconds, ur_stream_t s);
/* a += b (gradient accumulation at forks), elementwise over `count` elements */
int ur_add_inplace(void* a, const void* b, int dtype, long long count, ur_stream_t s);
/* layout / dtype conversion at the model boundary */
int ur_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int n, int c, int h, int w,
                    ur_stream_t s);
int ur_nhwc_to_nchw(const void* src, int src_dtype, void* dst, int dst_dtype, int n, int c, int h, int w,
                    ur_stream_t s);

/* input packing, the step before the hot path (Segmentation/Cityscapes/DataInstance/Cityscapes.py:191-194,
 * 220-223, 238, 256): uint8 HWC pixels -> NHWC activation, out[ch] = (px[ch] - sub[ch]) * mul[ch] for ch < c and 0 for
 * c <= ch < out_c (sub / mul: HOST arrays of c <= 4 floats; Normalize(mean, std) is sub = 255 mean, mul = 1/(255 std);
 * divide != 0 computes (px - sub) / mul instead, correctly rounded: image / 255.0 is sub = 0, mul = 255);
 * label ids -> int64 classes through a HOST table of 256 bytes. */
int ur_pack_image_u8(const uint8_t* img, void* out, int dtype, long long pixels, int c, int out_c, const float* sub,
                     const float* mul, int divide, ur_stream_t s);
int ur_remap_labels_u8(const uint8_t* ids, int64_t* out, long long count, const uint8_t* lut256, ur_stream_t s);

/* ---------------------------------------------------------------- losses and masks
 * CrossEntropyLoss(ignore_index) (train_UResNetAblitionStudy.py:66-67, CNN-CIFAR10.py:129),
 * BCEWithLogitsLoss (Ice-Seg/TrainingCode.py:97), argmax(1) (train_UResNetAblitionStudy.py:73),
 * sigmoid(x) > 0.5 (Ice-Seg/TrainingCode.py:132). */
/* workspace (bytes) for the loss reductions over n rows / elements */
size_t ur_loss_workspace(long long n);
/* Forward: lse[rows] = log-sum-exp per row; out3 = {sum of NLL over valid rows, number of valid
 * rows, mean}.  Rows whose target equals ignore_index (or lies outside [0, classes)) are skipped. */
int ur_ce_fwd(const void* logits, int dtype, const int64_t* target, long long rows, int classes, int ld,
              long long ignore_index, float* lse, float* out3, void* workspace, size_t ws_bytes, ur_stream_t s);
/* Backward: dlogits = (softmax - onehot) * gout[0] / count[0] on valid rows, 0 elsewhere.  count is
 * read from device memory (out3 + 1 of the forward, or a globally reduced copy); gout may be NULL (1). */
int ur_ce_bwd(const void* logits, int dtype, const int64_t* target, long long rows, int classes, int ld,
              long long ignore_index, const float* lse, const float* count, const float* gout, void* dlogits,
              int dld, ur_stream_t s);
/* ur_ce_bwd for the 1x1 head behind the loss (S/C:127-133 -> train_UResNetAblitionStudy.py:62-68): the gradient rows are
 * written `dld` columns wide (>= classes, zero fill; dld * element size a multiple of 16 and at most 64 bytes) -- the form
 * the streaming 1x1 kernels read -- and colsum[classes] receives the per-class sums of the stored gradient, which is that
 * head's bias gradient.  Saves a pass that widens the rows and a pass that sums them. */
size_t ur_ce_bwd_padded_workspace(long long rows);
int ur_ce_bwd_padded_supported(int dtype, int classes, int dld);
int ur_ce_bwd_padded(const void* logits, int dtype, const int64_t* target, long long rows, int classes,
                     long long ignore_index, const float* lse, const float* count, const float* gout, void* dlogits, int dld,
                     float* colsum, void* workspace, size_t ws_bytes, ur_stream_t s);
/* ur_ce_fwd without a workspace argument and without any allocation.  No counterpart in the reference: it exists for
 * callers that cannot allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  The per-workgroup partial sums live in a device array inside the library (8 KiB per GPU, held
 * until the process ends; each GPU has its own copy, whose address is resolved -- a runtime query that may load the
 * code object -- by ur_ce_fwd_noalloc_prepare() or by the first call made with that GPU current: a caller that
 * captures the call into a HIP graph calls ur_ce_fwd_noalloc_prepare() once per GPU BEFORE the capture begins; after
 * that the call enqueues two kernels and nothing else).  Same kernels, same results as ur_ce_fwd bit for bit.  rows > ur_ce_fwd_noalloc_max_rows() (2^24) is
 * rejected with UR_ERR_INVALID.  All calls on one GPU share the array: calls that overlap on one GPU (different
 * streams or threads) race; calls on one stream, or otherwise ordered, are safe and need no clearing in between. */
long long ur_ce_fwd_noalloc_max_rows(void);
int ur_ce_fwd_noalloc_prepare(void);
int ur_ce_fwd_noalloc(const void* logits, int dtype, const int64_t* target, long long rows, int classes, int ld,
                      long long ignore_index, float* lse, float* out3, ur_stream_t s);
/* BCEWithLogits, mean over all elements: out3 = {sum, count, mean} */
int ur_bce_fwd(const void* logits, int dtype, const float* target, long long count, float* out3, void* workspace,
               size_t ws_bytes, ur_stream_t s);
int ur_bce_bwd(const void* logits, int dtype, const float* target, long long count, const float* gout,
               void* dlogits, ur_stream_t s);
/* first-maximum argmax over the channel axis (bit-exact class map given identical logits) */
int ur_argmax_channels(const void* logits, int dtype, long long rows, int classes, int ld, int64_t* out,
                       ur_stream_t s);
/* out = sigmoid(x) > 0.5 evaluated in fp32 as the reference writes it */
int ur_sigmoid_mask(const void* logits, int dtype, long long count, uint8_t* out, ur_stream_t s);
/* Validation metrics of train_UResNetAblitionStudy.py:70-100 without leaving the GPU.
 * cm[t * classes + p] += 1 for every pixel with target t != ignore_index (0 <= t < classes) and p = argmax(logits)
 * (same first-maximum rule as ur_argmax_channels); cm is 64-bit and accumulates across calls. */
int ur_confusion_matrix(const void* logits, int dtype, const int64_t* target, long long rows, int classes, int ld,
                        long long ignore_index, unsigned long long* cm, ur_stream_t s);
/* acc7 += (precision, recall, accuracy, f1, iou, 1, *loss) of one confusion matrix: macro averages over the labels
 * present in targets or predictions, zero denominators score 0 (sklearn average='macro', zero_division=0; :82-89).
 * Dividing acc7[0..4] and acc7[6] by acc7[5] gives the per-batch means that validate() reports (:144-162).
 * loss: device scalar of the batch (NULL: none). */
int ur_seg_metrics_accumulate(const unsigned long long* cm, int classes, const float* loss, double* acc7, ur_stream_t s);

/* ---------------------------------------------------------------- optimisers (multi-tensor)
 * optim.Adam(lr) (train_UResNetAblitionStudy.py:203) and optim.SGD(lr) (CNN-CIFAR10.py:130).
 * ptrs: device array of ntensors*{4|2} pointers (param, grad, exp_avg, exp_avg_sq | param, grad);
 * sizes: device array of element counts; total = sum(sizes). */
int ur_adam_multi(const void* const* ptrs, const long long* sizes, int ntensors, long long max_size, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  ur_stream_t s);
/* Adam with device-resident step counts (same update as ur_adam_multi; replaces the `step` bookkeeping of
 * torch.optim.Adam, which train_UResNetAblitionStudy.py:122 advances once per optimizer.step()).  ptrs holds
 * ntensors*5 pointers: (param, grad, exp_avg, exp_avg_sq, int32 step counter).  The call first adds 1 to every
 * counter, then updates each tensor with the bias corrections of ITS counter.  No argument changes from step to
 * step, so the launch pair can be captured in a HIP graph and replayed. */
int ur_adam_multi_dev(const void* const* ptrs, const long long* sizes, int ntensors, long long max_size, float lr,
                      float beta1, float beta2, float eps, float weight_decay, float grad_scale, ur_stream_t s);
int ur_sgd_multi(const void* const* ptrs, const long long* sizes, int ntensors, long long max_size, float lr,
                 float weight_decay, float grad_scale, ur_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
