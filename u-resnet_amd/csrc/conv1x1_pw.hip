// Pointwise (1x1, stride 1) convolution on full-resolution activations, bf16: forward (+bias) and backward-data.
//
// These layers (the EUCB 1x1 projection, the segmentation head, the gradients of both; reference
// Segmentation/Cityscapes/modules/U_ResNet.py:53-55,127-133) read or write 80 / 32 / 24 / 16 channels per pixel
// over 3 M pixels: a few FLOP per byte, so the kernel is a stream with a small matrix product in the middle:
//   * a wave owns 32-pixel chunks and never meets another wave (no workgroup barrier).  A chunk is one contiguous
//     span of memory (or 32 equal pieces of rows when the tensor is a channel window): lanes fetch it as consecutive
//     16-byte pieces, park it in the wave's own LDS rows (pitch = an odd number of 16-byte units: conflict-free
//     ds_read_b128) and read it back as MFMA operands;
//   * D[channel][pixel] = W[channel][k] . X^T[k][pixel] with v_mfma_f32_32x32x16_bf16: the weight matrix is the A
//     operand and lives in registers for the whole kernel (converted from the fp32 parameter by each wave once, no
//     packing step); the activations are the B operand: lane (pixel, half) supplies channels [16s + 8 half, +8);
//   * results go back through LDS so that the global stores are consecutive 16-byte pieces as well;
//   * the next chunk's global loads are in flight while the current one is multiplied and stored.
//   grid = one resident round of 4-wave workgroups; a wave takes chunks wave_id, wave_id + waves, ...
#include "ur_common.h"

// tensors streamed once per launch go through non-temporal accesses (0.2 % of the training step); -DUR_PW_NT=0: ordinary
#ifndef UR_PW_NT
#define UR_PW_NT 1
#endif
namespace {

constexpr int NT = 256;

struct PwParams {
    const bf16_t* x;      // [P][in_ld], K channels used
    const float* w;       // A[m][k] = w[m * sm + k * sk] for m < wM, k < wK (zero elsewhere)
    const float* bias;    // [wM] or null
    bf16_t* y;            // [P][out_ld], M channels written
    long long P;
    int K, M, in_ld, out_ld;
    int wM, wK, sm, sk;
    int pitch_in, pitch_out;   // LDS row pitches in bytes
    int flat;                  // output rows are contiguous in memory and in LDS (out_ld == M): copy the chunk as bytes
    const float* in_aff;       // AFF: [2][K] per-channel scale then shift applied to x on its way into LDS (a folded BatchNorm)
    int in_relu;               // AFF: ReLU after the multiply-add
    // A logical tensor that lives in two dense ones (a channel concatenation that was never materialised): input channels
    // [K1, K) come from x2 (pitch in_ld2, constants in_aff2 = [2][K - K1]) when x2 is set; output channels [M1, M) go to
    // y2 (pitch out_ld2) when y2 is set.  All boundaries are multiples of 8 channels (whole 16-byte pieces).
    const bf16_t* x2; int K1, in_ld2; const float* in_aff2;
    bf16_t* y2; int M1, out_ld2;
    // BNA (backward-data of a convolution whose input is relu(BatchNorm(v)) in training mode): the product g is the gradient of
    // the activated tensor, which is not stored; stored is the BatchNorm's input gradient
    //   gamma * invstd * (g' - s0 - xhat * s1),  g' = g where the activation passed (xhat * gamma + beta > 0), else 0,
    //   xhat = (v - mean) * invstd,  (s0, s1) = bn_sums[c] * bn_inv_count
    // -- ur_bn_act_bwd_apply's value on the bf16-rounded g, to fp32 rounding.  v lies in bn_v (and bn_v2 for the channels that go to y2),
    // with the pitches of y / y2; the per-channel vectors cover all M channels in order.
    const bf16_t* bn_v; const bf16_t* bn_v2;
    const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta; const float* bn_sums;
    float bn_inv_count;
};

// x -> bf16(max(x * scale + shift, 0)) on 8 consecutive channels; cs = the channels' scales in LDS, their shifts K floats on
__device__ __forceinline__ bf16x8_t pw_affine8(const bf16x8_t& v, const float* cs, int K, int relu) {
    const float4 s0 = *reinterpret_cast<const float4*>(cs), s1 = *reinterpret_cast<const float4*>(cs + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(cs + K), h1 = *reinterpret_cast<const float4*>(cs + K + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = fmaf((float)v[e], sc[e], sh[e]);
        if (relu) f = fmaxf(f, 0.f);
        o[e] = (bf16_t)f;
    }
    return o;
}

// Asynchronous global -> LDS copy (LDS-DMA): every lane supplies its own 16-byte source, the data land at (wave-uniform) LDS
// byte address dst + lane * 16; inactive lanes copy nothing.  Counted in vmcnt.
__device__ __forceinline__ void pw_lds_dma16(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}

// LDS row pitch for rows of `bytes`: the next odd multiple of 16 bytes above it
int odd_pitch(int bytes) {
    int q = bytes / 16 + 1;
    if (!(q & 1)) ++q;
    return q * 16;
}

template <int MT, int KS, bool BIAS, bool AFF, bool BNA = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void conv1x1_pw_kernel(PwParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 31, half = lane >> 5;
    // per wave: input rows, output rows and (BNA) two buffers of 2 MT KiB for the BatchNorm input of this and the next chunk
    constexpr int VBUF = BNA ? 2 * MT * 1024 : 0;
    const size_t wave_bytes = (size_t)32 * (p.pitch_in + p.pitch_out) + 2 * VBUF;
    unsigned char* lin = lds + (size_t)wave * wave_bytes;
    unsigned char* lout = lin + 32 * p.pitch_in;
    unsigned char* vbuf = lout + 32 * p.pitch_out;
    // AFF: the input's per-channel constants sit behind the four waves' rows (the only workgroup barrier of the kernel)
    const float* aff = reinterpret_cast<const float*>(lds + (size_t)(NT / 64) * wave_bytes);
    if constexpr (AFF) {
        float* a = const_cast<float*>(aff);
        const int ka = p.in_aff2 ? p.K1 : p.K;                 // channels whose constants are in in_aff
        for (int i = threadIdx.x; i < 2 * p.K; i += NT) {
            const int which = i >= p.K ? 1 : 0, c = i - which * p.K;
            a[i] = c < ka ? p.in_aff[which * ka + c] : p.in_aff2[which * (p.K - ka) + c - ka];
        }
        __syncthreads();
    }

    bf16x8_t wf[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int m = mt * 32 + px;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = ks * 16 + half * 8 + e;
                wf[mt][ks][e] = (bf16_t)((m < p.wM && k < p.wK) ? p.w[(size_t)m * p.sm + (size_t)k * p.sk] : 0.f);
            }
        }
    // bias of the 4 channel runs this lane receives per block: channel = mt*32 + 8*j + 4*half + i
    float bs[BIAS ? MT : 1][4][4];
    if constexpr (BIAS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = mt * 32 + 8 * j + 4 * half + i;
                    bs[mt][j][i] = c < p.wM ? p.bias[c] : 0.f;
                }
    }

    // input pieces of a chunk: piece = i*64 + lane -> (pixel, 8-channel group); the same for every chunk
    const int gi = p.K >> 3, pieces_in = 32 * gi;
    int ipix[KS], isrc[KS], ilds[KS], ich[AFF ? KS : 1];
    bool isecond[KS];                                          // the piece comes from the second tensor
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        const int id = i * 64 + lane;
        const int pix = id / gi, grp = id - pix * gi;
        ipix[i] = id < pieces_in ? pix : 32;                   // 32: no such piece
        isecond[i] = p.x2 != nullptr && grp * 8 >= p.K1;
        isrc[i] = isecond[i] ? pix * p.in_ld2 + grp * 8 - p.K1 : pix * p.in_ld + grp * 8;
        ilds[i] = pix * p.pitch_in + grp * 16;
        if constexpr (AFF) ich[i] = grp * 8;
    }
    // output pieces: (pixel, group) of M/8 groups per pixel, or plain 16-byte pieces of the contiguous chunk
    // BNA: a lane keeps ONE channel group (lane % groups) for all its pieces, so that the group's BatchNorm constants fit in
    // registers: pass i covers pixels (lane / groups) + (64 / groups) i; the lanes beyond groups * (64 / groups) idle
    constexpr int NO = 2 * MT, NB = MT == 1 ? 2 : MT == 2 ? 4 : 7;      // pieces per lane: dense order / BNA passes
    const int go = p.M >> 3, pieces_out = p.flat ? (32 * p.M * 2) / 16 : 32 * go;
    int opix[NO], odst[NO], olds[NO];
    bool osecond[NO];                                          // the piece goes to the second tensor
    // (BNA: piece i of a lane is pixel b_p0 + b_ppp * i of its group: offsets are base + i * stride, nothing is tabulated)
    const int b_ppp = BNA ? 64 / go : 1, b_grp = BNA ? lane % go : 0, b_p0 = BNA ? lane / go : 0;
    const bool b_second = BNA && p.y2 != nullptr && b_grp * 8 >= p.M1;
    const int b_ld = b_second ? p.out_ld2 : p.out_ld;
    const int b_dst0 = b_second ? b_p0 * p.out_ld2 + b_grp * 8 - p.M1 : b_p0 * p.out_ld + b_grp * 8;
    const int b_lds0 = b_p0 * p.pitch_out + b_grp * 16;
    // The BatchNorm's per-channel constants sit behind the four waves' rows, folded to four per channel ([4][M]):
    //   k = gamma * invstd, sh = beta - mean * k:  the activation passed where v * k + sh > 0 (the forward's own test);
    //   gamma * invstd * (g' - s0 - xhat * s1) = k * g' + (q * v + r),  q = -k * s1 * invstd,  r = -k * s0 - q * mean
    const float* bnc = reinterpret_cast<const float*>(lds + (size_t)(NT / 64) * wave_bytes);
    if constexpr (BNA) {
        float* a = const_cast<float*>(bnc);
        for (int c = threadIdx.x; c < p.M; c += NT) {
            const float m = p.bn_mean[c], is = p.bn_invstd[c], k = p.bn_gamma[c] * is;
            const float s0 = p.bn_sums[2 * c] * p.bn_inv_count, s1 = p.bn_sums[2 * c + 1] * p.bn_inv_count;
            const float q = -k * s1 * is;
            a[c] = k; a[p.M + c] = p.bn_beta[c] - m * k; a[2 * p.M + c] = q; a[3 * p.M + c] = -k * s0 - q * m;
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < (BNA ? 0 : NO); ++i) {
        const int id = i * 64 + lane;
        osecond[i] = false;
        if (p.flat) {
            opix[i] = id < pieces_out ? 0 : 32; odst[i] = id * 8; olds[i] = id * 16;
        } else {
            const int pix = id / (go > 0 ? go : 1), grp = id - pix * go;
            osecond[i] = p.y2 != nullptr && grp * 8 >= p.M1;
            opix[i] = id < pieces_out ? pix : 32; olds[i] = pix * p.pitch_out + grp * 16;
            odst[i] = osecond[i] ? pix * p.out_ld2 + grp * 8 - p.M1 : pix * p.out_ld + grp * 8;
        }
    }

    const long long chunks = (p.P + 31) >> 5;
    const long long wstep = (long long)gridDim.x * (NT / 64);
    long long ch = (long long)blockIdx.x * (NT / 64) + wave;
    const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};

    auto fetch = [&](long long chunk, bf16x8_t* dst) {
        const long long first = chunk * 32;
        const bf16_t* src = p.x + first * p.in_ld;
        const bf16_t* src2 = p.x2 ? p.x2 + first * p.in_ld2 : src;
        const int live = (int)(p.P - first < 32 ? p.P - first : 32);
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const bf16x8_t* q = reinterpret_cast<const bf16x8_t*>((isecond[i] ? src2 : src) + isrc[i]);
            dst[i] = ipix[i] < live ? (UR_PW_NT ? __builtin_nontemporal_load(q) : *q) : zero;
        }
    };

    // BNA: the BatchNorm's input at a chunk's output positions goes straight to LDS (LDS-DMA, no registers), in the dense
    // piece order (pixel, group) -- which is where a lane of the output stage finds its pieces: lane + pass * (64 / go) * go.
    // The copy for chunk c + 1 is issued before chunk c is worked on, so it has a whole iteration to land.
    auto fetch_v = [&](long long chunk, int buf) {
        const long long first = chunk * 32;
        const int live = (int)(p.P - first < 32 ? p.P - first : 32);
        const bf16_t* v1 = p.bn_v + first * p.out_ld;
        const bf16_t* v2 = p.bn_v2 ? p.bn_v2 + first * p.out_ld2 : v1;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)vbuf + buf * VBUF);
        // piece i * 64 + lane = (pixel, group), stepped from piece `lane` without a division (and without a table: registers)
        int pix = b_p0, grp = b_grp;
        asm volatile("" : "+v"(pix), "+v"(grp));               // (or the compiler keeps all the addresses in registers across the loop)
        const int dg = 64 - b_ppp * go;
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            if (pix < live) {
                const bool second = p.bn_v2 != nullptr && grp * 8 >= p.M1;
                pw_lds_dma16(second ? v2 + pix * p.out_ld2 + grp * 8 - p.M1 : v1 + pix * p.out_ld + grp * 8, dst + i * 1024);
            }
            pix += b_ppp; grp += dg;
            if (grp >= go) { grp -= go; pix += 1; }
        }
    };
    bf16x8_t nxt[KS];
    int vb = 0;
    if (ch < chunks) {
        fetch(ch, nxt);
        if constexpr (BNA) fetch_v(ch, 0);
    }
    for (; ch < chunks; ch += wstep) {
        if constexpr (BNA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this chunk's copy has landed
#pragma unroll
        for (int i = 0; i < KS; ++i)
            if (ipix[i] < 32) {
                if constexpr (AFF) *reinterpret_cast<bf16x8_t*>(lin + ilds[i]) = pw_affine8(nxt[i], aff + ich[i], p.K, p.in_relu);
                else *reinterpret_cast<bf16x8_t*>(lin + ilds[i]) = nxt[i];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (ch + wstep < chunks) {
            fetch(ch + wstep, nxt);
            if constexpr (BNA) fetch_v(ch + wstep, vb ^ 1);
        }
        bf16x8_t cur[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(lin + px * p.pitch_in + (ks * 16 + half * 8) * 2);
            cur[ks] = (ks * 16 + half * 8 < p.K) ? v : zero;       // beyond K the row holds padding, not zeros
        }
        f32x16_t acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = BIAS ? bs[BIAS ? mt : 0][i >> 2][i & 3] : 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[mt][ks], cur[ks], acc[mt], 0, 0, 0);
        }
        const long long first = ch * 32;
        const int live = (int)(p.P - first < 32 ? p.P - first : 32);
        if (p.flat && live < 32) {
            // ragged last chunk of a contiguous tensor: its byte count need not be a multiple of 16
            if (px < live) {
                bf16_t* out = p.y + (first + px) * p.out_ld;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int c = mt * 32 + 8 * (i >> 2) + 4 * half + (i & 3);
                        if (c < p.M) out[c] = (bf16_t)acc[mt][i];
                    }
            }
        } else {
            const int row = px * (p.flat ? p.M * 2 : p.pitch_out);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = mt * 32 + 8 * j + 4 * half;
                    if ((p.M & 3) == 0) {
                        if (c < p.M) {
                            bf16x4_t o = {(bf16_t)acc[mt][j * 4], (bf16_t)acc[mt][j * 4 + 1], (bf16_t)acc[mt][j * 4 + 2],
                                          (bf16_t)acc[mt][j * 4 + 3]};
                            *reinterpret_cast<bf16x4_t*>(lout + row + c * 2) = o;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (c + i < p.M) *reinterpret_cast<bf16_t*>(lout + row + (c + i) * 2) = (bf16_t)acc[mt][j * 4 + i];
                    }
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            bf16_t* out = p.y + first * p.out_ld;
            bf16_t* out2 = p.y2 ? p.y2 + first * p.out_ld2 : out;
            if constexpr (BNA) {
                // the lane's eight channels' constants, fetched here where the accumulators' registers are free again
                float ck[8], csh[8], cq[8], cr[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float* c0 = bnc + b_grp * 8 + 4 * h;
                    const float4 a = *reinterpret_cast<const float4*>(c0), b = *reinterpret_cast<const float4*>(c0 + p.M);
                    const float4 c = *reinterpret_cast<const float4*>(c0 + 2 * p.M), d = *reinterpret_cast<const float4*>(c0 + 3 * p.M);
                    const float ak[4] = {a.x, a.y, a.z, a.w}, bs[4] = {b.x, b.y, b.z, b.w}, cc[4] = {c.x, c.y, c.z, c.w}, dr[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ck[4 * h + j] = ak[j]; csh[4 * h + j] = bs[j]; cq[4 * h + j] = cc[j]; cr[4 * h + j] = dr[j]; }
                }
                int dst0 = b_dst0, lds0 = b_lds0, l16 = lane * 16;
                asm volatile("" : "+v"(dst0), "+v"(lds0), "+v"(l16));     // (as above)
                bf16_t* o = (b_second ? out2 : out) + dst0;
                const unsigned char* vsrc = vbuf + vb * VBUF + l16;
                vb ^= 1;
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    if (b_p0 < b_ppp && b_p0 + b_ppp * i < live) {
                        bf16x8_t o8 = *reinterpret_cast<const bf16x8_t*>(lout + lds0 + i * b_ppp * p.pitch_out);
                        const bf16x8_t v8 = *reinterpret_cast<const bf16x8_t*>(vsrc + i * b_ppp * go * 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = (float)v8[e];
                            const float gv = fmaf(v, ck[e], csh[e]) > 0.f ? (float)o8[e] : 0.f;
                            o8[e] = (bf16_t)fmaf(ck[e], gv, fmaf(cq[e], v, cr[e]));
                        }
                        bf16x8_t* q = reinterpret_cast<bf16x8_t*>(o + i * b_ppp * b_ld);
                        if (UR_PW_NT) __builtin_nontemporal_store(o8, q);
                        else *q = o8;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < NO; ++i)
                    if (opix[i] < live) {
                        const bf16x8_t o8 = *reinterpret_cast<const bf16x8_t*>(lout + olds[i]);
                        bf16x8_t* q = reinterpret_cast<bf16x8_t*>((osecond[i] ? out2 : out) + odst[i]);
                        if (UR_PW_NT) __builtin_nontemporal_store(o8, q);
                        else *q = o8;
                    }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

bool shape_ok(const ur_conv_desc* d) {
    return d->dtype == UR_BF16 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->oh == d->h && d->ow == d->w;
}

// Register budget: MT*KS weight quads (4 registers each) + 16*MT accumulators + 2*KS operand quads.  Up to 6 weight
// quads the kernel keeps 3+ waves per SIMD and beats the generic implicit GEMM (80 -> 19: 141 vs 189 us, 24 -> 80:
// 123 vs 195 us on 3.1 M pixels); wider layers (64 <-> 128 channels) drop to one wave per SIMD and lose to it.
bool tiles_ok(int mt, int ks) { return mt >= 1 && mt <= 3 && ks >= 1 && ks <= 5 && mt * ks <= 6; }

// the output side needs either 8 | channels and pitch (16-byte pieces of rows) or a contiguous tensor (any count)
bool out_ok(int M, int out_ld) { return (M % 8 == 0 && out_ld % 8 == 0) || out_ld == M; }
size_t lds_bytes(int K, int M, int out_ld) {
    const bool flat = !(M % 8 == 0 && out_ld % 8 == 0);
    return (size_t)(NT / 64) * 32 * (odd_pitch(K * 2) + (flat ? M * 2 : odd_pitch(M * 2)));
}

// One resident round: as many workgroups as the device holds at this instantiation's register count, so that every
// wave walks the same number of chunks and all finish together (no partially filled last round).
template <int MT, int KS, bool BIAS, bool AFF, bool BNA = false>
void launch_one(long long chunks, hipStream_t st, const PwParams& p) {
    const size_t lds = (size_t)(NT / 64) * (32 * (p.pitch_in + p.pitch_out) + (BNA ? 4 * MT * 1024 : 0)) +
                       (AFF ? (size_t)2 * p.K * sizeof(float) : 0) + (BNA ? (size_t)4 * p.M * sizeof(float) : 0);
    int per_cu = 0, dev = 0, cus = 0;
    if (BNA && lds > 64 * 1024)              // (more than the default limit of dynamic LDS per workgroup; a CU has 160 KiB)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_pw_kernel<MT, KS, BIAS, AFF, BNA>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1x1_pw_kernel<MT, KS, BIAS, AFF, BNA>, NT, lds) != hipSuccess || per_cu < 1)
        per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    const unsigned grid = (unsigned)ur_grid_1d(chunks, NT / 64, per_cu * cus);
    hipLaunchKernelGGL((conv1x1_pw_kernel<MT, KS, BIAS, AFF, BNA>), dim3(grid), dim3(NT), lds, st, p);
}

template <int MT>
int launch_ks(int ks, long long chunks, hipStream_t st, const PwParams& p) {
    switch (ks) {
#define UR_PW_CASE(KS_)                                                 \
    case KS_:                                                           \
        if constexpr (MT * KS_ <= 6) {                                  \
            if (p.in_aff && p.bias) launch_one<MT, KS_, true, true>(chunks, st, p);   \
            else if (p.in_aff) launch_one<MT, KS_, false, true>(chunks, st, p);       \
            else if (p.bias) launch_one<MT, KS_, true, false>(chunks, st, p);         \
            else launch_one<MT, KS_, false, false>(chunks, st, p);                    \
        }                                                               \
        break;
        UR_PW_CASE(1) UR_PW_CASE(2) UR_PW_CASE(3) UR_PW_CASE(4) UR_PW_CASE(5)
#undef UR_PW_CASE
        default: ur_set_error("conv1x1_pw: %d reduction steps", ks); return UR_ERR_INVALID;
    }
    return ur_check_launch("conv1x1_pw");
}

// the BNA instantiations: backward-data shapes of the heads (24 -> 80: MT 3, KS 2; 16 -> 32: MT 1, KS 1) and their neighbours
template <int MT>
int launch_bna(int ks, long long chunks, hipStream_t st, const PwParams& p) {
    switch (ks) {
        case 1: launch_one<MT, 1, false, false, true>(chunks, st, p); break;
        case 2: launch_one<MT, 2, false, false, true>(chunks, st, p); break;
        default: ur_set_error("conv1x1_pw (BatchNorm backward in the epilogue): %d reduction steps", ks); return UR_ERR_INVALID;
    }
    return ur_check_launch("conv1x1_pw");
}

int launch(PwParams p, hipStream_t st) {
    const int mt = ur_cdiv(p.M, 32), ks = ur_cdiv(p.K, 16);
    UR_REQUIRE(tiles_ok(mt, ks), "conv1x1_pw: %d -> %d channels is outside the register budget", p.K, p.M);
    UR_REQUIRE(p.K % 8 == 0 && p.in_ld % 8 == 0 && ((uintptr_t)p.x & 15) == 0,
               "conv1x1_pw: input channels and pitch must be multiples of 8 and the tensor 16-byte aligned");
    UR_REQUIRE(out_ok(p.M, p.out_ld) && ((uintptr_t)p.y & 15) == 0,
               "conv1x1_pw: the output needs 8 | channels and pitch, or a contiguous tensor; 16-byte aligned");
    p.flat = !(p.M % 8 == 0 && p.out_ld % 8 == 0);
    p.pitch_in = odd_pitch(p.K * 2);
    p.pitch_out = p.flat ? p.M * 2 : odd_pitch(p.M * 2);
    UR_REQUIRE(lds_bytes(p.K, p.M, p.out_ld) + 2 * p.K * sizeof(float) <= 64 * 1024, "conv1x1_pw: %d -> %d channels exceeds the LDS budget", p.K, p.M);
    UR_REQUIRE(!p.in_aff || ((uintptr_t)p.in_aff & 3) == 0, "conv1x1_pw: misaligned input constants");
    if (p.x2)
        UR_REQUIRE(p.K1 > 0 && p.K1 < p.K && p.K1 % 8 == 0 && p.in_ld2 % 8 == 0 && p.in_ld2 >= p.K - p.K1 && p.in_ld >= p.K1 &&
                       ((uintptr_t)p.x2 & 15) == 0 && (!p.in_aff || p.in_aff2),
                   "conv1x1_pw: the two-tensor input needs 8 | channel split and pitches, 16-byte alignment and constants for both");
    if (p.y2)
        UR_REQUIRE(!p.flat && p.M1 > 0 && p.M1 < p.M && p.M1 % 8 == 0 && p.out_ld2 % 8 == 0 && p.out_ld2 >= p.M - p.M1 &&
                       p.out_ld >= p.M1 && ((uintptr_t)p.y2 & 15) == 0,
                   "conv1x1_pw: the two-tensor output needs 8 | channel split and pitches and 16-byte alignment");
    if (p.P <= 0) return UR_OK;
    const long long chunks = (p.P + 31) / 32;
    if (p.bn_v) {
        UR_REQUIRE(!p.flat && !p.bias && !p.in_aff && ks <= 2 && p.M >= 8, "conv1x1_pw: BatchNorm backward in the epilogue needs 8 | channels and <= 32 reduced channels");
        UR_REQUIRE(((uintptr_t)p.bn_v & 15) == 0 && ((uintptr_t)p.bn_v2 & 15) == 0 && (!p.y2) == (!p.bn_v2),
                   "conv1x1_pw: the BatchNorm input must be 16-byte aligned and split as the output is");
        switch (mt) {
            case 1: return launch_bna<1>(ks, chunks, st, p);
            case 2: return launch_bna<2>(ks, chunks, st, p);
            default: return launch_bna<3>(ks, chunks, st, p);
        }
    }
    switch (mt) {
        case 1: return launch_ks<1>(ks, chunks, st, p);
        case 2: return launch_ks<2>(ks, chunks, st, p);
        default: return launch_ks<3>(ks, chunks, st, p);
    }
}

}  // namespace

extern "C" {

int ur_conv1x1_pw_supported(const ur_conv_desc* d, int dgrad) {
    if (!d || !shape_ok(d)) return 0;
    const int K = dgrad ? d->cout : d->cin, M = dgrad ? d->cin : d->cout;
    const int in_ld = dgrad ? (d->out_ld ? d->out_ld : d->cout) : (d->in_ld ? d->in_ld : d->cin);
    const int out_ld = dgrad ? (d->in_ld ? d->in_ld : d->cin) : (d->out_ld ? d->out_ld : d->cout);
    return (K % 8 == 0 && in_ld % 8 == 0 && tiles_ok(ur_cdiv(M, 32), ur_cdiv(K, 16)) && out_ok(M, out_ld) &&
            lds_bytes(K, M, out_ld) <= 64 * 1024) ? 1 : 0;
}

int ur_conv1x1_pw_fwd(const void* x, const float* w, int w_cout, int w_cin, const float* bias, void* y,
                      const ur_conv_desc* d, ur_stream_t s) {
    UR_REQUIRE(x && w && y && d, "conv1x1_pw_fwd: null argument");
    UR_REQUIRE(shape_ok(d), "conv1x1_pw_fwd: not a bf16 1x1 stride-1 convolution");
    UR_REQUIRE(w_cout <= d->cout && w_cin <= d->cin, "conv1x1_pw_fwd: weight %dx%d exceeds the tensors' %dx%d channels",
               w_cout, w_cin, d->cout, d->cin);
    PwParams p{};
    p.x = (const bf16_t*)x; p.w = w; p.bias = bias; p.y = (bf16_t*)y;
    p.P = (long long)d->n * d->h * d->w;
    p.K = d->cin; p.M = d->cout; p.in_ld = d->in_ld ? d->in_ld : d->cin; p.out_ld = d->out_ld ? d->out_ld : d->cout;
    p.wM = w_cout; p.wK = w_cin; p.sm = w_cin; p.sk = 1;
    return launch(p, (hipStream_t)s);
}

int ur_conv1x1_pw_fwd_affine(const void* x, const float* in_aff, int in_relu, const float* w, int w_cout, int w_cin,
                             const float* bias, void* y, const ur_conv_desc* d, ur_stream_t s) {
    UR_REQUIRE(x && in_aff && w && y && d, "conv1x1_pw_fwd_affine: null argument");
    UR_REQUIRE(shape_ok(d), "conv1x1_pw_fwd_affine: not a bf16 1x1 stride-1 convolution");
    UR_REQUIRE(w_cout <= d->cout && w_cin <= d->cin, "conv1x1_pw_fwd_affine: weight %dx%d exceeds the tensors' %dx%d channels",
               w_cout, w_cin, d->cout, d->cin);
    PwParams p{};
    p.x = (const bf16_t*)x; p.w = w; p.bias = bias; p.y = (bf16_t*)y;
    p.P = (long long)d->n * d->h * d->w;
    p.K = d->cin; p.M = d->cout; p.in_ld = d->in_ld ? d->in_ld : d->cin; p.out_ld = d->out_ld ? d->out_ld : d->cout;
    p.wM = w_cout; p.wK = w_cin; p.sm = w_cin; p.sk = 1;
    p.in_aff = in_aff; p.in_relu = in_relu;
    return launch(p, (hipStream_t)s);
}

int ur_conv1x1_pw_fwd_affine_pair(const void* xa, int ca, int ld_a, const void* xb, int cb, int ld_b, const float* aff_a,
                                  const float* aff_b, int in_relu, const float* w, int w_cout, int w_cin, const float* bias, void* y,
                                  const ur_conv_desc* d, ur_stream_t s) {
    UR_REQUIRE(xa && xb && aff_a && aff_b && w && y && d, "conv1x1_pw_fwd_affine_pair: null argument");
    UR_REQUIRE(shape_ok(d), "conv1x1_pw_fwd_affine_pair: not a bf16 1x1 stride-1 convolution");
    UR_REQUIRE(ca + cb == d->cin && w_cout <= d->cout && w_cin <= d->cin,
               "conv1x1_pw_fwd_affine_pair: %d + %d channels against a descriptor of %d, weight %dx%d", ca, cb, d->cin, w_cout, w_cin);
    PwParams p{};
    p.x = (const bf16_t*)xa; p.w = w; p.bias = bias; p.y = (bf16_t*)y;
    p.P = (long long)d->n * d->h * d->w;
    p.K = d->cin; p.M = d->cout; p.in_ld = ld_a ? ld_a : ca; p.out_ld = d->out_ld ? d->out_ld : d->cout;
    p.wM = w_cout; p.wK = w_cin; p.sm = w_cin; p.sk = 1;
    p.in_aff = aff_a; p.in_relu = in_relu;
    p.x2 = (const bf16_t*)xb; p.K1 = ca; p.in_ld2 = ld_b ? ld_b : cb; p.in_aff2 = aff_b;
    return launch(p, (hipStream_t)s);
}

int ur_conv1x1_pw_dgrad_pair(const void* dy, const float* w, int w_cout, int w_cin, void* dxa, int ca, int ld_a, void* dxb, int cb,
                             int ld_b, const ur_conv_desc* d, ur_stream_t s) {
    UR_REQUIRE(dy && w && dxa && dxb && d, "conv1x1_pw_dgrad_pair: null argument");
    UR_REQUIRE(shape_ok(d), "conv1x1_pw_dgrad_pair: not a bf16 1x1 stride-1 convolution");
    UR_REQUIRE(ca + cb == d->cin && w_cout <= d->cout && w_cin <= d->cin,
               "conv1x1_pw_dgrad_pair: %d + %d channels against a descriptor of %d, weight %dx%d", ca, cb, d->cin, w_cout, w_cin);
    PwParams p{};
    p.x = (const bf16_t*)dy; p.w = w; p.bias = nullptr; p.y = (bf16_t*)dxa;
    p.P = (long long)d->n * d->h * d->w;
    p.K = d->cout; p.M = d->cin; p.in_ld = d->out_ld ? d->out_ld : d->cout; p.out_ld = ld_a ? ld_a : ca;
    p.wM = w_cin; p.wK = w_cout; p.sm = 1; p.sk = w_cin;
    p.y2 = (bf16_t*)dxb; p.M1 = ca; p.out_ld2 = ld_b ? ld_b : cb;
    return launch(p, (hipStream_t)s);
}

int ur_conv1x1_pw_dgrad_bn_pair(const void* dy, const float* w, int w_cout, int w_cin, const void* va, void* dva, int ca, int ld_a,
                                const void* vb, void* dvb, int cb, int ld_b, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, const float* sums, const ur_conv_desc* d, ur_stream_t s) {
    UR_REQUIRE(dy && w && va && dva && mean && invstd && gamma && beta && sums && d, "conv1x1_pw_dgrad_bn_pair: null argument");
    UR_REQUIRE((vb == nullptr) == (dvb == nullptr), "conv1x1_pw_dgrad_bn_pair: the second tensor needs both its input and its gradient");
    UR_REQUIRE(shape_ok(d), "conv1x1_pw_dgrad_bn_pair: not a bf16 1x1 stride-1 convolution");
    UR_REQUIRE(ca + (vb ? cb : 0) == d->cin && w_cout <= d->cout && w_cin <= d->cin,
               "conv1x1_pw_dgrad_bn_pair: %d + %d channels against a descriptor of %d, weight %dx%d", ca, vb ? cb : 0, d->cin, w_cout, w_cin);
    PwParams p{};
    p.x = (const bf16_t*)dy; p.w = w; p.bias = nullptr; p.y = (bf16_t*)dva;
    p.P = (long long)d->n * d->h * d->w;
    p.K = d->cout; p.M = d->cin; p.in_ld = d->out_ld ? d->out_ld : d->cout; p.out_ld = ld_a ? ld_a : ca;
    p.wM = w_cin; p.wK = w_cout; p.sm = 1; p.sk = w_cin;
    if (vb) { p.y2 = (bf16_t*)dvb; p.M1 = ca; p.out_ld2 = ld_b ? ld_b : cb; }
    p.bn_v = (const bf16_t*)va; p.bn_v2 = (const bf16_t*)vb;
    p.bn_mean = mean; p.bn_invstd = invstd; p.bn_gamma = gamma; p.bn_beta = beta; p.bn_sums = sums;
    p.bn_inv_count = 1.f / (float)p.P;
    return launch(p, (hipStream_t)s);
}

int ur_conv1x1_pw_dgrad(const void* dy, const float* w, int w_cout, int w_cin, void* dx, const ur_conv_desc* d,
                        ur_stream_t s) {
    UR_REQUIRE(dy && w && dx && d, "conv1x1_pw_dgrad: null argument");
    UR_REQUIRE(shape_ok(d), "conv1x1_pw_dgrad: not a bf16 1x1 stride-1 convolution");
    UR_REQUIRE(w_cout <= d->cout && w_cin <= d->cin, "conv1x1_pw_dgrad: weight %dx%d exceeds the tensors' %dx%d channels",
               w_cout, w_cin, d->cout, d->cin);
    PwParams p{};
    p.x = (const bf16_t*)dy; p.w = w; p.bias = nullptr; p.y = (bf16_t*)dx;
    p.P = (long long)d->n * d->h * d->w;
    p.K = d->cout; p.M = d->cin; p.in_ld = d->out_ld ? d->out_ld : d->cout; p.out_ld = d->in_ld ? d->in_ld : d->cin;
    p.wM = w_cin; p.wK = w_cout; p.sm = 1; p.sk = w_cin;
    return launch(p, (hipStream_t)s);
}

}  // extern "C"
