// Backward passes of the row-wise operators the trainable grounding head is made of (train_walkgpt.py:347-350 leaves the mask decoder,
// text_hidden_fcs (CTP), the projector and the LLM's LoRA / head trainable; the reference gets these gradients from torch autograd over
// nn.Linear / nn.LayerNorm / nn.GELU / nn.ReLU / F.scaled_dot_product-style attention / F.interpolate / the loss functions of
// utils/utils_walkgpt.py).  Gradients of a Linear are GEMMs and run on gemm.hip (dX = dY W, dW = dY^T X on transposed copies); this
// file holds what is not a GEMM:
//   wg_colsum_f32            db[n] = sum_m dY[m][n]                                   (bias gradient)
//   wg_act_bf16 / _bwd       y = act(x);  dx = dy * act'(x)                           (GELU erf / quick-GELU / ReLU as separate operators)
//   wg_layernorm_bwd_bf16    dx, dgamma, dbeta of a row LayerNorm
//   wg_ctp_tail_bwd_bf16     backward of CalibratedTextProjector's tail (utils_walkgpt.py:321-327): LayerNorm -> + text_type -> L2
//                            normalise -> * exp(log_temp)
//   wg_attn_bwd_f32          dq, dk, dv of softmax(scale q k^T) v for the decoder's / projector's small attentions
//   wg_hyper_mask_dot_bwd    gradients of masks = hyper_in @ upscaled (mask_decoder.py:150-160)
//   wg_postprocess_bwd_f32   adjoint of the two bilinear resamples of Sam.postprocess_masks (sam.py:137-172)
//   wg_mask_losses_bwd_f32   d(sigmoid_ce_loss + dice_loss)/d logits (utils_walkgpt.py:76-120)          (the last two live in misc.hip)
//   wg_avgpool_tokens_bwd / wg_mean_tokens_bwd / wg_sigmoid_gate_bwd        MSQP's pooling and gate (utils_walkgpt.py:195-217,256-257)
//   wg_resample_tokens_bwd_f32, wg_splice_multimodal_bwd_bf16                the path from the language model's input embeddings back to the
//                            projector's tokens and to embed_tokens (llava_arch.py:252-259, :265-518)
// All HBM-bound row or element kernels: fp32 arithmetic, bf16 activations, fp32 accumulation of parameter gradients (atomics).
#include "wg_common.h"

namespace {

// ---- column sums ----------------------------------------------------------------------------------------------------------------------
// x [R, C] bf16 -> out[C] += sum over rows (fp32; the caller zeroes out).  A workgroup takes 256 rows x 512 columns: lane = 8 columns.
__global__ __launch_bounds__(256) void wg_colsum_kernel(const bf16* x, long ldx, float* out, int R, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 512 + lane * 8;
    if (c >= C) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int r0 = blockIdx.y * 256;
    const int r1 = r0 + 256 < R ? r0 + 256 : R;
    for (int r = r0 + wave; r < r1; r += 4) {
        const bf16x8 t = *(const bf16x8*)(x + (long)r * ldx + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)t[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(out + c + e, s[e]);
}

// The same without atomics: every workgroup (256 rows x 512 columns) leaves its partial row in part[blockIdx.y][C]; wg_fold_rows_kernel sums the
// partial rows in order.  (Fixed order: the same bits every run.)
__global__ __launch_bounds__(256) void wg_colsum_part_kernel(const bf16* x, long ldx, float* part, int R, int C) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 512 + lane * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int r0 = blockIdx.y * 256;
    const int r1 = r0 + 256 < R ? r0 + 256 : R;
    if (c < C) {
        for (int r = r0 + wave; r < r1; r += 4) {
            const bf16x8 t = *(const bf16x8*)(x + (long)r * ldx + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)t[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = s[e];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int cc = blockIdx.x * 512 + i;
        if (cc < C) part[(long)blockIdx.y * C + cc] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}
// out[c] = sum_b part[b][c] (b ascending), n = number of columns; fp32 or bf16 out
__global__ __launch_bounds__(256) void wg_fold_rows_kernel(const float* part, int blocks, long n, void* out, int out_f32) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    float s = 0.f;
    for (int b = 0; b < blocks; ++b) s += part[(long)b * n + c];
    if (out_f32) ((float*)out)[c] = s; else ((bf16*)out)[c] = (bf16)s;
}

// ---- activations as separate operators ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wg_act_grad(float x, int act) {
    switch (act) {
        case WG_ACT_GELU_ERF: {
            // d/dx [x Phi(x)] = Phi(x) + x phi(x)
            const float cdf = 0.5f * (1.0f + wg_erf(x * 0.70710678118654752440f));
            return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
        }
        case WG_ACT_QUICK_GELU: {
            const float s = 1.0f / (1.0f + __expf(-1.702f * x));
            return s + 1.702f * x * s * (1.0f - s);
        }
        case WG_ACT_RELU: return x > 0.f ? 1.0f : 0.f;
        default: return 1.0f;
    }
}
template <bool BWD>
__global__ __launch_bounds__(256) void wg_act_kernel(const bf16* x, const bf16* dy, bf16* out, long n8, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const bf16x8 t = *(const bf16x8*)(x + i * 8);
        bf16x8 o;
        if (BWD) {
            const bf16x8 g = *(const bf16x8*)(dy + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)g[e] * wg_act_grad((float)t[e], act));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)wg_act((float)t[e], act);
        }
        *(bf16x8*)(out + i * 8) = o;
    }
}

// ---- LayerNorm backward, rows of 64 VEC <= 512 channels (the head's widths: 64, 256, 512), deterministic ------------------------------------------
// The general kernel below spreads a row over lane * 8 columns (half the wave idle at C = 256), carries registers for 4096 columns and ends in
// 2 C atomics per wave: 90 us for the decoder's 32 768 x 256 image-token rows, 16 % of a head step.  Here lane l holds columns VEC l .. VEC l + VEC - 1
// (every lane busy), a wave walks a contiguous run of rows with its dgamma / dbeta partials in 2 VEC registers, the four waves of a workgroup add
// theirs in LDS, and the workgroup leaves ONE partial row pair in the workspace [blocks][2][C]; wg_ln_partials_kernel sums them in block order.
template <int VEC>
__global__ __launch_bounds__(256) void wg_layernorm_bwd_small_kernel(const bf16* x, long ldx, const bf16* gamma, const bf16* dy, long lddy, bf16* dx, long lddx,
                                                                     float* part, int M, int rows_per_block, float eps) {
    typedef __attribute__((ext_vector_type(VEC))) __bf16 vec_t;
    constexpr int C = 64 * VEC;
    __shared__ float red[4][2][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = lane * VEC;
    float gm[VEC], ag[VEC], ab[VEC];
    {
        const vec_t t = *(const vec_t*)(gamma + d);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { gm[e] = VEC == 1 ? (float)((const bf16*)&t)[0] : (float)t[e]; ag[e] = ab[e] = 0.f; }
    }
    const float invc = 1.0f / (float)C;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    for (int m = r0 + wave; m < r1; m += 4) {
        const vec_t tx = *(const vec_t*)(x + (long)m * ldx + d), tg = *(const vec_t*)(dy + (long)m * lddy + d);
        float v[VEC], g[VEC], s = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[e] = VEC == 1 ? (float)((const bf16*)&tx)[0] : (float)tx[e];
            g[e] = VEC == 1 ? (float)((const bf16*)&tg)[0] : (float)tg[e];
            s += v[e];
        }
        const float mean = wg_wave_sum(s) * invc;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { v[e] -= mean; q += v[e] * v[e]; }
        const float rstd = 1.0f / sqrtf(wg_wave_sum(q) * invc + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[e] *= rstd;                                              // xhat
            const float gg = g[e] * gm[e];
            sg += gg;
            sgx += gg * v[e];
            ag[e] += g[e] * v[e];
            ab[e] += g[e];
        }
        const float mg = wg_wave_sum(sg) * invc, mgx = wg_wave_sum(sgx) * invc;
        vec_t o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bf16 r = (bf16)(rstd * (g[e] * gm[e] - mg - v[e] * mgx));
            if (VEC == 1) ((bf16*)&o)[0] = r; else o[e] = r;
        }
        *(vec_t*)(dx + (long)m * lddx + d) = o;
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { red[wave][0][d + e] = ag[e]; red[wave][1][d + e] = ab[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        part[((long)blockIdx.x * 2 + which) * C + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
}

// dgamma | dbeta [2][C] = sum over the `blocks` partial row pairs, in block order; out bf16 or fp32
__global__ __launch_bounds__(256) void wg_ln_partials_kernel(const float* part, int blocks, int C, void* dgamma, void* dbeta, int out_f32) {
    __shared__ float red[4];
    const int col = blockIdx.x;          // 0 .. 2 C - 1: (which, c)
    float s = 0.f;
    for (int b = threadIdx.x; b < blocks; b += 256) s += part[(long)b * 2 * C + col];
    s = wg_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = red[0] + red[1] + red[2] + red[3];
        void* dst = col < C ? dgamma : dbeta;
        const int c = col < C ? col : col - C;
        if (out_f32) ((float*)dst)[c] = t; else ((bf16*)dst)[c] = (bf16)t;
    }
}

// ---- LayerNorm backward -----------------------------------------------------------------------------------------------------------------
// xhat = (x - mean) rstd,  g = dy gamma:   dx = rstd (g - mean(g) - xhat mean(g xhat)),  dgamma += dy xhat,  dbeta += dy.
// One wave per row at a time (rows strided over the grid's waves), the row read twice; a wave keeps its columns' dgamma / dbeta partials in
// registers over all its rows and adds them to the fp32 outputs once (C <= 4096: 8 chunks of 512 columns; the row itself stays in registers).
constexpr int LNB_CH = 8;
__global__ __launch_bounds__(256) void wg_layernorm_bwd_kernel(const bf16* x, long ldx, const bf16* gamma, const bf16* dy, long lddy, bf16* dx,
                                                               long lddx, float* dgamma, float* dbeta, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    float ag[LNB_CH][8], ab[LNB_CH][8];
#pragma unroll
    for (int c = 0; c < LNB_CH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) ag[c][e] = ab[c][e] = 0.f;
    const float invc = 1.0f / (float)C;
    for (int m = wid; m < M; m += nw) {
        const bf16* xr = x + (long)m * ldx;
        const bf16* gr = dy + (long)m * lddy;
        float v[LNB_CH][8];                    // the row, then xhat (statistics exactly as the forward kernel forms them: two passes)
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < LNB_CH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < C) {
                const bf16x8 t = *(const bf16x8*)(xr + d);
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[c][e] = (float)t[e]; s += v[c][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
            }
        }
        const float mean = wg_wave_sum(s) * invc;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < LNB_CH; ++c) {
            if (c * 512 + lane * 8 < C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[c][e] -= mean; q += v[c][e] * v[c][e]; }
            }
        }
        const float rstd = 1.0f / sqrtf(wg_wave_sum(q) * invc + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int c = 0; c < LNB_CH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < C) {
                const bf16x8 g = *(const bf16x8*)(gr + d), gm = *(const bf16x8*)(gamma + d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[c][e] *= rstd;                                   // xhat
                    const float gg = (float)g[e] * (float)gm[e];
                    sg += gg;
                    sgx += gg * v[c][e];
                    ag[c][e] += (float)g[e] * v[c][e];
                    ab[c][e] += (float)g[e];
                }
            }
        }
        const float mg = wg_wave_sum(sg) * invc, mgx = wg_wave_sum(sgx) * invc;
        bf16* dr = dx + (long)m * lddx;
#pragma unroll
        for (int c = 0; c < LNB_CH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < C) {
                const bf16x8 g = *(const bf16x8*)(gr + d), gm = *(const bf16x8*)(gamma + d);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)(rstd * ((float)g[e] * (float)gm[e] - mg - v[c][e] * mgx));
                *(bf16x8*)(dr + d) = o;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < LNB_CH; ++c) {
        const int d = c * 512 + lane * 8;
        if (d < C) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                atomicAdd(dgamma + d + e, ag[c][e]);
                atomicAdd(dbeta + d + e, ab[c][e]);
            }
        }
    }
}

// Rows wider than 4096 (a 13B language model's 5120 in front of CTP): the same gradients without the row or the column partials in
// registers -- dx from three passes over the row (L2-resident), {mean, rstd} left per row; dgamma / dbeta by a column-sum kernel over
// dy * xhat and dy.
__global__ __launch_bounds__(256) void wg_layernorm_bwd_wide_dx_kernel(const bf16* x, long ldx, const bf16* gamma, const bf16* dy, long lddy, bf16* dx,
                                                                       long lddx, float* stats, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const bf16* xr = x + (long)m * ldx;
    const bf16* gr = dy + (long)m * lddy;
    const float invc = 1.0f / (float)C;
    float s = 0.f;
    for (int d = lane * 8; d < C; d += 512) {
        const bf16x8 t = *(const bf16x8*)(xr + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)t[e];
    }
    const float mean = wg_wave_sum(s) * invc;
    float q = 0.f;
    for (int d = lane * 8; d < C; d += 512) {
        const bf16x8 t = *(const bf16x8*)(xr + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = (float)t[e] - mean; q += v * v; }
    }
    const float rstd = 1.0f / sqrtf(wg_wave_sum(q) * invc + eps);
    float sg = 0.f, sgx = 0.f;
    for (int d = lane * 8; d < C; d += 512) {
        const bf16x8 t = *(const bf16x8*)(xr + d), g = *(const bf16x8*)(gr + d), gm = *(const bf16x8*)(gamma + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gg = (float)g[e] * (float)gm[e];
            sg += gg;
            sgx += gg * ((float)t[e] - mean) * rstd;
        }
    }
    const float mg = wg_wave_sum(sg) * invc, mgx = wg_wave_sum(sgx) * invc;
    bf16* dr = dx + (long)m * lddx;
    for (int d = lane * 8; d < C; d += 512) {
        const bf16x8 t = *(const bf16x8*)(xr + d), g = *(const bf16x8*)(gr + d), gm = *(const bf16x8*)(gamma + d);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(rstd * ((float)g[e] * (float)gm[e] - mg - ((float)t[e] - mean) * rstd * mgx));
        *(bf16x8*)(dr + d) = o;
    }
    if (lane == 0) { stats[2 * (long)m] = mean; stats[2 * (long)m + 1] = rstd; }
}
__global__ __launch_bounds__(256) void wg_layernorm_bwd_wide_cols_kernel(const bf16* x, long ldx, const bf16* dy, long lddy, const float* stats, float* dgamma,
                                                                         float* dbeta, int M, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 512 + lane * 8;
    if (c >= C) return;
    float ag[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ab[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int r0 = blockIdx.y * 256;
    const int r1 = r0 + 256 < M ? r0 + 256 : M;
    for (int r = r0 + wave; r < r1; r += 4) {
        const bf16x8 t = *(const bf16x8*)(x + (long)r * ldx + c), g = *(const bf16x8*)(dy + (long)r * lddy + c);
        const float mean = stats[2 * (long)r], rstd = stats[2 * (long)r + 1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ag[e] += (float)g[e] * ((float)t[e] - mean) * rstd;
            ab[e] += (float)g[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        atomicAdd(dgamma + c + e, ag[e]);
        atomicAdd(dbeta + c + e, ab[e]);
    }
}

// ---- y = x / max(|x|, eps) * exp(t): the tail of CalibratedTextProjector behind its LayerNorm and type embedding (utils_walkgpt.py:325-327:
// F.normalize(dim=-1, eps 1e-12) * log_temp.exp()).  One wave per row, C <= 512.
//   forward:  y = x e^t / n,  n = max(|x|, eps)
//   backward: dx = e^t / n (dy - xh (xh . dy)),  xh = x / n   (for |x| >= eps);   dt += sum_rows dy . y
template <bool BWD>
__global__ __launch_bounds__(256) void wg_l2norm_scale_kernel(const bf16* x, const bf16* dy, const bf16* log_temp, bf16* out, float* dtemp, int M, int C,
                                                              float eps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int d = lane * 8;
    const bool on = d < C;
    float v[8], g[8];
    float n2 = 0.f, dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = g[e] = 0.f;
    if (on) {
        const bf16x8 t = *(const bf16x8*)(x + (long)m * C + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = (float)t[e]; n2 += v[e] * v[e]; }
        if (BWD) {
            const bf16x8 u = *(const bf16x8*)(dy + (long)m * C + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) { g[e] = (float)u[e]; dot += g[e] * v[e]; }
        }
    }
    const float nrm = fmaxf(sqrtf(wg_wave_sum(n2)), eps);
    const float et = __expf((float)log_temp[0]), k = et / nrm;
    if (BWD) {
        dot = wg_wave_sum(dot);                               // x . dy
        if (lane == 0) atomicAdd(dtemp, dot * k);             // dy . y
        const float c = dot / (nrm * nrm);
        if (on) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)(k * (g[e] - v[e] * c));
            *(bf16x8*)(out + (long)m * C + d) = o;
        }
    } else if (on) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(v[e] * k);
        *(bf16x8*)(out + (long)m * C + d) = o;
    }
}

// ---- attention backward for the head's small attentions ---------------------------------------------------------------------------------------
// o = softmax(scale q k^T) v per (batch, head); q [B, Lq, D], k / v [B, Lk, D], D = H * hd, all contiguous bf16.  The trainable attentions
// (two-way transformer: 6 tokens <-> 4096 image tokens, transformer.py:185-240; MSQP's CrossAttnBlock: 4..12 queries, utils_walkgpt.py:163-185;
// TinyCrossAttn: 1 query) always have ONE short side (<= 16 rows).  A lane owns a row of the LONG side and keeps its 16 scores in registers:
//   BYQ  (few keys):    lane = query i.  softmax over its <= 16 keys locally; dq_i written directly; dk / dv are sums over lanes (wave
//                       reduction -> LDS -> one fp32 atomic per element and workgroup).
//   BYKEY (few queries): lane = key j.  The softmax runs over ALL keys: log-sum-exp and D_i = dO_i . O_i of every query come from a pre-pass
//                       (wg_attn_rowstats_kernel); dk_j / dv_j written directly; dq is the sum over lanes.
// fp32 arithmetic; the short side's rows sit in LDS as fp32.
constexpr int ATT_S = 16;     // rows of the short side
constexpr int ATT_HD = 128;   // head dim limit

__global__ __launch_bounds__(256) void wg_attn_rowstats_kernel(const bf16* q, const bf16* k, const bf16* o, const bf16* dout, float* stats, int H, int hd,
                                                               int Lq, int Lk, float scale) {
    // one workgroup per (batch*head, query): stats[.][0] = log sum_j exp(s_ij), stats[.][1] = dO_i . O_i
    __shared__ float qs[ATT_HD];
    __shared__ float red[4];
    const int i = blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh % H;
    const int D = H * hd;
    const long qoff = ((long)b * Lq + i) * D + h * hd;
    if (threadIdx.x < hd) qs[threadIdx.x] = (float)q[qoff + threadIdx.x];
    __syncthreads();
    float m = -3.0e38f, l = 0.f;
    for (int j = threadIdx.x; j < Lk; j += 256) {
        const bf16* kr = k + ((long)b * Lk + j) * D + h * hd;
        float s = 0.f;
        for (int d = 0; d < hd; d += 8) {
            const bf16x8 t = *(const bf16x8*)(kr + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += qs[d + e] * (float)t[e];
        }
        s *= scale;
        const float mn = fmaxf(m, s);
        l = l * __expf(m - mn) + __expf(s - mn);
        m = mn;
    }
    // combine (m, l) over the workgroup
    const float wm = wg_wave_max(m);
    l = wg_wave_sum(l * __expf(m - wm));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wm;
    __syncthreads();
    const float gm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l * __expf(wm - gm);
    __syncthreads();
    if (threadIdx.x == 0) {
        const float lt = red[0] + red[1] + red[2] + red[3];
        float dsum = 0.f;
        for (int d = 0; d < hd; ++d) dsum += (float)dout[qoff + d] * (float)o[qoff + d];
        stats[((long)bh * Lq + i) * 2 + 0] = gm + __logf(lt);
        stats[((long)bh * Lq + i) * 2 + 1] = dsum;
    }
}

// a: the long side's rows this workgroup's lanes own (queries for BYQ, keys for BYKEY); s_*: the short side
template <bool BYKEY>
__global__ __launch_bounds__(256) void wg_attn_bwd_kernel(const bf16* q, const bf16* k, const bf16* v, const bf16* dout, const float* stats, bf16* dlong_a,
                                                          bf16* dlong_b, float* part_a, float* part_b, long plane, int H, int hd, int Lq, int Lk, float scale) {
    // BYQ:   dlong_a = dq (bf16, direct);           part_a = dk, part_b = dv partials
    // BYKEY: dlong_a = dk, dlong_b = dv (direct);    part_a = dq partials
    // The short side's gradient is a sum over every row of the long side.  Round 4, no atomics: each wave adds into LDS rows of its own, the four waves'
    // rows are summed in wave order, and the workgroup leaves ONE partial [B, Ls, D] plane (part_*[blockIdx.x * plane + ...], output layout) for
    // wg_fold_rows_kernel to sum in workgroup order -- the same bits every run.  (LDS sized by the head dim: dynamic.)
    extern __shared__ float att_smem[];
    float* s_a = att_smem;                        // [ATT_S][hd]  BYQ: K rows          BYKEY: Q rows
    float* s_b = s_a + ATT_S * hd;                // [ATT_S][hd]  BYQ: V rows          BYKEY: dO rows
    float* acc_a = s_b + ATT_S * hd;              // [4 waves][ATT_S][hd]  BYQ: dK     BYKEY: dQ
    float* acc_b = acc_a + 4 * ATT_S * hd;        // [4 waves][ATT_S][hd]  BYQ: dV     (BYKEY: unused, not allocated)
    __shared__ float s_st[ATT_S][2];
    const int wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / H, h = bh % H, D = H * hd;
    const int Ll = BYKEY ? Lk : Lq, Ls = BYKEY ? Lq : Lk;
    const int lane = threadIdx.x & 63;
    const bf16* sa_src = BYKEY ? q : k;
    const bf16* sb_src = BYKEY ? dout : v;
    for (int t = threadIdx.x; t < Ls * hd; t += 256) {
        const int r = t / hd, d = t % hd;
        const long off = ((long)b * Ls + r) * D + h * hd + d;
        s_a[r * hd + d] = (float)sa_src[off];
        s_b[r * hd + d] = (float)sb_src[off];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            acc_a[(w * ATT_S + r) * hd + d] = 0.f;
            if (!BYKEY) acc_b[(w * ATT_S + r) * hd + d] = 0.f;
        }
    }
    if (BYKEY && threadIdx.x < Ls * 2) s_st[threadIdx.x >> 1][threadIdx.x & 1] = stats[((long)bh * Lq + (threadIdx.x >> 1)) * 2 + (threadIdx.x & 1)];
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;
    const bool on = row < Ll;
    const int rr = on ? row : Ll - 1;
    const long loff = ((long)b * Ll + rr) * D + h * hd;
    // BYQ: la = q_i, lb = dO_i;   BYKEY: la = k_j, lb = v_j
    const bf16* la = (BYKEY ? k : q) + loff;
    const bf16* lb = (BYKEY ? v : dout) + loff;
    float sc[ATT_S], dp[ATT_S];
#pragma unroll
    for (int r = 0; r < ATT_S; ++r) sc[r] = dp[r] = 0.f;
    for (int d = 0; d < hd; d += 8) {
        const bf16x8 ta = *(const bf16x8*)(la + d), tb = *(const bf16x8*)(lb + d);
#pragma unroll
        for (int r = 0; r < ATT_S; ++r) {
            if (r < Ls) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (BYKEY) {               // s_ij = q_i . k_j ;  dp_ij = dO_i . v_j
                        sc[r] += s_a[r * hd + d + e] * (float)ta[e];
                        dp[r] += s_b[r * hd + d + e] * (float)tb[e];
                    } else {                   // s_ij = q_i . k_j ;  dp_ij = dO_i . v_j
                        sc[r] += (float)ta[e] * s_a[r * hd + d + e];
                        dp[r] += (float)tb[e] * s_b[r * hd + d + e];
                    }
                }
            }
        }
    }
    float pr[ATT_S], ds[ATT_S];
    if (BYKEY) {
#pragma unroll
        for (int r = 0; r < ATT_S; ++r) {
            pr[r] = (r < Ls && on) ? __expf(sc[r] * scale - s_st[r][0]) : 0.f;
            ds[r] = pr[r] * (dp[r] - (r < Ls ? s_st[r][1] : 0.f)) * scale;
        }
    } else {
        float m = -3.0e38f, l = 0.f, dsum = 0.f;
#pragma unroll
        for (int r = 0; r < ATT_S; ++r)
            if (r < Ls) m = fmaxf(m, sc[r] * scale);
#pragma unroll
        for (int r = 0; r < ATT_S; ++r) {
            pr[r] = r < Ls ? __expf(sc[r] * scale - m) : 0.f;
            l += pr[r];
        }
        const float il = on ? 1.0f / l : 0.f;
#pragma unroll
        for (int r = 0; r < ATT_S; ++r) { pr[r] *= il; dsum += pr[r] * dp[r]; }
#pragma unroll
        for (int r = 0; r < ATT_S; ++r) ds[r] = pr[r] * (dp[r] - dsum) * scale;
    }
    // second sweep over the head dim: the long side's own gradient (direct) and the short side's (reduced over the lanes)
    for (int d = 0; d < hd; d += 8) {
        const bf16x8 ta = *(const bf16x8*)(la + d), tb = *(const bf16x8*)(lb + d);
        float o1[8], o2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o1[e] = o2[e] = 0.f;
#pragma unroll
        for (int r = 0; r < ATT_S; ++r) {
            if (r < Ls) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (BYKEY) {
                        o1[e] += ds[r] * s_a[r * hd + d + e];          // dk_j = sum_i ds_ij q_i
                        o2[e] += pr[r] * s_b[r * hd + d + e];          // dv_j = sum_i p_ij dO_i
                        const float c = wg_wave_sum(ds[r] * (float)ta[e]);      // dq_i += sum_j ds_ij k_j
                        if (lane == 0) acc_a[(wave * ATT_S + r) * hd + d + e] += c;
                    } else {
                        o1[e] += ds[r] * s_a[r * hd + d + e];          // dq_i = sum_j ds_ij k_j
                        const float c1 = wg_wave_sum(ds[r] * (float)ta[e]);     // dk_j += sum_i ds_ij q_i
                        const float c2 = wg_wave_sum(pr[r] * (float)tb[e]);     // dv_j += sum_i p_ij dO_i
                        if (lane == 0) { acc_a[(wave * ATT_S + r) * hd + d + e] += c1; acc_b[(wave * ATT_S + r) * hd + d + e] += c2; }
                    }
                }
            }
        }
        if (on) {
            bf16x8 w1, w2;
#pragma unroll
            for (int e = 0; e < 8; ++e) { w1[e] = (bf16)o1[e]; w2[e] = (bf16)o2[e]; }
            *(bf16x8*)(dlong_a + loff + d) = w1;
            if (BYKEY) *(bf16x8*)(dlong_b + loff + d) = w2;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < Ls * hd; t += 256) {
        const int r = t / hd, d = t % hd;
        const long off = ((long)b * Ls + r) * D + h * hd + d;
        const int f = r * hd + d, ws = ATT_S * hd;
        part_a[(long)blockIdx.x * plane + off] = acc_a[f] + acc_a[ws + f] + acc_a[2 * ws + f] + acc_a[3 * ws + f];
        if (!BYKEY) part_b[(long)blockIdx.x * plane + off] = acc_b[f] + acc_b[ws + f] + acc_b[2 * ws + f] + acc_b[3 * ws + f];
    }
}

// ---- MSQP pieces (utils_walkgpt.py:195-217,256-257) and the path from the language model's input back to the projector ------------------------
// avg-pool s x s over the token grid: dx[b, y, x, :] = dy[b, y / s, x / s, :] / s^2
__global__ __launch_bounds__(256) void wg_avgpool_bwd_kernel(const bf16* dy, bf16* dx, int B, int H, int W, int C, int s) {
    const int Ho = H / s, Wo = W / s, cpr = C / 8;
    const long total = (long)B * H * W * cpr;
    const float inv = 1.0f / (float)(s * s);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cpr) * 8;
        const long t = i / cpr;
        const int x = (int)(t % W), y = (int)((t / W) % H), b = (int)(t / ((long)W * H));
        bf16x8 o;
        if (y / s < Ho && x / s < Wo) {
            const bf16x8 g = *(const bf16x8*)(dy + (((long)b * Ho + y / s) * Wo + x / s) * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)g[e] * inv);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)0.f;
        }
        *(bf16x8*)(dx + t * C + c) = o;
    }
}
// mean over the L tokens: dx[b, r, :] = dy[b, :] / L
__global__ __launch_bounds__(256) void wg_mean_tokens_bwd_kernel(const bf16* dy, bf16* dx, int B, int L, int C) {
    const int cpr = C / 8;
    const long total = (long)B * L * cpr;
    const float inv = 1.0f / (float)L;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cpr) * 8;
        const long t = i / cpr;
        const int b = (int)(t / L);
        const bf16x8 g = *(const bf16x8*)(dy + (long)b * C + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)g[e] * inv);
        *(bf16x8*)(dx + t * C + c) = o;
    }
}
// y = x sigmoid(l):  dx = dy sigmoid(l),  dl[r] = sigmoid'(l) sum_c dy x   (a wave per row)
__global__ __launch_bounds__(256) void wg_gate_bwd_kernel(const bf16* x, const float* logit, const bf16* dy, bf16* dx, float* dlogit, long rows, int C) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        const float g = 1.0f / (1.0f + __expf(-logit[r]));
        float dot = 0.f;
        for (int c = lane * 8; c < C; c += 512) {
            const bf16x8 v = *(const bf16x8*)(x + r * C + c), d = *(const bf16x8*)(dy + r * C + c);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) { dot += (float)v[e] * (float)d[e]; o[e] = (bf16)((float)d[e] * g); }
            *(bf16x8*)(dx + r * C + c) = o;
        }
        dot = wg_wave_sum(dot);
        if (lane == 0) dlogit[r] = dot * g * (1.0f - g);
    }
}
// adjoint of the token resample (llava_arch.py:252-259; projector.hip wg_resample_kernel): [n, t*t, C] -> += into fp32 [n, p*p, C]
__global__ __launch_bounds__(256) void wg_resample_bwd_kernel(const bf16* dy, float* dx, int n, int p, int t, int C) {
    const int cpr = C / 8;
    const long total = (long)n * t * t * cpr;
    const float sc = (float)p / (float)t;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cpr) * 8;
        const long cell = i / cpr;
        const int ox = (int)(cell % t), oy = (int)((cell / t) % t), b = (int)(cell / ((long)t * t));
        float sy = __fsub_rn(__fmul_rn(sc, (float)oy + 0.5f), 0.5f), sx = __fsub_rn(__fmul_rn(sc, (float)ox + 0.5f), 0.5f);
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 < p - 1 ? y0 : p - 1;
        x0 = x0 < p - 1 ? x0 : p - 1;
        const int y1 = y0 + (y0 < p - 1 ? 1 : 0), x1 = x0 + (x0 < p - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const bf16x8 g = *(const bf16x8*)(dy + cell * C + c);
        float* base = dx + (long)b * p * p * C + c;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (float)g[e];
            atomicAdd(base + (long)(y0 * p + x0) * C + e, v * (1.f - ly) * (1.f - lx));
            atomicAdd(base + (long)(y0 * p + x1) * C + e, v * (1.f - ly) * lx);
            atomicAdd(base + (long)(y1 * p + x0) * C + e, v * ly * (1.f - lx));
            atomicAdd(base + (long)(y1 * p + x1) * C + e, v * ly * lx);
        }
    }
}
// adjoint of the multimodal splice (splice.hip wg_splice_gather_kernel): a workgroup per spliced token sends its gradient row to the image
// feature it was copied from (bf16, one writer) or adds it to its embedding row (fp32 atomics: a token id occurs many times)
__global__ __launch_bounds__(256) void wg_splice_bwd_kernel(const long* ids, const int* img_pos, const bf16* dembeds, bf16* dimg, float* dtable, int L, int T,
                                                            int H, int V) {
    const int Lo = L + T - 1;
    const int r = blockIdx.x / Lo, j = blockIdx.x % Lo;
    const int s = img_pos[r];
    const bool is_img = j >= s && j < s + T;
    const bf16* src = dembeds + ((long)r * Lo + j) * H;
    if (is_img) {
        bf16* dst = dimg + ((long)r * T + (j - s)) * H;
        for (int d = threadIdx.x * 8; d < H; d += 2048) *(bf16x8*)(dst + d) = *(const bf16x8*)(src + d);
    } else if (dtable) {
        const int i = j < s ? j : j - T + 1;
        const long id = ids[(long)r * L + i];
        if (id < 0 || id >= V) return;
        float* dst = dtable + id * H;
        for (int d = threadIdx.x * 8; d < H; d += 2048) {
            const bf16x8 g = *(const bf16x8*)(src + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(dst + d + e, (float)g[e]);
        }
    }
}

// ---- region-alignment InfoNCE (utils_walkgpt.py:8-73, top_k form) as differentiable pieces -------------------------------------------------
// (1) top-k pooled positive: u [M, D] (= W_k^T W_q z: the folded query of TinyCrossAttn), kt [M, Kt, D] the Kt raw SAM tokens with the largest
//     attention weight (constants).  alpha = the attention weights renormalised over those Kt = softmax of their scores; v = sum alpha_k kt_k.
//     A wave per row; D <= 512, Kt <= 16.  Backward: du = sum_k ds_k kt_k / sqrt(D), ds = alpha (da - sum alpha da), da_k = dv . kt_k.
template <bool BWD>
__global__ __launch_bounds__(256) void wg_topk_pool_kernel(const bf16* u, const bf16* kt, const bf16* dv, bf16* out, int M, int Kt, int D) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int d = lane * 8;
    const bool on = d < D;
    float uv[8], gv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) uv[e] = gv[e] = 0.f;
    if (on) {
        const bf16x8 t = *(const bf16x8*)(u + (long)m * D + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) uv[e] = (float)t[e];
        if (BWD) {
            const bf16x8 g = *(const bf16x8*)(dv + (long)m * D + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] = (float)g[e];
        }
    }
    const float scale = 1.0f / sqrtf((float)D);
    float sc[16], da[16];
    float mx = -3.0e38f;
    for (int k = 0; k < 16; ++k) {
        sc[k] = -3.0e38f; da[k] = 0.f;
        if (k < Kt) {
            float s = 0.f, a = 0.f;
            if (on) {
                const bf16x8 t = *(const bf16x8*)(kt + ((long)m * Kt + k) * D + d);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s += uv[e] * (float)t[e]; a += gv[e] * (float)t[e]; }
            }
            sc[k] = wg_wave_sum(s) * scale;
            da[k] = BWD ? wg_wave_sum(a) : 0.f;
            mx = fmaxf(mx, sc[k]);
        }
    }
    float l = 0.f;
    for (int k = 0; k < 16; ++k) { sc[k] = k < Kt ? __expf(sc[k] - mx) : 0.f; l += sc[k]; }
    float dsum = 0.f;
    for (int k = 0; k < 16; ++k) { sc[k] /= l; dsum += sc[k] * da[k]; }
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (on) {
        for (int k = 0; k < Kt; ++k) {
            const bf16x8 t = *(const bf16x8*)(kt + ((long)m * Kt + k) * D + d);
            const float w = BWD ? sc[k] * (da[k] - dsum) * scale : sc[k];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += w * (float)t[e];
        }
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (bf16)o[e];
        *(bf16x8*)(out + (long)m * D + d) = r;
    }
}

// (1b) the same pooling over ANY number of tokens (top_k > 16, or no refinement at all: TinyCrossAttn's own softmax over the N tokens of the
//     row, utils_walkgpt.py:338-356 with W_v / out applied to the pooled raw token afterwards): tokens [rows, Kt, D], query m pools row
//     row_of[m] (null: row m).  A workgroup per query, its four waves take keys wave, wave + 4, ... with an online softmax each and meet in LDS.
//     Forward: one pass.  Backward: pass 1 leaves the softmax statistics and dsum = sum_k p_k da_k, pass 2 forms du = sum_k p_k (da_k - dsum) kt_k
//     / sqrt(D) (scores and da recomputed: nothing of size Kt is stored).
template <bool BWD>
__global__ __launch_bounds__(256) void wg_pool_rows_kernel(const bf16* u, const bf16* tokens, const int* row_of, const bf16* dv, bf16* out, int M, int Kt, int D) {
    __shared__ float red[4][4];            // per wave: max, sum, weighted da sum
    __shared__ float acc[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x;
    const int d = lane * 8 < D ? lane * 8 : 0;
    const bool on = lane * 8 < D;
    const bf16* kt = tokens + (long)(row_of ? row_of[m] : m) * Kt * D;
    float uv[8], gv[8];
    {
        const bf16x8 t = *(const bf16x8*)(u + (long)m * D + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) { uv[e] = on ? (float)t[e] : 0.f; gv[e] = 0.f; }
        if (BWD) {
            const bf16x8 g = *(const bf16x8*)(dv + (long)m * D + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] = on ? (float)g[e] : 0.f;
        }
    }
    const float scale = 1.0f / sqrtf((float)D);
    // ---- pass 1: online softmax over this wave's keys; forward accumulates the pooled token, backward the da-weighted sum ---------------
    float mx = -3.0e38f, l = 0.f, wda = 0.f;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = wave; k < Kt; k += 4) {
        const bf16x8 t = *(const bf16x8*)(kt + (long)k * D + d);
        float s = 0.f, a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s += uv[e] * (float)t[e]; a += gv[e] * (float)t[e]; }
        s = wg_wave_sum(s) * scale;
        if (BWD) a = wg_wave_sum(a);
        const float mn = fmaxf(mx, s);
        const float r = __expf(mx - mn), w = __expf(s - mn);
        l = l * r + w;
        if (BWD) wda = wda * r + w * a;
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = o[e] * r + w * (float)t[e];
        }
        mx = mn;
    }
    if (lane == 0) { red[wave][0] = mx; red[wave][1] = l; red[wave][2] = wda; }
    __syncthreads();
    float gm = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    float gl = 0.f, gda = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float f = __expf(red[w][0] - gm);      // (a wave without keys: max -3e38, weight 0)
        gl += red[w][1] * f;
        gda += red[w][2] * f;
    }
    if (!BWD) {
        const float f = __expf(mx - gm) / gl;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[wave][lane * 8 + e] = o[e] * f;
    } else {
        // ---- pass 2 (backward): du = sum_k p_k (da_k - dsum) kt_k / sqrt(D) ---------------------------------------------------------------
        const float dsum = gda / gl, inv = 1.0f / gl;
        for (int k = wave; k < Kt; k += 4) {
            const bf16x8 t = *(const bf16x8*)(kt + (long)k * D + d);
            float s = 0.f, a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { s += uv[e] * (float)t[e]; a += gv[e] * (float)t[e]; }
            s = wg_wave_sum(s) * scale;
            a = wg_wave_sum(a);
            const float w = __expf(s - gm) * inv * (a - dsum) * scale;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += w * (float)t[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[wave][lane * 8 + e] = o[e];
    }
    __syncthreads();
    if (wave == 0 && on) {
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (bf16)(acc[0][lane * 8 + e] + acc[1][lane * 8 + e] + acc[2][lane * 8 + e] + acc[3][lane * 8 + e]);
        *(bf16x8*)(out + (long)m * D + lane * 8) = r;
    }
}

// (2) the loss tail: pos_m = z_m . vp_m, logits_m = [pos_m, sim[m, :]] / T with the columns of m's own row masked, loss = mean_m (lse_m - pos_m / T).
//     A workgroup per m.  Forward leaves loss_m and lse_m; backward: p_j = exp(logit_j - lse):  dsim[m, j] = g p_j / (T M),
//     dz_m (positive term only) = g (p_0 - 1) / (T M) vp_m,  dvp_m = g (p_0 - 1) / (T M) z_m.
template <bool BWD>
__global__ __launch_bounds__(256) void wg_nce_tail_kernel(const bf16* z, const bf16* vp, const float* sim, const int* own_row, float* loss_m, float* lse_m,
                                                          float g, bf16* dz, bf16* dvp, float* dsim, int M, int rows, int N, int D, float inv_t, int exclude,
                                                          const float* g_dev = nullptr) {
    __shared__ float red[8];
    if (BWD && g_dev) g = g_dev[0];      // upstream gradient read on the device (no host synchronisation in the caller's backward pass)
    const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long R = (long)rows * N;
    float pd = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) pd += (float)z[(long)m * D + d] * (float)vp[(long)m * D + d];
    pd = wg_wave_sum(pd);
    if (lane == 0) red[wave] = pd;
    __syncthreads();
    const float pos = (red[0] + red[1] + red[2] + red[3]) * inv_t;
    __syncthreads();
    const long own0 = exclude ? (long)own_row[m] * N : -1, own1 = exclude ? own0 + N : -1;
    const float* sr = sim + (long)m * R;
    if (!BWD) {
        float mx = pos;
        for (long j = threadIdx.x; j < R; j += 256)
            if (j < own0 || j >= own1) mx = fmaxf(mx, sr[j] * inv_t);
        mx = wg_wave_max(mx);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        float l = 0.f;
        for (long j = threadIdx.x; j < R; j += 256)
            if (j < own0 || j >= own1) l += __expf(sr[j] * inv_t - mx);
        l = wg_wave_sum(l);
        if (lane == 0) red[wave] = l;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float lse = mx + __logf(red[0] + red[1] + red[2] + red[3] + __expf(pos - mx));
            lse_m[m] = lse;
            loss_m[m] = lse - pos;
        }
    } else {
        const float lse = lse_m[m], k = g * inv_t / (float)M;
        for (long j = threadIdx.x; j < R; j += 256) dsim[(long)m * R + j] = (j < own0 || j >= own1) ? k * __expf(sr[j] * inv_t - lse) : 0.f;
        const float c = k * (__expf(pos - lse) - 1.0f);
        for (int d = threadIdx.x; d < D; d += 256) {
            dz[(long)m * D + d] = (bf16)(c * (float)vp[(long)m * D + d]);
            dvp[(long)m * D + d] = (bf16)(c * (float)z[(long)m * D + d]);
        }
    }
}

// ---- masks = hyper_in @ upscaled (mask_decoder.py:150-160) on channels-last rows, all prompts in one launch, and its gradients ------------------
// up [P, HW, C] bf16 (C = 32: the upscaled embedding, one row per output pixel), hyper [P, K, C] bf16 (K <= 4 mask tokens' hypernetwork outputs)
// -> masks [P, K, HW] fp32.  A thread per pixel, the prompt's K hyper rows in LDS.
// Backward: dup[p, x, :] = sum_k dm[p, k, x] hyper[p, k, :];  dhyper[p, k, :] = sum_x dm[p, k, x] up[p, x, :] (one partial per workgroup, folded in a fixed order: no atomics).
constexpr int HM_C = 32, HM_K = 4;
template <int MODE>   // 0 forward, 1 backward
__global__ __launch_bounds__(256) void wg_hyper_rows_kernel(const bf16* up, const bf16* hyper, const float* dm, float* masks, bf16* dup, float* dhyper, int HW,
                                                            int K) {
    __shared__ float hs[HM_K][HM_C];
    __shared__ float red[4][HM_K][HM_C];
    const int p = blockIdx.y;
    // every row of hs is written: the unrolled sums below run over all HM_K rows with g[k] = 0 for k >= K, and 0 * (whatever bit pattern
    // an earlier kernel left in LDS, NaN included) must not reach dup
    if (threadIdx.x < HM_K * HM_C) hs[threadIdx.x / HM_C][threadIdx.x % HM_C] = threadIdx.x < K * HM_C ? (float)hyper[((long)p * K) * HM_C + threadIdx.x] : 0.f;
    __syncthreads();
    float acc[HM_K][HM_C];
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < HM_K; ++k)
#pragma unroll
            for (int c = 0; c < HM_C; ++c) acc[k][c] = 0.f;
    }
    for (int x = blockIdx.x * 256 + threadIdx.x; x < HW; x += gridDim.x * 256) {
        float u[HM_C];
        const bf16* ur = up + ((long)p * HW + x) * HM_C;
#pragma unroll
        for (int c = 0; c < HM_C; c += 8) {
            const bf16x8 t = *(const bf16x8*)(ur + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) u[c + e] = (float)t[e];
        }
        if (MODE == 0) {
            for (int k = 0; k < K; ++k) {
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < HM_C; ++c) d += u[c] * hs[k][c];
                masks[((long)p * K + k) * HW + x] = d;
            }
        } else {
            float g[HM_K];
#pragma unroll
            for (int k = 0; k < HM_K; ++k) g[k] = k < K ? dm[((long)p * K + k) * HW + x] : 0.f;
            bf16* dr = dup + ((long)p * HW + x) * HM_C;
#pragma unroll
            for (int c = 0; c < HM_C; c += 8) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float d = 0.f;
#pragma unroll
                    for (int k = 0; k < HM_K; ++k) d += g[k] * hs[k][c + e];
                    o[e] = (bf16)d;
                }
                *(bf16x8*)(dr + c) = o;
            }
#pragma unroll
            for (int k = 0; k < HM_K; ++k)
#pragma unroll
                for (int c = 0; c < HM_C; ++c) acc[k][c] += g[k] * u[c];
        }
    }
    if (MODE == 1) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < HM_K; ++k)
#pragma unroll
            for (int c = 0; c < HM_C; ++c) {
                const float v = wg_wave_sum(acc[k][c]);
                if (lane == 0) red[wave][k][c] = v;
            }
        __syncthreads();
        if (threadIdx.x < K * HM_C) {       // this workgroup's partial of dhyper[p]: plane blockIdx.x of [gridDim.x][P * K * C] (summed in order by wg_fold_rows_kernel)
            const int k = threadIdx.x / HM_C, c = threadIdx.x % HM_C;
            dhyper[((long)blockIdx.x * gridDim.y + p) * K * HM_C + k * HM_C + c] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
        }
    }
}

}  // namespace

// Test support: leave `pattern` in the first 64 KiB of every compute unit's LDS (a kernel's LDS is not cleared between launches; tests use
// this to prove that a kernel reads no LDS word it has not written: tests/test_gpu_backward.py poisons with a NaN pattern).
__global__ __launch_bounds__(256) void wg_fill_lds_kernel(unsigned pattern, unsigned* sink) {
    __shared__ unsigned buf[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = pattern;
    __syncthreads();
    if (sink && buf[(threadIdx.x * 61) & 16383] != pattern) *sink = 1;   // keeps the stores alive
}
extern "C" int wg_debug_fill_lds_u32(unsigned pattern, void* sink, void* stream) {
    hipLaunchKernelGGL(wg_fill_lds_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, pattern, (unsigned*)sink);
    return wg_check_launch("wg_debug_fill_lds_u32");
}

extern "C" int wg_hyper_rows_f32(const void* up, const void* hyper, float* masks, int P, int HW, int C, int K, void* stream) {
    WG_REQUIRE(up && hyper && masks && P > 0 && HW > 0 && C == HM_C && K > 0 && K <= HM_K, "hyper_rows: C must be %d, K <= %d", HM_C, HM_K);
    int gx = (HW + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(wg_hyper_rows_kernel<0>, dim3(gx, P), dim3(256), 0, (hipStream_t)stream, (const bf16*)up, (const bf16*)hyper, nullptr, masks, nullptr, nullptr, HW, K);
    return wg_check_launch("wg_hyper_rows_f32");
}

// dhyper [P, K, C] fp32 is WRITTEN (round 4: fixed-order sums, no atomics); workspace: wg_hyper_rows_bwd_workspace_floats(P, HW, K) floats.
static int wg_hyper_gx(int HW) { const int gx = (HW + 255) / 256; return gx > 64 ? 64 : gx; }
extern "C" long wg_hyper_rows_bwd_workspace_floats(int P, int HW, int K) { return (long)wg_hyper_gx(HW) * P * K * HM_C; }
extern "C" int wg_hyper_rows_bwd_f32(const void* up, const void* hyper, const float* dmasks, void* dup, float* dhyper, float* workspace, long workspace_floats, int P, int HW,
                                     int C, int K, void* stream) {
    WG_REQUIRE(up && hyper && dmasks && dup && dhyper && workspace && P > 0 && HW > 0 && C == HM_C && K > 0 && K <= HM_K, "hyper_rows_bwd: C must be %d, K <= %d", HM_C, HM_K);
    const int gx = wg_hyper_gx(HW);
    WG_REQUIRE(workspace_floats >= (long)gx * P * K * HM_C, "hyper_rows_bwd: workspace too small (need %ld floats)", (long)gx * P * K * HM_C);
    hipLaunchKernelGGL(wg_hyper_rows_kernel<1>, dim3(gx, P), dim3(256), 0, (hipStream_t)stream, (const bf16*)up, (const bf16*)hyper, dmasks, nullptr, (bf16*)dup, workspace, HW, K);
    const long n = (long)P * K * HM_C;
    hipLaunchKernelGGL(wg_fold_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, gx, n, dhyper, 1);
    return wg_check_launch("wg_hyper_rows_bwd_f32");
}

extern "C" int wg_topk_pool_bf16(const void* u, const void* kt, void* v, int M, int Kt, int D, void* stream) {
    WG_REQUIRE(u && kt && v && M > 0 && Kt > 0 && Kt <= 16 && D % 8 == 0 && D <= 512, "topk_pool: Kt <= 16, D <= 512");
    hipLaunchKernelGGL(wg_topk_pool_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)u, (const bf16*)kt, nullptr, (bf16*)v, M, Kt, D);
    return wg_check_launch("wg_topk_pool_bf16");
}

extern "C" int wg_topk_pool_bwd_bf16(const void* u, const void* kt, const void* dv, void* du, int M, int Kt, int D, void* stream) {
    WG_REQUIRE(u && kt && dv && du && M > 0 && Kt > 0 && Kt <= 16 && D % 8 == 0 && D <= 512, "topk_pool_bwd: Kt <= 16, D <= 512");
    hipLaunchKernelGGL(wg_topk_pool_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)u, (const bf16*)kt, (const bf16*)dv, (bf16*)du, M, Kt, D);
    return wg_check_launch("wg_topk_pool_bwd_bf16");
}

extern "C" int wg_pool_rows_bf16(const void* u, const void* tokens, const int* row_of, void* v, int M, int Kt, int D, void* stream) {
    WG_REQUIRE(u && tokens && v && M > 0 && Kt > 0 && D % 8 == 0 && D > 0 && D <= 512, "pool_rows: D <= 512, a multiple of 8");
    hipLaunchKernelGGL(wg_pool_rows_kernel<false>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)u, (const bf16*)tokens, row_of, nullptr, (bf16*)v, M, Kt, D);
    return wg_check_launch("wg_pool_rows_bf16");
}

extern "C" int wg_pool_rows_bwd_bf16(const void* u, const void* tokens, const int* row_of, const void* dv, void* du, int M, int Kt, int D, void* stream) {
    WG_REQUIRE(u && tokens && dv && du && M > 0 && Kt > 0 && D % 8 == 0 && D > 0 && D <= 512, "pool_rows_bwd: D <= 512, a multiple of 8");
    hipLaunchKernelGGL(wg_pool_rows_kernel<true>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)u, (const bf16*)tokens, row_of, (const bf16*)dv, (bf16*)du, M, Kt, D);
    return wg_check_launch("wg_pool_rows_bwd_bf16");
}

extern "C" int wg_nce_tail_f32(const void* z, const void* vp, const float* sim, const int* own_row, float* loss_m, float* lse_m, int M, int rows, int N, int D,
                               float temperature, int exclude_same_row, void* stream) {
    WG_REQUIRE(z && vp && sim && own_row && loss_m && lse_m && M > 0 && rows > 0 && N > 0 && D > 0 && temperature > 0.f, "nce_tail: bad arguments");
    hipLaunchKernelGGL(wg_nce_tail_kernel<false>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)z, (const bf16*)vp, sim, own_row, loss_m, lse_m, 0.f,
                       nullptr, nullptr, nullptr, M, rows, N, D, 1.0f / temperature, exclude_same_row);
    return wg_check_launch("wg_nce_tail_f32");
}

extern "C" int wg_nce_tail_bwd_f32(const void* z, const void* vp, const float* sim, const int* own_row, const float* lse_m, float g, void* dz, void* dvp,
                                   float* dsim, int M, int rows, int N, int D, float temperature, int exclude_same_row, void* stream) {
    WG_REQUIRE(z && vp && sim && own_row && lse_m && dz && dvp && dsim && M > 0 && rows > 0 && N > 0 && D > 0 && temperature > 0.f, "nce_tail_bwd: bad arguments");
    hipLaunchKernelGGL(wg_nce_tail_kernel<true>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)z, (const bf16*)vp, sim, own_row, nullptr,
                       (float*)lse_m, g, (bf16*)dz, (bf16*)dvp, dsim, M, rows, N, D, 1.0f / temperature, exclude_same_row, nullptr);
    return wg_check_launch("wg_nce_tail_bwd_f32");
}
// ... with the upstream gradient in device memory (one float)
extern "C" int wg_nce_tail_bwd_dev_f32(const void* z, const void* vp, const float* sim, const int* own_row, const float* lse_m, const float* g, void* dz, void* dvp,
                                       float* dsim, int M, int rows, int N, int D, float temperature, int exclude_same_row, void* stream) {
    WG_REQUIRE(z && vp && sim && own_row && lse_m && g && dz && dvp && dsim && M > 0 && rows > 0 && N > 0 && D > 0 && temperature > 0.f, "nce_tail_bwd_dev: bad arguments");
    hipLaunchKernelGGL(wg_nce_tail_kernel<true>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)z, (const bf16*)vp, sim, own_row, nullptr,
                       (float*)lse_m, 0.f, (bf16*)dz, (bf16*)dvp, dsim, M, rows, N, D, 1.0f / temperature, exclude_same_row, g);
    return wg_check_launch("wg_nce_tail_bwd_dev_f32");
}

extern "C" int wg_avgpool_tokens_bwd_bf16(const void* dy, void* dx, int B, int H, int W, int C, int s, void* stream) {
    WG_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C % 8 == 0 && s > 0 && H % s == 0 && W % s == 0, "avgpool_bwd: bad arguments");
    const long total = (long)B * H * W * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wg_avgpool_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (bf16*)dx, B, H, W, C, s);
    return wg_check_launch("wg_avgpool_tokens_bwd_bf16");
}

extern "C" int wg_mean_tokens_bwd_bf16(const void* dy, void* dx, int B, int L, int C, void* stream) {
    WG_REQUIRE(dy && dx && B > 0 && L > 0 && C % 8 == 0, "mean_tokens_bwd: bad arguments");
    const long total = (long)B * L * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wg_mean_tokens_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (bf16*)dx, B, L, C);
    return wg_check_launch("wg_mean_tokens_bwd_bf16");
}

extern "C" int wg_sigmoid_gate_bwd_bf16(const void* x, const float* logit, const void* dy, void* dx, float* dlogit, long rows, int C, void* stream) {
    WG_REQUIRE(x && logit && dy && dx && dlogit && rows > 0 && C % 8 == 0, "sigmoid_gate_bwd: bad arguments");
    const int blocks = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
    hipLaunchKernelGGL(wg_gate_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, logit, (const bf16*)dy, (bf16*)dx, dlogit, rows, C);
    return wg_check_launch("wg_sigmoid_gate_bwd_bf16");
}

extern "C" int wg_resample_tokens_bwd_f32(const void* dy, float* dx, int n, int p, int t, int C, void* stream) {
    WG_REQUIRE(dy && dx && n > 0 && p > 0 && t > 0 && C % 8 == 0, "resample_tokens_bwd: bad arguments");
    const long total = (long)n * t * t * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wg_resample_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, dx, n, p, t, C);
    return wg_check_launch("wg_resample_tokens_bwd_f32");
}

extern "C" int wg_splice_multimodal_bwd_bf16(const long* ids, const int* img_pos, const void* dembeds, void* dimage_features, float* dtable, int rows,
                                             int L, int T, int H, int V, void* stream) {
    WG_REQUIRE(ids && img_pos && dembeds && dimage_features && rows > 0 && L > 0 && T > 0 && V > 0 && H > 0 && H % 8 == 0, "splice_bwd: bad arguments");
    hipLaunchKernelGGL(wg_splice_bwd_kernel, dim3((unsigned)(rows * (L + T - 1))), dim3(256), 0, (hipStream_t)stream, ids, img_pos, (const bf16*)dembeds,
                       (bf16*)dimage_features, dtable, L, T, H, V);
    return wg_check_launch("wg_splice_multimodal_bwd_bf16");
}

extern "C" int wg_l2norm_scale_bf16(const void* x, const void* log_temp, void* y, int M, int C, float eps, void* stream) {
    WG_REQUIRE(x && log_temp && y && M > 0 && C > 0 && C % 8 == 0 && C <= 512, "l2norm_scale: C = %d must be a multiple of 8, at most 512", C);
    hipLaunchKernelGGL(wg_l2norm_scale_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, nullptr, (const bf16*)log_temp,
                       (bf16*)y, nullptr, M, C, eps);
    return wg_check_launch("wg_l2norm_scale_bf16");
}

extern "C" int wg_l2norm_scale_bwd_bf16(const void* x, const void* dy, const void* log_temp, void* dx, float* dlog_temp, int M, int C, float eps,
                                        void* stream) {
    WG_REQUIRE(x && dy && log_temp && dx && dlog_temp && M > 0 && C > 0 && C % 8 == 0 && C <= 512, "l2norm_scale_bwd: C = %d must be a multiple of 8, at most 512", C);
    hipLaunchKernelGGL(wg_l2norm_scale_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)dy,
                       (const bf16*)log_temp, (bf16*)dx, dlog_temp, M, C, eps);
    return wg_check_launch("wg_l2norm_scale_bwd_bf16");
}

// q [B,Lq,D], k / v [B,Lk,D], o / dout [B,Lq,D] contiguous bf16 (D = heads * head_dim); min(Lq, Lk) <= 16, head_dim % 8 == 0, <= 128.
// dq / dk / dv: bf16 for the long side(s) written directly, fp32 (WRITTEN, round 4: fixed-order sums) for the short side: the caller passes BOTH forms
// for every gradient and reads the one that applies (wg_attn_bwd_short_side tells which side is short).
// workspace: wg_attn_bwd_workspace_floats(B, heads, head_dim, Lq, Lk) floats (softmax statistics + one partial plane per workgroup of long-side rows).
extern "C" int wg_attn_bwd_short_side(int Lq, int Lk) { return Lk <= ATT_S ? 1 : (Lq <= ATT_S ? 0 : -1); }   // 1: keys short, 0: queries short

extern "C" long wg_attn_bwd_workspace_floats(int B, int heads, int head_dim, int Lq, int Lk) {
    const int side = wg_attn_bwd_short_side(Lq, Lk);
    if (side < 0) return 0;
    const long D = (long)heads * head_dim, nblk = ((side == 1 ? Lq : Lk) + 255) / 256;
    const long plane = (long)B * (side == 1 ? Lk : Lq) * D;
    return (long)B * heads * Lq * 2 + nblk * plane * (side == 1 ? 2 : 1);
}

extern "C" int wg_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* dout, void* dq_bf16, void* dk_bf16, void* dv_bf16,
                                float* dq_f32, float* dk_f32, float* dv_f32, float* workspace, long workspace_floats, int B, int heads, int head_dim, int Lq, int Lk,
                                float scale, void* stream) {
    WG_REQUIRE(q && k && v && o && dout && workspace, "attn_bwd: null operand");
    WG_REQUIRE(B > 0 && heads > 0 && Lq > 0 && Lk > 0 && head_dim % 8 == 0 && head_dim <= ATT_HD, "attn_bwd: head_dim = %d must be a multiple of 8, at most %d", head_dim, ATT_HD);
    const int side = wg_attn_bwd_short_side(Lq, Lk);
    WG_REQUIRE(side >= 0, "attn_bwd: one side must have at most %d rows (Lq = %d, Lk = %d)", ATT_S, Lq, Lk);
    WG_REQUIRE(workspace_floats >= wg_attn_bwd_workspace_floats(B, heads, head_dim, Lq, Lk), "attn_bwd: workspace too small (need %ld floats)",
               wg_attn_bwd_workspace_floats(B, heads, head_dim, Lq, Lk));
    hipStream_t st = (hipStream_t)stream;
    float* stats = workspace;
    float* part = workspace + (long)B * heads * Lq * 2;
    const long D = (long)heads * head_dim;
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) {
        (void)hipFuncSetAttribute((const void*)wg_attn_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (2 + 8) * ATT_S * ATT_HD * 4);
        (void)hipFuncSetAttribute((const void*)wg_attn_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (2 + 4) * ATT_S * ATT_HD * 4);
    }
    if (side == 1) {   // few keys: a lane per query
        WG_REQUIRE(dq_bf16 && dk_f32 && dv_f32, "attn_bwd: dq (bf16), dk / dv (fp32) outputs");
        const int nblk = (Lq + 255) / 256;
        const long plane = (long)B * Lk * D;
        hipLaunchKernelGGL(wg_attn_bwd_kernel<false>, dim3(nblk, B * heads), dim3(256), (size_t)(2 + 8) * ATT_S * head_dim * 4, st, (const bf16*)q, (const bf16*)k,
                           (const bf16*)v, (const bf16*)dout, stats, (bf16*)dq_bf16, nullptr, part, part + nblk * plane, plane, heads, head_dim, Lq, Lk, scale);
        hipLaunchKernelGGL(wg_fold_rows_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, part, nblk, plane, dk_f32, 1);
        hipLaunchKernelGGL(wg_fold_rows_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, part + nblk * plane, nblk, plane, dv_f32, 1);
    } else {           // few queries: a lane per key, softmax statistics first
        WG_REQUIRE(dk_bf16 && dv_bf16 && dq_f32, "attn_bwd: dk / dv (bf16), dq (fp32) outputs");
        const int nblk = (Lk + 255) / 256;
        const long plane = (long)B * Lq * D;
        hipLaunchKernelGGL(wg_attn_rowstats_kernel, dim3(Lq, B * heads), dim3(256), 0, st, (const bf16*)q, (const bf16*)k, (const bf16*)o, (const bf16*)dout,
                           stats, heads, head_dim, Lq, Lk, scale);
        hipLaunchKernelGGL(wg_attn_bwd_kernel<true>, dim3(nblk, B * heads), dim3(256), (size_t)(2 + 4) * ATT_S * head_dim * 4, st, (const bf16*)q, (const bf16*)k,
                           (const bf16*)v, (const bf16*)dout, stats, (bf16*)dk_bf16, (bf16*)dv_bf16, part, nullptr, plane, heads, head_dim, Lq, Lk, scale);
        hipLaunchKernelGGL(wg_fold_rows_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, part, nblk, plane, dq_f32, 1);
    }
    return wg_check_launch("wg_attn_bwd_bf16");
}

extern "C" int wg_colsum_f32(const void* x, long ldx, float* out, int R, int C, void* stream) {
    WG_REQUIRE(x && out && R > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, "colsum: x [R, C] bf16 with C %% 8 == 0, 16-byte rows");
    hipLaunchKernelGGL(wg_colsum_kernel, dim3((C + 511) / 512, (R + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, out, R, C);
    return wg_check_launch("wg_colsum_f32");
}

// Column sums without atomics: out [C] is WRITTEN (fp32, or bf16 with out_f32 = 0); workspace: ceil(R / 256) * C floats.
extern "C" long wg_colsum_det_workspace_floats(int R, int C) { return (long)((R + 255) / 256) * C; }
extern "C" int wg_colsum_det_f32(const void* x, long ldx, void* out, int out_f32, float* workspace, long workspace_floats, int R, int C, void* stream) {
    WG_REQUIRE(x && out && workspace && R > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, "colsum_det: x [R, C] bf16 with C %% 8 == 0, 16-byte rows");
    const int rb = (R + 255) / 256;
    WG_REQUIRE(workspace_floats >= (long)rb * C && rb <= 65535, "colsum_det: workspace too small (need %ld floats)", (long)rb * C);
    hipLaunchKernelGGL(wg_colsum_part_kernel, dim3((C + 511) / 512, rb), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, workspace, R, C);
    hipLaunchKernelGGL(wg_fold_rows_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace, rb, (long)C, out, out_f32);
    return wg_check_launch("wg_colsum_det_f32");
}

extern "C" int wg_act_bf16(const void* x, void* y, long n, int act, void* stream) {
    WG_REQUIRE(x && y && n > 0 && n % 8 == 0 && act >= 0 && act <= 3 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "act: n %% 8 == 0, 16-byte aligned");
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wg_act_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, nullptr, (bf16*)y, n / 8, act);
    return wg_check_launch("wg_act_bf16");
}

extern "C" int wg_act_bwd_bf16(const void* x, const void* dy, void* dx, long n, int act, void* stream) {
    WG_REQUIRE(x && dy && dx && n > 0 && n % 8 == 0 && act >= 0 && act <= 3 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0,
               "act_bwd: n %% 8 == 0, 16-byte aligned");
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wg_act_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)dy, (bf16*)dx, n / 8, act);
    return wg_check_launch("wg_act_bwd_bf16");
}

// Rows of 64, 128, 256 or 512 channels, deterministic (no atomics): dgamma / dbeta are written (bf16, or fp32 with out_f32), not accumulated;
// workspace = wg_layernorm_bwd_det_workspace_floats(M, C) floats.  Returns -2 (nothing launched) for other widths: the caller falls back on
// wg_layernorm_bwd_bf16.
static int wg_lnb_blocks(int M) {
    int b = (M + 63) / 64;               // >= 64 rows (16 per wave) per workgroup ...
    return b > 1024 ? 1024 : b;          // ... and at most 1024 partial row pairs to fold
}
extern "C" long wg_layernorm_bwd_det_workspace_floats(int M, int C) { return (long)wg_lnb_blocks(M) * 2 * C; }
extern "C" int wg_layernorm_bwd_det_bf16(const void* x, long ldx, const void* gamma, const void* dy, long lddy, void* dx, long lddx, void* dgamma, void* dbeta,
                                         int out_f32, float* workspace, long workspace_floats, int M, int C, float eps, void* stream) {
    WG_REQUIRE(x && gamma && dy && dx && dgamma && dbeta && workspace && M > 0, "layernorm_bwd_det: null operand");
    if (!(C == 64 || C == 128 || C == 256 || C == 512)) return -2;
    WG_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0,
               "layernorm_bwd_det: misaligned operand");
    const int blocks = wg_lnb_blocks(M);
    WG_REQUIRE(workspace_floats >= (long)blocks * 2 * C, "layernorm_bwd_det: workspace too small (need %ld floats)", (long)blocks * 2 * C);
    const int rpb = (M + blocks - 1) / blocks;
#define WG_LNB(V) hipLaunchKernelGGL(wg_layernorm_bwd_small_kernel<V>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, (const bf16*)gamma, \
                                     (const bf16*)dy, lddy, (bf16*)dx, lddx, workspace, M, rpb, eps)
    if (C == 64) WG_LNB(1); else if (C == 128) WG_LNB(2); else if (C == 256) WG_LNB(4); else WG_LNB(8);
#undef WG_LNB
    hipLaunchKernelGGL(wg_ln_partials_kernel, dim3(2 * C), dim3(256), 0, (hipStream_t)stream, workspace, blocks, C, dgamma, dbeta, out_f32);
    return wg_check_launch("wg_layernorm_bwd_det_bf16");
}

extern "C" int wg_layernorm_bwd_bf16(const void* x, long ldx, const void* gamma, const void* dy, long lddy, void* dx, long lddx, float* dgamma,
                                     float* dbeta, float* row_stats, int M, int C, float eps, void* stream) {
    WG_REQUIRE(x && gamma && dy && dx && dgamma && dbeta, "layernorm_bwd: null operand");
    WG_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "layernorm_bwd: C = %d must be a multiple of 8", C);
    WG_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0, "layernorm_bwd: misaligned operand");
    if (C > LNB_CH * 512) {   // wide rows: `row_stats` (2 M floats of workspace) is required
        WG_REQUIRE(row_stats, "layernorm_bwd: rows wider than %d need the row_stats workspace (2 * M floats)", LNB_CH * 512);
        hipLaunchKernelGGL(wg_layernorm_bwd_wide_dx_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, (const bf16*)gamma,
                           (const bf16*)dy, lddy, (bf16*)dx, lddx, row_stats, M, C, eps);
        hipLaunchKernelGGL(wg_layernorm_bwd_wide_cols_kernel, dim3((C + 511) / 512, (M + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx,
                           (const bf16*)dy, lddy, row_stats, dgamma, dbeta, M, C);
        return wg_check_launch("wg_layernorm_bwd_bf16(wide)");
    }
    // ~16 rows per wave: every wave ends with 2 C atomics, so few waves with many rows each (4096 rows of 256: 1 M atomics on 512 addresses
    // took 122 us with one or two rows per wave)
    int blocks = (M + 63) / 64;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(wg_layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, (const bf16*)gamma, (const bf16*)dy,
                       lddy, (bf16*)dx, lddx, dgamma, dbeta, M, C, eps);
    return wg_check_launch("wg_layernorm_bwd_bf16");
}
