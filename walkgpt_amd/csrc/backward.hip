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
//   wg_mask_losses_bwd_f32   d(sigmoid_ce_loss + dice_loss)/d logits (utils_walkgpt.py:76-120)
// All HBM-bound row or element kernels: fp32 arithmetic, bf16 activations, fp32 accumulation of parameter gradients (atomics).
#include "wg_common.h"

namespace {

// ---- column sums ----------------------------------------------------------------------------------------------------------------------
// x [R, C] bf16 -> out[C] += sum over rows (fp32; the caller zeroes out).  A workgroup takes 64 rows x 512 columns: lane = 8 columns.
__global__ __launch_bounds__(256) void wg_colsum_kernel(const bf16* x, long ldx, float* out, int R, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 512 + lane * 8;
    if (c >= C) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int r0 = blockIdx.y * 64;
    const int r1 = r0 + 64 < R ? r0 + 64 : R;
    for (int r = r0 + wave; r < r1; r += 4) {
        const bf16x8 t = *(const bf16x8*)(x + (long)r * ldx + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)t[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(out + c + e, s[e]);
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

}  // namespace

extern "C" int wg_colsum_f32(const void* x, long ldx, float* out, int R, int C, void* stream) {
    WG_REQUIRE(x && out && R > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, "colsum: x [R, C] bf16 with C %% 8 == 0, 16-byte rows");
    hipLaunchKernelGGL(wg_colsum_kernel, dim3((C + 511) / 512, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, out, R, C);
    return wg_check_launch("wg_colsum_f32");
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

extern "C" int wg_layernorm_bwd_bf16(const void* x, long ldx, const void* gamma, const void* dy, long lddy, void* dx, long lddx, float* dgamma,
                                     float* dbeta, int M, int C, float eps, void* stream) {
    WG_REQUIRE(x && gamma && dy && dx && dgamma && dbeta, "layernorm_bwd: null operand");
    WG_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= LNB_CH * 512 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0,
               "layernorm_bwd: C = %d must be a multiple of 8, at most %d", C, LNB_CH * 512);
    WG_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0, "layernorm_bwd: misaligned operand");
    int blocks = (M + 3) / 4;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(wg_layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, (const bf16*)gamma, (const bf16*)dy,
                       lddy, (bf16*)dx, lddx, dgamma, dbeta, M, C, eps);
    return wg_check_launch("wg_layernorm_bwd_bf16");
}
