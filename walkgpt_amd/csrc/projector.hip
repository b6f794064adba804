// Wavefront-level kernels of the Multi-Scale Query Projector and the Calibrated Text Projector
// (/root/reference/utils/utils_walkgpt.py:195-217, 259-327) and the LLM-side token resample
// (/root/reference/model/llava_walkgpt/model/llava_arch.py:252-259).  All HBM-bound; fp32 arithmetic, bf16 storage.
#include "wg_common.h"

// ---- _pool_grid_tokens (:195-201): avg_pool2d(kernel = stride = s) on channels-last tokens ---------------------------
// x [B, H, W, C] -> y [B, H/s, W/s, C].  One thread per (output token, 8 channels): s*s 16-byte loads, one 16-byte store.
__global__ __launch_bounds__(256) void wg_avgpool_kernel(const bf16* x, bf16* y, int B, int H, int W, int C, int s) {
    const int Ho = H / s, Wo = W / s, cpr = C / 8;
    const long total = (long)B * Ho * Wo * cpr;
    const float inv = 1.0f / (float)(s * s);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr) * 8;
        const long t = i / cpr;
        const int ox = (int)(t % Wo), oy = (int)((t / Wo) % Ho), b = (int)(t / ((long)Wo * Ho));
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int dy = 0; dy < s; ++dy)
            for (int dx = 0; dx < s; ++dx) {
                const bf16x8 v = *(const bf16x8*)(x + (((long)b * H + oy * s + dy) * W + ox * s + dx) * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(acc[e] * inv);
        *(bf16x8*)(y + t * C + c) = o;
    }
}

extern "C" int wg_avgpool_tokens_bf16(const void* x, void* y, int B, int H, int W, int C, int s, void* stream) {
    WG_REQUIRE(x && y && B > 0 && s > 0 && H % s == 0 && W % s == 0 && C % 8 == 0, "avgpool_tokens: bad arguments");
    const long total = (long)B * (H / s) * (W / s) * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wg_avgpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, B, H, W, C, s);
    return wg_check_launch("wg_avgpool_tokens_bf16");
}

// ---- _global_token (:256-257): mean over all tokens.  x [B, L, C] -> y [B, C].  Block = (batch, 64 channels):
// 8 channel-chunk lanes x 32 row lanes, LDS reduction over the row lanes. ---------------------------------------------
__global__ __launch_bounds__(256) void wg_mean_tokens_kernel(const bf16* x, bf16* y, int L, int C) {
    __shared__ float red[32][64];
    const int b = blockIdx.y;
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 64 + cl * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < C) {
        for (int r = rl; r < L; r += 32) {
            const bf16x8 v = *(const bf16x8*)(x + ((long)b * L + r) * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cl * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int r = 0; r < 32; ++r) s += red[r][threadIdx.x];
        const int cc = blockIdx.x * 64 + threadIdx.x;
        if (cc < C) y[(long)b * C + cc] = (bf16)(s / (float)L);
    }
}

extern "C" int wg_mean_tokens_bf16(const void* x, void* y, int B, int L, int C, void* stream) {
    WG_REQUIRE(x && y && B > 0 && L > 0 && C % 8 == 0, "mean_tokens: bad arguments");
    hipLaunchKernelGGL(wg_mean_tokens_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, L, C);
    return wg_check_launch("wg_mean_tokens_bf16");
}

// ---- SegAwareGate tail (:213-217): y[r, :] = x[r, :] * sigmoid(logit[r]) ------------------------------------------------
__global__ __launch_bounds__(256) void wg_gate_kernel(const bf16* x, const float* logit, bf16* y, long rows, int C) {
    const int cpr = C / 8;
    const long total = rows * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cpr;
        const int c = (int)(i % cpr) * 8;
        const float g = 1.0f / (1.0f + __expf(-logit[r]));
        const bf16x8 v = *(const bf16x8*)(x + r * C + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)v[e] * g);
        *(bf16x8*)(y + r * C + c) = o;
    }
}

extern "C" int wg_sigmoid_gate_bf16(const void* x, const float* logit, void* y, long rows, int C, void* stream) {
    WG_REQUIRE(x && logit && y && rows > 0 && C % 8 == 0, "sigmoid_gate: bad arguments");
    const long total = rows * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wg_gate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, logit, (bf16*)y, rows, C);
    return wg_check_launch("wg_sigmoid_gate_bf16");
}

// ---- CTP tail (:321-327): LayerNorm(C) -> + text_type -> L2 normalise (eps 1e-12) -> * exp(log_temp). -----------------
// One wave per row, the row stays in registers; C <= 512.
__global__ __launch_bounds__(256) void wg_ctp_tail_kernel(const bf16* x, long ldx, const bf16* gamma, const bf16* beta,
                                                          const bf16* text_type, const bf16* log_temp, bf16* y, long ldy, int M,
                                                          int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int d = lane * 8;
    float v[8];
    wg_ctp_tail_row(x + (long)m * ldx, gamma, beta, text_type, log_temp, C, eps, lane, v);
    if (d < C) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        *(bf16x8*)(y + (long)m * ldy + d) = o;
    }
}

extern "C" int wg_ctp_tail_bf16(const void* x, long ldx, const void* gamma, const void* beta, const void* text_type,
                                const void* log_temp, void* y, long ldy, int M, int C, float eps, void* stream) {
    WG_REQUIRE(x && gamma && beta && text_type && log_temp && y, "ctp_tail: null operand");
    WG_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 512, "ctp_tail: C=%d must be a multiple of 8, at most 512", C);
    hipLaunchKernelGGL(wg_ctp_tail_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       (const bf16*)gamma, (const bf16*)beta, (const bf16*)text_type, (const bf16*)log_temp, (bf16*)y, ldy, M, C, eps);
    return wg_check_launch("wg_ctp_tail_bf16");
}

// ---- token resample (llava_arch.py:252-259): [n, p*p, C] -> fp32 bilinear (align_corners False) -> [n, t*t, C] -------
__global__ __launch_bounds__(256) void wg_resample_kernel(const bf16* x, bf16* y, int n, int p, int t, int C) {
    const int cpr = C / 8;
    const long total = (long)n * t * t * cpr;
    const float sc = (float)p / (float)t;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
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
        const bf16* base = x + (long)b * p * p * C + c;
        const bf16x8 a = *(const bf16x8*)(base + (long)(y0 * p + x0) * C), bb = *(const bf16x8*)(base + (long)(y0 * p + x1) * C);
        const bf16x8 cc = *(const bf16x8*)(base + (long)(y1 * p + x0) * C), dd = *(const bf16x8*)(base + (long)(y1 * p + x1) * C);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = (bf16)((1.f - ly) * ((1.f - lx) * (float)a[e] + lx * (float)bb[e]) + ly * ((1.f - lx) * (float)cc[e] + lx * (float)dd[e]));
        *(bf16x8*)(y + cell * C + c) = o;
    }
}

extern "C" int wg_resample_tokens_bf16(const void* x, void* y, int n, int p, int t, int C, void* stream) {
    WG_REQUIRE(x && y && n > 0 && p > 0 && t > 0 && C % 8 == 0, "resample_tokens: bad arguments");
    const long total = (long)n * t * t * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wg_resample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, n, p, t, C);
    return wg_check_launch("wg_resample_tokens_bf16");
}
