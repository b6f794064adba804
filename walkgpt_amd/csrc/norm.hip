// Row LayerNorm (bf16 in/out, fp32 statistics), one wave per row.
//
// Replaces nn.LayerNorm in the SAM blocks (eps 1e-6, image_encoder.py:177-193 via build_sam.py:73), the CLIP
// layers (eps 1e-5), MSQP/CTP (utils_walkgpt.py:163-185,302-327), the two-way transformer
// (transformer.py:151-182) and -- because the build keeps activations channels-last -- LayerNorm2d
// (common.py:31-43: biased variance over C, eps inside the sqrt), including the neck and the mask-decoder upscaler.
//
// HBM-bound: 2 bytes read + 2 bytes written per element.  The row stays in registers between the
// statistics and the normalisation, loads/stores are 16 bytes per lane.
#include "wg_common.h"

struct LnArgs {
    const bf16* x; long ldx;
    const bf16* gamma; const bf16* beta;
    bf16* y; long ldy;
    int M, D;
    float eps;
    int act;  // optional activation fused behind the affine (upscaler: LN2d -> GELU)
};

template <int MAXC>
__global__ __launch_bounds__(256) void wg_layernorm_kernel(LnArgs a) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const bf16* x = a.x + (long)m * a.ldx;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int d = c * 512 + lane * 8;
        if (d < a.D) {
            const bf16x8 t = *(const bf16x8*)(x + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[c][e] = (float)t[e]; s += v[c][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
        }
    }
    const float mean = wg_wave_sum(s) / (float)a.D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int d = c * 512 + lane * 8;
        if (d < a.D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float t = v[c][e] - mean; q += t * t; }
        }
    }
    const float rstd = 1.0f / sqrtf(wg_wave_sum(q) / (float)a.D + a.eps);
    bf16* y = a.y + (long)m * a.ldy;
    // (the activation is resolved once per wave, not per element: see WG_ACT_SWITCH)
    WG_ACT_SWITCH(a.act,
        _Pragma("unroll") for (int c = 0; c < MAXC; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < a.D) {
                const bf16x8 gm = *(const bf16x8*)(a.gamma + d);
                const bf16x8 bt = *(const bf16x8*)(a.beta + d);
                bf16x8 o;
                _Pragma("unroll") for (int e = 0; e < 8; e += 2) {
                    f32x2 t = {(v[c][e] - mean) * rstd * (float)gm[e] + (float)bt[e], (v[c][e + 1] - mean) * rstd * (float)gm[e + 1] + (float)bt[e + 1]};
                    t = wg_act2<ACT>(t);
                    o[e] = (bf16)t.x; o[e + 1] = (bf16)t.y;
                }
                *(bf16x8*)(y + d) = o;
            }
        })
}

// D <= 128 (the mask decoder's LayerNorm2d over 64 channels runs on P*16384 rows): D/8 lanes per row, 512/D rows per
// wave, so every lane moves 16 bytes; statistics by xor-shuffles inside the lane group.
template <int D>
__global__ __launch_bounds__(256) void wg_layernorm_small_kernel(LnArgs a) {
    constexpr int LPR = D / 8;          // lanes per row
    constexpr int RPW = 64 / LPR;       // rows per wave
    const int lane = threadIdx.x & 63;
    const long m = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int d = (lane % LPR) * 8;
    const bool on = m < a.M;
    float v[8];
    float s = 0.f;
    if (on) {
        const bf16x8 t = *(const bf16x8*)(a.x + m * a.ldx + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = (float)t[e]; s += v[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float t = v[e] - mean; q += t * t; }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)D + a.eps);
    if (on) {
        const bf16x8 gm = *(const bf16x8*)(a.gamma + d);
        const bf16x8 bt = *(const bf16x8*)(a.beta + d);
        bf16x8 o;
        WG_ACT_SWITCH(a.act,
            _Pragma("unroll") for (int e = 0; e < 8; e += 2) {
                f32x2 t = {(v[e] - mean) * rstd * (float)gm[e] + (float)bt[e], (v[e + 1] - mean) * rstd * (float)gm[e + 1] + (float)bt[e + 1]};
                t = wg_act2<ACT>(t);
                o[e] = (bf16)t.x; o[e + 1] = (bf16)t.y;
            })
        *(bf16x8*)(a.y + m * a.ldy + d) = o;
    }
}

extern "C" int wg_layernorm_rows(const void* x, long ldx, const void* gamma, const void* beta, void* y, long ldy,
                                 int M, int D, float eps, int act, void* stream) {
    WG_REQUIRE(x && gamma && beta && y, "layernorm: null operand");
    WG_REQUIRE(M > 0 && D > 0 && D % 8 == 0, "layernorm: D=%d must be a positive multiple of 8", D);
    WG_REQUIRE(D <= 8192, "layernorm: D=%d exceeds 8192", D);
    WG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= D && ldy >= D, "layernorm: bad leading dimension");
    LnArgs a{(const bf16*)x, ldx, (const bf16*)gamma, (const bf16*)beta, (bf16*)y, ldy, M, D, eps, act};
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((M + 3) / 4), block(256);
    if (D == 32 || D == 64 || D == 128) {
        const int rpw = 512 / D;
        dim3 g2((unsigned)((M + 4 * rpw - 1) / (4 * rpw)));
        if (D == 32) hipLaunchKernelGGL(wg_layernorm_small_kernel<32>, g2, block, 0, st, a);
        else if (D == 64) hipLaunchKernelGGL(wg_layernorm_small_kernel<64>, g2, block, 0, st, a);
        else hipLaunchKernelGGL(wg_layernorm_small_kernel<128>, g2, block, 0, st, a);
        return wg_check_launch("wg_layernorm_rows(small)");
    }
    if (D <= 1024) hipLaunchKernelGGL(wg_layernorm_kernel<2>, grid, block, 0, st, a);
    else if (D <= 2048) hipLaunchKernelGGL(wg_layernorm_kernel<4>, grid, block, 0, st, a);
    else if (D <= 4096) hipLaunchKernelGGL(wg_layernorm_kernel<8>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(wg_layernorm_kernel<16>, grid, block, 0, st, a);
    return wg_check_launch("wg_layernorm_rows");
}

// Row statistics only (mean, 1/sqrt(var + eps)) -> stats[m] = {mean, rstd} fp32: the LayerNorm whose affine map is folded into
// the consuming GEMM (wg_gemm_ln_bias_act_bf16) needs nothing else -- half the HBM traffic of a full LayerNorm pass.
template <int MAXC, int R>   // R rows per wave, their loads all in flight before the first reduction (the kernel is latency-bound)
__global__ __launch_bounds__(256) void wg_row_stats_kernel(const bf16* x, long ldx, float* stats, int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int m0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (m0 >= M) return;
    float v[R][MAXC][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = m0 + r < M ? m0 + r : M - 1;
        const bf16* p = x + (long)m * ldx;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int d = c * 512 + lane * 8;
            bf16x8 t = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            if (d < D) t = *(const bf16x8*)(p + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[r][c][e] = (float)t[e];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[r][c][e];
        const float mean = wg_wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c * 512 + lane * 8 < D) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float t = v[r][c][e] - mean; q += t * t; }
            }
        }
        const float rstd = 1.0f / sqrtf(wg_wave_sum(q) / (float)D + eps);
        if (lane == 0 && m0 + r < M) { stats[2 * (long)(m0 + r)] = mean; stats[2 * (long)(m0 + r) + 1] = rstd; }
    }
}

extern "C" int wg_row_stats_bf16(const void* x, long ldx, float* stats, int M, int D, float eps, void* stream) {
    WG_REQUIRE(x && stats && M > 0 && D > 0 && D % 8 == 0 && D <= 8192 && ldx % 8 == 0 && ldx >= D, "row_stats: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256);
    const int R = (D <= 2048 && M >= 4096) ? 2 : 1;
    dim3 grid((M + 4 * R - 1) / (4 * R));
    if (D <= 1024 && R == 2) hipLaunchKernelGGL((wg_row_stats_kernel<2, 2>), grid, block, 0, st, (const bf16*)x, ldx, stats, M, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL((wg_row_stats_kernel<2, 1>), grid, block, 0, st, (const bf16*)x, ldx, stats, M, D, eps);
    else if (D <= 2048 && R == 2) hipLaunchKernelGGL((wg_row_stats_kernel<4, 2>), grid, block, 0, st, (const bf16*)x, ldx, stats, M, D, eps);
    else if (D <= 2048) hipLaunchKernelGGL((wg_row_stats_kernel<4, 1>), grid, block, 0, st, (const bf16*)x, ldx, stats, M, D, eps);
    else if (D <= 4096) hipLaunchKernelGGL((wg_row_stats_kernel<8, 1>), grid, block, 0, st, (const bf16*)x, ldx, stats, M, D, eps);
    else hipLaunchKernelGGL((wg_row_stats_kernel<16, 1>), grid, block, 0, st, (const bf16*)x, ldx, stats, M, D, eps);
    return wg_check_launch("wg_row_stats_bf16");
}
