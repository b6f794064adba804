// fp32 verification route (NOT a product path, not benched): the CLIP tower -> projector arithmetic with fp32 storage and exact fp32 matrix
// math, to hold north_star's "text logits within 1e-4 abs of the reference CPU path" on the GPU at all -- the bf16 path sits 0.3-0.7 away
// from fp32 because its WEIGHTS are bf16 (tests/test_gpu_modules.py::clip_calibration), which no kernel can undo.
//
// gfx950 has no TF32-like mode: v_mfma_f32_16x16x4_f32 is an exact k-ordered fp32 fmaf chain at the fp32 vector rate (MI355X_MICROARCH.md,
// Matrix cores), so "fp32 on the matrix pipe" is bit-compatible with a plain fp32 dot product.  Kernels are deliberately plain (one wave
// per 16 x 16 output tile, operands straight from global memory; a wave per LayerNorm row; a wave per attention query): they exist to be
// obviously right.
//   wg_f32_gemm_bias_act   custom_clip.py:50-104 call sites of HF CLIPAttention / CLIPMLP linears, the patch embedding as a GEMM over
//                          patch rows, llava_arch.py:36-42 (mm_projector)
//   wg_f32_layernorm       HF CLIP pre_layrnorm / layer_norm1 / layer_norm2 (eps 1e-5)
//   wg_f32_mha             HF CLIPAttention (eager): softmax(q k^T * scale + key bias) v, custom_clip.py:27-38 mask
//   wg_f32_mha_ex          the same with queries and keys / values in different tensors (nn.MultiheadAttention of MSQP's CrossAttnBlock,
//                          utils_walkgpt.py:163-185) and an additive bias per (batch, head, query, key): SAM's decomposed relative position
//                          term, image_encoder.py:321-392, formed by the caller from the unscaled queries
#include "wg_common.h"

// C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ R[m % res_mod or m, :]); all fp32, K % 4 == 0.
__global__ __launch_bounds__(64) void wg_f32_gemm_kernel(const float* A, long lda, const float* W, long ldw, const float* bias, const float* R, long ldr,
                                                         int res_mod, float* C, long ldc, int M, int N, int K, int act) {
    const int lane = threadIdx.x;
    const int tm = blockIdx.y * 16, tn = blockIdx.x * 16;
    const int r16 = lane & 15, kq = lane >> 4;
    // operand maps of mfma_f32_16x16x4f32: lane l holds A[row l & 15][k = l >> 4] and B[k = l >> 4][col l & 15]
    const int am = tm + r16 < M ? tm + r16 : M - 1, wn = tn + r16 < N ? tn + r16 : N - 1;
    const float* ap = A + (long)am * lda + kq;
    const float* wp = W + (long)wn * ldw + kq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k], wp[k], acc, 0, 0, 0);
    // C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int n = tn + r16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = tm + kq * 4 + j;
        if (m < M && n < N) {
            float v = acc[j] + (bias ? bias[n] : 0.f);
            if (act == WG_ACT_GELU_ERF) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            else if (act == WG_ACT_QUICK_GELU) v = v / (1.0f + expf(-1.702f * v));
            else if (act == WG_ACT_RELU) v = fmaxf(v, 0.f);
            if (R) v += R[(long)(res_mod > 0 ? m % res_mod : m) * ldr + n];
            C[(long)m * ldc + n] = v;
        }
    }
}

extern "C" int wg_f32_gemm_bias_act(const float* A, long lda, const float* W, long ldw, const float* bias, const float* residual, long ldr, int res_row_mod,
                                    float* C, long ldc, int M, int N, int K, int act, void* stream) {
    WG_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0 && K % 4 == 0, "f32_gemm: bad arguments (K must be a multiple of 4)");
    hipLaunchKernelGGL(wg_f32_gemm_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(64), 0, (hipStream_t)stream, A, lda, W, ldw, bias, residual, ldr,
                       res_row_mod, C, ldc, M, N, K, act);
    return wg_check_launch("wg_f32_gemm_bias_act");
}

// y = (x - mean) * rsqrt(var + eps) * gamma + beta per row (biased variance, two passes over the row in registers / global memory)
__global__ __launch_bounds__(64) void wg_f32_layernorm_kernel(const float* x, const float* gamma, const float* beta, float* y, long rows, int C, float eps) {
    const long row = blockIdx.x;
    const int lane = threadIdx.x;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wg_wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wg_wave_sum(q) / (float)C + eps);
    for (int c = lane; c < C; c += 64) y[row * C + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}

extern "C" int wg_f32_layernorm(const float* x, const float* gamma, const float* beta, float* y, long rows, int C, float eps, void* stream) {
    WG_REQUIRE(x && gamma && beta && y && rows > 0 && C > 0, "f32_layernorm: bad arguments");
    hipLaunchKernelGGL(wg_f32_layernorm_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, x, gamma, beta, y, rows, C, eps);
    return wg_check_launch("wg_f32_layernorm");
}

// o[b, i, h*hd + d] = sum_j softmax_j(scale * q_i . k_j + key_bias[b, j]) v[j, d]: a wave per (batch, head, query), lanes over keys, then over d.
// q, k, v, o: [B, L, ld] rows with the head at columns h*hd; hd <= 128.
__global__ __launch_bounds__(64) void wg_f32_mha_kernel(const float* q, const float* k, const float* v, float* o, const float* key_bias, long ld, long ldo,
                                                        int heads, int hd, int Lq, int Lk, float scale) {
    extern __shared__ float p[];      // [Lk] probabilities of this query
    const int lane = threadIdx.x;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const float* qi = q + ((long)b * Lq + i) * ld + h * hd;
    float mx = -INFINITY;
    for (int j = lane; j < Lk; j += 64) {
        const float* kj = k + ((long)b * Lk + j) * ld + h * hd;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(qi[d] * scale, kj[d], s);      // (q scaled first: HF CLIPAttention scales the projected query)
        if (key_bias) s += key_bias[(long)b * Lk + j];
        p[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wg_wave_max(mx);
    float l = 0.f;
    for (int j = lane; j < Lk; j += 64) {
        const float e = expf(p[j] - mx);
        p[j] = e;
        l += e;
    }
    l = wg_wave_sum(l);
    __syncthreads();
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < Lk; ++j) acc = fmaf(p[j], v[((long)b * Lk + j) * ld + h * hd + d], acc);
        o[((long)b * Lq + i) * ldo + h * hd + d] = acc / l;
    }
}

extern "C" int wg_f32_mha(const float* q, const float* k, const float* v, float* o, const float* key_bias, long ld, long ldo, int B, int heads, int head_dim,
                          int Lq, int Lk, float scale, void* stream) {
    WG_REQUIRE(q && k && v && o && B > 0 && heads > 0 && head_dim > 0 && Lq > 0 && Lk > 0 && Lk <= 16384, "f32_mha: bad arguments");
    hipLaunchKernelGGL(wg_f32_mha_kernel, dim3(Lq, heads, B), dim3(64), (size_t)Lk * 4, (hipStream_t)stream, q, k, v, o, key_bias, ld, ldo, heads, head_dim,
                       Lq, Lk, scale);
    return wg_check_launch("wg_f32_mha");
}

// o[b, i, h*hd + d] = sum_j softmax_j(scale * q_i . k_j + key_bias[b, j] + attn_bias[b, h, i, j]) v[j, d]; q rows [B, Lq, ldq], k / v rows [B, Lk, ldkv].
__global__ __launch_bounds__(64) void wg_f32_mha_ex_kernel(const float* q, long ldq, const float* k, const float* v, long ldkv, float* o, long ldo,
                                                           const float* key_bias, const float* attn_bias, int heads, int hd, int Lq, int Lk, float scale) {
    extern __shared__ float p[];      // [Lk] probabilities of this query
    const int lane = threadIdx.x;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const float* qi = q + ((long)b * Lq + i) * ldq + h * hd;
    const float* ab = attn_bias ? attn_bias + (((long)b * heads + h) * Lq + i) * Lk : nullptr;
    float mx = -INFINITY;
    for (int j = lane; j < Lk; j += 64) {
        const float* kj = k + ((long)b * Lk + j) * ldkv + h * hd;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(qi[d] * scale, kj[d], s);
        if (key_bias) s += key_bias[(long)b * Lk + j];
        if (ab) s += ab[j];
        p[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wg_wave_max(mx);
    float l = 0.f;
    for (int j = lane; j < Lk; j += 64) {
        const float e = expf(p[j] - mx);
        p[j] = e;
        l += e;
    }
    l = wg_wave_sum(l);
    __syncthreads();
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < Lk; ++j) acc = fmaf(p[j], v[((long)b * Lk + j) * ldkv + h * hd + d], acc);
        o[((long)b * Lq + i) * ldo + h * hd + d] = acc / l;
    }
}

extern "C" int wg_f32_mha_ex(const float* q, long ldq, const float* k, const float* v, long ldkv, float* o, long ldo, const float* key_bias,
                             const float* attn_bias, int B, int heads, int head_dim, int Lq, int Lk, float scale, void* stream) {
    WG_REQUIRE(q && k && v && o && B > 0 && heads > 0 && head_dim > 0 && Lq > 0 && Lk > 0 && Lk <= 16384, "f32_mha_ex: bad arguments");
    // every head's head_dim columns of a row must lie inside the row's pitch (q, the k | v packing behind ldkv, and o): a short pitch from
    // the caller would read or write past the rows without any message
    WG_REQUIRE(ldq >= (long)heads * head_dim && ldkv >= (long)heads * head_dim && ldo >= (long)heads * head_dim,
               "f32_mha_ex: a row pitch (ldq %ld, ldkv %ld, ldo %ld) is smaller than heads * head_dim = %d", ldq, ldkv, ldo, heads * head_dim);
    hipLaunchKernelGGL(wg_f32_mha_ex_kernel, dim3(Lq, heads, B), dim3(64), (size_t)Lk * 4, (hipStream_t)stream, q, ldq, k, v, ldkv, o, ldo, key_bias,
                       attn_bias, heads, head_dim, Lq, Lk, scale);
    return wg_check_launch("wg_f32_mha_ex");
}
