// Region-alignment (InfoNCE) loss forward -- SURVEY.md §8f row 1.
//
// Replaces infonce_loss() + TinyCrossAttn.forward() of the reference (/root/reference/utils/utils_walkgpt.py:8-73,330-357,
// called at model/walkgpt.py:459-473) for the configuration WalkGPT trains with (normalize=True, top_k optional).
//
// Work split (host side: walkgpt_amd/utils_walkgpt.py):
//   * the two [M,256]x[256,256] query projections run on wg_gemm; q.(Wk kv_n) is folded to (Wk^T q).kv_n, so the 4096
//     tokens of a row are never projected (the reference projects K and V of every gathered row: 2 x 2.1 GFLOP per [SEG]);
//   * ST = [Zn ; Wk^T q] . tokens^T  -> fp32 [2M, rows*N] by wg_gemm (tokens = the SAM embedding rows, read once);
//   * wg_nce_attn_*:  per [SEG] m, softmax over its own row's N tokens -> attn_w, then either the top-k refinement
//     (alpha-weighted sum of the k raw tokens) or the attention-pooled raw token (the host applies Wv / Wo to it:
//     sum_n a_n (Wv kv_n + b) = Wv (sum_n a_n kv_n) + b because the weights sum to one);
//   * wg_nce_loss_*:  per m, cosine of Zn with the positive, masked log-sum-exp over all rows*N cosines / temperature.
// HBM-bound, tiny next to the encoders: ST is 2M*rows*N*4 bytes (29 MB at M = 112, rows = 8, N = 4096), read twice.
#include "wg_common.h"

__device__ __forceinline__ float wg_block_sum(float v, float* red) {   // 256 threads
    v = wg_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float wg_block_max(float v, float* red) {
    v = wg_wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// 1 / max(||x_r||, eps) for every row (F.normalize's denominator, utils_walkgpt.py:46-54); one wave per row.
__global__ __launch_bounds__(256) void wg_row_inv_norm_kernel(const bf16* x, long ld, float* out, long R, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const bf16* p = x + r * ld;
    float s = 0.f;
    for (int d = lane * 8; d < D; d += 512) {
        const bf16x8 v = *(const bf16x8*)(p + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[e] * (float)v[e];
    }
    s = wg_wave_sum(s);
    if (lane == 0) out[r] = 1.0f / fmaxf(sqrtf(s), eps);
}

// rows of Z scaled to unit length, written as bf16 (the GEMM operand; the same rounded values feed the positive term)
__global__ __launch_bounds__(256) void wg_l2_normalize_rows_kernel(const bf16* x, long ldx, bf16* y, long ldy, long R, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = (float)x[r * ldx + d]; s += v * v; }
    s = wg_wave_sum(s);
    const float inv = 1.0f / fmaxf(sqrtf(s), eps);
    for (int d = lane; d < D; d += 64) y[r * ldy + d] = (bf16)((float)x[r * ldx + d] * inv);
}

struct NceArgs {
    const float* ST;        // [2M, ldst]: row m = Zn_m . token_t, row M+m = (Wk^T q_m) . token_t
    long ldst;
    const float* inv_norm;  // [rows*N]
    const bf16* tokens;     // [rows*N, D]
    long ldt;
    const bf16* Zn;         // [M, D] unit rows
    const int* seg_row;     // [M]
    float* attn_w;          // [M, N]
    float* vraw;            // [M, D]: top-k refined positive, or the attention-pooled raw token
    const float* vpos;      // [M, D]: positive features entering the loss
    float* loss_m;          // [M]
    float* logits;          // optional [M, 1 + rows*N]
    int M, N, rows, D, top_k, exclude_same_row;
    float attn_scale, inv_temp;
};

// One workgroup per [SEG] token: softmax over its row's tokens, then top-k refinement or attention pooling.
__global__ __launch_bounds__(256) void wg_nce_attn_kernel(NceArgs a) {
    extern __shared__ float sm[];      // [N] attention weights, then 8 reduction slots
    float* red = sm + a.N;
    __shared__ int pick_idx[32];
    __shared__ float pick_val[32];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int row = a.seg_row[m];
    const float* lg = a.ST + (long)(a.M + m) * a.ldst + (long)row * a.N;
    float mx = -3.0e38f;
    for (int n = tid; n < a.N; n += 256) { const float v = lg[n] * a.attn_scale; sm[n] = v; mx = fmaxf(mx, v); }
    mx = wg_block_max(mx, red);
    float s = 0.f;
    for (int n = tid; n < a.N; n += 256) { const float e = __expf(sm[n] - mx); sm[n] = e; s += e; }
    s = wg_block_sum(s, red);
    const float inv = 1.0f / s;
    for (int n = tid; n < a.N; n += 256) { const float w = sm[n] * inv; sm[n] = w; a.attn_w[(long)m * a.N + n] = w; }
    __syncthreads();
    const bf16* tok = a.tokens + (long)row * a.N * a.ldt;
    if (a.top_k > 0 && a.top_k < a.N) {
        // torch.topk(attn_w, k): k rounds of block arg-max (lowest index wins ties), each winner masked out
        for (int k = 0; k < a.top_k; ++k) {
            float bv = -1.f;
            int bi = 0x7fffffff;
            for (int n = tid; n < a.N; n += 256) { const float v = sm[n]; if (v > bv) { bv = v; bi = n; } }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            __syncthreads();
            if ((tid & 63) == 0) { red[tid >> 6] = bv; red[4 + (tid >> 6)] = __int_as_float(bi); }
            __syncthreads();
            if (tid == 0) {
                float v = red[0]; int i = __float_as_int(red[4]);
                for (int w = 1; w < 4; ++w) {
                    const float ov = red[w]; const int oi = __float_as_int(red[4 + w]);
                    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
                }
                pick_idx[k] = i; pick_val[k] = v;
                sm[i] = -2.f;
            }
            __syncthreads();
        }
        float tot = 0.f;
        for (int k = 0; k < a.top_k; ++k) tot += pick_val[k];
        const float invt = 1.0f / (tot + 1e-12f);
        for (int d = tid; d < a.D; d += 256) {
            float acc = 0.f;
            for (int k = 0; k < a.top_k; ++k) acc += pick_val[k] * invt * (float)tok[(long)pick_idx[k] * a.ldt + d];
            a.vraw[(long)m * a.D + d] = acc;
        }
    } else {
        for (int d = tid; d < a.D; d += 256) {
            float acc = 0.f;
            for (int n = 0; n < a.N; ++n) acc += sm[n] * (float)tok[(long)n * a.ldt + d];
            a.vraw[(long)m * a.D + d] = acc;
        }
    }
}

// One workgroup per [SEG] token: positive cosine + masked log-sum-exp over every token of every row.
__global__ __launch_bounds__(256) void wg_nce_loss_kernel(NceArgs a) {
    __shared__ float red[8];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int row = a.seg_row[m];
    float vv = 0.f, zv = 0.f;
    for (int d = tid; d < a.D; d += 256) {
        const float v = a.vpos[(long)m * a.D + d];
        vv += v * v;
        zv += v * (float)a.Zn[(long)m * a.D + d];
    }
    vv = wg_block_sum(vv, red);
    zv = wg_block_sum(zv, red);
    const float pos = zv / fmaxf(sqrtf(vv), 1e-12f) * a.inv_temp;
    const long T = (long)a.rows * a.N;
    const float* st = a.ST + (long)m * a.ldst;
    float* lo = a.logits ? a.logits + (long)m * (T + 1) : nullptr;
    if (lo && tid == 0) lo[0] = pos;
    const long own0 = (long)row * a.N, own1 = own0 + a.N;
    // two passes (max, then sum of exponentials): the row is 4*T bytes and stays in L2
    float mx = pos;
    for (long t = tid; t < T; t += 256) {
        const bool masked = a.exclude_same_row && t >= own0 && t < own1;
        const float v = masked ? -INFINITY : st[t] * a.inv_norm[t] * a.inv_temp;
        if (lo) lo[1 + t] = v;
        mx = fmaxf(mx, v);
    }
    mx = wg_block_max(mx, red);
    float s = 0.f;
    for (long t = tid; t < T; t += 256) {
        const bool masked = a.exclude_same_row && t >= own0 && t < own1;
        if (!masked) s += __expf(st[t] * a.inv_norm[t] * a.inv_temp - mx);
    }
    s = wg_block_sum(s, red);
    if (tid == 0) a.loss_m[m] = logf(s + __expf(pos - mx)) + mx - pos;   // cross entropy with the positive at index 0
}

__global__ __launch_bounds__(64) void wg_mean_f32_kernel(const float* x, int n, float* out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += x[i];
    s = wg_wave_sum(s);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

extern "C" int wg_row_inv_norm_bf16(const void* x, long ld, float* out, long R, int D, float eps, void* stream) {
    WG_REQUIRE(x && out && R > 0 && D > 0 && D % 8 == 0 && ld % 8 == 0 && ((uintptr_t)x & 15) == 0, "row_inv_norm: bad arguments");
    hipLaunchKernelGGL(wg_row_inv_norm_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ld, out, R, D, eps);
    return wg_check_launch("wg_row_inv_norm_bf16");
}

extern "C" int wg_l2_normalize_rows_bf16(const void* x, long ldx, void* y, long ldy, long R, int D, float eps, void* stream) {
    WG_REQUIRE(x && y && R > 0 && D > 0, "l2_normalize_rows: bad arguments");
    hipLaunchKernelGGL(wg_l2_normalize_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, (bf16*)y, ldy, R, D, eps);
    return wg_check_launch("wg_l2_normalize_rows_bf16");
}

// attention weights of every [SEG] over its own row + the refined / pooled raw positive (see the header of this file)
extern "C" int wg_nce_attn_f32(const float* ST, long ldst, const void* tokens, long ldt, const int* seg_row, float* attn_w,
                               float* vraw, int M, int N, int rows, int D, int top_k, float attn_scale, void* stream) {
    WG_REQUIRE(ST && tokens && seg_row && attn_w && vraw, "nce_attn: null operand");
    WG_REQUIRE(M > 0 && N > 0 && rows > 0 && D > 0 && top_k <= 32, "nce_attn: bad shape (top_k <= 32)");
    WG_REQUIRE((size_t)(N + 8) * 4 <= 150 * 1024, "nce_attn: N=%d tokens per row exceed the LDS budget", N);
    NceArgs a{};
    a.ST = ST; a.ldst = ldst; a.tokens = (const bf16*)tokens; a.ldt = ldt; a.seg_row = seg_row; a.attn_w = attn_w; a.vraw = vraw;
    a.M = M; a.N = N; a.rows = rows; a.D = D; a.top_k = top_k; a.attn_scale = attn_scale;
    const size_t lds = (size_t)(N + 8) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)wg_nce_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(wg_nce_attn_kernel, dim3(M), dim3(256), lds, (hipStream_t)stream, a);
    return wg_check_launch("wg_nce_attn_f32");
}

// per-[SEG] cross entropy (loss_m), their mean (loss), optionally the full logit rows [M, 1 + rows*N]
extern "C" int wg_nce_loss_f32(const float* ST, long ldst, const float* inv_norm, const void* Zn, const float* vpos,
                               const int* seg_row, float* loss_m, float* loss, float* logits, int M, int N, int rows, int D,
                               int exclude_same_row, float temperature, void* stream) {
    WG_REQUIRE(ST && inv_norm && Zn && vpos && seg_row && loss_m && loss, "nce_loss: null operand");
    WG_REQUIRE(M > 0 && N > 0 && rows > 0 && D > 0 && temperature > 0.f, "nce_loss: bad arguments");
    NceArgs a{};
    a.ST = ST; a.ldst = ldst; a.inv_norm = inv_norm; a.Zn = (const bf16*)Zn; a.vpos = vpos; a.seg_row = seg_row;
    a.loss_m = loss_m; a.logits = logits; a.M = M; a.N = N; a.rows = rows; a.D = D; a.exclude_same_row = exclude_same_row;
    a.inv_temp = 1.0f / temperature;
    hipLaunchKernelGGL(wg_nce_loss_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(wg_mean_f32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)loss_m, M, loss);
    return wg_check_launch("wg_nce_loss_f32");
}
