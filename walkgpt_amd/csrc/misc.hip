// HBM-bound data-layout and elementwise kernels of the hot path (no GEMM shape -> no MFMA): patch gathering,
// 3x3 im2row, broadcast adds, layout transposes, the dense positional encoding, the hyper-network mask product,
// the fused double-bilinear mask post-processing and the mask score.
#include "wg_common.h"

// ------------------------------------------------------------------------------------------------------------
// Patch gather ("im2row" of a stride == kernel conv): images NCHW bf16 -> rows [B*gh*gw, Kpad], column order
// (c, ky, kx) = the flattened Conv2d weight [D, C, P, P] (PatchEmbed image_encoder.py:422-426; CLIP patch conv).
// Columns >= C*P*P (CLIP: 588 -> 640) are zero so the GEMM can run 64-deep K slabs.
// One thread produces 8 consecutive columns (16-byte store); for P % 8 == 0 they come from one 16-byte load.
// ------------------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void wg_patchify_kernel(const bf16* img, bf16* rows, int B, int C, int H, int W, int P,
                                                          int Kpad) {
    const int gh = H / P, gw = W / P;
    const int cpr = Kpad / 8;
    const long total = (long)B * gh * gw * cpr;
    const int K = C * P * P;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        const long r = i / cpr;
        const int px = (int)(r % gw), py = (int)((r / gw) % gh), b = (int)(r / ((long)gw * gh));
        bf16x8 v;
        const int k0 = ch * 8;
        if (VEC) {
            if (k0 < K) {
                const int c = k0 / (P * P), ky = (k0 / P) % P, kx = k0 % P;
                v = *(const bf16x8*)(img + (((long)b * C + c) * H + py * P + ky) * W + px * P + kx);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                if (k < K) {
                    const int c = k / (P * P), ky = (k / P) % P, kx = k % P;
                    v[e] = img[(((long)b * C + c) * H + py * P + ky) * W + px * P + kx];
                } else {
                    v[e] = (bf16)0.f;
                }
            }
        }
        *(bf16x8*)(rows + r * Kpad + k0) = v;
    }
}

extern "C" int wg_patchify_bf16(const void* images, void* rows, int B, int C, int H, int W, int P, int Kpad, void* stream) {
    WG_REQUIRE(images && rows, "patchify: null operand");
    WG_REQUIRE(B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0, "patchify: image %dx%d not a multiple of patch %d", H, W, P);
    WG_REQUIRE(Kpad % 8 == 0 && Kpad >= C * P * P, "patchify: Kpad=%d too small or not a multiple of 8", Kpad);
    const long total = (long)B * (H / P) * (W / P) * (Kpad / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    const bool vec = (P % 8 == 0) && (W % 8 == 0) && (((uintptr_t)images & 15) == 0);
    if (vec) hipLaunchKernelGGL(wg_patchify_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)images, (bf16*)rows, B, C, H, W, P, Kpad);
    else hipLaunchKernelGGL(wg_patchify_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)images, (bf16*)rows, B, C, H, W, P, Kpad);
    return wg_check_launch("wg_patchify_bf16");
}

// ------------------------------------------------------------------------------------------------------------
// 3x3 / pad 1 im2row on channels-last tokens: x [B, H, W, C] -> rows [B*H*W, 9*C], column order (ky, kx, c);
// the neck's second conv (image_encoder.py:99-106) then is one GEMM against the weight re-laid as [O, (ky,kx,c)].
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wg_im2row3x3_kernel(const bf16* x, bf16* rows, int B, int H, int W, int C) {
    const int cpr = 9 * C / 8;
    const int cpc = C / 8;
    const long total = (long)B * H * W * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        const long r = i / cpr;
        const int xx = (int)(r % W), yy = (int)((r / W) % H), b = (int)(r / ((long)W * H));
        const int tap = ch / cpc, c0 = (ch % cpc) * 8;
        const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
        bf16x8 v;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
            v = *(const bf16x8*)(x + (((long)b * H + sy) * W + sx) * C + c0);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
        }
        *(bf16x8*)(rows + r * 9 * C + ch * 8) = v;
    }
}

extern "C" int wg_im2row3x3_bf16(const void* x, void* rows, int B, int H, int W, int C, void* stream) {
    WG_REQUIRE(x && rows && B > 0 && H > 0 && W > 0 && C % 8 == 0, "im2row3x3: bad arguments");
    const long total = (long)B * H * W * (9 * C / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wg_im2row3x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)rows, B, H, W, C);
    return wg_check_launch("wg_im2row3x3_bf16");
}

// ------------------------------------------------------------------------------------------------------------
// out[r, :] = a[r, :] + b[r % b_rows, :]    (queries + query_pe, keys + key_pe, src + no_mask_embed, cls + pos ...)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wg_add_rows_kernel(const bf16* a, long lda, const bf16* b, long ldb, int b_rows,
                                                          bf16* out, long ldo, long rows, int cols) {
    const int cpr = cols / 8;
    const long total = rows * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cpr;
        const int c = (int)(i % cpr) * 8;
        const bf16x8 x = *(const bf16x8*)(a + r * lda + c);
        const bf16x8 y = *(const bf16x8*)(b + (r % b_rows) * ldb + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)x[e] + (float)y[e]);
        *(bf16x8*)(out + r * ldo + c) = o;
    }
}

extern "C" int wg_add_rows_bf16(const void* a, long lda, const void* b, long ldb, int b_rows, void* out, long ldo, long rows,
                                int cols, void* stream) {
    WG_REQUIRE(a && b && out && rows > 0 && cols > 0 && cols % 8 == 0 && b_rows > 0, "add_rows: bad arguments");
    WG_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0, "add_rows: leading dimensions must be multiples of 8");
    const long total = rows * (cols / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wg_add_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, lda, (const bf16*)b, ldb,
                       b_rows, (bf16*)out, ldo, rows, cols);
    return wg_check_launch("wg_add_rows_bf16");
}

// ------------------------------------------------------------------------------------------------------------
// Channels-last tokens [B, HW, C] -> NCHW [B, C, HW] (the [B,256,64,64] embedding the reference API returns).
// 32x32 tiles through LDS so both sides are coalesced.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wg_tokens_to_nchw_kernel(const bf16* x, bf16* y, int HW, int C) {
    __shared__ bf16 tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int t = t0 + k, c = c0 + tx;
        tile[k][tx] = (t < HW && c < C) ? x[((long)b * HW + t) * C + c] : (bf16)0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, t = t0 + tx;
        if (t < HW && c < C) y[((long)b * C + c) * HW + t] = tile[tx][k];
    }
}

extern "C" int wg_tokens_to_nchw_bf16(const void* x, void* y, int B, int HW, int C, void* stream) {
    WG_REQUIRE(x && y && B > 0 && HW > 0 && C > 0, "tokens_to_nchw: bad arguments");
    hipLaunchKernelGGL(wg_tokens_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, (bf16*)y, HW, C);
    return wg_check_launch("wg_tokens_to_nchw_bf16");
}

__global__ __launch_bounds__(256) void wg_nchw_to_tokens_kernel(const bf16* x, bf16* y, int HW, int C) {
    __shared__ bf16 tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, t = t0 + tx;
        tile[k][tx] = (t < HW && c < C) ? x[((long)b * C + c) * HW + t] : (bf16)0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int t = t0 + k, c = c0 + tx;
        if (t < HW && c < C) y[((long)b * HW + t) * C + c] = tile[tx][k];
    }
}

extern "C" int wg_nchw_to_tokens_bf16(const void* x, void* y, int B, int HW, int C, void* stream) {
    WG_REQUIRE(x && y && B > 0 && HW > 0 && C > 0, "nchw_to_tokens: bad arguments");
    hipLaunchKernelGGL(wg_nchw_to_tokens_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, (bf16*)y, HW, C);
    return wg_check_launch("wg_nchw_to_tokens_bf16");
}

// ------------------------------------------------------------------------------------------------------------
// Dense positional encoding as token rows (prompt_encoder.py:203-229): pe[y*w+x, :] = [sin | cos](2 pi ((2c-1) @ G)),
// c = ((x+0.5)/w, (y+0.5)/h), G = positional_encoding_gaussian_matrix [2, F] (fp32 buffer).  Input independent:
// computed once per model.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wg_dense_pe_kernel(const float* G, float* pe, int h, int w, int F) {
    const long total = (long)h * w * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const int t = (int)(i / F);
        const int x = t % w, y = t / w;
        const float cx = 2.0f * ((x + 0.5f) / w) - 1.0f, cy = 2.0f * ((y + 0.5f) / h) - 1.0f;
        const float v = 6.283185307179586f * (cx * G[f] + cy * G[F + f]);
        pe[(long)t * 2 * F + f] = sinf(v);
        pe[(long)t * 2 * F + F + f] = cosf(v);
    }
}

extern "C" int wg_dense_pe_f32(const float* gaussian, float* pe_tokens, int h, int w, int num_feats, void* stream) {
    WG_REQUIRE(gaussian && pe_tokens && h > 0 && w > 0 && num_feats > 0, "dense_pe: bad arguments");
    const long total = (long)h * w * num_feats;
    hipLaunchKernelGGL(wg_dense_pe_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gaussian,
                       pe_tokens, h, w, num_feats);
    return wg_check_launch("wg_dense_pe_f32");
}

__global__ __launch_bounds__(256) void wg_cast_f32_bf16_kernel(const float* x, bf16* y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (bf16)x[i];
}
extern "C" int wg_cast_f32_to_bf16(const float* x, void* y, long n, void* stream) {
    WG_REQUIRE(x && y && n > 0, "cast: bad arguments");
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(wg_cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (bf16*)y, n);
    return wg_check_launch("wg_cast_f32_to_bf16");
}

// ------------------------------------------------------------------------------------------------------------
// Hyper-network mask product (mask_decoder.py:150-160): masks[t, k, Y, X] = sum_c hyper[t, k, c] * up[t, c, Y, X].
// `up` arrives as the pixel-shuffled output of the two transposed-conv GEMMs:
//   rows  = t*h*w*4 + (y*w + x)*4 + (dy*2 + dx)        (first ConvT sub-pixel)
//   cols  = (dy2*2 + dx2)*Cu + c                        (second ConvT sub-pixel, Cu = 32 channels)
// and lands at Y = 4y + 2dy + dy2, X = 4x + 2dx + dx2.  One thread per (row, second sub-pixel): a 32-long dot.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wg_hyper_mask_kernel(const bf16* up, const bf16* hyper, float* masks, int T, int h,
                                                            int w, int Cu, int nmask_total, int k0, int nk) {
    const long total = (long)T * h * w * 4 * 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int sub2 = (int)(i & 3);
        const long row = i >> 2;
        const int sub1 = (int)(row & 3);
        const long pix = row >> 2;
        const int x = (int)(pix % w), y = (int)((pix / w) % h), t = (int)(pix / ((long)w * h));
        const int Y = 4 * y + 2 * (sub1 >> 1) + (sub2 >> 1), X = 4 * x + 2 * (sub1 & 1) + (sub2 & 1);
        const bf16* u = up + row * (4 * Cu) + sub2 * Cu;
        float uf[32];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bf16x8 v = *(const bf16x8*)(u + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) uf[8 * c + e] = (float)v[e];
        }
        for (int k = 0; k < nk; ++k) {
            const bf16* hp = hyper + ((long)t * nmask_total + k0 + k) * Cu;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) s += uf[c] * (float)hp[c];
            masks[(((long)t * nk + k) * (4 * h) + Y) * (4 * w) + X] = s;
        }
    }
}

extern "C" int wg_hyper_mask_dot(const void* up, const void* hyper, float* masks, int T, int h, int w, int channels,
                                 int nmask_total, int first_mask, int num_masks, void* stream) {
    WG_REQUIRE(up && hyper && masks && T > 0 && h > 0 && w > 0, "hyper_mask_dot: bad arguments");
    WG_REQUIRE(channels == 32, "hyper_mask_dot: %d channels (only 32 = transformer_dim/8 compiled)", channels);
    WG_REQUIRE(first_mask >= 0 && num_masks > 0 && first_mask + num_masks <= nmask_total, "hyper_mask_dot: bad mask slice");
    const long total = (long)T * h * w * 16;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(wg_hyper_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)up, (const bf16*)hyper,
                       masks, T, h, w, channels, nmask_total, first_mask, num_masks);
    return wg_check_launch("wg_hyper_mask_dot");
}

// ------------------------------------------------------------------------------------------------------------
// Sam.postprocess_masks (sam.py:137-172) as ONE pass: bilinear low-res -> img_size^2, crop to (in_h, in_w), bilinear
// -> (out_h, out_w); the img_size^2 intermediate (4 MB / mask) is never materialised: each output pixel evaluates
// its 4 taps of the second resample, each of which evaluates 4 taps of the first.  PyTorch's align_corners=False
// source-index rule is reproduced exactly: src = max(scale*(dst+0.5)-0.5, 0), i0 = floor, i1 = i0 + (i0 < in-1).
// Algorithmic bytes: read N*lh*lw*4, write N*out_h*out_w*4.
// ------------------------------------------------------------------------------------------------------------
// The four source-index scales of the two resamples (torch: input size / output size in fp32), divided on the host: a device-side `/` under
// -ffast-math is a reciprocal approximation times the numerator.  (Through double: a ratio of two small integers is never a rounding tie.)
struct PostScales { float s1y, s1x, s2y, s2x; };
static PostScales wg_post_scales(int lh, int lw, int img, int in_h, int in_w, int out_h, int out_w) {
    return PostScales{(float)((double)lh / (double)img), (float)((double)lw / (double)img), (float)((double)in_h / (double)out_h),
                      (float)((double)in_w / (double)out_w)};
}

// One IEEE operation, one rounding, whatever the translation unit's floating-point mode: HIP's __fmul_rn / __fadd_rn are plain `*` / `+` unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined, and under -ffast-math the back end fuses and re-associates them (differently from one
// instantiation of a kernel to the next: the fused-score form of the postprocess kernel differed from the plain one by an ulp in column 0).
__device__ __forceinline__ float wg_mul_rn(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float wg_add_rn(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float wg_sub_rn(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void wg_src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    // separate roundings (no FMA contraction) so the source index and weight match PyTorch's CPU kernel bit for bit
    float s = wg_sub_rn(wg_mul_rn(scale, (float)dst + 0.5f), 0.5f);
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 < in_size - 1 ? i0 : in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = wg_sub_rn(s, (float)i0);
}

// grid (output column blocks, output rows, masks): no 64-bit index division per pixel (the first version spent more on `i % out_w` than on
// the sixteen taps); the four first-resample index sets a pixel needs (two rows, two columns) are formed once and shared by its four taps.
// SCORE: the workgroup also leaves {sum of sigmoid(x) over x > 0, count of x > 0} of its 256 pixels in score_ws -- the first pass of the mask
// score (model/walkgpt.py:540-542) without reading the masks back.
// SCORE 2: the second pass too -- the workgroup whose arrival completes a mask (an agent-scope ticket per mask: `tickets`, zero before the first
// launch and left zero) folds that mask's partials in the order wg_mask_score_final_kernel uses and writes score[n]: one dependent launch
// (4.2 us of a 211-us decode) less.  Hand-over as MI355X_MICROARCH.md prescribes for "the workgroup whose add came last": the partial is ONE
// 8-byte sc1 store by the lane that then drains it (s_waitcnt vmcnt(0)) and adds; the last arriver's wave reads every partial with sc1 loads
// after its add has returned (and one acquire, in that workgroup only) -- no release fence, no L2 write-back.
template <int SCORE, int ROWS>
__global__ __launch_bounds__(256) void wg_postprocess_kernel(const float* __restrict__ low, float* __restrict__ out, float* score_ws, int N, int lh, int lw, PostScales sc,
                                                             int in_h, int in_w, int out_h, int out_w, float* score, unsigned* tickets) {
    __shared__ float ssum[4], scnt[4];
    __shared__ unsigned last;
    const int ox = blockIdx.x * 256 + threadIdx.x, n = blockIdx.z;
    float s_ = 0.f, c_ = 0.f;     // (a pixel outside the row contributes nothing to the score)
    if (ox < out_w) {
        const float s1y = sc.s1y, s1x = sc.s1x, s2y = sc.s2y, s2x = sc.s2x;
        const float* m = low + (long)n * lh * lw;
        // the column taps of this thread's output column are the same for every row the workgroup covers (ROWS of them: the fused score wants
        // few, fat workgroups -- its tickets are adds to ONE word per mask and serialise at ~13 ns each)
        int x0, x1, xa[2][2];            // [second-resample tap][first-resample tap]
        float lx, lxa[2];
        wg_src_index(ox, s2x, in_w, x0, x1, lx);
        wg_src_index(x0, s1x, lw, xa[0][0], xa[0][1], lxa[0]);
        wg_src_index(x1, s1x, lw, xa[1][0], xa[1][1], lxa[1]);
        // rows in chunks of up to four: all 64 taps of a chunk are requested before the first is used (row by row, a store between two rows' loads
        // made every row its own memory round trip); rows past the image repeat the last one and are not stored
        constexpr int CH = ROWS < 4 ? ROWS : 4;
#pragma unroll
        for (int r0 = 0; r0 < ROWS; r0 += CH) {
            float tap[CH][2][2][4], ly[CH], lya[CH][2];
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int oy = blockIdx.y * ROWS + r0 + r;
                int y0, y1, ya[2][2];
                wg_src_index(oy < out_h ? oy : out_h - 1, s2y, in_h, y0, y1, ly[r]);
                wg_src_index(y0, s1y, lh, ya[0][0], ya[0][1], lya[r][0]);
                wg_src_index(y1, s1y, lh, ya[1][0], ya[1][1], lya[r][1]);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        tap[r][j][i][0] = m[ya[j][0] * lw + xa[i][0]]; tap[r][j][i][1] = m[ya[j][0] * lw + xa[i][1]];
                        tap[r][j][i][2] = m[ya[j][1] * lw + xa[i][0]]; tap[r][j][i][3] = m[ya[j][1] * lw + xa[i][1]];
                    }
            }
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int oy = blockIdx.y * ROWS + r0 + r;
                float up[2][2];              // the img^2 intermediate at (y0|y1, x0|x1), operation by operation as torch's first resample
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float wy = wg_sub_rn(1.f, lya[r][j]), wx = wg_sub_rn(1.f, lxa[i]);
                        up[j][i] = wg_add_rn(wg_mul_rn(wy, wg_add_rn(wg_mul_rn(wx, tap[r][j][i][0]), wg_mul_rn(lxa[i], tap[r][j][i][1]))),
                                             wg_mul_rn(lya[r][j], wg_add_rn(wg_mul_rn(wx, tap[r][j][i][2]), wg_mul_rn(lxa[i], tap[r][j][i][3]))));
                    }
                const float vy = wg_sub_rn(1.f, ly[r]), vx = wg_sub_rn(1.f, lx);
                const float res = wg_add_rn(wg_mul_rn(vy, wg_add_rn(wg_mul_rn(vx, up[0][0]), wg_mul_rn(lx, up[0][1]))),
                                            wg_mul_rn(ly[r], wg_add_rn(wg_mul_rn(vx, up[1][0]), wg_mul_rn(lx, up[1][1]))));
                if (oy < out_h) {
                    out[((long)n * out_h + oy) * out_w + ox] = res;
                    if (SCORE != 0 && res > 0.f) { s_ += 1.0f / (1.0f + __expf(-res)); c_ += 1.f; }
                }
            }
        }
    }
    if constexpr (SCORE != 0) {
        s_ = wg_wave_sum(s_);
        c_ = wg_wave_sum(c_);
        if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s_; scnt[threadIdx.x >> 6] = c_; }
        __syncthreads();
        const int nblk = (int)(gridDim.y * gridDim.x);
        const long blk = (long)n * nblk + blockIdx.y * gridDim.x + blockIdx.x;
        if constexpr (SCORE == 1) {
            if (threadIdx.x == 0) {
                score_ws[blk * 2 + 0] = ssum[0] + ssum[1] + ssum[2] + ssum[3];
                score_ws[blk * 2 + 1] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
            }
        } else {
            if (threadIdx.x == 0) {
                const f32x2 pr = {ssum[0] + ssum[1] + ssum[2] + ssum[3], scnt[0] + scnt[1] + scnt[2] + scnt[3]};
                __hip_atomic_store((unsigned long long*)(score_ws + blk * 2), __builtin_bit_cast(unsigned long long, pr), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned t = __hip_atomic_fetch_add(tickets + n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = t == (unsigned)nblk - 1u ? 1u : 0u;
                if (t == (unsigned)nblk - 1u) {   // (several workgroups share a CU here, outside what the guide measured sc1 loads alone for: + the acquire)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();                      // (the ticket's return is behind this barrier for every wave of the workgroup)
            if (last && threadIdx.x < 64) {
                float s = 0.f, c = 0.f;
                for (int i = threadIdx.x; i < nblk; i += 64) {
                    const unsigned long long raw = __hip_atomic_load((const unsigned long long*)(score_ws + ((long)n * nblk + i) * 2), __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_AGENT);
                    const f32x2 pr = __builtin_bit_cast(f32x2, raw);
                    s += pr.x;
                    c += pr.y;
                }
                s = wg_wave_sum(s);
                c = wg_wave_sum(c);
                if (threadIdx.x == 0) {
                    score[n] = s / (c + 1e-6f);
                    __hip_atomic_store(tickets + n, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
                }
            }
        }
    }
}

// Adjoint of the two resamples (training: d loss / d low-res logits from d loss / d masks).  Rounds 2-3 scattered: every output pixel added its gradient
// to the 16 low-res taps it was blended from with fp32 atomics.
// Round 4, a GATHER: a thread per low-res pixel sums, in a fixed order, the output pixels whose taps reach it -- no atomics (the
// scatter form issued 16 fp32 atomics per output pixel: 604 us for eight 448 x 448 masks, the largest kernel of a head step), dlow is written,
// not accumulated, and two runs give the same bits.  Both resamples are separable, so the weight of output pixel (oy, ox) on low-res pixel (r, c) is
// Wy(oy, r) Wx(ox, c); each factor is found by running the FORWARD index rule on the candidate and keeping what lands on r (c): border clamps and
// the crop need no case analysis.  Candidates: a superset of the output rows (columns) whose second-resample taps fall on intermediate rows whose
// first-resample taps fall on r -- (2 / s1 + 3) / s2 + 3 of them (7 at 1024 -> 448, 17 at 768 -> 1080).
__device__ __forceinline__ float wg_post_w(int o, float s2, int in_sz, float s1, int low_sz, int r) {
    int t0, t1, a0, a1;
    float l, la, w;
    wg_src_index(o, s2, in_sz, t0, t1, l);
    wg_src_index(t0, s1, low_sz, a0, a1, la);
    w = (1.f - l) * ((a0 == r ? 1.f - la : 0.f) + (a1 == r ? la : 0.f));
    wg_src_index(t1, s1, low_sz, a0, a1, la);
    return w + l * ((a0 == r ? 1.f - la : 0.f) + (a1 == r ? la : 0.f));
}
__device__ __forceinline__ void wg_post_range(int r, float s1, float s2, int in_sz, int out_sz, int& lo, int& hi) {
    float ylo = ((float)r - 0.5f) / s1 - 1.5f, yhi = ((float)r + 1.5f) / s1 + 0.5f;      // intermediate rows that can touch r (+-1)
    ylo = ylo < 0.f ? 0.f : ylo;
    yhi = yhi > (float)(in_sz - 1) ? (float)(in_sz - 1) : yhi;
    const float olo = (ylo - 0.5f) / s2 - 1.5f, ohi = (yhi + 1.5f) / s2 + 0.5f;          // output rows whose taps can touch those (+-1)
    lo = olo < 0.f ? 0 : (int)olo;
    hi = ohi > (float)(out_sz - 1) ? out_sz - 1 : (int)ohi;
}
__global__ __launch_bounds__(256) void wg_postprocess_bwd_gather_kernel(const float* __restrict__ dout, float* __restrict__ dlow, int lh, int lw, PostScales sc, int in_h,
                                                                        int in_w, int out_h, int out_w) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, n = blockIdx.z;
    if (c >= lw) return;
    int ylo, yhi, xlo, xhi;
    wg_post_range(r, sc.s1y, sc.s2y, in_h, out_h, ylo, yhi);
    wg_post_range(c, sc.s1x, sc.s2x, in_w, out_w, xlo, xhi);
    const float* g = dout + (long)n * out_h * out_w;
    float acc = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = wg_post_w(oy, sc.s2y, in_h, sc.s1y, lh, r);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int ox = xlo; ox <= xhi; ++ox) {
            const float wx = wg_post_w(ox, sc.s2x, in_w, sc.s1x, lw, c);
            row += wx * g[(long)oy * out_w + ox];
        }
        acc += wy * row;
    }
    dlow[((long)n * lh + r) * lw + c] = acc;
}

extern "C" int wg_postprocess_masks_bwd_f32(const float* dout, float* dlow, int N, int low_h, int low_w, int img_size, int in_h, int in_w, int out_h,
                                            int out_w, void* stream) {
    WG_REQUIRE(dout && dlow && N > 0 && low_h > 0 && low_w > 0 && img_size > 0 && in_h > 0 && in_w > 0 && in_h <= img_size && in_w <= img_size &&
                   out_h > 0 && out_w > 0, "postprocess_bwd: bad arguments");
    WG_REQUIRE(low_h <= 65535 && N <= 65535, "postprocess_bwd: more than 65535 low-res rows or masks per call");
    // (dlow is overwritten: callers written against the scatter form zero it first, which is harmless)
    hipLaunchKernelGGL(wg_postprocess_bwd_gather_kernel, dim3((unsigned)((low_w + 255) / 256), (unsigned)low_h, (unsigned)N), dim3(256), 0, (hipStream_t)stream,
                       dout, dlow, low_h, low_w, wg_post_scales(low_h, low_w, img_size, in_h, in_w, out_h, out_w), in_h, in_w, out_h, out_w);
    return wg_check_launch("wg_postprocess_masks_bwd_f32");
}

extern "C" int wg_postprocess_masks_f32(const float* low_res, float* out, int N, int low_h, int low_w, int img_size,
                                        int in_h, int in_w, int out_h, int out_w, void* stream) {
    WG_REQUIRE(low_res && out && N > 0 && low_h > 0 && low_w > 0 && img_size > 0, "postprocess: bad arguments");
    WG_REQUIRE(in_h > 0 && in_w > 0 && in_h <= img_size && in_w <= img_size && out_h > 0 && out_w > 0,
               "postprocess: crop (%d,%d) must lie inside the %d^2 padded image", in_h, in_w, img_size);
    WG_REQUIRE(out_h <= 65535 && N <= 65535, "postprocess: more than 65535 output rows or masks per call");
    hipLaunchKernelGGL((wg_postprocess_kernel<0, 1>), dim3((unsigned)((out_w + 255) / 256), (unsigned)out_h, (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, low_res, out, nullptr, N, low_h, low_w, wg_post_scales(low_h, low_w, img_size, in_h, in_w, out_h, out_w), in_h, in_w,
                       out_h, out_w, nullptr, nullptr);
    return wg_check_launch("wg_postprocess_masks_f32");
}


// ------------------------------------------------------------------------------------------------------------
// Mask score (walkgpt.py:540-542, :737): sum(sigmoid(x) [x>0]) / (count[x>0] + 1e-6) per mask.
// Two passes: `nblk` blocks per mask write (sum, count) partials into the caller's workspace [N*nblk*2] floats, then
// one wave per mask folds them in a fixed order (bitwise reproducible; no atomics).  HBM-bound: reads N*hw*4 bytes.
// ------------------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void wg_mask_score_partial_kernel(const float* masks, float* ws, long hw, int nblk) {
    __shared__ float ssum[4], scnt[4];
    const int n = blockIdx.y, blk = blockIdx.x;
    const float* m = masks + (long)n * hw;
    float s = 0.f, c = 0.f;
    if (VEC) {  // hw % 4 == 0 and 16-byte aligned base: 16-byte loads
        for (long i = (long)blk * 256 + threadIdx.x; i < (hw >> 2); i += (long)nblk * 256) {
            const f32x4 x = *(const f32x4*)(m + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x[e] > 0.f) { s += 1.0f / (1.0f + __expf(-x[e])); c += 1.f; }
        }
    } else {
        for (long i = (long)blk * 256 + threadIdx.x; i < hw; i += (long)nblk * 256)
            if (m[i] > 0.f) { s += 1.0f / (1.0f + __expf(-m[i])); c += 1.f; }
    }
    s = wg_wave_sum(s);
    c = wg_wave_sum(c);
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; scnt[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[((long)n * nblk + blk) * 2 + 0] = ssum[0] + ssum[1] + ssum[2] + ssum[3];
        ws[((long)n * nblk + blk) * 2 + 1] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
    }
}

__global__ __launch_bounds__(64) void wg_mask_score_final_kernel(const float* ws, float* score, int nblk) {
    const int n = blockIdx.x;
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) { s += ws[((long)n * nblk + i) * 2]; c += ws[((long)n * nblk + i) * 2 + 1]; }
    s = wg_wave_sum(s);
    c = wg_wave_sum(c);
    if (threadIdx.x == 0) score[n] = s / (c + 1e-6f);
}

static long wg_mask_score_blocks(long hw) {
    long nblk = (hw + 4095) / 4096;
    return nblk < 1 ? 1 : (nblk > 64 ? 64 : nblk);
}

extern "C" long wg_mask_score_workspace_floats(int N, long hw) { return (long)N * wg_mask_score_blocks(hw) * 2; }

extern "C" int wg_mask_score_f32(const float* masks, float* score, float* workspace, long workspace_floats, int N, long hw,
                                 void* stream) {
    WG_REQUIRE(masks && score && workspace && N > 0 && hw > 0, "mask_score: bad arguments");
    const long nblk = wg_mask_score_blocks(hw);
    WG_REQUIRE(workspace_floats >= (long)N * nblk * 2, "mask_score: workspace too small (need %ld floats)", (long)N * nblk * 2);
    const bool vec = (((uintptr_t)masks & 15) == 0) && (hw & 3) == 0;
    if (vec) hipLaunchKernelGGL(wg_mask_score_partial_kernel<true>, dim3((unsigned)nblk, N), dim3(256), 0, (hipStream_t)stream, masks, workspace, hw, (int)nblk);
    else hipLaunchKernelGGL(wg_mask_score_partial_kernel<false>, dim3((unsigned)nblk, N), dim3(256), 0, (hipStream_t)stream, masks, workspace, hw, (int)nblk);
    hipLaunchKernelGGL(wg_mask_score_final_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, workspace, score, (int)nblk);
    return wg_check_launch("wg_mask_score_f32");
}

// Sam.postprocess_masks (sam.py:137-172) and the mask score of the result (model/walkgpt.py:540-542, :737) in one pass over the output: the
// postprocess workgroups leave their {sum, count} partials in `workspace` (wg_postprocess_score_workspace_floats floats), one wave per mask
// folds them in a fixed order.  out [N, out_h, out_w] fp32, score [N] fp32.
extern "C" long wg_postprocess_score_workspace_floats(int N, int out_h, int out_w) { return (long)N * out_h * ((out_w + 255) / 256) * 2; }

extern "C" int wg_postprocess_masks_score_f32(const float* low_res, float* out, float* score, float* workspace, long workspace_floats, int N,
                                              int low_h, int low_w, int img_size, int in_h, int in_w, int out_h, int out_w, void* stream) {
    WG_REQUIRE(low_res && out && score && workspace && N > 0 && low_h > 0 && low_w > 0 && img_size > 0, "postprocess_score: bad arguments");
    WG_REQUIRE(in_h > 0 && in_w > 0 && in_h <= img_size && in_w <= img_size && out_h > 0 && out_w > 0,
               "postprocess_score: crop (%d,%d) must lie inside the %d^2 padded image", in_h, in_w, img_size);
    WG_REQUIRE(out_h <= 65535 && N <= 65535, "postprocess_score: more than 65535 output rows or masks per call");
    const int gx = (out_w + 255) / 256;
    WG_REQUIRE(workspace_floats >= (long)N * out_h * gx * 2, "postprocess_score: workspace too small (need %ld floats)", (long)N * out_h * gx * 2);
    hipLaunchKernelGGL((wg_postprocess_kernel<1, 1>), dim3((unsigned)gx, (unsigned)out_h, (unsigned)N), dim3(256), 0, (hipStream_t)stream, low_res, out,
                       workspace, N, low_h, low_w, wg_post_scales(low_h, low_w, img_size, in_h, in_w, out_h, out_w), in_h, in_w, out_h, out_w, nullptr, nullptr);
    hipLaunchKernelGGL(wg_mask_score_final_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, workspace, score, out_h * gx);
    return wg_check_launch("wg_postprocess_masks_score_f32");
}

// The same in ONE launch: `tickets` holds N words that are zero before the first call and that every call leaves zero (calls sharing them
// must be ordered, e.g. on one stream); same pixels; scores equal to the two-launch form's up to the summation order (a workgroup
// covers eight output rows here, one there).
extern "C" int wg_postprocess_masks_score_fused_f32(const float* low_res, float* out, float* score, float* workspace, long workspace_floats, unsigned* tickets,
                                                    int N, int low_h, int low_w, int img_size, int in_h, int in_w, int out_h, int out_w, void* stream) {
    WG_REQUIRE(low_res && out && score && workspace && tickets && N > 0 && low_h > 0 && low_w > 0 && img_size > 0, "postprocess_score_fused: bad arguments");
    WG_REQUIRE(in_h > 0 && in_w > 0 && in_h <= img_size && in_w <= img_size && out_h > 0 && out_w > 0,
               "postprocess_score_fused: bad sizes (input %dx%d inside %d, output %dx%d)", in_h, in_w, img_size, out_h, out_w);
    WG_REQUIRE(((uintptr_t)workspace & 7) == 0, "postprocess_score_fused: the workspace must be 8-byte aligned");
    const int gx = (out_w + 255) / 256;
    WG_REQUIRE(workspace_floats >= (long)N * out_h * gx * 2, "postprocess_score_fused: workspace too small (need %ld floats)", (long)N * out_h * gx * 2);
    constexpr int ROWS = 8;
    hipLaunchKernelGGL((wg_postprocess_kernel<2, ROWS>), dim3((unsigned)gx, (unsigned)((out_h + ROWS - 1) / ROWS), (unsigned)N), dim3(256), 0, (hipStream_t)stream, low_res, out,
                       workspace, N, low_h, low_w, wg_post_scales(low_h, low_w, img_size, in_h, in_w, out_h, out_w), in_h, in_w, out_h, out_w, score, tickets);
    return wg_check_launch("wg_postprocess_masks_score_fused_f32");
}

// ------------------------------------------------------------------------------------------------------------
// per-mask statistics of the predicted logits against the ground truth, one pass over HBM.
//   MODE 0  eval metric   intersectionAndUnionGPU(pred > 0, gt, K = 2, ignore_index)  (utils/utils.py:192-204, called
//           evaluation_walkgpt.py:936-944): joint counts n[o][t] over pixels whose gt is not `ignore`; the thresholding
//           happens here, so logits never leave the GPU un-thresholded.
//   MODE 1  loss forward  sigmoid_ce_loss / dice_loss (utils/utils_walkgpt.py:76-120): sums of bce, sigmoid*t, sigmoid, t.
// Same two-pass, atomics-free reduction as the mask score: nblk blocks per mask -> partials [N, nblk, 4] -> fold.
// ------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void wg_mask_stats_partial_kernel(const float* pred, const float* gt, float* ws, long hw, int nblk,
                                                                    float ignore) {
    __shared__ float red[4][4];
    const int n = blockIdx.y, blk = blockIdx.x;
    const float* p = pred + (long)n * hw;
    const float* t = gt + (long)n * hw;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blk * 256 + threadIdx.x; i < hw; i += (long)nblk * 256) {
        const float x = p[i], y = t[i];
        if (MODE == 0) {
            if (y != ignore) {
                const int o = x > 0.f ? 1 : 0, g = (int)y;
                a[0] += (o == 0 && g == 0) ? 1.f : 0.f;
                a[1] += (o == 0 && g == 1) ? 1.f : 0.f;
                a[2] += (o == 1 && g == 0) ? 1.f : 0.f;
                a[3] += (o == 1 && g == 1) ? 1.f : 0.f;
            }
        } else {
            const float s = 1.0f / (1.0f + __expf(-x));
            a[0] += fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x)));   // binary_cross_entropy_with_logits, stable form
            a[1] += s * y;
            a[2] += s;
            a[3] += y;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = wg_wave_sum(a[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = a[k];
    __syncthreads();
    if (threadIdx.x < 4)
        ws[((long)n * nblk + blk) * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

template <int MODE>
__global__ __launch_bounds__(64) void wg_mask_stats_final_kernel(const float* ws, float* out, int nblk, long hw, float scale, float eps) {
    const int n = blockIdx.x;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nblk; i += 64)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += ws[((long)n * nblk + i) * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = wg_wave_sum(a[k]);
    if (threadIdx.x == 0) {
        if (MODE == 0) {   // [inter0, inter1, union0, union1, target0, target1]
            const float i0 = a[0], i1 = a[3], o0 = a[0] + a[1], o1 = a[2] + a[3], t0 = a[0] + a[2], t1 = a[1] + a[3];
            float* o = out + (long)n * 6;
            o[0] = i0; o[1] = i1; o[2] = o0 + t0 - i0; o[3] = o1 + t1 - i1; o[4] = t0; o[5] = t1;
        } else if (MODE == 1) {   // [mean bce, dice loss] of this mask
            out[(long)n * 2 + 0] = a[0] / (float)hw;
            out[(long)n * 2 + 1] = 1.0f - (2.0f * a[1] / scale + eps) / (a[2] / scale + a[3] / scale + eps);
        } else {                  // MODE 2 (loss backward): the dice loss's numerator and denominator
            out[(long)n * 2 + 0] = 2.0f * a[1] / scale + eps;
            out[(long)n * 2 + 1] = a[2] / scale + a[3] / scale + eps;
        }
    }
}

static long wg_mask_stats_blocks(long hw) {
    long nblk = (hw + 8191) / 8192;
    return nblk < 1 ? 1 : (nblk > 64 ? 64 : nblk);
}
extern "C" long wg_mask_stats_workspace_floats(int N, long hw) { return (long)N * wg_mask_stats_blocks(hw) * 4; }

extern "C" int wg_mask_iou_f32(const float* pred_logits, const float* gt, float* out6, float* workspace, long workspace_floats,
                               int N, long hw, float ignore_value, void* stream) {
    WG_REQUIRE(pred_logits && gt && out6 && workspace && N > 0 && hw > 0, "mask_iou: bad arguments");
    const long nblk = wg_mask_stats_blocks(hw);
    WG_REQUIRE(workspace_floats >= (long)N * nblk * 4, "mask_iou: workspace too small (need %ld floats)", (long)N * nblk * 4);
    hipLaunchKernelGGL(wg_mask_stats_partial_kernel<0>, dim3((unsigned)nblk, N), dim3(256), 0, (hipStream_t)stream, pred_logits, gt,
                       workspace, hw, (int)nblk, ignore_value);
    hipLaunchKernelGGL(wg_mask_stats_final_kernel<0>, dim3(N), dim3(64), 0, (hipStream_t)stream, workspace, out6, (int)nblk, hw, 1.f, 0.f);
    return wg_check_launch("wg_mask_iou_f32");
}

extern "C" int wg_mask_losses_f32(const float* pred_logits, const float* targets, float* out2, float* workspace,
                                  long workspace_floats, int N, long hw, float dice_scale, float dice_eps, void* stream) {
    WG_REQUIRE(pred_logits && targets && out2 && workspace && N > 0 && hw > 0, "mask_losses: bad arguments");
    const long nblk = wg_mask_stats_blocks(hw);
    WG_REQUIRE(workspace_floats >= (long)N * nblk * 4, "mask_losses: workspace too small (need %ld floats)", (long)N * nblk * 4);
    hipLaunchKernelGGL(wg_mask_stats_partial_kernel<1>, dim3((unsigned)nblk, N), dim3(256), 0, (hipStream_t)stream, pred_logits, targets,
                       workspace, hw, (int)nblk, 0.f);
    hipLaunchKernelGGL(wg_mask_stats_final_kernel<1>, dim3(N), dim3(64), 0, (hipStream_t)stream, workspace, out2, (int)nblk, hw,
                       dice_scale, dice_eps);
    return wg_check_launch("wg_mask_losses_f32");
}

// d(g_bce * sigmoid_ce_loss + g_dice * dice_loss) / d logits (utils_walkgpt.py:76-120; g_* already carry the 1 / (num_masks + 1e-8) of the final
// mean and the upstream gradient).  Per mask: bce_mean' = (s - y) / hw;  dice = 1 - num / den, num = 2 sum(s y) / scale + eps,
// den = (sum s + sum y) / scale + eps:  d dice / d s_i = (num - 2 y_i den) / (scale den^2),  ds / dx = s (1 - s).
__global__ __launch_bounds__(256) void wg_mask_losses_bwd_kernel(const float* pred, const float* gt, const float* numden, float* dpred, long hw, float g_bce,
                                                                 float g_dice, float scale, const float* g_dev) {
    const int n = blockIdx.y;
    if (g_dev) {   // upstream gradients read on the device (no host synchronisation in the caller's backward pass)
        g_bce = g_dev[0];
        g_dice = g_dev[1];
    }
    const float num = numden[(long)n * 2], den = numden[(long)n * 2 + 1];
    const float kb = g_bce / (float)hw, kd = g_dice / (scale * den * den);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const float x = pred[(long)n * hw + i], y = gt[(long)n * hw + i];
        const float sg = 1.0f / (1.0f + __expf(-x));
        dpred[(long)n * hw + i] = kb * (sg - y) + kd * (num - 2.0f * y * den) * sg * (1.0f - sg);
    }
}

static int wg_mask_losses_bwd_impl(const float* pred_logits, const float* targets, float* dpred, float* workspace, long workspace_floats, int N, long hw,
                                   float g_bce, float g_dice, const float* g_dev, float dice_scale, float dice_eps, void* stream);
extern "C" int wg_mask_losses_bwd_f32(const float* pred_logits, const float* targets, float* dpred, float* workspace, long workspace_floats, int N, long hw,
                                      float g_bce, float g_dice, float dice_scale, float dice_eps, void* stream) {
    return wg_mask_losses_bwd_impl(pred_logits, targets, dpred, workspace, workspace_floats, N, hw, g_bce, g_dice, nullptr, dice_scale, dice_eps, stream);
}
// ... with the two upstream gradients {g_bce, g_dice} in device memory (graph-capturable backward passes: no host read)
extern "C" int wg_mask_losses_bwd_dev_f32(const float* pred_logits, const float* targets, float* dpred, float* workspace, long workspace_floats, int N, long hw,
                                          const float* g2, float dice_scale, float dice_eps, void* stream) {
    WG_REQUIRE(g2, "mask_losses_bwd_dev: null gradient pointer");
    return wg_mask_losses_bwd_impl(pred_logits, targets, dpred, workspace, workspace_floats, N, hw, 0.f, 0.f, g2, dice_scale, dice_eps, stream);
}
static int wg_mask_losses_bwd_impl(const float* pred_logits, const float* targets, float* dpred, float* workspace, long workspace_floats, int N, long hw,
                                   float g_bce, float g_dice, const float* g_dev, float dice_scale, float dice_eps, void* stream) {
    WG_REQUIRE(pred_logits && targets && dpred && workspace && N > 0 && hw > 0, "mask_losses_bwd: bad arguments");
    const long nblk = wg_mask_stats_blocks(hw);
    WG_REQUIRE(workspace_floats >= (long)N * nblk * 4 + 2L * N, "mask_losses_bwd: workspace too small (need %ld floats)", (long)N * nblk * 4 + 2L * N);
    float* numden = workspace + (long)N * nblk * 4;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wg_mask_stats_partial_kernel<1>, dim3((unsigned)nblk, N), dim3(256), 0, st, pred_logits, targets, workspace, hw, (int)nblk, 0.f);
    hipLaunchKernelGGL(wg_mask_stats_final_kernel<2>, dim3(N), dim3(64), 0, st, workspace, numden, (int)nblk, hw, dice_scale, dice_eps);
    long gx = (hw + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(wg_mask_losses_bwd_kernel, dim3((unsigned)gx, N), dim3(256), 0, st, pred_logits, targets, numden, dpred, hw, g_bce, g_dice, dice_scale, g_dev);
    return wg_check_launch("wg_mask_losses_bwd_f32");
}


// ---------------------------------------------------------------------------------------------------------------------
// the cost matrix of match_pred() (/root/reference/utils/matcher.py:93-133, called at
// evaluation_walkgpt.py:751 and train_walkgpt.py:937) -- both mask sets bilinearly sampled at one shared set of points
// (point_sample :64-90 = grid_sample(2p-1, align_corners=False, zero padding)), then
//   C[p,t] = mean_n( BCE(x_pn, 1) y_tn + BCE(x_pn, 0) (1 - y_tn) )  +  1 - (2 sum_n s_pn y_tn + 1) / (sum_n s_pn + sum_n y_tn + 1)
// (batch_sigmoid_ce_loss :33-56, batch_dice_loss :10-25; s = sigmoid(x)).  The Hungarian assignment on the [P,T] matrix stays
// on the host (scipy), as in the reference.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wg_point_sample_kernel(const float* masks, const float* points, float* out, int H, int W, int NP) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= NP) return;
    const float* m = masks + (long)blockIdx.y * H * W;
    // grid_sample, align_corners=False: pixel coordinate = ((2p - 1 + 1) * size - 1) / 2 = p * size - 0.5
    const float x = points[2 * n] * (float)W - 0.5f, y = points[2 * n + 1] * (float)H - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = x - fx, wy1 = y - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    auto at = [&](int yy, int xx) { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? m[(long)yy * W + xx] : 0.f; };
    out[(long)blockIdx.y * NP + n] = at(y0, x0) * wy0 * wx0 + at(y0, x0 + 1) * wy0 * wx1 + at(y0 + 1, x0) * wy1 * wx0 + at(y0 + 1, x0 + 1) * wy1 * wx1;
}

__global__ __launch_bounds__(256) void wg_match_cost_kernel(const float* sp, const float* st, float* cost, int T, int NP) {
    __shared__ float red[4][4];
    const int p = blockIdx.x / T, t = blockIdx.x % T;
    const float* x = sp + (long)p * NP;
    const float* y = st + (long)t * NP;
    float ce = 0.f, num = 0.f, ss = 0.f, sy = 0.f;
    for (int n = threadIdx.x; n < NP; n += 256) {
        const float v = x[n], g = y[n];
        const float l1p = log1pf(__expf(-fabsf(v)));
        const float pos = fmaxf(v, 0.f) - v + l1p;     // BCE-with-logits against 1
        const float neg = fmaxf(v, 0.f) + l1p;         //                 against 0
        ce += pos * g + neg * (1.f - g);
        const float s = 1.f / (1.f + __expf(-v));
        num += s * g; ss += s; sy += g;
    }
    float v4[4] = {ce, num, ss, sy};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v4[i] = wg_wave_sum(v4[i]);
        if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = v4[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = red[i][0] + red[i][1] + red[i][2] + red[i][3];
        cost[blockIdx.x] = r[0] / (float)NP + 1.f - (2.f * r[1] + 1.f) / (r[2] + r[3] + 1.f);
    }
}

extern "C" long wg_match_cost_workspace_floats(int P, int T, int NP) { return (long)(P + T) * NP; }

extern "C" int wg_match_cost_f32(const float* pred_logits, const float* targets, const float* points, float* cost, float* workspace,
                                 long workspace_floats, int P, int T, int H, int W, int NP, void* stream) {
    WG_REQUIRE(pred_logits && targets && points && cost && workspace, "match_cost: null operand");
    WG_REQUIRE(P > 0 && T > 0 && H > 0 && W > 0 && NP > 0, "match_cost: bad shape");
    WG_REQUIRE(workspace_floats >= (long)(P + T) * NP, "match_cost: workspace too small (need %ld floats)", (long)(P + T) * NP);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wg_point_sample_kernel, dim3((NP + 255) / 256, P), dim3(256), 0, st, pred_logits, points, workspace, H, W, NP);
    hipLaunchKernelGGL(wg_point_sample_kernel, dim3((NP + 255) / 256, T), dim3(256), 0, st, targets, points, workspace + (long)P * NP, H, W, NP);
    hipLaunchKernelGGL(wg_match_cost_kernel, dim3(P * T), dim3(256), 0, st, (const float*)workspace, (const float*)(workspace + (long)P * NP), cost, T, NP);
    return wg_check_launch("wg_match_cost_f32");
}
