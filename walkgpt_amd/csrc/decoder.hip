// Fused kernels of SAM's mask decoder (gfx950).  The chain is latency-bound and tiny next to the encoders, so these kernels spend
// their freedom on PRECISION (fp32 state, bf16 weights, activations fed to the MFMAs as a bf16 hi + lo pair: ~16 significant bits)
// and on launch count, not on FLOP/s.
//
//   wg_upscale_mask_bf16   output_upscaling + hypernetwork product of mask_decoder.py:140-160 in ONE launch:
//                          ConvT(256->64,k2,s2) -> LayerNorm2d(eps 1e-6) -> GELU -> ConvT(64->32,k2,s2) -> GELU -> <hyper_in, .>
//                          Every step is local to one image token (a k2/s2 transposed convolution maps a token to its own 2x2 block),
//                          so a token row goes 256 ch -> 4 sub-pixels x 64 ch -> 16 sub-sub-pixels x 32 ch -> 16 logits without leaving
//                          the workgroup; the [P,64,2h,2w] and [P,32,4h,4w] tensors of the reference never exist.
#include "wg_common.h"

namespace {

// A and B fragments of mfma_f32_16x16x32_bf16: lane (l16 = lane & 15, kg = lane >> 4) holds k = 8*kg .. 8*kg+7 of row / column l16;
// the accumulator holds rows 4*kg .. 4*kg+3 of column l16.
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// sum over the 16 lanes of a DPP row (lanes 16*kg .. 16*kg+15); every lane of the row ends with the total
__device__ __forceinline__ float row16_sum(float v) {
    v += WG_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
    v += WG_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
    v += WG_DPP(v, 0x124);   // row_ror:4
    v += WG_DPP(v, 0x128);   // row_ror:8
    return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + wg_erf(x * 0.70710678118654752440f)); }

struct UpArgs {
    const bf16* x; long ldx;           // image tokens [P*hw, 256]
    const bf16* w1; const bf16* b1;    // ConvT 1 as a GEMM: [(dy,dx,64), 256], bias [64]
    const bf16* g1; const bf16* be1;   // LayerNorm2d(64)
    const bf16* w2; const bf16* b2;    // ConvT 2 as a GEMM: [(dy,dx,32), 64], bias [32]
    const float* hyper;                // [P, nmask_total, 32]
    float* out;                        // [P, num_masks, 4h, 4w]
    int P, h, w, nmask_total, first_mask, num_masks;
    float eps;
};

constexpr int UP_ROWS = 64;            // token rows per workgroup (four waves x 16)
constexpr int UP_PITCH = 72;           // bf16 elements per LDS row of the 64-channel intermediate (144 B: conflict-free b128 reads)

__global__ __launch_bounds__(256) void wg_upscale_mask_kernel(UpArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 vhi[4][64 * UP_PITCH];
    __shared__ __attribute__((aligned(16))) bf16 vlo[4][64 * UP_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const long hw = (long)a.h * a.w;
    const long total = (long)a.P * hw;
    const long r0 = (long)blockIdx.x * UP_ROWS + wave * 16;
    if (r0 >= total) return;            // (whole wave: rows come in multiples of 16, hw % 16 == 0 is checked by the host)

    // ---- GEMM 1: u[16 rows][256] = x[16][256] . w1^T -------------------------------------------------------------------------
    bf16x8 xa[8];
    {
        const bf16* xp = a.x + (r0 + l16) * a.ldx + 8 * kg;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xa[ks] = *(const bf16x8*)(xp + 32 * ks);
    }
    f32x4 u[16];
#pragma unroll
    for (int nb = 0; nb < 16; ++nb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bf16* wp = a.w1 + (long)(nb * 16 + l16) * 256 + 8 * kg;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = mfma16(xa[ks], *(const bf16x8*)(wp + 32 * ks), acc);
        u[nb] = acc;
    }
    // ---- bias, LayerNorm2d over the 64 channels of each (row, sub-pixel), GELU; -> LDS as bf16 hi + lo, one row per (row, sub-pixel)
    float b1v[4], g1v[4], be1v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b1v[j] = (float)a.b1[16 * j + l16];
        g1v[j] = (float)a.g1[16 * j + l16];
        be1v[j] = (float)a.be1[16 * j + l16];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[4], sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = u[4 * s + j][i] + b1v[j];
                sum += v[j];
            }
            const float mean = row16_sum(sum) * (1.0f / 64.0f);
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] -= mean;
                sq += v[j] * v[j];
            }
            const float rstd = 1.0f / sqrtf(row16_sum(sq) * (1.0f / 64.0f) + a.eps);
            const int srow = (4 * kg + i) * 4 + s;     // LDS row of (token row 4*kg+i, sub-pixel s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = gelu_erf(v[j] * rstd * g1v[j] + be1v[j]);
                const bf16 hi = (bf16)y;
                vhi[wave][srow * UP_PITCH + 16 * j + l16] = hi;
                vlo[wave][srow * UP_PITCH + 16 * j + l16] = (bf16)(y - (float)hi);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();      // the slab is private to the wave and LDS operations of one wave complete in order

    // ---- GEMM 2: z[64 sub-pixel rows][128] = v[64][64] . w2^T (hi + lo), bias, GELU, dot with hyper_in -> 16 logits per token --------
    bf16x8 wb[8][2];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wb[nb][ks] = *(const bf16x8*)(a.w2 + (long)(nb * 16 + l16) * 64 + 32 * ks + 8 * kg);
    float b2v[2];
    b2v[0] = (float)a.b2[l16];
    b2v[1] = (float)a.b2[16 + l16];
    const long p = r0 / hw;                 // prompt of this wave's rows (16 | hw: a wave never straddles two prompts)
    const int t0 = (int)(r0 - p * hw);      // first token of the wave inside its prompt
    const int H4 = 4 * a.h, W4 = 4 * a.w;
    for (int mk = 0; mk < a.num_masks; ++mk) {
        const float* hy = a.hyper + ((long)p * a.nmask_total + a.first_mask + mk) * 32;
        const float hy0 = hy[l16], hy1 = hy[16 + l16];
        float* outp = a.out + ((long)p * a.num_masks + mk) * H4 * W4;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {      // 16 LDS rows = token rows 4*mb .. 4*mb+3 of the wave x 4 sub-pixels
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ah[ks] = *(const bf16x8*)(&vhi[wave][(mb * 16 + l16) * UP_PITCH + 32 * ks + 8 * kg]);
                al[ks] = *(const bf16x8*)(&vlo[wave][(mb * 16 + l16) * UP_PITCH + 32 * ks + 8 * kg]);
            }
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {  // sub-sub-pixel (dy2, dx2): columns 32*ss .. 32*ss+31 = N blocks 2*ss, 2*ss+1
                f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int nb = 2 * ss + half;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        acc = mfma16(ah[ks], wb[nb][ks], acc);
                        acc = mfma16(al[ks], wb[nb][ks], acc);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) part[i] += gelu_erf(acc[i] + b2v[half]) * (half == 0 ? hy0 : hy1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float logit = row16_sum(part[i]);
                    // accumulator row 4*kg+i of block mb = LDS row mb*16 + 4*kg + i = (token row 4*mb + kg, sub-pixel i)
                    if (l16 == 4 * ss + i) {
                        const int t = t0 + 4 * mb + kg;
                        const int ty = t / a.w, tx = t % a.w;
                        const int y = 4 * ty + 2 * (i >> 1) + (ss >> 1), x = 4 * tx + 2 * (i & 1) + (ss & 1);
                        outp[(long)y * W4 + x] = logit;
                    }
                }
            }
        }
    }
}

}  // namespace

// mask_decoder.py:140-160 after the transformer: `upscaled = output_upscaling(src)`; `masks = hyper_in @ upscaled`.
// x [P*h*w, 256] bf16 image tokens (channels-last rows), w1 [(dy,dx,64), 256] / w2 [(dy,dx,32), 64] the two transposed convolutions
// re-laid as GEMM weights, hyper [P, nmask_total, 32] fp32 -> out [P, num_masks, 4h, 4w] fp32 (masks first_mask .. +num_masks-1).
extern "C" int wg_upscale_mask_bf16(const void* x, long ldx, const void* w1, const void* b1, const void* ln_g, const void* ln_b, float eps,
                                    const void* w2, const void* b2, const float* hyper, float* out, int P, int h, int w,
                                    int nmask_total, int first_mask, int num_masks, void* stream) {
    WG_REQUIRE(x && w1 && b1 && ln_g && ln_b && w2 && b2 && hyper && out, "upscale_mask: null operand");
    WG_REQUIRE(P > 0 && h > 0 && w > 0 && ((long)h * w) % 16 == 0, "upscale_mask: h*w must be a positive multiple of 16");
    WG_REQUIRE(ldx % 8 == 0 && (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2) & 15) == 0, "upscale_mask: misaligned operand");
    WG_REQUIRE(first_mask >= 0 && num_masks > 0 && first_mask + num_masks <= nmask_total, "upscale_mask: bad mask range");
    UpArgs a{(const bf16*)x, ldx, (const bf16*)w1, (const bf16*)b1, (const bf16*)ln_g, (const bf16*)ln_b, (const bf16*)w2, (const bf16*)b2,
             hyper, out, P, h, w, nmask_total, first_mask, num_masks, eps};
    const long rows = (long)P * h * w;
    hipLaunchKernelGGL(wg_upscale_mask_kernel, dim3((unsigned)((rows + UP_ROWS - 1) / UP_ROWS)), dim3(256), 0, (hipStream_t)stream, a);
    return wg_check_launch("wg_upscale_mask");
}

// =====================================================================================================================================
//   wg_dec_tokens_f32      the token side of one TwoWayAttentionBlock (transformer.py:151-182) -- self attention, norm1, token->image
//                          cross attention over all image tokens, norm2, MLP, norm3, and the k / v projections that the image->token
//                          attention of the same block reads -- or of the tail of the transformer (final token->image attention,
//                          norm_final_attn: transformer.py:96-106) plus the hypernetwork MLPs and the IoU head (mask_decoder.py:146-160)
//                          in ONE launch, one workgroup per prompt.  The six tokens of a prompt live in LDS as fp32 for the whole
//                          kernel; Linear layers run on the matrix pipe with the tokens as a 16-row A operand (rows 6..15 zero) split
//                          into a bf16 hi + lo pair, weights (bf16, the checkpoint's precision) streamed from L2 as B fragments.
// =====================================================================================================================================
namespace {

constexpr int TK_THREADS = 512;
constexpr int TK_N = 6;          // tokens per prompt: iou + 4 mask tokens + 1 text prompt (mask_decoder.py:125-132)
constexpr int TK_C = 256;        // transformer_dim
constexpr int TK_KMAX = 2048;    // widest Linear input (mlp.lin2)

struct LinW { const bf16* w; const bf16* b; };
struct NormW { const bf16* g; const bf16* b; };
struct AttnW { LinW q, k, v, o; };

struct TokArgs {
    // mode 0: one TwoWayAttentionBlock; mode 1: final attention + heads
    int mode, skip_pe, P, hw;
    float* queries;            // [P, 6, 256] fp32 in / out
    const float* pe;           // [P, 6, 256] fp32: the prompt tokens as they entered the transformer (query_pe)
    AttnW self_attn; NormW norm1;
    LinW t2i_q, t2i_o; NormW norm2;          // (mode 1: final_attn_token_to_image q / out, norm_final_attn)
    LinW lin1, lin2; NormW norm3;
    LinW i2t_k, i2t_v;
    const bf16* Kimg; const bf16* Vimg; long ld_img, img_bs;   // projected image tokens: [P | 1, hw, ld_img] (img_bs = 0: shared by all prompts)
    bf16* k_i2t; bf16* v_i2t;                // out (mode 0): [P, 6, 128] bf16 keys / values of the image->token attention
    LinW hyper[4][3]; LinW iou[3];           // mode 1
    float* hyper_out;                        // [P, 4, 32] fp32
    float* iou_out;                          // [P, 4] fp32
    float eps;
};

// y[r][n] = act(x[r][:] . W[n][:] + b[n]) (+ res[r][n]) for r < rows (<= 6), n < N.  x, y, res: LDS fp32.  sh / sl: LDS staging of the
// bf16 hi / lo split of x, [8][K + 8].  Every thread of the workgroup calls it; the result is visible to all on return.
template <int K>
__device__ void tok_linear(const float* x, int ldx, int rows, LinW W, int N, float* y, int ldy, int act, const float* res, int ldres,
                           bf16* sh, bf16* sl) {
    constexpr int PITCH = K + 8;
    const int tid = threadIdx.x;
    for (int i = tid; i < rows * K; i += TK_THREADS) {
        const int r = i / K, c = i % K;
        const float v = x[r * ldx + c];
        const bf16 hi = (bf16)v;
        sh[r * PITCH + c] = hi;
        sl[r * PITCH + c] = (bf16)(v - (float)hi);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    // two column blocks of 16 per pass: 16 independent 1-KiB weight fragments in flight per wave (the loop is a chain of L2 / HBM round
    // trips, not of MFMAs)
    constexpr int NWAVES = TK_THREADS / 64;
    const bool live = l16 < rows;
    const int arow = live ? l16 : 0;
    for (int nb0 = wave; nb0 * 16 < N; nb0 += 2 * NWAVES) {
        const int n_a = nb0 * 16 + l16, n_b = (nb0 + NWAVES) * 16 + l16;
        const bf16* wpa = W.w + (long)(n_a < N ? n_a : N - 1) * K + 8 * kg;
        const bf16* wpb = W.w + (long)(n_b < N ? n_b : N - 1) * K + 8 * kg;
        f32x4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int ks = 0; ks < K / 32; ++ks) {
            const bf16x8 fa = *(const bf16x8*)(wpa + 32 * ks);
            const bf16x8 fb = *(const bf16x8*)(wpb + 32 * ks);
            // rows >= `rows` of the 16-row A operand are zero: read a valid row and select (no branch inside the loop -- at a
            // control-flow join hipcc waits for every weight load in flight)
            bf16x8 ah = *(const bf16x8*)(sh + arow * PITCH + 32 * ks + 8 * kg);
            bf16x8 al = *(const bf16x8*)(sl + arow * PITCH + 32 * ks + 8 * kg);
            ah = live ? ah : zero;
            al = live ? al : zero;
            acc_a = mfma16(ah, fa, acc_a);
            acc_a = mfma16(al, fa, acc_a);
            acc_b = mfma16(ah, fb, acc_b);
            acc_b = mfma16(al, fb, acc_b);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n = u == 0 ? n_a : n_b;
            const f32x4 acc = u == 0 ? acc_a : acc_b;
            if (n < N) {
                const float bias = W.b ? (float)W.b[n] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * kg + i;
                    if (r < rows) {
                        float v = acc[i] + bias;
                        if (act == WG_ACT_RELU) v = fmaxf(v, 0.f);
                        if (res) v += res[r * ldres + n];
                        y[r * ldy + n] = v;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// LayerNorm over the last dimension (256) of `rows` LDS rows, in place; one wave per row.
__device__ void tok_layernorm(float* x, int rows, NormW nw, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < rows; r += TK_THREADS / 64) {
        float v[4], s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = x[r * TK_C + lane + 64 * j];
            s += v[j];
        }
        const float mean = wg_wave_sum(s) * (1.0f / TK_C);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] -= mean;
            sq += v[j] * v[j];
        }
        const float rstd = 1.0f / sqrtf(wg_wave_sum(sq) * (1.0f / TK_C) + eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[r * TK_C + lane + 64 * j] = v[j] * rstd * (float)nw.g[lane + 64 * j] + (float)nw.b[lane + 64 * j];
    }
    __syncthreads();
}

// out[r][:] = a[r][:] + b[r][:]  (6 x 256)
__device__ void tok_add(float* out, const float* a, const float* b) {
    for (int i = threadIdx.x; i < TK_N * TK_C; i += TK_THREADS) out[i] = a[i] + b[i];
    __syncthreads();
}

// softmax(q k^T / sqrt(32)) v over the six tokens themselves: q, k, v [6][256] LDS fp32 (8 heads x 32) -> o [6][256]
__device__ void tok_self_attention(const float* q, const float* k, const float* v, float* o) {
    const int tid = threadIdx.x;
    if (tid < 8 * TK_N * 8) {               // (head, query, quarter of the head's 32 output dims)
        const int hq = tid >> 3, part = tid & 7;
        const int h = hq / TK_N, t = hq % TK_N;
        float s[TK_N], m = -1e30f;
#pragma unroll
        for (int j = 0; j < TK_N; ++j) {
            float acc = 0.f;
            for (int d = 0; d < 32; ++d) acc += q[t * TK_C + h * 32 + d] * k[j * TK_C + h * 32 + d];
            s[j] = acc * 0.17677669529663687f;   // 1 / sqrt(32)
            m = fmaxf(m, s[j]);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < TK_N; ++j) {
            s[j] = __expf(s[j] - m);
            l += s[j];
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < TK_N; ++j) acc += s[j] * v[j * TK_C + h * 32 + part * 4 + d];
            o[t * TK_C + h * 32 + part * 4 + d] = acc * inv;
        }
    }
    __syncthreads();
}

// token -> image attention: q [6][128] LDS fp32 (8 heads x 16, unscaled), K / V image rows [hw][ld] bf16 in global memory.
// wave = head; a lane owns keys lane, lane + 64, ...; online softmax per lane, merged across the wave at the end.  o [6][128] LDS fp32.
__device__ void tok_image_attention(float* q, const bf16* Kp, const bf16* Vp, long ld, int hw, float* o) {
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    // the six query rows are wave-uniform: they stay in LDS (scaled once; q is scratch of the caller) and are re-read per key as
    // broadcast 16-byte reads, which frees 96 registers for keeping the next pass's keys in flight
    for (int i = threadIdx.x; i < TK_N * 128; i += TK_THREADS) q[i] *= 0.25f;   // 1 / sqrt(16)
    __syncthreads();
    float m[TK_N], l[TK_N], acc[TK_N][16];
#pragma unroll
    for (int t = 0; t < TK_N; ++t) {
        m[t] = -1e30f;
        l[t] = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[t][d] = 0.f;
    }
    // the keys of the next pass are requested before the current pass is computed (the loop was a chain of strided-read round trips)
    auto ldk = [&](int j, bf16x8* kk, bf16x8* vv) __attribute__((always_inline)) {
        const int jj = j < hw ? j : hw - 1;
        kk[0] = *(const bf16x8*)(Kp + (long)jj * ld + h * 16); kk[1] = *(const bf16x8*)(Kp + (long)jj * ld + h * 16 + 8);
        vv[0] = *(const bf16x8*)(Vp + (long)jj * ld + h * 16); vv[1] = *(const bf16x8*)(Vp + (long)jj * ld + h * 16 + 8);
    };
    bf16x8 kn[2], vn[2];
    ldk(lane, kn, vn);
    for (int j = lane; j < hw; j += 64) {
        const bf16x8 k0 = kn[0], k1 = kn[1], v0 = vn[0], v1 = vn[1];
        ldk(j + 64, kn, vn);
        int qo = h * 16;
        asm volatile("" : "+v"(qo));   // keeps the (loop-invariant) query reads inside the loop: hoisted they are 96 registers again
        float kf[16], vf[16];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            kf[d] = (float)k0[d]; kf[8 + d] = (float)k1[d];
            vf[d] = (float)v0[d]; vf[8 + d] = (float)v1[d];
        }
#pragma unroll
        for (int t = 0; t < TK_N; ++t) {
            float s = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                const f32x4 qv = *(const f32x4*)(q + t * 128 + qo + 4 * d4);
                s += qv[0] * kf[4 * d4] + qv[1] * kf[4 * d4 + 1] + qv[2] * kf[4 * d4 + 2] + qv[3] * kf[4 * d4 + 3];
            }
            if (s > m[t]) {                       // rare after the first few keys: rescale the running state
                const float a = __expf(m[t] - s);
                l[t] *= a;
#pragma unroll
                for (int d = 0; d < 16; ++d) acc[t][d] *= a;
                m[t] = s;
            }
            const float p = __expf(s - m[t]);
            l[t] += p;
#pragma unroll
            for (int d = 0; d < 16; ++d) acc[t][d] += p * vf[d];
        }
    }
#pragma unroll
    for (int t = 0; t < TK_N; ++t) {
        const float M = wg_wave_max(m[t]);
        const float a = __expf(m[t] - M);
        const float L = wg_wave_sum(l[t] * a);
        const float inv = 1.0f / L;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const float s = wg_wave_sum(acc[t][d] * a);
            if (lane == d) o[t * 128 + h * 16 + d] = s * inv;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(TK_THREADS) void wg_dec_tokens_kernel(TokArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = (float*)smem;                 // [6][256] queries
    float* pes = qs + TK_N * TK_C;            // [6][256] query_pe
    float* t0 = pes + TK_N * TK_C;            // [6][256] scratch
    float* t1 = t0 + TK_N * TK_C;
    float* t2 = t1 + TK_N * TK_C;
    float* t3 = t2 + TK_N * TK_C;
    float* hid = t3 + TK_N * TK_C;            // [6][2048] MLP hidden
    bf16* sh = (bf16*)(hid + TK_N * TK_KMAX); // [8][2048 + 8] hi
    bf16* sl = sh + 8 * (TK_KMAX + 8);        // [8][2048 + 8] lo
    const int p = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < TK_N * TK_C; i += TK_THREADS) {
        qs[i] = a.queries[(long)p * TK_N * TK_C + i];
        pes[i] = a.pe[(long)p * TK_N * TK_C + i];
    }
    __syncthreads();
    const bf16* Kp = a.Kimg + (long)p * a.img_bs * a.ld_img;
    const bf16* Vp = a.Vimg + (long)p * a.img_bs * a.ld_img;

    if (a.mode == 0) {
        // ---- self attention (transformer.py:153-160): layer 0 replaces the queries and skips the positional term ---------------------
        if (a.skip_pe) {
            tok_linear<TK_C>(qs, TK_C, TK_N, a.self_attn.q, TK_C, t0, TK_C, 0, nullptr, 0, sh, sl);
            tok_linear<TK_C>(qs, TK_C, TK_N, a.self_attn.k, TK_C, t1, TK_C, 0, nullptr, 0, sh, sl);
        } else {
            tok_add(t3, qs, pes);
            tok_linear<TK_C>(t3, TK_C, TK_N, a.self_attn.q, TK_C, t0, TK_C, 0, nullptr, 0, sh, sl);
            tok_linear<TK_C>(t3, TK_C, TK_N, a.self_attn.k, TK_C, t1, TK_C, 0, nullptr, 0, sh, sl);
        }
        tok_linear<TK_C>(qs, TK_C, TK_N, a.self_attn.v, TK_C, t2, TK_C, 0, nullptr, 0, sh, sl);
        tok_self_attention(t0, t1, t2, t3);
        tok_linear<TK_C>(t3, TK_C, TK_N, a.self_attn.o, TK_C, qs, TK_C, 0, a.skip_pe ? nullptr : qs, TK_C, sh, sl);
        tok_layernorm(qs, TK_N, a.norm1, a.eps);
    }
    // ---- token -> image attention (:162-167; mode 1: :96-106), internal width 128 ---------------------------------------------------------
    tok_add(t3, qs, pes);
    tok_linear<TK_C>(t3, TK_C, TK_N, a.t2i_q, 128, t0, 128, 0, nullptr, 0, sh, sl);
    tok_image_attention(t0, Kp, Vp, a.ld_img, a.hw, t1);
    tok_linear<128>(t1, 128, TK_N, a.t2i_o, TK_C, qs, TK_C, 0, qs, TK_C, sh, sl);
    tok_layernorm(qs, TK_N, a.norm2, a.eps);
    if (a.mode == 0) {
        // ---- MLP (:169-172), norm3, then k / v of the image -> token attention (:174-178) ---------------------------------------------------
        tok_linear<TK_C>(qs, TK_C, TK_N, a.lin1, TK_KMAX, hid, TK_KMAX, WG_ACT_RELU, nullptr, 0, sh, sl);
        tok_linear<TK_KMAX>(hid, TK_KMAX, TK_N, a.lin2, TK_C, qs, TK_C, 0, qs, TK_C, sh, sl);
        tok_layernorm(qs, TK_N, a.norm3, a.eps);
        tok_add(t3, qs, pes);
        tok_linear<TK_C>(t3, TK_C, TK_N, a.i2t_k, 128, t0, 128, 0, nullptr, 0, sh, sl);
        tok_linear<TK_C>(qs, TK_C, TK_N, a.i2t_v, 128, t1, 128, 0, nullptr, 0, sh, sl);
        for (int i = tid; i < TK_N * 128; i += TK_THREADS) {
            a.k_i2t[(long)p * TK_N * 128 + i] = (bf16)t0[i];
            a.v_i2t[(long)p * TK_N * 128 + i] = (bf16)t1[i];
        }
    } else {
        // ---- hypernetwork MLPs on the mask tokens (rows 1..4), IoU head on row 0 (mask_decoder.py:146-160) -------------------------------
        for (int i = 0; i < 5; ++i) {
            const LinW* mlp = i < 4 ? a.hyper[i] : a.iou;
            const float* x = qs + (i < 4 ? 1 + i : 0) * TK_C;
            tok_linear<TK_C>(x, TK_C, 1, mlp[0], TK_C, t0, TK_C, WG_ACT_RELU, nullptr, 0, sh, sl);
            tok_linear<TK_C>(t0, TK_C, 1, mlp[1], TK_C, t1, TK_C, WG_ACT_RELU, nullptr, 0, sh, sl);
            tok_linear<TK_C>(t1, TK_C, 1, mlp[2], i < 4 ? 32 : 4, t2, TK_C, 0, nullptr, 0, sh, sl);
            if (i < 4) {
                if (tid < 32) a.hyper_out[((long)p * 4 + i) * 32 + tid] = t2[tid];
            } else if (tid < 4) {
                a.iou_out[(long)p * 4 + tid] = t2[tid];
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < TK_N * TK_C; i += TK_THREADS) a.queries[(long)p * TK_N * TK_C + i] = qs[i];
}

constexpr size_t TK_LDS = (size_t)(6 * TK_N * TK_C + TK_N * TK_KMAX) * 4 + (size_t)2 * 8 * (TK_KMAX + 8) * 2;

}  // namespace

// Flat pointer table of wg_dec_tokens_f32 (all bf16 device pointers, weight then bias / gamma then beta):
//   mode 0 (a TwoWayAttentionBlock, transformer.py:151-182), 26 entries:
//     self_attn q,k,v,out (8) | norm1 (2) | cross_attn_token_to_image q,out (4) | norm2 (2) | mlp lin1,lin2 (4) | norm3 (2) |
//     cross_attn_image_to_token k,v (4)
//   mode 1 (final_attn_token_to_image + norm_final_attn + output_hypernetworks_mlps + iou_prediction_head), 36 entries:
//     final attn q,out (4) | norm_final_attn (2) | 4 x 3 hypernetwork layers (24) | 3 IoU-head layers (6)
extern "C" int wg_dec_tokens_f32(int mode, int skip_pe, float* queries, const float* query_pe, const void* const* weights, int n_weights,
                                 const void* Kimg, const void* Vimg, long ld_img, long img_rows_per_prompt, int hw, void* k_i2t,
                                 void* v_i2t, float* hyper_out, float* iou_out, int P, float eps, void* stream) {
    WG_REQUIRE(queries && query_pe && weights && Kimg && Vimg, "dec_tokens: null operand");
    WG_REQUIRE(P > 0 && hw > 0 && ld_img % 8 == 0, "dec_tokens: bad shape");
    WG_REQUIRE((((uintptr_t)Kimg | (uintptr_t)Vimg) & 15) == 0, "dec_tokens: misaligned image projections");
    WG_REQUIRE((mode == 0 && n_weights == 26 && k_i2t && v_i2t) || (mode == 1 && n_weights == 36 && hyper_out && iou_out),
               "dec_tokens: mode %d needs %d weight pointers and its outputs", mode, mode == 0 ? 26 : 36);
    for (int i = 0; i < n_weights; ++i) WG_REQUIRE(weights[i] && ((uintptr_t)weights[i] & 15) == 0, "dec_tokens: weight %d null or misaligned", i);
    TokArgs a{};
    a.mode = mode; a.skip_pe = skip_pe; a.P = P; a.hw = hw; a.queries = queries; a.pe = query_pe; a.eps = eps;
    a.Kimg = (const bf16*)Kimg; a.Vimg = (const bf16*)Vimg; a.ld_img = ld_img; a.img_bs = img_rows_per_prompt;
    a.k_i2t = (bf16*)k_i2t; a.v_i2t = (bf16*)v_i2t; a.hyper_out = hyper_out; a.iou_out = iou_out;
    int c = 0;
    auto lin = [&]() { LinW l{(const bf16*)weights[c], (const bf16*)weights[c + 1]}; c += 2; return l; };
    auto nrm = [&]() { NormW n{(const bf16*)weights[c], (const bf16*)weights[c + 1]}; c += 2; return n; };
    if (mode == 0) {
        a.self_attn.q = lin(); a.self_attn.k = lin(); a.self_attn.v = lin(); a.self_attn.o = lin(); a.norm1 = nrm();
        a.t2i_q = lin(); a.t2i_o = lin(); a.norm2 = nrm();
        a.lin1 = lin(); a.lin2 = lin(); a.norm3 = nrm();
        a.i2t_k = lin(); a.i2t_v = lin();
    } else {
        a.t2i_q = lin(); a.t2i_o = lin(); a.norm2 = nrm();
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) a.hyper[i][j] = lin();
        for (int j = 0; j < 3; ++j) a.iou[j] = lin();
    }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)wg_dec_tokens_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(wg_dec_tokens_kernel, dim3(P), dim3(TK_THREADS), TK_LDS, (hipStream_t)stream, a);
    return wg_check_launch("wg_dec_tokens");
}
