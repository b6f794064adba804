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
    const bf16* w1; const bf16* b1;    // ConvT 1 as a GEMM: [(dy,dx,64), 256] in fragment order (wg_tile_weight_bf16), bias [64]
    const bf16* g1; const bf16* be1;   // LayerNorm2d(64)
    const bf16* w2; const bf16* b2;    // ConvT 2 as a GEMM: [(dy,dx,32), 64], bias [32]
    const float* hyper;                // [P, nmask_total, 32]
    float* out;                        // [P, num_masks, 4h, 4w]
    int P, h, w, nmask_total, first_mask, num_masks;
    float eps;
};

constexpr int UP_ROWS = 16;            // token rows per step of a workgroup
constexpr int UP_PITCH = 72;           // bf16 elements per LDS row of a 64-channel operand (144 B: conflict-free b128 reads)

// Work split: the four waves of a workgroup own the four sub-pixels (dy, dx) of the first transposed convolution.  A wave keeps ITS 64
// rows of w1 (the 64 channels of its sub-pixel, 32 KB) in registers for the whole kernel and the workgroup walks 16-token-row groups:
// LayerNorm2d runs over exactly a wave's 64 channels, the second convolution reads only them, so from the x rows to the logits nothing
// crosses waves.  (First version: a wave per 16 rows doing all four sub-pixels -- every wave re-streamed the 128 KB of w1 from L2, 512 KB
// per workgroup through one compute unit: 80 us at P = 8.)
__global__ __launch_bounds__(256) void wg_upscale_mask_kernel(UpArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 vhi[4][16 * UP_PITCH];
    __shared__ __attribute__((aligned(16))) bf16 vlo[4][16 * UP_PITCH];
    __shared__ __attribute__((aligned(16))) bf16 w2s[128 * UP_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const long hw = (long)a.h * a.w;
    const long groups = (long)a.P * hw / UP_ROWS;        // hw % 16 == 0 is checked by the host: a group never straddles two prompts
    // w2 [(dy2,dx2,32), 64] -> LDS (16 KB, shared by the four waves)
    for (int i = threadIdx.x; i < 128 * 8; i += 256) *(bf16x8*)(w2s + (i >> 3) * UP_PITCH + 8 * (i & 7)) = *(const bf16x8*)(a.w2 + (long)i * 8);
    // this wave's rows of w1 as B fragments: column block nb (16 channels) x k step ks
    bf16x8 wf[4][8];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wf[nb][ks] = *(const bf16x8*)(a.w1 + ((long)((wave * 4 + nb) * 8 + ks) * 64 + lane) * 8);   // fragment order
    float b1v[4], g1v[4], be1v[4], b2v[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b1v[j] = (float)a.b1[16 * j + l16];
        g1v[j] = (float)a.g1[16 * j + l16];
        be1v[j] = (float)a.be1[16 * j + l16];
    }
    b2v[0] = (float)a.b2[l16];
    b2v[1] = (float)a.b2[16 + l16];
    __syncthreads();
    const int H4 = 4 * a.h, W4 = 4 * a.w;
    bf16x8 xa[8];
    long grp = blockIdx.x;
    if (grp < groups) {
        const bf16* xp = a.x + (grp * UP_ROWS + l16) * a.ldx + 8 * kg;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xa[ks] = *(const bf16x8*)(xp + 32 * ks);
    }
    for (; grp < groups; grp += gridDim.x) {
        // ---- GEMM 1: u[16 rows][64] = x[16][256] . w1[sub-pixel `wave`]^T ---------------------------------------------------------------
        f32x4 u[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = mfma16(xa[ks], wf[nb][ks], acc);
            u[nb] = acc;
        }
        if (grp + gridDim.x < groups) {       // the next group's rows travel under the rest of this one
            const bf16* xp = a.x + ((grp + gridDim.x) * UP_ROWS + l16) * a.ldx + 8 * kg;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) xa[ks] = *(const bf16x8*)(xp + 32 * ks);
        }
        // ---- bias, LayerNorm2d over the 64 channels, GELU -> the wave's LDS slab as a bf16 hi + lo pair (A operand of GEMM 2) ----------------
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[4], sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = u[j][i] + b1v[j];
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
#pragma unroll
            for (int j = 0; j < 4; j += 2) {      // (pairs: hipcc selects the packed fp32 forms, wg_act2 uses the bare v_exp / v_rcp)
                const f32x2 y = wg_act2<WG_ACT_GELU_ERF>((f32x2){v[j] * rstd * g1v[j] + be1v[j], v[j + 1] * rstd * g1v[j + 1] + be1v[j + 1]});
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const bf16 hi = (bf16)y[e];
                    vhi[wave][(4 * kg + i) * UP_PITCH + 16 * (j + e) + l16] = hi;
                    vlo[wave][(4 * kg + i) * UP_PITCH + 16 * (j + e) + l16] = (bf16)(y[e] - (float)hi);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();      // the slab is private to the wave and LDS operations of one wave complete in order
        // ---- GEMM 2: z[16 rows][128] = v[16][64] . w2^T (hi + lo), bias, GELU ----------------------------------------------------------
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ah[ks] = *(const bf16x8*)(&vhi[wave][l16 * UP_PITCH + 32 * ks + 8 * kg]);
            al[ks] = *(const bf16x8*)(&vlo[wave][l16 * UP_PITCH + 32 * ks + 8 * kg]);
        }
        f32x4 gz[8];                          // gelu(z): column block nb = (sub-sub-pixel nb >> 1, channel half nb & 1)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 wb = *(const bf16x8*)(w2s + (nb * 16 + l16) * UP_PITCH + 32 * ks + 8 * kg);
                acc = mfma16(ah[ks], wb, acc);
                acc = mfma16(al[ks], wb, acc);
            }
            const f32x2 g01 = wg_act2<WG_ACT_GELU_ERF>((f32x2){acc[0] + b2v[nb & 1], acc[1] + b2v[nb & 1]});
            const f32x2 g23 = wg_act2<WG_ACT_GELU_ERF>((f32x2){acc[2] + b2v[nb & 1], acc[3] + b2v[nb & 1]});
            gz[nb] = (f32x4){g01.x, g01.y, g23.x, g23.y};
        }
        __builtin_amdgcn_wave_barrier();      // (the slab is rewritten by the next group)
        // ---- <hyper_in, .> over the 32 channels: 16 logits per token, this wave's four ------------------------------------------------------
        const long r0 = grp * UP_ROWS;
        const long p = r0 / hw;
        const int t0 = (int)(r0 - p * hw);
        const int t = t0 + 4 * kg + (l16 & 3);          // lane (kg, l16 = 4*ss + i) stores the logit of token row 4*kg + i, sub-sub-pixel ss
        const int ty = t / a.w, tx = t % a.w;
        const int ss_l = l16 >> 2;
        const long opix = (long)(4 * ty + 2 * (wave >> 1) + (ss_l >> 1)) * W4 + 4 * tx + 2 * (wave & 1) + (ss_l & 1);
        for (int mk = 0; mk < a.num_masks; ++mk) {
            const float* hy = a.hyper + ((long)p * a.nmask_total + a.first_mask + mk) * 32;
            const float hy0 = hy[l16], hy1 = hy[16 + l16];
            // the 16 (sub-sub-pixel, token row) sums of a mask: all of them through each DPP step together (independent neighbours: no wait states)
            float part[16];
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)
#pragma unroll
                for (int i = 0; i < 4; ++i) part[4 * ss + i] = gz[2 * ss][i] * hy0 + gz[2 * ss + 1][i] * hy1;
#pragma unroll
            for (int e = 0; e < 16; ++e) part[e] += WG_DPP(part[e], 0xB1);
#pragma unroll
            for (int e = 0; e < 16; ++e) part[e] += WG_DPP(part[e], 0x4E);
#pragma unroll
            for (int e = 0; e < 16; ++e) part[e] += WG_DPP(part[e], 0x124);
#pragma unroll
            for (int e = 0; e < 16; ++e) part[e] += WG_DPP(part[e], 0x128);
            float mine = part[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mine = (l16 == e) ? part[e] : mine;
            a.out[((long)p * a.num_masks + mk) * H4 * W4 + opix] = mine;
        }
    }
}

// =====================================================================================================================================
// Image side of a TwoWayAttentionBlock after the token kernel (transformer.py:173-180): every image token attends to the SIX prompt
// tokens, goes through out_proj, adds itself and is normalised:
//     keys = norm4(keys + out_proj(softmax(q k^T / 4) v))          q = the [.., 2d:3d] columns of the fused image-side projection
// One wave per 16 image tokens, nothing leaves the wave: lane (l16 = token, kg) computes the scores of heads kg>>1, 2+(kg>>1), ... (three
// keys each for the two lanes that share a head, swapped with one permlane) and
// the half (kg & 1) of each head's 16 outputs -- exactly the B fragment (k = 32 ks + 8 kg .. +7) of the transposed product
// out^T[256][16 tokens] = Wo[256][128] . o^T, whose accumulators hold four CONSECUTIVE channels of one token per lane: residual, LayerNorm
// statistics (in-lane + two cross-lane steps) and the bf16 store all run on 8-byte pieces.  Wo sits in LDS (shared by the four waves),
// o is fed as a bf16 hi + lo pair.  Replaces three launches (attention with 6 keys, GEMM + residual, LayerNorm).
// =====================================================================================================================================
struct I2tArgs {
    const bf16* q; long ldq;            // [rows or hw][ldq]: projected image tokens (128 columns used)
    const bf16* kq; const bf16* vq;     // [P, 6, 128]: projected prompt tokens
    const bf16* wo; const bf16* bo;     // out_proj [256, 128], [256]
    const bf16* res; long ldr;          // the image tokens themselves [rows or hw][ldr]
    const bf16* res_bias;               // [256] or null: constant row added to res
    int row_mod;                        // hw when q / res are shared by all prompts (one image), 0 otherwise
    const int* pimg;                    // or: prompt p's q / res rows are those of image pimg[p] (hw rows per image)
    const bf16* g; const bf16* b; float eps;
    bf16* out;                          // [P * hw, 256]
    long rows; int hw;
};
constexpr int I2_KEYS = 6;             // prompt tokens per query: iou + 4 mask tokens + the text prompt
constexpr int I2_PITCH = 136;           // bf16 per LDS row of Wo (272 B: conflict-free b128 reads)
constexpr int I2_WAVES = 8;             // two per SIMD: the LDS round trips of one hide under the arithmetic of the other
constexpr int I2_LDS = 256 * I2_PITCH * 2 + 3 * 256 * 4 + I2_WAVES * 2 * I2_KEYS * 128 * 4;

__device__ __forceinline__ float kg_sum(float v) {      // sum over the four lanes that share l16
    float x, y;
    wg_permlane_swap<0>(v, x, y); v = x + y;
    wg_permlane_swap<1>(v, x, y); return x + y;
}

__global__ __launch_bounds__(64 * I2_WAVES) void wg_dec_i2t_rows_kernel(I2tArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char i2_smem[];
    bf16* wos = (bf16*)i2_smem;                                   // [256][I2_PITCH]
    float* vec = (float*)(i2_smem + 256 * I2_PITCH * 2);           // bias | gamma | beta, fp32 [3][256]
    float* kvs = vec + 3 * 256;                                    // per wave: k [6][128] | v [6][128], fp32
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    for (int i = threadIdx.x; i < 256 * 16; i += 64 * I2_WAVES) *(bf16x8*)(wos + (i >> 4) * I2_PITCH + 8 * (i & 15)) = *(const bf16x8*)(a.wo + (long)i * 8);
    if (threadIdx.x < 256) {
        const int i = threadIdx.x;
        vec[i] = (float)a.bo[i] + (a.res_bias ? (float)a.res_bias[i] : 0.f);
        vec[256 + i] = (float)a.g[i];
        vec[512 + i] = (float)a.b[i];
    }
    __syncthreads();
    float* ks_ = kvs + wave * 2 * I2_KEYS * 128;
    float* vs_ = ks_ + I2_KEYS * 128;
    const long groups = a.rows / 16;
    for (long grp = (long)blockIdx.x * I2_WAVES + wave; grp < groups; grp += (long)gridDim.x * I2_WAVES) {
        const long r0 = grp * 16;
        const long p = r0 / a.hw;                                  // hw % 16 == 0: a group never straddles two prompts
        const long row = r0 + l16;
        const long srow = a.pimg ? (long)a.pimg[p] * a.hw + (row - p * a.hw) : (a.row_mod > 0 ? row % a.row_mod : row);   // row of q / res
        // prompt tokens of this prompt -> the wave's slab
        for (int i = lane; i < 2 * I2_KEYS * 16; i += 64) {
            const bf16x8 t = *(const bf16x8*)((i < I2_KEYS * 16 ? a.kq : a.vq) + p * I2_KEYS * 128 + (i % (I2_KEYS * 16)) * 8);
            *(f32x4*)(ks_ + i * 8) = (f32x4){(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
            *(f32x4*)(ks_ + i * 8 + 4) = (f32x4){(float)t[4], (float)t[5], (float)t[6], (float)t[7]};
        }
        bf16x8 qv[4][2];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            const bf16* qp = a.q + srow * a.ldq + (2 * hh + (kg >> 1)) * 16;
            qv[hh][0] = *(const bf16x8*)qp;
            qv[hh][1] = *(const bf16x8*)(qp + 8);
        }
        bf16x4 rv[16];                                             // residual: channels nb*16 + 4*kg .. +3 of token l16
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) rv[nb] = *(const bf16x4*)(a.res + srow * a.ldr + nb * 16 + 4 * kg);
        __builtin_amdgcn_wave_barrier();
        // ---- attention over the six prompt tokens: head 2*hh + (kg >> 1), output dims 8*(kg & 1) .. +7 -----------------------------------
        bf16x8 oh[4], ol[4];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            const int hoff = (2 * hh + (kg >> 1)) * 16;
            float qf[16];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                qf[d] = (float)qv[hh][0][d] * 0.25f;               // 1 / sqrt(16)
                qf[8 + d] = (float)qv[hh][1][d] * 0.25f;
            }
            // the two lanes that share (token, head) -- kg even / odd, lane ^ 16 -- take three keys each and swap the scores:
            // v_permlane16_swap leaves the even row's value in its first operand and the odd row's in the second, on BOTH lanes
            float sc[I2_KEYS];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float* kp = ks_ + (3 * (kg & 1) + t) * 128 + hoff;
                float acc = 0.f;
#pragma unroll
                for (int d4 = 0; d4 < 4; ++d4) {
                    const f32x4 kk = *(const f32x4*)(kp + 4 * d4);
                    acc += qf[4 * d4] * kk[0] + qf[4 * d4 + 1] * kk[1] + qf[4 * d4 + 2] * kk[2] + qf[4 * d4 + 3] * kk[3];
                }
                wg_permlane_swap<0>(acc, sc[t], sc[3 + t]);
            }
            float m = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(sc[4], sc[5]));
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < I2_KEYS; ++j) {
                sc[j] = __expf(sc[j] - m);
                l += sc[j];
            }
            const float inv = 1.0f / l;
            float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < I2_KEYS; ++j) {
                const float* vp = vs_ + j * 128 + hoff + 8 * (kg & 1);
                const f32x4 v0 = *(const f32x4*)vp, v1 = *(const f32x4*)(vp + 4);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    o[d] += sc[j] * v0[d];
                    o[4 + d] += sc[j] * v1[d];
                }
            }
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const float y = o[d] * inv;
                const bf16 hi = (bf16)y;
                oh[hh][d] = hi;
                ol[hh][d] = (bf16)(y - (float)hi);
            }
        }
        __builtin_amdgcn_wave_barrier();                           // (the slab is rewritten by the next group)
        // ---- out^T = Wo . o^T, + bias + residual; LayerNorm over the 256 channels of each token ------------------------------------------
        f32x4 acc[16];
        float sum = 0.f;
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 wf = *(const bf16x8*)(wos + (nb * 16 + l16) * I2_PITCH + 32 * ks + 8 * kg);
                c = mfma16(wf, oh[ks], c);
                c = mfma16(wf, ol[ks], c);
            }
            const f32x4 bias = *(const f32x4*)(vec + nb * 16 + 4 * kg);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c[i] += bias[i] + (float)rv[nb][i];
                sum += c[i];
            }
            acc[nb] = c;
        }
        const float mean = kg_sum(sum) * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int nb = 0; nb < 16; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[nb][i] -= mean;
                sq += acc[nb][i] * acc[nb][i];
            }
        const float rstd = 1.0f / sqrtf(kg_sum(sq) * (1.0f / 256.0f) + a.eps);
        bf16* op = a.out + row * 256 + 4 * kg;
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) {
            const f32x4 gm = *(const f32x4*)(vec + 256 + nb * 16 + 4 * kg), bt = *(const f32x4*)(vec + 512 + nb * 16 + 4 * kg);
            bf16x4 y;
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = (bf16)(acc[nb][i] * rstd * gm[i] + bt[i]);
            *(bf16x4*)(op + nb * 16) = y;
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
    const long groups = rows / UP_ROWS;
    hipLaunchKernelGGL(wg_upscale_mask_kernel, dim3((unsigned)(groups < 512 ? groups : 512)), dim3(256), 0, (hipStream_t)stream, a);
    return wg_check_launch("wg_upscale_mask");
}

// =====================================================================================================================================
// Token side of SAM's two-way transformer (transformer.py:62-182) and the decoder heads (mask_decoder.py:146-160).
//
// The six tokens of a prompt are tiny (6 x 256 fp32) but every Linear that touches them streams its weights: 2.8 MB per block, and ONE
// compute unit ingests ~30 GB/s with plain loads (in-kernel stamps of the one-kernel-per-block version: 183 us = 83 us walking the 4096
// image keys + 63 us streaming the MLP's 2 MB + 37 us for everything else).  So the two heavy pieces run as kernels of their own, spread over
// the chip, and hand small fp32 partials across a kernel boundary (the cheapest grid-wide synchronisation there is):
//
//   wg_dec_tokens_f32        one workgroup per prompt, the tokens in LDS as fp32; executes the stages named by a bit mask:
//                              SUM_MLP  x = x + lin2 bias + sum of the MLP partials; norm3; k / v of the image->token attention
//                              SELF     self attention of the six tokens (+ query_pe unless skip_pe), norm1
//                              Q_T2I    q of the token->image attention -> q_t2i [P,6,128]
//                              COMBINE  merge the attention partials; out_proj + residual; norm2 (or norm_final_attn)
//                              INIT     (with the first SELF) build the tokens from the output-token table and the prompt rows
//   wg_dec_attn_partial_f32  token->image attention: grid (prompt, head, split of 1024 keys), a wave per 256 keys, online softmax per lane
//                            merged across the wave, then across the four waves in LDS -> {m[6], l[6], o[6][16]} per (prompt, head, split)
//   wg_dec_mlp_partial_f32   the MLP (256 -> 2048 -> ReLU -> 256): grid (prompt, 8 slices of 256 hidden units) -> partial [6][256] each
//   wg_dec_heads_f32         the four hypernetwork MLPs and the IoU head: grid (prompt, 5), one 3-layer MLP each
//
// Linear layers run on the matrix pipe with the tokens as a 16-row A operand (rows 6..15 zero) split into a bf16 hi + lo pair (~16
// significant bits); weights are bf16 (the checkpoint's precision), streamed as B fragments.
//
// WEIGHT LAYOUT.  Every weight matrix these kernels read is pre-tiled in FRAGMENT ORDER (wg_tile_weight_bf16, once per checkpoint):
//     T[nb][ks][lane][j] = W[16 nb + (lane & 15)][32 ks + 8 (lane >> 4) + j],     rows >= N zero,
// i.e. the 1 KiB a wave loads for MFMA step ks of column block nb is contiguous.  With the row-major [N][K] matrix the same instruction
// touches 16 rows x 64 B -- 16 half-used cache lines -- and ONE compute unit pulled 33 GB/s through that pattern even when the matrix was
// L2-resident, against 96 GB/s for contiguous 1-KiB pieces (tools/micro/cu_ingest.hip; 29 vs 36 GB/s when cold): these kernels are exactly
// that streaming, a workgroup per prompt walking 0.2 - 0.7 MB of weights.
// =====================================================================================================================================
namespace {

// 1024 threads: the token kernels are chains of short phases on ONE compute unit (stage, weight pass, LDS arithmetic, LayerNorm); their weight passes
// keep the same bytes in flight at 8 or 16 waves, but everything between the passes is per-thread serial work, and 16 waves halve it.  Round 6,
// one-image decode 198.5 -> 185.3 us, B = 8: 286 -> 272, T = 14: 356 -> 343, B = 8 at T = 14 unchanged (notes/r06_experiments.md section 14).
#ifndef WG_DEC_THREADS
#define WG_DEC_THREADS 1024
#endif
constexpr int TK_THREADS = WG_DEC_THREADS;
constexpr int TK_N = 6;          // tokens per prompt: iou + 4 mask tokens + 1 text prompt (mask_decoder.py:125-132)
constexpr int TK_C = 256;        // transformer_dim
constexpr int TK_HID = 2048;     // MLP width
#ifndef WG_DEC_MLP_SLICES
#define WG_DEC_MLP_SLICES 16
#endif
constexpr int TK_SLICES = WG_DEC_MLP_SLICES;     // MLP slices (2048 / TK_SLICES hidden units each: 8 or 16)
constexpr int TK_SLW = 2048 / TK_SLICES;         // hidden units per slice
constexpr int TK_PART = TK_N * 18;   // floats of one attention partial: m[6] | l[6] | o[6][16]

struct LinW { const bf16* w; const bf16* b; };
struct NormW { const bf16* g; const bf16* b; };
struct AttnW { LinW q, k, v, o; };

enum { ST_SUM_MLP = 1, ST_SELF = 2, ST_Q_T2I = 4, ST_COMBINE = 8, ST_INIT = 16 };

struct TokArgs {
    int stages, skip_pe, P, n_splits;
    float* queries;            // [P, 6, 256] fp32 in / out
    float* pe;                 // [P, 6, 256] fp32: the prompt tokens as they entered the transformer (query_pe); written by INIT
    const float* init_tokens;  // INIT: [5, 256] fp32 = cat(iou_token.weight, mask_tokens.weight)   (mask_decoder.py:125-128)
    const bf16* init_prompt;   //       [P, 256] bf16 = the sparse prompt embedding of each query  (:129-132)
    const bf16* tail_g; const bf16* tail_b; const bf16* tail_tt; const bf16* tail_lt; float tail_eps;   // INIT, optional: init_prompt holds the rows
                               //       BEFORE the text projector's tail (utils_walkgpt.py:324-327), which this launch applies (LayerNorm gamma, beta,
                               //       text_type, log_temp): the [SEG] embedding is rounded to bf16 exactly as wg_ctp_tail_bf16 leaves it
    AttnW self_attn; NormW norm1;
    LinW t2i_q, t2i_o; NormW norm2;          // (tail: final_attn_token_to_image q / out, norm_final_attn)
    const bf16* lin2_b; NormW norm3;         // SUM_MLP: bias of mlp.lin2, norm3
    LinW i2t_k, i2t_v;
    float* q_t2i;                            // [P, 6, 128] fp32: out of Q_T2I
    const float* attn_part;                  // [P, 8, n_splits, TK_PART]: in of COMBINE
    const float* mlp_part;                   // [P, TK_SLICES, 6, 256]: in of SUM_MLP
    bf16* k_i2t; bf16* v_i2t;                // out of SUM_MLP: [P, 6, 128] bf16 keys / values of the image->token attention
    float eps;
};

// ---- Linear layers on the six token rows ---------------------------------------------------------------------------------------------------
// One step of the chain = stage the inputs as bf16 hi + lo pairs in LDS (tok_stage), barrier, then ONE pass (tok_mm) over the column blocks
// of every Linear that reads those inputs -- q, k and v of an attention are one pass, not three: the chain is a sequence of L2 / HBM round
// trips (stage -> weights -> write), so what counts is the number of passes, not their width.
struct TokJob {
    const bf16* sh; const bf16* sl;      // staged input rows [8][K + 8] (hi, lo)
    LinW W; int N;                       // y[r][n] = act(x[r][:] . W[n][:] + b[n]) (+ res[r][n]),  W in fragment order (below)
    float* y; int ldy; int act;
    const float* res; int ldres;         // LDS fp32
};

template <int K, int THREADS = TK_THREADS>
__device__ __forceinline__ void tok_stage(const float* x, int ldx, const float* add, int rows, bf16* sh, bf16* sl) {
    constexpr int PITCH = K + 8;
    for (int i = threadIdx.x; i < rows * K; i += THREADS) {
        const int r = i / K, c = i % K;
        const float v = x[r * ldx + c] + (add ? add[r * ldx + c] : 0.f);
        const bf16 hi = (bf16)v;
        sh[r * PITCH + c] = hi;
        sl[r * PITCH + c] = (bf16)(v - (float)hi);
    }
}

// The column blocks (16 outputs) of all NJ jobs form one list; a wave takes blocks wave, wave + NWAVES, ... two at a time, so 16 independent
// 1-KiB weight fragments are in flight per wave.  Rows >= `rows` of the 16-row A operand are zero: a valid row is read and a select applied
// (no branch inside the loop -- at a control-flow join hipcc waits for every weight load in flight).  Ends with a barrier.
template <int K, int NJ, int THREADS = TK_THREADS>
__device__ void tok_mm(const TokJob (&jobs)[NJ], int rows) {
    constexpr int PITCH = K + 8;
    constexpr int NWAVES = THREADS / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // scalar: block -> job selection below stays in SGPRs
    const int l16 = lane & 15, kg = lane >> 4;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool live = l16 < rows;
    const int arow = live ? l16 : 0;
    int total = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) total += (jobs[j].N + 15) / 16;
    for (int b0 = wave; b0 < total; b0 += 2 * NWAVES) {
        // block -> (job, column block) by a chain of selects over the statically indexed job table (a run-time index into it would put
        // the table in scratch memory); a block past the end repeats the pass's first one and is not written
        TokJob jsel[2];
        int nb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int b = b0 + u * NWAVES;
            if (b >= total) b = b0;
            jsel[u] = jobs[0];
            bool go = true;
#pragma unroll
            for (int q = 1; q < NJ; ++q) {
                const int nblk = (jobs[q - 1].N + 15) / 16;
                go = go && b >= nblk;
                if (go) { b -= nblk; jsel[u] = jobs[q]; }
            }
            nb[u] = b;
        }
        const TokJob& ja = jsel[0];
        const TokJob& jq = jsel[1];
        const int n_a = nb[0] * 16 + l16, n_b = nb[1] * 16 + l16;
        const bf16* wpa = ja.W.w + (long)nb[0] * (K / 32) * 512 + lane * 8;      // fragment (nb, ks) = 1 KiB at ((nb * K/32 + ks) * 64 + lane) * 8
        const bf16* wpb = jq.W.w + (long)nb[1] * (K / 32) * 512 + lane * 8;
        const bf16* aha = ja.sh + arow * PITCH + 8 * kg, *ala = ja.sl + arow * PITCH + 8 * kg;
        const bf16* ahb = jq.sh + arow * PITCH + 8 * kg, *alb = jq.sl + arow * PITCH + 8 * kg;
        // The columns' bias values travel WITH the weight fragments: requested in front of them, no branch (a missing bias reads a weight word and
        // is dropped by a select), and they are what the accumulators start from -- so the first MFMA's wait covers them.  Added behind the loop,
        // hipcc sank the load to its use: one more dependent trip to L2 in every pass of every token kernel (round 6).
        const bf16 bra = *((ja.W.b ? ja.W.b : ja.W.w) + (n_a < ja.N ? n_a : 0)), brb = *((jq.W.b ? jq.W.b : jq.W.w) + (n_b < jq.N ? n_b : 0));
        const float bias_a = ja.W.b ? (float)bra : 0.f, bias_b = jq.W.b ? (float)brb : 0.f;
        f32x4 acc_a = {bias_a, bias_a, bias_a, bias_a}, acc_b = {bias_b, bias_b, bias_b, bias_b};
#pragma unroll 8
        for (int ks = 0; ks < K / 32; ++ks) {
            const bf16x8 fa = *(const bf16x8*)(wpa + 512 * ks);
            const bf16x8 fb = *(const bf16x8*)(wpb + 512 * ks);
            bf16x8 h0 = *(const bf16x8*)(aha + 32 * ks), l0 = *(const bf16x8*)(ala + 32 * ks);
            bf16x8 h1 = *(const bf16x8*)(ahb + 32 * ks), l1 = *(const bf16x8*)(alb + 32 * ks);
            h0 = live ? h0 : zero; l0 = live ? l0 : zero;
            h1 = live ? h1 : zero; l1 = live ? l1 : zero;
            acc_a = mfma16(h0, fa, acc_a);
            acc_a = mfma16(l0, fa, acc_a);
            acc_b = mfma16(h1, fb, acc_b);
            acc_b = mfma16(l1, fb, acc_b);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && b0 + NWAVES >= total) break;
            const TokJob& jj = u == 0 ? ja : jq;
            const int n = u == 0 ? n_a : n_b;
            const f32x4 acc = u == 0 ? acc_a : acc_b;
            if (n < jj.N) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * kg + i;
                    if (r < rows) {
                        float v = acc[i];
                        if (jj.act == WG_ACT_RELU) v = fmaxf(v, 0.f);
                        if (jj.res) v += jj.res[r * jj.ldres + n];
                        jj.y[r * jj.ldy + n] = v;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// one Linear: y = act(x W^T + b) (+ res); x, y, res in LDS (fp32), every thread of the workgroup calls it, visible to all on return
template <int K, int THREADS = TK_THREADS>
__device__ void tok_linear(const float* x, int ldx, int rows, LinW W, int N, float* y, int ldy, int act, const float* res, int ldres,
                           bf16* sh, bf16* sl) {
    tok_stage<K, THREADS>(x, ldx, nullptr, rows, sh, sl);
    __syncthreads();
    const TokJob job[1] = {{sh, sl, W, N, y, ldy, act, res, ldres}};
    tok_mm<K, 1, THREADS>(job, rows);
}

// LayerNorm over the last dimension (256) of `rows` LDS rows, in place; one wave per row.  Its gamma / beta come from memory: tok_ln_load requests
// them EARLY -- in front of the Linear whose output the LayerNorm reads (a barrier inside keeps the loads there) -- so that the trip to L2 passes
// under that Linear's weight pass instead of standing alone behind it (round 6: every token kernel is a chain of such trips).
struct LnRegs { float g[4], b[4]; };
__device__ __forceinline__ LnRegs tok_ln_load(NormW nw) {
    const int lane = threadIdx.x & 63;
    LnRegs r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.g[j] = (float)nw.g[lane + 64 * j];
        r.b[j] = (float)nw.b[lane + 64 * j];
    }
    return r;
}
template <int THREADS = TK_THREADS>
__device__ void tok_layernorm(float* x, int rows, const LnRegs& p, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < rows; r += THREADS / 64) {
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
        for (int j = 0; j < 4; ++j) x[r * TK_C + lane + 64 * j] = v[j] * rstd * p.g[j] + p.b[j];
    }
    __syncthreads();
}
template <int THREADS = TK_THREADS>
__device__ void tok_layernorm(float* x, int rows, NormW nw, float eps) {
    tok_layernorm<THREADS>(x, rows, tok_ln_load(nw), eps);
}

// softmax(q k^T / sqrt(32)) v over the six tokens themselves: q, k, v [6][256] LDS fp32 (8 heads x 32) -> o [6][256]
__device__ void tok_self_attention(const float* q, const float* k, const float* v, float* o) {
    const int tid = threadIdx.x;
    if (tid < 8 * TK_N * 8) {               // (head, query, eighth of the head's 32 output dims)
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

// COMBINE for token rows row0 .. row0 + rows - 1 of prompt p: merge the key splits of every (head, row),
//     o = sum_s o_s e^(m_s - M) / sum_s l_s e^(m_s - M),
// then x = LayerNorm(x + out_proj(o)).  x: LDS [rows][256] (residual in, result out); ao: LDS scratch [rows][128].
struct CombineW { const float* part; int n_splits; LinW wo; NormW nw; float eps; };
template <int THREADS = TK_THREADS>
__device__ void tok_combine(const CombineW& c, long p, int row0, int rows, float* x, float* ao, bf16* sh, bf16* sl) {
    for (int i = threadIdx.x; i < 8 * rows * 16; i += THREADS) {
        const int d = i & 15, r = (i >> 4) % rows, h = i / (16 * rows), t = row0 + r;
        const float* pp = c.part + (p * 8 + h) * c.n_splits * TK_PART;
        float M = -1e30f, L = 0.f, o = 0.f;
        for (int s0 = 0; s0 < c.n_splits; s0 += 4) {        // four splits' loads in flight at a time
            float mm[4], ll[4], oo[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = s0 + u < c.n_splits;
                const float* q_ = pp + (in ? s0 + u : s0) * TK_PART;
                mm[u] = in ? q_[t] : -1e30f;
                ll[u] = in ? q_[TK_N + t] : 0.f;
                oo[u] = in ? q_[2 * TK_N + t * 16 + d] : 0.f;
            }
            const float Mn = fmaxf(fmaxf(fmaxf(mm[0], mm[1]), fmaxf(mm[2], mm[3])), M);
            const float sc = __expf(M - Mn);
            L *= sc;
            o *= sc;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float w = __expf(mm[u] - Mn);
                L += ll[u] * w;
                o += oo[u] * w;
            }
            M = Mn;
        }
        ao[r * 128 + h * 16 + d] = o / L;
    }
    __syncthreads();
    const LnRegs ln = tok_ln_load(c.nw);
    tok_linear<128, THREADS>(ao, 128, rows, c.wo, TK_C, x, TK_C, 0, x, TK_C, sh, sl);
    tok_layernorm<THREADS>(x, rows, ln, c.eps);
}

__global__ __launch_bounds__(TK_THREADS) void wg_dec_tokens_kernel(TokArgs a) {
    __shared__ float qs[TK_N * TK_C], pes[TK_N * TK_C], t0[TK_N * TK_C], t1[TK_N * TK_C], t2[TK_N * TK_C], t3[TK_N * TK_C];
    __shared__ __attribute__((aligned(16))) bf16 sh[2][8 * (TK_C + 8)], sl[2][8 * (TK_C + 8)];
    const int p = blockIdx.x;
    const int tid = threadIdx.x;
    const bool tail = (a.stages & ST_INIT) && a.tail_g;
    LnRegs ln3 = {};
    if (a.stages & ST_SUM_MLP) ln3 = tok_ln_load(a.norm3);      // (with the loads of the token rows and the MLP partials below, not behind them)
    if (tail) {                           // the text projector's tail on this prompt's row: one launch (4.6 us of a decode) less
        if (tid < 64) {
            float v[8];
            wg_ctp_tail_row(a.init_prompt + (long)p * TK_C, a.tail_g, a.tail_b, a.tail_tt, a.tail_lt, TK_C, a.tail_eps, tid, v);
            if (tid * 8 < TK_C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) t0[tid * 8 + e] = (float)(bf16)v[e];
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < TK_N * TK_C; i += TK_THREADS) {
        float v, pv;
        if (a.stages & ST_INIT) {         // tokens = cat(output tokens, prompt) (mask_decoder.py:125-132) = queries = query_pe of the first block
            v = pv = i < 5 * TK_C ? a.init_tokens[i] : tail ? t0[i - 5 * TK_C] : (float)a.init_prompt[(long)p * TK_C + i - 5 * TK_C];
            a.pe[(long)p * TK_N * TK_C + i] = pv;
        } else {
            v = a.queries[(long)p * TK_N * TK_C + i];
            pv = a.pe[(long)p * TK_N * TK_C + i];
        }
        if (a.stages & ST_SUM_MLP) {      // x + mlp(x) (transformer.py:169-171): the eight slices' partial sums and the bias of lin2
            v += (float)a.lin2_b[i % TK_C];
            const float* mp = a.mlp_part + (long)p * TK_SLICES * TK_N * TK_C + i;
#pragma unroll
            for (int sl_ = 0; sl_ < TK_SLICES; ++sl_) v += mp[sl_ * TK_N * TK_C];
        }
        qs[i] = v;
        pes[i] = pv;
    }
    __syncthreads();
    // SUM_MLP closes a block and the SELF / Q_T2I that follow open the next one: every Linear of both reads the SAME rows (norm3's output, with
    // and without the positional term), so they are ONE pass over the weights -- the chain is a sequence of L2 round trips through one CU and a
    // pass costs ~3.5 us whatever its width (round 4: the second launch of a decode ran four passes, 21 us; the last one two).
    const bool sum = a.stages & ST_SUM_MLP, self = a.stages & ST_SELF, q_t2i = a.stages & ST_Q_T2I;
    const bool self_in_sum = sum && self, q_in_sum = sum && !self && q_t2i;
    if (sum) {
        // ---- norm3, then k / v of the image -> token attention (:172-178) ------------------------------------------------------------------
        tok_layernorm(qs, TK_N, ln3, a.eps);
        tok_stage<TK_C>(qs, TK_C, pes, TK_N, sh[0], sl[0]);
        tok_stage<TK_C>(qs, TK_C, nullptr, TK_N, sh[1], sl[1]);
        __syncthreads();
        float* ki = t3;                    // [6][128] each
        float* vi = t3 + TK_N * 128;
        if (self_in_sum) {                 // + q, k, v of the next block's self attention (:153-160)
            const int qk = a.skip_pe ? 1 : 0;
            const TokJob jobs[5] = {{sh[0], sl[0], a.i2t_k, 128, ki, 128, 0, nullptr, 0}, {sh[1], sl[1], a.i2t_v, 128, vi, 128, 0, nullptr, 0},
                                    {sh[qk], sl[qk], a.self_attn.q, TK_C, t0, TK_C, 0, nullptr, 0},
                                    {sh[qk], sl[qk], a.self_attn.k, TK_C, t1, TK_C, 0, nullptr, 0},
                                    {sh[1], sl[1], a.self_attn.v, TK_C, t2, TK_C, 0, nullptr, 0}};
            tok_mm<TK_C, 5>(jobs, TK_N);
        } else if (q_in_sum) {             // + q of the token -> image attention that follows (the tail: :96-99)
            const TokJob jobs[3] = {{sh[0], sl[0], a.i2t_k, 128, ki, 128, 0, nullptr, 0}, {sh[1], sl[1], a.i2t_v, 128, vi, 128, 0, nullptr, 0},
                                    {sh[0], sl[0], a.t2i_q, 128, t0, 128, 0, nullptr, 0}};
            tok_mm<TK_C, 3>(jobs, TK_N);
        } else {
            const TokJob jobs[2] = {{sh[0], sl[0], a.i2t_k, 128, ki, 128, 0, nullptr, 0}, {sh[1], sl[1], a.i2t_v, 128, vi, 128, 0, nullptr, 0}};
            tok_mm<TK_C, 2>(jobs, TK_N);
        }
        for (int i = tid; i < TK_N * 128; i += TK_THREADS) {
            a.k_i2t[(long)p * TK_N * 128 + i] = (bf16)ki[i];
            a.v_i2t[(long)p * TK_N * 128 + i] = (bf16)vi[i];
            if (q_in_sum) a.q_t2i[(long)p * TK_N * 128 + i] = t0[i];
        }
        __syncthreads();
    }
    if (self) {
        // ---- self attention (:153-160): layer 0 replaces the queries and skips the positional term; q, k, v in one pass -----------------------
        if (!self_in_sum) {
            tok_stage<TK_C>(qs, TK_C, a.skip_pe ? nullptr : pes, TK_N, sh[0], sl[0]);
            tok_stage<TK_C>(qs, TK_C, nullptr, TK_N, sh[1], sl[1]);
            __syncthreads();
            const TokJob qkv[3] = {{sh[0], sl[0], a.self_attn.q, TK_C, t0, TK_C, 0, nullptr, 0},
                                   {sh[0], sl[0], a.self_attn.k, TK_C, t1, TK_C, 0, nullptr, 0},
                                   {sh[1], sl[1], a.self_attn.v, TK_C, t2, TK_C, 0, nullptr, 0}};
            tok_mm<TK_C, 3>(qkv, TK_N);
        }
        tok_self_attention(t0, t1, t2, t3);
        const LnRegs ln1 = tok_ln_load(a.norm1);
        tok_linear<TK_C>(t3, TK_C, TK_N, a.self_attn.o, TK_C, qs, TK_C, 0, a.skip_pe ? nullptr : qs, TK_C, sh[0], sl[0]);
        tok_layernorm(qs, TK_N, ln1, a.eps);
    }
    if (q_t2i && !q_in_sum) {
        // ---- q of the token -> image attention (:162-165; tail: :96-101), internal width 128 ------------------------------------------------
        tok_stage<TK_C>(qs, TK_C, pes, TK_N, sh[0], sl[0]);
        __syncthreads();
        const TokJob qj[1] = {{sh[0], sl[0], a.t2i_q, 128, t0, 128, 0, nullptr, 0}};
        tok_mm<TK_C, 1>(qj, TK_N);
        for (int i = tid; i < TK_N * 128; i += TK_THREADS) a.q_t2i[(long)p * TK_N * 128 + i] = t0[i];
        __syncthreads();
    }
    if (a.stages & ST_COMBINE) {
        const CombineW c{a.attn_part, a.n_splits, a.t2i_o, a.norm2, a.eps};
        tok_combine(c, p, 0, TK_N, qs, t1, sh[0], sl[0]);
    }
    for (int i = tid; i < TK_N * TK_C; i += TK_THREADS) a.queries[(long)p * TK_N * TK_C + i] = qs[i];
}

// token -> image attention, one workgroup of four waves per (prompt, head, split of 1024 keys), a wave per 256 keys; the waves' online-softmax
// states are merged in LDS.  q [P,6,128] fp32 (unscaled); K / V image rows [hw][ld] bf16.
constexpr int AT_WAVES = 4;
constexpr int AT_KEYS = 256 * AT_WAVES;    // keys per workgroup
struct AttnPartArgs {
    const float* q; const bf16* K; const bf16* V; long ld, img_bs; const int* pimg; int head_stride, hw, n_splits; float* part;
};
// max / sum over the 16 lanes of a DPP row; every lane of the row ends with the result
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, WG_DPP(v, 0xB1));
    v = fmaxf(v, WG_DPP(v, 0x4E));
    v = fmaxf(v, WG_DPP(v, 0x124));
    v = fmaxf(v, WG_DPP(v, 0x128));
    return v;
}
__device__ __forceinline__ float row16_add(float v) {
    v += WG_DPP(v, 0xB1);
    v += WG_DPP(v, 0x4E);
    v += WG_DPP(v, 0x124);
    v += WG_DPP(v, 0x128);
    return v;
}

__global__ __launch_bounds__(64 * AT_WAVES) void wg_dec_attn_partial_kernel(AttnPartArgs a) {
    __shared__ __attribute__((aligned(16))) float qsh[TK_N * 16];
    __shared__ float wpart[AT_WAVES * 4][TK_PART];       // one softmax state per 16-lane row of every wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int split = blockIdx.x % a.n_splits, h = (blockIdx.x / a.n_splits) & 7, p = blockIdx.x / (a.n_splits * 8);
    for (int i = threadIdx.x; i < TK_N * 16; i += 64 * AT_WAVES)
        qsh[i] = a.q[((long)p * TK_N + i / 16) * 128 + h * 16 + (i & 15)] * 0.25f;   // 1 / sqrt(16)
    __syncthreads();
    const long img = a.pimg ? a.pimg[p] : p;             // the image whose tokens prompt p attends to (first block: shared by an image's prompts)
    const bf16* Kp = a.K + img * a.img_bs * a.ld + h * a.head_stride;
    const bf16* Vp = a.V + img * a.img_bs * a.ld + h * a.head_stride;
    const int j0 = split * AT_KEYS + wave * 256;
    // all of this lane's four keys are requested before the first one is used
    bf16x8 kk[4][2], vv[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = j0 + u * 64 + lane;
        const long jj = j < a.hw ? j : a.hw - 1;
        kk[u][0] = *(const bf16x8*)(Kp + jj * a.ld); kk[u][1] = *(const bf16x8*)(Kp + jj * a.ld + 8);
        vv[u][0] = *(const bf16x8*)(Vp + jj * a.ld); vv[u][1] = *(const bf16x8*)(Vp + jj * a.ld + 8);
    }
    // scores of the four keys against the six tokens, then ONE softmax state per lane: no running maximum, no rescaling
    float sc[4][TK_N];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool valid = j0 + u * 64 + lane < a.hw;
        float kf[16];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            kf[d] = (float)kk[u][0][d];
            kf[8 + d] = (float)kk[u][1][d];
        }
#pragma unroll
        for (int t = 0; t < TK_N; ++t) {
            float s_ = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                const f32x4 qv = *(const f32x4*)(qsh + t * 16 + 4 * d4);
                s_ += qv[0] * kf[4 * d4] + qv[1] * kf[4 * d4 + 1] + qv[2] * kf[4 * d4 + 2] + qv[3] * kf[4 * d4 + 3];
            }
            sc[u][t] = valid ? s_ : -1e30f;
        }
    }
    float m[TK_N], l[TK_N], acc[TK_N][16];
#pragma unroll
    for (int t = 0; t < TK_N; ++t) {
        m[t] = row16_max(fmaxf(fmaxf(sc[0][t], sc[1][t]), fmaxf(sc[2][t], sc[3][t])));     // the maximum of the lane's 16-lane row
        l[t] = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[t][d] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool valid = j0 + u * 64 + lane < a.hw;
        float vf[16];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            vf[d] = (float)vv[u][0][d];
            vf[8 + d] = (float)vv[u][1][d];
        }
#pragma unroll
        for (int t = 0; t < TK_N; ++t) {
            const float pr = valid ? __expf(sc[u][t] - m[t]) : 0.f;
            l[t] += pr;
#pragma unroll
            for (int d = 0; d < 16; ++d) acc[t][d] += pr * vf[d];
        }
    }
    // lanes -> 16-lane rows by DPP (the cross-row steps of a full wave reduction are permlane swaps with wait states: two thirds of the
    // instructions of the first version of this kernel); the rows' states meet in LDS
    const int row = lane >> 4, l16 = lane & 15;
    float* mine = wpart[wave * 4 + row];
    // step by step over ALL values (a DPP read of a register written by the previous instruction costs wait states; independent neighbours do not)
#define WG_ROW_STEP(ctrl)                                                   \
    _Pragma("unroll") for (int t = 0; t < TK_N; ++t) {                      \
        l[t] += WG_DPP(l[t], ctrl);                                         \
        _Pragma("unroll") for (int d = 0; d < 16; ++d) acc[t][d] += WG_DPP(acc[t][d], ctrl); \
    }
    WG_ROW_STEP(0xB1)
    WG_ROW_STEP(0x4E)
    WG_ROW_STEP(0x124)
    WG_ROW_STEP(0x128)
#undef WG_ROW_STEP
#pragma unroll
    for (int t = 0; t < TK_N; ++t) {
        float ov = acc[t][0];                 // lane l16 of the row writes output dim l16 (every lane of the row holds every total)
#pragma unroll
        for (int d = 1; d < 16; ++d) ov = l16 == d ? acc[t][d] : ov;
        mine[2 * TK_N + t * 16 + l16] = ov;
        if (l16 == 0) { mine[t] = m[t]; mine[TK_N + t] = l[t]; }
    }
    __syncthreads();
    // rows and waves -> workgroup
    float* out = a.part + (((long)p * 8 + h) * a.n_splits + split) * TK_PART;
    if (threadIdx.x < TK_N * 16) {
        const int t = threadIdx.x >> 4, d = threadIdx.x & 15;
        float M = -1e30f;
#pragma unroll
        for (int w_ = 0; w_ < AT_WAVES * 4; ++w_) M = fmaxf(M, wpart[w_][t]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < AT_WAVES * 4; ++w_) {
            const float w = __expf(wpart[w_][t] - M);
            L += wpart[w_][TK_N + t] * w;
            o += wpart[w_][2 * TK_N + t * 16 + d] * w;
        }
        out[2 * TK_N + t * 16 + d] = o;
        if (d == 0) { out[t] = M; out[TK_N + t] = L; }
    }
}

// one slice (256 hidden units) of the MLP of a prompt's six tokens: part[p][slice] = relu(x lin1_s^T + b1_s) lin2[:, slice]^T.
// With `comb.part` set, x is first taken through COMBINE (every slice repeats it: 6 x 128 x 256 MACs against a launch saved) and slice 0
// writes it to x_out -- a buffer other than x_in, which the other slices are still reading.
struct MlpArgs { const float* x; CombineW comb; float* x_out; LinW lin1; const bf16* lin2_w; float* part; };
__global__ __launch_bounds__(TK_THREADS) void wg_dec_mlp_partial_kernel(MlpArgs a) {
    __shared__ float xs[TK_N * TK_C], hs[TK_N * TK_C], ys[TK_N * TK_C];
    __shared__ __attribute__((aligned(16))) bf16 sh[8 * (TK_C + 8)], sl[8 * (TK_C + 8)];
    const int slice = blockIdx.x % TK_SLICES, p = blockIdx.x / TK_SLICES;
    for (int i = threadIdx.x; i < TK_N * TK_C; i += TK_THREADS) xs[i] = a.x[(long)p * TK_N * TK_C + i];
    __syncthreads();
    if (a.comb.part) {
        tok_combine(a.comb, p, 0, TK_N, xs, hs, sh, sl);
        if (slice == 0)
            for (int i = threadIdx.x; i < TK_N * TK_C; i += TK_THREADS) a.x_out[(long)p * TK_N * TK_C + i] = xs[i];
    }
    const LinW l1{a.lin1.w + (long)slice * TK_SLW * TK_C, a.lin1.b + slice * TK_SLW};   // rows slice*TK_SLW .. of lin1.weight = TK_SLW / 16 column blocks
    tok_linear<TK_C>(xs, TK_C, TK_N, l1, TK_SLW, hs, TK_C, WG_ACT_RELU, nullptr, 0, sh, sl);
    const LinW l2{a.lin2_w + (long)slice * TK_SLW * TK_C, nullptr};     // columns slice*TK_SLW .. of lin2.weight [256, 2048], tiled per slice
    tok_linear<TK_SLW>(hs, TK_C, TK_N, l2, TK_C, ys, TK_C, 0, nullptr, 0, sh, sl);
    float* out = a.part + ((long)p * TK_SLICES + slice) * TK_N * TK_C;
    for (int i = threadIdx.x; i < TK_N * TK_C; i += TK_THREADS) out[i] = ys[i];
}

// hypernetwork MLPs on the mask tokens (rows 1..4), IoU head on row 0 (mask_decoder.py:146-160): one (prompt, head) pair per workgroup.
// With `comb.part` set, the workgroup first takes ITS token row through COMBINE (final attention out_proj + norm_final_attn).
struct HeadArgs { const float* x; CombineW comb; LinW mlp[5][3]; float* hyper_out; float* iou_out; };
__global__ __launch_bounds__(TK_THREADS) void wg_dec_heads_kernel(HeadArgs a) {
    __shared__ float x0[TK_C], x1[TK_C], x2[TK_C];
    __shared__ __attribute__((aligned(16))) bf16 sh[8 * (TK_C + 8)], sl[8 * (TK_C + 8)];
    const int i = blockIdx.x % 5, p = blockIdx.x / 5;
    const int row = i < 4 ? 1 + i : 0;
    const float* x = a.x + ((long)p * TK_N + row) * TK_C;
    for (int c = threadIdx.x; c < TK_C; c += TK_THREADS) x0[c] = x[c];
    __syncthreads();
    if (a.comb.part) tok_combine(a.comb, p, row, 1, x0, x1, sh, sl);
    tok_linear<TK_C>(x0, TK_C, 1, a.mlp[i][0], TK_C, x1, TK_C, WG_ACT_RELU, nullptr, 0, sh, sl);
    tok_linear<TK_C>(x1, TK_C, 1, a.mlp[i][1], TK_C, x2, TK_C, WG_ACT_RELU, nullptr, 0, sh, sl);
    tok_linear<TK_C>(x2, TK_C, 1, a.mlp[i][2], i < 4 ? 32 : 4, x0, TK_C, 0, nullptr, 0, sh, sl);
    if (i < 4) {
        if (threadIdx.x < 32) a.hyper_out[((long)p * 4 + i) * 32 + threadIdx.x] = x0[threadIdx.x];
    } else if (threadIdx.x < 4) {
        a.iou_out[(long)p * 4 + threadIdx.x] = x0[threadIdx.x];
    }
}

// W [N][K] (row stride ld) -> fragment order T[ceil(N / 16)][K / 32][64][8] (see WEIGHT LAYOUT above); one thread per 16-byte piece
__global__ void wg_tile_weight_kernel(const bf16* w, long ld, int N, int K, bf16* t) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // piece index = (nb * K/32 + ks) * 64 + lane
    const int KS = K / 32;
    const long total = (long)((N + 15) / 16) * KS * 64;
    if (i >= total) return;
    const int lane = (int)(i & 63), ks = (int)((i >> 6) % KS), nb = (int)((i >> 6) / KS);
    const int n = nb * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4);
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n < N) v = *(const bf16x8*)(w + (long)n * ld + k);
    *(bf16x8*)(t + i * 8) = v;
}

}  // namespace

// One-time re-layout of a weight matrix for the token-side kernels: W [N][K] bf16 (row stride ld >= K, K % 32 == 0, 16-byte aligned rows)
// -> tiled [ceil(N / 16)][K / 32][64][8] bf16 (rows beyond N zero).  A K-slice of a wider matrix is tiled by passing W + k0 with its ld.
extern "C" int wg_tile_weight_bf16(const void* W, long ld, int N, int K, void* tiled, void* stream) {
    WG_REQUIRE(W && tiled && N > 0 && K > 0 && K % 32 == 0 && ld >= K && ld % 8 == 0, "tile_weight: bad arguments");
    WG_REQUIRE((((uintptr_t)W | (uintptr_t)tiled) & 15) == 0, "tile_weight: misaligned operand");
    const long total = (long)((N + 15) / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(wg_tile_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)W, ld, N, K,
                       (bf16*)tiled);
    return wg_check_launch("wg_tile_weight");
}

// Flat pointer table of wg_dec_tokens_f32 (bf16 device pointers, weight then bias / gamma then beta; entries of stages that are not
// requested may be null), 24 entries:
//   self_attn q,k,v,out (8) | norm1 (2) | token->image attention q,out (4) | norm2 or norm_final_attn (2) | mlp.lin2 bias (1) + unused (1) |
//   norm3 (2) | image->token attention k,v (4)
static int wg_dec_tokens_impl(int stages, int skip_pe, float* queries, float* query_pe, const float* init_tokens, const void* init_prompt,
                              const void* const* prompt_tail, float prompt_tail_eps, const void* const* weights, int n_weights, float* q_t2i,
                              const float* attn_partials, int n_splits, const float* mlp_partials, void* k_i2t, void* v_i2t, int P, float eps, void* stream);

extern "C" int wg_dec_tokens_f32(int stages, int skip_pe, float* queries, float* query_pe, const float* init_tokens, const void* init_prompt,
                                 const void* const* weights, int n_weights, float* q_t2i, const float* attn_partials, int n_splits, const float* mlp_partials, void* k_i2t, void* v_i2t,
                                 int P, float eps, void* stream) {
    return wg_dec_tokens_impl(stages, skip_pe, queries, query_pe, init_tokens, init_prompt, nullptr, 0.f, weights, n_weights, q_t2i, attn_partials, n_splits,
                              mlp_partials, k_i2t, v_i2t, P, eps, stream);
}

// The first launch of a decode (stages must carry INIT) with the text projector's tail folded in: init_prompt [P, 256] bf16 holds the rows BEFORE
// the tail (the output of CalibratedTextProjector.net[3]); prompt_tail = {LayerNorm gamma, beta, text_type, log_temp} (bf16, 256 / 256 / 256 / 1).
extern "C" int wg_dec_tokens_ctp_f32(int stages, int skip_pe, float* queries, float* query_pe, const float* init_tokens, const void* init_prompt,
                                     const void* const* prompt_tail, float prompt_tail_eps, const void* const* weights, int n_weights, float* q_t2i,
                                     const float* attn_partials, int n_splits, const float* mlp_partials, void* k_i2t, void* v_i2t, int P, float eps,
                                     void* stream) {
    WG_REQUIRE((stages & ST_INIT) && prompt_tail && prompt_tail[0] && prompt_tail[1] && prompt_tail[2] && prompt_tail[3],
               "dec_tokens_ctp: the tail is part of INIT and needs gamma, beta, text_type and log_temp");
    WG_REQUIRE((((uintptr_t)prompt_tail[0] | (uintptr_t)prompt_tail[1] | (uintptr_t)prompt_tail[2] | (uintptr_t)init_prompt) & 15) == 0,
               "dec_tokens_ctp: misaligned tail operands");
    return wg_dec_tokens_impl(stages, skip_pe, queries, query_pe, init_tokens, init_prompt, prompt_tail, prompt_tail_eps, weights, n_weights, q_t2i,
                              attn_partials, n_splits, mlp_partials, k_i2t, v_i2t, P, eps, stream);
}

static int wg_dec_tokens_impl(int stages, int skip_pe, float* queries, float* query_pe, const float* init_tokens, const void* init_prompt,
                              const void* const* prompt_tail, float prompt_tail_eps, const void* const* weights, int n_weights, float* q_t2i,
                              const float* attn_partials, int n_splits, const float* mlp_partials, void* k_i2t, void* v_i2t, int P, float eps, void* stream) {
    WG_REQUIRE(queries && query_pe && weights && n_weights == 24 && P > 0, "dec_tokens: bad arguments");
    WG_REQUIRE(stages > 0 && stages < 32, "dec_tokens: bad stage mask %d", stages);
    WG_REQUIRE(!(stages & ST_INIT) || (init_tokens && init_prompt && !(stages & ST_SUM_MLP)), "dec_tokens: INIT needs the output tokens and the prompt rows (and cannot follow an MLP)");
    TokArgs a{};
    a.stages = stages; a.skip_pe = skip_pe; a.P = P; a.n_splits = n_splits; a.queries = queries; a.pe = query_pe; a.eps = eps;
    a.init_tokens = init_tokens; a.init_prompt = (const bf16*)init_prompt;
    if (prompt_tail) {
        a.tail_g = (const bf16*)prompt_tail[0]; a.tail_b = (const bf16*)prompt_tail[1]; a.tail_tt = (const bf16*)prompt_tail[2];
        a.tail_lt = (const bf16*)prompt_tail[3]; a.tail_eps = prompt_tail_eps;
    }
    a.q_t2i = q_t2i; a.attn_part = attn_partials; a.mlp_part = mlp_partials; a.k_i2t = (bf16*)k_i2t; a.v_i2t = (bf16*)v_i2t;
    const bf16* const* w = (const bf16* const*)weights;
    auto need = [&](int lo, int hi) { for (int i = lo; i < hi; ++i) if (!w[i] || ((uintptr_t)w[i] & 15)) return false; return true; };
    a.self_attn = AttnW{{w[0], w[1]}, {w[2], w[3]}, {w[4], w[5]}, {w[6], w[7]}};
    a.norm1 = NormW{w[8], w[9]};
    a.t2i_q = LinW{w[10], w[11]}; a.t2i_o = LinW{w[12], w[13]}; a.norm2 = NormW{w[14], w[15]};
    a.lin2_b = w[16]; a.norm3 = NormW{w[18], w[19]};
    a.i2t_k = LinW{w[20], w[21]}; a.i2t_v = LinW{w[22], w[23]};
    if (stages & ST_SELF) WG_REQUIRE(need(0, 10), "dec_tokens: SELF needs the self-attention and norm1 weights");
    if (stages & ST_Q_T2I) WG_REQUIRE(need(10, 12) && q_t2i, "dec_tokens: Q_T2I needs the q projection and its output buffer");
    if (stages & ST_COMBINE) WG_REQUIRE(need(12, 16) && attn_partials && n_splits > 0, "dec_tokens: COMBINE needs out_proj, the norm and the attention partials");
    if (stages & ST_SUM_MLP) WG_REQUIRE(need(16, 17) && need(18, 24) && mlp_partials && k_i2t && v_i2t, "dec_tokens: SUM_MLP needs lin2 bias, norm3, k / v projections, the MLP partials and its outputs");
    hipLaunchKernelGGL(wg_dec_tokens_kernel, dim3(P), dim3(TK_THREADS), 0, (hipStream_t)stream, a);
    return wg_check_launch("wg_dec_tokens");
}

// q [P,6,128] fp32; Kimg / Vimg: head h's 16 columns at Kimg + h * head_stride of each of the [P or 1][hw] rows (stride ld_img) -- head_stride 16:
// plain [.., 128] blocks; 32: the projection's columns ordered [K_h | V_h] per head, one 64-byte piece per (key, head) -- (img_rows_per_prompt = 0: one image shared by all prompts; prompt_image != null: prompt p reads image prompt_image[p], images
// img_rows_per_prompt rows apart);
// partials [P, 8, n_splits, 108] fp32 with n_splits = ceil(hw / 1024).
extern "C" int wg_dec_attn_partial_f32(const float* q, const void* Kimg, const void* Vimg, long ld_img, int head_stride, long img_rows_per_prompt,
                                       const int* prompt_image, int hw,
                                       float* partials, int n_splits, int P, void* stream) {
    WG_REQUIRE(q && Kimg && Vimg && partials && P > 0 && hw > 0 && ld_img % 8 == 0, "dec_attn_partial: bad arguments");
    WG_REQUIRE(n_splits == (hw + AT_KEYS - 1) / AT_KEYS, "dec_attn_partial: n_splits must be ceil(hw / 1024)");
    WG_REQUIRE((((uintptr_t)Kimg | (uintptr_t)Vimg) & 15) == 0, "dec_attn_partial: misaligned image projections");
    WG_REQUIRE(head_stride == 16 || head_stride == 32, "dec_attn_partial: head_stride must be 16 (K and V blocks) or 32 (K | V interleaved per head)");
    AttnPartArgs a{q, (const bf16*)Kimg, (const bf16*)Vimg, ld_img, img_rows_per_prompt, prompt_image, head_stride, hw, n_splits, partials};
    hipLaunchKernelGGL(wg_dec_attn_partial_kernel, dim3((unsigned)(P * 8 * n_splits)), dim3(64 * AT_WAVES), 0, (hipStream_t)stream, a);
    return wg_check_launch("wg_dec_attn_partial");
}

// combine: 6 pointers {attention partials (fp32 [P,8,n_splits,108]), out_proj weight [256,128], bias, LayerNorm gamma, beta} or null
static bool wg_parse_combine(const void* const* combine, int n_splits, float eps, CombineW& c) {
    c = CombineW{nullptr, 0, LinW{nullptr, nullptr}, NormW{nullptr, nullptr}, eps};
    if (!combine) return true;
    for (int i = 0; i < 5; ++i)
        if (!combine[i]) return false;
    if (n_splits <= 0 || ((uintptr_t)combine[1] & 15)) return false;
    c.part = (const float*)combine[0]; c.n_splits = n_splits;
    c.wo = LinW{(const bf16*)combine[1], (const bf16*)combine[2]};
    c.nw = NormW{(const bf16*)combine[3], (const bf16*)combine[4]};
    return true;
}

// x [P,6,256] fp32; lin1 [2048,256] + bias, lin2 weight [256,2048] tiled in S = wg_dec_mlp_slices() K-slices (bias added by the SUM_MLP stage); partials [P, S, 6, 256] fp32.
// combine == null: x holds the tokens after norm2.  combine != null (5 pointers: attention partials, token->image out_proj weight, bias,
// norm2 gamma, beta): x holds the tokens BEFORE the COMBINE stage, which this launch runs itself; the tokens after norm2 go to x_out
// (a buffer other than x).
extern "C" int wg_dec_mlp_slices(void) { return TK_SLICES; }      // slices of wg_dec_mlp_partial_f32's partials (the host sizes its buffers and tiles lin2 by it)

extern "C" int wg_dec_mlp_partial_f32(const float* x, const void* const* combine, int n_splits, float eps, float* x_out, const void* lin1_w,
                                      const void* lin1_b, const void* lin2_w, float* partials, int P, void* stream) {
    WG_REQUIRE(x && lin1_w && lin1_b && lin2_w && partials && P > 0, "dec_mlp_partial: bad arguments");
    WG_REQUIRE((((uintptr_t)lin1_w | (uintptr_t)lin2_w) & 15) == 0, "dec_mlp_partial: misaligned weights");
    MlpArgs a{x, CombineW{}, x_out, LinW{(const bf16*)lin1_w, (const bf16*)lin1_b}, (const bf16*)lin2_w, partials};
    WG_REQUIRE(wg_parse_combine(combine, n_splits, eps, a.comb), "dec_mlp_partial: bad combine table");
    WG_REQUIRE(!combine || (x_out && x_out != x), "dec_mlp_partial: combine needs x_out, distinct from x");
    hipLaunchKernelGGL(wg_dec_mlp_partial_kernel, dim3((unsigned)(P * TK_SLICES)), dim3(TK_THREADS), 0, (hipStream_t)stream, a);
    return wg_check_launch("wg_dec_mlp_partial");
}

// x [P,6,256] fp32; weights: 30 bf16 pointers = 5 MLPs (hypernetwork 0..3, IoU head) x 3 layers x (weight, bias) -> hyper_out [P,4,32],
// iou_out [P,4] fp32.  combine == null: x holds the tokens after norm_final_attn; otherwise (table as above, with the final attention's
// out_proj and norm_final_attn) x holds them before the final COMBINE, which every workgroup runs on its own token row.
extern "C" int wg_dec_heads_f32(const float* x, const void* const* combine, int n_splits, float eps, const void* const* weights, int n_weights,
                                float* hyper_out, float* iou_out, int P, void* stream) {
    WG_REQUIRE(x && weights && n_weights == 30 && hyper_out && iou_out && P > 0, "dec_heads: bad arguments");
    HeadArgs a{};
    a.x = x; a.hyper_out = hyper_out; a.iou_out = iou_out;
    WG_REQUIRE(wg_parse_combine(combine, n_splits, eps, a.comb), "dec_heads: bad combine table");
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 3; ++j) {
            const void* wp = weights[(i * 3 + j) * 2];
            WG_REQUIRE(wp && ((uintptr_t)wp & 15) == 0 && weights[(i * 3 + j) * 2 + 1], "dec_heads: weight %d null or misaligned", i * 3 + j);
            a.mlp[i][j] = LinW{(const bf16*)wp, (const bf16*)weights[(i * 3 + j) * 2 + 1]};
        }
    hipLaunchKernelGGL(wg_dec_heads_kernel, dim3((unsigned)(P * 5)), dim3(TK_THREADS), 0, (hipStream_t)stream, a);
    return wg_check_launch("wg_dec_heads");
}

// transformer.py:173-180 for one block: q [rows | hw][ldq] bf16 (the 128 q columns of the image-side projection), kq / vq [P,6,128] bf16
// (wg_dec_tokens_f32, SUM_MLP), out_proj [256,128] + bias, res = the image tokens (bf16 rows, stride ldr), norm4 -> out [P*hw, 256] bf16.
// row_mod = hw when q and res hold ONE image shared by all P prompts, 0 when they have P*hw rows.
extern "C" int wg_dec_i2t_rows_bf16(const void* q, long ldq, const void* kq, const void* vq, const void* wo, const void* bo, const void* res,
                                    long ldr, const void* res_bias, int row_mod, const int* prompt_image, const void* ln_g, const void* ln_b, float eps, void* out, int P, int hw, void* stream) {
    WG_REQUIRE(q && kq && vq && wo && bo && res && ln_g && ln_b && out, "dec_i2t_rows: null operand");
    WG_REQUIRE(P > 0 && hw > 0 && hw % 16 == 0 && (row_mod == 0 || row_mod == hw), "dec_i2t_rows: hw must be a positive multiple of 16");
    WG_REQUIRE(ldq % 8 == 0 && ldr % 4 == 0, "dec_i2t_rows: misaligned leading dimension");
    WG_REQUIRE((((uintptr_t)q | (uintptr_t)kq | (uintptr_t)vq | (uintptr_t)wo) & 15) == 0 && (((uintptr_t)res | (uintptr_t)out) & 7) == 0,
               "dec_i2t_rows: misaligned operand");
    I2tArgs a{(const bf16*)q, ldq, (const bf16*)kq, (const bf16*)vq, (const bf16*)wo, (const bf16*)bo, (const bf16*)res, ldr, (const bf16*)res_bias, row_mod, prompt_image,
              (const bf16*)ln_g, (const bf16*)ln_b, eps, (bf16*)out, (long)P * hw, hw};
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) (void)hipFuncSetAttribute((const void*)wg_dec_i2t_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, I2_LDS);
    const long wgs = (a.rows / 16 + I2_WAVES - 1) / I2_WAVES;
    const long cus = wg_cu_count(dev);
    hipLaunchKernelGGL(wg_dec_i2t_rows_kernel, dim3((unsigned)(wgs < cus ? wgs : cus)), dim3(64 * I2_WAVES), I2_LDS, (hipStream_t)stream, a);
    return wg_check_launch("wg_dec_i2t_rows");
}
