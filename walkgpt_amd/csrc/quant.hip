// Row quantisation to OCP e4m3 for the fp8 GEMM (gemm.hip: wg_gemm_fp8_bias_act; BASELINE config C5).
//   wg_quantize_rows_fp8        q[m][:] = e4m3(x[m][:] / s[m]),  s[m] = max|x[m][:]| / 448   (activations per call, weights once)
//   wg_layernorm_quantize_fp8   the same on LayerNorm(x) -- the pre-LN of a transformer block and the quantisation of its output
//                               are one pass over the row (statistics in fp32, as SURVEY.md 8d C5 asks)
// One wave per row, the row held in registers (K <= 5120), 16-byte loads, 8-byte stores.  HBM-bound: 3 bytes per element.
#include "wg_common.h"

namespace {

constexpr int QMAXIT = 10;          // 10 x 64 lanes x 8 values = 5120 columns
constexpr float E4M3_MAX = 448.0f;

template <bool LN>
__global__ __launch_bounds__(256) void wg_quantize_rows_kernel(const bf16* x, long ldx, const bf16* gamma, const bf16* beta, float eps,
                                                               unsigned char* q, long ldq, float* scale, int M, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16* xr = x + (long)row * ldx;
    float v[QMAXIT][8];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < QMAXIT; ++it) {
        const int c = it * 512 + lane * 8;
        if (c < K) {
            const bf16x8 t = *(const bf16x8*)(xr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[it][e] = (float)t[e];
                sum += v[it][e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[it][e] = 0.f;
        }
    }
    if (LN) {
        const float mean = wg_wave_sum(sum) / (float)K;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < QMAXIT; ++it) {
            const int c = it * 512 + lane * 8;
            if (c < K) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[it][e] -= mean;
                    sq += v[it][e] * v[it][e];
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wg_wave_sum(sq) / (float)K + eps);
#pragma unroll
        for (int it = 0; it < QMAXIT; ++it) {
            const int c = it * 512 + lane * 8;
            if (c < K) {
                const bf16x8 gm = *(const bf16x8*)(gamma + c), bt = *(const bf16x8*)(beta + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = v[it][e] * rstd * (float)gm[e] + (float)bt[e];
            }
        }
    }
    float amax = 0.f;
#pragma unroll
    for (int it = 0; it < QMAXIT; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[it][e]));
    amax = wg_wave_max(amax);
    const float s = amax > 0.f ? amax * (1.0f / E4M3_MAX) : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) scale[row] = s;
    unsigned char* qr = q + (long)row * ldq;
#pragma unroll
    for (int it = 0; it < QMAXIT; ++it) {
        const int c = it * 512 + lane * 8;
        if (c < K) {
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[it][0] * inv, v[it][1] * inv, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[it][2] * inv, v[it][3] * inv, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[it][4] * inv, v[it][5] * inv, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[it][6] * inv, v[it][7] * inv, hi, true);
            *(u32x2*)(qr + c) = (u32x2){(unsigned)lo, (unsigned)hi};
        }
    }
}

int launch(bool ln, const void* x, long ldx, const void* gamma, const void* beta, float eps, void* q, long ldq, float* scale, int M,
           int K, void* stream) {
    WG_REQUIRE(x && q && scale && (!ln || (gamma && beta)), "quantize_rows: null operand");
    WG_REQUIRE(M > 0 && K > 0 && K % 8 == 0 && K <= QMAXIT * 512, "quantize_rows: K = %d must be a multiple of 8, at most %d", K, QMAXIT * 512);
    WG_REQUIRE(ldx % 8 == 0 && ldq % 8 == 0 && ldx >= K && ldq >= K, "quantize_rows: leading dimensions must be multiples of 8 covering the row");
    WG_REQUIRE((((uintptr_t)x & 15) | ((uintptr_t)q & 7)) == 0, "quantize_rows: misaligned operand");
    const dim3 grid((unsigned)((M + 3) / 4)), block(256);
    if (ln)
        hipLaunchKernelGGL(wg_quantize_rows_kernel<true>, grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx, (const bf16*)gamma,
                           (const bf16*)beta, eps, (unsigned char*)q, ldq, scale, M, K);
    else
        hipLaunchKernelGGL(wg_quantize_rows_kernel<false>, grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx, nullptr, nullptr, 0.f,
                           (unsigned char*)q, ldq, scale, M, K);
    return wg_check_launch("wg_quantize_rows_fp8");
}

// x [M,K] bf16 -> e4m3 bytes [M,K] + one E8M0 scale per (row, 32 columns): the power of two at or above max|block| / 448 (the rule of
// the GEMM epilogue's MX output, gemm.hip wg_flush_slab_mx).  Scale planes [K/32][pitch]; inside every `group`-row group (128: the A
// side of the MX GEMMs, 8 MFMA fragments per lane; 64: the W side, 4 fragments) row r sits at (r % 16) * (group / 16) + r / 16.
// A lane takes 8 adjacent columns of a row, four lanes one block, 16 rows per 256-thread workgroup pass; HBM-bound (3 bytes / value).
// part != null (K % 256 == 0): also the rows' {sum, sum of squares} per 256-column tile, [K/256][part_mpad][2] fp32 -- the partial sums a
// GEMM with a folded LayerNorm reads (gemm.hip LNMODE 2), for a tensor no GEMM epilogue produced them for (the chain's entry).
__global__ __launch_bounds__(256) void wg_quantize_mx_kernel(const bf16* x, long ldx, unsigned char* q, long ldq, unsigned char* mx, long pitch,
                                                             int group, int M, int K, float* part, long part_mpad) {
    const int cpr = K / 8;                                  // 8-column pieces per row
    const long total = (long)M * cpr;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int row = (int)(idx / cpr), c = (int)(idx % cpr) * 8;
        const bf16x8 t = *(const bf16x8*)(x + (long)row * ldx + c);
        float v[8];
        float am = 0x1p-100f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = (float)t[e];
            am = fmaxf(am, fabsf(v[e]));
        }
        if (part) {                                         // 32 lanes = one 256-column tile of one row (K % 256 == 0: never a partial half-wave)
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sv += v[e];
                qv += v[e] * v[e];
            }
            sv += WG_DPP(sv, 0xB1); qv += WG_DPP(qv, 0xB1);
            sv += WG_DPP(sv, 0x4E); qv += WG_DPP(qv, 0x4E);
            sv += WG_DPP(sv, 0x124); qv += WG_DPP(qv, 0x124);
            sv += WG_DPP(sv, 0x128); qv += WG_DPP(qv, 0x128);
            float a, b;
            wg_permlane_swap<0>(sv, a, b); sv = a + b;
            wg_permlane_swap<0>(qv, a, b); qv = a + b;
            if ((threadIdx.x & 31) == 0) *(f32x2*)(part + 2 * ((long)(c >> 8) * part_mpad + row)) = (f32x2){sv, qv};
        }
        am = fmaxf(am, WG_DPP(am, 0xB1));                   // K % 32 == 0: the four lanes of a block are one aligned quad, all active
        am = fmaxf(am, WG_DPP(am, 0x4E));
        const unsigned bits = __builtin_bit_cast(unsigned, am * (1.0f / E4M3_MAX));
        unsigned e8 = (bits >> 23) + ((bits & 0x7FFFFFu) ? 1u : 0u);
        e8 = e8 > 253u ? 253u : e8;
        const float inv = __builtin_bit_cast(float, (254u - e8) << 23);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
        *(u32x2*)(q + (long)row * ldq + c) = (u32x2){(unsigned)lo, (unsigned)hi};
        if ((c & 31) == 0) {
            const int r = row % group;
            mx[(long)(c >> 5) * pitch + (row - r) + (r % 16) * (group / 16) + r / 16] = (unsigned char)e8;
        }
    }
}

}  // namespace

extern "C" int wg_quantize_mx_fp8(const void* x, long ldx, void* q, long ldq, void* mx, long pitch, int group, int M, int K, float* part,
                                  long part_mpad, void* stream) {
    WG_REQUIRE(x && q && mx, "quantize_mx: null operand");
    WG_REQUIRE(!part || (K % 256 == 0 && part_mpad >= M && ((uintptr_t)part & 7) == 0), "quantize_mx: row partials need K %% 256 == 0 and a pitch covering M");
    WG_REQUIRE(M > 0 && K > 0 && K % 32 == 0 && (group == 64 || group == 128), "quantize_mx: K = %d must be a multiple of 32, group 64 or 128", K);
    WG_REQUIRE(ldx % 8 == 0 && ldq % 8 == 0 && ldx >= K && ldq >= K, "quantize_mx: leading dimensions must be multiples of 8 covering the row");
    WG_REQUIRE(pitch >= ((long)M + group - 1) / group * group, "quantize_mx: scale pitch %ld does not cover %d rows in groups of %d", pitch, M, group);
    WG_REQUIRE((((uintptr_t)x & 15) | ((uintptr_t)q & 7)) == 0, "quantize_mx: misaligned operand");
    const long total = (long)M * (K / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(wg_quantize_mx_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, (unsigned char*)q, ldq,
                       (unsigned char*)mx, pitch, group, M, K, part, part_mpad);
    return wg_check_launch("wg_quantize_mx_fp8");
}

extern "C" int wg_quantize_rows_fp8(const void* x, long ldx, void* q, long ldq, float* scale, int M, int K, void* stream) {
    return launch(false, x, ldx, nullptr, nullptr, 0.f, q, ldq, scale, M, K, stream);
}

extern "C" int wg_layernorm_quantize_fp8(const void* x, long ldx, const void* gamma, const void* beta, float eps, void* q, long ldq,
                                         float* scale, int M, int K, void* stream) {
    return launch(true, x, ldx, gamma, beta, eps, q, ldq, scale, M, K, stream);
}
