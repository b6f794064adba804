// Shared device/host helpers for the walkgpt_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define WG_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- error plumbing (host) -------------------------------------------------
// Codes returned by every wg_* entry point (see include/walkgpt_hip.h).
enum {
    WG_OK = 0,
    WG_ERR_BAD_ARG = -1,
    WG_ERR_UNSUPPORTED = -2,
    WG_ERR_LAUNCH = -3,
};
void wg_set_error(const char* fmt, ...);
#define WG_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            wg_set_error(__VA_ARGS__);        \
            return WG_ERR_BAD_ARG;            \
        }                                     \
    } while (0)
int wg_check_launch(const char* what);
int wg_cu_count(int device);   // compute units of a device (cached): persistent grids are sized from it, not from a constant
// One-time setup per (call site, device): hipFuncSetAttribute and occupancy answers belong to a device, and a process may drive
// several.  `static WgPerDevice once; int dev; if (once.first(&dev)) { ...set attributes... }`
struct WgPerDevice {
    bool done[64] = {};
    bool first(int* dev_out) {
        int d = 0;
        (void)hipGetDevice(&d);
        *dev_out = d;
        if (d < 0 || d >= 64) return true;      // out of the table: repeat the (idempotent) setup every time
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

// ---- device helpers --------------------------------------------------------
__device__ __forceinline__ float wg_bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 wg_f2bf(float x) { return (bf16)x; }

// ---- wave-wide reductions on the VALU (DPP + gfx950 permlane swaps) instead of six ds_bpermute round trips through the LDS crossbar.
// v_permlaneNN_swap exchanges the upper half (odd rows) of its first register with the lower half (even rows) of the second;
// fed with two copies of v it leaves {lo, lo} and {hi, hi}, whose combination is the xor-32 (xor-16) butterfly step.
// Inline asm on purpose: __builtin_amdgcn_permlane{16,32}_swap of this hipcc (ROCm 7.2) hands back the FIRST register for both
// elements of its result, i.e. code using the builtin silently computes op(lo, lo) (checked on the GPU: a wave sum of 1..64
// returned 544).  s_nop 1 covers the VALU-write -> permlane-read hazard the compiler would otherwise pad.
template <int ROWS32> __device__ __forceinline__ void wg_permlane_swap(float v, float& a, float& b) {
    a = v; b = v;
    if constexpr (ROWS32) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// value of lane (l ^ 32) combined with lane l
__device__ __forceinline__ float wg_xor32_max(float v) { float a, b; wg_permlane_swap<1>(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float wg_xor32_sum(float v) { float a, b; wg_permlane_swap<1>(v, a, b); return a + b; }
#define WG_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float wg_wave_sum(float v) {
    v += WG_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
    v += WG_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
    v += WG_DPP(v, 0x124);   // row_ror:4
    v += WG_DPP(v, 0x128);   // row_ror:8  -> every lane of a 16-lane row holds the row total
    float a, b;
    wg_permlane_swap<0>(v, a, b); v = a + b;
    wg_permlane_swap<1>(v, a, b); return a + b;
}

// CTP tail of ONE row by one wave (utils_walkgpt.py:321-327): LayerNorm(C) -> + text_type -> L2 normalise (eps 1e-12) -> * exp(log_temp); lane l holds
// elements 8 l .. 8 l + 7 (C <= 512, C % 8 == 0).  v[e] returns the fp32 results (the caller rounds to bf16).  Shared by wg_ctp_tail_kernel and the
// INIT stage of wg_dec_tokens_kernel, so the folded form is the same arithmetic bit for bit.
__device__ __forceinline__ void wg_ctp_tail_row(const bf16* xrow, const bf16* gamma, const bf16* beta, const bf16* text_type, const bf16* log_temp, int C,
                                                float eps, int lane, float (&v)[8]) {
    const int d = lane * 8 < C ? lane * 8 : 0;      // (lanes past the row read its first piece and contribute nothing)
    const bool on = lane * 8 < C;
    // every operand is requested before the first reduction: the row's statistics are three dependent wave reductions, and loads issued
    // between them would each add a memory round trip to a chain that is latency from end to end
    const bf16x8 t = *(const bf16x8*)(xrow + d);
    const bf16x8 gm = *(const bf16x8*)(gamma + d), bt = *(const bf16x8*)(beta + d), tt = *(const bf16x8*)(text_type + d);
    const float lt = (float)log_temp[0];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = on ? (float)t[e] : 0.f; s += v[e]; }
    const float mean = wg_wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float u = v[e] - mean; q += on ? u * u : 0.f; }
    const float rstd = 1.0f / sqrtf(wg_wave_sum(q) / (float)C + eps);
    float n2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        v[e] = on ? (v[e] - mean) * rstd * (float)gm[e] + (float)bt[e] + (float)tt[e] : 0.f;
        n2 += v[e] * v[e];
    }
    const float nrm = fmaxf(sqrtf(wg_wave_sum(n2)), 1e-12f);
    const float sc = __expf(lt) / nrm;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= sc;
}
__device__ __forceinline__ float wg_wave_max(float v) {
    v = fmaxf(v, WG_DPP(v, 0xB1));
    v = fmaxf(v, WG_DPP(v, 0x4E));
    v = fmaxf(v, WG_DPP(v, 0x124));
    v = fmaxf(v, WG_DPP(v, 0x128));
    float a, b;
    wg_permlane_swap<0>(v, a, b); v = fmaxf(a, b);
    wg_permlane_swap<1>(v, a, b); return fmaxf(a, b);
}

// activation codes shared by the GEMM epilogue and the elementwise kernels
enum { WG_ACT_NONE = 0, WG_ACT_GELU_ERF = 1, WG_ACT_QUICK_GELU = 2, WG_ACT_RELU = 3 };

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. below fp32 resolution of the GELU it feeds): about a
// dozen VALU instructions instead of libm erff's ~40, which matters because the GEMM epilogue runs it on every output.
__device__ __forceinline__ float wg_erf(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float e = exp2f(-1.4426950408889634f * ax * ax);
    const float r = 1.0f - p * t * e;
    return copysignf(r, x);
}

__device__ __forceinline__ float wg_act(float x, int act) {
    switch (act) {
        case WG_ACT_GELU_ERF: return 0.5f * x * (1.0f + wg_erf(x * 0.70710678118654752440f));
        case WG_ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case WG_ACT_RELU: return fmaxf(x, 0.0f);
        default: return x;
    }
}

// ---- epilogue form of the activations: compile-time code, two values per instruction ---------------------------------------
// The GEMM epilogues run the activation on every output (128 values per lane on a 256x256 tile), with the matrix pipe idle
// meanwhile, so their cost is main-loop time.  These versions are straight-line (no per-element switch), written on float
// pairs so that hipcc selects v_pk_mul/v_pk_fma_f32, and use the bare v_exp_f32 / v_rcp_f32 (no denormal range scaling:
// exp2 of a very negative argument flushing to zero is exactly what both formulas want).
//   GELU(x) = x/2 + |x/2| * (1 - p(t) t e^{-x^2/2}),  t = 1 / (1 + 0.3275911 |x| / sqrt2)   (A&S 7.1.26, as wg_erf above)
template <int ACT> __device__ __forceinline__ f32x2 wg_act2(f32x2 x) {
    if constexpr (ACT == WG_ACT_GELU_ERF) {
        const f32x2 ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
        const f32x2 d = ax * 0.23164189f + 1.0f;                       // 0.3275911 / sqrt(2)
        const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        f32x2 p = t * 1.061405429f - 1.453152027f;
        p = p * t + 1.421413741f;
        p = p * t - 0.284496736f;
        p = p * t + 0.254829592f;
        const f32x2 a = (x * x) * -0.72134752f;                         // -(x^2 / 2) log2(e)
        const f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
        const f32x2 r = 1.0f - (p * t) * e;                             // erf(|x| / sqrt2)
        const f32x2 hx = x * 0.5f, ahx = ax * 0.5f;
        return hx + ahx * r;
    } else if constexpr (ACT == WG_ACT_QUICK_GELU) {
        const f32x2 a = x * -2.4554669f;                                // -1.702 log2(e)
        const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + 1.0f;
        return x * (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    } else if constexpr (ACT == WG_ACT_RELU) {
        return (f32x2){fmaxf(x.x, 0.0f), fmaxf(x.y, 0.0f)};
    } else {
        return x;
    }
}

// The GEMM epilogues' form (bf16 outputs only: wg_epi_pack below, the persistent kernel's LayerNorm-fold epilogue).  (round 6)
//   GELU(x) = x Phi(x),  Phi(x) = 1 / (1 + exp(-x (a + b x^2 + c x^4))),  x^2 clamped to 49
// A minimax fit of the sigmoid form to the erf form over the whole line: |GELU_fit - GELU_erf| <= 2.6e-5 absolute (tools/fit_gelu.py prints the
// fit and its fp32 error), i.e. 1/150 of a bf16 rounding step at |y| = 1, and every value that leaves through here is rounded to bf16.  Per PAIR of
// values 6 packed multiplies / FMAs, 2 minima and 4 transcendentals against 15 + 2 + 4 for the A&S 7.1.26 form above (the lin1 GEMM spends a
// third of every tile in this epilogue with the matrix pipe idle).  The A&S form stays wherever fp32 results leave the chip or a fused kernel is
// compared bit for bit with an unfused chain (decoder.hip, norm.hip, wg_act); coefficients below are (a, b, c) * -log2(e).
template <int ACT> __device__ __forceinline__ f32x2 wg_act2e(f32x2 x) {
    if constexpr (ACT == WG_ACT_GELU_ERF) {
        f32x2 x2 = x * x;
        x2 = (f32x2){fminf(x2.x, 49.0f), fminf(x2.y, 49.0f)};
        f32x2 p = x2 * 0.0010142630198970437f - 0.10677572339773178f;
        p = p * x2 - 2.301121234893799f;
        const f32x2 v = p * x;
        const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)} + 1.0f;      // (exp2 overflowing to +inf: rcp gives 0, the limit)
        return x * (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    } else {
        return wg_act2<ACT>(x);
    }
}

// four accumulator values (+ bias) -> activation -> packed bf16
// (EXACT: the A&S form of GELU -- the fp8 kernels keep it: their end-to-end mask test sits at the edge of its 1e-3 IoU bound, and a 2.6e-5 change
// in front of an e4m3 quantisation moves individual pixels)
template <int ACT, bool EXACT = false> __device__ __forceinline__ bf16x4 wg_epi_pack(f32x4 v, const float* b) {
    f32x2 lo = {v[0] + b[0], v[1] + b[1]}, hi = {v[2] + b[2], v[3] + b[3]};
    lo = EXACT ? wg_act2<ACT>(lo) : wg_act2e<ACT>(lo);
    hi = EXACT ? wg_act2<ACT>(hi) : wg_act2e<ACT>(hi);
    return (bf16x4){(bf16)lo.x, (bf16)lo.y, (bf16)hi.x, (bf16)hi.y};
}

// run BODY with `ACT` bound to the compile-time value of the (wave-uniform) runtime activation code
#define WG_ACT_SWITCH(act, ...)                                                                  \
    switch (act) {                                                                               \
        case WG_ACT_GELU_ERF: { constexpr int ACT = WG_ACT_GELU_ERF; __VA_ARGS__ } break;        \
        case WG_ACT_QUICK_GELU: { constexpr int ACT = WG_ACT_QUICK_GELU; __VA_ARGS__ } break;    \
        case WG_ACT_RELU: { constexpr int ACT = WG_ACT_RELU; __VA_ARGS__ } break;                \
        default: { constexpr int ACT = WG_ACT_NONE; __VA_ARGS__ } break;                         \
    }
