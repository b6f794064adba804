// Shared device/host helpers for the walkgpt_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
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

// ---- device helpers --------------------------------------------------------
__device__ __forceinline__ float wg_bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 wg_f2bf(float x) { return (bf16)x; }

__device__ __forceinline__ float wg_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wg_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
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
