// Do a wave's MFMAs and its SIMD partner's VALU work overlap, and does it depend on WHERE the MFMA operands live (arch VGPRs vs AGPRs)?
// 512-thread workgroups (waves w and w + 4 share a SIMD), one per CU.  Waves 0-3 run `ma`, waves 4-7 run `mb`:
//   0 idle   1 MFMA 32x32x16 bf16, accumulators in arch VGPRs   2 the same, accumulators in AGPRs   3 accumulators AND A/B operands in AGPRs
//   4 VALU: v_exp_f32 + v_fma_f32 + v_add_f32 chain groups (the softmax mix)   5 VALU: v_fma_f32 only
// hipcc --offload-arch=gfx950 -O3 tools/micro/coexec_probe.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE> __device__ __forceinline__ float body(int iters, float seed) {
    if constexpr (MODE == 0) return 0.f;
    if constexpr (MODE == 1) {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        }
        return c0[0] + c1[1] + c2[2] + c3[3];
    }
    if constexpr (MODE == 2 || MODE == 3) {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
        if constexpr (MODE == 2) {
            for (int i = 0; i < iters; ++i)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                             "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                             : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
        } else {
            for (int i = 0; i < iters; ++i)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                             "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                             : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "a"(a), "a"(b));
        }
        return c0[0] + c1[1] + c2[2] + c3[3];
    }
    if constexpr (MODE == 4) {   // 16 independent chains: exp, fma, add per element per iteration
        float x[16];
        for (int e = 0; e < 16; ++e) x[e] = seed * 1e-3f + e * 1e-4f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float t = __builtin_amdgcn_exp2f(x[e]);
                t = __builtin_fmaf(t, 0.5f, -0.25f);
                x[e] = t + x[e] * 0.125f;
            }
        }
        float s = 0.f;
        for (int e = 0; e < 16; ++e) s += x[e];
        return s;
    }
    if constexpr (MODE == 5) {
        float x[16];
        for (int e = 0; e < 16; ++e) x[e] = seed * 1e-3f + e * 1e-4f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                x[e] = __builtin_fmaf(x[e], 0.999f, 0.001f);
                x[e] = __builtin_fmaf(x[e], 1.001f, -0.001f);
                x[e] = __builtin_fmaf(x[e], 0.998f, 0.002f);
            }
        }
        float s = 0.f;
        for (int e = 0; e < 16; ++e) s += x[e];
        return s;
    }
    return 0.f;
}

template <int MA, int MB> __global__ __launch_bounds__(512) void k(float* out, int iters_a, int iters_b) {
    const int wave = threadIdx.x >> 6;
    float r = wave < 4 ? body<MA>(iters_a, (float)threadIdx.x) : body<MB>(iters_b, (float)threadIdx.x);
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MA, int MB> float run(float* d, int ia, int ib) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MA, MB><<<256, 512>>>(d, ia, ib);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MA, MB><<<256, 512>>>(d, ia, ib);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1000.f;
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    const int IM = 20000;   // 80000 MFMAs per wave = 2.56 M cycles = ~1.2 ms
    const int IV = 6000;    // MODE 4: 16 x (exp 8 + fma 4 + fma 4) = 256 cycles per iteration -> ~1.5 M cycles
    const int IF = 8000;    // MODE 5: 48 fma x 4 = 192 cycles per iteration
    printf("alone:   mfma(vgpr acc) %.0f us | mfma(agpr acc) %.0f us | mfma(all agpr) %.0f us | valu(exp mix) %.0f us | valu(fma) %.0f us\n",
           run<1, 0>(d, IM, 0), run<2, 0>(d, IM, 0), run<3, 0>(d, IM, 0), run<4, 0>(d, IV, 0), run<5, 0>(d, IF, 0));
    printf("partner: mfma(vgpr acc) + valu(exp mix) %.0f us | mfma(agpr acc) + valu(exp mix) %.0f us | mfma(all agpr) + valu(exp mix) %.0f us\n",
           run<1, 4>(d, IM, IV), run<2, 4>(d, IM, IV), run<3, 4>(d, IM, IV));
    printf("partner: mfma(vgpr acc) + valu(fma) %.0f us | mfma(agpr acc) + valu(fma) %.0f us | mfma(all agpr) + valu(fma) %.0f us\n",
           run<1, 5>(d, IM, IF), run<2, 5>(d, IM, IF), run<3, 5>(d, IM, IF));
    printf("partner: mfma + mfma (vgpr) %.0f us | valu + valu (exp mix) %.0f us\n", run<1, 1>(d, IM, IM), run<4, 4>(d, IV, IV));
    return 0;
}
