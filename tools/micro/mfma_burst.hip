// Micro-benchmark: cycles of ONE wave's burst of 64 v_mfma_f32_16x16x32_bf16 (the matrix half-phase of a 128 x 64 wave tile over a 64-deep slab),
// by operand order, priority and what the SIMD's other wave does meanwhile.  The persistent GEMMs measure 18.5 cycles per MFMA in such bursts
// (stamps in csrc/gemm_fr.hip) against 16.1 for a bare MFMA loop (tools/micro/dma_issue_cost.hip): which ingredient costs the difference?
//   ORDER 0: for ks, j, i : acc[i][j] += w[ks][j] . a[ks][i]   (the weight fragment stays for eight MFMAs)
//   ORDER 1: for ks, i, j                                      (the activation fragment stays for four)
//   ORDER 2: operands of a bare loop: acc[m & 31], a[m & 3], b[(m >> 2) & 3]
//   PRIO  0/1: s_setprio 1 around the burst
//   PARTNER 0: 4 waves per workgroup (one per SIMD)   1: 8 waves, waves 4-7 wait at the barrier during the burst (ping-pong of bursts)
//           2: 8 waves, both halves burst at the same time
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_burst.hip -o tools/micro/_bin/mfma_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define MFMA(acc, w, a) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a))
#define STAMP(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")

template <int ORDER, int PRIO, int PARTNER>
__global__ __launch_bounds__(PARTNER ? 512 : 256, PARTNER ? 2 : 1) void k(unsigned long long* out, const unsigned* seed, int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
    f32x4 acc[8][4];
    u32x4 a[2][8], w[2][4];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < 2; ++ks) {
        for (int i = 0; i < 8; ++i) a[ks][i] = *(const u32x4*)(seed + ((ks * 8 + i) * 64 + lane) * 4);
        for (int j = 0; j < 4; ++j) w[ks][j] = *(const u32x4*)(seed + ((16 + ks * 4 + j) * 64 + lane) * 4);
    }
    __syncthreads();
    unsigned long long sum = 0, t0, t1;
    if (PARTNER == 1 && grp == 1) asm volatile("s_barrier" ::: "memory");
    for (int it = 0; it < iters; ++it) {
        if (PARTNER == 1) asm volatile("s_barrier" ::: "memory");      // (the partner's burst)
        asm volatile("s_barrier" ::: "memory");
        STAMP(t0);
        if (PRIO) asm volatile("s_setprio 1");
        if constexpr (ORDER == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) MFMA(acc[i][j], w[ks][j], a[ks][i]);
        } else if constexpr (ORDER == 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) MFMA(acc[i][j], w[ks][j], a[ks][i]);
        } else {
#pragma unroll
            for (int m = 0; m < 64; ++m) MFMA(acc[(m & 31) >> 2][m & 3], a[0][m & 3], w[0][(m >> 2) & 3]);
        }
        if (PRIO) asm volatile("s_setprio 0");
        STAMP(t1);
        sum += t1 - t0;
    }
    if (PARTNER == 1 && grp == 0) asm volatile("s_barrier" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (lane == 0) out[blockIdx.x * 8 + wave] = sum;
}

static unsigned long long* g_out; static unsigned* g_seed; static float* g_sink;
template <int ORDER, int PRIO, int PARTNER> static void run(const char* what) {
    const int iters = 400;
    hipMemset(g_out, 0, 256 * 8 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<ORDER, PRIO, PARTNER>), dim3(256), dim3(PARTNER ? 512 : 256), 0, 0, g_out, g_seed, iters, g_sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), g_out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int i = 0; i < 256; ++i) { a.push_back((double)h[i * 8] / iters); b.push_back((double)h[i * 8 + (PARTNER ? 4 : 3)] / iters); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("order %d prio %d partner %d  %-70s cycles per 64-MFMA burst: wave 0 med %6.0f (%.2f / MFMA) | wave %d med %6.0f\n", ORDER, PRIO, PARTNER, what, a[128], a[128] / 64,
           PARTNER ? 4 : 3, b[128]);
    fflush(stdout);
}
int main() {
    hipMalloc(&g_out, 256 * 8 * 8); hipMalloc(&g_seed, 24 * 64 * 16); hipMalloc(&g_sink, 4096);
    std::vector<unsigned> s(24 * 64 * 4);
    for (auto& x : s) {      // random bf16 pairs of moderate magnitude
        const unsigned e0 = 0x3f00 + (rand() & 0xff), e1 = 0x3f00 + (rand() & 0xff);
        x = ((e0 | ((rand() & 1) << 15)) << 16) | (e1 | ((rand() & 1) << 15));
    }
    hipMemcpy(g_seed, s.data(), s.size() * 4, hipMemcpyHostToDevice);
    run<2, 0, 0>("bare-loop operands, one wave per SIMD");
    run<0, 0, 0>("weight fragment stays 8 MFMAs, one wave per SIMD");
    run<1, 0, 0>("activation fragment stays 4 MFMAs, one wave per SIMD");
    run<0, 1, 0>("weight fragment stays 8, setprio, one wave per SIMD");
    run<2, 0, 1>("bare-loop operands, partner parked at the barrier");
    run<0, 0, 1>("weight fragment stays 8, partner parked at the barrier");
    run<1, 0, 1>("activation fragment stays 4, partner parked at the barrier");
    run<0, 1, 1>("weight fragment stays 8, setprio, partner parked at the barrier");
    run<1, 1, 1>("activation fragment stays 4, setprio, partner parked at the barrier");
    run<0, 0, 2>("weight fragment stays 8, both waves of the SIMD burst together");
    run<2, 0, 2>("bare-loop operands, both waves of the SIMD burst together");
    return 0;
}
