// Micro-benchmark: what an LDS-DMA piece (1 KiB per wave-instruction) costs the MFMA stream of the SIMD it is issued on.
// One 512- or 256-thread workgroup per CU, every CU busy.  An "iteration" is one 64-deep slab of a 256 x 256 bf16 tile: 2048 cycles of
// v_mfma_f32_16x16x32_bf16 per SIMD (128 MFMAs for one wave per SIMD, 64 each for two) and 64 pieces per CU.  Reported: cycles per iteration
// (median over workgroups, measured by wave 0 and by wave 4 / the last wave) against the 2048 of the matrix pipe.
//
// ROLE  0: every wave computes and issues its share of the pieces (64 / waves per iteration), evenly spaced between its MFMAs
//       1: (8 waves) waves 0-3 compute 128 MFMAs and issue nothing, waves 4-7 only load (16 pieces each): a dedicated loader beside each MFMA wave
//       2: (8 waves) every wave computes 64 MFMAs, only waves 4-7 issue (16 pieces each)
//       3: no pieces at all (the MFMA stream alone)
// FORM  0: global_load_lds_dwordx4 with a 64-bit per-lane address   1: the same with an SGPR base + 32-bit per-lane offset
//       2: buffer_load_dwordx4 ... offen lds (descriptor in SGPRs, 32-bit per-lane offset)
// hipcc --offload-arch=gfx950 -O3 tools/micro/dma_issue_cost.hip -o tools/micro/_bin/dma_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Args {
    const char* A;             // >= 4 MiB, L2-resident after the first touch
    unsigned long long* stamps;   // [grid][4]
    float* sink;
    int iters;
};

template <int FORM>
__device__ __forceinline__ void dma_piece(const char* gbase, unsigned voff, unsigned lds_dst, __amdgpu_buffer_rsrc_t rs) {
    if constexpr (FORM == 0) {
        const char* p = gbase + voff;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(lds_dst) : "memory");
    } else if constexpr (FORM == 1) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
    } else {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(lds_dst) : "memory");
    }
}

template <int WAVES, int ROLE, int FORM>
__global__ __launch_bounds__(WAVES * 64) void k(Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool computes = ROLE != 1 || wave < 4;
    const bool loads = ROLE == 0 || ((ROLE == 1 || ROLE == 2) && wave >= 4);
    constexpr int NMFMA = (WAVES == 4 || ROLE == 1) ? 128 : 64;          // per wave and iteration
    constexpr int NPIECE = ROLE == 0 ? 64 / WAVES : 16;                    // per loading wave and iteration
    // the wave's source rows: 8 rows x 128 B per piece out of a 1536-byte-pitch panel, as the GEMM's slabs
    const int r = lane >> 3, c = (lane & 7) ^ ((r >> 1) & 7);
    const unsigned voff0 = (unsigned)(((blockIdx.x & 3) * 256 + wave * 8 + r) * 1536 + c * 16);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, 4u << 20, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;   // (LDS addresses are 32-bit offsets)

    f32x4 acc[32];
    bf16x8 a[4], b[4];
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(0.37f * ((lane * 7 + e * 3 + i) % 11) - 1.7f); b[i][e] = (__bf16)(0.21f * ((lane * 5 + e + i * 2) % 13) - 1.2f); }
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < g.iters; ++it) {
        const unsigned kofs = (unsigned)(it % 12) * 128;
        if (computes) {
            constexpr int STEP = NMFMA / NPIECE;
#pragma unroll
            for (int m = 0; m < NMFMA; ++m) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 31]) : "v"(a[m & 3]), "v"(b[(m >> 2) & 3]));
                if (loads && (m % STEP) == STEP / 2) {
                    const int p = m / STEP;
                    dma_piece<FORM>(g.A, voff0 + kofs + (unsigned)(p & 3) * 64 * 1536, lds0 + ((it & 1) * 8 + (p & 7)) * 8192, rs);
                }
            }
            if (loads) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPIECE) : "memory");
        } else if (loads) {
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) {
                dma_piece<FORM>(g.A, voff0 + kofs + (unsigned)(p & 3) * 64 * 1536, lds0 + ((it & 1) * 8 + (p & 7)) * 8192, rs);
                // pace: the loader keeps one iteration's pieces in flight
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPIECE) : "memory");
            }
            // ... and does not run ahead of the MFMA waves: one barrier per iteration for everyone
        }
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) g.sink[threadIdx.x] = s;
    if (lane == 0 && (wave == 0 || wave == WAVES - 1)) g.stamps[blockIdx.x * 4 + (wave ? 1 : 0)] = c1 - c0;
}

static Args g_args;
template <int WAVES, int ROLE, int FORM>
static void run(const char* what) {
    const int lds = 131072;
    hipFuncSetAttribute((const void*)k<WAVES, ROLE, FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipMemset(g_args.stamps, 0, 256 * 4 * 8);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<WAVES, ROLE, FORM>), dim3(256), dim3(WAVES * 64), lds, 0, g_args);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(256 * 4);
    hipMemcpy(st.data(), g_args.stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int i = 0; i < 256; ++i) { a.push_back((double)st[i * 4] / g_args.iters); b.push_back((double)st[i * 4 + 1] / g_args.iters); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("waves %d role %d form %d  %-58s cycles/iteration: wave0 med %6.0f (min %6.0f max %6.0f) | last wave med %6.0f\n", WAVES, ROLE, FORM, what, a[128], a[0], a[255], b[128]);
    fflush(stdout);
}

int main() {
    char* A; unsigned long long* st; float* sink;
    hipMalloc(&A, 8u << 20); hipMalloc(&st, 256 * 4 * 8); hipMalloc(&sink, 4096);
    {
        std::vector<unsigned> h(2 << 20);
        for (auto& x : h) x = ((unsigned)rand() * 2654435761u) & 0xBF7FBF7Fu;
        hipMemcpy(A, h.data(), 8u << 20, hipMemcpyHostToDevice);
    }
    g_args.A = A; g_args.stamps = st; g_args.sink = sink; g_args.iters = 600;
    run<4, 3, 0>("1 wave/SIMD, MFMA only");
    run<8, 3, 0>("2 waves/SIMD, MFMA only");
    run<4, 0, 0>("1 wave/SIMD, 16 pieces per wave, 64-bit vaddr");
    run<4, 0, 1>("1 wave/SIMD, 16 pieces per wave, saddr + voffset");
    run<4, 0, 2>("1 wave/SIMD, 16 pieces per wave, buffer offen");
    run<8, 0, 0>("2 waves/SIMD, 8 pieces per wave, 64-bit vaddr");
    run<8, 0, 1>("2 waves/SIMD, 8 pieces per wave, saddr + voffset");
    run<8, 0, 2>("2 waves/SIMD, 8 pieces per wave, buffer offen");
    run<8, 1, 0>("MFMA wave + dedicated loader wave per SIMD, 64-bit vaddr");
    run<8, 1, 1>("MFMA wave + dedicated loader wave per SIMD, saddr + voffset");
    run<8, 1, 2>("MFMA wave + dedicated loader wave per SIMD, buffer offen");
    run<8, 2, 0>("2 compute waves/SIMD, waves 4-7 issue 16 each, 64-bit vaddr");
    run<8, 2, 2>("2 compute waves/SIMD, waves 4-7 issue 16 each, buffer offen");
    return 0;
}
