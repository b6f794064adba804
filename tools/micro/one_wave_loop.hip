// Micro-benchmark / feasibility probe: a 256 x 256 bf16 GEMM loop with ONE wave per SIMD (4 waves per workgroup, wave tile 128 x 128, the 64
// accumulators = 256 registers in AGPRs, up to 512 registers per lane), every instruction placed by hand, operands as in the real kernels:
//   * slabs of 64 k-values (A rows 0..255 | W rows 0..255, 128-byte LDS rows, source-side XOR swizzle) brought in by LDS-DMA two slabs ahead
//     into two 64 KiB slots; a wave issues 16 of a slab's 64 pieces, spaced between its MFMAs;
//   * fragments double-buffered in registers per k-step (2 x 16 ds_read_b128): the reads of k-step n + 1 sit between the MFMAs of k-step n;
//   * ONE workgroup barrier per slab (slab s + 1 has landed everywhere / the slot of slab s is free);
//   * EPI = 1: at every tile seam (every TILE_SLABS slabs) the instruction mix of a bias epilogue for the wave's 64 accumulators
//     (v_accvgpr_read, packed add, bf16 pack, permlane swap, 16-byte stores to a scratch tile) is spread over the 256 MFMAs of the last slab
//     of the tile and the first slab of the next ("rolling" epilogue: an accumulator is drained behind its last MFMA and re-targeted with
//     C = 0 by the next tile).  Values are not meaningful; the instruction streams and memory operations are.
// Reported: cycles per slab (2048 = matrix-pipe-bound) with and without the seam work, against the production kernel's 2430-2530 in the loop
// plus 7 100 (bias) ... 15 400 (GELU) cycles of epilogue per tile with nothing beside it.
// hipcc --offload-arch=gfx950 -O3 tools/micro/one_wave_loop.hip -o tools/micro/_bin/one_wave_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct Args {
    const char* A;                 // 8 MiB panel (L2 / MALL resident after the first touch)
    char* C;                       // scratch output: [grid][256 x 256 bf16]
    unsigned long long* stamps;    // [grid][4]
    float* sink;
    int iters, tile_slabs;
};

#ifndef ABL
#define ABL 0      // timing-only ablations: 1 no LDS-DMA pieces, 2 no fragment reads, 4 no barrier, 8 no vmcnt wait, 16 no lgkmcnt waits
#endif
#ifndef BAR_AT
#define BAR_AT 36        // the slab's barrier sits behind this MFMA of k-step 0 (the reads of k-step 1's fragments are issued behind MFMAs 1, 3, .. 31)
#endif
#ifndef PIECE_EVERY
#define PIECE_EVERY 5    // a piece of slab it + 2 every this many MFMAs from the barrier on
#endif
constexpr int NP0 = (63 - BAR_AT) / PIECE_EVERY;     // pieces issued in what is left of k-step 0
#define MFMA(acc, w, a) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a))
#define MFMA0(acc, w, a) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(a))
template <int OFF> __device__ __forceinline__ void rd(u32x4& d, unsigned ad) {
    if (ABL & 2) { asm volatile("" : "+v"(d)); return; }
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(OFF));
}
__device__ __forceinline__ void dma(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned lds_dst) {
    if (ABL & 1) return;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}
// piece wave + 4 K of a slab: rows 8 (4 K mod 32) .. of A (K < 8) or W; two scalar adds (compile-time constants), M0, the load
template <int K> __device__ __forceinline__ void issue_piece(unsigned vbase, __amdgpu_buffer_rsrc_t rs, unsigned dbase, unsigned sbase) {
    constexpr unsigned SO = (unsigned)(((4 * K) & 31) * 8 * 1536 + (K >> 3) * 393216), DO = (unsigned)(K * 4096);
    if (ABL & 1) return;
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %4 offen lds" ::"v"(vbase), "s"(rs), "s"(dbase), "n"(DO), "s"(sbase + SO) : "memory", "scc");
}
template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

// the seam work of ONE accumulator (4 values of a row segment): read it out of the AGPRs, add a bias pair, pack to bf16
__device__ __forceinline__ void drain(const f32x4& acc, float b0, float b1, unsigned& lo, unsigned& hi) {
    float x0, x1, x2, x3;
    asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "a"(acc[0]), "a"(acc[1]), "a"(acc[2]), "a"(acc[3]));
    asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b0), "v"(b1));
    asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5" : "=v"(lo), "=v"(hi) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void k(Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, 8u << 20, 0x00020000);
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.C + (size_t)blockIdx.x * 131072), 0, 131072, 0x00020000);

    // fragment addresses: A(i, ks) = slot + vA[ks] + i * 2048, W(j, ks) = slot + vW[ks] + j * 2048
    const unsigned swz = (unsigned)((fr >> 1) & 7);
    unsigned vA[2], vW[2];
    for (int ks = 0; ks < 2; ++ks) {
        vA[ks] = lds0 + (unsigned)((wm * 128 + fr) * 128) + ((((unsigned)(ks * 4 + fq)) ^ swz) << 4);
        vW[ks] = lds0 + 32768u + (unsigned)((wn * 128 + fr) * 128) + ((((unsigned)(ks * 4 + fq)) ^ swz) << 4);
    }
    // DMA: piece p = wave + 4 k (k = 0..15): 8 rows of 128 bytes; rows of a 1536-byte-pitch panel, swizzled at the source
    const int r = lane >> 3, cs = (lane & 7) ^ ((((wave & 1) * 4) + (lane >> 4)) & 7);
    const unsigned vbase = (unsigned)(r * 1536 + cs * 16);
    // per slab: one scalar source base and one scalar LDS base; per piece two scalar adds (compile-time constants), M0, the load
    unsigned sbase = 0, dbase = 0;
    auto slab_base = [&](int slab, int slot) __attribute__((always_inline)) {
        const int panel = (blockIdx.x * 7 + slab / g.tile_slabs) & 7;                      // another row panel per tile
        sbase = __builtin_amdgcn_readfirstlane((unsigned)(panel * 1048576 + wave * 8 * 1536 + (slab % 12) * 128));
        dbase = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(slot * 65536 + wave * 1024));
    };
    auto issue = [&](auto kc) __attribute__((always_inline)) { issue_piece<decltype(kc)::value>(vbase, rs, dbase, sbase); };

    f32x4 acc[64];
    u32x4 F[2][16];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        asm volatile("v_accvgpr_write_b32 %0, 0\n\tv_accvgpr_write_b32 %1, 0\n\tv_accvgpr_write_b32 %2, 0\n\tv_accvgpr_write_b32 %3, 0" : "=a"(acc[i][0]), "=a"(acc[i][1]), "=a"(acc[i][2]), "=a"(acc[i][3]));
        (void)z;
    }
    // prologue: slabs 0 and 1 requested, slab 0 landed, fragments of (0, ks 0) read
    slab_base(0, 0);
    sfor<0, 16>([&](auto kc) { issue(kc); });
    slab_base(1, 1);
    sfor<0, 16>([&](auto kc) { issue(kc); });
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    sfor<0, 8>([&](auto ic) { constexpr int I = decltype(ic)::value; rd<I * 2048>(F[0][I], vA[0]); });
    sfor<0, 8>([&](auto jc) { constexpr int J = decltype(jc)::value; rd<J * 2048>(F[0][8 + J], vW[0]); });
    sfor<0, 16>([&](auto jc) { constexpr int J = decltype(jc)::value; F[1][J] = F[0][J]; });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float b0 = (float)lane * 0.001f, b1 = 0.5f;
    int seam = g.tile_slabs - 1;                  // the slab index (mod tile) that ends a tile
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    // the seam work of accumulators X and Y (compile-time indices) -> one 16-byte store
    auto seam_pair = [&](auto xc, auto yc, unsigned off) __attribute__((always_inline)) {
        constexpr int X = decltype(xc)::value, Y = decltype(yc)::value;
        unsigned lo0, hi0, lo1, hi1;
        drain(acc[X], b0, b1, lo0, hi0);
        drain(acc[Y], b0, b1, lo1, hi1);
        asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(hi0), "+v"(lo1), "+v"(lo0), "+v"(hi1));
        const u32x4 o = {lo0, hi0, lo1, hi1};
        __builtin_amdgcn_raw_buffer_store_b128(o, crs, off, 0, 0);
    };
    using std::integral_constant;
    // one slab; SEAM 0: loop only, 1: last slab of a tile (accumulators 0..31 drained behind their last MFMAs), 2: first slab of the next (32..63)
    auto slab = [&](int it, auto seamc) __attribute__((always_inline)) {
        constexpr int SEAM = decltype(seamc)::value;
        const int slot = it & 1;
        const unsigned sl = (unsigned)slot * 65536u, sn = (unsigned)(slot ^ 1) * 65536u;
        slab_base(it + 2, slot);
        // ---- k-step 0: MFMAs from F[0]; the reads of (it, ks 1) into F[1] behind every second MFMA of the first half
        sfor<0, 64>([&](auto mc) {
            constexpr int M = decltype(mc)::value;
            constexpr int J = M >> 3, I = M & 7;
            MFMA(acc[I * 8 + J], F[0][8 + J], F[0][I]);
            if constexpr ((M & 1) == 1 && M < 32) {
                constexpr int R = M >> 1;
                if constexpr (R < 8) rd<R * 2048>(F[1][8 + R], vW[1] + sl);       // (W fragments first: the next k-step starts with W_0)
                else rd<(R - 8) * 2048>(F[1][R - 8], vA[1] + sl);
            }
            if constexpr (SEAM != 0 && (M & 3) == 3) {
                constexpr int Q = M >> 2;             // 0..15
                constexpr int B0 = SEAM == 1 ? 0 : 32;
                const unsigned off = (unsigned)(((wm * 128 + (Q & 7) * 16 + fr) * 256 + wn * 128 + (Q >> 3) * 64) * 2 + fq * 16);
                seam_pair(integral_constant<int, B0 + Q>{}, integral_constant<int, (B0 + Q + 16) & 63>{}, off);
            }
            if constexpr (M == BAR_AT) {
                // the fragments of (it, ks 1) are in registers: the slot of slab it is free once every wave is here; this wave's pieces of slab
                // it + 1 (requested most of a slab ago) have landed
                if (!(ABL & 16)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!(ABL & 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SEAM != 0 ? (M + 1) / 4 : 0) : "memory");     // (the seam's stores are younger)
                if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
            }
            if constexpr (M > BAR_AT && ((M - BAR_AT) % PIECE_EVERY) == PIECE_EVERY / 2 && (M - BAR_AT) / PIECE_EVERY < NP0) issue(integral_constant<int, (M - BAR_AT) / PIECE_EVERY>{});
        });
        // ---- k-step 1: MFMAs from F[1]; 16 pieces of slab it + 2 into the slot just freed, the reads of (it + 1, ks 0) into F[0]
        sfor<0, 64>([&](auto mc) {
            constexpr int M = decltype(mc)::value;
            constexpr int J = M >> 3, I = M & 7;
            MFMA(acc[I * 8 + J], F[1][8 + J], F[1][I]);
            if constexpr ((M % PIECE_EVERY) == 0 && NP0 + M / PIECE_EVERY < 16) issue(integral_constant<int, NP0 + M / PIECE_EVERY>{});
            if constexpr ((M & 1) == 1 && M >= 16 && M < 48) {
                constexpr int R = (M - 16) >> 1;
                if constexpr (R < 8) rd<R * 2048>(F[0][8 + R], vW[0] + sn);
                else rd<(R - 8) * 2048>(F[0][R - 8], vA[0] + sn);
            }
            if constexpr (SEAM != 0 && (M & 3) == 3) {
                constexpr int Q = M >> 2;
                constexpr int B0 = SEAM == 1 ? 16 : 48;
                const unsigned off = (unsigned)(((wm * 128 + (Q & 7) * 16 + fr) * 256 + wn * 128 + (Q >> 3) * 64 + 32) * 2 + fq * 16);
                seam_pair(integral_constant<int, B0 + Q>{}, integral_constant<int, (B0 + Q + 16) & 63>{}, off);
            }
        });
        if (!(ABL & 16)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    const int ntiles = g.iters / g.tile_slabs;
    int it = 0;
    for (int t = 0; t < ntiles; ++t) {
        if constexpr (EPI) slab(it++, integral_constant<int, 2>{});
        else slab(it++, integral_constant<int, 0>{});
        for (int s2 = 2; s2 < g.tile_slabs; ++s2) slab(it++, integral_constant<int, 0>{});
        if constexpr (EPI) slab(it++, integral_constant<int, 1>{});
        else slab(it++, integral_constant<int, 0>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        float x;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[i][0]));
        s += x;
    }
    if (s == 12345.678f) g.sink[threadIdx.x] = s;
    if (lane == 0) g.stamps[blockIdx.x * 4 + wave] = c1 - c0;
}

static Args g_args;
template <int EPI> static void run(const char* what, int tile_slabs) {
    const int lds = 131072;
    g_args.tile_slabs = tile_slabs;
    hipFuncSetAttribute((const void*)k<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipMemset(g_args.stamps, 0, 256 * 4 * 8);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<EPI>), dim3(256), dim3(256), lds, 0, g_args);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); exit(1); }
    std::vector<unsigned long long> st(256 * 4);
    hipMemcpy(st.data(), g_args.stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> a;
    for (int i = 0; i < 256; ++i) a.push_back((double)st[i * 4] / (g_args.iters / tile_slabs * tile_slabs));
    std::sort(a.begin(), a.end());
    printf("epi %d tile %2d slabs  %-62s cycles/slab: med %6.0f (min %6.0f max %6.0f) -> per tile %7.0f\n", EPI, tile_slabs, what, a[128], a[0], a[255], a[128] * tile_slabs);
    fflush(stdout);
}

int main() {
    char *A, *C; unsigned long long* st; float* sink;
    hipMalloc(&A, 8u << 20); hipMalloc(&C, 256u * 131072u); hipMalloc(&st, 256 * 4 * 8); hipMalloc(&sink, 4096);
    {
        std::vector<unsigned> h(2 << 20);
        for (auto& x : h) x = ((unsigned)rand() * 2654435761u) & 0xBF7FBF7Fu;
        hipMemcpy(A, h.data(), 8u << 20, hipMemcpyHostToDevice);
    }
    g_args.A = A; g_args.C = C; g_args.stamps = st; g_args.sink = sink; g_args.iters = 600;
    printf("ABL=%d\n", ABL);
    run<0>("loop only (no seam work)", 12);
    if (ABL) return 0;
    run<1>("bias-epilogue instruction mix spread over the two seam slabs", 12);
    run<1>("the same, K = 1024 tiles", 16);
    run<1>("the same, K = 3072 tiles", 48);
    return 0;
}
