// Micro-benchmark: how many bytes per clock ONE compute unit can pull from L2 (and beyond) into LDS in the access pattern of the persistent
// 256 x 256 GEMM (walkgpt_amd/csrc/gemm.hip): 64-deep K slabs of an [M][K] and an [N][K] bf16 operand, 8 rows x 128 bytes per 1-KiB piece, tiles
// walked in the GEMM's XCD-aware order by 256 persistent workgroups.  Nothing is computed from the data: this is the ceiling of the operand
// fetch, the quantity profiles/r04_gemm_phases.md found at 26 B/clk/CU against the 32 B/clk a 256 x 256 tile needs to keep the matrix pipe busy.
//
// Variants (template parameters):
//   NLOAD     waves of the 512-thread workgroup that issue loads (1, 2, 4, 8); the others idle or run MFMAs (MFMA = 1: waves 4-7 issue
//             back-to-back v_mfma_f32_16x16x32_bf16 from registers for the whole kernel)
//   TRANSPORT 0 LDS-DMA (global_load_lds_dwordx4), 1 global_load_dwordx4 into registers (data dropped), 2 registers + ds_write_b128,
//             3 A by LDS-DMA and W through registers + ds_write_b128
//   INFLIGHT  pieces a loading wave keeps outstanding (counted vmcnt)
// hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_ceiling.hip -o tools/micro/_bin/fetch_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

struct Args {
    const char* A; const char* W;
    long lda, ldw;             // bytes per row
    int tiles_m, tiles_n, nk;  // 256 x 256 tiles, K / 64 slabs
    int col_block;
    int resident;              // 1: every tile reads panels (tm % 4, tn % 4): the whole working set sits in every XCD's L2
    unsigned long long* stamps;  // [grid][4]: cycles, 100 MHz ticks, bytes
    float* sink;
};

__device__ __forceinline__ void tile_of(int v, const Args& g, int& tm, int& tn) {
    const int nwg = g.tiles_m * g.tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    const int cb = g.col_block;
    if (cb <= 0 || cb >= g.tiles_n) { tm = wgid / g.tiles_n; tn = wgid % g.tiles_n; }
    else {
        const int per = g.tiles_m * cb, b = wgid / per, c0 = b * cb;
        const int w = (g.tiles_n - c0) < cb ? (g.tiles_n - c0) : cb;
        const int idx = wgid - b * per;
        tm = idx / w; tn = c0 + idx % w;
    }
    if (g.resident) { tm &= 3; tn &= 3; }
}

template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N_) : "memory"); }

template <int NLOAD, int TRANSPORT, int INFLIGHT, int MFMA>
__global__ __launch_bounds__(512) void fetch_k(Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    volatile int* flag = (volatile int*)(smem + 131072);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (wave >= NLOAD) {
        if (MFMA && wave >= 4) {
            f32x4 c[8];
            bf16x8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
            for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            while (*flag == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
            }
            float s = 0.f;
            for (int i = 0; i < 8; ++i) s += c[i][0];
            if (s == 12345.678f) g.sink[threadIdx.x] = s;
        }
        return;
    }
    // ---- loader wave: pieces p = wave, wave + NLOAD, ... of every slab's 64 (0-31: A rows 8p.., 32-63: W rows 8(p-32)..)
    constexpr int PPW = 64 / NLOAD;
    const int nwg = g.tiles_m * g.tiles_n;
    unsigned long long bytes = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    u32x4 hold[(TRANSPORT == 0) ? 1 : 8];
    for (auto& h : hold) h = (u32x4){0u, 0u, 0u, 0u};
    unsigned acc = 0;
    for (int v = blockIdx.x; v < nwg; v += gridDim.x) {
        int tm, tn;
        tile_of(v, g, tm, tn);
        // per-lane source of piece parity 0 / 1 (the swizzle of the GEMM: chunk c of row r sits in slot c ^ ((r >> 1) & 7))
        const char* baseA[2];
        const char* baseW[2];
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int r = par * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            baseA[par] = g.A + ((long)tm * 256 + r) * g.lda + c * 16;
            baseW[par] = g.W + ((long)tn * 256 + r) * g.ldw + c * 16;
        }
        for (int kt = 0; kt < g.nk; ++kt) {
            char* lds = smem + (kt & 1) * 65536;
#pragma unroll
            for (int u = 0; u < PPW; ++u) {
                const int p = wave + u * NLOAD;          // 0..63
                const bool isw = p >= 32;
                const int pp = p & 31;
                const char* src = (isw ? baseW[pp & 1] + (long)(pp >> 1) * 16 * g.ldw : baseA[pp & 1] + (long)(pp >> 1) * 16 * g.lda) + kt * 128;
                char* dst = lds + p * 1024;
                const bool dma = TRANSPORT == 0 || (TRANSPORT == 3 && !isw);
                if (dma) {
                    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst), 16, 0, 0);
                    if constexpr (TRANSPORT == 0) wait_vmcnt<INFLIGHT>();
                } else {
                    // register path, loads the compiler can see (it counts them itself): a slot's previous load is consumed, then the slot is loaded again;
                    // eight slots = eight loads in flight per wave.  (An asm load with a deferred wait is not safe across the loop's back edge: hipcc may
                    // copy the destination registers before the data has landed and hand the old registers to something else.)
                    const int slot = u & 7;
                    if (u >= 8 || kt > 0) {
                        if constexpr (TRANSPORT == 1) acc ^= hold[slot][0] ^ hold[slot][3];
                        else *(u32x4*)(lds + ((p * 1024 + 8192) & 65535) + lane * 16) = hold[slot];
                    }
                    hold[slot] = *(const u32x4*)src;
                }
            }
            bytes += (unsigned long long)PPW * 1024;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (TRANSPORT != 0) { for (int i = 0; i < 8; ++i) acc ^= hold[i][1]; }
    if (acc == 0x12345u) g.sink[threadIdx.x] = 1.f;
    if (wave == 0) {
        if (lane == 0) {
            g.stamps[blockIdx.x * 4 + 0] = c1 - c0;
            g.stamps[blockIdx.x * 4 + 1] = r1 - r0;
            g.stamps[blockIdx.x * 4 + 2] = bytes * NLOAD;
        }
        *flag = 1;
    }
}

static Args g_args;
static int g_grid = 256;
template <int NLOAD, int TRANSPORT, int INFLIGHT, int MFMA>
static void run(const char* shape, const char* what) {
    const int lds = 131072 + 64;
    hipFuncSetAttribute((const void*)fetch_k<NLOAD, TRANSPORT, INFLIGHT, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 6;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((fetch_k<NLOAD, TRANSPORT, INFLIGHT, MFMA>), dim3(g_grid), dim3(512), lds, 0, g_args);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fetch_k<NLOAD, TRANSPORT, INFLIGHT, MFMA>), dim3(g_grid), dim3(512), lds, 0, g_args);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<unsigned long long> st(g_grid * 4);
    hipMemcpy(st.data(), g_args.stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> bpc, gbs, ghz;
    double total = 0;
    for (int b = 0; b < g_grid; ++b) {
        const double cyc = (double)st[b * 4], ticks = (double)st[b * 4 + 1], by = (double)st[b * 4 + 2];
        if (cyc <= 0 || ticks <= 0) continue;
        bpc.push_back(by / cyc); gbs.push_back(by / (ticks * 10.0)); ghz.push_back(cyc / (ticks * 10.0)); total += by;
    }
    std::sort(bpc.begin(), bpc.end()); std::sort(gbs.begin(), gbs.end()); std::sort(ghz.begin(), ghz.end());
    const size_t m = bpc.size() / 2;
    printf("%-10s %-34s L%d inflight %2d mfma %d | %6.1f us | B/clk/CU med %5.1f (min %5.1f max %5.1f) | GB/s/CU med %5.1f | %4.2f GHz | chip %5.2f TB/s\n", shape, what,
           NLOAD, INFLIGHT, MFMA, ms * 1e3, bpc[m], bpc.front(), bpc.back(), gbs[m], ghz[m], total / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

static void set_shape(const char* A, const char* W, int M, int N, int K, int resident) {
    g_args.A = A; g_args.W = W; g_args.lda = (long)K * 2; g_args.ldw = (long)K * 2;
    g_args.tiles_m = M / 256; g_args.tiles_n = N / 256; g_args.nk = K / 64; g_args.resident = resident;
    const long panel = 256L * K * 2, wbytes = (long)N * K * 2;
    const int cb = (int)((3L << 19) / panel);
    g_args.col_block = (wbytes > (3L << 20) && cb >= 2 && cb < g_args.tiles_n) ? cb : 0;
}

int main(int argc, char** argv) {
    const size_t abytes = 32768ull * 3072 * 2, wbytes = 8192ull * 8192 * 2;
    char *A, *W; unsigned long long* st; float* sink;
    hipMalloc(&A, abytes); hipMalloc(&W, wbytes); hipMalloc(&st, 256 * 4 * 8); hipMalloc(&sink, 4096);
    // random bytes (the clock the chip holds depends on the data when MFMAs run beside the loads; the loads themselves do not care)
    {
        std::vector<unsigned> h(1 << 22);
        for (auto& x : h) x = (unsigned)rand() * 2654435761u;
        for (size_t o = 0; o < abytes; o += h.size() * 4) hipMemcpy(A + o, h.data(), std::min(h.size() * 4, abytes - o), hipMemcpyHostToDevice);
        for (size_t o = 0; o < wbytes; o += h.size() * 4) hipMemcpy(W + o, h.data(), std::min(h.size() * 4, wbytes - o), hipMemcpyHostToDevice);
    }
    g_args.stamps = st; g_args.sink = sink;
    struct Shape { const char* name; int M, N, K, resident; };
    const Shape shapes[] = {{"resident", 32768, 2304, 768, 1}, {"sam_qkv", 32768, 2304, 768, 0}, {"sam_lin2", 32768, 768, 3072, 0}, {"8k", 8192, 8192, 8192, 0}};
    for (const Shape& s : shapes) {
        set_shape(A, W, s.M, s.N, s.K, s.resident);
        run<8, 0, 8, 0>(s.name, "DMA");
        run<8, 0, 16, 0>(s.name, "DMA");
        run<8, 0, 32, 0>(s.name, "DMA");
        run<4, 0, 16, 0>(s.name, "DMA");
        run<4, 0, 32, 0>(s.name, "DMA");
        run<4, 0, 32, 1>(s.name, "DMA");
        run<2, 0, 32, 0>(s.name, "DMA");
        run<2, 0, 32, 1>(s.name, "DMA");
        run<1, 0, 32, 0>(s.name, "DMA");
        run<1, 0, 63, 0>(s.name, "DMA");
    }
    if (argc > 1)
    for (const Shape& s : shapes) {
        set_shape(A, W, s.M, s.N, s.K, s.resident);
        run<8, 1, 8, 0>(s.name, "registers (dropped)");
        run<4, 1, 8, 0>(s.name, "registers (dropped)");
        run<4, 1, 8, 1>(s.name, "registers (dropped)");
        run<8, 2, 8, 0>(s.name, "registers + ds_write_b128");
        run<4, 2, 8, 0>(s.name, "registers + ds_write_b128");
        run<4, 2, 8, 1>(s.name, "registers + ds_write_b128");
    }
    return 0;
}
