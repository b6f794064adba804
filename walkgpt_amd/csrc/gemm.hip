// bf16 GEMM with fused epilogue for gfx950:  C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ R)
//
// Replaces every nn.Linear / 1x1-conv / patchify-conv on the hot path (SURVEY.md §8a rows a1,a2,a4,a6,a7,
// a8,a9,a11,a13,a14): qkv/proj/MLP of the SAM and CLIP blocks, the neck convolutions (after im2row), the
// MSQP/CTP linears and the mask decoder linears.
//
// Design (MI355X_MICROARCH.md / cdna_hip_programming.md §5):
//   * both operands are K-contiguous ("B^T" form: nn.Linear weights are [N,K]) so A and W tiles are staged
//     the same way: 64-deep K slabs, 128-byte LDS rows, LDS-DMA (global_load_lds_dwordx4, 1 KiB per
//     wave-instruction = 8 rows), double-buffered.
//   * the LDS image is lane-linear (hardware constraint of LDS-DMA), so the bank-conflict swizzle is applied
//     to the per-lane SOURCE address and undone on the ds_read_b128: 16-byte chunk c of row r lives in slot
//     c ^ ((r >> 1) & 7).  16 rows x one chunk then cover all 16 slots of the 256-byte bank row.
//   * MFMA 16x16x32 bf16, operands swapped (W fragment as the A operand) so that each lane ends with four
//     consecutive N elements of one output row: the epilogue packs them into one 8-byte store and reads
//     bias / residual with the same shape.
//   * workgroup -> tile map is XCD-aware: blocks that share an XCD (same id mod 8) walk one contiguous
//     stripe of the tile grid, so the A row panel and the weight slab are re-read from that XCD's L2.
#include "wg_common.h"
#include "gemm_args.h"
#include <type_traits>

// Diagnostic build only (-DWG_GEMM_STAMP, tools/gemm_stamps.py): lane 0 of waves 0 and 4 of workgroup 0 records s_memtime at the
// half-phase boundaries of K slabs 2..9 of the ping-pong loop into the LDS bytes behind the two slabs.  No stamp in normal builds.
#ifdef WG_GEMM_STAMP
__device__ unsigned* wg_gemm_stamp_ptr = nullptr;
extern "C" int wg_debug_gemm_stamps(unsigned* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(wg_gemm_stamp_ptr), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#define WG_GSTAMP(k)                                                                                                              \
    do {                                                                                                                          \
        if (blockIdx.x == 0 && (wave & 3) == 0 && kt >= 2 && kt < 10) {                                                            \
            const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();                                                          \
            if (lane == 0) ((volatile unsigned*)(smem + 2 * STAGE))[(((wave >> 2) * 8 + (kt - 2)) * 4 + c) * 8 + (k)] = now;       \
        }                                                                                                                         \
    } while (0)
// tile-level stamps: [wave half][k]: 0 kernel entry, 1 first slab landed, 2 main loop done, 3 epilogue done
#define WG_TSTAMP(k)                                                                                                              \
    do {                                                                                                                          \
        if ((blockIdx.x == 0 || blockIdx.x == 1024) && (wave & 3) == 0) {                                                         \
            const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();                                                          \
            if (lane == 0 && wg_gemm_stamp_ptr) wg_gemm_stamp_ptr[512 + (blockIdx.x ? 16 : 0) + (wave >> 2) * 8 + (k)] = now;    \
        }                                                                                                                         \
    } while (0)
// persistent kernel (tools/gemm_phase_stamps.py): phase sums over all tiles of a workgroup in scalar registers (no store inside the loops);
// k = 0 tile start (first slab requested earlier), 1 first slab landed and barrier passed, 2 main loop done, 3 epilogue issued (stores may
// still drain under the next tile)
#define WG_PSTAMP(k)                                                                                                              \
    do {                                                                                                                          \
        unsigned long long now;                                                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");                                               \
        if ((k) > 0) pst_sum[(k) - 1] += now - pst_prev;                                                                          \
        pst_prev = now;                                                                                                           \
    } while (0)
// ... and inside the epilogue (relative to pst_e)
#define WG_ESTAMP(k)                                                                                                              \
    do {                                                                                                                          \
        unsigned long long now;                                                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");                                               \
        pst_sum[(k)] += now - pst_e;                                                                                              \
        pst_e = now;                                                                                                              \
    } while (0)
#else
#define WG_GSTAMP(k) do { } while (0)
#define WG_TSTAMP(k) do { } while (0)
#define WG_PSTAMP(k) do { } while (0)
#define WG_ESTAMP(k) do { } while (0)
#endif

template <int N_> __device__ __forceinline__ void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N_) : "memory"); }

// swizzle of 16-byte chunk slots inside a K slab row (BK*2 bytes): spreads 16 rows x one chunk over all 16 slots of
// the 256-byte LDS bank row for ds_read_b128.
// BK = 32 (64-byte rows, four rows per bank row): the 16x16x32 fragment read touches rows {R..R+3, R+12..R+15} at chunk c and
// rows {R+4..R+11} at chunk c+1 inside one 16-lane group; the permutation {0,2,3,1} of (row>>2)&3 sends those to 16
// distinct slots (the identity map would fold rows R and R+4 onto the same banks).
template <int BK> __device__ __forceinline__ int wg_swz(int row) {
    return BK == 64 ? (row >> 1) & 7 : (0x78 >> (2 * ((row >> 2) & 3))) & 3;
}

// Staged epilogue, memory side.  The wave's 64 x WTN staging slab (rows padded to SROW bytes) goes out as whole row
// segments, 16 bytes per lane; residual rows come in the same shape.  Both use raw buffer instructions so that rows past M are
// dropped (stores) or read as zero (loads) by the hardware range check instead of by per-lane branches, and
//   * all slab reads and residual adds happen BEFORE the first store, and
//   * the code between the first residual load and the last store is branch-free,
// because loads and stores share one in-order counter: a residual consumed between stores, or a store sequence cut by
// exec-mask branches, makes hipcc guard every store with `s_waitcnt vmcnt(0)` -- each store then waits until memory has
// acknowledged the previous one (microseconds per tile with the matrix pipe idle).
#define WG_RSRC_FLAGS 0x00020000   // raw buffer, 32-bit data format
template <int SROW, int CH, int NIT, bool HAS_R>
__device__ __forceinline__ void wg_flush_slab(const char* stg, const u32x4* rres, __amdgpu_buffer_rsrc_t crs, int ldc, int N, int row0, int nbase, int el) {
    constexpr int RPS = 64 / CH;
    bf16x8 o[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) o[it] = *(const bf16x8*)(stg + (it * RPS + el / CH) * SROW + (el % CH) * 16);
    if (HAS_R) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const bf16x8 r = __builtin_bit_cast(bf16x8, rres[it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[it][e] = (bf16)((float)o[it][e] + (float)r[e]);
        }
    }
    // pin the values here: otherwise LLVM sinks each add (and its wait) down to its store
    u32x4 t[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        t[it] = __builtin_bit_cast(u32x4, o[it]);
        asm volatile("" : "+v"(t[it]));
    }
    // columns past N: an offset beyond the descriptor's extent (< 2^31), so the range check drops those lanes too -- a branch
    // around the stores would make hipcc assume at the join that they may not have been issued and tighten every later wait
    const int n = nbase + (el % CH) * 8;
    const int off0 = n < N ? ((row0 + el / CH) * ldc + n) * 2 : (int)0x80000000;
#pragma unroll
    for (int it = 0; it < NIT; ++it) __builtin_amdgcn_raw_buffer_store_b128(t[it], crs, off0 + it * RPS * ldc * 2, 0, 0);
}
template <int CH, int NIT>
__device__ __forceinline__ void wg_load_residual(u32x4* rres, __amdgpu_buffer_rsrc_t rrs, int ldr, int res_mod, int row0, int nbase, int el) {
    constexpr int RPS = 64 / CH;
    const int n = nbase + (el % CH) * 8;
    // rows row0 .. row0 + 63 of the residual, modulo res_mod when it is set (a broadcast table: pos_embed, the decoder's positional rows).  The 64
    // rows wrap at most once when res_mod >= 64: ONE scalar modulo for the slab's first row and a compare + subtract per load, instead of a vector
    // integer division per load behind a branch (round 6: ~400 vector instructions per tile of every residual GEMM, whether or not res_mod was set)
    const int base = res_mod > 0 ? __builtin_amdgcn_readfirstlane(row0) % res_mod : row0;
    int off = ((base + el / CH) * ldr + n) * 2;
    if (res_mod >= 64 || res_mod <= 0) {
        const int wrap = res_mod > 0 ? res_mod - (base + el / CH) : 0x7FFFFFFF;      // rows until this lane's row index wraps
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int o = off + it * RPS * ldr * 2;
            rres[it] = __builtin_amdgcn_raw_buffer_load_b128(rrs, it * RPS >= wrap ? o - res_mod * ldr * 2 : o, 0, 0);
        }
    } else {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int m = row0 + it * RPS + el / CH;
            rres[it] = __builtin_amdgcn_raw_buffer_load_b128(rrs, ((m % res_mod) * ldr + n) * 2, 0, 0);
        }
    }
}

typedef __attribute__((ext_vector_type(8))) int i32x8;

// Experiment knobs of the persistent 256x256 kernel (diagnostic builds only, tools/build_variant.py): cache policy bits of its output
// stores and of its operand loads (gfx950 aux: 1 = sc0, 2 = nt, 16 = sc1).  Default 0 = the plain policy.
// WG_GEMM_TAIL (persistent bf16 kernel): MFMAs of a cluster issued BEHIND its hand-over barrier (0: the barrier closes the cluster)
#ifndef WG_GEMM_TAIL
#define WG_GEMM_TAIL 0
#endif
#ifndef WG_GEMM_C_AUX
#define WG_GEMM_C_AUX 0
#endif
#ifndef WG_GEMM_A_AUX
#define WG_GEMM_A_AUX 0
#endif
#ifndef WG_GEMM_W_AUX
#define WG_GEMM_W_AUX 0
#endif
__device__ __forceinline__ void wg_mfma16_acc(f32x4& acc, bf16x8 a, bf16x8 b) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); }

// FP8 = true: the same kernel on e4m3 (OCP) operands.  A K slab is still 128 bytes per row -- 128 fp8 values instead of 64 bf16 -- so
// staging, swizzle, barriers and the ping-pong schedule are unchanged; a fragment is the two 16-byte chunks k = 16*fq .. 16*fq+15
// and k = 64+16*fq .. 64+16*fq+15 of its row -- the K order of v_mfma_scale_f32_16x16x128_f8f6f4's operand registers, measured with
// per-block scales (tools/micro/mx_dbg.py: with 32 contiguous bytes per lane half of every block took the scale of another one):
// lane group q supplies the E8M0 scale of K block q = k 32q .. 32q+31.  ONE such MFMA replaces the two bf16 16x16x32 steps of a
// slab at the same matrix-pipe time for twice the K.  Here both block scales are 1.0 (E8M0 127): the operands carry fp32 scales per
// row (A) and per output channel (W), applied to the accumulators before the epilogue.  The persistent kernel
// (wg_gemm_pp_persist_kernel<.., FP8>) feeds real MX block scales to the same instruction.
template <int BM, int BN, int BK, int STAGES, int WM, int WN, bool STAGED, int PIPE, bool FP8 = false>
__global__ __launch_bounds__(WM* WN * 64, ((BM / WM) * (BN / WN) / 64 > 192 ? 1 : 2)) void wg_gemm_kernel(GemmArgs g) {
    static_assert(PIPE == 0 || (PIPE == 2 && BK == 64 && STAGES == 2), "the ping-pong schedule is written for two 64-deep slabs");
    static_assert(!FP8 || PIPE == 2, "fp8 operands run on the ping-pong schedule only");
    static_assert((BM / WM) % 64 == 0, "the staged epilogue walks the wave tile 64 rows at a time");
    static_assert(BK == 32 || BK == 64, "K slab depth");
    constexpr int NT = WM * WN * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FI = WTM / 16, FJ = WTN / 16;
    constexpr int ROWB = BK * 2;                  // bytes per LDS row
    constexpr int CPR = BK / 8;                   // 16-byte chunks per row
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int RPI = 64 / CPR;                 // rows per LDS-DMA wave-instruction
    constexpr int ROWS_PER_ROUND = (NT / 64) * RPI;
    constexpr int G = (BM + BN) / ROWS_PER_ROUND; // LDS-DMA instructions per wave per stage
    static_assert(BM % ROWS_PER_ROUND == 0 && BN % ROWS_PER_ROUND == 0, "tile rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware, bijective remap of the linear block id (cdna_hip_programming.md §5 "XCD swizzle").
    const int nwg = g.tiles_m * g.tiles_n;
    int wgid;
    {
        const int orig = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int tile_m, tile_n;
    wg_tile_of(wgid, g.tiles_m, g.tiles_n, g.col_block, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // per-lane source pointers of this wave's LDS-DMA pieces (K offset 0); they advance by BK elements per stage
    const bf16* srcA[BM / ROWS_PER_ROUND];
    const bf16* srcW[BN / ROWS_PER_ROUND];
#pragma unroll
    for (int i = 0; i < BM / ROWS_PER_ROUND; ++i) {
        const int r = i * ROWS_PER_ROUND + wave * RPI + lane / CPR;
        const int c = (lane % CPR) ^ wg_swz<BK>(r);
        int gr = m0 + r;
        gr = gr < g.M ? gr : g.M - 1;
        srcA[i] = g.A + (long)gr * g.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < BN / ROWS_PER_ROUND; ++i) {
        const int r = i * ROWS_PER_ROUND + wave * RPI + lane / CPR;
        const int c = (lane % CPR) ^ wg_swz<BK>(r);
        int gr = n0 + r;
        gr = gr < g.N ? gr : g.N - 1;
        srcW[i] = g.W + (long)gr * g.ldw + c * 8;
    }

    auto stage = [&](int kt, int buf) {
        char* ldsA = smem + buf * STAGE;
        char* ldsW = ldsA + BM * ROWB;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < BM / ROWS_PER_ROUND; ++i)
            __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(srcA[i] + k0), WG_LDS_PTR(ldsA + (i * ROWS_PER_ROUND + wave * RPI) * ROWB), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BN / ROWS_PER_ROUND; ++i)
            __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(srcW[i] + k0), WG_LDS_PTR(ldsW + (i * ROWS_PER_ROUND + wave * RPI) * ROWB), 16, 0, 0);
    };

    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (PIPE == 2) {
        // ---- ping-pong between the two waves of every SIMD (waves w and w+4 = the two M halves of the tile) -------------
        // A slab's 64 MFMAs per wave are cut into two clusters of 32 (64 rows x 64 columns x K 64).  Every cluster is two
        // half-phases separated by s_barrier:  M = this cluster's ds_reads + the next slab's LDS-DMA pieces, retire them;
        // C = the 32 MFMAs.  The bottom-half waves run one half-phase behind the top-half ones (one extra barrier up front,
        // one at the end for the others), so on each SIMD one wave is always in C while its partner is in M: the matrix
        // pipe sees back-to-back clusters and LDS / DMA latency sits under them.
        // Why two clusters and not four (the 8-phase template of cdna_hip_programming.md): in-kernel stamps
        // (tools/gemm_stamps.py) showed every s_barrier releasing ~125 cycles after its last arrival; eight barriers per slab
        // were ~20 % of the loop.  Both W halves were already resident, so merging the two clusters that share an A half
        // costs no registers: +2..10 % (8192^3: 1071 -> 1183 TFLOP/s, SAM lin2 899 -> 979).
        // Hazards: reads are retired (lgkmcnt(0)) inside their own M half-phase, so a buffer is re-staged at the earliest
        // one barrier after its last read was complete.  Slab kt+1 is sent during slab kt: W and the early A rows (6 pieces
        // per wave) in the first M, the late A rows (2 pieces) in the second; the counted waits leave exactly the pieces
        // that are not needed by the next M in flight (memory operations retire in issue order).
        static_assert(FI == 8 && FJ == 4, "written for 128x64 wave tiles");
        static_assert(BM / ROWS_PER_ROUND == 4 && BN / ROWS_PER_ROUND == 4, "piece schedule assumes four 64-row rounds per operand");
        const int grp = wm;
        bf16x8 af[4][2], wf2[2][2][2];
        auto read_a = [&](const char* ldsA, int ci) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int r = wm * WTM + (4 * ci + i) * 16 + fr;
                    af[i][ks] = *(const bf16x8*)(ldsA + r * ROWB + (((ks * 4 + fq) ^ wg_swz<BK>(r)) << 4));
                }
        };
        auto read_w = [&](const char* ldsW, int cj) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int r = wn * WTN + (2 * cj + j) * 16 + fr;
                    wf2[cj][j][ks] = *(const bf16x8*)(ldsW + r * ROWB + (((ks * 4 + fq) ^ wg_swz<BK>(r)) << 4));
                }
        };
        auto piece = [&](int kt, int which) {   // which: 0,1 = W round pairs; 2 = A rounds 0,2 (early); 3 = A rounds 1,3 (late)
            char* ldsA = smem + (kt & 1) * STAGE;
            char* ldsW = ldsA + BM * ROWB;
            const int k0 = kt * BK;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (which < 2) {
                    const int i = which * 2 + u;
                    __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(srcW[i] + k0), WG_LDS_PTR(ldsW + (i * ROWS_PER_ROUND + wave * RPI) * ROWB), 16, 0, 0);
                } else {
                    const int i = (which - 2) + 2 * u;
                    __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(srcA[i] + k0), WG_LDS_PTR(ldsA + (i * ROWS_PER_ROUND + wave * RPI) * ROWB), 16, 0, 0);
                }
            }
        };
        // first slab in the order the loop consumes it; its late A rows (the two youngest pieces) may still be in flight when
        // cluster 0 starts -- the loop's own counted wait in its first M half-phase covers them
        WG_TSTAMP(0);
        const unsigned sw1 = 0x7F7F7F7Fu;   // both operands' block scales: 1.0 (E8M0 127); the fp32 per-row / per-channel scales follow in the epilogue
        piece(0, 0); piece(0, 1); piece(0, 2); piece(0, 3);
        wg_wait_vmcnt<2>();
        __builtin_amdgcn_s_barrier();
        WG_TSTAMP(1);
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const char* ldsA = smem + (kt & 1) * STAGE;
            const char* ldsW = ldsA + BM * ROWB;
            const bool more = kt + 1 < nk;
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {
                const int c = sc;   // (stamp index)
                // ---- M half-phase: 16 reads + 6 pieces, then 8 reads + 2 pieces
                WG_GSTAMP(0);
                read_a(ldsA, sc);
                if (sc == 0) { read_w(ldsW, 0); read_w(ldsW, 1); }
                if (more) {
                    if (sc == 0) { piece(kt + 1, 0); piece(kt + 1, 1); piece(kt + 1, 2); }
                    else piece(kt + 1, 3);
                }
                // sc 0: the late A rows of THIS slab (sent in the previous slab's sc 1) must be in before sc 1 reads them;
                // sc 1: everything of slab kt+1 except the two pieces just sent must be in before its sc 0
                if (sc == 0) { if (more) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
                else { if (more) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
                __builtin_amdgcn_sched_barrier(0);
                WG_GSTAMP(1);
                __builtin_amdgcn_s_barrier();
                WG_GSTAMP(2);
                // ---- C half-phase: 64 rows x 64 columns x K 64
                __builtin_amdgcn_s_setprio(1);
                if constexpr (FP8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4 a0 = __builtin_bit_cast(u32x4, af[i][0]), a1 = __builtin_bit_cast(u32x4, af[i][1]);
                        const i32x8 a8 = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const u32x4 w0 = __builtin_bit_cast(u32x4, wf2[j >> 1][j & 1][0]), w1 = __builtin_bit_cast(u32x4, wf2[j >> 1][j & 1][1]);
                            const i32x8 w8 = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
                            // Volatile asm, not the builtin: as a pure value the builtin's cluster 0 was sunk by hipcc below cluster 1's M
                            // half-phase (both clusters back to back, their reads uncovered).
                            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3" : "+v"(acc[4 * sc + i][j]) : "v"(w8), "v"(a8), "v"(sw1));
                        }
                    }
                } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[4 * sc + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[j >> 1][j & 1][ks], af[i][ks], acc[4 * sc + i][j], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                WG_GSTAMP(3);
                __builtin_amdgcn_s_barrier();
                WG_GSTAMP(4);
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();   // last half-phase: only the lagging group works
#ifdef WG_GEMM_STAMP
        __syncthreads();
        if (blockIdx.x == 0 && wg_gemm_stamp_ptr)
            for (int i = tid; i < 2 * 8 * 4 * 8; i += NT) wg_gemm_stamp_ptr[i] = ((unsigned*)(smem + 2 * STAGE))[i];
#endif
    } else {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) stage(s, s);

    int buf = 0, nbuf = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt must have landed; up to STAGES-2 younger stages may stay in flight across the barrier
        // (counted vmcnt + raw s_barrier: a __syncthreads() here would drain the LDS-DMA queue, cdna_hip_programming.md §5)
        const int younger = nk - 1 - kt;
        if (STAGES >= 4 && younger >= 2) wg_wait_vmcnt<2 * G>();
        else if (STAGES >= 3 && younger >= 1) wg_wait_vmcnt<G>();
        else wg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + STAGES - 1 < nk) stage(kt + STAGES - 1, nbuf);
        const char* ldsA = smem + buf * STAGE;
        const char* ldsW = ldsA + BM * ROWB;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[FI], wf[FJ];
            const int c = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int r = wm * WTM + i * 16 + fr;
                af[i] = *(const bf16x8*)(ldsA + r * ROWB + ((c ^ wg_swz<BK>(r)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const int r = wn * WTN + j * 16 + fr;
                wf[j] = *(const bf16x8*)(ldsW + r * ROWB + ((c ^ wg_swz<BK>(r)) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        buf = buf + 1 == STAGES ? 0 : buf + 1;
        nbuf = nbuf + 1 == STAGES ? 0 : nbuf + 1;
    }
    }
    __syncthreads();  // every wave is done with the last K slab before the staging slabs reuse the LDS
    if constexpr (PIPE == 2) { WG_TSTAMP(2); }

    // ---- epilogue -------------------------------------------------------------------------------------------------------
    // acc[i][j][e]: output row m = .. + i*16 + (lane&15), column n = .. + j*16 + (lane>>4)*4 + e.
    const int nbase = n0 + wn * WTN;
    if constexpr (FP8) {
        // dequantise: C[m][n] = scale_a[m] * scale_w[n] * sum_k Aq[m][k] Wq[n][k]
        float sw[FJ][4];
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = nbase + j * 16 + fq * 4 + e;
                sw[j][e] = g.scale_w[n < g.N ? n : g.N - 1];
            }
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int m = m0 + wm * WTM + i * 16 + fr;
            const float sa = g.scale_a[m < g.M ? m : g.M - 1];
#pragma unroll
            for (int j = 0; j < FJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] *= sa * sw[j][e];
        }
    }
    if (STAGED) {
        // bf16 output: bias + activation in registers, then the wave's sub-tile goes through a private LDS slab
        // (64 rows at a time, rows padded by 16 bytes) so that residual loads and output stores are whole 128-byte
        // row segments, 16 bytes per lane, instead of 8-byte pieces of 16 different rows.
        constexpr int SROW = WTN * 2 + 16;
        constexpr int CH = WTN / 8;          // 16-byte chunks per row
        constexpr int RPI = 64 / CH;         // rows per wave-instruction
        char* stg = smem + wave * (64 * SROW);
        float bv[FJ][4];
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int n = nbase + j * 16 + fq * 4;
            if (g.bias && n < g.N) {
                const bf16x4 b = *(const bf16x4*)(g.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[j][e] = (float)b[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[j][e] = 0.f;
            }
        }
        // One straight-line copy of the rest per (residual? yes/no): with the residual handled by `if (g.R)` inside a common
        // body, hipcc cannot tell at the control-flow joins that no load is pending and guards the second slab's stores with
        // `s_waitcnt vmcnt(0)` -- which waits for the first slab's stores to be acknowledged.
        const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, g.c_bytes, WG_RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)g.R, 0, g.r_bytes, WG_RSRC_FLAGS);
        auto finish = [&](auto has_r) __attribute__((always_inline)) {
            constexpr bool HAS_R = decltype(has_r)::value;
            // residual rows of the whole wave tile: issued first (a load issued behind stores could only be waited for
            // together with them), consumed after the LDS round trips
            u32x4 rres[WTM / 64][64 / RPI];
            if (HAS_R) {
#pragma unroll
                for (int half = 0; half < WTM / 64; ++half)
                    wg_load_residual<CH, 64 / RPI>(rres[half], rrs, (int)g.ldr, g.res_mod, m0 + wm * WTM + half * 64, nbase, lane);
            }
#pragma unroll
            for (int half = 0; half < WTM / 64; ++half) {
                WG_ACT_SWITCH(g.act,
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) {
                        _Pragma("unroll") for (int j = 0; j < FJ; ++j)
                            *(bf16x4*)(stg + (i * 16 + fr) * SROW + (j * 16 + fq * 4) * 2) = wg_epi_pack<ACT, FP8>(acc[half * 4 + i][j], bv[j]);
                    })
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                wg_flush_slab<SROW, CH, 64 / RPI, HAS_R>(stg, rres[half], crs, (int)g.ldc, g.N, m0 + wm * WTM + half * 64, nbase, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        };
        // (Measured alternative: no LDS round trip -- one v_permlane16_swap between the packed halves of fragments j and j+1 gives
        // every lane 16 contiguous bytes, stored as 64-byte row segments.  Correct, but 3-6 % slower than this staged form on every
        // shape: half-line stores cost more than the LDS transpose.)
        if (g.R) finish(std::true_type{}); else finish(std::false_type{});
        if constexpr (PIPE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); WG_TSTAMP(3); }
        return;
    }
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        const int m = m0 + wm * WTM + i * 16 + fr;
        if (m >= g.M) continue;
        const long rrow = g.R ? (long)(g.res_mod > 0 ? m % g.res_mod : m) * g.ldr : 0;
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int n = nbase + j * 16 + fq * 4;
            if (n >= g.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (g.bias) {
                const bf16x4 b = *(const bf16x4*)(g.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)b[e];
            }
            if (g.act != WG_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = wg_act(v[e], g.act);
            }
            if (g.R) {
                const bf16x4 rr = *(const bf16x4*)(g.R + rrow + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
            }
            if (g.out_f32) {
                *(f32x4*)((float*)g.C + (long)m * g.ldc + n) = (f32x4){v[0], v[1], v[2], v[3]};
            } else {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
                *(bf16x4*)((bf16*)g.C + (long)m * g.ldc + n) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent-tile variant (bf16 output, staged epilogue).  One workgroup per CU walks tiles v, v+grid, v+2*grid ...
// (same XCD for all of them, consecutive tiles of an XCD stay neighbours).  Per tile the fixed costs of the plain
// kernel -- first-slab latency (~2.5 us) and the store-issue-bound epilogue (~5 us of a ~30 us tile at K = 768) --
// are taken off the matrix pipe's critical path:
//   * bias for the tile is loaded before its main loop, the residual rows of the first 64-row slab during the last K
//     slab, so the epilogue starts without a memory wait;
//   * after the last slab's barrier the NEXT tile's first slab is sent by LDS-DMA into buffer 0 while the epilogue
//     stages through buffer 1's region (+ one 9 KiB extension for wave 7);
//   * the next main loop starts behind `s_waitcnt vmcnt(#stores)`: memory operations retire in issue order, so this
//     waits for the slab (older) and leaves the epilogue's 16-byte output stores (younger) draining in the background.
// ---------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64, 2) void wg_gemm_persist_kernel(GemmArgs g) {
    constexpr int BK = 64;
    constexpr int NT = WM * WN * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FI = WTM / 16, FJ = WTN / 16;
    constexpr int ROWB = BK * 2;
    constexpr int CPR = BK / 8;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int RPI = 64 / CPR;
    constexpr int ROWS_PER_ROUND = (NT / 64) * RPI;
    constexpr int SROW = WTN * 2 + 16;
    constexpr int CH = WTN / 8;
    constexpr int RPS = 64 / CH;                         // rows per store instruction
    constexpr int NSTORE = (WTM / 64) * (64 / RPS);      // output store instructions per wave per tile
    constexpr int SLAB = 64 * SROW;                      // per-wave staging bytes
    constexpr int FIT = STAGE / SLAB;                    // waves whose staging slab fits inside buffer 1's region
    static_assert(WTM % 64 == 0, "staged epilogue walks 64 rows at a time");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    char* stg = wave < FIT ? smem + STAGE + wave * SLAB : smem + 2 * STAGE + (wave - FIT) * SLAB;

    const int nwg = g.tiles_m * g.tiles_n;
    const int nk = g.K / BK;
    auto tile_of = [&](int v, int& m0, int& n0) {
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        m0 = (wgid / g.tiles_n) * BM;
        n0 = (wgid % g.tiles_n) * BN;
    };

    // element offsets (32-bit: operands are < 2^31 elements, checked by the launcher) of this wave's LDS-DMA pieces
    unsigned offA[BM / ROWS_PER_ROUND], offW[BN / ROWS_PER_ROUND];
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < BM / ROWS_PER_ROUND; ++i) {
            const int r = i * ROWS_PER_ROUND + wave * RPI + lane / CPR;
            const int c = (lane % CPR) ^ wg_swz<BK>(r);
            int gr = m0 + r;
            gr = gr < g.M ? gr : g.M - 1;
            offA[i] = (unsigned)((long)gr * g.lda + c * 8);
        }
#pragma unroll
        for (int i = 0; i < BN / ROWS_PER_ROUND; ++i) {
            const int r = i * ROWS_PER_ROUND + wave * RPI + lane / CPR;
            const int c = (lane % CPR) ^ wg_swz<BK>(r);
            int gr = n0 + r;
            gr = gr < g.N ? gr : g.N - 1;
            offW[i] = (unsigned)((long)gr * g.ldw + c * 8);
        }
    };
    auto stage = [&](int kt, int buf) {
        char* ldsA = smem + buf * STAGE;
        char* ldsW = ldsA + BM * ROWB;
        const bf16* baseA = g.A + kt * BK;
        const bf16* baseW = g.W + kt * BK;
#pragma unroll
        for (int i = 0; i < BM / ROWS_PER_ROUND; ++i)
            __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(baseA + offA[i]), WG_LDS_PTR(ldsA + (i * ROWS_PER_ROUND + wave * RPI) * ROWB), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BN / ROWS_PER_ROUND; ++i)
            __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(baseW + offW[i]), WG_LDS_PTR(ldsW + (i * ROWS_PER_ROUND + wave * RPI) * ROWB), 16, 0, 0);
    };

    int v = blockIdx.x;
    int m0, n0;
    tile_of(v, m0, n0);
    set_sources(m0, n0);
    stage(0, 0);
    bool stores_in_flight = false;

    while (true) {
        const int nbase = n0 + wn * WTN;
        // bias of this tile, kept packed; loaded while the wave has to wait for its first slab anyway
        bf16x4 bvp[FJ];
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int n = nbase + j * 16 + fq * 4;
            if (g.bias && n < g.N) bvp[j] = *(const bf16x4*)(g.bias + n);
            else bvp[j] = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        }
        f32x4 acc[FI][FJ];
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt) {
            if (kt == 0 && stores_in_flight) wg_wait_vmcnt<NSTORE>();   // slab 0 landed; the previous tile's stores may still drain
            else wg_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
            const char* ldsA = smem + (kt & 1) * STAGE;
            const char* ldsW = ldsA + BM * ROWB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[FI], wf[FJ];
                const int c = ks * 4 + fq;
#pragma unroll
                for (int i = 0; i < FI; ++i) {
                    const int r = wm * WTM + i * 16 + fr;
                    af[i] = *(const bf16x8*)(ldsA + r * ROWB + ((c ^ wg_swz<BK>(r)) << 4));
                }
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    const int r = wn * WTN + j * 16 + fr;
                    wf[j] = *(const bf16x8*)(ldsW + r * ROWB + ((c ^ wg_swz<BK>(r)) << 4));
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of the last slab is done
        __builtin_amdgcn_s_barrier();

        // next tile: coordinates, sources, first slab on its way before the epilogue touches memory
        const int vn = v + gridDim.x;
        const bool has_next = vn < nwg;
        const int cm0 = m0;
        // opaque copies of the lane coordinates: keeps hipcc from hoisting the epilogue's (tile-invariant) address
        // arithmetic out of the tile loop, where it would stay live across the main loop and push it into spills
        int el = lane, efr = fr, efq = fq;
        asm volatile("" : "+v"(el), "+v"(efr), "+v"(efq));
        if (has_next) {
            tile_of(vn, m0, n0);
            set_sources(m0, n0);
            stage(0, 0);
        }

        const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, g.c_bytes, WG_RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)g.R, 0, g.r_bytes, WG_RSRC_FLAGS);
        auto finish = [&](auto has_r) __attribute__((always_inline)) {   // see wg_gemm_kernel's staged epilogue
            constexpr bool HAS_R = decltype(has_r)::value;
            u32x4 rres[WTM / 64][64 / RPS];   // (ordinary loads: hipcc waits for the slab in flight too before their first use)
            if (HAS_R) {
#pragma unroll
                for (int half = 0; half < WTM / 64; ++half)
                    wg_load_residual<CH, 64 / RPS>(rres[half], rrs, (int)g.ldr, g.res_mod, cm0 + wm * WTM + half * 64, nbase, el);
            }
#pragma unroll
            for (int half = 0; half < WTM / 64; ++half) {
                WG_ACT_SWITCH(g.act,
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) {
                        _Pragma("unroll") for (int j = 0; j < FJ; ++j) {
                            const float b[4] = {(float)bvp[j][0], (float)bvp[j][1], (float)bvp[j][2], (float)bvp[j][3]};
                            *(bf16x4*)(stg + (i * 16 + efr) * SROW + (j * 16 + efq * 4) * 2) = wg_epi_pack<ACT>(acc[half * 4 + i][j], b);
                        }
                    })
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                wg_flush_slab<SROW, CH, 64 / RPS, HAS_R>(stg, rres[half], crs, (int)g.ldc, g.N, cm0 + wm * WTM + half * 64, nbase, el);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        };
        if (g.R) finish(std::true_type{}); else finish(std::false_type{});
        if (!has_next) break;
        v = vn;
        // every wave issues all NSTORE store instructions of a tile (rows / columns outside the matrix are dropped by the buffer
        // range check, not branched around), and residual loads are consumed before the stores issue: the count is exact
        stores_in_flight = true;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 128x128 tiles whose LAST row tile absorbs up to 16 extra rows (bf16 output, staged epilogue).
// CLIP's token matrices have M = B*1025 = 64*128 + 8 rows at B = 8: with plain tiling every GEMM of the tower pays a
// 65th, almost empty row of tiles -- a whole extra round over the CUs (520 workgroups on 512 slots for N = 1024).  Here
// the 64th row tile carries 136 rows: one more 16-row MFMA fragment for the waves that own the bottom half, fetched by two
// extra LDS-DMA pieces per slab.  M = 8200 then tiles as 64 x N/128 workgroups = whole rounds at two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wg_gemm_tail_kernel(GemmArgs g) {
    constexpr int BM = 128, BN = 128, BK = 64, WM = 2, WN = 2, XR = 16;
    constexpr int WTM = 64, WTN = 64, FI = 4, FJ = 4;
    constexpr int ROWB = 128, STAGE = (BM + XR + BN) * ROWB;
    constexpr int ROWS_PER_ROUND = 32;
    constexpr int SROW = WTN * 2 + 16, CH = 8, RPS = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;

    const int nwg = g.tiles_m * g.tiles_n;
    int wgid;
    {
        const int orig = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tile_m = wgid / g.tiles_n, tile_n = wgid % g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bool tail = (tile_m == g.tiles_m - 1) && (m0 + BM < g.M);   // this tile also owns rows m0+128 .. M-1 (<= 16)

    const bf16* srcA[BM / ROWS_PER_ROUND];
    const bf16* srcW[BN / ROWS_PER_ROUND];
    const bf16* srcX;
#pragma unroll
    for (int i = 0; i < BM / ROWS_PER_ROUND; ++i) {
        const int r = i * ROWS_PER_ROUND + wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ wg_swz<BK>(r);
        int gr = m0 + r;
        gr = gr < g.M ? gr : g.M - 1;
        srcA[i] = g.A + (long)gr * g.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < BN / ROWS_PER_ROUND; ++i) {
        const int r = i * ROWS_PER_ROUND + wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ wg_swz<BK>(r);
        int gr = n0 + r;
        gr = gr < g.N ? gr : g.N - 1;
        srcW[i] = g.W + (long)gr * g.ldw + c * 8;
    }
    {   // the 16 extra rows: waves 0 and 1 fetch 8 rows each
        const int r = BM + (wave & 1) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ wg_swz<BK>(r);
        int gr = m0 + r;
        gr = gr < g.M ? gr : g.M - 1;
        srcX = g.A + (long)gr * g.lda + c * 8;
    }
    auto stage = [&](int kt, int buf) {
        char* ldsA = smem + buf * STAGE;
        char* ldsW = ldsA + (BM + XR) * ROWB;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < BM / ROWS_PER_ROUND; ++i)
            __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(srcA[i] + k0), WG_LDS_PTR(ldsA + (i * ROWS_PER_ROUND + wave * 8) * ROWB), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BN / ROWS_PER_ROUND; ++i)
            __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(srcW[i] + k0), WG_LDS_PTR(ldsW + (i * ROWS_PER_ROUND + wave * 8) * ROWB), 16, 0, 0);
        if (tail && wave < 2)
            __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(srcX + k0), WG_LDS_PTR(ldsA + (BM + wave * 8) * ROWB), 16, 0, 0);
    };

    f32x4 acc[FI][FJ], accx[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
        accx[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < FI; ++i) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool own_tail = tail && wm == WM - 1;   // the bottom-half waves run the extra fragment row (wave-uniform)

    const int nk = g.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        wg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* ldsA = smem + (kt & 1) * STAGE;
        const char* ldsW = ldsA + (BM + XR) * ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[FI], wf[FJ], xf;
            const int c = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int r = wm * WTM + i * 16 + fr;
                af[i] = *(const bf16x8*)(ldsA + r * ROWB + ((c ^ wg_swz<BK>(r)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const int r = wn * WTN + j * 16 + fr;
                wf[j] = *(const bf16x8*)(ldsW + r * ROWB + ((c ^ wg_swz<BK>(r)) << 4));
            }
            if (own_tail) {
                const int r = BM + fr;
                xf = *(const bf16x8*)(ldsA + r * ROWB + ((c ^ wg_swz<BK>(r)) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
            if (own_tail) {
#pragma unroll
                for (int j = 0; j < FJ; ++j) accx[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf, accx[j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    __syncthreads();

    // ---- staged epilogue: 64 rows of the wave tile, then (bottom waves of the last row tile) the 16 extra rows ------
    const int nbase = n0 + wn * WTN;
    char* stg = smem + wave * (64 * SROW);
    float bv[FJ][4];
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
        const int n = nbase + j * 16 + fq * 4;
        if (g.bias && n < g.N) {
            const bf16x4 b = *(const bf16x4*)(g.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[j][e] = (float)b[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[j][e] = 0.f;
        }
    }
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, g.c_bytes, WG_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)g.R, 0, g.r_bytes, WG_RSRC_FLAGS);
    auto finish = [&](auto has_r) __attribute__((always_inline)) {   // see wg_gemm_kernel's staged epilogue
        constexpr bool HAS_R = decltype(has_r)::value;
        u32x4 rres[8], rrx[2];
        if (HAS_R) {
            wg_load_residual<CH, 8>(rres, rrs, (int)g.ldr, g.res_mod, m0 + wm * WTM, nbase, lane);
            if (own_tail) wg_load_residual<CH, 2>(rrx, rrs, (int)g.ldr, g.res_mod, m0 + BM, nbase, lane);
        }
        WG_ACT_SWITCH(g.act,
            _Pragma("unroll") for (int i = 0; i < FI; ++i) {
                _Pragma("unroll") for (int j = 0; j < FJ; ++j)
                    *(bf16x4*)(stg + (i * 16 + fr) * SROW + (j * 16 + fq * 4) * 2) = wg_epi_pack<ACT>(acc[i][j], bv[j]);
            })
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        wg_flush_slab<SROW, CH, 8, HAS_R>(stg, rres, crs, (int)g.ldc, g.N, m0 + wm * WTM, nbase, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (own_tail) {
            WG_ACT_SWITCH(g.act,
                _Pragma("unroll") for (int j = 0; j < FJ; ++j)
                    *(bf16x4*)(stg + fr * SROW + (j * 16 + fq * 4) * 2) = wg_epi_pack<ACT>(accx[j], bv[j]);)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            wg_flush_slab<SROW, CH, 2, HAS_R>(stg, rrx, crs, (int)g.ldc, g.N, m0 + BM, nbase, lane);
        }
    };
    if (g.R) finish(std::true_type{}); else finish(std::false_type{});
}

static int launch_tail(GemmArgs& g, hipStream_t st) {
    g.tiles_m = g.M / 128;             // the last row tile takes the M % 128 (<= 16) leftover rows
    g.tiles_n = (g.N + 127) / 128;
    constexpr int lds = 2 * (128 + 16 + 128) * 128;
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) (void)hipFuncSetAttribute((const void*)wg_gemm_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(wg_gemm_tail_kernel, dim3(g.tiles_m * g.tiles_n), dim3(256), lds, st, g);
    return wg_check_launch("wg_gemm_bias_act_bf16(tail)");
}

// Small / ragged shapes (N of 1, 4, 32 ..., K not a multiple of 64): one wave per output row, lanes split K.
// Used by the gate's 128->1 linear, the IoU head, the hyper-network output layers; never on the FLOP-heavy path.
__global__ __launch_bounds__(256) void wg_gemm_rowwave_kernel(GemmArgs g) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= g.M) return;
    const bf16* a = g.A + (long)m * g.lda;
    const long rrow = g.R ? (long)(g.res_mod > 0 ? m % g.res_mod : m) * g.ldr : 0;
    for (int n = 0; n < g.N; ++n) {
        const bf16* w = g.W + (long)n * g.ldw;
        float s = 0.f;
        for (int k = lane; k < g.K; k += 64) s += (float)a[k] * (float)w[k];
        s = wg_wave_sum(s);
        if (lane == 0) {
            if (g.bias) s += (float)g.bias[n];
            s = wg_act(s, g.act);
            if (g.R) s += (float)g.R[rrow + n];
            if (g.out_f32) ((float*)g.C)[(long)m * g.ldc + n] = s;
            else ((bf16*)g.C)[(long)m * g.ldc + n] = (bf16)s;
        }
    }
}

// Skinny activations (M <= 16: the [SEG] hidden states through text_hidden_fcs, the MSQP / CTP token rows).  Such a GEMM only streams its
// weight matrix, and a compute unit ingests ~30 GB/s with plain loads, so a 128-wide tile (32 workgroups for N = 4096) leaves the matrix
// behind 32 straws: 42 us for the 32 MB of text_hidden_fcs[0].  Here one workgroup owns 16 output columns, its four waves split K,
// and the rows ride as a 16-row MFMA A operand (rows >= M zero); the partial sums meet in LDS.
// NW waves split K: 4 for short rows, 16 at K >= 2048 -- what a wave can keep in flight (8 fragments of 1 KB) over the ~2 us a miss takes
// is ~4 GB/s, so the 4 MB of text_hidden_fcs[0] (32 workgroups at N = 512) need every wave the compute units can hold.
// LN: the rows go through LayerNorm (two exact passes over the row for mean and variance, fp32) on their way into the MFMA -- every
// workgroup repeats the statistics of the <= 16 rows (they are L2-resident and tiny next to its weight columns), which saves the separate
// LayerNorm launch in front of text_hidden_fcs[0] (utils_walkgpt.py:321-323).
template <int NW, bool LN, bool TILED>
__global__ __launch_bounds__(64 * NW) void wg_gemm_skinny_kernel(GemmArgs g) {
    __shared__ f32x4 red[NW - 1][64];
    __shared__ float stat[NW][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const int n = blockIdx.x * 16 + l16;                 // N % 16 == 0
    const int m0 = blockIdx.y * 16;                      // rows m0 .. m0 + 15 (M <= 128: up to eight row blocks, each re-reading its weight
                                                         // columns from L2 -- still N / 16 x M / 16 workgroups spread over the chip)
    const int kw = g.K / NW;                             // K % (32 NW) == 0: every wave takes whole 32-deep MFMA steps
    const bool live = m0 + l16 < g.M;
    // TILED: fragment (column block, k step) is one contiguous KiB (3x the per-CU streaming rate of 16 rows x 64 B, tools/micro/cu_ingest.hip)
    const bf16* wp = TILED ? g.W + (((long)blockIdx.x * (g.K / 32) + wave * (kw / 32)) * 64 + lane) * 8 : g.W + (long)n * g.ldw + wave * kw + 8 * kg;
    constexpr int WSTEP = TILED ? 16 : 1;                // elements of W per element of k
    const bf16* ap = g.A + (long)(live ? m0 + l16 : 0) * g.lda + wave * kw + 8 * kg;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    float mean = 0.f, rstd = 1.f;
    if constexpr (LN) {
        // sum over the wave's K slice of row l16 (lanes kg = 0..3 hold interleaved 8-element pieces), then over the waves through LDS
        auto row_total = [&](float v) {
            float x, y;
            wg_permlane_swap<0>(v, x, y); v = x + y;
            wg_permlane_swap<1>(v, x, y); v = x + y;
            __syncthreads();                             // (stat is reused by the second statistic)
            if (kg == 0) stat[wave][l16] = v;
            __syncthreads();
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += stat[w][l16];
            return t;
        };
        float s = 0.f;
        for (int k = 0; k < kw; k += 32) {
            const bf16x8 a = *(const bf16x8*)(ap + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)a[j];
        }
        mean = row_total(s) / (float)g.K;
        float sq = 0.f;
        for (int k = 0; k < kw; k += 32) {
            const bf16x8 a = *(const bf16x8*)(ap + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = (float)a[j] - mean;
                sq += d * d;
            }
        }
        rstd = 1.0f / sqrtf(row_total(sq) / (float)g.K + g.sk_eps);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < kw; k += 32) {
        const bf16x8 b = *(const bf16x8*)(wp + k * WSTEP);
        bf16x8 a = *(const bf16x8*)(ap + k);
        if constexpr (LN) {
            const bf16x8 gm = *(const bf16x8*)(g.sk_gamma + wave * kw + 8 * kg + k), bt = *(const bf16x8*)(g.sk_beta + wave * kw + 8 * kg + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (bf16)(((float)a[j] - mean) * rstd * (float)gm[j] + (float)bt[j]);
        }
        a = live ? a : zero;                             // select, not branch: a join inside the loop would serialise the loads
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    if (wave) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) acc += red[w][lane];
    const float bias = g.bias ? (float)g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                        // accumulator: rows 4*kg + i of column l16
        const int m = m0 + 4 * kg + i;
        if (m >= g.M) continue;
        float v = wg_act(acc[i] + bias, g.act);
        if (g.R) v += (float)g.R[(long)(g.res_mod > 0 ? m % g.res_mod : m) * g.ldr + n];
        if (g.out_f32) ((float*)g.C)[(long)m * g.ldc + n] = v;
        else ((bf16*)g.C)[(long)m * g.ldc + n] = (bf16)v;
    }
}

template <int BM, int BN, int BK, int STAGES, int WM, int WN, bool STAGED, int PIPE, bool FP8 = false>
static int launch_tile_impl(GemmArgs& g, hipStream_t st) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    {   // tile order (wg_tile_of): column blocks when the weight matrix cannot stay in an XCD's 4 MiB L2 beside the streaming A
        // panels (> 3 MiB) and at least two of its column panels (BN x K bf16) fit in ~1.5 MiB; otherwise plain row-major,
        // which reads A once.  Measured (rocprofv3 FETCH_SIZE, tools/pmc_gemm.py): CLIP qkv 125 -> 92 MB, CLIP fc1 170 -> 114 MB,
        // SAM lin1 292 -> 227 MB per launch; SAM qkv (3.5 MB of weights) unchanged; no effect on run time either way.
        const long panel = (long)BN * g.K * 2, wbytes = (long)g.N * g.K * 2;
        const int cb = (int)((3L << 19) / (panel > 0 ? panel : 1));
        g.col_block = (wbytes > (3L << 20) && cb >= 2 && cb < g.tiles_n) ? cb : 0;
    }
    constexpr int lds_main = STAGES * (BM + BN) * BK * 2;
    constexpr int lds_stg = WM * WN * 64 * ((BN / WN) * 2 + 16);
#ifdef WG_GEMM_STAMP
    constexpr int lds = (lds_main > lds_stg ? lds_main : lds_stg) + 4096;
#else
    constexpr int lds = lds_main > lds_stg ? lds_main : lds_stg;
#endif
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) (void)hipFuncSetAttribute((const void*)wg_gemm_kernel<BM, BN, BK, STAGES, WM, WN, STAGED, PIPE, FP8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((wg_gemm_kernel<BM, BN, BK, STAGES, WM, WN, STAGED, PIPE, FP8>), dim3(g.tiles_m * g.tiles_n), dim3(WM * WN * 64), lds, st, g);
    return wg_check_launch(FP8 ? "wg_gemm_fp8_bias_act" : "wg_gemm_bias_act_bf16");
}

template <int BM, int BN, int WM, int WN>
static int launch_persist(GemmArgs& g, hipStream_t st) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    constexpr int stage = (BM + BN) * 128;
    constexpr int slab = 64 * ((BN / WN) * 2 + 16);
    constexpr int fit = stage / slab;
    constexpr int extra = (WM * WN > fit) ? (WM * WN - fit) * slab : 0;
    constexpr int lds = 2 * stage + extra;
    constexpr int per_cu = lds <= 80 * 1024 ? 2 : 1;
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) (void)hipFuncSetAttribute((const void*)wg_gemm_persist_kernel<BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int nwg = g.tiles_m * g.tiles_n;
    int grid = wg_cu_count(dev) * per_cu;   // one (or two) resident workgroups per CU; a multiple of 8 keeps a workgroup on one XCD
    if (grid > nwg) grid = nwg;
    hipLaunchKernelGGL((wg_gemm_persist_kernel<BM, BN, WM, WN>), dim3(grid), dim3(WM * WN * 64), lds, st, g);
    return wg_check_launch("wg_gemm_bias_act_bf16(persistent)");
}

// STATS instances: spart[256 rows][4 column waves][2] -> {sum, sum of squares} of row `row_off + r` over the tile's 256 columns.
// Every wave issues exactly one store instruction (those of waves 4-7 fall outside the descriptor and are dropped): the tile loop's
// counted waits rely on a fixed number of memory operations per wave.
// (A thread per row, no cross-lane step: the pairwise form -- two lanes per row joined by `x + dpp(x)` on both sums -- came out of this
// hipcc with ONE v_mov_b32_dpp feeding both halves of a v_pk_add_f32 (op_sel_hi:[1,0]), i.e. sum-of-squares + the partner's SUM.)
__device__ __forceinline__ void wg_stats_combine(const float* spart, int tid, float* out, long rows_total, int row_off) {
    const int r = tid & 255;
    const f32x4 a = *(const f32x4*)(spart + r * 8), b = *(const f32x4*)(spart + r * 8 + 4);    // {s0,q0,s1,q1}, {s2,q2,s3,q3} of row r
    const f32x2 o2 = {(a[0] + a[2]) + (b[0] + b[2]), (a[1] + a[3]) + (b[1] + b[3])};
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (unsigned)(rows_total * 8), WG_RSRC_FLAGS);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o2), srs, tid >= 256 ? (int)0x80000000 : (row_off + r) * 8, 0, 0);
}

// One block-scaled fp8 MFMA with both operands' E8M0 scales taken from byte J of `sw` (SrcA = weights) and byte I of `sa` (SrcB =
// activations).  Volatile asm: see wg_gemm_kernel (the builtin's cluster gets sunk out of its half-phase).
template <int I, int J>
__device__ __forceinline__ void wg_mx_mfma(f32x4& acc, const i32x8& w8, const i32x8& a8, unsigned sw, unsigned sa) {
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[%5,%6,0] op_sel_hi:[%7,%8,0]"
                 : "+v"(acc) : "v"(w8), "v"(a8), "v"(sw), "v"(sa), "n"(J & 1), "n"(I & 1), "n"(J >> 1), "n"(I >> 1));
}
__device__ __forceinline__ i32x8 wg_i32x8_of(bf16x8 lo, bf16x8 hi) {
    const u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
    return (i32x8){(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent 256x256 tiles with the two-cluster ping-pong loop of wg_gemm_kernel (bf16 output, staged epilogue).
// Tile-level stamps on the K = 768 shapes put 5 % of a tile's life into the first-slab latency and ~6 % into waiting for
// its 128 KiB of output stores before the workgroup may retire.  One workgroup per CU walks tiles v, v+grid, ...:
//   * the next tile's first slab (and its bias row, by LDS-DMA) is sent right after the last slab's barrier and lands under
//     the epilogue, which stages through buffer 1's region (+ a 9 KiB extension for wave 7);
//   * the next main loop starts behind a COUNTED wait: memory operations retire in issue order, so `vmcnt(NSTORE)` waits
//     for the slab (older) and leaves the tile's 16 output stores per wave (younger) draining under the next loop;
//   * residual rows are loaded one 64-row slab at a time, the second slab's loads issued after the first slab's sums are
//     formed but BEFORE its stores (registers reused; the later wait for them then never includes a store).
// ---------------------------------------------------------------------------------------------------------------------
//   * LNMODE 0: bias / residual epilogue; 1: LayerNorm folded in, statistics {mean, rstd} per row read from g.ln_stats; 2: the same
//     with the statistics formed here from the partial sums g.ln_part that the producing GEMM (STATS) left;
//   * STATS: this GEMM's output is the input of a later LayerNorm: next to its bf16 rows it leaves, per row and 256-column tile,
//     {sum, sum of squares} of the values it stored.  The separate row-statistics pass over the residual stream (72 launches per
//     C2 step, 3.7 % of its kernel time; 10.8 % of C3's, where each of them queues behind a persistent grid) disappears: the
//     statistics ride on data the epilogue already holds in registers.  Round 2 tried this with per-64-column centred sums written
//     straight to memory from the store loop (one buffer store per output store): 11 spilled registers, +0.47 ms per step.  Here
//     the sums are taken from the packed registers that feed a slab's stores (two v_dot2c per bf16 pair) while the residual
//     registers are free, cross the waves through 8 KiB of LDS, and leave as ONE 8-byte store per row and tile behind the next
//     tile's first barrier.
//   * FP8: e4m3 operands with OCP-MX block scales on BOTH sides (GemmArgs::mx_a, mx_w), applied by the MFMA itself -- the
//     accumulators come out dequantised and every epilogue above runs unchanged.  The scale bytes of a slab travel like its operands:
//     by LDS-DMA with the early pieces of slab kt+1 (7 operations stay in flight at the first counted wait instead of 6).
//   * SEAM (round 6): how the operand slabs are addressed and when a tile's first slab is requested -- see the comment at `piece` below.  true: every
//     K with an even number of slabs (all shapes of the workload): 1 000-1 300 cycles less at every tile seam, +1.5 ... 6 % per GEMM
//     (profiles/r06_gemm_phases.md); false: the round-1..5 form, kept for odd nk.
template <int LNMODE, bool STATS, bool FP8 = false, bool SEAM = true>
__global__ __launch_bounds__(512, 2) void wg_gemm_pp_persist_kernel(GemmArgs g) {
    constexpr bool LN = LNMODE != 0;
    constexpr int BM = 256, BN = 256, BK = 64, WN = 4;
    constexpr int WTM = 128, WTN = 64, FJ = 4;
    constexpr int ROWB = 128, STAGE = (BM + BN) * ROWB, RPI = 8, RPR = 64;   // RPR: rows per LDS-DMA round of the 8 waves
    constexpr int SROW = WTN * 2 + 16, CH = 8, RPS = 8;
    constexpr int NIT = 64 / RPS;                        // store instructions per 64-row slab
    // ... per wave per tile.  FP8: + the same rows as e4m3 (8 bytes per lane) and two scale dwords per slab, issued whether or not an MX
    // copy was asked for (null descriptors drop them): the tile loop's counted waits need a fixed number of operations per tile
    constexpr int NSTORE = (WTM / 64) * NIT + (FP8 ? (WTM / 64) * (NIT + 2) : 0);
    constexpr int SLAB = 64 * SROW;
    constexpr int FIT = STAGE / SLAB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    const int grp = wm;
    // the lane id, re-derived where the tile loop needs it outside the main loop (two VALU instructions) instead of being kept in a
    // register across it: every instance sits at the 256-register limit
    auto lane_now = [&]() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    char* stg = wave < FIT ? smem + STAGE + wave * SLAB : smem + 2 * STAGE + (wave - FIT) * SLAB;
    char* biasbuf = smem + 2 * STAGE + SLAB;   // [2][256] bf16, double-buffered by tile parity

    const int nwg = g.tiles_m * g.tiles_n;
    const int nk = g.K / BK;
    auto tile_of = [&](int v, int& m0, int& n0) {
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        int tm, tn;
        wg_tile_of(wgid, g.tiles_m, g.tiles_n, g.col_block, tm, tn);
        m0 = tm * BM;
        n0 = tn * BN;
    };
    // SEAM: operand slabs come in by LDS-DMA through BUFFER descriptors: a lane's byte offset inside a 64-row round (row wave * 8 + lane / 8, its
    // swizzled 16-byte chunk -- the swizzle depends on the row's bits 1-3 only, so it is the same in all four rounds) is a constant of the
    // kernel, and everything that changes -- tile, round, slab -- travels in the instruction's SCALAR offset.  Two VGPRs instead of eight
    // 64-bit pointers, no per-piece vector arithmetic, no per-tile source set-up; rows past M / N read as zero through the descriptor's range
    // check (the scalar offset takes part in it on gfx950: tools/micro/buffer_soffset_range.hip), so ragged last tiles need no clamp.  That is
    // what lets the NEXT tile's first slab be requested from inside this tile's last slab (below).
    // !SEAM: eight per-lane 64-bit row pointers per tile (rows clamped to the matrix), a slab's column added per piece.
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, g.a_bytes, WG_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, g.w_bytes, WG_RSRC_FLAGS);
    [[maybe_unused]] int voffA = 0, voffW = 0;
    [[maybe_unused]] const bf16* srcA[4];
    [[maybe_unused]] const bf16* srcW[4];
    if constexpr (SEAM) {
        const int ln = lane_now();
        const int r = wave * RPI + (ln >> 3);
        const int c = (ln & 7) ^ wg_swz<BK>(r);
        voffA = (r * (int)g.lda + c * 8) * 2;
        voffW = (r * (int)g.ldw + c * 8) * 2;
    }
    auto set_sources = [&](int m0, int n0) {
        const int ln = FP8 ? lane_now() : lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * RPR + wave * RPI + (ln >> 3);
            const int c = (ln & 7) ^ wg_swz<BK>(r);
            int ga = m0 + r, gw = n0 + r;
            ga = ga < g.M ? ga : g.M - 1;
            gw = gw < g.N ? gw : g.N - 1;
            srcA[i] = g.A + (long)ga * g.lda + c * 8;
            srcW[i] = g.W + (long)gw * g.ldw + c * 8;
        }
    };
    // two LDS-DMA instructions of a slab into buffer `stage`.  SEAM: offA / offW = scalar byte offsets of the slab's first A / W row (tile origin row
    // * pitch + k column), the four 64-row rounds sit rsA / rsW bytes apart; !SEAM: offA = offW = the slab's k column (elements) on top of the
    // tile's row pointers.  which: 0,1 = W round pairs; 2 = A rounds 0,2 (early); 3 = A rounds 1,3 (late)
    const unsigned rsA = (unsigned)g.lda * (RPR * 2), rsW = (unsigned)g.ldw * (RPR * 2);
    auto piece = [&](int stage, unsigned offA, unsigned offW, int which, int only = -1) {
        char* ldsA = smem + stage * STAGE;
        char* ldsW = ldsA + BM * ROWB;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (only >= 0 && u != only) continue;
            if (which < 2) {
                const int i = which * 2 + u;
                if constexpr (SEAM)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, WG_LDS_PTR(ldsW + (i * RPR + wave * RPI) * ROWB), 16, voffW, (int)(offW + i * rsW), 0, WG_GEMM_W_AUX);
                else
                    __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(srcW[i] + offW), WG_LDS_PTR(ldsW + (i * RPR + wave * RPI) * ROWB), 16, 0, WG_GEMM_W_AUX);
            } else {
                const int i = (which - 2) + 2 * u;
                if constexpr (SEAM)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, WG_LDS_PTR(ldsA + (i * RPR + wave * RPI) * ROWB), 16, voffA, (int)(offA + i * rsA), 0, WG_GEMM_A_AUX);
                else
                    __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(srcA[i] + offA), WG_LDS_PTR(ldsA + (i * RPR + wave * RPI) * ROWB), 16, 0, WG_GEMM_A_AUX);
            }
        }
    };
    // LN: per tile parity  s[256] f32 | b'[256] f32 | stats[256][2] f32  (4 KiB), all by LDS-DMA like the bias row
    constexpr int EPIB = LN ? 4096 : 512;
    // LNMODE 2: + raw[ln_np <= 5][256][2] f32 partial sums of the next tile's rows (single-buffered: turned into the parity's
    // stats[] right behind the tile's first barrier, many barriers before the next DMA into it).  STATS: + part[256][4][2] f32.
    char* const rawbuf = biasbuf + 2 * EPIB;
    float* const spart = (float*)(biasbuf + 2 * EPIB);
    // FP8: the block scales of a slab -- 4 planes x 256 rows of A, 4 planes x 256 rows of W, one byte each -- ride the operands' LDS-DMA
    // pipeline: one more 4-byte-per-lane DMA per wave and slab (waves 0-3: A plane `wave`, waves 4-7: W plane `wave - 4`) into
    // mxbuf[slab parity][2 KiB], sent with the early pieces, read back as one dword per cluster (A) / per slab (W) in the M half-phases.
    // (As raw global loads into registers under the counted waits, hipcc's own copies of the destination registers ran ahead of the
    // data; as plain loads it drained the whole queue -- vmcnt(0) -- at every loop edge.)
    char* const mxbuf = biasbuf + 2 * EPIB + (LNMODE == 2 ? 5 * 2048 : 0) + (STATS ? 256 * 4 * 8 : 0);
    // (addresses: wave-uniform base + lane * 4, the lane id re-derived where it is needed -- nothing of this may stay in a VGPR across the
    // main loop: the bf16 instances already sit at the 256-register limit, and a spilled register's reload is a VMEM operation that makes
    // hipcc drain the whole queue)
    auto scale_piece = [&](int tm0, int tn0, int kt) {
        const unsigned char* ub = wave < 4 ? g.mx_a + (long)(4 * kt + wave) * g.mx_a_pitch + tm0 : g.mx_w + (long)(4 * kt + wave - 4) * g.mx_w_pitch + tn0;
        __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(ub + (unsigned)(lane_now() * 4)), WG_LDS_PTR(mxbuf + (kt & 1) * 2048 + wave * 256), 4, 0, 0);
    };
    // the epilogue operands of a tile (bias row / LayerNorm fold vectors and row statistics), by LDS-DMA into parity `par`
    auto first_ops = [&](int m0, int n0, int par) {
        const int lane = FP8 ? lane_now() : (tid & 63);   // (shadows the kernel's: see lane_now)
        if (LN) {
            if (wave == 0) {
                int n = n0 + lane * 4;
                n = n + 4 <= g.N ? n : 0;         // columns past N are never stored
                __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(g.ln_s + n), WG_LDS_PTR(biasbuf + par * EPIB), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(g.ln_b + n), WG_LDS_PTR(biasbuf + par * EPIB + 1024), 16, 0, 0);
            } else if (wave == 1) {
                if constexpr (LNMODE == 2) {
                    // rows m0 .. m0+255 of every partial plane (the planes are padded to whole row tiles: no clamp needed)
                    int ol = lane;                               // opaque: keeps the address out of the registers that live across the main loop
                    asm volatile("" : "+v"(ol));
                    const float* src = g.ln_part + 2 * ((long)m0 + ol * 2);
#pragma unroll 1
                    for (int p = 0; p < g.ln_np; ++p) {
                        __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(src), WG_LDS_PTR(rawbuf + p * 2048), 16, 0, 0);
                        __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(src + 256), WG_LDS_PTR(rawbuf + p * 2048 + 1024), 16, 0, 0);
                        src += 2 * g.ln_mpad;
                    }
                } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    int m = m0 + u * 128 + lane * 2;
                    m = m < g.M ? m : 0;          // rows past M are never stored; the buffer holds an even number of rows
                    __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(g.ln_stats + 2 * (long)m), WG_LDS_PTR(biasbuf + par * EPIB + 2048 + u * 1024), 16, 0, 0);
                }
                }
            }
        } else if (g.bias && wave == 0 && lane < 32) {   // the tile's 256 bias values: one 512-byte LDS-DMA
            int n = n0 + lane * 8;
            n = n + 8 <= g.N ? n : 0;             // columns past N are never stored
            __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(g.bias + n), WG_LDS_PTR(biasbuf + par * 512), 16, 0, 0);
        }
    };
    // a tile's first slab (into buffer 0)
    auto first_pieces = [&](int m0, int n0) {
        if constexpr (FP8) scale_piece(m0, n0, 0);
        if constexpr (!SEAM) set_sources(m0, n0);
        const unsigned oA = SEAM ? (unsigned)m0 * (unsigned)g.lda * 2u : 0u, oW = SEAM ? (unsigned)n0 * (unsigned)g.ldw * 2u : 0u;
        piece(0, oA, oW, 0); piece(0, oA, oW, 1); piece(0, oA, oW, 2); piece(0, oA, oW, 3);
    };
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, g.c_bytes, WG_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)g.R, 0, g.r_bytes, WG_RSRC_FLAGS);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(g.Cq, 0, g.cq_bytes, WG_RSRC_FLAGS);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)g.mx_c, 0, g.mx_c_bytes, WG_RSRC_FLAGS);
    int sp_off = 0;    // STATS: row index (tile column * mpad + first row) of the tile whose row sums sit in spart

    int v = blockIdx.x;
    int m0, n0;
    tile_of(v, m0, n0);
    first_ops(m0, n0, 0);
    first_pieces(m0, n0);
    bool stores_in_flight = false;
    int par = 0;
#ifdef WG_GEMM_STAMP
    unsigned long long pst_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pst_prev = 0, pst_e = 0;      // 0 first-slab wait, 1 main loop, 2 epilogue | 3.. its parts
    unsigned pst_tiles = 0;
    const unsigned long long pst_c0 = __builtin_amdgcn_s_memtime(), pst_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    while (true) {
        WG_PSTAMP(0);
        const int nbase = n0 + wn * WTN;
        // The next tile of this workgroup, known before the main loop starts (scalar arithmetic, off the seam between two tiles).  With an even
        // number of slabs its first slab is requested INSIDE this tile's last slab, exactly where a slab kt + 1 would be (buffer 0 is free from
        // slab nk - 2 on, and the epilogue stages through buffer 1): the operand pipeline never drains at a tile seam, and the epilogue starts
        // with nothing to set up.  (Odd nk: buffer 0 holds the last slab; the first slab is then sent behind the loop, as before round 6.)
        const int vn = v + gridDim.x;
        const bool has_next = vn < nwg;
        int m0n = m0, n0n = n0;
        if (SEAM && has_next) tile_of(vn, m0n, n0n);      // (!SEAM: behind the main loop, where it always was -- nothing more live across the loop)
        const bool seam = SEAM && has_next && !(nk & 1);
        f32x4 acc[8][FJ];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 af[4][2], wf2[2][2][2];
        [[maybe_unused]] unsigned sb = 0x7F7F7F7Fu, swc = 0x7F7F7F7Fu;   // FP8: the clusters' A scales (byte i = fragment i) / the slab's W scales (byte j)
        auto read_a = [&](const char* ldsA, int ci) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int r = wm * WTM + (4 * ci + i) * 16 + fr;
                    af[i][ks] = *(const bf16x8*)(ldsA + r * ROWB + (((ks * 4 + fq) ^ wg_swz<BK>(r)) << 4));
                }
        };
        auto read_w = [&](const char* ldsW, int cj) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int r = wn * WTN + (2 * cj + j) * 16 + fr;
                    wf2[cj][j][ks] = *(const bf16x8*)(ldsW + r * ROWB + (((ks * 4 + fq) ^ wg_swz<BK>(r)) << 4));
                }
        };
        // the whole first slab (older than the previous tile's stores) has landed; those stores may still be draining
        // (the slab's pieces are the oldest operations in the queue: behind them only the NSTORE output stores of the previous tile)
        if (stores_in_flight) wg_wait_vmcnt<NSTORE>(); else wg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if constexpr (LNMODE == 2) {
            // this tile's row statistics from the partial sums that arrived with its first slab:  mean = S / K,
            // var = Q / K - mean^2 (fp32 sums of bf16 values; |mean| stays within a few sigma on a residual stream), rstd = (var + eps)^-1/2
            // (by the waves of group 1, which wait at their extra barrier anyway while group 0 is in its first load half-phase)
            int ct = (FP8 ? wave * 64 + lane_now() : tid) - 256;
            asm volatile("" : "+v"(ct));
            if (ct >= 0) {
                float S = 0.f, Q = 0.f;
#pragma unroll 1
                for (int p = 0; p < g.ln_np; ++p) {
                    const f32x2 v = *(const f32x2*)(rawbuf + p * 2048 + ct * 8);
                    S += v.x; Q += v.y;
                }
                const float invk = 1.0f / (float)(FP8 ? 2 * g.K : g.K);     // (fp8: g.K counts byte pairs)
                const float mean = S * invk;
                const float var = fmaxf(Q * invk - mean * mean, 0.f);
                *(f32x2*)(biasbuf + par * EPIB + 2048 + ct * 8) = (f32x2){mean, __builtin_amdgcn_rsqf(var + g.ln_eps)};
            }
        }
        if constexpr (STATS) {
            // the previous tile's row sums: the four column waves' shares meet here (their LDS writes are behind the barrier above)
            // (group 1's threads carry the store -- see above; every wave issues it: the counted waits need the same number of operations per wave)
            if (stores_in_flight) wg_stats_combine(spart, (FP8 ? wave * 64 + lane_now() : tid) ^ 256, g.stats_part, g.tiles_n * g.stats_mpad, sp_off);
        }
        WG_PSTAMP(1);
        if (grp == 1) __builtin_amdgcn_s_barrier();
        // One slab = two clusters; each cluster an M half-phase (fragment reads + requests for later slabs, then the counted wait and the hand-over
        // barrier) and its 64 MFMAs.  Requests of the seam flow (even nk; E(s) = the six "early" pieces of slab s: all of W and the A rows of the
        // waves' first clusters, L(s) = the two "late" pieces: the A rows of their second clusters):
        //     M(kt, first cluster):  L(kt + 1)            -- those rows of buffer (kt + 1) & 1 were last read in slab kt - 1's second cluster
        //     M(kt, second cluster): E(kt + 2)            -- into the buffer slab kt itself sits in: W and the first clusters' A rows were read in
        //                                                    M(kt, first cluster) by both groups, one barrier ago
        // so every piece has a whole slab (~1.2 us) between request and first use.  Rounds 1-6 sent E(kt + 1) in M(kt, first) -- three quarters of the
        // operand bytes with HALF a slab of lead, which is what the K >= 3072 shapes (A from beyond L2) paid 2800 instead of 2350 cycles per slab for.
        // Slab 0 also sends E(1) (nothing could send it earlier: the previous tile's epilogue stages through buffer 1); past the tile's end the
        // same slots carry the NEXT tile's slab 0 (E'(0) in slab nk - 2, L'(0) in slab nk - 1, both into buffer 0).  fp8: the slab's scale bytes
        // S(kt + 1) go first in M(kt, first cluster) (their buffer is read in both clusters, so they keep the old timing).
        // Queue, oldest first, at the wait that ends M(kt, first): .. L(kt) | E(kt+1) | [S] L(kt+1)  -> L(kt) is in at vmcnt(8) (fp8: 9);
        // at the wait that ends M(kt, second): E(kt+1) | [S] L(kt+1) | E(kt+2)  -> E(kt+1) and S(kt+1) are in at vmcnt(8), or vmcnt(2) when no
        // E(kt+2) was sent.  Slab 0's first wait needs nothing (the tile-start wait covered slab 0; the queue may still hold the previous tile's stores).
        // Odd nk (tests only) keeps the flow of rounds 1-5: E(kt + 1) and L(kt + 1) both sent in slab kt, first slab sent behind the previous tile's loop.
        // Straight instances in sequence (first slab, loop body, last slab): an if / else between two instances is a control-flow merge of 128
        // accumulators, which this hipcc resolves with spills.
        auto slab = [&](int kt, auto firstc, auto lastc, const bool more, unsigned lA, unsigned lW, unsigned eA, unsigned eW, int sm0, int sn0, int skt) __attribute__((always_inline)) {
            constexpr bool last = decltype(lastc)::value;
            const bool first = kt == 0;      // (last: nk >= 2)
            // SEAM: `more` = E is sent in the second cluster (first / loop instances; L always is) | the next tile's L'(0) is sent (last instance)
            // !SEAM: `more` = slab kt + 1 exists (eA / eW its k offset)
            const char* ldsA = smem + (kt & 1) * STAGE;
            const char* ldsW = ldsA + BM * ROWB;
            const int pst = (kt + 1) & 1;
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {
                // Requests sit BETWEEN the fragment reads, at most one per read: issued back to back (round 6 first had them behind the reads with only
                // scalar instructions in between, where rounds 1-5 had happened to keep a 64-bit vector add between any two) they cost the
                // K >= 3072 shapes 2-6 %: profiles/r06_gemm_phases.md.  (An LDS-DMA write and an LDS read may alias as far as hipcc knows, so it
                // keeps this order.)
                auto ra = [&](int i, int ks) __attribute__((always_inline)) {
                    const int r = wm * WTM + (4 * sc + i) * 16 + fr;
                    af[i][ks] = *(const bf16x8*)(ldsA + r * ROWB + (((ks * 4 + fq) ^ wg_swz<BK>(r)) << 4));
                };
                auto rw = [&](int cj, int j, int ks) __attribute__((always_inline)) {
                    const int r = wn * WTN + (2 * cj + j) * 16 + fr;
                    wf2[cj][j][ks] = *(const bf16x8*)(ldsW + r * ROWB + (((ks * 4 + fq) ^ wg_swz<BK>(r)) << 4));
                };
                if constexpr (!SEAM) {      // (odd nk only: the order of rounds 1-5; interleaved, its eight row pointers spill)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { ra(i, 0); ra(i, 1); }
                    if (sc == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { rw(q >> 1, q & 1, 0); rw(q >> 1, q & 1, 1); }
                    }
                    if (more) {
                        if (sc == 0) { piece(pst, eA, eW, 0); piece(pst, eA, eW, 1); piece(pst, eA, eW, 2); }
                        else piece(pst, eA, eW, 3);
                    }
                } else if (sc == 0) {
                    const bool sendL = !last || more;
                    if constexpr (FP8) {
                        if (sendL) scale_piece(sm0, sn0, skt);
                    }
                    ra(0, 0); ra(0, 1); ra(1, 0);
                    if (first) piece(pst, lA, lW, 0, 0);         // (slab 0: E(1) rides here, in front of L(1); its k offset is L's)
                    ra(1, 1); ra(2, 0); ra(2, 1);
                    if (first) piece(pst, lA, lW, 0, 1);
                    ra(3, 0); ra(3, 1);
                    if (first) piece(pst, lA, lW, 1, 0);
                    rw(0, 0, 0); rw(0, 0, 1);
                    if (first) piece(pst, lA, lW, 1, 1);
                    rw(0, 1, 0); rw(0, 1, 1);
                    if (first) piece(pst, lA, lW, 2, 0);
                    rw(1, 0, 0);
                    if (first) piece(pst, lA, lW, 2, 1);
                    rw(1, 0, 1);
                    if (sendL) piece(pst, lA, 0, 3, 0);
                    rw(1, 1, 0); rw(1, 1, 1);
                    if (sendL) piece(pst, lA, 0, 3, 1);
                } else {
                    const bool sendE = !last && more;
                    const int est = kt & 1;      // E(kt + 2) goes into this slab's own buffer
                    ra(0, 0);
                    if (sendE) piece(est, eA, eW, 0, 0);
                    ra(0, 1);
                    if (sendE) piece(est, eA, eW, 0, 1);
                    ra(1, 0);
                    if (sendE) piece(est, eA, eW, 1, 0);
                    ra(1, 1);
                    if (sendE) piece(est, eA, eW, 1, 1);
                    ra(2, 0); ra(2, 1);
                    if (sendE) piece(est, eA, eW, 2, 0);
                    ra(3, 0); ra(3, 1);
                    if (sendE) piece(est, eA, eW, 2, 1);
                }
                if constexpr (FP8) {
                    const char* mxs = mxbuf + (kt & 1) * 2048;
                    const int ol = lane_now();
                    sb = *(const unsigned*)(mxs + (ol >> 4) * 256 + wm * WTM + (ol & 15) * 8 + 4 * sc);
                    if (sc == 0) {
                        swc = *(const unsigned*)(mxs + 1024 + (ol >> 4) * 256 + wn * WTN + (ol & 15) * 4);
                        if constexpr (!SEAM) {
                            if (more) scale_piece(sm0, sn0, skt);
                        }
                    }
                }
                if constexpr (SEAM) {
                    if (sc == 0) {
                        if (first) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        else if (last && !more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        else if (FP8) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                    } else {
                        if (last) {
                            if (more) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (seam: nothing of THIS tile is outstanding any more)
                            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        } else if (more) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                    }
                } else if (sc == 0) {
                    // late A rows of this slab must be in.  In slab 0 they are (first-slab wait above) and the queue may still hold
                    // the previous tile's stores in front of the six pieces just sent: do not wait for those here.
                    if (!more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else if (kt == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    else if (FP8) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                } else {
                    if (!more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
                if constexpr (FP8) {
                    const i32x8 w0 = wg_i32x8_of(wf2[0][0][0], wf2[0][0][1]), w1 = wg_i32x8_of(wf2[0][1][0], wf2[0][1][1]);
                    const i32x8 w2 = wg_i32x8_of(wf2[1][0][0], wf2[1][0][1]), w3 = wg_i32x8_of(wf2[1][1][0], wf2[1][1][1]);
#define WG_MX_ROW(I)                                                                                                          \
    {                                                                                                                         \
        const i32x8 a8 = wg_i32x8_of(af[I][0], af[I][1]);                                                                     \
        wg_mx_mfma<I, 0>(acc[4 * sc + I][0], w0, a8, swc, sb); wg_mx_mfma<I, 1>(acc[4 * sc + I][1], w1, a8, swc, sb);         \
        wg_mx_mfma<I, 2>(acc[4 * sc + I][2], w2, a8, swc, sb); wg_mx_mfma<I, 3>(acc[4 * sc + I][3], w3, a8, swc, sb);         \
    }
                    WG_MX_ROW(0) WG_MX_ROW(1) WG_MX_ROW(2) WG_MX_ROW(3)
#undef WG_MX_ROW
                } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (WG_GEMM_TAIL > 0 && ks * 16 + i * 4 + j == 32 - WG_GEMM_TAIL) {
                                // The hand-over barrier WG_GEMM_TAIL MFMAs before the end of the cluster: the partner group (parked at this barrier
                                // since its load half-phase ended) is released while this wave still has matrix work to issue, so the barrier's
                                // release latency passes under the tail instead of under an idle matrix pipe.  The tail outranks the partner's
                                // fresh cluster (priority 2 against 1: ties go to the older wave, which would starve a younger wave's tail).
                                __builtin_amdgcn_sched_barrier(0);
                                __builtin_amdgcn_s_setprio(2);
                                __builtin_amdgcn_s_barrier();
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            wg_mfma16_acc(acc[4 * sc + i][j], wf2[j >> 1][j & 1][ks], af[i][ks]);
                        }
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                // (the tile's last barrier -- group 1 behind its last cluster, group 0 one cluster earlier -- is left out in the seam flow: every LDS
                // read of the tile was over at the barrier in front of that cluster, so group 0 starts its epilogue beside group 1's last
                // MFMAs instead of idling through them)
                if (FP8 || WG_GEMM_TAIL == 0) {
                    if (!(SEAM && last && sc == 1 && grp == 1)) __builtin_amdgcn_s_barrier();
                }
            }
        };
        {
            using T_ = std::true_type;
            using F_ = std::false_type;
            if constexpr (SEAM) {
                // (one call per instance, their arguments chosen by selects: a branch around an instance is the merge described above)
                const unsigned nA = (unsigned)m0n * (unsigned)g.lda * 2u, nW = (unsigned)n0n * (unsigned)g.ldw * 2u;      // the next tile's slab 0
                unsigned pA = (unsigned)m0 * (unsigned)g.lda * 2u + BK * 2, pW = (unsigned)n0 * (unsigned)g.ldw * 2u + BK * 2;      // slab kt + 1
                int kt = 0;
                for (; kt + 1 < nk; ++kt) {
                    const bool inner = kt + 2 < nk;
                    slab(kt, F_{}, F_{}, inner || seam, pA, pW, inner ? pA + BK * 2 : nA, inner ? pW + BK * 2 : nW, m0, n0, kt + 1);
                    pA += BK * 2;
                    pW += BK * 2;
                }
                slab(kt, F_{}, T_{}, seam, nA, nW, 0, 0, m0n, n0n, 0);
            } else {
                for (int kt = 0; kt < nk; ++kt) slab(kt, F_{}, F_{}, kt + 1 < nk, 0, 0, (unsigned)(kt + 1) * BK, (unsigned)(kt + 1) * BK, m0, n0, kt + 1);
            }
        }
        if (!SEAM && grp == 0) __builtin_amdgcn_s_barrier();
        // every wave is past its last LDS read of this tile (each M half-phase retired its reads before its barrier)
        WG_PSTAMP(2);

        const int cm0 = m0;
        // opaque copies of the lane coordinates: keeps hipcc from hoisting the epilogue's tile-invariant address arithmetic
        // out of the tile loop, where it would stay live across the main loop
        int el = lane, efr = fr, efq = fq;
        if constexpr (FP8) {
            el = lane_now();
            efr = el & 15;
            efq = el >> 4;
        }
        asm volatile("" : "+v"(el), "+v"(efr), "+v"(efq));
        float bv[FJ][4];
        if (!LN) {
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                bf16x4 b = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
                if (g.bias) b = *(const bf16x4*)(biasbuf + par * 512 + (wn * WTN + j * 16 + efq * 4) * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[j][e] = (float)b[e];
            }
        }
        const char* lnbuf = biasbuf + par * EPIB;
        if (has_next) {
            if constexpr (!SEAM) tile_of(vn, m0n, n0n);
            m0 = m0n;
            n0 = n0n;
            first_ops(m0, n0, par ^ 1);
            if (!seam) first_pieces(m0, n0);
        }
#ifdef WG_GEMM_STAMP
        pst_e = pst_prev;
        WG_ESTAMP(3);
#endif
        auto finish = [&](auto has_r) __attribute__((always_inline)) {
            constexpr bool HAS_R = decltype(has_r)::value;
            u32x4 rres[NIT];
            if (HAS_R) wg_load_residual<CH, NIT>(rres, rrs, (int)g.ldr, g.res_mod, cm0 + wm * WTM, nbase, el);
#pragma unroll
            for (int half = 0; half < WTM / 64; ++half) {
                if (LN) {
                    // LayerNorm folded into the GEMM: y = LN(x) W^T + b = rstd * (x W'^T - mean * s) + b' with W' = W * gamma (per
                    // input column), s = row sums of W', b' = b + W beta.  A holds the RAW rows x (statistics: wg_row_stats_bf16).
                    float lnr[4], lnm[4];   // rstd and mean * rstd of this lane's 4 accumulator rows of this half
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x2 st2 = *(const f32x2*)(lnbuf + 2048 + (wm * WTM + (half * 4 + i) * 16 + efr) * 8);
                        lnr[i] = st2.y;
                        lnm[i] = st2.x * st2.y;
                    }
                    WG_ACT_SWITCH(g.act,
                        _Pragma("unroll") for (int j = 0; j < FJ; ++j) {
                            const f32x4 s4 = *(const f32x4*)(lnbuf + (wn * WTN + j * 16 + efq * 4) * 4);
                            const f32x4 b4 = *(const f32x4*)(lnbuf + 1024 + (wn * WTN + j * 16 + efq * 4) * 4);
                            _Pragma("unroll") for (int i = 0; i < 4; ++i) {
                                const float r = lnr[i]; const float mr = lnm[i];
                                const f32x4 a4 = acc[half * 4 + i][j];
                                f32x2 lo = {a4[0] * r + (b4[0] - mr * s4[0]), a4[1] * r + (b4[1] - mr * s4[1])};
                                f32x2 hi = {a4[2] * r + (b4[2] - mr * s4[2]), a4[3] * r + (b4[3] - mr * s4[3])};
                                lo = FP8 ? wg_act2<ACT>(lo) : wg_act2e<ACT>(lo); hi = FP8 ? wg_act2<ACT>(hi) : wg_act2e<ACT>(hi);
                                *(bf16x4*)(stg + (i * 16 + efr) * SROW + (j * 16 + efq * 4) * 2) = (bf16x4){(bf16)lo.x, (bf16)lo.y, (bf16)hi.x, (bf16)hi.y};
                            }
                        })
                } else
                WG_ACT_SWITCH(g.act,
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) {
                        _Pragma("unroll") for (int j = 0; j < FJ; ++j)
                            *(bf16x4*)(stg + (i * 16 + efr) * SROW + (j * 16 + efq * 4) * 2) = wg_epi_pack<ACT, FP8>(acc[half * 4 + i][j], bv[j]);
                    })
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                WG_ESTAMP(4 + 2 * half);
                // slab reads + residual sums, then (first slab) the second slab's residual loads into the same registers, then
                // the stores: no load is ever issued behind a store that a later wait would have to cover
                constexpr int RPSl = 64 / CH;
                bf16x8 o[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) o[it] = *(const bf16x8*)(stg + (it * RPSl + el / CH) * SROW + (el % CH) * 16);
                if (HAS_R) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const bf16x8 r = __builtin_bit_cast(bf16x8, rres[it]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[it][e] = (bf16)((float)o[it][e] + (float)r[e]);
                    }
                }
                u32x4 t[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    t[it] = __builtin_bit_cast(u32x4, o[it]);
                    asm volatile("" : "+v"(t[it]));
                }
                if constexpr (STATS) {
                    // row sums of what is about to be stored (taken here, while the residual registers are free): lane el holds 8
                    // consecutive columns of row it * 8 + el / 8 of this 64-row slab, the 8 lanes el % 8 share a row.  Partial of this
                    // wave's 64 columns -> spart[row][wn] = {sum, sum of squares}.
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        // v_dot2c_f32_bf16: one instruction per pair and sum (was: unpack, add, multiply-add on fp32 pairs -- the residual + row-sum
                        // epilogue is vector-bound).  Each dword goes through a scalar copy first: handed a vector ELEMENT, this hipcc's builtin read
                        // dword 0 for every k (and as inline asm the DPP steps below would follow it without the wait states the compiler only inserts behind
                        // instructions it can see: hence the builtin on a scalar copy).
                        float sv = 0.f, qv = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            unsigned pr = t[it][k];
                            asm volatile("" : "+v"(pr));
                            const bf16x2 p2 = __builtin_bit_cast(bf16x2, pr);
                            sv = __builtin_amdgcn_fdot2_f32_bf16(p2, (bf16x2){(bf16)1.0f, (bf16)1.0f}, sv, false);
                            qv = __builtin_amdgcn_fdot2_f32_bf16(p2, p2, qv, false);
                        }
                        sv += WG_DPP(sv, 0xB1); qv += WG_DPP(qv, 0xB1);      // quad_perm [1,0,3,2]
                        sv += WG_DPP(sv, 0x4E); qv += WG_DPP(qv, 0x4E);      // quad_perm [2,3,0,1]
                        sv += WG_DPP(sv, 0x141); qv += WG_DPP(qv, 0x141);    // row_half_mirror: the other quad of the 8 lanes
                        if ((el & 7) == 0)
                            *(f32x2*)(spart + ((wm * WTM + half * 64 + it * RPSl + (el >> 3)) * 4 + wn) * 2) = (f32x2){sv, qv};
                    }
                }
                // FP8, MX copy: the final values go back into the wave's slab (same 16 bytes this lane took them from) and are turned into
                // e4m3 + block scales one row segment at a time AFTER the bf16 stores, when their 32 registers are free -- done from the
                // registers in the store loop this cost 16-86 spilled registers, and a spill's reload is a VMEM operation the counted
                // waits do not know about.
                [[maybe_unused]] const bool mxc = FP8 && g.mx_c != nullptr;
                if constexpr (FP8) {
                    if (mxc) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) *(u32x4*)(stg + (it * RPSl + el / CH) * SROW + (el % CH) * 16) = t[it];
                    }
                }
                if (HAS_R && half + 1 < WTM / 64) wg_load_residual<CH, NIT>(rres, rrs, (int)g.ldr, g.res_mod, cm0 + wm * WTM + (half + 1) * 64, nbase, el);
                const int n = nbase + (el % CH) * 8;
                const int off0 = n < g.N ? ((cm0 + wm * WTM + half * 64 + el / CH) * (int)g.ldc + n) * 2 : (int)0x80000000;
#pragma unroll
                for (int it = 0; it < NIT; ++it) __builtin_amdgcn_raw_buffer_store_b128(t[it], crs, off0 + it * RPSl * (int)g.ldc * 2, 0, WG_GEMM_C_AUX);
                if constexpr (FP8) {
                    // one E8M0 scale per (row, 32 columns): the power of two at or above max|block| / 448 (no value saturates), byte
                    // 127 + its exponent; the four lanes of a quad hold one block of a row.  Scale bytes: rows r, r+16, .. of a 128-row
                    // group are adjacent (GemmArgs::mx_a), which makes the 4 scales a lane collects over the even (odd) iterations of a
                    // 64-row slab one aligned dword.  Without an MX copy the same stores leave with null descriptors (fixed count per tile).
                    const int qoff0 = n < g.N ? (cm0 + wm * WTM + half * 64 + el / CH) * (int)g.ldcq + n : (int)0x80000000;
                    unsigned se = 0, so = 0;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        u32x2 q = {0u, 0u};
                        if (mxc) {
                            const u32x4 tv = *(const u32x4*)(stg + (it * RPSl + el / CH) * SROW + (el % CH) * 16);
                            float am = 0x1p-100f;
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                am = fmaxf(am, fmaxf(fabsf(__builtin_bit_cast(float, tv[k] << 16)), fabsf(__builtin_bit_cast(float, tv[k] & 0xFFFF0000u))));
                            am = fmaxf(am, WG_DPP(am, 0xB1));
                            am = fmaxf(am, WG_DPP(am, 0x4E));
                            const unsigned bits = __builtin_bit_cast(unsigned, am * (1.0f / 448.0f));
                            unsigned e8 = (bits >> 23) + ((bits & 0x7FFFFFu) ? 1u : 0u);
                            e8 = e8 > 253u ? 253u : e8;
                            const float inv = __builtin_bit_cast(float, (254u - e8) << 23);
#define WG_LO(k) (__builtin_bit_cast(float, tv[k] << 16) * inv)
#define WG_HI(k) (__builtin_bit_cast(float, tv[k] & 0xFFFF0000u) * inv)
                            int lo = 0, hi = 0;
                            lo = __builtin_amdgcn_cvt_pk_fp8_f32(WG_LO(0), WG_HI(0), lo, false);
                            lo = __builtin_amdgcn_cvt_pk_fp8_f32(WG_LO(1), WG_HI(1), lo, true);
                            hi = __builtin_amdgcn_cvt_pk_fp8_f32(WG_LO(2), WG_HI(2), hi, false);
                            hi = __builtin_amdgcn_cvt_pk_fp8_f32(WG_LO(3), WG_HI(3), hi, true);
#undef WG_LO
#undef WG_HI
                            q = (u32x2){(unsigned)lo, (unsigned)hi};
                            if (it & 1) so |= e8 << (8 * (it >> 1));
                            else se |= e8 << (8 * (it >> 1));
                        }
                        __builtin_amdgcn_raw_buffer_store_b64(q, qrs, qoff0 + it * RPSl * (int)g.ldcq, 0, 0);
                    }
                    const int soff = (n < g.N && (el & 3) == 0) ? (n >> 5) * (int)g.mx_c_pitch + cm0 + wm * WTM + (el / CH) * 8 + 4 * half : (int)0x80000000;
                    __builtin_amdgcn_raw_buffer_store_b32(se, srs, soff, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(so, srs, soff + 64, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                WG_ESTAMP(5 + 2 * half);
            }
        };
        if (g.R) finish(std::true_type{}); else finish(std::false_type{});
        if constexpr (STATS) sp_off = (nbase / BN) * (int)g.stats_mpad + cm0;
#ifdef WG_GEMM_STAMP
        WG_PSTAMP(3);
        ++pst_tiles;
#endif
        if (!has_next) break;
        v = vn;
        par ^= 1;
        // every wave issues all NSTORE store instructions of a tile (rows / columns outside the matrix are dropped by the buffer
        // range check, not branched around) and every load of the epilogue is consumed before the last stores issue: exact count
        stores_in_flight = true;
    }
#ifdef WG_GEMM_STAMP
    if (wg_gemm_stamp_ptr && lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 32) {      // [workgroup][wave half]: 3 phase sums, tiles, cycles, 100 MHz ticks
        unsigned* o = wg_gemm_stamp_ptr + (blockIdx.x * 2 + (wave >> 2)) * 8;
        o[0] = (unsigned)pst_sum[0]; o[1] = (unsigned)pst_sum[1]; o[2] = (unsigned)pst_sum[2]; o[3] = pst_tiles;
        o[4] = (unsigned)(__builtin_amdgcn_s_memtime() - pst_c0); o[5] = (unsigned)(__builtin_amdgcn_s_memrealtime() - pst_r0);
        unsigned* e = wg_gemm_stamp_ptr + 512 + (blockIdx.x * 2 + (wave >> 2)) * 8;      // epilogue parts: next-tile setup | rows 0-63: compute + LDS, read + stores | rows 64-127
        for (int q = 0; q < 5; ++q) e[q] = (unsigned)pst_sum[3 + q];
    }
#endif
    if constexpr (STATS) {   // the last tile's row sums
        __syncthreads();
        wg_stats_combine(spart, FP8 ? wave * 64 + lane_now() : tid, g.stats_part, g.tiles_n * g.stats_mpad, sp_off);
    }
}

// The persistent kernel addresses A and W through 32-bit buffer offsets that reach up to 255 rows past the last one (those read as zero): both
// operands, padded to whole tiles, must stay below 4 GiB.
static bool wg_pp_operands_ok(int M, int N, long lda, long ldw) {
    return (long)(M + 256) * lda < (1L << 31) && (long)(N + 256) * ldw < (1L << 31);
}

static int launch_pp_persist(GemmArgs& g, hipStream_t st) {
    WG_REQUIRE(wg_pp_operands_ok(g.M, g.N, g.lda, g.ldw), "persistent gemm: an operand (padded to whole 256-row tiles) is larger than 4 GiB");
    g.tiles_m = (g.M + 255) / 256;
    g.tiles_n = (g.N + 255) / 256;
    g.a_bytes = (unsigned)(((long)(g.M - 1) * g.lda + g.K) * 2);
    g.w_bytes = (unsigned)(((long)(g.N - 1) * g.ldw + g.K) * 2);
    {
        const long panel = 256L * g.K * 2, wbytes = (long)g.N * g.K * 2;
        const int cb = (int)((3L << 19) / (panel > 0 ? panel : 1));
        g.col_block = (wbytes > (3L << 20) && cb >= 2 && cb < g.tiles_n) ? cb : 0;
        if (g.col_block && (long)g.M * g.K * 2 > (128L << 20)) {
            // A (ViT-H at bs = 32: 335 MB) does not stay in the 256 MB Infinity Cache from one column block to the next, and its panels are 640 KiB:
            // narrow blocks then make every XCD stream 16 row panels at a time, at different rows than its neighbours.  What measured best on those
            // shapes (tools/bench_gemm_colblock.py, notes/r06_experiments.md section 16): blocks of 4-10 tile columns that are a whole number of XCD
            // ranges (an XCD owns nwg / 8 consecutive tiles of the order: the XCDs then walk the SAME row panels at the same time, in different
            // column blocks, and share them through the Infinity Cache) -- lin1 (20 tile columns): 5, -6.6 % -- and plain row-major when no such
            // width exists (qkv, proj: -2 ... -3 % against blocks of 2).
            const long nwg = (long)g.tiles_m * g.tiles_n;
            g.col_block = 0;
            for (int c = 4; c <= 10 && c < g.tiles_n; ++c)
                if (((long)g.tiles_m * c * 8) % nwg == 0) { g.col_block = c; break; }
        }
        static const char* force = getenv("WG_GEMM_COLBLOCK");      // experiments: tile order override (0 = row-major)
        if (force && *force) {
            g.col_block = atoi(force);
#ifndef WG_GEMM_EXPERIMENT
            if (g.col_block < 0) g.col_block = 0;      // the timing-only orders (resident 4 x 4 tiles, row bands) exist in diagnostic builds only
#else
            if (g.col_block < -1 && (((-g.col_block) >> 4) == 0 || ((-g.col_block) & 15) == 0)) g.col_block = 0;      // band height / block width 0
#endif
        }
    }
    constexpr int stage = 512 * 128, slab = 64 * (64 * 2 + 16);
    constexpr int base = 2 * stage + (8 - stage / slab) * slab;
    // + the double-buffered bias row / LayerNorm operands, the raw partial sums of the next tile (mode 2), the row-sum exchange (STATS)
    constexpr int lds_plain = base + 2 * 512, lds_stats = base + 2 * 512 + 256 * 4 * 8, lds_ln = base + 2 * 4096, lds_lnp = base + 2 * 4096 + 5 * 2048;
    static_assert(lds_lnp + 4096 <= 160 * 1024 && lds_stats + 4096 <= 160 * 1024, "LDS budget of the persistent GEMM");
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) {   // the attribute is per device (a process may drive several)
#define WG_PP_ATTR(LN_, ST_, F8_, BYTES_)                                                                                                                \
    (void)hipFuncSetAttribute((const void*)wg_gemm_pp_persist_kernel<LN_, ST_, F8_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES_);          \
    (void)hipFuncSetAttribute((const void*)wg_gemm_pp_persist_kernel<LN_, ST_, F8_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES_)
        WG_PP_ATTR(0, false, false, lds_plain); WG_PP_ATTR(0, true, false, lds_stats); WG_PP_ATTR(1, false, false, lds_ln); WG_PP_ATTR(2, false, false, lds_lnp);
        WG_PP_ATTR(0, false, true, lds_plain + 4096); WG_PP_ATTR(0, true, true, lds_stats + 4096); WG_PP_ATTR(2, false, true, lds_lnp + 4096);
#undef WG_PP_ATTR
    }
    const int nwg = g.tiles_m * g.tiles_n;
    const int cus = wg_cu_count(dev);
    const int grid = nwg < cus ? nwg : cus;   // one resident workgroup per CU; a multiple of 8 keeps a workgroup's tiles on one XCD
    // Which flow (template parameter SEAM): the seamless one whenever the number of slabs is even (its first-slab request lands in buffer 0
    // during the last slab); the round-1..5 flow for odd nk.
    const int nk = g.K / 64;
    bool seam = (nk & 1) == 0;
    static const char* force_seam = getenv("WG_GEMM_SEAM");      // =0: the round-1..5 flow everywhere (A/B runs)
    if (force_seam && *force_seam) seam = seam && atoi(force_seam) != 0;
#define WG_PP_LAUNCH(LN_, ST_, F8_, BYTES_)                                                                                           \
    do {                                                                                                                              \
        if (seam) hipLaunchKernelGGL((wg_gemm_pp_persist_kernel<LN_, ST_, F8_, true>), dim3(grid), dim3(512), BYTES_, st, g);         \
        else hipLaunchKernelGGL((wg_gemm_pp_persist_kernel<LN_, ST_, F8_, false>), dim3(grid), dim3(512), BYTES_, st, g);             \
    } while (0)
    if (g.mx_w) {   // fp8 operands with MX block scales: the same three epilogue flavours (4 KiB more LDS for the slabs' scale bytes)
        if (g.ln_part) WG_PP_LAUNCH(2, false, true, lds_lnp + 4096);
        else if (g.stats_part) WG_PP_LAUNCH(0, true, true, lds_stats + 4096);
        else WG_PP_LAUNCH(0, false, true, lds_plain + 4096);
    } else if (g.ln_part) WG_PP_LAUNCH(2, false, false, lds_lnp);
    else if (g.ln_stats) WG_PP_LAUNCH(1, false, false, lds_ln);
    else if (g.stats_part) WG_PP_LAUNCH(0, true, false, lds_stats);
    else WG_PP_LAUNCH(0, false, false, lds_plain);
#undef WG_PP_LAUNCH
    return wg_check_launch("wg_gemm_bias_act_bf16(ping-pong persistent)");
}

template <int BM, int BN, int BK, int STAGES, int WM, int WN, int PIPE = 0>
static int launch_tile(GemmArgs& g, hipStream_t st) {
    // staged (LDS-transposed, 16-byte) epilogue for bf16 outputs whose rows are 16-byte addressable
    const bool staged = !g.out_f32 && g.N % 8 == 0 && g.ldc % 8 == 0 && g.c_bytes != 0 &&
                        (!g.R || (g.ldr % 8 == 0 && ((uintptr_t)g.R & 15) == 0 && g.r_bytes != 0));
    return staged ? launch_tile_impl<BM, BN, BK, STAGES, WM, WN, true, PIPE>(g, st)
                  : launch_tile_impl<BM, BN, BK, STAGES, WM, WN, false, PIPE>(g, st);
}

// Tile choice (measured, tools/bench_gemm.py and bench.py with WG_GEMM_TILE=n on MI355X): the 256x256 kernel
// (8 waves, 1 workgroup / CU) is ~15 % better per tile-slot than the 128x128 one, and with the CLIP tower and the SAM
// branch on two streams a partially filled last round is back-filled by the other stream -- so wave quantisation
// decides only for shapes that cannot fill even one round of big tiles.
//
// allow_tail: also consider the tail-absorbing 128x128 kernel (tile 12).  It wins by 15-40 % per GEMM on M = B*1025
// shapes when the GEMMs run back to back on one stream (+3 % end to end), but its 68 KiB workgroups cannot share a CU
// with the other stream's 128 KiB ones, and with the two-stream overlap it measured 3-4 % SLOWER end to end -- so the
// host only asks for it in single-stream runs.
extern "C" int wg_gemm_pick_tile_ex(int M, int N, int allow_tail) {
    if (M <= 16 && N % 16 == 0) return 5;   // a handful of rows: weight streaming spread over N / 16 workgroups (needs K % 128 == 0)
    if (M <= 128) return 1;             // decoder token-side linears: one or two workgroups
    if (allow_tail && M % 128 >= 1 && M % 128 <= 16 && N % 8 == 0) {
        // ragged M just above a multiple of 128 (CLIP: 8 * 1025): fold the leftover rows into the last row tile when that
        // removes a nearly empty round of 128x128 tiles and the big tiles would not fill the chip well either
        const long t = (long)(M / 128) * ((N + 127) / 128);
        const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
        const double e128 = (double)t / (double)(((t + 511) / 512) * 512);
        const double e256 = (double)t256 / (double)(((t256 + 255) / 256) * 256);
        if (e128 >= 0.9 && e128 > e256) return 12;
    }
    if (N < 256 || (N < 512 && N % 256 != 0 && N % 128 == 0))
        return 11;                      // N = 128 / 384 (decoder image-side projections, 64->128 transposed conv): 128x128 tiles
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (t256 < 64 && t128 > t256) return 11;  // too few big tiles to matter: spread over more CUs
    // round quantisation: 256x256 tiles run one workgroup per CU (256 slots), 128x128 tiles two (512 slots).  The big tile's
    // main loop is ~15-20 % faster, so the small one wins only when it wastes clearly fewer slots in its last round
    // (SAM proj / lin2: 384 big tiles = 1.5 rounds vs 1536 small ones = 3 whole rounds: +7..12 % for that GEMM alone).
    // Only for callers that run one stream (allow_tail): with the towers overlapped on several streams the other
    // stream's kernels fill those slots anyway and the rule costs 1-2 % end to end.
    if (allow_tail) {
        const double e256 = (double)t256 / (double)(((t256 + 255) / 256) * 256);
        const double e128 = (double)t128 / (double)(((t128 + 511) / 512) * 512);
        if (e128 > 1.2 * e256) return 11;
    }
    return 16;                                // 256x256 tiles, ping-pong schedule, persistent (falls back to 14 = one tile per workgroup
                                              // when the output cannot take the staged epilogue)
}

extern "C" int wg_gemm_pick_tile(int M, int N) { return wg_gemm_pick_tile_ex(M, N, 0); }
// The same choice with K in view: exactly the kernel wg_gemm_bias_act_bf16 launches for a shape whose operands qualify for the MFMA
// path (the skinny kernel needs K % 128 == 0 and falls back to the 128x128 tiles otherwise) -- host-side accounting asks this one.
// A few dozen rows (MSQP's query tokens x batch: M = 32..96 at 1024 / 4096 columns): the 128x128 tiles put ONE workgroup per 128
// columns and walk K in a single chain (14 us at K = 1024, 40 us at K = 4096, whatever M <= 128 is); the skinny kernel spreads the same
// weights over (M / 16) x (N / 16) workgroups: 5 us + ~6.6 ns per workgroup and 1024 of K (tools/micro/skinny_bench.py under rocprofv3).
static bool wg_skinny_wins(int M, int N, int K) {
    if (M > 128 || N % 16 != 0 || K % 128 != 0) return false;
    const double kk = K / 1024.0;
    const double blocks = (double)((M + 15) / 16) * (N / 16);
    return 4.5 + (K >= 2048 ? 5.0 : 0.0) + blocks * kk * 0.0066 < 14.0 * (kk < 1.0 ? 1.0 : kk);
}
extern "C" int wg_gemm_pick_tile_mnk(int M, int N, int K, int allow_tail) {
    const int t = wg_gemm_pick_tile_ex(M, N, allow_tail);
    if (t == 1 && wg_skinny_wins(M, N, K)) return 5;
    return (t == 5 && K % 128 != 0) ? 1 : t;
}

struct GemmStatsIO {   // row statistics travelling between two persistent GEMMs (see wg_gemm_pp_persist_kernel)
    const float* ln_part = nullptr; int ln_np = 0; long ln_mpad = 0; float ln_eps = 0.f;   // consumer
    float* stats_part = nullptr; long stats_mpad = 0;                                      // producer
};
static int wg_gemm_dispatch(const void* A, long lda, const void* W, long ldw, const void* bias, const void* residual, long ldr,
                            int res_row_mod, void* C, long ldc, int M, int N, int K, int act, int out_f32, int tile_hint,
                            const float* ln_stats, const float* ln_s, const float* ln_b, void* stream, const bf16* sk_gamma = nullptr,
                            const bf16* sk_beta = nullptr, float sk_eps = 0.f, int sk_tiled = 0, const GemmStatsIO* sio = nullptr);

extern "C" int wg_gemm_bias_act_bf16(const void* A, long lda, const void* W, long ldw, const void* bias,
                                     const void* residual, long ldr, int res_row_mod, void* C, long ldc, int M, int N,
                                     int K, int act, int out_f32, int tile_hint, void* stream) {
    return wg_gemm_dispatch(A, lda, W, ldw, bias, residual, ldr, res_row_mod, C, ldc, M, N, K, act, out_f32, tile_hint, nullptr, nullptr,
                            nullptr, stream);
}

// Skinny rows, optionally with the LayerNorm in front and / or the weight matrix in fragment order:
//     C = act(LN?(A; gamma, beta, eps) . W^T + b)  for M <= 128, N % 16 == 0, K % 128 == 0       (text_hidden_fcs[0], utils_walkgpt.py:321-323)
// gamma == beta == null: no LayerNorm.  w_tiled: W was re-laid by wg_tile_weight_bf16 (ldw ignored).  wg_gemm_skinny_ln_supported() tells the
// caller whether the shape qualifies.
extern "C" int wg_gemm_skinny_ln_supported(int M, int N, int K, long lda, long ldw, long ldc) {
    return (M >= 1 && M <= 128 && N % 16 == 0 && K % 128 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0) ? 1 : 0;
}
extern "C" int wg_gemm_skinny_ln_bias_act_bf16(const void* A, long lda, const void* gamma, const void* beta, float eps, const void* W, long ldw,
                                               int w_tiled, const void* bias, void* C, long ldc, int M, int N, int K, int act, int out_f32,
                                               void* stream) {
    WG_REQUIRE((gamma != nullptr) == (beta != nullptr) && (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "gemm_skinny: LayerNorm operands");
    WG_REQUIRE(wg_gemm_skinny_ln_supported(M, N, K, lda, w_tiled ? 8 : ldw, ldc) && (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) == 0,
               "gemm_skinny: shape M=%d N=%d K=%d is not a skinny one", M, N, K);
    return wg_gemm_dispatch(A, lda, W, w_tiled ? K : ldw, bias, nullptr, 0, 0, C, ldc, M, N, K, act, out_f32, 5, nullptr, nullptr, nullptr, stream,
                            (const bf16*)gamma, (const bf16*)beta, eps, w_tiled);
}

// C = act(LayerNorm(A) . W^T + b) with the LayerNorm folded in: A = the raw rows, Wg = W * gamma (bf16), colsum[n] = sum_k Wg[n,k],
// bias_f32[n] = b[n] + sum_k W[n,k] beta[k], stats[m] = {mean, rstd} of row m of A (wg_row_stats_bf16; the buffer must hold
// M rounded up to an even number of rows, 16-byte aligned: rows travel to LDS in pairs).  Runs on the persistent
// 256x256 kernel only: wg_gemm_ln_supported() tells the caller whether this shape / alignment qualifies.
extern "C" int wg_gemm_ln_supported(int M, int N, int K, long lda, long ldw, long ldc) {
    return (wg_gemm_pick_tile(M, N) == 16 && K % 64 == 0 && N % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 &&
            (long)(M + 256) * lda < (1L << 31) && (long)(N + 256) * ldw < (1L << 31) && ((long)(M - 1) * ldc + N) * 2 < (1L << 31)) ? 1 : 0;      // (operands: wg_pp_operands_ok)
}
extern "C" int wg_gemm_ln_bias_act_bf16(const void* A, long lda, const void* Wg, long ldw, const float* bias_f32, const float* colsum,
                                        const float* stats, void* C, long ldc, int M, int N, int K, int act, void* stream) {
    WG_REQUIRE(bias_f32 && colsum && stats, "gemm_ln: null operand");
    WG_REQUIRE((((uintptr_t)bias_f32 | (uintptr_t)colsum) & 15) == 0 && ((uintptr_t)stats & 15) == 0, "gemm_ln: misaligned operand");
    WG_REQUIRE(wg_gemm_ln_supported(M, N, K, lda, ldw, ldc) && ((uintptr_t)C & 15) == 0, "gemm_ln: shape M=%d N=%d K=%d does not run on the persistent 256x256 kernel", M, N, K);
    return wg_gemm_dispatch(A, lda, Wg, ldw, nullptr, nullptr, 0, 0, C, ldc, M, N, K, act, 0, 16, stats, colsum, bias_f32, stream);
}

// The producing side of the row statistics: wg_gemm_bias_act_bf16 that also leaves, for a later LayerNorm over its output rows,
// row_partials[tile column][mpad][2] = {sum, sum of squares} (fp32) of the bf16 values it stored, per row and 256-column tile
// (mpad = M rounded up to 256; N / 256 planes).  Replaces the wg_row_stats_bf16 pass in front of wg_gemm_ln_bias_act_bf16
// (SAM: image_encoder.py:177-178,191 norm1 / norm2 behind lin2 / proj; CLIP: layer_norm1 / 2 behind fc2 / out_proj).
// Shapes: the persistent 256x256 kernel with N % 256 == 0 (wg_gemm_row_partials_supported).
extern "C" int wg_gemm_row_partials_supported(int M, int N, int K, long lda, long ldw, long ldc) {
    return (wg_gemm_ln_supported(M, N, K, lda, ldw, ldc) && N % 256 == 0 && N <= 1280 &&
            ((long)(N / 256) * (long)((M + 255) / 256 * 256) * 8 < (1L << 31))) ? 1 : 0;
}
extern "C" int wg_gemm_bias_act_stats_bf16(const void* A, long lda, const void* W, long ldw, const void* bias, const void* residual, long ldr,
                                           int res_row_mod, void* C, long ldc, int M, int N, int K, int act, float* row_partials, long mpad,
                                           void* stream) {
    WG_REQUIRE(row_partials && ((uintptr_t)row_partials & 15) == 0, "gemm_stats: row_partials must be a 16-byte aligned buffer");
    WG_REQUIRE(mpad % 256 == 0 && mpad >= M, "gemm_stats: mpad = %ld must be M = %d rounded up to a multiple of 256", mpad, M);
    WG_REQUIRE(wg_gemm_row_partials_supported(M, N, K, lda, ldw, ldc) && ((uintptr_t)C & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) &&
                   (!residual || (ldr % 8 == 0 && ((uintptr_t)residual & 15) == 0)),
               "gemm_stats: shape M=%d N=%d K=%d does not run on the persistent 256x256 kernel with whole column tiles", M, N, K);
    GemmStatsIO sio;
    sio.stats_part = row_partials; sio.stats_mpad = mpad;
    return wg_gemm_dispatch(A, lda, W, ldw, bias, residual, ldr, res_row_mod, C, ldc, M, N, K, act, 0, 16, nullptr, nullptr, nullptr, stream,
                            nullptr, nullptr, 0.f, 0, &sio);
}
// The consuming side: wg_gemm_ln_bias_act_bf16 with the statistics given as the producer's partial sums (n_partials planes of mpad
// rows; n_partials * 256 = K = the LayerNorm width, n_partials <= 5).  eps: the LayerNorm's.
extern "C" int wg_gemm_lnp_bias_act_bf16(const void* A, long lda, const void* Wg, long ldw, const float* bias_f32, const float* colsum,
                                         const float* row_partials, int n_partials, long mpad, float eps, void* C, long ldc, int M, int N, int K,
                                         int act, void* stream) {
    WG_REQUIRE(bias_f32 && colsum && row_partials, "gemm_lnp: null operand");
    WG_REQUIRE((((uintptr_t)bias_f32 | (uintptr_t)colsum | (uintptr_t)row_partials) & 15) == 0, "gemm_lnp: misaligned operand");
    WG_REQUIRE(n_partials >= 1 && n_partials <= 5 && n_partials * 256 == K, "gemm_lnp: %d partial planes do not cover K = %d", n_partials, K);
    WG_REQUIRE(mpad % 256 == 0 && mpad >= M, "gemm_lnp: mpad = %ld must be M = %d rounded up to a multiple of 256", mpad, M);
    WG_REQUIRE(wg_gemm_ln_supported(M, N, K, lda, ldw, ldc) && ((uintptr_t)C & 15) == 0, "gemm_lnp: shape M=%d N=%d K=%d does not run on the persistent 256x256 kernel", M, N, K);
    GemmStatsIO sio;
    sio.ln_part = row_partials; sio.ln_np = n_partials; sio.ln_mpad = mpad; sio.ln_eps = eps;
    return wg_gemm_dispatch(A, lda, Wg, ldw, nullptr, nullptr, 0, 0, C, ldc, M, N, K, act, 0, 16, nullptr, colsum, bias_f32, stream,
                            nullptr, nullptr, 0.f, 0, &sio);
}

static int wg_gemm_dispatch(const void* A, long lda, const void* W, long ldw, const void* bias, const void* residual, long ldr,
                            int res_row_mod, void* C, long ldc, int M, int N, int K, int act, int out_f32, int tile_hint,
                            const float* ln_stats, const float* ln_s, const float* ln_b, void* stream, const bf16* sk_gamma,
                            const bf16* sk_beta, float sk_eps, int sk_tiled, const GemmStatsIO* sio) {
    WG_REQUIRE(A && W && C, "gemm: null operand");
    WG_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
    WG_REQUIRE(act >= 0 && act <= 3, "gemm: bad activation %d", act);
    WG_REQUIRE(lda >= K && ldw >= K && ldc >= N, "gemm: leading dimension smaller than the row");
    WG_REQUIRE(!residual || ldr >= N, "gemm: residual leading dimension smaller than N");
    GemmArgs g;
    g.A = (const bf16*)A; g.lda = lda; g.W = (const bf16*)W; g.ldw = ldw;
    g.bias = (const bf16*)bias; g.R = (const bf16*)residual; g.ldr = ldr; g.res_mod = res_row_mod;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.act = act; g.out_f32 = out_f32;
    g.tiles_m = g.tiles_n = 0;
    g.col_block = 0;
    g.ln_stats = ln_stats; g.ln_s = ln_s; g.ln_b = ln_b;
    g.ln_part = sio ? sio->ln_part : nullptr; g.ln_np = sio ? sio->ln_np : 0; g.ln_mpad = sio ? sio->ln_mpad : 0; g.ln_eps = sio ? sio->ln_eps : 0.f;
    g.stats_part = sio ? sio->stats_part : nullptr; g.stats_mpad = sio ? sio->stats_mpad : 0;
    g.scale_a = g.scale_w = nullptr;
    g.sk_gamma = sk_gamma; g.sk_beta = sk_beta; g.sk_eps = sk_eps; g.sk_tiled = sk_tiled;
    {   // byte extents for the staged epilogue's buffer descriptors (it addresses C and R with 32-bit byte offsets)
        const long cb = ((long)(M - 1) * ldc + N) * 2;
        const long rrows = res_row_mod > 0 ? (res_row_mod < M ? res_row_mod : M) : M;
        const long rb = residual ? ((rrows - 1) * ldr + N) * 2 : 0;
        g.c_bytes = cb < (1L << 31) ? (unsigned)cb : 0;
        g.r_bytes = rb < (1L << 31) ? (unsigned)rb : 0;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool mfma_ok = (K % 64 == 0) && (N % 4 == 0) && (lda % 8 == 0) && (ldw % 8 == 0) && (ldc % 4 == 0) &&
                         (!residual || ldr % 4 == 0) && (((uintptr_t)A | (uintptr_t)W) % 16 == 0) &&
                         ((uintptr_t)C % 16 == 0) && (!bias || (uintptr_t)bias % 8 == 0) &&
                         (!residual || (uintptr_t)residual % 8 == 0) && N >= 16;
    if (!mfma_ok || tile_hint == 3) {
        hipLaunchKernelGGL(wg_gemm_rowwave_kernel, dim3((M + 3) / 4), dim3(256), 0, st, g);
        return wg_check_launch("wg_gemm_bias_act_bf16(rowwave)");
    }
    int tile = tile_hint;
    if (tile <= 0) tile = wg_gemm_pick_tile_mnk(M, N, K, 0);
    const bool can_stage = !out_f32 && N % 8 == 0 && ldc % 8 == 0 && g.c_bytes != 0 &&
                           (!residual || (ldr % 8 == 0 && ((uintptr_t)residual & 15) == 0 && g.r_bytes != 0));
    const bool small_ops = (long)M * lda < (1L << 31) && (long)N * ldw < (1L << 31);
    if (tile == 11 && !(can_stage && small_ops)) tile = 1;
    if (tile == 17) {   // the experimental one-barrier persistent kernel (tools/micro/gemm_fr.hip): diagnostic builds only (tools/build_variant.py <tag> -DWG_GEMM_FR)
#ifdef WG_GEMM_FR
        if (can_stage && small_ops && wg_gemm_fr_supports(g)) return wg_launch_gemm_fr(g, st);
#endif
        tile = 16;
    }
    if (tile == 16 && !(can_stage && small_ops && wg_pp_operands_ok(M, N, lda, ldw) && (!bias || ((uintptr_t)bias % 16 == 0 && N % 8 == 0)))) {
        WG_REQUIRE(!(g.ln_part || g.stats_part || g.ln_stats), "gemm: this operand layout cannot take the persistent kernel's staged epilogue");
        tile = 14;
    }
    if (tile == 5 && !(M <= 128 && N % 16 == 0 && K % 128 == 0)) tile = 1;
    if (tile != 1 && tile != 2 && tile != 5 && tile != 11 && tile != 12 && tile != 14 && tile != 16) tile = 1;
    if (tile == 12 && !(can_stage && M >= 128 && M % 128 >= 1 && M % 128 <= 16)) tile = 1;
    switch (tile) {
        case 5:
            {
                const bool wide = K >= 2048 && K % 512 == 0;
                const int variant = (wide ? 4 : 0) | (g.sk_gamma ? 2 : 0) | (g.sk_tiled ? 1 : 0);
                const dim3 grid(N / 16, (M + 15) / 16), blk(wide ? 1024 : 256);
                switch (variant) {
                    case 0: hipLaunchKernelGGL((wg_gemm_skinny_kernel<4, false, false>), grid, blk, 0, st, g); break;
                    case 1: hipLaunchKernelGGL((wg_gemm_skinny_kernel<4, false, true>), grid, blk, 0, st, g); break;
                    case 2: hipLaunchKernelGGL((wg_gemm_skinny_kernel<4, true, false>), grid, blk, 0, st, g); break;
                    case 3: hipLaunchKernelGGL((wg_gemm_skinny_kernel<4, true, true>), grid, blk, 0, st, g); break;
                    case 4: hipLaunchKernelGGL((wg_gemm_skinny_kernel<16, false, false>), grid, blk, 0, st, g); break;
                    case 5: hipLaunchKernelGGL((wg_gemm_skinny_kernel<16, false, true>), grid, blk, 0, st, g); break;
                    case 6: hipLaunchKernelGGL((wg_gemm_skinny_kernel<16, true, false>), grid, blk, 0, st, g); break;
                    default: hipLaunchKernelGGL((wg_gemm_skinny_kernel<16, true, true>), grid, blk, 0, st, g); break;
                }
            }
            return wg_check_launch("wg_gemm_bias_act_bf16(skinny)");
        case 12: return launch_tail(g, st);                            // 128x128 tiles, last row tile absorbs M % 128 <= 16 rows
        case 11: return launch_persist<128, 128, 2, 2>(g, st);         // persistent 128x128 tiles, 2 workgroups / CU
        case 2: return launch_tile<256, 256, 64, 2, 2, 4>(g, st);      // 256x256, plain two-slab loop (best at K >= 8192)
        case 14: return launch_tile<256, 256, 64, 2, 2, 4, 2>(g, st);  // 256x256, ping-pong between the two waves of each SIMD
        case 16: return launch_pp_persist(g, st);                      // the same loop, persistent tiles (first slab / store drain hidden)
        default: return launch_tile<128, 128, 64, 2, 2, 2>(g, st);  //  64 KiB LDS, 4 waves, 2 workgroups / CU
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) GEMM with per-row / per-output-channel scales (BASELINE config C5):
//     C[M,N] (bf16) = act( scale_a[m] scale_w[n] sum_k Aq[m,k] Wq[n,k] + bias[n] ) (+ R)
// Persistent 256x256 fp8 GEMM with OCP-MX block scales on both operands (wg_gemm_pp_persist_kernel<.., FP8 = true>):
//   C[M,N] bf16 = act(sum_k deq(Aq)[m,k] deq(Wq)[n,k] + bias[n]) (+ residual),  deq(x)[r,k] = e4m3(x[r,k]) * 2^(mx[k / 32][pos(r)] - 127).
// Aq [M,K], Wq [N,K] e4m3 bytes; a_mx [K/32][a_pitch] in the 128-row-group layout of GemmArgs::mx_a, w_mx [K/32][w_pitch] in the
// 64-row-group layout of GemmArgs::mx_w (both written by wg_quantize_mx_fp8 or by a producing GEMM's epilogue).  Optional, as in the
// bf16 kernel it shares its loop and epilogues with:
//   * LayerNorm of A's rows folded in (ln_colsum != null): Wq holds the gamma-scaled weight, ln_colsum [N] the row sums of its
//     DEQUANTISED values, ln_bias [N] = b + W beta (fp32; `bias` unused), ln_part [ln_np][ln_mpad][2] the {sum, sum of squares} partials
//     of the rows of the bf16 tensor Aq was quantised from (left by the GEMM that produced it); K = ln_np * 256;
//   * stats_part != null (N % 256 == 0): this GEMM leaves those partials for its own output;
//   * Cq != null (N % 32 == 0): the stored values once more as e4m3 [M][ldcq] + block scales c_mx [N/32][c_pitch] -- the A operand of
//     the next call; C may then be null (MLP hidden layer: only the fp8 form is ever read).
extern "C" int wg_gemm_mxfp8(const void* Aq, long lda, const void* a_mx, long a_pitch, const void* Wq, long ldw, const void* w_mx, long w_pitch,
                             const void* bias, const float* ln_colsum, const float* ln_bias, const float* ln_part, int ln_np, long ln_mpad,
                             float ln_eps, const void* residual, long ldr, int res_row_mod, void* C, long ldc, void* Cq, long ldcq, void* c_mx,
                             long c_pitch, float* stats_part, long stats_mpad, int M, int N, int K, int act, void* stream) {
    WG_REQUIRE(Aq && Wq && (C || Cq) && a_mx && w_mx, "gemm_mxfp8: null operand");
    WG_REQUIRE(M > 0 && N > 0 && K > 0 && K % 128 == 0, "gemm_mxfp8: K = %d must be a positive multiple of 128", K);
    WG_REQUIRE(act >= 0 && act <= 3, "gemm_mxfp8: bad activation %d", act);
    WG_REQUIRE(lda >= K && ldw >= K && (!C || ldc >= N) && lda % 16 == 0 && ldw % 16 == 0, "gemm_mxfp8: leading dimensions must cover the row and be multiples of 16");
    WG_REQUIRE(N % 8 == 0 && ldc % 8 == 0 && (!residual || (ldr >= N && ldr % 8 == 0)), "gemm_mxfp8: N, ldc, ldr must be multiples of 8");
    WG_REQUIRE((((uintptr_t)Aq | (uintptr_t)Wq | (uintptr_t)C) & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) &&
                   (!residual || ((uintptr_t)residual & 15) == 0) && (((uintptr_t)a_mx | (uintptr_t)w_mx) & 3) == 0, "gemm_mxfp8: misaligned operand");
    WG_REQUIRE((long)M * lda < (1L << 31) && (long)N * ldw < (1L << 31), "gemm_mxfp8: operand larger than 2 GiB");
    const long mpad = ((long)M + 255) / 256 * 256, npad = ((long)N + 255) / 256 * 256;
    WG_REQUIRE(a_pitch >= mpad && a_pitch % 4 == 0 && w_pitch >= npad && w_pitch % 4 == 0 && (K / 32) * a_pitch < (1L << 31) && (K / 32) * w_pitch < (1L << 31),
               "gemm_mxfp8: scale pitches must be multiples of 4 covering M, N rounded up to 256 (%ld, %ld)", mpad, npad);
    WG_REQUIRE(!ln_colsum || (ln_bias && ln_part && ln_np >= 1 && ln_np <= 5 && ln_np * 256 == K && ln_mpad % 256 == 0 && ln_mpad >= M && N % 4 == 0 &&
                              (((uintptr_t)ln_colsum | (uintptr_t)ln_bias | (uintptr_t)ln_part) & 15) == 0),
               "gemm_mxfp8: folded LayerNorm needs colsum, bias, partial sums in K / 256 = %d planes of a multiple-of-256 pitch covering M", K / 256);
    WG_REQUIRE(!stats_part || (N % 256 == 0 && stats_mpad % 256 == 0 && stats_mpad >= M && !ln_colsum), "gemm_mxfp8: row partials need N %% 256 == 0 and a pitch covering M (not together with a folded LayerNorm)");
    WG_REQUIRE(!Cq || (c_mx && N % 32 == 0 && ldcq >= N && ldcq % 8 == 0 && c_pitch >= mpad && c_pitch % 4 == 0 && ((uintptr_t)Cq & 7) == 0 && ((uintptr_t)c_mx & 3) == 0),
               "gemm_mxfp8: the fp8 copy needs N %% 32 == 0, ldcq %% 8 == 0 and a scale pitch covering %ld rows", mpad);
    GemmArgs g;
    g.A = (const bf16*)Aq; g.lda = lda / 2; g.W = (const bf16*)Wq; g.ldw = ldw / 2;      // the kernel counts 2-byte elements: byte PAIRS
    g.bias = (const bf16*)bias; g.R = (const bf16*)residual; g.ldr = ldr; g.res_mod = res_row_mod;
    g.C = C; g.ldc = C ? ldc : 0; g.M = M; g.N = N; g.K = K / 2; g.act = act; g.out_f32 = 0;
    g.tiles_m = g.tiles_n = 0;
    g.col_block = 0;
    g.ln_stats = nullptr; g.ln_s = ln_colsum; g.ln_b = ln_bias;
    g.ln_part = ln_colsum ? ln_part : nullptr; g.ln_np = ln_np; g.ln_mpad = ln_mpad; g.ln_eps = ln_eps;
    g.stats_part = stats_part; g.stats_mpad = stats_mpad;
    g.scale_a = g.scale_w = nullptr;
    g.mx_a = (const unsigned char*)a_mx; g.mx_a_pitch = a_pitch; g.mx_w = (const unsigned char*)w_mx; g.mx_w_pitch = w_pitch;
    g.Cq = Cq; g.ldcq = ldcq; g.mx_c = (unsigned char*)(Cq ? c_mx : nullptr); g.mx_c_pitch = c_pitch;
    g.sk_gamma = g.sk_beta = nullptr; g.sk_eps = 0.f; g.sk_tiled = 0;
    const long cb = C ? ((long)(M - 1) * ldc + N) * 2 : 0;
    const long rrows = res_row_mod > 0 ? (res_row_mod < M ? res_row_mod : M) : M;
    const long rb = residual ? ((rrows - 1) * ldr + N) * 2 : 0;
    const long qb = Cq ? (long)(M - 1) * ldcq + N : 0, sb = Cq ? (long)(N / 32) * c_pitch : 0;
    WG_REQUIRE(cb < (1L << 31) && rb < (1L << 31) && qb < (1L << 31) && sb < (1L << 31), "gemm_mxfp8: output larger than 2 GiB");
    g.c_bytes = (unsigned)cb;
    g.r_bytes = (unsigned)rb;
    g.cq_bytes = (unsigned)qb;
    g.mx_c_bytes = (unsigned)sb;
    return launch_pp_persist(g, (hipStream_t)stream);
}

// fp8 GEMM with per-row (A) and per-output-channel (W) fp32 scales, for widths the MX kernel above does not take (and as the form a
// caller with its own row quantisation binds).
// Aq [M,K] and Wq [N,K] are e4m3 bytes, K-contiguous, produced by wg_quantize_rows_fp8 (activations: per call; weights: once).
// Runs the 256x256 ping-pong kernel with block-scaled MFMAs (see wg_gemm_kernel<..., FP8 = true>); K % 128 == 0 (one 128-byte slab),
// leading dimensions multiples of 16 bytes.
extern "C" int wg_gemm_fp8_bias_act(const void* Aq, long lda, const float* scale_a, const void* Wq, long ldw, const float* scale_w,
                                    const void* bias, const void* residual, long ldr, int res_row_mod, void* C, long ldc, int M, int N,
                                    int K, int act, void* stream) {
    WG_REQUIRE(Aq && Wq && C && scale_a && scale_w, "gemm_fp8: null operand");
    WG_REQUIRE(M > 0 && N > 0 && K > 0 && K % 128 == 0, "gemm_fp8: K = %d must be a positive multiple of 128", K);
    WG_REQUIRE(act >= 0 && act <= 3, "gemm_fp8: bad activation %d", act);
    WG_REQUIRE(lda >= K && ldw >= K && ldc >= N && lda % 16 == 0 && ldw % 16 == 0, "gemm_fp8: leading dimensions must cover the row and be multiples of 16");
    WG_REQUIRE(N % 8 == 0 && ldc % 8 == 0 && (!residual || (ldr >= N && ldr % 8 == 0)), "gemm_fp8: N, ldc, ldr must be multiples of 8");
    WG_REQUIRE((((uintptr_t)Aq | (uintptr_t)Wq | (uintptr_t)C) & 15) == 0 && (!bias || ((uintptr_t)bias & 7) == 0) &&
                   (!residual || ((uintptr_t)residual & 15) == 0), "gemm_fp8: misaligned operand");
    WG_REQUIRE((long)M * lda < (1L << 31) && (long)N * ldw < (1L << 31), "gemm_fp8: operand larger than 2 GiB");
    GemmArgs g;
    // the kernel's address arithmetic counts 2-byte elements: hand it byte PAIRS (a 128-byte slab = "64 elements")
    g.A = (const bf16*)Aq; g.lda = lda / 2; g.W = (const bf16*)Wq; g.ldw = ldw / 2;
    g.bias = (const bf16*)bias; g.R = (const bf16*)residual; g.ldr = ldr; g.res_mod = res_row_mod;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K / 2; g.act = act; g.out_f32 = 0;
    g.tiles_m = g.tiles_n = 0;
    g.col_block = 0;
    g.ln_stats = nullptr; g.ln_s = nullptr; g.ln_b = nullptr;
    g.ln_part = nullptr; g.ln_np = 0; g.ln_mpad = 0; g.ln_eps = 0.f; g.stats_part = nullptr; g.stats_mpad = 0;
    g.scale_a = scale_a; g.scale_w = scale_w;
    g.sk_gamma = g.sk_beta = nullptr; g.sk_eps = 0.f; g.sk_tiled = 0;
    const long cb = ((long)(M - 1) * ldc + N) * 2;
    const long rrows = res_row_mod > 0 ? (res_row_mod < M ? res_row_mod : M) : M;
    const long rb = residual ? ((rrows - 1) * ldr + N) * 2 : 0;
    WG_REQUIRE(cb < (1L << 31) && rb < (1L << 31), "gemm_fp8: output larger than 2 GiB");
    g.c_bytes = (unsigned)cb;
    g.r_bytes = (unsigned)rb;
    return launch_tile_impl<256, 256, 64, 2, 2, 4, true, 2, true>(g, (hipStream_t)stream);
}
