// Software-pipelined fused attention for head_dim 64 on gfx950: SAM's global attention (64 x 64 grid, decomposed rel-pos;
// image_encoder.py:235-260, 321-392) and, on request (wg_attn_pipe_mode), plain attention without a key bias (CLIP; custom_clip.py:50-104).
//
// Same arithmetic, LDS images and fragment maps as wg_attn_kernel (attn.hip): S^T = K . Q^T on mfma 32x32x16 so that a lane owns
// 32 scores of ONE query, P^T stays in registers as the B operand of O^T += V^T . P^T, V^T by ds_read_b64_tr_b16, K / V tiles of
// 64 keys by LDS-DMA from running per-lane pointers, lazy rescale (2^6), rel-pos width term as the C operand of the first S^T
// MFMA, height term folded into the exponent offset, p = exp2(s * scale * log2 e + offset) in fp32.  What differs is the ORDER of the
// work inside a wave:
//
//   wg_attn_kernel runs a tile as one dependent chain -- S^T MFMAs, maximum, exponentials, P.V MFMAs, barrier -- and with two
//   waves per SIMD in lockstep a tile costs the SUM of its matrix segment and its vector segment plus the LDS / barrier latencies
//   between them: ~3300 cycles for 1024 cycles of MFMA work per SIMD (stamps).
//   Here every wave runs a three-stage pipeline over the key tiles: in iteration t its matrix instructions are S^T of tile t+1 (into
//   the second score buffer), P.V of tile t-1 and that tile's row sums (an all-ones A operand: 4 MFMAs instead of 32 v_add), its
//   vector instructions the exponentials of tile t and the maximum of tile t+1 -- no data dependence between the streams.  The
//   instruction stream is hand-placed: every instruction of the loop body is its own `asm volatile` statement (hipcc keeps volatile asm
//   statements in program order and allocates their registers; the same pipeline written with builtins was scheduled into clusters:
//   rounds 1-3): one MFMA, then 2 v_fma + 2 v_exp + 1 v_cvt_pk of the exponential phase, the LDS fragment reads of the
//   MFMAs three to four slots ahead and the counted lgkmcnt wait of the next one.  The first four K fragments of an iteration are read
//   in the previous one (K tiles live in a ring of three), the row maximum, its cross-half exchange and the exponent offset are
//   finished inside the stream: between the last MFMA of an iteration and the barrier only the rescale decision is left.
//   Measured (B = 8, 12 heads, alone on the GPU): 580 us against 604 us; stamps: 2100 cycles per tile and SIMD at 2.0 GHz, issue-bound --
//   the two waves of a SIMD present ~1800 cycles of instruction issue per tile (MFMA 8, v_exp 8, everything else ~4 each; they do not
//   overlap), which is why variants that trade vector instructions for MFMAs one for one ran at the same speed (notes/r04_experiments.md).
//
// Hazards this file owns (the compiler sees opaque statements):
//   * LDS reads are asm: every consumer sits behind an `s_waitcnt lgkmcnt(N)` counted from the issue order written at the iteration
//     (LDS operations return in order; the loop body contains no scalar loads -- an s_load inside the loop would share the counter).
//   * MFMA results are read by vector instructions at least two MFMAs and their fillers later (maximum), one iteration later
//     (exponentials) or after the loop (O^T, row sums); P^T fragments written by v_cvt_pk are read by MFMAs one iteration later.  The
//     rescale decision for tile t+1 is taken behind the LAST MFMA of iteration t: everything still at the old maximum -- O^T, the row
//     sums, and the P^T fragments of tile t, none of whose MFMAs has been issued -- is rescaled together, so the decision never splits a
//     pending tile (cdna_hip_programming.md T13; tests/test_gpu_attention.py forces it late in the loop).
//   * Operands the COMPILER computes for an asm MFMA (ones, zero-initialised accumulators, the rel-pos C operand) are pinned two wait
//     states ahead of the first use: its hazard recogniser does not know the statement is an MFMA (tools/lint_asm_hazards.py).
//   * LDS-DMA: tile K(t+3) / V(t) are requested in iteration t into the slots whose last reads ended before the barrier that opens the
//     iteration; `s_waitcnt vmcnt(0)` + s_barrier close it.
#include "attn_common.h"
#include <type_traits>

template <int I, int N, class F> __device__ __forceinline__ void wg_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wg_static_for<I + 1, N>(f);
    }
}

// ---- one instruction per statement ------------------------------------------------------------------------------------------
// Every MFMA operand lives in arch VGPRs.  (Round 4 tried the accumulation half of the register file through "a" constraints: hipcc shuttles such values through
// v_accvgpr_write in front of the asm MFMAs without the wait states -- wrong rows; the diff is kept in notes/r04_attn_pipe_agpr.diff, not in the product sources.)
#define PA_F "v"
#define PA_FO "=v"
#define PA_FIO "+v"
// S^T (scores, arch VGPRs): A = K fragment, B = Q fragment
#define PA_MFMA_ACC(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : PA_F(a), PA_F(b))
#define PA_MFMA_C(d, a, b, c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : PA_F(a), PA_F(b), "v"(c))
#define PA_MFMA_Z(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : PA_F(a), PA_F(b))
// rank-3 offset update of the scores: both operands come off the vector ALU -- possibly materialised by the COMPILER right in front of this
// statement (its hazard recogniser does not know the statement is an MFMA: a v_mov of a zero dword one instruction ahead was read stale, and
// a stale NaN pattern times the other operand's zero poisoned whole rows).  The two wait states of that hazard therefore live inside.
#define PA_MFMA_VV(acc, a, b) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// O^T (accumulator file): A = V^T fragment, B = P^T (arch VGPRs)
#define PA_MFMA_O(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : PA_FIO(acc) : PA_F(a), "v"(b))
#define PA_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
#define PA_MAX3(d, a, b, c) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))
#define PA_FMA_S(d, a, s, c) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(s), "v"(c))
#define PA_EXP(d, a) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(a))
#define PA_ADD(acc, a) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(a))
#define PA_CVT(d, lo, hi) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi))

template <int OFF> __device__ __forceinline__ u32x4 pa_ds_read_b128(unsigned lds_addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : PA_FO(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF> __device__ __forceinline__ u32x2 pa_ds_read_tr(unsigned lds_addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : PA_FO(v) : "v"(lds_addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ float pa_ds_read_b32(unsigned lds_addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}

// Diagnostic build only (-DWG_ATTN_STAMP, tools/attn_pipe_stamps.py): lane 0 of every wave of workgroup 0 stores s_memtime at five points of
// its first 14 iterations.  The stamp drains the LDS queue (s_memtime shares lgkmcnt), so the build's timings are read as shares only.
#ifdef WG_ATTN_STAMP
__device__ unsigned* wg_attn_pipe_stamp_ptr = nullptr;
extern "C" int wg_debug_attn_pipe_stamps(unsigned* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(wg_attn_pipe_stamp_ptr), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
// segment sums in scalar registers (no store, no vector-memory operation inside the loop): k = 0 iteration start, 1 head done, 2 main stream done,
// 3 requested tiles landed, 4 barrier passed
#define PA_STAMP(k)                                                                                                   \
    do {                                                                                                              \
        unsigned long long now;                                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");                                   \
        if ((k) > 0) st_sum[(k) - 1] += now - st_prev;                                                                \
        st_prev = now;                                                                                                \
    } while (0)
#else
#define PA_STAMP(k) do { } while (0)
#endif

// S: 0 = plain (CLIP: no bias, a lone key past a multiple of 64 is folded in after the loop), 64 = SAM global attention.
// NW waves of 32 queries; needs an even number >= 4 of key tiles (the launcher checks).
template <int S, int NW>
__global__ __launch_bounds__(NW * 64, 2) void wg_attn_pipe_kernel(AttnArgs a) {
    constexpr int HD = 64, ROWB = 128, TILE = 64 * ROWB, TILE2 = 2 * TILE, KSTEPS = 4, DB = 2;
    constexpr bool GRID = (S > 0);
    static_assert(S == 0 || S == 64, "plain or the 64 x 64 grid");
    constexpr int SS = S * S;
    constexpr int SP = GRID ? S + 1 : 1;          // relh table row (fp32 words, odd => conflict-free)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KV_BYTES = 5 * TILE;             // K tiles: ring of three | V tiles: ring of two
    char* kv = smem;
    float* tab = (float*)(smem + KV_BYTES);        // grid: per-wave rel table

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql_lane = lane & 31, hi = lane >> 5;

    // ---- block id -> (batch, head, q chunk), as wg_attn_kernel ---------------------------------------------------------------
    const int bid = blockIdx.x;
    const int groups = a.B * a.heads;
    int grp, qc;
    if (a.qchunks > 1 && (groups & 7) == 0) {
        const int per = 8 * a.qchunks;
        const int blk = bid / per, rem = bid % per;
        grp = blk * 8 + (rem & 7);
        qc = rem >> 3;
    } else {
        grp = bid / a.qchunks;
        qc = bid % a.qchunks;
    }
    const int head = grp % a.heads;
    const int b = grp / a.heads;
    const int Lq = GRID ? SS : a.Lq;
    const int hcol = head * HD;

    // ---- this lane's query ------------------------------------------------------------------------------------------------------
    const int ql_raw = (qc * NW + wave) * 32 + ql_lane;
    const int ql = ql_raw < Lq ? ql_raw : Lq - 1;
    const bool qvalid = ql_raw < Lq;
    constexpr int SD = GRID ? S : 1;
    const int qh = GRID ? ql / SD : 0, qw = GRID ? ql % SD : 0;
    const long qrow = GRID ? (long)b * SS + ql : (long)b * a.q_bs + ql;
    bf16x8 qf[KSTEPS];
    {
        const bf16* qp = a.Q + qrow * a.ldq + hcol + 8 * hi;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) qf[s] = *(const bf16x8*)(qp + 16 * s);
    }

    // ---- K / V staging: one running per-lane source pointer per piece (1 KiB = 8 key rows), swizzle on the source address ---------
    constexpr int NPW = 8 / NW > 0 ? 8 / NW : 1;   // pieces per wave per tile and operand
    static_assert(NW == 8 || NW == 4, "eight or four waves");
    const bf16* runp[2][NPW];
    const int nkeys = GRID ? SS : a.Lk;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int ci = (wave + i * NW) * 64 + lane;
            const int row = ci >> 3, cs = ci & 7;
            const int c = cs ^ (o ? swzV<HD>(row) : swzK<HD>(row));
            const long r = (long)b * (GRID ? SS : a.k_bs) + row;
            runp[o][i] = (o ? a.V + r * a.ldv : a.K + r * a.ldk) + hcol + c * 8;
        }
    const unsigned strideK = (unsigned)(64L * a.ldk), strideV = (unsigned)(64L * a.ldv);
    // K tile -> ring slot `slot` (0..2) / V tile j -> slot j & 1; per operand the tiles are staged in order 0, 1, 2, ...  (every tile the
    // loop stages is a whole tile: the launcher guarantees 64 | number of keys walked)
    auto stage = [&](int slot, int o) __attribute__((always_inline)) {
        char* dst = kv + (o ? 3 * TILE + (slot & 1) * TILE : slot * TILE);
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(runp[o][i]), WG_LDS_PTR(dst + (wave + i * NW) * 1024), 16, 0, 0);
            runp[o][i] += o ? strideV : strideK;
        }
    };
    const bool lone_key = !GRID && (nkeys & 63) == 1;
    const int nt = nkeys / 64;
    stage(0, 0);
    stage(1, 0);
    stage(2, 0);

    // ---- rel-pos tables (grid), as wg_attn_kernel: width term -> 32 registers (C operand), height term -> LDS ---------------------------
    f32x16 relw_c[2];
    float* mytab = tab + wave * 32 * SP;
    if constexpr (GRID) {
        constexpr int NJB = (2 * S - 1 + 31) / 32;
        auto rel_pass = [&](int which) __attribute__((always_inline)) {
            const bf16* rel = which == 0 ? a.rel_w : a.rel_h;
            const int qpos = which == 0 ? qw : qh;
            for (int k = S + hi; k < SP; k += 2) mytab[ql_lane * SP + k] = NEG_BIG;
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) {
                int j = jb * 32 + ql_lane;
                j = j < 2 * S - 1 ? j : 2 * S - 2;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
                    const bf16x8 rf = *(const bf16x8*)(rel + (long)j * HD + 16 * s + 8 * hi);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf, qf[s], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jj = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const int kpos = qpos + S - 1 - jj;
                    if (jj < 2 * S - 1 && kpos >= 0 && kpos < S) mytab[ql_lane * SP + kpos] = acc[r] * LOG2E;
                }
            }
        };
        const float inv_sc2 = 1.0f / (a.scale * LOG2E);
        rel_pass(0);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int kw = (e & 3) + 8 * (e >> 2) + 4 * hi;
            relw_c[e >> 4][e & 15] = mytab[ql_lane * SP + kw] * inv_sc2;      // pre-divided by scale * log2 e: C operand of the first S^T MFMA
        }
        asm volatile("" ::: "memory");
        rel_pass(1);
    }
    // p = exp2(s * sc2 + noff): scale and offset stay on the vector ALU (one v_fma per score).  Measured alternative (round 4,
    // notes/r04_experiments.md): q pre-multiplied by scale * log2 e and the offset added by a rank-3 MFMA update of the scores -- 32 v_fma
    // fewer per tile, the same run time, and the second bf16 rounding of q cost 0.07-0.11 of absolute error on scores of several hundred
    // (tests/test_gpu_attention.py::test_attention_large_uneven_scores): not kept.
    const float sc2 = a.scale * LOG2E;
    // softmax denominators from the matrix pipe: osum += 1 . P^T (an all-ones A operand) beside every k-step of P.V -- every element of osum
    // ends up holding the row sum over BOTH lane halves' keys.  The loop is bound by instruction ISSUE (stamps + PMC): 4 MFMAs
    // (8 issue cycles each) replace 32 v_add (4 each), on a pipe that is ~55 % busy
    f32x16 osum;
#pragma unroll
    for (int r = 0; r < 16; ++r) osum[r] = 0.f;
    u32x4 ones_full = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};   // A operand of the row-sum MFMAs
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
    unsigned relh_ad = lds0 + KV_BYTES + (unsigned)((wave * 32 + ql_lane) * SP * 4);    // this lane's relh row, entry t

    // ---- fragment addresses ---------------------------------------------------------------------------------------------------------
    // K fragment of S^T MFMA (kb, s): row kb*32 + ql_lane, 16-byte chunk (2s + hi) ^ swzK(row); the swizzle does not depend on kb
    unsigned kad[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) kad[s] = lds0 + ql_lane * ROWB + (((2 * s + hi) ^ swzK<HD>(ql_lane)) << 4);
    // V^T fragments (see wg_attn_kernel): one lane-dependent base per d block, compile-time offsets per k-step
    unsigned vt_ad[DB];
    {
        const int g = lane >> 4, i16 = lane & 15;
        const int rq = i16 >> 2, cp = i16 & 3;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const int col = 32 * d + 16 * (g & 1) + 4 * cp;
            const int chunk = col >> 3;
            vt_ad[d] = lds0 + 3 * TILE + (4 * hi + rq) * ROWB + ((chunk ^ swzV<HD>(4 * hi + rq)) << 4) + (col & 7) * 2;
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Everything the compiler computed above and an asm MFMA below reads as an operand is pinned HERE, two wait states ahead of the first
    // of them: in one variant of this kernel hipcc placed the last v_cvt_pk of a query prescale directly in front of the first S^T MFMA
    // (opaque to its hazard recogniser), which then read the old register -- wrong rows that came and went with the register allocation
    // (tools/lint_asm_hazards.py scans the ISA for this pattern; tests/test_cabi_and_host.py runs it).
    asm volatile("s_nop 1" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
    asm volatile("s_nop 1" : "+v"(ones_full), "+v"(osum));
    if constexpr (GRID) asm volatile("s_nop 1" : "+v"(relw_c[0]), "+v"(relw_c[1]));

    if (!GRID && (qc * NW + wave) * 32 >= Lq) {
        // a wave without a single query (CLIP's 1025 = 32 blocks + 1): it stages its share of the tiles and keeps the barriers
        __builtin_amdgcn_s_barrier();
        for (int t = 0, slot = 0; t < nt; ++t) {
            if (t + 3 < nt) stage(slot, 0);
            slot = slot == 2 ? 0 : slot + 1;
            stage(t, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---- pipeline state ---------------------------------------------------------------------------------------------------------------
    f32x16 sa[2][2];      // [score buffer][key block]: S^T of tile t (being exponentiated) and of tile t+1 (being accumulated)
    u32x4 pf[2][4];       // [buffer][k-step]: P^T fragments of tile t (being written) and of tile t-1 (feeding P.V)
    f32x16 ot[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) pf[i][k] = (u32x4){0u, 0u, 0u, 0u};
    float m_run = NEG_BIG;
    constexpr float RESCALE_THR = 6.0f;
#ifdef WG_ATTN_STAMP
    unsigned long long st_sum[4] = {0, 0, 0, 0}, st_prev = 0;
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // S^T MFMA g = (kb = g >> 2, s = g & 3) into score buffer SB; its K fragment from the tile whose ring slot is baked into `ad`
    u32x4 kf[4];
    unsigned kx[KSTEPS], ky[KSTEPS];      // fragment addresses of the two K tiles an iteration reads (roles alternate, see iter)
    auto k_read = [&kf](auto gc, const unsigned (&ad)[KSTEPS]) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        constexpr int kb = g >> 2, s = g & 3;
        kf[s] = pa_ds_read_b128<kb * 32 * ROWB>(ad[s]);
    };
    auto qk_mfma = [&sa, &qf, &relw_c, &kf](auto gc, auto sbc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, SB = decltype(sbc)::value;
        constexpr int kb = g >> 2, s = g & 3;
        const bf16x8 kfr = __builtin_bit_cast(bf16x8, kf[s]);
        if constexpr (s == 0) {
            if constexpr (GRID) PA_MFMA_C(sa[SB][kb], kfr, qf[0], relw_c[kb]);
            else PA_MFMA_Z(sa[SB][kb], kfr, qf[0]);
        } else {
            PA_MFMA_ACC(sa[SB][kb], kfr, qf[s]);
        }
    };
    // P.V MFMA j = (ks = j >> 1, d = j & 1): V^T fragment pair in slot j & 3
    u32x2 vta[4], vtb[4];
    auto v_read = [&vta, &vtb, &vt_ad](auto jc, auto bufc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, BUF = decltype(bufc)::value;
        constexpr int ks = j >> 1, d = j & 1;
        vta[j & 3] = pa_ds_read_tr<BUF * TILE + ks * 16 * ROWB>(vt_ad[d]);
        vtb[j & 3] = pa_ds_read_tr<BUF * TILE + ks * 16 * ROWB + 8 * ROWB>(vt_ad[d]);
    };
    auto pv_mfma = [&ot, &pf, &vta, &vtb](auto jc, auto pbc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, PB = decltype(pbc)::value;
        constexpr int ks = j >> 1, d = j & 1;
        const u32x4 vv = {vta[j & 3][0], vta[j & 3][1], vtb[j & 3][0], vtb[j & 3][1]};
        PA_MFMA_O(ot[d], vv, pf[PB][ks]);
    };

    // running maximum of a score buffer: 16 v_max3 in four chains (an operand written by the statement two ahead makes hipcc pad an s_nop:
    // its hazard recogniser assumes the worst of an asm statement); ops 0-3 start the chains (elements 0..11), ops 4-13 extend them by two
    // elements each (12..31), ops 14, 15 join them into mt.  Ops 0-5 read key block 0 only.
    float mch[4], mt;
    auto max_op = [&mch, &mt, &sa](auto sbc, auto ic) __attribute__((always_inline)) {
        constexpr int SB = decltype(sbc)::value, i = decltype(ic)::value;
        if constexpr (i < 4) PA_MAX3(mch[i], sa[SB][0][3 * i], sa[SB][0][3 * i + 1], sa[SB][0][3 * i + 2]);
        else if constexpr (i < 14) {
            constexpr int e = 12 + 2 * (i - 4);
            PA_MAX3(mch[i & 3], mch[i & 3], sa[SB][e >> 4][e & 15], sa[SB][(e + 1) >> 4][(e + 1) & 15]);
        } else if constexpr (i == 14) PA_MAX3(mch[0], mch[0], mch[1], mch[2]);
        else PA_MAX3(mt, mch[0], mch[3], mch[3]);
    };
    // exponent offset of a tile: p = exp2(s * sc2 + noff), noff = rh - m_run (one instruction, redone by the rare rescale path)
    float mx = 0.f, noff = 0.f;
    auto offset_of = [&noff, &m_run](float rh) __attribute__((always_inline)) {
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(noff) : "v"(rh), "v"(m_run));
    };
    // maximum of the tile over both lane halves (the other half of a query's keys lives in lane ^ 32), height term added: 4 instructions
    auto exchange = [&mt, &mx, sc2](float rh) __attribute__((always_inline)) {
        float a_, b_;
        PA_FMA_S(a_, mt, sc2, rh);
        asm volatile("v_mov_b32 %0, %1" : "=v"(b_) : "v"(a_));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a_), "+v"(b_));
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(a_), "v"(b_));
    };
    // The rescale decision for the tile whose maximum sits in mx: lazy rescale of everything still at the old maximum -- O^T, the row sums and,
    // PEND >= 0, the P^T fragments of the tile whose P.V product has not been issued yet (buffer PEND) -- and the offset pieces again.
    auto decide = [&](float rh, auto pendc) __attribute__((always_inline)) {
        constexpr int PEND = decltype(pendc)::value;
        if (__any(mx > m_run + RESCALE_THR)) {
            // O^T and the row sums come out of MFMAs the hazard recogniser cannot see: the wait states are TIED to the registers (an untied
            // s_nop orders nothing: hipcc may schedule its own multiplies by alpha, ordered only behind the asm MFMA, ahead of the nop)
            asm volatile("s_nop 15\n\ts_nop 1" : "+v"(osum));
#pragma unroll
            for (int d = 0; d < DB; ++d) asm volatile("" : "+v"(ot[d]));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) osum[r] *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
            if constexpr (PEND >= 0) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned pr = pf[PEND][ks][w];
                        const float lo = __builtin_bit_cast(float, pr << 16) * alpha, hi_ = __builtin_bit_cast(float, pr & 0xFFFF0000u) * alpha;
                        unsigned o;
                        PA_CVT(o, lo, hi_);
                        pf[PEND][ks][w] = o;
                    }
            }
            offset_of(rh);
        }
    };
    using IM1 = std::integral_constant<int, -1>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- pre-loop: S^T of tile 0, its maximum and offset; the first K fragments of tile 1 -------------------------------------------------
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        kx[s] = kad[s] + 1 * TILE;       // K tile 1 (ring slot 1)
        ky[s] = kad[s] + 2 * TILE;       // K tile 2 (ring slot 2)
    }
    int kslot = 0;                       // ring slot of the K tile the next iteration requests (tile t + 3)
    {
        wg_static_for<0, 4>([&k_read, &kad](auto g) { k_read(g, kad); });
        wg_static_for<0, 8>([&k_read, &qk_mfma, &kad](auto gc) {
            constexpr int g = decltype(gc)::value;
            PA_LGKM(g < 4 ? 3 : 7 - g);
            qk_mfma(gc, I0{});
            if constexpr (g < 4) k_read(std::integral_constant<int, g + 4>{}, kad);
        });
        wg_static_for<0, 4>([&k_read, &kx](auto g) { k_read(g, kx); });
        float rh = 0.f;
        if constexpr (GRID) {
            rh = pa_ds_read_b32(relh_ad);
            relh_ad += 4;
        }
        // the S^T MFMAs' results before the first vector read: the wait states are tied to the score registers they cover (as at the rescale and behind
        // the loop; an untied s_nop orders nothing against compiler code that reads the scores)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sa[0][0]), "+v"(sa[0][1]));
        wg_static_for<0, 16>([&max_op](auto i) { max_op(I0{}, i); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        exchange(rh);
        offset_of(rh);
        decide(rh, IM1{});
        __builtin_amdgcn_s_barrier();
    }

    // ---- one iteration (P = t & 1, Q = P ^ 1).
    //   vector stream: the exponentials of tile t (score buffer P, offset already added) -> P^T buffer P; the maximum of tile t+1 (score
    //                  buffer Q), its exchange and offset pieces; behind the last MFMA the rescale decision and the offset MFMAs of tile t+1
    //   matrix stream: S^T of tile t+1 into score buffer Q (K fragments g0..g3 were read in the previous iteration, g4..g7 come from
    //                  address set A), then P.V of tile t-1 (P^T buffer Q, V tile in slot Q)
    //   LDS reads:     rh(t+1) | K g4..g7 of tile t+1 (behind MFMA 0..3) | V pairs 0..3 (behind MFMA 4..7) | behind P.V MFMA j = 0..3: V
    //                  pair j+4, then K g = j of tile t+2 (address set B) for the next iteration
    //   requests:      K tile t+3 -> ring slot kslot, V tile t -> slot P
    // QK: t + 1 < nt.  PV: t > 0.  KP: t + 2 < nt.
    auto iter = [&](int t, auto pc, auto do_qk_c, auto do_pv_c, auto do_kp_c, unsigned (&kA)[KSTEPS], unsigned (&kB)[KSTEPS]) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value, Q = P ^ 1;
        constexpr bool DO_QK = decltype(do_qk_c)::value, DO_PV = decltype(do_pv_c)::value, DO_KP = decltype(do_kp_c)::value;
        using BQ = std::integral_constant<int, Q>;
        PA_STAMP(0);
        if (t + 3 < nt) stage(kslot, 0);
        stage(t, 1);
        float rh = 0.f;
        if constexpr (GRID && DO_QK) {
            rh = pa_ds_read_b32(relh_ad);
            relh_ad += 4;
        }
        if constexpr (!DO_QK) wg_static_for<0, 4>([&v_read](auto j) { v_read(j, BQ{}); });
        PA_STAMP(1);
        // exponential phase, skewed over the MFMA gaps: group k = fma of elements 2k, 2k+1 | exp of 2k-2, 2k-1 | pack of 2k-6, 2k-5 (an
        // operand written by the statement just ahead would make hipcc pad an s_nop).  The offset of THIS tile is copied first: the tail
        // of the iteration overwrites noff with the next tile's while the last groups still run
        float x[32], pe[32];
        const float noff_t = noff;
        auto group = [&x, &pe, &sa, &pf, &noff_t, sc2](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < 16) {
                constexpr int e = 2 * k;
                PA_FMA_S(x[e], sa[P][e >> 4][e & 15], sc2, noff_t);
                PA_FMA_S(x[e + 1], sa[P][(e + 1) >> 4][(e + 1) & 15], sc2, noff_t);
            }
            if constexpr (k >= 1 && k < 17) {
                constexpr int e = 2 * k - 2;
                PA_EXP(pe[e], x[e]);
                PA_EXP(pe[e + 1], x[e + 1]);
            }
            if constexpr (k >= 3 && k < 19) {
                constexpr int e = 2 * k - 6;       // element e = 16*kb + r -> fragment kb*2 + (r >> 3), dword (r & 7) >> 1
                unsigned o;
                PA_CVT(o, pe[e], pe[e + 1]);
                pf[P][(e >> 4) * 2 + ((e & 15) >> 3)][(e & 7) >> 1] = o;
            }
        };
        group(I0{});
        // MFMA slots: 0..7 S^T g = m | 8..19: per k-step ks = (m-8)/3 two P.V MFMAs (j = 2ks, 2ks+1) and the row-sum MFMA
        wg_static_for<0, 20>([&group, &k_read, &v_read, &qk_mfma, &pv_mfma, &max_op, &exchange, &offset_of, &rh, &kA, &kB, &osum, &ones_full, &pf](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr bool RH = GRID && DO_QK;
            if constexpr (m < 8) {
                if constexpr (DO_QK) {
                    // m >= 4: K g = m landed?  issued behind it: K m+1..7, then the V pairs
                    if constexpr (m >= 4) PA_LGKM(DO_PV ? (m == 4 ? 3 : 2 * (m - 4) + (7 - m)) : 7 - m);
                    qk_mfma(mc, BQ{});
                    if constexpr (m < 4) k_read(std::integral_constant<int, m + 4>{}, kA);
                    else if constexpr (DO_PV) v_read(std::integral_constant<int, m - 4>{}, BQ{});
                }
            } else {
                constexpr int ks = (m - 8) / 3, r = (m - 8) % 3;
                if constexpr (r < 2) {
                    constexpr int j = 2 * ks + r;
                    if constexpr (DO_PV) {
                        // V pair j landed?  behind it, in issue order: [V j+1 .. V 3] [V 4, Kp 0, V 5, Kp 1, ...] as far as issued
                        if constexpr (!DO_QK) PA_LGKM(j < 5 ? 6 : 2 * (7 - j));
                        else if constexpr (DO_KP) PA_LGKM(j == 0 ? 6 : j == 1 ? 7 : j == 2 ? 8 : j == 3 ? 9 : j == 4 ? 10 : j == 5 ? 7 : j == 6 ? 4 : 1);
                        else PA_LGKM(j < 5 ? 6 : 2 * (7 - j));
                        pv_mfma(std::integral_constant<int, j>{}, BQ{});
                        if constexpr (j < 4) v_read(std::integral_constant<int, j + 4>{}, BQ{});
                    }
                    if constexpr (j < 4 && DO_KP) k_read(std::integral_constant<int, j>{}, kB);
                } else if constexpr (DO_PV) {
                    PA_MFMA_VV(osum, ones_full, pf[Q][ks]);      // row sums of tile t-1, k-step ks
                }
            }
            if constexpr (m < 18) group(std::integral_constant<int, m + 1>{});
            // maximum of tile t+1: key block 0 is complete behind slot 3, key block 1 behind slot 7 (+ the MFMA's own latency: the reads
            // below start two MFMAs and their fillers later); then the exchange and the offset pieces on the assumption of no rescale
            if constexpr (DO_QK) {
                if constexpr (m >= 5 && m <= 7) {
                    max_op(BQ{}, std::integral_constant<int, 2 * (m - 5)>{});
                    max_op(BQ{}, std::integral_constant<int, 2 * (m - 5) + 1>{});
                } else if constexpr (m >= 10 && m <= 13) {
                    max_op(BQ{}, std::integral_constant<int, 6 + 2 * (m - 10)>{});
                    max_op(BQ{}, std::integral_constant<int, 7 + 2 * (m - 10)>{});
                } else if constexpr (m == 14) {
                    max_op(BQ{}, std::integral_constant<int, 14>{});
                } else if constexpr (m == 15) {
                    max_op(BQ{}, std::integral_constant<int, 15>{});
                } else if constexpr (m == 16) {
                    if constexpr (RH) PA_LGKM(15);      // (rh is the oldest LDS operation of the iteration: long landed, and lgkmcnt is in order)
                    exchange(rh);
                    offset_of(rh);
                }
            }
        });
        PA_STAMP(2);
        if constexpr (DO_QK) {
            decide(rh, std::integral_constant<int, P>{});      // pending: P^T of tile t, just written
            // the fragment addresses of the K tile two iterations ahead take over set A (ring slot kslot, the one just requested)
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) kA[s] = kad[s] + (unsigned)(kslot * TILE);
            kslot = kslot == 2 ? 0 : kslot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        PA_STAMP(3);
        __builtin_amdgcn_s_barrier();
        PA_STAMP(4);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // t = 0 | 1 .. nt-3 | nt-2 | nt-1   (nt even, >= 4)
    iter(0, I0{}, T_{}, F_{}, T_{}, kx, ky);
    int t = 1;
    for (; t + 1 < nt - 2; t += 2) {
        iter(t, I1{}, T_{}, T_{}, T_{}, ky, kx);
        iter(t + 1, I0{}, T_{}, T_{}, T_{}, kx, ky);
    }
    iter(nt - 3, I1{}, T_{}, T_{}, T_{}, ky, kx);
    iter(nt - 2, I0{}, T_{}, T_{}, F_{}, kx, ky);
    iter(nt - 1, I1{}, F_{}, T_{}, F_{}, ky, kx);
    // P.V and row sums of the last tile (odd: P^T buffer 1, V tile in slot 1)
    {
        wg_static_for<0, 4>([&v_read](auto j) { v_read(j, I1{}); });
        wg_static_for<0, 8>([&v_read, &pv_mfma, &osum, &ones_full, &pf](auto jc) {
            constexpr int j = decltype(jc)::value;
            PA_LGKM(j < 5 ? 6 : 2 * (7 - j));
            pv_mfma(jc, I1{});
            if constexpr (j < 4) v_read(std::integral_constant<int, j + 4>{}, I1{});
            if constexpr (j & 1) PA_MFMA_VV(osum, ones_full, pf[1][j >> 1]);
        });
    }
    // the last MFMAs' results (opaque to the hazard recogniser) before the first vector read: wait states tied to the registers they cover
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(osum));
#pragma unroll
    for (int d = 0; d < DB; ++d) asm volatile("" : "+v"(ot[d]));
    float l_run = osum[0];      // (every element holds the row sum over both lane halves' keys)
#ifdef WG_ATTN_STAMP
    if (blockIdx.x < 8 && lane == 0 && wg_attn_pipe_stamp_ptr)
#pragma unroll
        for (int k = 0; k < 4; ++k) wg_attn_pipe_stamp_ptr[(blockIdx.x * 8 + wave) * 4 + k] = (unsigned)st_sum[k];
    if (blockIdx.x < 8 && lane == 0 && wave == 0 && wg_attn_pipe_stamp_ptr) {      // loop length in core cycles and in 100 MHz ticks: the clock the chip held
        wg_attn_pipe_stamp_ptr[256 + blockIdx.x * 2] = (unsigned)(__builtin_amdgcn_s_memtime() - st_c0);
        wg_attn_pipe_stamp_ptr[256 + blockIdx.x * 2 + 1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
    }
#endif

    if constexpr (!GRID) {
        if (lone_key) {      // the key past the last whole tile (CLIP's class token), on the vector ALU: as wg_attn_kernel
            const long krow = (long)b * a.k_bs + (nkeys - 1);
            const bf16* kp = a.K + krow * a.ldk + hcol + 8 * hi;
            float dot = 0.f;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const bf16x8 kfv = *(const bf16x8*)(kp + 16 * s);
#pragma unroll
                for (int e = 0; e < 8; ++e) dot += (float)qf[s][e] * (float)kfv[e];
            }
            const float sv = wg_xor32_sum(dot) * sc2;
            const float m_new = fmaxf(m_run, sv);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            const float pl = __builtin_amdgcn_exp2f(sv - m_new);
            m_run = m_new;
            l_run = l_run * alpha + pl;
            const bf16* vp = a.V + krow * a.ldv + hcol + 4 * hi;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const bf16x4 v4 = *(const bf16x4*)(vp + 32 * d + 8 * g4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ot[d][g4 * 4 + e] = ot[d][g4 * 4 + e] * alpha + pl * (float)v4[e];
                }
        }
    }
    // ---- epilogue: O = O^T / l, 8-byte stores ----------------------------------------------------------------------------------------
    const float l_tot = l_run;
    if (GRID && a.Oq) {            // fp8 chain: e4m3 + block scales instead of bf16 (uniform branch)
        wg_attn_store_mx<DB>(ot, 1.0f / l_tot, qvalid, qrow, hcol, hi, a);
        return;
    }
    if (qvalid) {
        const float inv = 1.0f / l_tot;
        const long orow = GRID ? qrow : (long)b * a.o_bs + ql;
        bf16* op = a.O + orow * a.ldo + hcol;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(ot[d][g4 * 4 + e] * inv);
                *(bf16x4*)(op + 32 * d + 8 * g4 + 4 * hi) = o;
            }
    }
}

template <int S, int NW>
static int launch_pipe(const AttnArgs& a, hipStream_t st) {
    size_t lds = 5 * 64 * 128;      // K ring of three, V ring of two
    if (S > 0) lds += (size_t)NW * 32 * (S + 1) * 4;
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) (void)hipFuncSetAttribute((const void*)wg_attn_pipe_kernel<S, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((wg_attn_pipe_kernel<S, NW>), dim3(a.B * a.heads * a.qchunks), dim3(NW * 64), lds, st, a);
    return wg_check_launch("wg_attn(pipelined)");
}

// Which cases the pipelined kernel takes: 0 none, 1 (default) SAM global attention only -- the case it wins (580 vs 604 us at B = 8, 12 heads) --,
// 2 also plain attention without a key bias on whole 64-key tiles (+ at most the one lone key), an even number >= 4 of them: measured equal to
// wg_attn_kernel there (CLIP 55.2 vs 54.5 us, 4096 keys 663 vs 640), kept for the tests and for other chips.  WG_ATTN_PIPE in the environment
// sets the initial mode; wg_attn_pipe_mode() changes it (returns the previous one; a negative argument only queries).
static int g_pipe_mode = -1;
extern "C" int wg_attn_pipe_mode(int mode) {
    if (g_pipe_mode < 0) {
        const char* e = getenv("WG_ATTN_PIPE");
        g_pipe_mode = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
    }
    const int prev = g_pipe_mode;
    if (mode >= 0 && mode <= 2) g_pipe_mode = mode;
    return prev;
}
bool wg_attn_pipe_takes(const AttnArgs& a, int head_dim, int S, int nw) {
    const int mode = wg_attn_pipe_mode(-1);
    if (mode == 0 || head_dim != 64 || (nw != 8 && nw != 4)) return false;
    if (S == 64) return a.Hg == 64 && a.nW == 1;
    if (mode < 2 || S != 0 || a.key_bias) return false;
    const int nt = a.Lk / 64;
    return (a.Lk % 64) <= 1 && nt >= 4 && (nt & 1) == 0;
}
int wg_attn_pipe_launch(const AttnArgs& a, int S, int nw, hipStream_t st) {
    if (S == 64) return nw == 8 ? launch_pipe<64, 8>(a, st) : launch_pipe<64, 4>(a, st);
    return nw == 8 ? launch_pipe<0, 8>(a, st) : launch_pipe<0, 4>(a, st);
}
