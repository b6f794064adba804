// Software-pipelined fused attention for head_dim 64 on gfx950: SAM's global attention (64 x 64 grid, decomposed rel-pos;
// image_encoder.py:235-260, 321-392) and CLIP's 1025-key self-attention without a key mask (HF CLIPAttention; custom_clip.py:50-104).
//
// Same arithmetic, LDS images and fragment maps as wg_attn_kernel (attn.hip): S^T = K . Q^T on mfma 32x32x16 so that a lane owns
// 32 scores of ONE query, P^T stays in registers as the B operand of O^T += V^T . P^T, V^T by ds_read_b64_tr_b16, K / V tiles of
// 64 keys by LDS-DMA from running per-lane pointers, lazy rescale (2^6), rel-pos width term as the C operand of the first S^T
// MFMA, height term folded into the exponent offset.  What differs is the ORDER of the work inside a wave:
//
//   wg_attn_kernel runs a tile as one dependent chain -- S^T MFMAs, maximum, exponentials, P.V MFMAs, barrier -- and with two
//   waves per SIMD in lockstep a tile costs the SUM of its matrix segment (2 x 512 cycles) and its vector segment (2 x ~650)
//   plus the LDS / barrier latencies between them: ~3300 cycles for 1024 cycles of MFMA work per SIMD (stamps).
//   Here every wave runs a three-stage pipeline over the key tiles: in iteration t its matrix instructions are S^T of tile
//   t+1 (into the second score buffer) and P.V of tile t-1, its vector instructions the softmax of tile t -- three tiles with no
//   data dependence between them.  The instruction stream is hand-placed: every instruction of the loop body is its own `asm
//   volatile` statement (hipcc keeps volatile asm statements in program order and allocates their registers; it scheduled the
//   same pipeline written with builtins into clusters: rounds 1-3), one MFMA followed by ~7 vector instructions of the
//   exponential phase and the LDS fragment reads of the MFMAs three to four slots ahead, behind counted lgkmcnt waits.
//   MI355X_MICROARCH.md: an MFMA holds the SIMD's vector issue for 8 of its 32 cycles, a wave issues one vector instruction per 4
//   cycles -- the two waves of a SIMD fill each other's gaps, and the matrix pipe sees MFMAs from both.
//
// Hazards this file owns (the compiler sees opaque statements):
//   * LDS reads are asm: every consumer sits behind an `s_waitcnt lgkmcnt(N)` counted from the issue order written below (LDS
//     operations return in order; the loop body contains no scalar loads -- check the ISA after edits: an s_load inside the loop
//     would share the counter).
//   * MFMA results are read by vector instructions one iteration later (S^T) or after the loop (O^T); P^T fragments written by
//     v_cvt_pk are read by MFMAs one iteration later.  The rare rescale path multiplies O^T (written by the previous iteration's
//     last MFMAs, > 40 instructions and a barrier earlier) and the pending P^T of tile t-1, whose P.V MFMAs all come later in the
//     iteration: the decision never splits a pending tile (cdna_hip_programming.md T13).
//   * LDS-DMA: tile K(t+2) / V(t) are requested in iteration t into the buffers whose last reads ended before the barrier that
//     opens the iteration; `s_waitcnt vmcnt(0)` + s_barrier close it.
#include "attn_common.h"
#include <type_traits>

template <int I, int N, class F> __device__ __forceinline__ void wg_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wg_static_for<I + 1, N>(f);
    }
}

// ---- one instruction per statement ------------------------------------------------------------------------------------------
#define PA_MFMA_ACC(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define PA_MFMA_C(d, a, b, c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c))
#define PA_MFMA_Z(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b))
#define PA_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
#define PA_MAX3(d, a, b, c) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))
#define PA_FMA_S(d, a, s, c) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(s), "v"(c))
#define PA_EXP(d, a) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(a))
#define PA_ADD(acc, a) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(a))
#define PA_CVT(d, lo, hi) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi))

template <int OFF> __device__ __forceinline__ u32x4 pa_ds_read_b128(unsigned lds_addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ float pa_ds_read_b32(unsigned lds_addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}

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
    char* kv = smem;                               // [2 buffers][K tile | V tile]
    float* tab = (float*)(smem + 2 * TILE2);       // grid: per-wave rel table

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
    const int qh = GRID ? ql / S : 0, qw = GRID ? ql % S : 0;
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
    // tile j of K (o = 0) or V (o = 1) -> buffer j & 1; per operand the tiles are staged in order 0, 1, 2, ...  (every tile the loop
    // stages is a whole tile: the launcher guarantees 64 | number of keys walked)
    auto stage = [&](int j, int o) __attribute__((always_inline)) {
        char* dst = kv + (j & 1) * TILE2 + (o ? TILE : 0);
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
            relw_c[e >> 4][e & 15] = mytab[ql_lane * SP + kw] * inv_sc2;
        }
        asm volatile("" ::: "memory");
        rel_pass(1);
    }
    const float sc2 = a.scale * LOG2E;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
    unsigned relh_ad = lds0 + 2 * TILE2 + (unsigned)((wave * 32 + ql_lane) * SP * 4);    // this lane's relh row, entry t

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
            vt_ad[d] = lds0 + TILE + (4 * hi + rq) * ROWB + ((chunk ^ swzV<HD>(4 * hi + rq)) << 4) + (col & 7) * 2;
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (!GRID && (qc * NW + wave) * 32 >= Lq) {
        // a wave without a single query (CLIP's 1025 = 32 blocks + 1): it stages its share of the tiles and keeps the barriers
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nt; ++t) {
            if (t + 2 < nt) stage(t + 2, 0);
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
    float m_run = NEG_BIG, la = 0.f, lb = 0.f;
    constexpr float RESCALE_THR = 6.0f;

    // S^T MFMA g = (kb = g >> 2, s = g & 3) of the tile in buffer BUF into score buffer SB
    u32x4 kf[4];
    auto k_read = [&kf, &kad](auto gc, auto bufc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, BUF = decltype(bufc)::value;
        constexpr int kb = g >> 2, s = g & 3;
        kf[s] = pa_ds_read_b128<BUF * TILE2 + kb * 32 * ROWB>(kad[s]);
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
        vta[j & 3] = wg_ds_read_tr<BUF * TILE2 + ks * 16 * ROWB>(vt_ad[d]);
        vtb[j & 3] = wg_ds_read_tr<BUF * TILE2 + ks * 16 * ROWB + 8 * ROWB>(vt_ad[d]);
    };
    auto pv_mfma = [&ot, &pf, &vta, &vtb](auto jc, auto pbc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, PB = decltype(pbc)::value;
        constexpr int ks = j >> 1, d = j & 1;
        const u32x4 vv = {vta[j & 3][0], vta[j & 3][1], vtb[j & 3][0], vtb[j & 3][1]};
        PA_MFMA_ACC(ot[d], __builtin_bit_cast(bf16x8, vv), __builtin_bit_cast(bf16x8, pf[PB][ks]));
    };

    // ---- pre-loop: S^T of tile 0 ------------------------------------------------------------------------------------------------------
    {
        using B0 = std::integral_constant<int, 0>;
        wg_static_for<0, 4>([&k_read](auto g) { k_read(g, B0{}); });
        wg_static_for<0, 8>([&k_read, &qk_mfma](auto gc) {
            constexpr int g = decltype(gc)::value;
            PA_LGKM(g < 4 ? 3 : 7 - g);
            qk_mfma(gc, B0{});
            if constexpr (g < 4) k_read(std::integral_constant<int, g + 4>{}, B0{});
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- one iteration: softmax of tile t (score buffer P) beside S^T of tile t+1 (into buffer P^1; K tile in LDS buffer P^1) and
    //      P.V of tile t-1 (P^T buffer P^1; V tile in LDS buffer P^1).  P = t & 1.
    auto iter = [&](int t, auto pc, auto do_qk_c, auto do_pv_c) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value, Q = P ^ 1;
        constexpr bool DO_QK = decltype(do_qk_c)::value, DO_PV = decltype(do_pv_c)::value;
        using BQ = std::integral_constant<int, Q>;
        // tiles for the iterations to come
        if (t + 2 < nt) stage(t + 2, 0);
        stage(t, 1);
        // LDS issue order (the counted waits below follow it): rh | K g0..g3 | K g4..g7 (one behind each of MFMA 0..3) | V pairs 0..3
        // (behind MFMA 4..7) | V pairs 4..7 (behind P.V MFMA 0..3)
        float rh = 0.f;
        if constexpr (GRID) rh = pa_ds_read_b32(relh_ad);
        if constexpr (DO_QK) wg_static_for<0, 4>([&](auto g) { k_read(g, BQ{}); });
        else if constexpr (DO_PV) wg_static_for<0, 4>([&](auto j) { v_read(j, BQ{}); });
        // running maximum of the tile: two chains over the 32 scores of this lane
        float ma, mb;
        PA_MAX3(ma, sa[P][0][0], sa[P][0][1], sa[P][0][2]);
        PA_MAX3(mb, sa[P][0][3], sa[P][0][4], sa[P][0][5]);
        wg_static_for<0, 13>([&ma, &mb, &sa](auto ic) {
            constexpr int i = decltype(ic)::value, e = 6 + 2 * i;
            if constexpr ((i & 1) == 0) PA_MAX3(ma, ma, sa[P][e >> 4][e & 15], sa[P][(e + 1) >> 4][(e + 1) & 15]);
            else PA_MAX3(mb, mb, sa[P][e >> 4][e & 15], sa[P][(e + 1) >> 4][(e + 1) & 15]);
        });
        float mt;
        PA_MAX3(mt, ma, mb, mb);
        if constexpr (GRID) {
            PA_LGKM(DO_QK || DO_PV ? (DO_QK ? 4 : 8) : 0);
            relh_ad += 4;
        }
        PA_FMA_S(mt, mt, sc2, rh);
        mt = wg_xor32_max(mt);      // the other half of the keys of this query lives in lane ^ 32
        if (__any(mt > m_run + RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            la *= alpha;
            lb *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
            if constexpr (DO_PV) {      // the pending P^T of tile t-1 (none of its P.V MFMAs has been issued yet)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned pr = pf[Q][ks][w];
                        const float lo = __builtin_bit_cast(float, pr << 16) * alpha, hi_ = __builtin_bit_cast(float, pr & 0xFFFF0000u) * alpha;
                        unsigned o;
                        PA_CVT(o, lo, hi_);
                        pf[Q][ks][w] = o;
                    }
            }
        }
        const float noff = rh - m_run;      // p = exp2(s * sc2 + noff)
        // exponential phase, skewed over the MFMA gaps: group k = fma of elements 2k, 2k+1 | exp of 2k-2, 2k-1 | sum + pack of 2k-4, 2k-3
        float x[32], pe[32];
        auto group = [&x, &pe, &sa, &pf, &la, &lb, &noff, sc2](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < 16) {
                constexpr int e = 2 * k;
                PA_FMA_S(x[e], sa[P][e >> 4][e & 15], sc2, noff);
                PA_FMA_S(x[e + 1], sa[P][(e + 1) >> 4][(e + 1) & 15], sc2, noff);
            }
            if constexpr (k >= 1 && k < 17) {
                constexpr int e = 2 * k - 2;
                PA_EXP(pe[e], x[e]);
                PA_EXP(pe[e + 1], x[e + 1]);
            }
            if constexpr (k >= 2) {
                constexpr int e = 2 * k - 4;       // element e = 16*kb + r -> fragment kb*2 + (r >> 3), dword (r & 7) >> 1
                PA_ADD(la, pe[e]);
                PA_ADD(lb, pe[e + 1]);
                unsigned o;
                PA_CVT(o, pe[e], pe[e + 1]);
                pf[P][(e >> 4) * 2 + ((e & 15) >> 3)][(e & 7) >> 1] = o;
            }
        };
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
        wg_static_for<0, 16>([&group, &k_read, &v_read, &qk_mfma, &pv_mfma](auto mc) {
            constexpr int m = decltype(mc)::value;
            if constexpr (m < 8) {
                if constexpr (DO_QK) {
                    // K g landed?  issued behind it: K g+1..g+3 (m < 4: + nothing else yet), then the V pairs
                    PA_LGKM(m < 5 ? 3 : (DO_PV ? 2 * (m - 4) + (7 - m) : 7 - m));
                    qk_mfma(mc, BQ{});
                    if constexpr (m < 4) k_read(std::integral_constant<int, m + 4>{}, BQ{});
                    else if constexpr (DO_PV) v_read(std::integral_constant<int, m - 4>{}, BQ{});
                }
            } else {
                constexpr int j = m - 8;
                if constexpr (DO_PV) {
                    PA_LGKM(j < 5 ? 6 : 2 * (7 - j));
                    pv_mfma(std::integral_constant<int, j>{}, BQ{});
                    if constexpr (j < 4) v_read(std::integral_constant<int, j + 4>{}, BQ{});
                }
            }
            group(std::integral_constant<int, m + 2>{});
        });
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    iter(0, I0{}, T_{}, F_{});
    for (int t = 1; t + 1 < nt; t += 2) {
        iter(t, I1{}, T_{}, T_{});
        iter(t + 1, I0{}, T_{}, T_{});
    }
    iter(nt - 1, I1{}, F_{}, T_{});
    // P.V of the last tile (odd: P^T buffer 1, V tile in LDS buffer 1)
    {
        wg_static_for<0, 4>([&](auto j) { v_read(j, I1{}); });
        wg_static_for<0, 8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            PA_LGKM(j < 5 ? 6 : 2 * (7 - j));
            pv_mfma(jc, I1{});
            if constexpr (j < 4) v_read(std::integral_constant<int, j + 4>{}, I1{});
        });
    }
    float l_run = la + lb;

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
            l_run = l_run * alpha + (hi == 0 ? pl : 0.f);
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
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // the last P.V MFMAs' results (opaque to the hazard recogniser) before the first read
    const float l_tot = wg_xor32_sum(l_run);
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
    size_t lds = 2 * 2 * 64 * 128;
    if (S > 0) lds += (size_t)NW * 32 * (S + 1) * 4;
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) (void)hipFuncSetAttribute((const void*)wg_attn_pipe_kernel<S, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((wg_attn_pipe_kernel<S, NW>), dim3(a.B * a.heads * a.qchunks), dim3(NW * 64), lds, st, a);
    return wg_check_launch("wg_attn(pipelined)");
}

// Does the pipelined kernel take this case?  head_dim 64; plain: no key bias, whole 64-key tiles (+ at most the one lone key), an even
// number >= 4 of them; grid: the 64 x 64 global attention over a 64 x 64 token grid (no padding, every tile one key row).
bool wg_attn_pipe_takes(const AttnArgs& a, int head_dim, int S, int nw) {
    static const char* off = getenv("WG_ATTN_PIPE");
    if (off && off[0] == '0') return false;
    if (head_dim != 64 || (nw != 8 && nw != 4)) return false;
    if (S == 64) return a.Hg == 64 && a.nW == 1 && nw == 8;
    if (S != 0 || a.key_bias) return false;
    const int nt = a.Lk / 64;
    return (a.Lk % 64) <= 1 && nt >= 4 && (nt & 1) == 0;
}
int wg_attn_pipe_launch(const AttnArgs& a, int S, int nw, hipStream_t st) {
    if (S == 64) return launch_pipe<64, 8>(a, st);
    return nw == 8 ? launch_pipe<0, 8>(a, st) : launch_pipe<0, 4>(a, st);
}
