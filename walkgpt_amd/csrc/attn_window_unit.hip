// Windowed attention for head_dim 64 (SAM ViT-B / ViT-L windows), whole-unit staging: wg_attn_window_kernel (attn.hip) with its K / V pipeline
// turned from "one 64-key tile ahead" into "one (window, head) ahead".
//
// Why (round 4, ablation builds of wg_attn_window_kernel<64, 14, 4>, B = 8 x 12 heads, alone on the chip: 93.7 us): with all tile arithmetic
// removed -- staging, waits, barriers and stores only -- that kernel still took 63 us, 2.5x the 25 us its 202 MB need at HBM speed: one 16-KiB
// tile in flight per workgroup behind a wait + barrier per tile is a latency chain.  Here a workgroup of NW = 7 waves (224 query slots for 196
// queries: one unit per (window, head), K / V staged once instead of once per query chunk) owns the CU, keeps TWO units' tiles in LDS
// (2 x 4 x 16 KiB) and requests the whole next unit when the current one starts: one barrier per unit, no wait inside it.
// Measured (same ablations on this kernel): skeleton 36 us (5.6 TB/s: the memory floor of 128-byte row segments), rel-pos passes +11,
// tiles +40 -> 88 us.  The requests now hide completely (draining them before the tiles or leaving them in flight: 88.6 / 87.7 us); what is
// left is the waves' own instruction streams -- ~1.9 us per wave and tile, seven waves on four SIMDs -- i.e. the same issue-bound picture as
// the global kernel before its hand-placed loop (notes/r04_experiments.md).  In the step: +0.4 % images/s.
// Arithmetic, layouts, rel-pos tables and the dead-key-block skip are those of wg_attn_window_kernel, line for line.
#include "attn_common.h"
#include <type_traits>

#define WG_RSRC_FLAGS 0x00020000   // raw buffer, 32-bit data format

template <int HD, int S, int NW>
__global__ __launch_bounds__(NW * 64) void wg_attn_window_unit_kernel(AttnArgs a, int total_units) {
    static_assert(S > 0 && S <= 32, "window sides up to 32 (padded rows of 16 / 32 slots)");
    constexpr int HDP = (HD == 80) ? 96 : (HD == 16 ? 32 : HD);
    constexpr int ROWB = (HD == 80) ? 208 : HDP * 2;
    constexpr int ROWBV = (HD == 80) ? 192 : ROWB;
    constexpr int TILE = 64 * ROWB, TILEV = 64 * ROWBV, TILE2 = TILE + TILEV;
    constexpr int KSTEPS = HD / 16;
    constexpr int DB = HDP / 32;
    constexpr int SS = S * S;
    constexpr int RP = S <= 16 ? 16 : 32;
    constexpr int RPT = 64 / RP;
    constexpr int NTG = (S + RPT - 1) / RPT;
    constexpr int SP = NTG * RPT + 1;
    constexpr int NRW = RP / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int UNIT = NTG * TILE2;                                 // all K / V tiles of one (window, head)
    char* kv = smem;                                                  // [2 units][NTG tiles][K tile | V tile]
    float* tab = (float*)(smem + 2 * UNIT);                           // per-wave rel-pos table in key space
    constexpr int RELROWS = 2 * S - 1;
    bf16* rels = (bf16*)(tab + NW * 32 * SP);                         // rel_w | rel_h table rows [2][RELROWS][HD], once per workgroup
    char* qslab = (char*)(rels + 2 * RELROWS * HD);                   // (head_dim 80 only: the lane-private width-term rows live behind the tables)
    // head_dim 80 is out of registers (215 before the unit loop's state): the width term of the bias (C operand of a tile's first S^T MFMAs,
    // 16 registers) lives in a lane-private LDS row there and is read per tile
    constexpr bool RELW_LDS = (HD == 80);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql_lane = lane & 31, hi = lane >> 5;
    float* relw_lds = (float*)qslab + (wave * 64 + lane) * 20;   // 80-byte pitch: conflict-free b128 reads

    // ---- a unit: (batch, window, head, query chunk) ---------------------------------------------------------------------------------
    struct Unit { int b, wy, wx, hcol, qh, qw, klim0, qvalid; long qrow; };
    const int groups = a.B * a.nW * a.nW * a.heads;
    auto decode = [&](int bid, Unit& u) __attribute__((always_inline)) {
        int grp, qc;
        if (a.qchunks > 1 && (groups & 7) == 0) {       // XCD-aware: all query chunks of a (window, head) on one XCD's L2
            const int per = 8 * a.qchunks;
            const int blk = bid / per, rem = bid % per;
            grp = blk * 8 + (rem & 7);
            qc = rem >> 3;
        } else {
            grp = bid / a.qchunks;
            qc = bid % a.qchunks;
        }
        const int head = grp % a.heads, bw = grp / a.heads, nw2 = a.nW * a.nW;
        u.b = bw / nw2;
        const int wi = bw % nw2;
        u.wy = wi / a.nW;
        u.wx = wi % a.nW;
        u.hcol = head * HD;
        const int ql_raw = (qc * NW + wave) * 32 + ql_lane;
        const int ql = ql_raw < SS ? ql_raw : SS - 1;
        u.qh = ql / S;
        u.qw = ql % S;
        const int gy = u.wy * S + u.qh, gx = u.wx * S + u.qw;
        const bool inside = gy < a.Hg && gx < a.Hg;
        u.qvalid = (ql_raw < SS && inside) ? 1 : 0;
        u.qrow = (long)u.b * a.Hg * a.Hg + (inside ? gy * a.Hg + gx : 0);
        u.klim0 = S < a.Hg - u.wy * S ? S : a.Hg - u.wy * S;     // window rows that exist in the image
    };
    auto load_q = [&](const Unit& u, bf16x8 (&q)[KSTEPS]) __attribute__((always_inline)) {
        const bf16* qp = a.Q + u.qrow * a.ldq + u.hcol + 8 * hi;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) q[s] = *(const bf16x8*)(qp + 16 * s);
    };
    // rel-pos table rows: the same for every unit -> LDS, once per workgroup (held in registers they cost 32 - 40 VGPRs across the unit loop)
    constexpr int NJB = (RELROWS + 31) / 32;
    for (int i = tid; i < 2 * RELROWS * HD / 8; i += NW * 64) {
        const int which = i / (RELROWS * HD / 8), r = i % (RELROWS * HD / 8);
        *(bf16x8*)(rels + (long)i * 8) = *(const bf16x8*)((which == 0 ? a.rel_w : a.rel_h) + (long)r * 8);
    }

    // ---- K / V staging: running per-lane source pointers, re-seeded per unit (see wg_attn_kernel) -------------------------------------
    constexpr int NINSTK = TILE / 1024, NINSTV = TILEV / 1024;
    constexpr int NPW = (NINSTK + NW - 1) / NW;
    static_assert(NPW <= 3, "running-pointer staging");
    const bf16* runp[2][NPW];
    unsigned coff[2][NPW];
    int kslot[2][NPW];
    bool padx[2][NPW];
    auto init_run = [&](const Unit& u) __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int cpr = (o ? ROWBV : ROWB) / 16, ninst = o ? NINSTV : NINSTK;
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                const int ii = wave + i * NW < ninst ? wave + i * NW : ninst - 1;
                const int ci = ii * 64 + lane;
                const int row = ci / cpr, cs = ci % cpr;
                int c = cs ^ (o ? swzV<HD>(row) : swzK<HD>(row));
                if (HDP != HD && c * 8 >= HD) c = HD / 8 - 1;
                coff[o][i] = c * 8;
                int kw = row % RP;
                kw = kw < S ? kw : S - 1;
                kslot[o][i] = row / RP;
                padx[o][i] = u.wx * S + kw >= a.Hg;
                const long r = (long)u.b * a.Hg * a.Hg + (long)(u.wy * S + kslot[o][i]) * a.Hg + u.wx * S + kw;
                if (o == 0) runp[0][i] = padx[0][i] ? a.padK + u.hcol + coff[0][i] : a.K + r * a.ldk + u.hcol + coff[0][i];
                else runp[1][i] = padx[1][i] ? a.padV + u.hcol + coff[1][i] : a.V + r * a.ldv + u.hcol + coff[1][i];
            }
        }
    };
    const unsigned strideK = (unsigned)((long)RPT * a.Hg * a.ldk);
    const unsigned strideV = (unsigned)((long)RPT * a.Hg * a.ldv);
    auto stage = [&](int t, int ubuf, bool isV, const Unit& u) __attribute__((always_inline)) {
        char* dst = kv + ubuf * UNIT + t * TILE2 + (isV ? TILE : 0);
        const int lim = u.klim0 - t * RPT;
        const int o = isV ? 1 : 0;
        const int ninst = isV ? NINSTV : NINSTK;
        const bf16* alt = (isV ? a.padV : a.padK) + u.hcol;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int ii = wave + i * NW;
            if (ii < ninst) {
                const bf16* src = kslot[o][i] < lim ? runp[o][i] : alt + coff[o][i];
                __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(src), WG_LDS_PTR(dst + ii * 1024), 16, 0, 0);
            }
            runp[o][i] += padx[o][i] ? 0u : (isV ? strideV : strideK);
        }
    };

    // ---- per-lane constants of the loop -------------------------------------------------------------------------------------------------
    float* mytab = tab + wave * 32 * SP;
    const float* relh_tab = mytab + ql_lane * SP;
    const float sc2 = a.scale * LOG2E;
    const float inv_sc2 = 1.0f / sc2;
    constexpr float RESCALE_THR = 6.0f;
    constexpr int NQK = 2 * KSTEPS;
    constexpr int NPV = 4 * DB;
    constexpr bool HALF_LAST = (S % RPT) != 0 && (S % RPT) * RP <= 32;   // the last tile holds keys in its first key block only
    unsigned vt_ad[DB];
    {
        const int g = lane >> 4, i16 = lane & 15;
        const int rq = i16 >> 2, cp = i16 & 3;
        const unsigned vbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem) + (unsigned)TILE;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const int col = 32 * d + 16 * (g & 1) + 4 * cp;
            const int chunk = col >> 3;
            vt_ad[d] = vbase + (4 * hi + rq) * ROWBV + ((chunk ^ swzV<HD>(4 * hi + rq)) << 4) + (col & 7) * 2;
        }
    }

    // a whole unit's K / V (NTG tiles) is requested at once, one unit ahead: NTG x 16 KiB in flight per CU while the current unit is
    // multiplied, one barrier per unit
    auto stage_unit = [&](int ubuf, const Unit& u) __attribute__((always_inline)) {
        init_run(u);
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
            stage(t, ubuf, false, u);
            stage(t, ubuf, true, u);
        }
    };
    static_assert(HDP == HD, "head_dim 64 / 32 / 128: every (d, g4) piece of the output exists");
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.O, 0, (int)((long)a.B * a.Hg * a.Hg * a.ldo * 2), WG_RSRC_FLAGS);
    Unit cur;
    decode(blockIdx.x, cur);
    bf16x8 qf[KSTEPS], qn[KSTEPS];
    load_q(cur, qn);
    stage_unit(0, cur);
    int ub = 0;

    for (int bid = blockIdx.x;;) {
        const int nbid = bid + (int)gridDim.x;
        const bool more = nbid < total_units;
        // this unit's tiles have landed (requested one unit ago) and every wave is done with the other buffer: refill it for the next unit
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (first unit: the rel-pos rows this wave parked in LDS)
        __builtin_amdgcn_s_barrier();                            // raw: __syncthreads() would drain the LDS-DMA just counted past
        // this unit's query fragments (loaded one unit ago) are taken HERE, in front of the next unit's requests: left to their first use
        // in the rel-pos pass, hipcc's own wait for them landed behind those requests as `vmcnt(0)` and drained the whole next unit
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            qf[s] = qn[s];
            asm volatile("" : "+v"(qf[s]));
        }
        if (more) {
            Unit nxt;
            decode(nbid, nxt);
            stage_unit(ub ^ 1, nxt);
            load_q(nxt, qn);
        }

        // ---- rel-pos tables of this unit's queries: T^T = Rel . Q^T by MFMA, scattered to key space (wave-private LDS) --------------------
        f32x16 relw_c;
        (void)relw_c;
        auto rel_pass = [&](int which) __attribute__((always_inline)) {
            const int qpos = which == 0 ? cur.qw : cur.qh;
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
                    const bf16x8 rf = *(const bf16x8*)(rels + ((long)which * RELROWS + j) * HD + 16 * s + 8 * hi);
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
#if !(defined(WG_ATTN_WIN_ABL) && (WG_ATTN_WIN_ABL & 2))
        rel_pass(0);
#endif
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = e % NRW;
            const int kw = (j & 3) + 8 * (j >> 2) + 4 * hi;
            const float wv = (kw < S ? mytab[ql_lane * SP + kw] : NEG_BIG) * inv_sc2;
            if constexpr (RELW_LDS) relw_lds[e] = wv;
            else relw_c[e] = wv;
        }
        asm volatile("" ::: "memory");
#if !(defined(WG_ATTN_WIN_ABL) && (WG_ATTN_WIN_ABL & 2))
        rel_pass(1);
#endif
        __builtin_amdgcn_wave_barrier();       // (the tables are wave-private)

        f32x16 ot[DB];
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
        float m_run = NEG_BIG;
        float l_run = 0.f;
        f32x16 sa[2];
        u32x2 vt[4][DB][2];
        bf16x8 pf[4];
        bf16x8 kfr[NQK];
        float rh[RPT];

        auto tile = [&](int t, auto half_c) __attribute__((always_inline)) {
            constexpr bool HALF = decltype(half_c)::value;
            constexpr int NE = HALF ? 16 : 32;
            const int buf = t;                 // (tile index inside the unit buffer)
#if defined(WG_ATTN_WIN_ABL) && (WG_ATTN_WIN_ABL & 4)
            if (t >= 0) return;
#endif
#pragma unroll
            for (int i = 0; i < RPT; ++i) rh[i] = relh_tab[t * RPT + i];
            const char* kbuf = kv + ub * UNIT + buf * TILE2;
#pragma unroll
            for (int g = 0; g < NQK; ++g) {
                if (HALF && (g & 1)) continue;
                const int kb = g & 1, s = g >> 1;
                const int row = kb * 32 + ql_lane;
                const int c = (2 * s + hi) ^ swzK<HD>(row);
                kfr[g] = *(const bf16x8*)(kbuf + row * ROWB + c * 16);
            }
#pragma unroll
            for (int g = 0; g < NQK; ++g) {
                if (HALF && (g & 1)) continue;
                const int kb = g & 1, s = g >> 1;
                if (s == 0) {                 // (+ width term of the bias: C operand)
                    if constexpr (RELW_LDS) {
                        f32x16 wc;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const f32x4 w4 = *(const f32x4*)(relw_lds + 4 * q4);
                            wc[4 * q4] = w4[0]; wc[4 * q4 + 1] = w4[1]; wc[4 * q4 + 2] = w4[2]; wc[4 * q4 + 3] = w4[3];
                        }
                        sa[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g], qf[s], wc, 0, 0, 0);
                    } else {
                        sa[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g], qf[s], relw_c, 0, 0, 0);
                    }
                }
                else sa[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g], qf[s], sa[kb], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) asm volatile("" : "+v"(rh[i]));
            // V^T fragments (inline-asm transposed reads, see wg_attn_kernel), behind the S^T MFMAs
#pragma unroll
            for (int g = 0; g < NPV; ++g) {
                const int ks = g / DB, d = g % DB;
                if (HALF && ks >= 2) continue;
                const unsigned ad = vt_ad[d] + (unsigned)(ub * UNIT + buf * TILE2);
                switch (ks) {
                    case 0: vt[0][d][0] = wg_ds_read_tr<0 * 16 * ROWBV>(ad); vt[0][d][1] = wg_ds_read_tr<0 * 16 * ROWBV + 8 * ROWBV>(ad); break;
                    case 1: vt[1][d][0] = wg_ds_read_tr<1 * 16 * ROWBV>(ad); vt[1][d][1] = wg_ds_read_tr<1 * 16 * ROWBV + 8 * ROWBV>(ad); break;
                    case 2: vt[2][d][0] = wg_ds_read_tr<2 * 16 * ROWBV>(ad); vt[2][d][1] = wg_ds_read_tr<2 * 16 * ROWBV + 8 * ROWBV>(ad); break;
                    default: vt[3][d][0] = wg_ds_read_tr<3 * 16 * ROWBV>(ad); vt[3][d][1] = wg_ds_read_tr<3 * 16 * ROWBV + 8 * ROWBV>(ad); break;
                }
            }
            // height term + running max
            float mt = NEG_BIG;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int kb = e >> 4, r = e & 15;
                const int sl0 = 32 * kb + (r & 3) + 8 * (r >> 2);
                const float v = sa[kb][r] * sc2 + rh[sl0 / RP];
                sa[kb][r] = v;
                mt = fmaxf(mt, v);
            }
            mt = wg_xor32_max(mt);
            if (__any(mt > m_run + RESCALE_THR)) {
                const float m_new = fmaxf(m_run, mt);
                const float alpha = wg_exp2(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
            }
            const float off = m_run;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int kb = e >> 4, r = e & 15;
                const float p = wg_exp2(sa[kb][r] - off);
                l_run += p;
                pf[kb * 2 + (r >> 3)][r & 7] = (bf16)p;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NPV; ++g) {
                const int ks = g / DB, d = g % DB;
                if (HALF && ks >= 2) continue;
                u32x4 vv = {vt[ks][d][0][0], vt[ks][d][0][1], vt[ks][d][1][0], vt[ks][d][1][1]};
                ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[ks], ot[d], 0, 0, 0);
            }
        };
        if constexpr (HALF_LAST) {
            for (int t = 0; t + 1 < NTG; ++t) tile(t, std::false_type());
            tile(NTG - 1, std::true_type());
        } else {
            for (int t = 0; t < NTG; ++t) tile(t, std::false_type());
        }

        // ---- O = O^T / l, 8-byte stores ------------------------------------------------------------------------------------------------
        // ---- O = O^T / l, 8-byte stores.  (Buffer stores: query slots outside the window or the image leave through an out-of-range offset, no
        // exec-mask branch around the stores.)  Tried and not kept: holding the packed output in registers and storing it behind the NEXT
        // unit's requests, so that the wait at the head of the loop finds no young stores to drain -- 97 us per launch against 88.
        const float l_tot = wg_xor32_sum(l_run);
        if (a.Oq) wg_attn_store_mx<DB>(ot, 1.0f / l_tot, cur.qvalid != 0, cur.qrow, cur.hcol, hi, a);      // fp8 chain (uniform branch)
        else {
            const float inv = 1.0f / l_tot;
            const unsigned ooff = cur.qvalid != 0 ? (unsigned)((cur.qrow * a.ldo + cur.hcol) * 2) : 0x80000000u;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)(ot[d][g4 * 4 + e] * inv);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ors, (int)(ooff + (unsigned)(32 * d + 8 * g4 + 4 * hi) * 2u), 0, 0);
                }
            }
        }
        if (!more) break;
        decode(nbid, cur);
        ub ^= 1;
        bid = nbid;
    }
}


// SAM windows of 14 x 14 at head_dim 64: 7 waves (224 query slots) own a (window, head); one persistent workgroup per CU.
// (the output leaves through 32-bit buffer offsets)
bool wg_attn_window_unit_takes(const AttnArgs& a) { return (long)a.B * a.Hg * a.Hg * a.ldo * 2 < (1L << 31); }

int wg_attn_window_unit_launch(AttnArgs a, int groups, hipStream_t st) {
    constexpr int HD = 64, S = 14, NW = 7;
    constexpr int TILE2 = 64 * 4 * HD, RPT = 4, NTG = (S + RPT - 1) / RPT, SP = NTG * RPT + 1;
    constexpr size_t lds = 2 * NTG * TILE2 + (size_t)NW * 32 * SP * 4 + 2 * (2 * S - 1) * HD * 2;
    static_assert(lds <= 160 * 1024, "two units of K / V tiles + the rel-pos tables in one CU's LDS");
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) (void)hipFuncSetAttribute((const void*)wg_attn_window_unit_kernel<HD, S, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    a.qchunks = 1;
    // one workgroup per CU, trimmed so that every workgroup walks the same number of units (+-1)
    const int cap = wg_cu_count(dev);
    int grid = groups;
    if (groups > cap) {
        const int rounds = (groups + cap - 1) / cap;
        grid = (groups + rounds - 1) / rounds;
        grid = (grid + 7) / 8 * 8;               // (a multiple of 8 keeps a workgroup's units on one XCD)
        if (grid > cap) grid = cap;
    }
    hipLaunchKernelGGL((wg_attn_window_unit_kernel<HD, S, NW>), dim3(grid), dim3(NW * 64), lds, st, a, groups);
    return wg_check_launch("wg_attn_window_unit");
}
