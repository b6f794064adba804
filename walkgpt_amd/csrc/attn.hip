// Fused multi-head attention for gfx950 (flash style: the score matrix never leaves registers).
//
// One kernel template, three users on the hot path:
//   S == 0  "plain"  : CLIP ViT-L/14 self-attention, 1025 keys + additive key-padding mask
//                      (custom_clip.py:27-38,50-104; HF CLIPAttention), MSQP cross-attention.
//   S == 14 "window" : SAM 14x14 windowed attention incl. decomposed rel-pos and the 64->70 zero padding whose
//                      pad tokens carry q=k=v=qkv-bias and take part in the softmax as keys
//                      (image_encoder.py:177-193, 235-260, 263-318, 321-392).
//   S == grid side   : SAM global attention (4096 keys at S=64) with decomposed rel-pos.
// window_partition / window_unpartition are pure address arithmetic here (no copies): keys and queries are fetched
// from / written to their natural [B*H*W, 3D] / [B*H*W, D] rows.
//
// Structure per workgroup: NW waves, each owning 32 query rows; K/V tiles of 64 keys are staged into LDS by
// LDS-DMA (double-buffered, one barrier per tile) and shared by all waves.
//   S^T = K . Q^T   (mfma 32x32x16, K fragment = A operand)  -> each lane holds 32 scores of ONE query column,
//                    so the running max / sum / rescale are per-lane scalars (no cross-lane traffic except one
//                    exchange with lane^32).
//   O^T += V^T . P^T (P^T is the S^T accumulator converted to bf16 in place: "accumulator as the next MFMA's
//                    B operand", k order permuted; the matching V^T fragment is read with ds_read_b64_tr_b16.)
//   rel-pos: T^T = Rel . Q^T by MFMA (table rows straight from global as the A operand), scattered into a
//            per-wave key-space table relh[q][kh], relw[q][kw] in LDS (pre-multiplied by log2 e); the main loop adds
//            relh[q][k / S] + relw[q][k % S].  For S == 64 a key tile is exactly one grid row: relw lives in 32
//            registers and relh costs one LDS read per tile.
// LDS swizzles (source-side, undone on the read): K rows for ds_read_b128, V rows for the transposed read.
#include "wg_common.h"
#include <type_traits>

#include "attn_common.h"

// Diagnostic build only (-DWG_ATTN_STAMP, tools/attn_stamps.py): lane 0 of every wave of workgroup 0 records s_memtime at the
// phase boundaries of its first 12 tiles into LDS and dumps them at the end.  No stamp executes in the normal build.
#ifdef WG_ATTN_STAMP
__device__ unsigned* wg_attn_stamp_ptr = nullptr;
extern "C" int wg_debug_attn_stamps(unsigned* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(wg_attn_stamp_ptr), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#define WG_STAMP(k)                                                                                                  \
    do {                                                                                                             \
        if (blockIdx.x == 0 && t < 12) {                                                                             \
            const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();                                             \
            if (lane == 0) ((volatile unsigned*)(smem + 156 * 1024))[(wave * 12 + t) * 8 + (k)] = now;              \
        }                                                                                                            \
    } while (0)
#else
#define WG_STAMP(k) do { } while (0)
#endif

template <int HD, int S, int NW, bool KB>
__global__ __launch_bounds__(NW * 64, 2) void wg_attn_kernel(AttnArgs a) {
    constexpr int HDP = (HD == 80) ? 96 : (HD == 16 ? 32 : HD);  // head dim padded to a multiple of 32 inside LDS
    // head_dim 80: rows padded once more to 13 chunks = 52 dwords, a pitch that sends 16 consecutive rows to 16 different bank
    // quads (the K fragment reads of 192-byte rows were 4-way conflicted: rows r and r + 4 shared their banks); the 13th chunk of
    // a row is never read
    constexpr int ROWB = (HD == 80) ? 208 : HDP * 2;   // bytes per K row
    // V rows of head_dim 80 keep the natural 192-byte pitch (12 chunks): four consecutive rows x 16 dwords then fall on disjoint bank
    // ranges for the transposed reads (pitch 48 dwords: 0, 48, 32, 16 mod 64), which the 208-byte pitch of the K rows (0, 52, 40, 28: rows
    // r and r + 1 overlap by four banks, a 2-way conflict) does not give; K keeps 208 for its ds_read_b128 pattern
    constexpr int ROWBV = (HD == 80) ? 192 : ROWB;     // bytes per V row
    constexpr int TILE = 64 * ROWB;        // bytes per K tile
    constexpr int TILEV = 64 * ROWBV;      // bytes per V tile
    constexpr int TILE2 = TILE + TILEV;    // one buffer = K tile | V tile
    constexpr int KSTEPS = HD / 16;        // QK^T runs over the real head dim only
    constexpr int DB = HDP / 32;           // PV d-blocks; columns >= HD hold don't-care data and are never stored
    constexpr bool GRID = (S > 0);
    constexpr int SS = GRID ? S * S : 0;
    // Grid mode key order: window rows padded to RP slots (16 / 32 / 64), RPT rows per 64-key tile.  A key slot's
    // column (kw) then depends only on the accumulator register, its row (kh) only on (tile, register): the width term
    // of the rel-pos bias is RP/2 registers per lane, the height term RPT LDS reads per tile, and padding slots are
    // masked by -inf entries in those two tables instead of per-element tests.
    constexpr int RP = S <= 16 ? 16 : (S <= 32 ? 32 : 64);
    constexpr int RPT = 64 / RP;
    constexpr int NTG = GRID ? (S + RPT - 1) / RPT : 0;   // key tiles per window
    constexpr int SP = GRID ? NTG * RPT + 1 : 1;          // relh table row (fp32 words, odd => conflict-free)
    constexpr int NRW = GRID ? RP / 2 : 1;                // width-bias registers per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kv = smem;                                  // [2 buffers][K tile | V tile]
    float* tab = (float*)(smem + 2 * TILE2);          // grid: per-wave rel table; plain: key bias row

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql_lane = lane & 31, hi = lane >> 5;

    // ---- decode the block id -> (batch, window, head, q chunk) ------------------------------------------------
    int bid = blockIdx.x;
    const int groups = a.B * (GRID ? a.nW * a.nW : 1) * a.heads;  // (batch, window, head) triples
    int grp, qc;
    if (a.qchunks > 1 && (groups & 7) == 0) {
        // XCD-aware: the 8 XCDs work on 8 different triples, all q chunks of a triple stay on one XCD's L2
        const int per = 8 * a.qchunks;
        const int blk = bid / per, rem = bid % per;
        grp = blk * 8 + (rem & 7);
        qc = rem >> 3;
    } else {
        grp = bid / a.qchunks;
        qc = bid % a.qchunks;
    }
    const int head = grp % a.heads;
    const int bw = grp / a.heads;
    int b, wy = 0, wx = 0;
    if (GRID) {
        const int nw2 = a.nW * a.nW;
        b = bw / nw2;
        const int wi = bw % nw2;
        wy = wi / a.nW;
        wx = wi % a.nW;
    } else {
        b = bw;
    }
    const int Lq = GRID ? SS : a.Lq;
    const int Lk = GRID ? NTG * 64 : a.Lk;   // grid mode: every slot of every tile is either a key or masked by the tables
    const int hcol = head * HD;

    // ---- this lane's query ---------------------------------------------------------------------------------------
    const int ql_raw = (qc * NW + wave) * 32 + ql_lane;
    const int ql = ql_raw < Lq ? ql_raw : Lq - 1;
    long qrow;
    bool qvalid = ql_raw < Lq;
    int qh = 0, qw = 0;
    if (GRID) {
        qh = ql / S;
        qw = ql % S;
        const int gy = wy * S + qh, gx = wx * S + qw;
        const bool inside = gy < a.Hg && gx < a.Hg;
        qvalid = qvalid && inside;
        qrow = (long)b * a.Hg * a.Hg + (inside ? gy * a.Hg + gx : 0);
    } else {
        qrow = (long)b * a.q_bs + ql;
    }
    bf16x8 qf[KSTEPS];
    {
        const bf16* qp = a.Q + qrow * a.ldq + hcol + 8 * hi;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) qf[s] = *(const bf16x8*)(qp + 16 * s);
    }

    // Window-sized tables (2S-1 <= 32 rows): the rel-pos rows of BOTH passes below are requested here, together with the query,
    // so that the prologue waits for global memory once instead of three times (query, width rows, height rows) -- with four key
    // tiles per window the prologue is as long as the loop.
    constexpr int NJB = GRID ? (2 * S - 1 + 31) / 32 : 1;
    constexpr bool REL_EARLY = GRID && NJB == 1;
    bf16x8 rel_early[REL_EARLY ? 2 : 1][KSTEPS];
    if constexpr (REL_EARLY) {
        const int j = ql_lane < 2 * S - 1 ? ql_lane : 2 * S - 2;
#pragma unroll
        for (int which = 0; which < 2; ++which)
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
                rel_early[which][s] = *(const bf16x8*)((which == 0 ? a.rel_w : a.rel_h) + (long)j * HD + 16 * s + 8 * hi);
    }

    // ---- K/V staging (LDS-DMA, swizzle on the source address) -----------------------------------------------
    // Piece i of this wave (1 KiB = 64 lanes x 16 bytes) covers fixed key slots of every tile, so its source is a running
    // per-lane pointer that advances by one tile of rows per call (softmax at head_dim 64 is VALU-bound: the general index
    // arithmetic -- window -> image coordinates, clamps, 64-bit row offsets -- was ~25 VALU instructions per piece per tile,
    // as much as the exponentials).  Slots that do not exist in a tile (keys beyond Lk; window rows beyond the window or the
    // image) read the substitute row instead: the last key (plain) or the qkv-bias row that stands for zero padding (grid);
    // slots beyond the window width do so in every tile and keep a stride of 0.  Their scores are masked by the bias tables.
    constexpr int NINSTK = TILE / 1024, NINSTV = TILEV / 1024;   // wave-instructions per K / V tile
    constexpr int NPW = (NINSTK + NW - 1) / NW;                    // ... per wave (NINSTV <= NINSTK)
    const bf16* altK = GRID ? a.padK + hcol : a.K + ((long)b * a.k_bs + (a.Lk - 1)) * a.ldk + hcol;
    const bf16* altV = GRID ? a.padV + hcol : a.V + ((long)b * a.k_bs + (a.Lk - 1)) * a.ldv + hcol;
    const int klim0 = GRID ? (S < a.Hg - wy * S ? S : a.Hg - wy * S) : a.Lk;   // rows (grid) / keys (plain) that exist
    constexpr bool RUNP = NPW <= 3;   // (one-wave workgroups issue every piece themselves: they keep the general form)
    constexpr int NRP = RUNP ? NPW : 1;
    const bf16* runp[2][NRP];
    unsigned coff[2][NRP];
    int kslot[2][NRP];
    bool padx[2][NRP];
    if constexpr (RUNP) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int cpr = (o ? ROWBV : ROWB) / 16, ninst = o ? NINSTV : NINSTK;
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                const int ii = wave + i * NW < ninst ? wave + i * NW : ninst - 1;
                const int ci = ii * 64 + lane;
                const int row = ci / cpr, cs = ci % cpr;        // key slot inside the tile, 16-byte chunk inside the row
                int c = cs ^ (o ? swzV<HD>(row) : swzK<HD>(row));
                if (HDP != HD && c * 8 >= HD) c = HD / 8 - 1;   // pad columns: any readable bytes will do
                coff[o][i] = c * 8;
                long r;
                if (GRID) {
                    int kw = row % RP;
                    kw = kw < S ? kw : S - 1;
                    kslot[o][i] = row / RP;
                    padx[o][i] = wx * S + kw >= a.Hg;
                    r = (long)b * a.Hg * a.Hg + (long)(wy * S + kslot[o][i]) * a.Hg + wx * S + kw;
                } else {
                    kslot[o][i] = row;
                    padx[o][i] = false;
                    r = (long)b * a.k_bs + row;
                }
                if (o == 0) runp[0][i] = padx[0][i] ? altK + coff[0][i] : a.K + r * a.ldk + hcol + coff[0][i];
                else runp[1][i] = padx[1][i] ? altV + coff[1][i] : a.V + r * a.ldv + hcol + coff[1][i];
            }
        }
    }
    const unsigned strideK = (unsigned)((GRID ? (long)RPT * a.Hg : 64L) * a.ldk);
    const unsigned strideV = (unsigned)((GRID ? (long)RPT * a.Hg : 64L) * a.ldv);
    // tile t of K (or V) -> buffer `buf`; per operand the tiles must be staged in order 0, 1, 2, ... (running pointers)
    auto stage = [&](int t, int buf, bool isV) __attribute__((always_inline)) {
        char* dst = kv + buf * TILE2 + (isV ? TILE : 0);
        const int lim = klim0 - t * (GRID ? RPT : 64);
        const int o = isV ? 1 : 0;
        const int cpr = (isV ? ROWBV : ROWB) / 16, ninst = isV ? NINSTV : NINSTK;
        if constexpr (RUNP) {
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                const int ii = wave + i * NW;
                if (ii < ninst) {
                    const bf16* src = kslot[o][i] < lim ? runp[o][i] : (isV ? altV : altK) + coff[o][i];
                    __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(src), WG_LDS_PTR(dst + ii * 1024), 16, 0, 0);
                }
                runp[o][i] += padx[o][i] ? 0u : (isV ? strideV : strideK);
            }
        } else {
            for (int ii = wave; ii < ninst; ii += NW) {
                const int ci = ii * 64 + lane;
                const int row = ci / cpr, cs = ci % cpr;
                int c = cs ^ (isV ? swzV<HD>(row) : swzK<HD>(row));
                if (HDP != HD && c * 8 >= HD) c = HD / 8 - 1;
                const bf16* src = (isV ? altV : altK) + c * 8;
                if (GRID) {
                    int kw = row % RP;
                    kw = kw < S ? kw : S - 1;
                    const int kh = t * RPT + row / RP;
                    if (row / RP < lim && wx * S + kw < a.Hg) {
                        const long r = (long)b * a.Hg * a.Hg + (long)(wy * S + kh) * a.Hg + wx * S + kw;
                        src = (isV ? a.V + r * a.ldv : a.K + r * a.ldk) + hcol + c * 8;
                    }
                } else if (row < lim) {
                    const long r = (long)b * a.k_bs + t * 64 + row;
                    src = (isV ? a.V + r * a.ldv : a.K + r * a.ldk) + hcol + c * 8;
                }
                __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(src), WG_LDS_PTR(dst + ii * 1024), 16, 0, 0);
            }
        }
    };

    // plain mode without a key bias, one key past a multiple of 64 (CLIP: 1024 patches + the class token): that key is folded in after the
    // loop on the vector ALU instead of costing a whole tile of MFMAs, exponentials and staging for one column
    const bool lone_key = !GRID && !KB && (Lk & 63) == 1 && Lk > 64;
    const int nt = GRID ? NTG : (lone_key ? Lk / 64 : (Lk + 63) / 64);
    stage(0, 0, false);
    stage(0, 0, true);

    // ---- rel-pos tables (grid) / key bias row (plain) -----------------------------------------------------------
    // Width term of the rel-pos bias, laid out as the S^T accumulator of a key block (element r of block kb belongs to padded-row
    // column slot (16*kb + r) % NRW) and pre-divided by scale*log2(e): it is the C operand of the first S^T MFMA of every tile,
    // so the matrix pipe adds it and the scores leave the MFMAs as (q.k + relw / sc2).
    constexpr int NC = (GRID && RP == 64) ? 2 : 1;
    f32x16 relw_c[NC];
    float* mytab = tab + wave * 32 * SP;
    // T^T = Rel . Q^T for one of the two tables, scattered to key space: mytab[q][kpos]
    auto rel_pass = [&](int which) __attribute__((always_inline)) {
        const bf16* rel = which == 0 ? a.rel_w : a.rel_h;
        const int qpos = which == 0 ? qw : qh;
        for (int k = S + hi; k < SP; k += 2) mytab[ql_lane * SP + k] = NEG_BIG;   // rows / columns beyond the window
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
            int j = jb * 32 + ql_lane;
            j = j < 2 * S - 1 ? j : 2 * S - 2;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                bf16x8 rf;
                if constexpr (REL_EARLY) rf = rel_early[which][s];
                else rf = *(const bf16x8*)(rel + (long)j * HD + 16 * s + 8 * hi);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf, qf[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;  // table row held in acc[r]
                const int kpos = qpos + S - 1 - jj;                        // key coordinate it belongs to
                if (jj < 2 * S - 1 && kpos >= 0 && kpos < S) mytab[ql_lane * SP + kpos] = acc[r] * LOG2E;
            }
        }
    };
    if constexpr (GRID) {
        // width table -> registers (through the LDS scatter); height table -> stays in LDS (same slot)
        const float inv_sc2 = 1.0f / (a.scale * LOG2E);
        // (mytab is private to the wave and a wave's LDS operations execute in order: the scatter, the reads below and the second scatter into
        // the same slot need no workgroup barrier)
        rel_pass(0);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = 0; e < 16 * NC; ++e) {
            const int j = e % NRW;
            const int kw = (j & 3) + 8 * (j >> 2) + 4 * hi;   // this lane's j-th column slot inside a padded row
            relw_c[e >> 4][e & 15] = (kw < S ? mytab[ql_lane * SP + kw] : NEG_BIG) * inv_sc2;
        }
        asm volatile("" ::: "memory");
        rel_pass(1);
    } else {
        if (KB) {
            for (int k = tid; k < nt * 64; k += NW * 64) {
                float v = k < Lk ? a.key_bias[(long)b * Lk + k] * LOG2E : NEG_BIG;
                tab[k] = fmaxf(v, NEG_BIG);
            }
        }
    }
    const float* relh_tab = mytab + ql_lane * SP;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- main loop ---------------------------------------------------------------------------------------------------
    f32x16 ot[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
    float m_run = NEG_BIG;
    float l_run = 0.f;   // running softmax denominator (this lane's half of the keys)
    const float sc2 = a.scale * LOG2E;

    // ---- main loop ---------------------------------------------------------------------------------------------------------------
    // Per tile: S^T MFMAs -> bias / running max -> exponentials -> P.V MFMAs, one barrier.  Measured alternatives (in-kernel
    // stamps, PMC, ablation builds): a ping-pong of the two workgroup halves across two barriers per tile, and
    // a software pipeline that interleaves S^T(t+1) with the exponentials of tile t and P.V(t) with the maxima of t+1 (+32 live
    // registers) were both SLOWER than this order once the per-tile index arithmetic of the staging was gone: the loop is a
    // latency chain in an in-order pipeline (matrix pipe 35 %, VALU 46 % busy, 18 % of the time together), not a throughput
    // problem of either unit.
    constexpr float RESCALE_THR = 6.0f;
    // RAW: no per-element bias is left to add on the VALU (none at all, or -- grid mode with one key row per tile -- the width term
    // comes out of the MFMAs and the height term is one value per tile): scores stay unscaled, p = exp2(s*sc2 - off)
    constexpr bool RAW = (!GRID && !KB) || (GRID && RPT == 1);
    constexpr int NQK = 2 * KSTEPS;              // S^T MFMAs per tile
    constexpr int NPV = 4 * DB;                  // P.V MFMAs per tile
    f32x16 sa[2];                                // scores of the tile
    u32x2 vt[4][DB][2];
    bf16x8 pf[4];  // P^T fragments: k-step (kb, s2) uses registers 8*s2 .. 8*s2+7 of block kb
    const bool ragged = !GRID && (Lk & 63) != 0 && !lone_key;  // plain mode: keys beyond Lk exist on the last tile only

    // S^T MFMA g of a tile: k-step s of key block kb (the two accumulation chains alternate); its K fragment is read from LDS
    // two MFMAs ahead
    bf16x8 kfr[NQK];
    auto qk_read = [&](const char* kbuf, int g) __attribute__((always_inline)) {
        const int kb = g & 1, s = g >> 1;
        const int row = kb * 32 + ql_lane;
        const int c = (2 * s + hi) ^ swzK<HD>(row);
        kfr[g] = *(const bf16x8*)(kbuf + row * ROWB + c * 16);
    };
    auto qk_one = [&](f32x16* st, int g) __attribute__((always_inline)) {
        const int kb = g & 1, s = g >> 1;
        if (s == 0) {
            if constexpr (GRID) {
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g], qf[s], relw_c[kb % NC], 0, 0, 0);
            } else {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g], qf[s], z, 0, 0, 0);
            }
        } else {
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g], qf[s], st[kb], 0, 0, 0);
        }
    };
    // bias + running max over score elements [e0, e1) of tile t (flattened index e = 16*kb + r)
    float rh[GRID ? RPT : 1];
    f32x4 kbv[2][4];
    auto bias_begin = [&](int t) __attribute__((always_inline)) {
        if constexpr (GRID) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) rh[i] = relh_tab[t * RPT + i];
        } else if constexpr (KB) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) kbv[kb][g4] = *(const f32x4*)(tab + t * 64 + kb * 32 + 8 * g4 + 4 * hi);
        }
    };
    auto bias_max = [&](f32x16* st, int t, int e0, int e1, float& mt, bool mask_tail) __attribute__((always_inline)) {
#pragma unroll
        for (int e = e0; e < e1; ++e) {
            const int kb = e >> 4, r = e & 15;
            float v = st[kb][r];
            if constexpr (GRID && RPT > 1) {
                const int sl0 = 32 * kb + (r & 3) + 8 * (r >> 2);        // slot in tile without the lane-half bit
                v = v * sc2 + rh[sl0 / RP];                                // (the width term is already in v: C operand of the MFMA)
            } else if constexpr (KB) {
                v = v * sc2 + kbv[kb][r >> 2][r & 3];
            }
            if (mask_tail && t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Lk) v = NEG_BIG;
            if (!RAW || mask_tail) st[kb][r] = v;
            mt = fmaxf(mt, v);
        }
    };
    // running max across the lane pair, lazy rescale of O; returns the exponent offset of the tile
    auto bias_end = [&](float mt) __attribute__((always_inline)) -> float {
        float rowh = 0.f;
        if constexpr (GRID && RPT == 1) rowh = rh[0];
        if constexpr (RAW) mt *= sc2;
        mt += rowh;
        mt = wg_xor32_max(mt);   // the other half of the keys of this query lives in lane ^ 32
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
        return m_run - rowh;
    };
    // V^T fragments of a tile: 4 k-steps x DB d-blocks x 2 transposed reads.  Lane (g = lane>>4, i16 = lane&15) supplies the
    // address of key row 16*ks + 4*hi + (i16>>2) (+8 for the second read), d columns 32*d + 16*(g&1) + 4*(i16&3) .. +3; the
    // swizzle bit(s) depend only on i16>>2, so each d block needs one lane-dependent base and compile-time offsets.
    unsigned vt_ad[DB];
    {
        const int g = lane >> 4, i16 = lane & 15;
        const int rq = i16 >> 2, cp = i16 & 3;
        const unsigned vbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem) + (unsigned)TILE;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const int col = 32 * d + 16 * (g & 1) + 4 * cp;
            const int chunk = col >> 3;
            // swzV depends on the key only through its low bits, which 16*ks and +8 leave untouched
            vt_ad[d] = vbase + (4 * hi + rq) * ROWBV + ((chunk ^ swzV<HD>(4 * hi + rq)) << 4) + (col & 7) * 2;
        }
    }
    // the two transposed reads that feed P.V MFMA g = ks*DB + d (the offset must be an immediate: switch on the unrolled ks)
    auto vt_pair = [&](int g, int buf) __attribute__((always_inline)) {
        const int ks = g / DB, d = g % DB;
        const unsigned ad = vt_ad[d] + (unsigned)(buf * TILE2);
        switch (ks) {
            case 0: vt[0][d][0] = wg_ds_read_tr<0 * 16 * ROWBV>(ad); vt[0][d][1] = wg_ds_read_tr<0 * 16 * ROWBV + 8 * ROWBV>(ad); break;
            case 1: vt[1][d][0] = wg_ds_read_tr<1 * 16 * ROWBV>(ad); vt[1][d][1] = wg_ds_read_tr<1 * 16 * ROWBV + 8 * ROWBV>(ad); break;
            case 2: vt[2][d][0] = wg_ds_read_tr<2 * 16 * ROWBV>(ad); vt[2][d][1] = wg_ds_read_tr<2 * 16 * ROWBV + 8 * ROWBV>(ad); break;
            default: vt[3][d][0] = wg_ds_read_tr<3 * 16 * ROWBV>(ad); vt[3][d][1] = wg_ds_read_tr<3 * 16 * ROWBV + 8 * ROWBV>(ad); break;
        }
    };
    // exponentials, row sum and bf16 packing of score elements [e0, e1)
    auto probs = [&](const f32x16* st, float off, int e0, int e1) __attribute__((always_inline)) {
#pragma unroll
        for (int e = e0; e < e1; ++e) {
            const int kb = e >> 4, r = e & 15;
            const float p = RAW ? wg_exp2(st[kb][r] * sc2 - off) : wg_exp2(st[kb][r] - off);
            l_run += p;
            pf[kb * 2 + (r >> 3)][r & 7] = (bf16)p;
        }
    };
    auto pv_one = [&](int g) __attribute__((always_inline)) {
        const int ks = g / DB, d = g % DB;
        u32x4 vv = {vt[ks][d][0][0], vt[ks][d][0][1], vt[ks][d][1][0], vt[ks][d][1][1]};
        const bf16x8 vf = __builtin_bit_cast(bf16x8, vv);
        ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], ot[d], 0, 0, 0);
    };

    // Windows whose last tile holds keys in its first key block only (S = 14 at four padded rows per tile: rows 12, 13 | 14, 15): the dead
    // block's S^T MFMAs, exponentials, V^T reads and P.V MFMAs are skipped there -- an eighth of a four-tile window's loop.
#ifndef WG_ATTN_STAMP   // (the diagnostic build keeps every wave to its closing barrier)
    if (!GRID && (qc * NW + wave) * 32 >= Lq) {
        // a wave without a single query (the last chunk of a ragged query count: CLIP's 1025 = 32 blocks + 1 leaves three such waves in
        // every ninth workgroup): it stages its share of the K / V tiles and keeps the barriers, nothing else -- its SIMD time goes to
        // the other workgroups of the CU
        for (int t = 0; t < nt; ++t) {
            if (t + 1 < nt) {
                stage(t + 1, (t & 1) ^ 1, false);
                stage(t + 1, (t & 1) ^ 1, true);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        return;
    }
#endif
    constexpr bool HALF_LAST = GRID && RPT > 1 && (S % RPT) != 0 && (S % RPT) * RP <= 32;
    auto tile = [&](int t, auto half_c) __attribute__((always_inline)) {
        constexpr bool HALF = decltype(half_c)::value;
        constexpr int NE = HALF ? 16 : 32;           // score elements per lane that can hold a key
        const int buf = t & 1;
        WG_STAMP(0);
        if (t + 1 < nt) {
            stage(t + 1, buf ^ 1, false);
            stage(t + 1, buf ^ 1, true);
        }
        bias_begin(t);
#pragma unroll
        for (int g = 0; g < NQK; ++g)
            if (!HALF || (g & 1) == 0) qk_read(kv + buf * TILE2, g);
#pragma unroll
        for (int g = 0; g < NQK; ++g)
            if (!HALF || (g & 1) == 0) qk_one(sa, g);
        // V^T fragments: inline-asm reads (invisible to hipcc's wait insertion), issued as early as the registers allow so that
        // their LDS latency hides under the softmax: right behind the S^T MFMAs, or -- key-bias variants, whose bias row
        // occupies 32 registers until it is applied -- after the running max.  Every LDS read hipcc does know about is retired
        // first: its wait would cover these too.
        constexpr bool EARLY_VT = !KB;
        if constexpr (GRID) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) asm volatile("" : "+v"(rh[i]));
        }
        if constexpr (EARLY_VT) {
#pragma unroll
            for (int g = 0; g < NPV; ++g)
                if (!HALF || g / DB < 2) vt_pair(g, buf);
        }
        float mt = NEG_BIG;
        if (ragged && t + 1 == nt) bias_max(sa, t, 0, NE, mt, true);
        else bias_max(sa, t, 0, NE, mt, false);
        const float off = bias_end(mt);
        if constexpr (!EARLY_VT) {
#pragma unroll
            for (int g = 0; g < NPV; ++g)
                if (!HALF || g / DB < 2) vt_pair(g, buf);
        }
        WG_STAMP(1);
        probs(sa, off, 0, NE);
        WG_STAMP(2);
        // O^T += V^T . P^T  (operands of the asm reads above: wait for them here, fenced from the MFMAs)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NPV; ++g)
            if (!HALF || g / DB < 2) pv_one(g);
        WG_STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        WG_STAMP(4);
    };
    if constexpr (HALF_LAST) {
        for (int t = 0; t + 1 < nt; ++t) tile(t, std::false_type());
        tile(nt - 1, std::true_type());
    } else {
        // (the same loop written out: instantiating it through the generic lambda above shifts hipcc's register allocation and one
        // head_dim-128 variant starts to spill)
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            WG_STAMP(0);
            if (t + 1 < nt) {
                stage(t + 1, buf ^ 1, false);
                stage(t + 1, buf ^ 1, true);
            }
            bias_begin(t);
    #pragma unroll
            for (int g = 0; g < NQK; ++g) qk_read(kv + buf * TILE2, g);
    #pragma unroll
            for (int g = 0; g < NQK; ++g) qk_one(sa, g);
            // V^T fragments: inline-asm reads (invisible to hipcc's wait insertion), issued as early as the registers allow so that
            // their LDS latency hides under the softmax: right behind the S^T MFMAs, or -- key-bias variants, whose bias row
            // occupies 32 registers until it is applied -- after the running max.  Every LDS read hipcc does know about is retired
            // first: its wait would cover these too.
            constexpr bool EARLY_VT = !KB;
            if constexpr (GRID) {
    #pragma unroll
                for (int i = 0; i < RPT; ++i) asm volatile("" : "+v"(rh[i]));
            }
            if constexpr (EARLY_VT) {
    #pragma unroll
                for (int g = 0; g < NPV; ++g) vt_pair(g, buf);
            }
            float mt = NEG_BIG;
            if (ragged && t + 1 == nt) bias_max(sa, t, 0, 32, mt, true);
            else bias_max(sa, t, 0, 32, mt, false);
            const float off = bias_end(mt);
            if constexpr (!EARLY_VT) {
    #pragma unroll
                for (int g = 0; g < NPV; ++g) vt_pair(g, buf);
            }
            WG_STAMP(1);
            probs(sa, off, 0, 32);
            WG_STAMP(2);
            // O^T += V^T . P^T  (operands of the asm reads above: wait for them here, fenced from the MFMAs)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
            for (int g = 0; g < NPV; ++g) pv_one(g);
            WG_STAMP(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            WG_STAMP(4);
        }
    }

#ifdef WG_ATTN_STAMP
    __syncthreads();
    if (blockIdx.x == 0 && wg_attn_stamp_ptr)
        for (int i = tid; i < NW * 12 * 8; i += NW * 64) wg_attn_stamp_ptr[i] = ((unsigned*)(smem + 156 * 1024))[i];
#endif
    if constexpr (!GRID && !KB) {
        if (lone_key) {
            const long krow = (long)b * a.k_bs + (Lk - 1);
            const bf16* kp = a.K + krow * a.ldk + hcol + 8 * hi;      // the lane's half of the head dims, as its query fragments
            float dot = 0.f;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const bf16x8 kf = *(const bf16x8*)(kp + 16 * s);
#pragma unroll
                for (int e = 0; e < 8; ++e) dot += (float)qf[s][e] * (float)kf[e];
            }
            const float sv = wg_xor32_sum(dot) * sc2;                  // the other half of the dims lives in lane ^ 32
            const float m_new = fmaxf(m_run, sv);
            const float alpha = wg_exp2(m_run - m_new);
            const float pl = wg_exp2(sv - m_new);
            m_run = m_new;
            l_run = l_run * alpha + (hi == 0 ? pl : 0.f);              // (the two lane halves' sums are added below)
            const bf16* vp = a.V + krow * a.ldv + hcol + 4 * hi;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if (HDP != HD && 32 * d + 8 * g4 + 4 * hi >= HD) continue;
                    const bf16x4 v4 = *(const bf16x4*)(vp + 32 * d + 8 * g4);   // O^T rows 32 d + 8 g4 + 4 hi .. +3 of this lane
#pragma unroll
                    for (int e = 0; e < 4; ++e) ot[d][g4 * 4 + e] = ot[d][g4 * 4 + e] * alpha + pl * (float)v4[e];
                }
        }
    }
    // ---- epilogue: O = O^T / l, 8-byte stores ---------------------------------------------------------------------------
    const float l_tot = wg_xor32_sum(l_run);
    if (qvalid) {
        const float inv = 1.0f / l_tot;
        long orow;
        if (GRID) orow = qrow;
        else orow = (long)b * a.o_bs + ql;
        bf16* op = a.O + orow * a.ldo + hcol;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                if (HDP != HD && 32 * d + 8 * g4 + 4 * hi >= HD) continue;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(ot[d][g4 * 4 + e] * inv);
                *(bf16x4*)(op + 32 * d + 8 * g4 + 4 * hi) = o;
            }
        }
    }
}

// =====================================================================================================================================
// Windowed attention (S <= 32: SAM's 14 x 14 windows), PERSISTENT form of wg_attn_kernel<HD, S, NW, false>.
//
// A window has 4 key tiles, and in-kernel stamps (tools/attn_window_stamps.py on the one-unit-per-workgroup kernel) put 40 % of a
// workgroup's life BEFORE its first tile: waiting for the query rows / the first K, V tile from memory and building the rel-pos tables,
// plus 7 % storing the output -- with one (ViT-H: 7 waves x 215 registers) or two workgroups per CU nothing else runs meanwhile.
// Here a workgroup walks units (window, head, query chunk) bid, bid + grid, ...: the NEXT unit's query rows are requested when the
// current unit's loop starts and its first K / V tile is staged (LDS-DMA) into the free buffer during the current unit's last tile, so
// the next unit begins with everything on chip; the rel-pos table rows (the same for every unit) are fetched once per workgroup.
// Arithmetic and data layout are those of wg_attn_kernel (see there); only the unit loop, the per-unit staging state and the
// dead-key-block skip of the last tile are spelled out again.
// =====================================================================================================================================
template <int HD, int S, int NW>
__global__ __launch_bounds__(NW * 64, 2) void wg_attn_window_kernel(AttnArgs a, int total_units) {
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
    static_assert(NTG % 2 == 0, "the next unit's first tile goes to buffer 0: an even number of tiles per unit");
    constexpr int SP = NTG * RPT + 1;
    constexpr int NRW = RP / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kv = smem;                                                  // [2 buffers][K tile | V tile]
    float* tab = (float*)(smem + 2 * TILE2);                          // per-wave rel-pos table in key space
    constexpr int RELROWS = 2 * S - 1;
    bf16* rels = (bf16*)(tab + NW * 32 * SP);                         // rel_w | rel_h table rows [2][RELROWS][HD], once per workgroup
    char* qslab = (char*)(rels + 2 * RELROWS * HD);                   // per wave: the NEXT unit's query fragments, KSTEPS x 1 KiB (LDS-DMA)
    // head_dim 80 is out of registers (215 before the unit loop's state): the width term of the bias (C operand of a tile's first S^T MFMAs,
    // 16 registers) lives in a lane-private LDS row there and is read per tile
    constexpr bool RELW_LDS = (HD == 80);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql_lane = lane & 31, hi = lane >> 5;
    float* relw_lds = (float*)(qslab + NW * KSTEPS * 1024) + (wave * 64 + lane) * 20;   // 80-byte pitch: conflict-free b128 reads

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
    char* myq = qslab + wave * KSTEPS * 1024;
    auto dma_q = [&](const Unit& u) __attribute__((always_inline)) {      // the same fragments, global -> LDS without passing through registers
        const bf16* qp = a.Q + u.qrow * a.ldq + u.hcol + 8 * hi;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) __builtin_amdgcn_global_load_lds(WG_GLOBAL_PTR(qp + 16 * s), WG_LDS_PTR(myq + s * 1024), 16, 0, 0);
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
    auto stage = [&](int t, int buf, bool isV, const Unit& u) __attribute__((always_inline)) {
        char* dst = kv + buf * TILE2 + (isV ? TILE : 0);
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

    Unit cur;
    decode(blockIdx.x, cur);
    bf16x8 qf[KSTEPS];
    load_q(cur, qf);
    init_run(cur);
    stage(0, 0, false, cur);
    stage(0, 0, true, cur);
    __syncthreads();                      // the rel-pos rows in LDS are visible to every wave

    for (int bid = blockIdx.x;;) {
        const int nbid = bid + (int)gridDim.x;
        const bool more = nbid < total_units;

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
        rel_pass(0);
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
        rel_pass(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

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
            const int buf = t & 1;
            if (t + 1 < NTG) {
                stage(t + 1, buf ^ 1, false, cur);
                stage(t + 1, buf ^ 1, true, cur);
                if (t == 0 && more) {             // the next unit's queries: requested behind tile 1's pieces, left in flight over this tile's wait
                    Unit nxt;
                    decode(nbid, nxt);
                    dma_q(nxt);
                }
            } else if (more) {               // last tile: the next unit's first tile into the free buffer (buffer 0: NTG is even)
                Unit nxt;
                decode(nbid, nxt);
                init_run(nxt);
                stage(0, buf ^ 1, false, nxt);
                stage(0, buf ^ 1, true, nxt);
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) rh[i] = relh_tab[t * RPT + i];
            const char* kbuf = kv + buf * TILE2;
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
                const unsigned ad = vt_ad[d] + (unsigned)(buf * TILE2);
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
            // memory operations retire in issue order: in tile 0 the KSTEPS query pieces are the youngest and may stay outstanding
            if (t == 0 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KSTEPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        if constexpr (HALF_LAST) {
            for (int t = 0; t + 1 < NTG; ++t) tile(t, std::false_type());
            tile(NTG - 1, std::true_type());
        } else {
            for (int t = 0; t < NTG; ++t) tile(t, std::false_type());
        }

        // ---- O = O^T / l, 8-byte stores ------------------------------------------------------------------------------------------------
        const float l_tot = wg_xor32_sum(l_run);
        if (cur.qvalid != 0) {
            const float inv = 1.0f / l_tot;
            bf16* op = a.O + cur.qrow * a.ldo + cur.hcol;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if (HDP != HD && 32 * d + 8 * g4 + 4 * hi >= HD) continue;
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)(ot[d][g4 * 4 + e] * inv);
                    *(bf16x4*)(op + 32 * d + 8 * g4 + 4 * hi) = o;
                }
            }
        }
        if (!more) break;
        decode(nbid, cur);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) qf[s] = *(const bf16x8*)(myq + s * 1024 + lane * 16);   // (landed before tile 1's barrier)
        bid = nbid;
    }
}

template <int HD, int S, int NW>
static int launch_attn_window(const AttnArgs& a, int groups, hipStream_t st) {
    constexpr int TILE = 64 * ((HD == 80) ? 208 + 192 : 4 * (HD == 16 ? 32 : HD));
    constexpr int RP = S <= 16 ? 16 : 32;
    constexpr int RPT = 64 / RP;
    constexpr int SP = ((S + RPT - 1) / RPT) * RPT + 1;
    const size_t lds = 2 * TILE + (size_t)NW * 32 * SP * 4 + 2 * (2 * S - 1) * HD * 2 + (size_t)NW * (HD / 16) * 1024 + (HD == 80 ? (size_t)NW * 64 * 80 : 0);
    static WgPerDevice once;
    static int per_cu_of[64] = {};
    int dev = 0;
    if (once.first(&dev) || per_cu_of[dev & 63] == 0) {
        (void)hipFuncSetAttribute((const void*)wg_attn_window_kernel<HD, S, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)wg_attn_window_kernel<HD, S, NW>, NW * 64, lds) != hipSuccess || nb < 1) nb = 1;
        per_cu_of[dev & 63] = nb;
    }
    const int per_cu = per_cu_of[dev & 63];
    // as many workgroups as the chip holds at once, trimmed so that every workgroup walks the same number of units (+-1) and the
    // XCD-aware unit order keeps its period
    const int total = groups * a.qchunks;
    const int period = 8 * a.qchunks;
    const int cap = wg_cu_count(dev) * per_cu;
    int grid = total;
    if (total > cap) {
        const int rounds = (total + cap - 1) / cap;
        grid = (total + rounds - 1) / rounds;
        grid = (grid + period - 1) / period * period;
        if (grid > cap) grid = cap / period * period;
    }
    hipLaunchKernelGGL((wg_attn_window_kernel<HD, S, NW>), dim3(grid), dim3(NW * 64), lds, st, a, total);
    return wg_check_launch("wg_attn(window)");
}

template <int HD, int S, int NW, bool KB>
static int launch_attn_impl(const AttnArgs& a, int groups, hipStream_t st) {
    constexpr int TILE = 64 * ((HD == 80) ? 208 + 192 : 4 * (HD == 16 ? 32 : HD));   // bytes of a K tile + a V tile (head_dim 80: 208- and 192-byte rows)
    constexpr int RP = S <= 16 ? 16 : (S <= 32 ? 32 : 64);
    constexpr int RPT = 64 / RP;
    constexpr int SP = S > 0 ? ((S + RPT - 1) / RPT) * RPT + 1 : 1;
    size_t lds = 2 * TILE;
    if (S > 0) lds += (size_t)NW * 32 * SP * 4;
    else lds += (size_t)((a.Lk + 63) / 64) * 64 * 4;
    if (lds > 160 * 1024) {
        wg_set_error("attention: LDS request %zu exceeds 160 KiB", lds);
        return WG_ERR_UNSUPPORTED;
    }
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) (void)hipFuncSetAttribute((const void*)wg_attn_kernel<HD, S, NW, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef WG_ATTN_STAMP
    lds = 160 * 1024;
#endif
    hipLaunchKernelGGL((wg_attn_kernel<HD, S, NW, KB>), dim3(groups * a.qchunks), dim3(NW * 64), lds, st, a);
    return wg_check_launch("wg_attn");
}

template <int HD, int S, int NW>
static int launch_attn(const AttnArgs& a, int groups, hipStream_t st) {
    if (S == 0 && a.key_bias) return launch_attn_impl<HD, S, NW, (S == 0)>(a, groups, st);
    return launch_attn_impl<HD, S, NW, false>(a, groups, st);
}

// Plain multi-head attention (optionally cross attention, optionally with an additive per-key bias).
// Q rows: (b*q_rows_per_batch + i), head h at columns [h*hd, (h+1)*hd); same for K, V, O.
extern "C" int wg_mha_bf16(const void* Q, long ldq, long q_rows_per_batch, const void* K, long ldk, const void* V, long ldv,
                           long k_rows_per_batch, void* O, long ldo, long o_rows_per_batch, const float* key_bias,
                           int B, int heads, int head_dim, int Lq, int Lk, float scale, void* stream) {
    WG_REQUIRE(Q && K && V && O, "mha: null operand");
    WG_REQUIRE(B > 0 && heads > 0 && Lq > 0 && Lk > 0, "mha: bad shape");
    WG_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "mha: leading dimensions must be multiples of 8");
    WG_REQUIRE((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) == 0 && ((uintptr_t)O & 7) == 0, "mha: misaligned operand");
    AttnArgs a{};
    a.Q = (const bf16*)Q; a.K = (const bf16*)K; a.V = (const bf16*)V; a.O = (bf16*)O;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.q_bs = q_rows_per_batch; a.k_bs = k_rows_per_batch; a.o_bs = o_rows_per_batch;
    a.key_bias = key_bias; a.B = B; a.heads = heads; a.Lq = Lq; a.Lk = Lk; a.scale = scale;
    hipStream_t st = (hipStream_t)stream;
    const int groups = B * heads;
    const int qblocks = (Lq + 31) / 32;
    // waves per workgroup: query-slot utilisation x a measured per-size factor (tools/bench_attn.py, 4096 and 1025 keys).  At head_dim
    // <= 64 four-wave workgroups (three or more of them per CU) beat one eight-wave workgroup by 5 % and three-wave ones by 13 %
    // even at 1025 queries, where they leave 11 % of the query slots empty (CLIP: 67 -> 58 us); at head_dim 128 eight waves win.
    int nw = 1;
    if (qblocks > 1) {
        const int cand[3] = {8, 4, 3};
        const double eff_small[3] = {0.95, 1.0, 0.87}, eff_128[3] = {1.0, 0.86, 0.0};
        double best = -1.0;
        for (int i = 0; i < 3; ++i) {
            const int c = cand[i];
            if (head_dim != 64 && c == 3) continue;
            const double util = (double)qblocks / (double)(((qblocks + c - 1) / c) * c);
            const double score = util * (head_dim == 128 ? eff_128[i] : eff_small[i]);
            if (score > best + 1e-9) { best = score; nw = c; }
        }
    }
    a.qchunks = (qblocks + nw - 1) / nw;
    if (wg_attn_pipe_takes(a, head_dim, 0, nw)) return wg_attn_pipe_launch(a, 0, nw, st);   // head_dim 64, no key bias, whole tiles: the pipelined loop
#define WG_MHA_CASE(HD_, NW_) if (head_dim == HD_ && nw == NW_) return launch_attn<HD_, 0, NW_>(a, groups, st);
    WG_MHA_CASE(64, 8) WG_MHA_CASE(64, 4) WG_MHA_CASE(64, 3) WG_MHA_CASE(64, 1)
    WG_MHA_CASE(128, 8) WG_MHA_CASE(128, 4) WG_MHA_CASE(128, 1)
    WG_MHA_CASE(32, 8) WG_MHA_CASE(32, 4) WG_MHA_CASE(32, 1)
    WG_MHA_CASE(16, 1)
#undef WG_MHA_CASE
    wg_set_error("mha: head_dim %d with %d query blocks not supported (32, 64, 128; 16 for <= 32 queries)", head_dim, qblocks);
    return WG_ERR_UNSUPPORTED;
}

// SAM ViT attention over a packed qkv buffer [B*Hg*Hg, 3*D] (q | k | v, heads contiguous inside each third).
// window == Hg: global attention; window < Hg: non-overlapping windows with zero padding to a multiple of
// `window`, pad positions acting as keys/values equal to qkv_bias (their outputs are dropped).
// WG_ATTN_WIN_UNIT=0: head_dim-64 windows on the tile-ahead kernel (A/B runs)
static bool wg_window_unit_on() {
    static const char* e = getenv("WG_ATTN_WIN_UNIT");
    return !(e && e[0] == '0');
}

static int wg_sam_attn_impl(const void* qkv, const void* qkv_bias, const void* rel_pos_h, const void* rel_pos_w, void* out, void* out_q, void* out_mx, long mx_pitch,
                            int B, int grid, int window, int heads, int head_dim, float scale, void* stream);

extern "C" int wg_sam_attn_relpos_bf16(const void* qkv, const void* qkv_bias, const void* rel_pos_h, const void* rel_pos_w,
                                       void* out, int B, int grid, int window, int heads, int head_dim, float scale,
                                       void* stream) {
    WG_REQUIRE(out, "sam_attn: null operand");
    return wg_sam_attn_impl(qkv, qkv_bias, rel_pos_h, rel_pos_w, out, nullptr, nullptr, 0, B, grid, window, heads, head_dim, scale, stream);
}

// The fp8 chain's form (config C5's nn.Linear chain, ViT-B / ViT-L geometry): the attention output leaves as the proj GEMM's MX operand -- e4m3 bytes
// out_q [B * grid^2, D] and E8M0 block scales out_mx [D / 32][mx_pitch] in wg_quantize_mx_fp8's group-128 layout, bit for bit what that pass makes of
// the bf16 output -- and the bf16 tensor is never written.  head_dim 64 only (a 32-column block must not straddle two heads: not ViT-H's 80), on the
// kernels that carry the epilogue (wg_sam_attn_mx_supported).
extern "C" int wg_sam_attn_mx_supported(int B, int grid, int window, int heads, int head_dim) {
    if (head_dim != 64 || B <= 0 || heads <= 0) return 0;
    AttnArgs a{};
    a.B = B; a.heads = heads; a.Hg = grid; a.nW = (grid + window - 1) / window; a.ldo = (long)heads * head_dim;
    if (window == 64) return wg_attn_pipe_takes(a, head_dim, 64, 8) ? 1 : 0;
    if (window == 14) return (wg_window_unit_on() && wg_attn_window_unit_takes(a)) ? 1 : 0;
    return 0;
}
extern "C" int wg_sam_attn_relpos_mx_bf16(const void* qkv, const void* qkv_bias, const void* rel_pos_h, const void* rel_pos_w, void* out_q, void* out_mx, long mx_pitch,
                                          int B, int grid, int window, int heads, int head_dim, float scale, void* stream) {
    WG_REQUIRE(out_q && out_mx && wg_sam_attn_mx_supported(B, grid, window, heads, head_dim), "sam_attn_mx: head_dim 64 with window 14 or a 64 x 64 global grid only");
    WG_REQUIRE(mx_pitch >= ((long)B * grid * grid + 127) / 128 * 128 && ((uintptr_t)out_q & 3) == 0, "sam_attn_mx: scale pitch must cover the rows in groups of 128");
    return wg_sam_attn_impl(qkv, qkv_bias, rel_pos_h, rel_pos_w, nullptr, out_q, out_mx, mx_pitch, B, grid, window, heads, head_dim, scale, stream);
}

static int wg_sam_attn_impl(const void* qkv, const void* qkv_bias, const void* rel_pos_h, const void* rel_pos_w, void* out, void* out_q, void* out_mx, long mx_pitch,
                            int B, int grid, int window, int heads, int head_dim, float scale, void* stream) {
    WG_REQUIRE(qkv && qkv_bias && rel_pos_h && rel_pos_w && (out || out_q), "sam_attn: null operand");
    WG_REQUIRE(B > 0 && grid > 0 && window > 0 && window <= grid && heads > 0, "sam_attn: bad shape");
    const long D = (long)heads * head_dim;
    WG_REQUIRE((((uintptr_t)qkv | (uintptr_t)qkv_bias | (uintptr_t)rel_pos_h | (uintptr_t)rel_pos_w) & 15) == 0 &&
                   ((uintptr_t)out & 7) == 0, "sam_attn: misaligned operand");
    AttnArgs a{};
    const bf16* base = (const bf16*)qkv;
    a.Q = base; a.K = base + D; a.V = base + 2 * D; a.O = (bf16*)out;
    a.ldq = a.ldk = a.ldv = 3 * D; a.ldo = D;
    a.Oq = (unsigned char*)out_q; a.ldoq = D; a.Omx = (unsigned char*)out_mx; a.mx_pitch = mx_pitch;
    a.padK = (const bf16*)qkv_bias + D; a.padV = (const bf16*)qkv_bias + 2 * D;
    a.rel_h = (const bf16*)rel_pos_h; a.rel_w = (const bf16*)rel_pos_w;
    a.B = B; a.heads = heads; a.Hg = grid; a.nW = (grid + window - 1) / window; a.scale = scale;
    hipStream_t st = (hipStream_t)stream;
    const int groups = B * a.nW * a.nW * heads;
    const int qblocks = (window * window + 31) / 32;
#define WG_SAM_CASE(HD_, S_, NW_)                                  \
    if (head_dim == HD_ && window == S_) {                         \
        a.qchunks = (qblocks + NW_ - 1) / NW_;                     \
        if (S_ == 64 && wg_attn_pipe_takes(a, head_dim, S_, NW_)) {                                              \
            /* (WG_ATTN_PIPE_NW=4: two four-wave workgroups per CU instead of one of eight waves -- measured equal, 577 vs 573 us) */ \
            static const char* nwe = getenv("WG_ATTN_PIPE_NW");                                                  \
            const int pnw = (nwe && nwe[0] == '4') ? 4 : 8;                                                      \
            a.qchunks = (qblocks + pnw - 1) / pnw;                                                               \
            return wg_attn_pipe_launch(a, S_, pnw, st);                                                          \
        }                                                                                                        \
        if (HD_ == 64 && S_ == 14 && wg_window_unit_on() && wg_attn_window_unit_takes(a)) return wg_attn_window_unit_launch(a, groups, st);      \
        if constexpr (S_ <= 32) return launch_attn_window<HD_, S_, NW_>(a, groups, st);   \
        else return launch_attn<HD_, S_, NW_>(a, groups, st);      \
    }
    WG_SAM_CASE(64, 14, 4)   // two 4-wave workgroups per window (one idle query slot in eight) measured 5 % faster than one of 7 waves
    WG_SAM_CASE(64, 64, 8)
    WG_SAM_CASE(64, 32, 8)
    WG_SAM_CASE(32, 14, 7)
    WG_SAM_CASE(32, 28, 5)
    WG_SAM_CASE(80, 14, 7)   // (two 4-wave workgroups per window: 166.8 vs 160.6 us at B = 8, 16 heads)
    WG_SAM_CASE(80, 64, 8)
#undef WG_SAM_CASE
    wg_set_error("sam_attn: (head_dim %d, window %d) has no compiled kernel", head_dim, window);
    return WG_ERR_UNSUPPORTED;
}
