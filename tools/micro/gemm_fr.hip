// One-barrier persistent 256 x 256 bf16 GEMM for gfx950 (experimental, tile_hint 17):  C[M,N] = act(A[M,K] . W[N,K]^T + bias[N])
//
// Same operands, tile geometry (8 waves = 2 x 4, wave tile 128 x 64, 64-deep K slabs of 128-byte LDS rows brought in by LDS-DMA, two slab
// slots, source-side XOR swizzle) and arithmetic (v_mfma_f32_16x16x32_bf16, weights as the A operand, k in ascending order: the sums are
// bit-identical, tests/test_gpu_gemm_norm.py) as wg_gemm_pp_persist_kernel (gemm.hip), on the shapes wg_gemm_fr_supports() accepts
// (image_encoder.py:238,257, common.py:25-26 and HF CLIP's q|k|v / out_proj / fc1 / fc2 behind custom_clip.py:50-104; bias / activation
// epilogues only).  NOT dispatched by default: it is the round-5 answer to "can a different loop structure beat the four-barrier ping-pong",
// and the answer measured is no except for the GELU shape (profiles/r05_gemm_phases.md, notes/r05_experiments.md section 2).
//
// Structure ("priority ping-pong"): the two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD) are half a slab apart and meet at
// ONE workgroup barrier per slab.
//   leading group (s_setprio 1):  barrier | load half-phase: 24 fragment reads of slab s, 12 LDS-DMA pieces of slab s + 1 per wave | 64-MFMA burst
//   trailing group:               load half-phase: 24 fragment reads, wait for its pieces | barrier | its 4 pieces of slab s + 2 | 64-MFMA burst
// so each group's burst has the matrix pipe while the other group reads and issues, with no hand-over latency between them, and every
// instruction of the loop is its own `asm volatile` statement (hipcc keeps their order and allocates the registers, as in attn_pipe.hip).
// The readings it is built on (tools/micro/*.hip): two MFMA-dense waves on a SIMD do not interleave (the older wave gets every slot), a
// 64-MFMA burst is 1024 cycles, a DMA piece costs 7-13 cycles spaced and ~70 back to back, the fetch ceiling (42-44 B/clk/CU) is far
// above what the loop uses (26).  The epilogue is LDS-free: v_permlane16_swap pairs two column blocks into 64-byte row segments, so a group's
// epilogue runs beside its partner's burst.  Earlier forms (free-running waves with one hand-placed stream each; slab-level ping-pong
// with two barriers) are in the history of this file (commit 01dce1e and the one after) and in the notes.
// The slab stream runs across tile boundaries (the issue side is ahead of the consuming side and walks the tile list itself).
#include "wg_common.h"
#include "gemm_args.h"
#include <type_traits>

#define FR_RSRC_FLAGS 0x00020000   // raw buffer, 32-bit data format
#define FR_OOB 0xC0000000u         // a byte offset beyond every operand the kernel accepts (wg_gemm_fr_supports): the range check drops the access

template <int I, int N, class F> __device__ __forceinline__ void fr_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        fr_for<I + 1, N>(f);
    }
}

// WG_GEMM_FR_ABL (timing-only builds, wrong results): bit 0 = no LDS-DMA pieces, bit 1 = no fragment reads, bit 2 = no barrier, bit 3 = no waits
#ifndef WG_GEMM_FR_ABL
#define WG_GEMM_FR_ABL 0
#endif
// ---- one instruction per statement --------------------------------------------------------------------------------------------
#define FR_MFMA(acc, w, a) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a))
#define FR_LGKM(n) do { if (!(WG_GEMM_FR_ABL & 8)) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); } while (0)
#define FR_VMCNT(n) do { if (!(WG_GEMM_FR_ABL & 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); } while (0)
template <int OFF> __device__ __forceinline__ void fr_read(u32x4& v, unsigned lds_addr) {
    if constexpr (WG_GEMM_FR_ABL & 2) asm volatile("" : "+v"(v) : "v"(lds_addr));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
}
// One LDS-DMA piece: 64 lanes x 16 bytes from (descriptor base + per-lane byte offset) to LDS at lds_dst + lane * 16.  M0 is written in the
// statement that reads it (hipcc does not preserve it around asm); offsets at or beyond the descriptor's extent are dropped by the range check
// and still count in vmcnt, which keeps every wave's count of outstanding operations independent of the data.
__device__ __forceinline__ void fr_dma(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned lds_dst) {
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);      // wave-uniform by construction; this makes it provably so (an "s" operand)
    if constexpr (WG_GEMM_FR_ABL & 1) asm volatile("" ::"v"(voff), "s"(lds_dst));
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(lds_dst) : "memory", "m0");
}

// Diagnostic build only (-DWG_GEMM_STAMP, tools/gemm_fr_stamps.py): s_memtime sums in scalar registers at points where the LDS queue is empty anyway
// (behind the wave's last fragment read of a slab): wait for this wave's pieces of the next slab, wait at the barrier, the slab, the epilogue.
#ifdef WG_GEMM_STAMP
__device__ unsigned* wg_gemm_fr_stamp_ptr = nullptr;
extern "C" int wg_debug_gemm_fr_stamps(unsigned* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(wg_gemm_fr_stamp_ptr), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#define FR_STAMP(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define FR_STAMP(var) do { } while (0)
#endif

// Experiment knob (tools/build_variant.py): WG_GEMM_FR_SLEEP = n > 0 makes the older half of the workgroup (waves 0-3, which win every
// arbitration against their SIMD partners by age) sleep n x 64 cycles at the start of every pass.
#ifndef WG_GEMM_FR_SLEEP
#define WG_GEMM_FR_SLEEP 0
#endif
#define FR_YIELD() do { if (WG_GEMM_FR_SLEEP > 0 && polite) asm volatile("s_sleep %0" ::"n"(WG_GEMM_FR_SLEEP)); } while (0)

// LDS map (bytes): slab slot s at s * 65536: A rows 0..255 (128 B each), then W rows 0..255; then the bias rows of two tiles.
constexpr int FR_SLOT = 65536, FR_WOFF = 32768, FR_BIAS = 2 * FR_SLOT, FR_LDS = FR_BIAS + 2 * 512;
constexpr int FR_NSTORE = 16;      // output stores per wave and tile

__global__ __launch_bounds__(512, 2) void wg_gemm_fr_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int grp = wm;                            // 0: waves 0-3, the leading group (priority 1); 1: waves 4-7, one half-phase behind
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int nwg = g.tiles_m * g.tiles_n;
    const int nk = g.K >> 6;
    auto tile_of = [&](int v, int& m0, int& n0) {
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        int tm, tn;
        wg_tile_of(wgid, g.tiles_m, g.tiles_n, g.col_block, tm, tn);
        m0 = tm * 256;
        n0 = tn * 256;
    };

    // ---- issue side: the slab stream by LDS-DMA, one slab ahead.  A slab is 64 pieces of 8 rows (0-31: A rows 8 p .., 32-63: W rows 8 (p - 32) ..):
    //      wave w of group 0 sends pieces w + 4 k, k = 0..11 (48 of them), wave w of group 1 pieces 48 + w + 4 k, k = 0..3.  The row block of
    //      a wave's pieces has the parity of w, so the swizzle ((row >> 1) & 7 = (block * 4 + lane / 16) & 7) is one value per lane.
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (unsigned)(((long)(g.M - 1) * g.lda + g.K) * 2), FR_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (unsigned)(((long)(g.N - 1) * g.ldw + g.K) * 2), FR_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)g.bias, 0, g.bias ? (unsigned)(g.N * 2) : 0u, FR_RSRC_FLAGS);
    const int w4 = wave & 3;
    const unsigned cs = (unsigned)((lane & 7) ^ (((w4 & 1) * 4 + (lane >> 4)) & 7));
    const unsigned vbaseA = ((unsigned)(lane >> 3) * (unsigned)g.lda + cs * 8) * 2;
    const unsigned vbaseW = ((unsigned)(lane >> 3) * (unsigned)g.ldw + cs * 8) * 2;
    int iv = blockIdx.x, ikt = 0, im0, in0;       // tile and slab the issue side is at (both groups walk the same list)
    bool ivalid = iv < nwg;
    tile_of(iv, im0, in0);
    unsigned islot = 0;
    int ipar = 0;
    // pieces [P0, P0 + 4 * NP) step 4 of the next slab of the stream; BIAS: wave 0 also sends the tile's bias row in front of its first slab
    auto issue_slab = [&](auto p0c, auto npc, auto biasc) {
        constexpr int P0 = decltype(p0c)::value, NP = decltype(npc)::value;
        const unsigned dA = ivalid ? (unsigned)((im0 * (int)g.lda + ikt * 64) * 2) : FR_OOB;
        const unsigned dW = ivalid ? (unsigned)((in0 * (int)g.ldw + ikt * 64) * 2) : FR_OOB;
        const unsigned dst = lds0 + islot * FR_SLOT + (unsigned)(w4 * 1024);
        if constexpr (decltype(biasc)::value) {
            if (ivalid && ikt == 0 && g.bias && wave == 0 && lane < 32) fr_dma((unsigned)((in0 + lane * 8) * 2), brs, lds0 + FR_BIAS + (unsigned)ipar * 512);
        }
        if (ivalid && ikt == 0) ipar ^= 1;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = P0 + 4 * k;                 // (+ w4: folded into vbase row and dst)
            const int rb = p & 31;
            // (the piece's scalar offset is formed on the scalar unit and pinned there: added to the lane part first it is loop-invariant, and
            // hipcc would keep one VGPR per piece alive across the main loop)
            unsigned so = p < 32 ? dA + (unsigned)((rb + w4) * 8 * (int)g.lda * 2) : dW + (unsigned)((rb + w4) * 8 * (int)g.ldw * 2);
            so = __builtin_amdgcn_readfirstlane(so);      // (uniform by construction; hipcc may have formed it through VALU integer divisions)
            asm volatile("" : "+s"(so));
            if (p < 32) fr_dma(vbaseA + so, ars, dst + (unsigned)(rb * 1024));
            else fr_dma(vbaseW + so, wrs, dst + FR_WOFF + (unsigned)(rb * 1024));
        }
        islot ^= 1;
        if (++ikt == nk) {
            ikt = 0;
            iv += gridDim.x;
            ivalid = iv < nwg;
            if (ivalid) tile_of(iv, im0, in0);
        }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;

    // ---- consuming side ---------------------------------------------------------------------------------------------------------------
    // fragment addresses: A(i, ks) = slot + vA[ks] + i * 2048, W(j, ks) = slot + vW[ks] + j * 2048 (the slot is toggled by xor 65536)
    const unsigned swz = (unsigned)((fr >> 1) & 7);
    unsigned vA[2], vW[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        vA[ks] = lds0 + (unsigned)((wm * 128 + fr) * 128) + ((((unsigned)(ks * 4 + fq)) ^ swz) << 4);
        vW[ks] = lds0 + FR_WOFF + (unsigned)((wn * 64 + fr) * 128) + ((((unsigned)(ks * 4 + fq)) ^ swz) << 4);
    }
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, g.c_bytes, FR_RSRC_FLAGS);

    f32x4 acc[8][4];
    u32x4 a[2][8], w[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+v"(acc[i][j]));      // materialised here, not sunk in front of the first MFMA
            }
    };
    auto read_frags = [&]() {      // the slab's 24 fragments, then the other slot
        fr_for<0, 4>([&](auto jc) { constexpr int J = decltype(jc)::value; fr_read<J * 2048>(w[0][J], vW[0]); });
        fr_for<0, 8>([&](auto ic) { constexpr int I = decltype(ic)::value; fr_read<I * 2048>(a[0][I], vA[0]); });
        fr_for<0, 4>([&](auto jc) { constexpr int J = decltype(jc)::value; fr_read<J * 2048>(w[1][J], vW[1]); });
        fr_for<0, 8>([&](auto ic) { constexpr int I = decltype(ic)::value; fr_read<I * 2048>(a[1][I], vA[1]); });
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { vA[ks] ^= FR_SLOT; vW[ks] ^= FR_SLOT; }
    };
    auto burst = [&]() {           // the matrix half-phase: 64 MFMAs, nothing else in the stream
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) FR_MFMA(acc[i][j], w[ks][j], a[ks][i]);
    };
    // ---- epilogue of the tile at (m0, n0): bias, activation, bf16, 64-byte row segments straight from registers (v_permlane16_swap pairs
    //      two column blocks: lane (fr, fq) ends with 16 bytes = columns jp * 32 + (fq & 1) * 16 + (fq >> 1) * 8 .. + 7 of row i * 16 + fr)
    auto epilogue = [&](int m0, int n0, int par) {
        // lane coordinates re-derived here (two VALU instructions) instead of living in registers across the main loop, which runs close to the
        // 256-register limit: a spilled value's reload is a vector-memory operation, and hipcc drains the whole queue (vmcnt(0)) behind it
        int el;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
        const int fr = el & 15, fq = el >> 4;
        // the last MFMAs' results before the first vector read (opaque to the hazard recogniser): the wait states are TIED to the accumulators they
        // cover (an untied s_nop orders nothing against a copy or spill of acc the compiler might place in between)
        asm volatile("s_nop 15" : "+v"(acc[0][0]));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
        float bv[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 b = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            if (g.bias) b = *(const bf16x4*)(smem + FR_BIAS + par * 512 + (wn * 64 + j * 16 + fq * 4) * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[j][e] = (float)b[e];
        }
        const int ncol = n0 + wn * 64 + (fq & 1) * 16 + (fq >> 1) * 8;
        const int mrow = m0 + wm * 128 + fr;
        WG_ACT_SWITCH(g.act,
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {
                u32x2 pk[4];
                _Pragma("unroll") for (int j = 0; j < 4; ++j) pk[j] = __builtin_bit_cast(u32x2, wg_epi_pack<ACT>(acc[i][j], bv[j]));
                _Pragma("unroll") for (int jp = 0; jp < 2; ++jp) {
                    unsigned x0 = pk[2 * jp][0], x1 = pk[2 * jp][1], y0 = pk[2 * jp + 1][0], y1 = pk[2 * jp + 1][1];
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x0), "+v"(y0));
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x1), "+v"(y1));
                    const int n = ncol + jp * 32;
                    const int off = n < g.N ? ((mrow + i * 16) * (int)g.ldc + n) * 2 : (int)0x80000000;
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){x0, x1, y0, y1}, crs, off, 0, 0);
                }
            })
    };

    // prologue: the stream's first slab by everyone (group 0: 48 pieces, group 1: 16)
    if (grp == 0) issue_slab(C0{}, std::integral_constant<int, 12>{}, C1{});
    else {
        issue_slab(std::integral_constant<int, 48>{}, std::integral_constant<int, 4>{}, C0{});
        issue_slab(std::integral_constant<int, 48>{}, std::integral_constant<int, 4>{}, C0{});      // (the trailing group stays one slab further ahead: see its loop)
    }
    FR_VMCNT(0);
#ifdef WG_GEMM_STAMP
    unsigned long long st_0 = 0, st_1 = 0, st_2 = 0, st_3 = 0;
    unsigned long long st_m = 0, st_mb = 0, st_c = 0, st_cb = 0, st_epi = 0, st_lat = 0;
    unsigned st_slabs = 0;
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif

    int v = blockIdx.x, m0, n0;
    tile_of(v, m0, n0);
    int par = 0;
    int pm0 = 0, pn0 = 0;
    bool pending = false;        // the previous tile's epilogue has not run yet (it runs in the load half-phase of this tile's first slab)
    if (grp == 0) {
        // ---- leading group: [barrier] load half-phase (12 pieces of the next slab, this slab's fragments) | matrix half-phase | own pieces landed ----
        asm volatile("s_setprio 1");      // this group's instructions win every arbitration against the SIMD partner's
        zero_acc();
        asm volatile("s_barrier" ::: "memory");      // slab 0 has landed
        while (true) {
            for (int kt = 0; kt < nk; ++kt) {
                FR_STAMP(st_1);
                if (kt == 0 && pending) {
                    epilogue(pm0, pn0, par ^ 1);
                    zero_acc();
                }
                read_frags();
                issue_slab(C0{}, std::integral_constant<int, 12>{}, C1{});
                FR_LGKM(0);
                FR_STAMP(st_2);
                burst();
                FR_VMCNT(0);      // this wave's pieces of the next slab (sent in this slab's load half-phase) have landed
                FR_STAMP(st_3);
                // the next slab has landed and nobody reads the other slot any more (straight behind the burst: no taken branch between the last
                // MFMA and the barrier that hands the matrix pipe over)
                asm volatile("s_barrier" ::: "memory");
#ifdef WG_GEMM_STAMP
                FR_STAMP(st_0);
                st_mb += st_0 - st_3; st_m += st_2 - st_1; st_c += st_3 - st_2; ++st_slabs;
#endif
            }
            pm0 = m0; pn0 = n0; pending = true;
            par ^= 1;
            const int vn = v + gridDim.x;
            if (vn >= nwg) break;
            v = vn;
            tile_of(v, m0, n0);
        }
        epilogue(pm0, pn0, par ^ 1);
    } else {
        // ---- trailing group: load half-phase (4 pieces first, then the fragments) | [barrier] | matrix half-phase, which gets the matrix pipe whenever
        //      the leading group is in ITS load half-phase (and is simply held up by the leading group's MFMAs otherwise: no hand-over latency) ----
        asm volatile("s_barrier" ::: "memory");      // slab 0 has landed
        zero_acc();
        while (true) {
            for (int kt = 0; kt < nk; ++kt) {
                FR_STAMP(st_0);
                if (kt == 0 && pending) {
                    epilogue(pm0, pn0, par ^ 1);
                    zero_acc();
                    read_frags();
                    FR_LGKM(0);
                    FR_VMCNT(FR_NSTORE);      // the four pieces are older than the epilogue's stores
                } else {
                    read_frags();
                    FR_LGKM(0);
                    FR_VMCNT(0);
                }
                FR_STAMP(st_1);
                asm volatile("s_barrier" ::: "memory");      // the next slab has landed; the leading group starts its load half-phase
                FR_STAMP(st_2);
                // this group's four pieces of the slab after the next one go out in front of the burst: the slot they land in has just been freed,
                // and they have a whole slab's time to land (sent from the load half-phase they would have half of that, behind the leading group's 48)
                issue_slab(std::integral_constant<int, 48>{}, std::integral_constant<int, 4>{}, C0{});
                burst();
                FR_STAMP(st_3);
#ifdef WG_GEMM_STAMP
                st_m += st_1 - st_0; st_mb += st_2 - st_1; st_c += st_3 - st_2; ++st_slabs;
#endif
            }
            pm0 = m0; pn0 = n0; pending = true;
            par ^= 1;
            const int vn = v + gridDim.x;
            if (vn >= nwg) break;
            v = vn;
            tile_of(v, m0, n0);
        }
        epilogue(pm0, pn0, par ^ 1);
    }
    FR_VMCNT(0);
    FR_LGKM(0);
#ifdef WG_GEMM_STAMP
    if (wg_gemm_fr_stamp_ptr && lane == 0 && blockIdx.x < 32) {      // [workgroup][wave][8]
        unsigned* o = wg_gemm_fr_stamp_ptr + (blockIdx.x * 8 + wave) * 8;
        o[0] = (unsigned)st_m; o[1] = (unsigned)st_mb; o[2] = (unsigned)st_c; o[3] = (unsigned)st_cb; o[4] = (unsigned)st_lat; o[5] = st_slabs;
        o[6] = (unsigned)(__builtin_amdgcn_s_memtime() - st_c0); o[7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
    }
#endif
}

int wg_gemm_fr_supports(const GemmArgs& g) {
    const long abytes = ((long)(g.M - 1) * g.lda + g.K) * 2, wbytes = ((long)(g.N - 1) * g.ldw + g.K) * 2;
    return (g.K % 64 == 0 && g.K >= 128 && g.N % 8 == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0 && g.ldc % 8 == 0 && !g.out_f32 && g.c_bytes != 0 &&
            abytes < (1L << 31) && wbytes < (1L << 31) && !g.R && !g.ln_stats && !g.ln_part && !g.stats_part && !g.mx_w &&
            (!g.bias || ((uintptr_t)g.bias % 16 == 0))) ? 1 : 0;
}

int wg_launch_gemm_fr(GemmArgs& g, hipStream_t st) {
    g.tiles_m = (g.M + 255) / 256;
    g.tiles_n = (g.N + 255) / 256;
    {
        const long panel = 256L * g.K * 2, wbytes = (long)g.N * g.K * 2;
        const int cb = (int)((3L << 19) / (panel > 0 ? panel : 1));
        g.col_block = (wbytes > (3L << 20) && cb >= 2 && cb < g.tiles_n) ? cb : 0;
    }
    static WgPerDevice once;
    int dev = 0;
    if (once.first(&dev)) (void)hipFuncSetAttribute((const void*)wg_gemm_fr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FR_LDS);
    const int nwg = g.tiles_m * g.tiles_n;
    const int cus = wg_cu_count(dev);
    const int grid = nwg < cus ? nwg : cus;
    hipLaunchKernelGGL(wg_gemm_fr_kernel, dim3(grid), dim3(512), FR_LDS, st, g);
    return wg_check_launch("wg_gemm_bias_act_bf16(free-running persistent)");
}

// lets tools and tests tell a -DWG_GEMM_FR build (tile 17 = this kernel) from the product library (tile 17 falls back to tile 16)
extern "C" int wg_gemm_fr_present() { return 1; }
