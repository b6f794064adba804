// Shared by attn.hip and attn_pipe.hip: argument block, LDS swizzles of the K / V tiles, the asm form of the transposed LDS read.
#pragma once
#include "wg_common.h"

#define LOG2E 1.4426950408889634f
#define NEG_BIG (-1.0e30f)

struct AttnArgs {
    const bf16* Q; const bf16* K; const bf16* V; bf16* O;
    long ldq, ldk, ldv, ldo;     // row strides (elements)
    long q_bs, k_bs, o_bs;       // rows per batch item (plain mode)
    const bf16* padK; const bf16* padV;  // grid mode: bias rows used for zero-padded window positions
    const bf16* rel_h; const bf16* rel_w;  // [2S-1, HD]
    const float* key_bias;       // plain mode: [B, Lk] additive, may be null
    int B, heads, Lq, Lk;
    int Hg;                      // grid mode: tokens per image side
    int nW;                      // windows per side
    int qchunks;                 // workgroups per (batch, window, head)
    float scale;
    // fp8 chain (grid mode, head_dim 64): the output leaves as e4m3 bytes + one E8M0 scale per (row, 32 columns) -- the A operand of the
    // MX GEMM behind the attention (gemm.hip GemmArgs::mx_a layout: rows r, r + 16, .. of a 128-row group adjacent) -- instead of bf16
    unsigned char* Oq; long ldoq; unsigned char* Omx; long mx_pitch;
};

template <int HD> __device__ __forceinline__ int swzK(int row) {
    if (HD == 16) return (row >> 2) & 3;   // stored as 64-byte rows like head_dim 32
    if (HD == 80) return 0;   // 208-byte padded rows (SAM ViT-H): the row pitch itself spreads the banks
    if (HD == 128) return row & 15;
    if (HD == 64) return (row >> 1) & 7;
    return (row >> 2) & 3;  // HD == 32
}
template <int HD> __device__ __forceinline__ int swzV(int row) {
    if (HD == 80) return 0;
    if (HD == 128) return (row & 3) << 2;
    if (HD == 64) return ((row >> 1) & 1) << 2;
    return 0;
}

// raw v_exp_f32: exp2f() adds a denormal-range fix-up (compare, select, ldexp) around it that triples the cost of the
// one instruction softmax cannot avoid; arguments here are <= RESCALE_THR and results below 2^-126 may flush to zero.
#if defined(WG_ATTN_ABL) && WG_ATTN_ABL == 1
__device__ __forceinline__ float wg_exp2(float x) { return x * 0.001f; }   // ablation build: no exponentials
#else
__device__ __forceinline__ float wg_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif

// ds_read_b64_tr_b16 through inline asm.  The builtin form makes hipcc park an `s_waitcnt vmcnt(0)` in front of the
// first transposed read (it cannot tell the V tile being read from the LDS-DMA still writing the NEXT tile), which
// drains the prefetch in the middle of every tile.  The asm form is invisible to that pass; the matching wait is
// wg_tr_wait() below (cdna_hip_programming.md §5.7: own wait + sched_barrier before the consumers).
template <int OFF> __device__ __forceinline__ u32x2 wg_ds_read_tr(unsigned lds_addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(OFF));
    return v;
}


// attn_pipe.hip: the software-pipelined kernel for head_dim 64 (SAM global attention, CLIP without a key mask)
bool wg_attn_pipe_takes(const AttnArgs& a, int head_dim, int S, int nw);
int wg_attn_pipe_launch(const AttnArgs& a, int S, int nw, hipStream_t st);

// attn_window_unit.hip: SAM's 14 x 14 windows at head_dim 64 with a whole (window, head) of K / V staged ahead
bool wg_attn_window_unit_takes(const AttnArgs& a);
int wg_attn_window_unit_launch(AttnArgs a, int groups, hipStream_t st);

// The epilogue of a head_dim-64 attention kernel in its fp8 form: lane (query, hi) holds rows 8 g4 + 4 hi + e of the two 32-row blocks of O^T,
// i.e. 16 of the 32 columns of the MX block (query row, head, d); its partner lane ^ 32 holds the other 16.  The rule is wg_quantize_mx_fp8's on
// the bf16-ROUNDED output (what the quantisation pass used to read back): scale = the power of two at or above max|block| / 448.
template <int DB>
__device__ __forceinline__ void wg_attn_store_mx(const f32x16 (&ot)[DB], float inv_l, bool valid, long row, int hcol, int hi, const AttnArgs& a) {
    const long r = row % 128;
    const long spos = (row - r) + (r % 16) * 8 + r / 16;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        float v[16], am = 0x1p-100f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i] = (float)(bf16)(ot[d][i] * inv_l);
            am = fmaxf(am, fabsf(v[i]));
        }
        am = wg_xor32_max(am);
        const unsigned bits = __builtin_bit_cast(unsigned, am * (1.0f / 448.0f));
        unsigned e8 = (bits >> 23) + ((bits & 0x7FFFFFu) ? 1u : 0u);
        e8 = e8 > 253u ? 253u : e8;
        const float s = __builtin_bit_cast(float, (254u - e8) << 23);
        if (valid) {
            unsigned char* qp = a.Oq + row * a.ldoq + hcol + 32 * d + 4 * hi;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * g4] * s, v[4 * g4 + 1] * s, w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * g4 + 2] * s, v[4 * g4 + 3] * s, w, true);
                *(unsigned*)(qp + 8 * g4) = (unsigned)w;
            }
            if (hi == 0) a.Omx[(long)((hcol + 32 * d) >> 5) * a.mx_pitch + spos] = (unsigned char)e8;
        }
    }
}
