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
