// Attention for the shapes where a 32-row MFMA query tile would be mostly padding: the two-way mask decoder
// (transformer.py:220-242: 6..20 tokens x 4096 image keys at 16 dims/head, 4096 image queries x 6..20 token keys,
// token self-attention at 32 dims/head) and MSQP's learned-query cross attention (utils_walkgpt.py:163-185:
// 4..12 queries x up to 4096 keys, 8 heads x 128).
//
// One wave per (batch, head, query).  Lanes are arranged [key slot][d slice]: a lane owns DPL consecutive head
// dims of one key per iteration, keeps a private online-softmax state (m, l, o[DPL]) over the keys it visits, and
// the wave merges the per-slot states at the end.  K and V rows are read once, 16 bytes per load.
// HBM/L2-bound: algorithmic bytes = Lk * 2 * hd * 2 per (batch, head, query) -- the queries of one (batch, head)
// run on neighbouring waves and re-read the same K/V rows from L2.
#include "wg_common.h"

struct SmallAttnArgs {
    const bf16* Q; const bf16* K; const bf16* V; bf16* O;
    long ldq, ldk, ldv, ldo;
    long q_bs, k_bs, o_bs;
    int B, heads, Lq, Lk;
    float scale;
};

template <int HD>
__global__ __launch_bounds__(256) void wg_attn_small_kernel(SmallAttnArgs a) {
    constexpr int DPL = HD < 32 ? HD : 32;   // head dims per lane
    constexpr int DL = HD / DPL;             // lanes per key
    constexpr int SLOTS = 64 / DL;           // keys per wave iteration
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)a.B * a.heads * a.Lq;
    if (wid >= total) return;
    const int qi = (int)(wid % a.Lq);
    const int h = (int)((wid / a.Lq) % a.heads);
    const int b = (int)(wid / ((long)a.Lq * a.heads));
    const int slot = lane / DL, ds = lane % DL;
    const int dcol = h * HD + ds * DPL;

    float q[DPL];
    {
        const bf16* qp = a.Q + ((long)b * a.q_bs + qi) * a.ldq + dcol;
#pragma unroll
        for (int c = 0; c < DPL / 8; ++c) {
            const bf16x8 t = *(const bf16x8*)(qp + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) q[8 * c + e] = (float)t[e] * a.scale * 1.4426950408889634f;
        }
    }
    float m = -1.0e30f, l = 0.f, o[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) o[d] = 0.f;

    for (int j0 = 0; j0 < a.Lk; j0 += SLOTS) {
        const int j = j0 + slot;
        const bool ok = j < a.Lk;
        const long row = (long)b * a.k_bs + (ok ? j : a.Lk - 1);
        const bf16* kp = a.K + row * a.ldk + dcol;
        const bf16* vp = a.V + row * a.ldv + dcol;
        float s = 0.f;
        bf16x8 vv[DPL / 8];
#pragma unroll
        for (int c = 0; c < DPL / 8; ++c) {
            const bf16x8 kk = *(const bf16x8*)(kp + 8 * c);
            vv[c] = *(const bf16x8*)(vp + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += q[8 * c + e] * (float)kk[e];
        }
#pragma unroll
        for (int w = 1; w < DL; w <<= 1) s += __shfl_xor(s, w, 64);
        if (!ok) s = -1.0e30f;
        const float mn = fmaxf(m, s);
        const float alpha = exp2f(m - mn);
        const float p = ok ? exp2f(s - mn) : 0.f;
        m = mn;
        l = l * alpha + p;
#pragma unroll
        for (int c = 0; c < DPL / 8; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[8 * c + e] = o[8 * c + e] * alpha + p * (float)vv[c][e];
    }
    // merge the key slots (lanes with equal d slice): butterfly over the slot bits
#pragma unroll
    for (int w = DL; w < 64; w <<= 1) {
        const float m2 = __shfl_xor(m, w, 64);
        const float l2 = __shfl_xor(l, w, 64);
        const float mn = fmaxf(m, m2);
        const float a1 = exp2f(m - mn), a2 = exp2f(m2 - mn);
        l = l * a1 + l2 * a2;
#pragma unroll
        for (int d = 0; d < DPL; ++d) {
            const float o2 = __shfl_xor(o[d], w, 64);
            o[d] = o[d] * a1 + o2 * a2;
        }
        m = mn;
    }
    if (slot == 0) {
        const float inv = 1.0f / l;
        bf16* op = a.O + ((long)b * a.o_bs + qi) * a.ldo + dcol;
#pragma unroll
        for (int c = 0; c < DPL / 8; ++c) {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (bf16)(o[8 * c + e] * inv);
            *(bf16x8*)(op + 8 * c) = t;
        }
    }
}

// Few keys (the decoder's image -> token attention: 4096 queries x 6..20 keys, transformer.py:176-180): one THREAD per
// (batch, head, query); the handful of K/V rows is shared by the whole block through L1.  q, o stay in registers.
template <int HD>
__global__ __launch_bounds__(256) void wg_attn_fewkeys_kernel(SmallAttnArgs a) {
    const long total = (long)a.B * a.heads * a.Lq;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    // heads fastest so that consecutive threads write consecutive 2*HD-byte pieces of one output row
    const int h = (int)(i % a.heads);
    const int qi = (int)((i / a.heads) % a.Lq);
    const int b = (int)(i / ((long)a.heads * a.Lq));
    const int dcol = h * HD;
    float q[HD], o[HD];
    {
        const bf16* qp = a.Q + ((long)b * a.q_bs + qi) * a.ldq + dcol;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            const bf16x8 t = *(const bf16x8*)(qp + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) q[8 * c + e] = (float)t[e] * a.scale * 1.4426950408889634f;
        }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    float m = -1.0e30f, l = 0.f;
    for (int j = 0; j < a.Lk; ++j) {
        const long row = (long)b * a.k_bs + j;
        const bf16* kp = a.K + row * a.ldk + dcol;
        const bf16* vp = a.V + row * a.ldv + dcol;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            const bf16x8 kk = *(const bf16x8*)(kp + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += q[8 * c + e] * (float)kk[e];
        }
        const float mn = fmaxf(m, s);
        const float alpha = exp2f(m - mn), p = exp2f(s - mn);
        m = mn;
        l = l * alpha + p;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            const bf16x8 vv = *(const bf16x8*)(vp + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[8 * c + e] = o[8 * c + e] * alpha + p * (float)vv[e];
        }
    }
    const float inv = 1.0f / l;
    bf16* op = a.O + ((long)b * a.o_bs + qi) * a.ldo + dcol;
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (bf16)(o[8 * c + e] * inv);
        *(bf16x8*)(op + 8 * c) = t;
    }
}

extern "C" int wg_mha_small_bf16(const void* Q, long ldq, long q_rows_per_batch, const void* K, long ldk, const void* V,
                                 long ldv, long k_rows_per_batch, void* O, long ldo, long o_rows_per_batch, int B,
                                 int heads, int head_dim, int Lq, int Lk, float scale, void* stream) {
    WG_REQUIRE(Q && K && V && O, "mha_small: null operand");
    WG_REQUIRE(B > 0 && heads > 0 && Lq > 0 && Lk > 0, "mha_small: bad shape");
    WG_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "mha_small: leading dimensions must be multiples of 8");
    WG_REQUIRE((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) == 0, "mha_small: misaligned operand");
    SmallAttnArgs a{(const bf16*)Q, (const bf16*)K, (const bf16*)V, (bf16*)O, ldq, ldk, ldv, ldo,
                    q_rows_per_batch, k_rows_per_batch, o_rows_per_batch, B, heads, Lq, Lk, scale};
    const long waves = (long)B * heads * Lq;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (Lk <= 32 && Lq >= 256 && (head_dim == 16 || head_dim == 32)) {
        dim3 g2((unsigned)((waves + 255) / 256));
        if (head_dim == 16) hipLaunchKernelGGL(wg_attn_fewkeys_kernel<16>, g2, block, 0, st, a);
        else hipLaunchKernelGGL(wg_attn_fewkeys_kernel<32>, g2, block, 0, st, a);
        return wg_check_launch("wg_mha_small_bf16(fewkeys)");
    }
    switch (head_dim) {
        case 16: hipLaunchKernelGGL(wg_attn_small_kernel<16>, grid, block, 0, st, a); break;
        case 32: hipLaunchKernelGGL(wg_attn_small_kernel<32>, grid, block, 0, st, a); break;
        case 64: hipLaunchKernelGGL(wg_attn_small_kernel<64>, grid, block, 0, st, a); break;
        case 128: hipLaunchKernelGGL(wg_attn_small_kernel<128>, grid, block, 0, st, a); break;
        default:
            wg_set_error("mha_small: head_dim %d not supported (16, 32, 64, 128)", head_dim);
            return WG_ERR_UNSUPPORTED;
    }
    return wg_check_launch("wg_mha_small_bf16");
}
