// The two GEMMs of a Linear's backward pass on the operands AS THEY LIE in memory (the reference gets them from torch autograd over nn.Linear;
// trainable Linears: train_walkgpt.py:347-350 -- mask decoder, text_hidden_fcs, out_mm_projector, lm_head / LoRA):
//   dX[M, K] = dY[M, N] . W[N, K]          "NN": the reduction index n is the ROW of the second operand
//   dW[N, K] = dY[M, N]^T . X[M, K]        "TN": the reduction index m is the row of BOTH operands;  db[n] = sum_m dY[m, n] rides along
// gemm.hip computes A[M, K] . W[N, K]^T only (reduction contiguous in both operands: what a forward pass needs), so rounds 2-3 ran these two
// through transposed COPIES -- three wg_tokens_to_nchw launches, row padding and dtype copies per Linear, ~540 of the 1 300 launches of a head
// step (profiles/r03_train_head.md).  Here a reduction-major tile is staged into LDS row by row as it is read (coalesced 16-byte pieces) and
// turned into MFMA fragments by ds_read_b64_tr_b16, the hardware transpose read (4 rows x 16 columns per 16-lane group, column-major out).
//
// Tile: 128 x 128 outputs per 256-thread workgroup (2 x 2 waves, 64 x 64 each = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16), reduction in chunks
// of 32 rows, the next chunk's global loads in flight (registers) while the current one is multiplied.  A long reduction with few output
// tiles (dW of a 256 x 256 Linear over 32 768 image tokens: 4 tiles) is cut into S splits, each leaving an fp32 partial tile; a second
// kernel sums the partials in a FIXED order and rounds to the parameter dtype -- no atomics, two runs give the same bits.
// HBM-light, MFMA-light: bound by one CU's ingest rate (~64 flop per operand byte); a head step has ~60 of these at 4-30 us each.
#include "wg_common.h"

namespace {

constexpr int BT = 128;            // output tile side
constexpr int RC = 32;             // reduction rows per chunk
constexpr int PITCH_T = 320;       // bytes per LDS row of a reduction-major chunk [RC][128 bf16]: 256 + 64 (the four rows of a tr-read block on
                                   // four different 16-bank quarters)
constexpr int PITCH_N = 80;        // bytes per LDS row of a reduction-minor chunk [128][RC bf16]: 64 + 16 (b128 reads of 16 rows conflict-free)

struct BwdGemmArgs {
    const bf16* A; long lda;       // TN: [R, Mo] (reduction rows);  NN: [Mo, R]
    const bf16* B; long ldb;       // [R, Ko]
    float* part;                   // [S][Mo][Ko] fp32 partial sums (S > 1), or null
    void* out; int out_f32;        // S == 1: the result itself, [Mo, Ko] bf16 or fp32 (ldo = Ko)
    float* colpart;                // TN, optional: column sums of A -> [S][Mo] (S > 1) ...
    void* colout;                  // ... or the sums themselves [Mo] (S == 1), same dtype rule
    int Mo, Ko, R, S, rows_per_split;
};

__device__ __forceinline__ u32x2 tr_read(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// fragment (32 output rows or columns x 16 reduction steps) of a reduction-major chunk: lane l -> output index l % 32, reduction 8 (l / 32) .. +7
__device__ __forceinline__ bf16x8 frag_t(unsigned base, int c0, int r0, int lane) {
    const int g = lane >> 4, i16 = lane & 15;
    const unsigned ad = base + (unsigned)((r0 + 8 * (g >> 1) + (i16 >> 2)) * PITCH_T + (c0 + 16 * (g & 1) + 4 * (i16 & 3)) * 2);
    const u32x2 lo = tr_read(ad), hi = tr_read(ad + 4 * PITCH_T);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const u32x4 v = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, v);
}

template <bool TA>
__global__ __launch_bounds__(256) void wg_gemm_bwd_kernel(BwdGemmArgs g) {
    __shared__ __attribute__((aligned(16))) char lds_a[TA ? RC * PITCH_T : BT * PITCH_N];
    __shared__ __attribute__((aligned(16))) char lds_b[RC * PITCH_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BT, k0 = blockIdx.x * BT, split = blockIdx.z;
    const int r_begin = split * g.rows_per_split;
    const int r_end = r_begin + g.rows_per_split < g.R ? r_begin + g.rows_per_split : g.R;
    const bool colsum = TA && (g.colpart || g.colout) && blockIdx.x == 0 && wn == 0;

    // ---- global -> registers: 16-byte pieces, two per operand and thread --------------------------------------------------------------------
    // reduction-major chunk [RC][128]: piece p = tid + 256 u -> row p / 16, 8 columns at 8 (p % 16); columns past the matrix re-read its last
    // piece (their products land in outputs that are not stored), rows past the reduction are zero
    bf16x8 ra[2], rb[2];
    auto load_t = [&](const bf16* src, long ld, int c0, int ncols, int r0, bf16x8 (&reg)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = tid + 256 * u, r = r0 + (p >> 4);
            int c = c0 + 8 * (p & 15);
            c = c + 8 <= ncols ? c : ncols - 8;
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            reg[u] = r < r_end ? *(const bf16x8*)(src + (long)r * ld + c) : z;
        }
    };
    // reduction-minor chunk [128][RC] (NN: A = dY rows): piece p -> row p / 4, reduction 8 (p % 4) .. +7
    auto load_n = [&](int r0, bf16x8 (&reg)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = tid + 256 * u;
            int m = m0 + (p >> 2);
            m = m < g.Mo ? m : g.Mo - 1;
            const int r = r0 + 8 * (p & 3);
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (r + 8 <= r_end) v = *(const bf16x8*)(g.A + (long)m * g.lda + r);
            else if (r < r_end) {                   // the reduction's ragged end: element by element
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (r + e < r_end) v[e] = g.A[(long)m * g.lda + r + e];
            }
            reg[u] = v;
        }
    };
    auto park = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = tid + 256 * u;
            if (TA) *(bf16x8*)(lds_a + (p >> 4) * PITCH_T + (p & 15) * 16) = ra[u];
            else *(bf16x8*)(lds_a + (p >> 2) * PITCH_N + (p & 3) * 16) = ra[u];
            *(bf16x8*)(lds_b + (p >> 4) * PITCH_T + (p & 15) * 16) = rb[u];
        }
    };
    auto fetch = [&](int r0) __attribute__((always_inline)) {
        if (TA) load_t(g.A, g.lda, m0, g.Mo, r0, ra);
        else load_n(r0, ra);
        load_t(g.B, g.ldb, k0, g.Ko, r0, rb);
    };

    f32x16 acc[2][2], accc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int v = 0; v < 16; ++v) { acc[i][0][v] = 0.f; acc[i][1][v] = 0.f; accc[i][v] = 0.f; }
    }
    const bf16 one = (bf16)1.0f;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};
    const unsigned abase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_a;
    const unsigned bbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_b;

    fetch(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += RC) {
        __syncthreads();                     // the previous chunk's fragment reads are done
        park();
        __syncthreads();
        if (r0 + RC < r_end) fetch(r0 + RC);
#pragma unroll
        for (int ks = 0; ks < RC / 16; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (TA) fa[i] = frag_t(abase, wm * 64 + i * 32, ks * 16, lane);
                else fa[i] = *(const bf16x8*)(lds_a + (wm * 64 + i * 32 + (lane & 31)) * PITCH_N + (ks * 16 + 8 * (lane >> 5)) * 2);
                fb[i] = frag_t(bbase, wn * 64 + i * 32, ks * 16, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (colsum) accc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], ones, accc[i], 0, 0, 0);
            }
        }
    }

    // ---- stores: accumulator register v of lane l = output (row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32) of its 32 x 32 block ----------------
    const long plane = (long)g.Mo * g.Ko;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = k0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m0 + wm * 64 + i * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
                if (row < g.Mo && col < g.Ko) {
                    const long o = (long)row * g.Ko + col;
                    if (g.part) g.part[split * plane + o] = acc[i][j][v];
                    else if (g.out_f32) ((float*)g.out)[o] = acc[i][j][v];
                    else ((bf16*)g.out)[o] = (bf16)acc[i][j][v];
                }
            }
        }
    if (colsum && (lane & 31) == 0) {      // every column of the ones-product holds the sums: take column 0
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m0 + wm * 64 + i * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
                if (row < g.Mo) {
                    if (g.colpart) g.colpart[(long)split * g.Mo + row] = accc[i][v];
                    else if (g.out_f32) ((float*)g.colout)[row] = accc[i][v];
                    else ((bf16*)g.colout)[row] = (bf16)accc[i][v];
                }
            }
    }
}

// out[i] = sum over the S partial planes, in plane order (fixed: deterministic), rounded once; elements n .. n + nc - 1 come from colpart
__global__ __launch_bounds__(256) void wg_sum_partials_kernel(const float* part, const float* colpart, int S, long n, long nc, void* out, void* colout, int out_f32) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n + nc) return;
    const bool c = i >= n;
    const float* src = c ? colpart + (i - n) : part + i;
    const long stride = c ? nc : n;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += src[k * stride];
    void* dst = c ? colout : out;
    const long o = c ? i - n : i;
    if (out_f32) ((float*)dst)[o] = s;
    else ((bf16*)dst)[o] = (bf16)s;
}

}  // namespace

// Splits of the reduction: enough workgroups to cover the chip a couple of times when the output has few tiles, at least 256 rows each.
extern "C" int wg_gemm_bwd_splits(int Mo, int Ko, int R) {
    const int tiles = ((Mo + BT - 1) / BT) * ((Ko + BT - 1) / BT);
    int s = (512 + tiles - 1) / tiles;
    const int by_rows = (R + 255) / 256;
    s = s < by_rows ? s : by_rows;
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}

extern "C" long wg_gemm_bwd_workspace_floats(int Mo, int Ko, int R, int with_colsum) {
    const int S = wg_gemm_bwd_splits(Mo, Ko, R);
    return S > 1 ? (long)S * ((long)Mo * Ko + (with_colsum ? Mo : 0)) : 0;
}

static int launch_bwd(bool ta, const void* A, long lda, const void* B, long ldb, void* out, void* colout, int out_f32, float* ws, long ws_floats, int Mo, int Ko, int R,
                      void* stream, const char* what) {
    WG_REQUIRE(A && B && out && Mo > 0 && Ko > 0 && R > 0, "%s: bad arguments", what);
    WG_REQUIRE(Ko % 8 == 0 && ldb % 8 == 0 && lda % 8 == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && (!ta || Mo % 8 == 0) && Ko >= 8 && (!ta || Mo >= 8),
               "%s: 16-byte rows (every leading dimension and every reduction-major width a multiple of 8 elements)", what);
    const int S = wg_gemm_bwd_splits(Mo, Ko, R);
    const long need = S > 1 ? (long)S * ((long)Mo * Ko + (colout ? Mo : 0)) : 0;
    WG_REQUIRE(need == 0 || (ws && ws_floats >= need), "%s: workspace too small (need %ld floats: wg_gemm_bwd_workspace_floats)", what, need);
    BwdGemmArgs g{};
    g.A = (const bf16*)A; g.lda = lda; g.B = (const bf16*)B; g.ldb = ldb; g.Mo = Mo; g.Ko = Ko; g.R = R; g.S = S; g.out_f32 = out_f32;
    g.rows_per_split = ((R + S - 1) / S + RC - 1) / RC * RC;
    if (S > 1) { g.part = ws; g.colpart = colout ? ws + (long)S * Mo * Ko : nullptr; }
    else { g.out = out; g.colout = colout; }
    const dim3 grid((Ko + BT - 1) / BT, (Mo + BT - 1) / BT, S);
    if (ta) hipLaunchKernelGGL(wg_gemm_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(wg_gemm_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, g);
    if (S > 1) {
        const long n = (long)Mo * Ko, nc = colout ? Mo : 0;
        hipLaunchKernelGGL(wg_sum_partials_kernel, dim3((unsigned)((n + nc + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g.part, g.colpart, S, n, nc, out, colout, out_f32);
    }
    return wg_check_launch(what);
}

// dW[N, K] (+ db[N]) = dY[M, N]^T . X[M, K]: out / colsum_out bf16 (out_f32 = 0) or fp32; workspace = wg_gemm_bwd_workspace_floats(N, K, M, db != null) floats.
extern "C" int wg_gemm_tn_bf16(const void* dY, long lddy, const void* X, long ldx, void* dW, void* db, int out_f32, float* workspace, long workspace_floats, int M,
                               int N, int K, void* stream) {
    return launch_bwd(true, dY, lddy, X, ldx, dW, db, out_f32, workspace, workspace_floats, N, K, M, stream, "wg_gemm_tn_bf16");
}

// dX[M, K] = dY[M, N] . W[N, K]: out bf16 (out_f32 = 0) or fp32; workspace = wg_gemm_bwd_workspace_floats(M, K, N, 0) floats (0 unless N is long and M, K small).
extern "C" int wg_gemm_nn_bf16(const void* dY, long lddy, const void* W, long ldw, void* dX, int out_f32, float* workspace, long workspace_floats, int M, int N, int K,
                               void* stream) {
    return launch_bwd(false, dY, lddy, W, ldw, dX, nullptr, out_f32, workspace, workspace_floats, M, K, N, stream, "wg_gemm_nn_bf16");
}
