// Argument block and tile order shared by the GEMM kernels (gemm.hip; tools/micro/gemm_fr.hip).
#pragma once
#include "wg_common.h"

struct GemmArgs {
    const bf16* A; long lda;
    const bf16* W; long ldw;
    const bf16* bias;
    const bf16* R; long ldr; int res_mod;
    void* C; long ldc;
    int M, N, K;
    int act;
    int out_f32;
    int tiles_m, tiles_n;
    int col_block;               // tile order: column blocks of this many tile columns, row-major inside a block (0 = plain row-major)
    const float* ln_stats;       // LayerNorm folded into this GEMM (persistent kernel only): [M][2] = {mean, rstd} of the rows of A,
    const float* ln_s;           //   [N] column sums of the (gamma-scaled, bf16) weight rows,
    const float* ln_b;           //   [N] folded bias  b + W beta:   C = rstd * (A W'^T - mean * s) + b'
    const float* ln_part;        // ... or, instead of ln_stats, the rows' statistics as partial sums left by the GEMM that PRODUCED A
    int ln_np; long ln_mpad;     //   (stats_part below): [ln_np][ln_mpad][2] = {sum x, sum x^2} per 256-column tile of A's row;
    float ln_eps;                //   mean / rstd are formed in this kernel (K = the LayerNorm width)
    float* stats_part;           // producer side (persistent kernel, bf16 output, N % 256 == 0): [tiles_n][stats_mpad][2] fp32 =
    long stats_mpad;             //   {sum, sum of squares} of the STORED (bf16-rounded) values of each output row over the tile's 256 columns
    unsigned c_bytes, r_bytes;   // extents of C and R for the staged epilogue's buffer descriptors (0: not addressable in 32 bits)
    unsigned a_bytes = 0, w_bytes = 0;   // persistent 256x256 kernel: extents of A and W for its LDS-DMA descriptors (set by launch_pp_persist)
    const float* scale_a;        // fp8 operands (wg_gemm_fp8_bias_act): per-row scale of A [M] and per-output-channel scale of W [N];
    const float* scale_w;        //   A, W then point at e4m3 bytes and lda / ldw / K count PAIRS of bytes (see the entry point)
    const unsigned char* mx_a = nullptr;   // persistent fp8 kernel, MX operand: E8M0 scale of every 32-value block of A's rows, [K/32][mx_a_pitch]
    long mx_a_pitch = 0;                   //   bytes, the rows of a 128-row group permuted to (row % 16) * 8 + row / 16 (a lane's 8 fragments = 8 adjacent bytes)
    unsigned char* mx_c = nullptr;         // ... and the block scales of its e4m3 output copy Cq (below), same layout [N/32][mx_c_pitch]: the next
    long mx_c_pitch = 0;                   //   GEMM's mx_a
    unsigned mx_c_bytes = 0;
    const unsigned char* mx_w = nullptr;   // persistent fp8 kernel: the weights' block scales, [K/32][mx_w_pitch], rows of a 64-row group at
    long mx_w_pitch = 0;                   //   (row % 16) * 4 + row / 16 (a lane's 4 column fragments = one dword)
    void* Cq = nullptr;                    // persistent fp8 kernel: the stored values once more as e4m3 bytes [M][ldcq] with block scales mx_c
    long ldcq = 0;                         //   (the next fp8 GEMM's A operand); C itself may then be null (c_bytes 0: its stores are dropped)
    unsigned cq_bytes = 0;
    const bf16* sk_gamma;        // skinny kernel only: LayerNorm(A) applied to the rows on their way into the MFMA (gamma, beta [K], eps)
    const bf16* sk_beta;
    float sk_eps;
    int sk_tiled;                // skinny kernel only: W is in MFMA fragment order (wg_tile_weight_bf16) instead of row-major [N][K]
};

// Linear tile index -> (tile row, tile column).  With col_block = c > 0 the grid is walked in blocks of c tile columns, row-major
// inside a block: the workgroups that share an XCD (a contiguous range of this order) then touch only c column panels of W,
// which stay in that XCD's 4 MiB L2 while the A row panels stream past (plain row-major makes every XCD cycle through ALL
// of W once per round of tiles: at N = 2304..4096 that is 3.5-8 MB per round, and the measured HBM-side reads were 2-4x
// the operands, profiles/r01_gemm_traffic_by_shape.md).
__device__ __forceinline__ void wg_tile_of(int wgid, int tiles_m, int tiles_n, int col_block, int& tile_m, int& tile_n) {
#ifdef WG_GEMM_EXPERIMENT      // timing-only tile orders (notes/r05_experiments.md section 2): diagnostic builds only, the product walks col_block >= 0
    if (col_block == -1) {      // timing-only experiment: every tile reads (and writes) one of 4 x 4 tiles, so all operands stay in every XCD's L2
        tile_m = (wgid / tiles_n) & 3;
        tile_n = (wgid % tiles_n) & 3;
        return;
    }
    if (col_block < 0) {
        // Row bands (experiment, WG_GEMM_COLBLOCK = -(band height * 16 + column block)): the grid is walked band by band of `bh` tile rows; inside a
        // band in blocks of `cb` tile columns, row-major inside a block.  The 32 tiles an XCD has in flight then share bh row panels of A, which
        // stay in its L2 while the band's column blocks pass: A leaves L2 once instead of once per column block.
        const int bh = (-col_block) >> 4, cb = (-col_block) & 15;
        const int per_band = bh * tiles_n;
        const int band = wgid / per_band;
        const int r0 = band * bh;
        const int h = (tiles_m - r0) < bh ? (tiles_m - r0) : bh;      // (the last band may be lower)
        const int idx = wgid - band * per_band;      // (bands above a lower last band are full)
        const int per_block = h * cb;
        const int b = idx / per_block;
        const int c0 = b * cb;
        const int wdt = (tiles_n - c0) < cb ? (tiles_n - c0) : cb;
        const int i2 = idx - b * per_block;
        tile_m = r0 + i2 / wdt;
        tile_n = c0 + i2 % wdt;
        return;
    }
#endif
    if (col_block <= 0 || col_block >= tiles_n) {
        tile_m = wgid / tiles_n;
        tile_n = wgid % tiles_n;
        return;
    }
    const int per_block = tiles_m * col_block;
    const int b = wgid / per_block;                    // column block
    const int c0 = b * col_block;
    const int w = (tiles_n - c0) < col_block ? (tiles_n - c0) : col_block;   // width of this (possibly last, narrower) block
    const int idx = wgid - b * per_block;
    tile_m = idx / w;
    tile_n = c0 + idx % w;
}


// the experimental one-barrier persistent 256x256 kernel (tools/micro/gemm_fr.hip, -DWG_GEMM_FR builds): same operands and epilogue
// semantics as wg_gemm_pp_persist_kernel
#ifdef WG_GEMM_FR
int wg_launch_gemm_fr(GemmArgs& g, hipStream_t st);
int wg_gemm_fr_supports(const GemmArgs& g);
#endif
