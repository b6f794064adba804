/* walkgpt_hip.h -- C ABI of libwalkgpt_hip.so: MI355X (gfx950) kernels for WalkGPT's grounded-segmentation forward path.
 *
 * The reference (rafiibnsultan/WalkGPT) has no FFI of its own: its hot path is a chain of stock torch.nn ops inside
 * Python nn.Modules.  Each entry point below therefore names the reference *expression* it replaces (file:line relative
 * to the reference repo); the Python modules in walkgpt_amd/ keep the reference's module/forward() surface and call
 * these through ctypes (walkgpt_amd/_lib.py).  INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers unless stated; `stream` is a hipStream_t passed as void*
 *   - bf16 = 16-bit brain float storage, all accumulation / statistics in fp32
 *   - "ld*" = leading dimension in ELEMENTS (row stride); rows are contiguous in their last dimension
 *   - return 0 on success, <0 on error (WG_ERR_*), never throw/abort; wg_last_error() gives the thread-local text
 *   - no allocation, no host synchronisation, no global mutable state: callers own every buffer; re-entrant across
 *     streams and threads; safe to capture into a hipGraph
 */
#ifndef WALKGPT_HIP_H
#define WALKGPT_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define WG_OK 0
#define WG_ERR_BAD_ARG (-1)
#define WG_ERR_UNSUPPORTED (-2)
#define WG_ERR_LAUNCH (-3)

/* activation codes of fused epilogues */
#define WG_ACT_NONE 0
#define WG_ACT_GELU_ERF 1   /* nn.GELU()            model/segment_anything/modeling/common.py:13-26, utils/utils_walkgpt.py:173.
                             * fp32 results and the fused decoder / norm kernels evaluate the erf form (A&S 7.1.26, 1.5e-7); the bf16 GEMM epilogues
                             * (0.2.1) evaluate a minimax sigmoid-form fit of it, |error| <= 2.6e-5 absolute, and round to bf16 (csrc/wg_common.h wg_act2e) */
#define WG_ACT_QUICK_GELU 2 /* x*sigmoid(1.702x)    HF CLIP MLP (custom_clip.py:50, third-party transformers) */
#define WG_ACT_RELU 3       /* nn.ReLU()            model/segment_anything/modeling/transformer.py:23, mask_decoder.py:186 */

int wg_version(void);               /* major*10000 + minor*100 + patch; 201 = 0.2.1: same entry points and argument lists as 0.2.0 (which changed them
                                     * against 0.1.0: INTEGRATION.md section 3); 0.2.1 = round 6's kernels (GELU form of the bf16 epilogues, above) */
const char* wg_last_error(void);    /* thread-local, valid until the next failing call on this thread */

/* C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ R[m % res_row_mod or m, :]).  bf16 in, bf16 or fp32 out.
 * Replaces every nn.Linear / 1x1 conv / stride==kernel conv (after wg_patchify_bf16) / 3x3 conv (after
 * wg_im2row3x3_bf16) / ConvTranspose2d(k2,s2) on the path:
 *   image_encoder.py:238 (qkv) :257 (proj) :422-426 (patch embed, + pos_embed :111-113 via R/res_row_mod)
 *   common.py:25 (MLP)   image_encoder.py:92-108 (neck)   transformer.py:222-224,240 (decoder projections)
 *   mask_decoder.py:53-63 (upscaler) :169-191 (hyper / IoU MLPs)   utils_walkgpt.py:171-175,209-212,226,249,312-316
 *   HF CLIPAttention / CLIPMLP linears (custom_clip.py:50-104 call site).
 * MFMA path needs K%64==0, N%4==0, N>=16, lda/ldw %8==0, ldc/ldr %4==0, 16-byte aligned A/W/C; anything else takes a
 * slower one-wave-per-row kernel.  tile_hint: 0 = auto (wg_gemm_pick_tile), 1 = 128x128 tiles (2 workgroups/CU),
 * 11 = persistent 128x128 tiles, 12 = 128x128 tiles whose last row tile absorbs M % 128 <= 16 rows, 2 = 256x256 tiles
 * (8 waves, 1 workgroup/CU), 14 = 256x256 tiles with the ping-pong schedule, 16 = the same as persistent tiles (one resident
 * workgroup per CU walks the tile grid; the auto choice for big shapes), 3 = force row-wave.  Variants that need the staged
 * 16-byte epilogue (11, 12, 16) fall back to 1 / 14 when the output is fp32, misaligned or larger than 2 GiB. */
int wg_gemm_pick_tile(int M, int N);
/* same, optionally allowing tile 12 = 128x128 tiles whose last row tile absorbs M % 128 <= 16 leftover rows (CLIP's
 * M = B*1025): faster when GEMMs run back to back on one stream, slower when two streams share the chip. */
int wg_gemm_pick_tile_ex(int M, int N, int allow_tail);
/* the same with K in view: the kernel id wg_gemm_bias_act_bf16 really launches (the skinny kernel 5 needs K % 128 == 0, else tile 1) */
int wg_gemm_pick_tile_mnk(int M, int N, int K, int allow_tail);
int wg_gemm_bias_act_bf16(const void* A, long lda, const void* W, long ldw, const void* bias, const void* residual,
                          long ldr, int res_row_mod, void* C, long ldc, int M, int N, int K, int act, int out_f32,
                          int tile_hint, void* stream);

/* Skinny rows (M <= 128: the [SEG] hidden states through text_hidden_fcs[0], utils/utils_walkgpt.py:321-323), one launch:
 *   C = act(LN?(A; gamma, beta, eps) . W^T + bias);  gamma == beta == null: no LayerNorm in front.
 * N % 16 == 0, K % 128 == 0; a workgroup per (16 output columns, 16 rows) repeats the exact two-pass row statistics of its rows in fp32.
 * w_tiled != 0: W is in fragment order (wg_tile_weight_bf16; ldw ignored) -- contiguous 1-KiB wave loads. */
int wg_gemm_skinny_ln_supported(int M, int N, int K, long lda, long ldw, long ldc);
int wg_gemm_skinny_ln_bias_act_bf16(const void* A, long lda, const void* gamma, const void* beta, float eps, const void* W, long ldw,
                                    int w_tiled, const void* bias, void* C, long ldc, int M, int N, int K, int act, int out_f32, void* stream);

/* The same product with the LayerNorm in front of it folded in (image_encoder.py:177-178 norm1 -> attn.qkv, :191 norm2 ->
 * mlp.lin1; HF CLIPEncoderLayer layer_norm1 -> q/k/v_proj, layer_norm2 -> mlp.fc1):
 *   C = act(rstd_m * (A Wg^T - mean_m * colsum) + bias_f32),  Wg = bf16(W * gamma), colsum[n] = sum_k Wg[n,k] (fp32),
 *   bias_f32[n] = b[n] + sum_k W[n,k] beta[k] (fp32), stats[m] = {mean, rstd} of row m of A from wg_row_stats_bf16.
 * A is the RAW residual stream, so the normalised rows never travel through HBM.  stats must be 16-byte aligned and hold M
 * rounded up to an even number of rows.  Runs on the persistent 256x256 kernel only: wg_gemm_ln_supported() != 0 says whether
 * a shape qualifies (callers otherwise run wg_layernorm_rows + wg_gemm_bias_act_bf16). */
int wg_row_stats_bf16(const void* x, long ldx, float* stats, int M, int D, float eps, void* stream);
int wg_gemm_ln_supported(int M, int N, int K, long lda, long ldw, long ldc);
int wg_gemm_ln_bias_act_bf16(const void* A, long lda, const void* Wg, long ldw, const float* bias_f32, const float* colsum,
                             const float* stats, void* C, long ldc, int M, int N, int K, int act, void* stream);

/* Row statistics handed from the GEMM that WRITES a residual-stream tensor to the LayerNorm-folded GEMM that reads it, instead of
 * a wg_row_stats_bf16 pass in between (the reference recomputes them inside nn.LayerNorm: image_encoder.py:177-178 norm1 behind the
 * previous block's mlp.lin2 (common.py:26) or the patch embedding (:422-426, :111-113); :191 norm2 behind attn.proj (:257); HF
 * CLIPEncoderLayer layer_norm1 behind the previous layer's mlp.fc2, layer_norm2 behind self_attn.out_proj):
 *   wg_gemm_bias_act_stats_bf16   = wg_gemm_bias_act_bf16 (bf16 output, persistent 256x256 kernel) that also writes
 *                                   row_partials[N / 256][mpad][2] fp32 = {sum, sum of squares} of the bf16 values it stored, per
 *                                   output row and 256-column tile; mpad = M rounded up to a multiple of 256 (rows >= M: unspecified).
 *   wg_gemm_lnp_bias_act_bf16     = wg_gemm_ln_bias_act_bf16 whose statistics are such partial sums (n_partials * 256 == K <= 1280):
 *                                   mean = S / K, rstd = (Q / K - mean^2 + eps)^-1/2 formed in the kernel.
 * wg_gemm_row_partials_supported() != 0: the producer's shape qualifies (N % 256 == 0, N <= 1280, plus wg_gemm_ln_supported's rules). */
int wg_gemm_row_partials_supported(int M, int N, int K, long lda, long ldw, long ldc);
int wg_gemm_bias_act_stats_bf16(const void* A, long lda, const void* W, long ldw, const void* bias, const void* residual, long ldr,
                                int res_row_mod, void* C, long ldc, int M, int N, int K, int act, float* row_partials, long mpad,
                                void* stream);
int wg_gemm_lnp_bias_act_bf16(const void* A, long lda, const void* Wg, long ldw, const float* bias_f32, const float* colsum,
                              const float* row_partials, int n_partials, long mpad, float eps, void* C, long ldc, int M, int N, int K,
                              int act, void* stream);

/* y = act(LayerNorm(x) * gamma + beta) per row, biased variance, eps inside the sqrt.
 * nn.LayerNorm: image_encoder.py:177,191 (eps 1e-6), transformer.py:157-181, utils_walkgpt.py:166-167,207,311,315,
 * HF CLIP layer norms; LayerNorm2d (common.py:31-43) on channels-last rows: neck (image_encoder.py:98,106),
 * upscaler (mask_decoder.py:57, with the following GELU fused through `act`). */
int wg_layernorm_rows(const void* x, long ldx, const void* gamma, const void* beta, void* y, long ldy, int M, int D,
                      float eps, int act, void* stream);

/* Multi-head attention softmax(scale * q k^T + key_bias) v, heads contiguous in the last dimension, batch b's rows
 * start at b * rows_per_batch (0 = shared by every batch item).  MFMA flash kernel, head_dim in {32, 64, 128}.
 * HF CLIPAttention with the additive key-padding mask of custom_clip.py:27-38 (key_bias = 0 / finfo.min, [B, Lk] fp32). */
int wg_mha_bf16(const void* Q, long ldq, long q_rows_per_batch, const void* K, long ldk, const void* V, long ldv,
                long k_rows_per_batch, void* O, long ldo, long o_rows_per_batch, const float* key_bias, int B,
                int heads, int head_dim, int Lq, int Lk, float scale, void* stream);

/* Same contract without key_bias, one wave per (batch, head, query); head_dim in {16, 32, 64, 128}.  For shapes with
 * few queries or few keys: two-way decoder attention (transformer.py:220-242) and MSQP cross attention
 * (nn.MultiheadAttention in utils_walkgpt.py:168,181). */
int wg_mha_small_bf16(const void* Q, long ldq, long q_rows_per_batch, const void* K, long ldk, const void* V, long ldv,
                      long k_rows_per_batch, void* O, long ldo, long o_rows_per_batch, int B, int heads, int head_dim,
                      int Lq, int Lk, float scale, void* stream);

/* SAM ViT attention on a packed qkv buffer [B*grid*grid, 3*heads*head_dim] -> out [B*grid*grid, heads*head_dim].
 * window == grid: global attention; window < grid: image_encoder.py:263-318 window partition with zero padding (pad
 * positions act as keys/values equal to qkv_bias, because the padding follows norm1), fused with
 * Attention.forward :235-260 and add_decomposed_rel_pos :321-392 (unscaled q, index q-k+S-1).
 * Compiled (head_dim, window): (64,14) (64,64) (64,32) (32,14) (32,28) (80,14) (80,64) -- 80 = SAM ViT-H. */
int wg_sam_attn_relpos_bf16(const void* qkv, const void* qkv_bias, const void* rel_pos_h, const void* rel_pos_w,
                            void* out, int B, int grid, int window, int heads, int head_dim, float scale, void* stream);
/* The same for the fp8 chain (head_dim 64; window 14 or the 64 x 64 global grid): the output leaves as e4m3 bytes out_q [B * grid^2, heads * 64] + E8M0 block
 * scales out_mx [heads * 2][mx_pitch] in wg_quantize_mx_fp8's group-128 layout -- bit for bit that pass's result on the bf16 output, which is never written
 * (the proj Linear behind the attention, image_encoder.py:235-260, takes it as its MX operand). */
int wg_sam_attn_mx_supported(int B, int grid, int window, int heads, int head_dim);
int wg_sam_attn_relpos_mx_bf16(const void* qkv, const void* qkv_bias, const void* rel_pos_h, const void* rel_pos_w, void* out_q, void* out_mx, long mx_pitch,
                               int B, int grid, int window, int heads, int head_dim, float scale, void* stream);

/* NCHW bf16 images -> rows [B*(H/P)*(W/P), Kpad], columns (c, ky, kx) zero padded to Kpad: the im2row of
 * Conv2d(kernel=stride=P)  (image_encoder.py:422-426; CLIP patch_embedding). */
int wg_patchify_bf16(const void* images, void* rows, int B, int C, int H, int W, int P, int Kpad, void* stream);

/* channels-last [B,H,W,C] -> rows [B*H*W, 9*C], columns (ky, kx, c), zero padding 1 (neck 3x3 conv, image_encoder.py:99-106). */
int wg_im2row3x3_bf16(const void* x, void* rows, int B, int H, int W, int C, void* stream);

/* out[r,:] = a[r,:] + b[r % b_rows,:]  (q + query_pe, k + key_pe: transformer.py:159-176; src + dense: mask_decoder.py:138). */
int wg_add_rows_bf16(const void* a, long lda, const void* b, long ldb, int b_rows, void* out, long ldo, long rows,
                     int cols, void* stream);

/* [B, HW, C] <-> [B, C, HW] (x.permute(0,3,1,2) image_encoder.py:118-124; flatten(2).permute(0,2,1) transformer.py:83-84). */
int wg_tokens_to_nchw_bf16(const void* x, void* y, int B, int HW, int C, void* stream);
int wg_nchw_to_tokens_bf16(const void* x, void* y, int B, int HW, int C, void* stream);

/* PositionEmbeddingRandom.forward (prompt_encoder.py:216-229) as rows [h*w, 2*num_feats] fp32 from the fp32 [2, num_feats] matrix. */
int wg_dense_pe_f32(const float* gaussian, float* pe_tokens, int h, int w, int num_feats, void* stream);
int wg_cast_f32_to_bf16(const float* x, void* y, long n, void* stream);

/* masks[t,k,Y,X] = sum_c hyper[t,first_mask+k,c] * up[t,c,Y,X] (mask_decoder.py:150-160); `up` is the pixel-shuffled
 * GEMM output of the two transposed convs: rows (t, y, x, dy, dx), columns (dy2, dx2, c), channels == 32. */
int wg_hyper_mask_dot(const void* up, const void* hyper, float* masks, int T, int h, int w, int channels, int nmask_total,
                      int first_mask, int num_masks, void* stream);

/* ---- fp8 (OCP e4m3) GEMM path, BASELINE config C5 (qkv / proj / MLP GEMMs in fp8, attention and LayerNorm statistics
 * stay bf16 / fp32).  Not in the reference (it runs bf16): the operation replaced is still `nn.Linear` (image_encoder.py:177-193). ----
 * wg_quantize_rows_fp8:       q[m][:] = e4m3(x[m][:] / scale[m]), scale[m] = max|x[m][:]| / 448.  x [M,K] bf16, q [M,K] bytes.
 * wg_layernorm_quantize_fp8:  the same on LayerNorm(x) (norm1 / norm2 of a block, fp32 statistics), one pass.
 * wg_gemm_fp8_bias_act:       C[M,N] bf16 = act(scale_a[m] scale_w[n] sum_k Aq[m,k] Wq[n,k] + bias[n]) (+ residual[m % res_row_mod]);
 *                             block-scaled MFMA 16x16x128 on the 256x256 ping-pong tile; K % 128 == 0, lda / ldw % 16 == 0. */
int wg_quantize_rows_fp8(const void* x, long ldx, void* q, long ldq, float* scale, int M, int K, void* stream);
int wg_layernorm_quantize_fp8(const void* x, long ldx, const void* gamma, const void* beta, float eps, void* q, long ldq, float* scale,
                              int M, int K, void* stream);
int wg_gemm_fp8_bias_act(const void* Aq, long lda, const float* scale_a, const void* Wq, long ldw, const float* scale_w, const void* bias,
                         const void* residual, long ldr, int res_row_mod, void* C, long ldc, int M, int N, int K, int act, void* stream);
/* wg_quantize_mx_fp8: x [M,K] bf16 -> e4m3 bytes + E8M0 block scales [K/32][pitch] (rows permuted inside `group`-row groups: 128 for an
 * activation operand, 64 for a weight operand).  The scale of a block is the power of two at or above max|block| / 448 (E8M0 byte =
 * 127 + exponent; no value saturates); inside a 128-row group row r sits at byte (r % 16) * 8 + r / 16 of its plane, inside a 64-row
 * group at (r % 16) * 4 + r / 16: the 8 (4) MFMA fragments of a lane are adjacent bytes.
 * wg_gemm_mxfp8: the persistent 256x256 fp8 GEMM with block scales on BOTH operands, applied inside the MFMA:
 *   C[M,N] bf16 = act(sum_k deq(Aq)[m,k] deq(Wq)[n,k] + bias[n]) (+ residual[m % res_row_mod]).
 *   Optional, as in the bf16 kernel it shares loop and epilogues with (wg_gemm_lnp_bias_act_bf16 / wg_gemm_bias_act_stats_bf16):
 *   ln_colsum != NULL: LayerNorm of A's rows folded in -- Wq = the gamma-scaled weight, ln_colsum [N] the row sums of its dequantised
 *     values, ln_bias [N] = b + W beta, ln_part [ln_np][ln_mpad][2] the {sum, sum of squares} partials of the rows Aq was quantised
 *     from (K = 256 ln_np); stats_part != NULL (N % 256 == 0): leaves such partials for its own output; Cq != NULL (N % 32 == 0): the
 *     stored values once more as e4m3 [M][ldcq] + block scales c_mx [N/32][c_pitch] (the next call's A operand; C may then be NULL). */
int wg_gemm_mxfp8(const void* Aq, long lda, const void* a_mx, long a_pitch, const void* Wq, long ldw, const void* w_mx, long w_pitch,
                  const void* bias, const float* ln_colsum, const float* ln_bias, const float* ln_part, int ln_np, long ln_mpad, float ln_eps,
                  const void* residual, long ldr, int res_row_mod, void* C, long ldc, void* Cq, long ldcq, void* c_mx, long c_pitch,
                  float* stats_part, long stats_mpad, int M, int N, int K, int act, void* stream);
int wg_quantize_mx_fp8(const void* x, long ldx, void* q, long ldq, void* mx, long pitch, int group, int M, int K, float* part, long part_mpad,
                       void* stream);   /* part != NULL (K % 256 == 0): + the rows' {sum, sum of squares} per 256-column tile, [K/256][part_mpad][2] */

/* mask_decoder.py:140-160 fused: `upscaled = output_upscaling(src)` (ConvT k2 s2 -> LayerNorm2d -> GELU -> ConvT k2 s2 -> GELU) and
 * `masks = hyper_in @ upscaled` in one launch; every step is local to an image token.  x [P*h*w, 256] bf16 token rows; w1 [(dy,dx,64),
 * 256], w2 [(dy,dx,32), 64]: the transposed convolutions re-laid as GEMM weights; hyper [P, nmask_total, 32] fp32;
 * out [P, num_masks, 4h, 4w] fp32.  fp32 inside (the 64-channel intermediate feeds the second MFMA as a bf16 hi + lo pair). */
int wg_upscale_mask_bf16(const void* x, long ldx, const void* w1, const void* b1, const void* ln_g, const void* ln_b, float eps,
                         const void* w2, const void* b2, const float* hyper, float* out, int P, int h, int w, int nmask_total,
                         int first_mask, int num_masks, void* stream);

/* Token side of SAM's two-way transformer (transformer.py:62-182) and the decoder heads (mask_decoder.py:146-160): the six tokens of a
 * prompt ([iou, mask0..3, prompt], mask_decoder.py:125-132) stay fp32; the two heavy pieces (token->image attention over the hw image
 * tokens, the 2048-wide MLP) are spread over the chip and hand fp32 partials across kernel boundaries.
 *
 * wg_dec_tokens_f32: one workgroup per prompt runs the stages named by the bit mask `stages`, in this order:
 *   1 SUM_MLP  x += mlp.lin2.bias + sum of the S = wg_dec_mlp_slices() MLP partials (:169-171); norm3; k / v of the image->token attention (:173-176)
 *              -> k_i2t / v_i2t [P,6,128] bf16
 *   2 SELF     self attention (+ query_pe unless skip_pe) and norm1 (:153-160)
 *   4 Q_T2I    q_proj(x + query_pe) of the token->image attention (:162-165; tail :96-99) -> q_t2i [P,6,128] fp32
 *   8 COMBINE  merge the attention partials; out_proj + residual; norm2 / norm_final_attn (:165-167; tail :100-101)
 *  16 INIT     (modifier of the first launch) queries = query_pe = cat(init_tokens [5,256] fp32, init_prompt [P,256] bf16)
 *              (mask_decoder.py:125-132: cat(iou_token, mask_tokens) | sparse prompt); both buffers are written
 *   queries [P,6,256] fp32 in / out; query_pe [P,6,256] fp32; weights: 24 bf16 device pointers (weight, bias / gamma, beta; slots of
 *   stages not requested may be null, so one launch can close block i (SUM_MLP) and open block i+1 (SELF, Q_T2I)):
 *   self_attn q,k,v,out [0..7] | norm1 [8,9] | token->image q,out [10..13] | norm2 or norm_final_attn [14,15] | mlp.lin2.bias [16] |
 *   unused [17] | norm3 [18,19] | image->token k,v [20..23].
 * wg_dec_attn_partial_f32: softmax(q k^T / 4) v per (prompt, head, split of 1024 keys), a wave per 256 keys.  Kimg / Vimg: the projected image
 *   tokens (k_proj(keys + key_pe), v_proj(keys)) as bf16 rows [P or 1][hw] with row stride ld_img, head h's 16 columns at + h * head_stride
 *   (16: two plain 128-column blocks; 32: columns ordered [K_h | V_h] per head, Vimg = Kimg + 16: one 64-byte piece per key and head)
 *   (img_rows_per_prompt = 0 when all prompts share one image; prompt_image [P] int32 != null: prompt p reads image prompt_image[p] --
 *   the first block, whose image tokens are those of the image for every one of its prompts); partials [P, 8, n_splits, 108] fp32 = {running max[6], sum[6], o[6][16]}, n_splits = ceil(hw / 1024).
 * wg_dec_mlp_partial_f32: slice s of S = wg_dec_mlp_slices() (16 since 0.2.1; 8 before) of mlp(x), w = 2048 / S: relu(x lin1[w s .. +w-1]^T + b1) lin2[:, w s .. +w-1]^T -> partials [P,S,6,256].
 * wg_dec_heads_f32: output_hypernetworks_mlps[i](x[:, 1 + i]) -> hyper_out [P,4,32]; iou_prediction_head(x[:, 0]) -> iou_out [P,4];
 *   weights: 30 bf16 pointers = (hypernetwork 0..3, IoU head) x layers[0..2] x (weight, bias).
 * Both take an optional `combine` table of 5 pointers {attention partials, out_proj weight [256,128], out_proj bias, LayerNorm gamma, beta}:
 *   when given, x holds the tokens BEFORE the COMBINE stage and the launch runs that stage itself (one launch less on the chain):
 *   the MLP kernel writes the tokens after norm2 to x_out (a buffer other than x); the heads kernel applies the final attention's
 *   out_proj + norm_final_attn to the token row each workgroup needs. */
/* Every weight MATRIX the four wg_dec_* token kernels take (not biases / LayerNorm vectors) is in fragment order, made once per checkpoint:
 *   wg_tile_weight_bf16: W [N][K] bf16 (row stride ld) -> T[ceil(N/16)][K/32][64][8],  T[nb][ks][lane][j] = W[16 nb + lane%16][32 ks + 8 (lane/16) + j]
 *   (rows >= N zero): the 1 KiB a wave loads per MFMA step is contiguous (3x the per-CU streaming rate of the row-major pattern).
 *   mlp.lin2.weight [256, 2048] is tiled per 256-column slice (8 calls with W + 256 s, ld 2048, N = K = 256 -> T + 65536 s). */
int wg_tile_weight_bf16(const void* W, long ld, int N, int K, void* tiled, void* stream);
int wg_dec_tokens_f32(int stages, int skip_pe, float* queries, float* query_pe, const float* init_tokens, const void* init_prompt,
                      const void* const* weights, int n_weights, float* q_t2i, const float* attn_partials, int n_splits, const float* mlp_partials, void* k_i2t, void* v_i2t, int P,
                      float eps, void* stream);
/* The first launch of a decode (INIT set) with the text projector's tail (utils_walkgpt.py:324-327) folded in: init_prompt [P,256] bf16 = the rows
 * BEFORE the tail (output of CalibratedTextProjector.net[3]); prompt_tail = {net[4] gamma, beta, text_type, log_temp} (bf16; 256 / 256 / 256 / 1
 * elements), prompt_tail_eps = net[4].eps.  Same bits as wg_ctp_tail_bf16 followed by wg_dec_tokens_f32. */
int wg_dec_tokens_ctp_f32(int stages, int skip_pe, float* queries, float* query_pe, const float* init_tokens, const void* init_prompt,
                          const void* const* prompt_tail, float prompt_tail_eps, const void* const* weights, int n_weights, float* q_t2i,
                          const float* attn_partials, int n_splits, const float* mlp_partials, void* k_i2t, void* v_i2t, int P, float eps, void* stream);
int wg_dec_attn_partial_f32(const float* q, const void* Kimg, const void* Vimg, long ld_img, int head_stride, long img_rows_per_prompt,
                            const int* prompt_image, int hw,
                            float* partials, int n_splits, int P, void* stream);
int wg_dec_mlp_slices(void);       /* S = slices of the MLP's 2048 hidden units in wg_dec_mlp_partial_f32 / wg_dec_tokens_f32 (partials [P, S, 6, 256]; lin2 tiled in S K-slices) */
int wg_dec_mlp_partial_f32(const float* x, const void* const* combine, int n_splits, float eps, float* x_out, const void* lin1_w,
                           const void* lin1_b, const void* lin2_w, float* partials, int P, void* stream);
int wg_dec_heads_f32(const float* x, const void* const* combine, int n_splits, float eps, const void* const* weights, int n_weights,
                     float* hyper_out, float* iou_out, int P, void* stream);

/* Image side of a TwoWayAttentionBlock after its token stages (transformer.py:173-180), one launch:
 *   keys = norm4(keys + out_proj(softmax(q k^T / 4) v)),  every image token attending to the six prompt tokens.
 * q [rows or hw][ldq] bf16: the q columns (128) of the fused image-side projection; kq / vq [P,6,128] bf16 (SUM_MLP stage above);
 * wo [256,128], bo [256]: cross_attn_image_to_token.out_proj; res: the image tokens themselves (bf16 rows, stride ldr); ln_g / ln_b: norm4;
 * res_bias [256] bf16 or null: a constant row added to res (the dense no-mask embedding when the caller folded it into the first block);
 * row_mod = hw when q and res hold ONE image shared by all P prompts, 0 when they hold P*hw rows; prompt_image [P] int32 != null: q and
 * res hold one block of hw rows per IMAGE and prompt p reads block prompt_image[p]; out [P*hw, 256] bf16.  hw % 16 == 0. */
int wg_dec_i2t_rows_bf16(const void* q, long ldq, const void* kq, const void* vq, const void* wo, const void* bo, const void* res, long ldr,
                         const void* res_bias, int row_mod, const int* prompt_image, const void* ln_g, const void* ln_b, float eps, void* out, int P, int hw, void* stream);

/* Sam.postprocess_masks (sam.py:137-172): bilinear to img_size^2, crop [:in_h,:in_w], bilinear to (out_h,out_w), one pass. */
int wg_postprocess_masks_f32(const float* low_res, float* out, int N, int low_h, int low_w, int img_size, int in_h,
                             int in_w, int out_h, int out_w, void* stream);

/* mask score = sum(sigmoid(x)[x>0]) / (count[x>0] + 1e-6) per mask (model/walkgpt.py:540-542, :737).  Two-pass,
 * atomics-free reduction; the caller provides wg_mask_score_workspace_floats(N, hw) floats of scratch. */
/* Both of the above in one pass over the output: out [N, out_h, out_w] fp32 and score [N] fp32; scratch = wg_postprocess_score_workspace_floats. */
long wg_postprocess_score_workspace_floats(int N, int out_h, int out_w);
int wg_postprocess_masks_score_f32(const float* low_res, float* out, float* score, float* workspace, long workspace_floats, int N, int low_h,
                                   int low_w, int img_size, int in_h, int in_w, int out_h, int out_w, void* stream);
/* The same in ONE launch (the workgroup that completes a mask folds its partials; same pixels, scores up to the summation order): `tickets` = N words, zero before the first call and
 * left zero by every call; calls that share them must be ordered (one stream). */
int wg_postprocess_masks_score_fused_f32(const float* low_res, float* out, float* score, float* workspace, long workspace_floats, unsigned* tickets,
                                         int N, int low_h, int low_w, int img_size, int in_h, int in_w, int out_h, int out_w, void* stream);
long wg_mask_score_workspace_floats(int N, long hw);
int wg_mask_score_f32(const float* masks, float* score, float* workspace, long workspace_floats, int N, long hw,
                      void* stream);

/* SURVEY.md 8(f) rows 1-2.  Per-mask one-pass statistics of fp32 logits [N, hw] against fp32 ground truth [N, hw]; scratch =
 * wg_mask_stats_workspace_floats(N, hw) floats.
 * wg_mask_iou_f32: intersectionAndUnionGPU(pred > 0, gt, K = 2, ignore) (utils/utils.py:192-204) with the threshold fused;
 *   out6[n] = {inter0, inter1, union0, union1, target0, target1}.
 * wg_mask_losses_f32: out2[n] = {mean BCE-with-logits, dice loss} of mask n (utils/utils_walkgpt.py:76-120). */
long wg_mask_stats_workspace_floats(int N, long hw);
int wg_mask_iou_f32(const float* pred_logits, const float* gt, float* out6, float* workspace, long workspace_floats, int N,
                    long hw, float ignore_value, void* stream);
int wg_mask_losses_f32(const float* pred_logits, const float* targets, float* out2, float* workspace, long workspace_floats,
                       int N, long hw, float dice_scale, float dice_eps, void* stream);

/* LLM-side multimodal splice = prepare_inputs_labels_for_multimodal (llava_arch.py:265-518) for rows
 * with exactly one IMAGE_TOKEN_INDEX placeholder, fused with the embed_tokens gather, plus WalkGPT's [SEG] read-out mask in
 * spliced coordinates (model/walkgpt.py:293-306).  ids [rows,L] int64; table [V,H] bf16; image_features [rows,T,H] bf16;
 * mask_in [rows,L] / vit_mask [rows,T] bool bytes (null = ones); labels_in [rows,L] int64 (optional); seg_ids [nseg] int64
 * (device, optional).  Outputs: embeds [rows,L+T-1,H]; mask_out, labels_out, seg_mask [rows,L+T-1] (optional);
 * img_pos / img_cnt [rows] int32 -- the caller must reject rows whose img_cnt != 1. */
int wg_splice_multimodal_bf16(const long* ids, const void* table, const void* image_features, const void* mask_in,
                              const void* vit_mask, const long* labels_in, const long* seg_ids, int nseg, void* embeds,
                              void* mask_out, long* labels_out, void* seg_mask, int* img_pos, int* img_cnt, int rows, int L, int T,
                              int H, int V, long image_token, long ignore_index, void* stream);

/* input pipeline.  frames [B,H,W,3] uint8 (HBM) -> images [B,3,S,S] bf16 (out_bf16) or fp32:
 * ResizeLongestSide(S).apply_image (segment_anything/utils/transforms.py:27-36 = Pillow's two-pass 8-bit bilinear resize,
 * reproduced bit for bit from its coefficient tables), (x - mean) / std and zero padding to S x S (utils/PAVE_dataset.py:115-121).
 * h_bounds [Wo,2] / h_kk [Wo,h_ksize] and v_bounds [Ho,2] / v_kk [Ho,v_ksize]: int32 tables of Resample.c's precompute_coeffs +
 * normalize_coeffs_8bpc for (W -> Wo) and (H -> Ho); null = that pass is the identity.  tmp: B*H*Wo*3 bytes; resized
 * (optional): the uint8 resized frames [B,Ho,Wo,3]; norm_lut: device fp32 [3][256] = (v - mean_c) / std_c in IEEE fp32. */
int wg_preprocess_frames_u8(const void* frames, void* tmp, void* resized, void* out, int out_bf16, const int* h_bounds,
                            const int* h_kk, int h_ksize, const int* v_bounds, const int* v_kk, int v_ksize, int B, int H, int W,
                            int Ho, int Wo, int S, const float* norm_lut, void* stream);

/* cost matrix of match_pred() (utils/matcher.py:93-133): P predicted logit masks and T target masks
 * [., H, W] fp32, sampled bilinearly (grid_sample, align_corners=False, zero padding) at NP shared points in [0,1]^2 (x, y);
 * cost[p*T + t] = batch_sigmoid_ce_loss + batch_dice_loss (:10-56).  Scratch: wg_match_cost_workspace_floats(P, T, NP). */
long wg_match_cost_workspace_floats(int P, int T, int NP);
int wg_match_cost_f32(const float* pred_logits, const float* targets, const float* points, float* cost, float* workspace,
                      long workspace_floats, int P, int T, int H, int W, int NP, void* stream);

/* region-alignment InfoNCE forward -- infonce_loss() + TinyCrossAttn.forward()
 * (utils/utils_walkgpt.py:8-73, 330-357; called at model/walkgpt.py:459-473), normalize=True.
 *   wg_row_inv_norm_bf16      out[r] = 1 / max(||x_r||, eps)                  (F.normalize of the SAM tokens, :53-54)
 *   wg_l2_normalize_rows_bf16 y_r = x_r / max(||x_r||, eps), bf16             (F.normalize of the [SEG] embeddings, :46-48)
 *   wg_nce_attn_f32           ST is the fp32 GEMM [2M, rows*N] = [Zn ; Wk^T q] . tokens^T (wg_gemm_bias_act_bf16, out_f32).
 *                             Per [SEG] m: attn_w[m, :] = softmax(ST[M+m, own row] * attn_scale) (:349-351); vraw[m, :] = the
 *                             top-k refinement sum_k alpha_k kv[idx_k] (:35-39) when 0 < top_k < N (top_k <= 32), else the
 *                             attention-pooled raw token sum_n attn_n kv_n (the caller applies Wv / Wo: weights sum to 1).
 *   wg_nce_loss_f32           pos = Zn_m . normalize(vpos_m); logits = [pos, ST[m, t] * inv_norm[t]] / temperature with the
 *                             own row at -inf if exclude_same_row; loss_m = cross entropy with label 0 (:56-71);
 *                             loss[0] = mean(loss_m); logits (optional, [M, 1 + rows*N]) receives the rows. */
int wg_row_inv_norm_bf16(const void* x, long ld, float* out, long R, int D, float eps, void* stream);
int wg_l2_normalize_rows_bf16(const void* x, long ldx, void* y, long ldy, long R, int D, float eps, void* stream);
int wg_nce_attn_f32(const float* ST, long ldst, const void* tokens, long ldt, const int* seg_row, float* attn_w, float* vraw,
                    int M, int N, int rows, int D, int top_k, float attn_scale, void* stream);
int wg_nce_loss_f32(const float* ST, long ldst, const float* inv_norm, const void* Zn, const float* vpos, const int* seg_row,
                    float* loss_m, float* loss, float* logits, int M, int N, int rows, int D, int exclude_same_row,
                    float temperature, void* stream);

/* MSQP pieces (utils/utils_walkgpt.py): _pool_grid_tokens :195-201, _global_token :256-257, SegAwareGate tail :213-217. */
int wg_avgpool_tokens_bf16(const void* x, void* y, int B, int H, int W, int C, int s, void* stream);
int wg_mean_tokens_bf16(const void* x, void* y, int B, int L, int C, void* stream);
int wg_sigmoid_gate_bf16(const void* x, const float* logit, void* y, long rows, int C, void* stream);

/* CTP tail (utils_walkgpt.py:321-327): normalize(LayerNorm(x) + text_type, dim=-1, eps 1e-12) * exp(log_temp). */
int wg_ctp_tail_bf16(const void* x, long ldx, const void* gamma, const void* beta, const void* text_type,
                     const void* log_temp, void* y, long ldy, int M, int C, float eps, void* stream);

/* [n, p*p, C] -> bilinear(align_corners=False) -> [n, t*t, C] (llava_arch.py:252-259; clip_encoder.py:47-49). */
int wg_resample_tokens_bf16(const void* x, void* y, int n, int p, int t, int C, void* stream);

/* ---- backward passes of the trainable grounding head (train_walkgpt.py:347-350: mask decoder, text_hidden_fcs, projector; the reference
 * obtains them from torch autograd over nn.Linear / nn.LayerNorm / nn.GELU / nn.ReLU ...).  A Linear's gradients are GEMMs on transposed
 * copies (wg_gemm_bias_act_bf16); these are the pieces that are not:
 * wg_colsum_f32:          out[c] += sum_r x[r][c]   (bias gradient; fp32, the caller zeroes `out`)
 * wg_act_bf16 / _bwd:     y = act(x); dx = dy * act'(x)  (act codes of the GEMM epilogue: 1 erf-GELU, 2 quick-GELU, 3 ReLU)
 * wg_layernorm_bwd_bf16:  dx [M,C] bf16, dgamma / dbeta [C] fp32 (+=) of y = LayerNorm(x) gamma + beta */
int wg_colsum_f32(const void* x, long ldx, float* out, int R, int C, void* stream);
/* The same without atomics (per-workgroup partial rows folded in a fixed order): out [C] is written, fp32 or bf16; workspace: wg_colsum_det_workspace_floats. */
long wg_colsum_det_workspace_floats(int R, int C);
int wg_colsum_det_f32(const void* x, long ldx, void* out, int out_f32, float* workspace, long workspace_floats, int R, int C, void* stream);
int wg_act_bf16(const void* x, void* y, long n, int act, void* stream);
int wg_act_bwd_bf16(const void* x, const void* dy, void* dx, long n, int act, void* stream);
int wg_layernorm_bwd_bf16(const void* x, long ldx, const void* gamma, const void* dy, long lddy, void* dx, long lddx, float* dgamma,
                          float* dbeta, float* row_stats, int M, int C, float eps, void* stream);   /* row_stats: 2 M floats, needed when C > 4096 */
/* The same for rows of 64 / 128 / 256 / 512 channels without atomics: a workgroup per run of rows leaves one partial {dgamma, dbeta} row pair,
 * summed in a fixed order; dgamma / dbeta are WRITTEN (bf16, or fp32 with out_f32).  workspace: wg_layernorm_bwd_det_workspace_floats(M, C) floats.
 * Returns -2 without launching for any other width. */
long wg_layernorm_bwd_det_workspace_floats(int M, int C);
int wg_layernorm_bwd_det_bf16(const void* x, long ldx, const void* gamma, const void* dy, long lddy, void* dx, long lddx, void* dgamma, void* dbeta,
                              int out_f32, float* workspace, long workspace_floats, int M, int C, float eps, void* stream);
/* wg_l2norm_scale_bf16 / _bwd: y = x / max(|x|, eps) * exp(log_temp), the tail of CalibratedTextProjector behind its LayerNorm and type
 *   embedding (utils_walkgpt.py:325-327) as a separate operator, and its backward (dx bf16; dlog_temp fp32 +=).  C <= 512.
 * wg_attn_bwd_bf16: gradients of o = softmax(scale q k^T) v per (batch, head) for the head's small attentions (two-way transformer,
 *   transformer.py:185-240; CrossAttnBlock / TinyCrossAttn, utils_walkgpt.py:163-185,330-357): q [B,Lq,D], k / v [B,Lk,D], o / dout [B,Lq,D]
 *   contiguous bf16, min(Lq, Lk) <= 16, head_dim % 8 == 0, <= 128.  The long side's gradients are written as bf16, the short side's
 *   WRITTEN as fp32 (round 4: per-wave LDS rows and per-workgroup partial planes summed in a fixed order, no atomics); wg_attn_bwd_short_side: 1 = keys short (dq bf16, dk / dv fp32), 0 = queries short (dk / dv
 *   bf16, dq fp32), -1 = unsupported.  workspace: wg_attn_bwd_workspace_floats floats.
 * wg_postprocess_masks_bwd_f32: adjoint of wg_postprocess_masks_f32 (d loss / d low-res logits; written, not accumulated -- a gather in a fixed
 *   order since round 4: no atomics, the same bits every run).
 * wg_mask_losses_bwd_f32: d(g_bce sigmoid_ce_loss + g_dice dice_loss) / d logits (utils_walkgpt.py:76-120); workspace
 *   wg_mask_stats_workspace_floats(N, hw) + 2 N floats. */
int wg_l2norm_scale_bf16(const void* x, const void* log_temp, void* y, int M, int C, float eps, void* stream);
int wg_l2norm_scale_bwd_bf16(const void* x, const void* dy, const void* log_temp, void* dx, float* dlog_temp, int M, int C, float eps, void* stream);
int wg_attn_bwd_short_side(int Lq, int Lk);
long wg_attn_bwd_workspace_floats(int B, int heads, int head_dim, int Lq, int Lk);
int wg_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* dout, void* dq_bf16, void* dk_bf16, void* dv_bf16,
                     float* dq_f32, float* dk_f32, float* dv_f32, float* workspace, long workspace_floats, int B, int heads, int head_dim, int Lq, int Lk,
                     float scale, void* stream);
int wg_postprocess_masks_bwd_f32(const float* dout, float* dlow, int N, int low_h, int low_w, int img_size, int in_h, int in_w, int out_h,
                                 int out_w, void* stream);
int wg_mask_losses_bwd_f32(const float* pred_logits, const float* targets, float* dpred, float* workspace, long workspace_floats, int N, long hw,
                           float g_bce, float g_dice, float dice_scale, float dice_eps, void* stream);
/* MSQP's pooling / gate (utils_walkgpt.py:195-217,256-257) and the way back from the language model's input embeddings (the splice of
 * llava_arch.py:265-518 and the resample of :252-259) to the projector's tokens and to embed_tokens (both in train_walkgpt.py's trainable_list):
 *   wg_avgpool_tokens_bwd_bf16 / wg_mean_tokens_bwd_bf16: dx of wg_avgpool_tokens_bf16 / wg_mean_tokens_bf16
 *   wg_sigmoid_gate_bwd_bf16: dx (bf16) and dlogit [rows] (fp32) of y = x * sigmoid(logit)
 *   wg_resample_tokens_bwd_f32: adjoint of wg_resample_tokens_bf16, += into fp32 [n, p*p, C] (zeroed by the caller)
 *   wg_splice_multimodal_bwd_bf16: gradient rows of the spliced embeddings [rows, L+T-1, H] -> image features [rows, T, H] (bf16) and, when
 *     dtable != NULL, += into the embedding table's gradient [V, H] (fp32); img_pos as wg_splice_multimodal_bf16 left it */
int wg_avgpool_tokens_bwd_bf16(const void* dy, void* dx, int B, int H, int W, int C, int s, void* stream);
int wg_mean_tokens_bwd_bf16(const void* dy, void* dx, int B, int L, int C, void* stream);
int wg_sigmoid_gate_bwd_bf16(const void* x, const float* logit, const void* dy, void* dx, float* dlogit, long rows, int C, void* stream);
int wg_resample_tokens_bwd_f32(const void* dy, float* dx, int n, int p, int t, int C, void* stream);
int wg_splice_multimodal_bwd_bf16(const long* ids, const int* img_pos, const void* dembeds, void* dimage_features, float* dtable, int rows, int L,
                                  int T, int H, int V, void* stream);
/* Region-alignment InfoNCE (utils_walkgpt.py:8-73, the top_k form WalkGPT calls: model/walkgpt.py:459-473) as differentiable pieces:
 *   wg_topk_pool_bf16 / _bwd: v_m = sum_k softmax_k(u_m . kt_mk / sqrt(D)) kt_mk over the Kt <= 16 selected SAM tokens kt [M, Kt, D] (constants);
 *     u [M, D] = the folded query W_k^T W_q z of TinyCrossAttn (:330-357); backward gives du.
 *   wg_nce_tail_f32 / _bwd: pos_m = z_m . vp_m, logits_m = [pos_m, sim[m, :]] / T (the rows*N columns of m's own image masked when
 *     exclude_same_row), loss_m = logsumexp - pos_m / T; backward (g = upstream gradient of the MEAN over m): dz (positive term), dvp, dsim. */
/* masks = hyper_in @ upscaled (mask_decoder.py:150-160) on channels-last rows for all prompts at once, and its gradients (training path):
 *   up [P, HW, 32] bf16, hyper [P, K <= 4, 32] bf16 -> masks [P, K, HW] fp32;  backward: dup [P, HW, 32] bf16, dhyper [P, K, 32] fp32 (+=). */
int wg_hyper_rows_f32(const void* up, const void* hyper, float* masks, int P, int HW, int C, int K, void* stream);
long wg_hyper_rows_bwd_workspace_floats(int P, int HW, int K);
int wg_hyper_rows_bwd_f32(const void* up, const void* hyper, const float* dmasks, void* dup, float* dhyper, float* workspace, long workspace_floats, int P, int HW,
                          int C, int K, void* stream);   /* dhyper written (fixed-order sums of per-workgroup partials in `workspace`: no atomics) */
/* fp32 verification route (csrc/fp32_ref.hip; not a product path, not benched): fp32 storage, exact fp32 matrix math on v_mfma_f32_16x16x4_f32,
 * plain kernels.  Exists so that north_star's "text logits within 1e-4 abs of the reference CPU path" can be held on the GPU on SOME route
 * (walkgpt_amd/fp32_route.py, tests/test_gpu_fp32_route.py); the bf16 path's distance to fp32 is set by its bf16 weights.
 *   wg_f32_gemm_bias_act: C = act(A . W^T + bias) (+ R[m % res_row_mod or m]), K % 4 == 0   (HF CLIP linears, patch embedding rows, llava_arch.py:36-42)
 *   wg_f32_layernorm:     biased-variance LayerNorm per row                                 (HF CLIP pre_layrnorm / layer_norm1 / layer_norm2)
 *   wg_f32_mha:           softmax(scale q k^T + key_bias[b, j]) v per head, rows [B, L, ld]  (HF CLIPAttention, custom_clip.py:27-38 mask) */
int wg_f32_gemm_bias_act(const float* A, long lda, const float* W, long ldw, const float* bias, const float* residual, long ldr, int res_row_mod,
                         float* C, long ldc, int M, int N, int K, int act, void* stream);
int wg_f32_layernorm(const float* x, const float* gamma, const float* beta, float* y, long rows, int C, float eps, void* stream);
int wg_f32_mha(const float* q, const float* k, const float* v, float* o, const float* key_bias, long ld, long ldo, int B, int heads, int head_dim,
               int Lq, int Lk, float scale, void* stream);
/* wg_f32_mha_ex: the same with queries [B, Lq, ldq] and keys / values [B, Lk, ldkv] in different tensors (nn.MultiheadAttention of CrossAttnBlock,
 * utils_walkgpt.py:163-185) and an optional additive bias attn_bias [B, heads, Lq, Lk] (SAM's decomposed relative position term, image_encoder.py:321-392,
 * which the caller forms from the UNSCALED queries, :247-249); the fp32 verification route of the SAM encoder -> MSQP -> language model path. */
int wg_f32_mha_ex(const float* q, long ldq, const float* k, const float* v, long ldkv, float* o, long ldo, const float* key_bias, const float* attn_bias,
                  int B, int heads, int head_dim, int Lq, int Lk, float scale, void* stream);
/* The software-pipelined attention kernel (csrc/attn_pipe.hip; head_dim 64): which cases wg_sam_attn_relpos_bf16 / wg_mha_bf16 hand to it.
 * 0 none, 1 (default) SAM global attention on a 64 x 64 grid (image_encoder.py:235-260 with window_size 0), 2 also plain attention without a key
 * bias on whole 64-key tiles.  Returns the previous mode; a negative argument only queries.  Process-wide, not per stream. */
int wg_attn_pipe_mode(int mode);
/* Test support: writes `pattern` over the first 64 KiB of every compute unit's LDS (2048 workgroups; sink: one device word, or NULL).  LDS is not
 * cleared between launches, so a kernel that reads a word it never wrote sees whatever ran before it; tests poison with NaN bits first. */
int wg_debug_fill_lds_u32(unsigned pattern, void* sink, void* stream);
/* The two GEMMs of a Linear's backward pass on the operands as they lie (no transposed copies; csrc/gemm_bwd.hip):
 *   wg_gemm_nn_bf16: dX[M,K] = dY[M,N] . W[N,K];   wg_gemm_tn_bf16: dW[N,K] = dY[M,N]^T . X[M,K] and, if db != null, db[N] = column sums of dY.
 * bf16 operands, fp32 accumulation, results bf16 (out_f32 = 0) or fp32.  A long reduction over few output tiles is split; the fp32 partials go
 * through `workspace` (wg_gemm_bwd_workspace_floats(rows of the result, columns of the result, reduction length, db != null) floats; 0 = none) and
 * are summed in a fixed order (no atomics: the same bits every run).  Every leading dimension, N and K must be multiples of 8. */
int wg_gemm_bwd_splits(int Mo, int Ko, int R);
long wg_gemm_bwd_workspace_floats(int Mo, int Ko, int R, int with_colsum);
int wg_gemm_tn_bf16(const void* dY, long lddy, const void* X, long ldx, void* dW, void* db, int out_f32, float* workspace, long workspace_floats, int M,
                    int N, int K, void* stream);
int wg_gemm_nn_bf16(const void* dY, long lddy, const void* W, long ldw, void* dX, int out_f32, float* workspace, long workspace_floats, int M, int N, int K,
                    void* stream);
int wg_topk_pool_bf16(const void* u, const void* kt, void* v, int M, int Kt, int D, void* stream);
int wg_topk_pool_bwd_bf16(const void* u, const void* kt, const void* dv, void* du, int M, int Kt, int D, void* stream);
/* The same pooling over any number of tokens (top_k > 16; or no top_k at all, utils_walkgpt.py:338-356: TinyCrossAttn's softmax over the whole row,
 * W_v / out applied to the pooled token by the caller): tokens [rows, Kt, D] bf16 constants, query m pools row row_of[m] (null: row m). */
int wg_pool_rows_bf16(const void* u, const void* tokens, const int* row_of, void* v, int M, int Kt, int D, void* stream);
int wg_pool_rows_bwd_bf16(const void* u, const void* tokens, const int* row_of, const void* dv, void* du, int M, int Kt, int D, void* stream);
int wg_nce_tail_f32(const void* z, const void* vp, const float* sim, const int* own_row, float* loss_m, float* lse_m, int M, int rows, int N, int D,
                    float temperature, int exclude_same_row, void* stream);
/* the same two backward entry points with their upstream gradients in DEVICE memory ({g_bce, g_dice} already scaled by 1 / (num_masks + 1e-8); one
 * float for the InfoNCE tail): no host read inside a backward pass, so it can be captured into a graph */
int wg_mask_losses_bwd_dev_f32(const float* pred_logits, const float* targets, float* dpred, float* workspace, long workspace_floats, int N, long hw,
                               const float* g2, float dice_scale, float dice_eps, void* stream);
int wg_nce_tail_bwd_dev_f32(const void* z, const void* vp, const float* sim, const int* own_row, const float* lse_m, const float* g, void* dz, void* dvp,
                            float* dsim, int M, int rows, int N, int D, float temperature, int exclude_same_row, void* stream);
int wg_nce_tail_bwd_f32(const void* z, const void* vp, const float* sim, const int* own_row, const float* lse_m, float g, void* dz, void* dvp,
                        float* dsim, int M, int rows, int N, int D, float temperature, int exclude_same_row, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WALKGPT_HIP_H */
