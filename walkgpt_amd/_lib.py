"""ctypes binding of libwalkgpt_hip.so (the C-ABI declared in include/walkgpt_hip.h).

There is NO fallback: if the shared library is missing or an entry point returns an error, the product path
raises.  (The CPU oracle lives under oracle/ and is only ever imported by tests, smoke() and bench.py's
cpu_baseline leg.)"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WG_LIB") or os.path.join(_HERE, "libwalkgpt_hip.so")   # WG_LIB: A/B against a variant build (tools/build_variant.py)

c_void_p, c_int, c_long, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

# name -> argtypes; every function returns int (0 ok / negative error) unless listed in _SPECIAL.
SIGNATURES = {
    "wg_gemm_bias_act_bf16": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_void_p, c_long,
                              c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "wg_row_stats_bf16": [c_void_p, c_long, c_void_p, c_int, c_int, c_float, c_void_p],
    "wg_gemm_ln_bias_act_bf16": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int,
                                 c_int, c_int, c_void_p],
    "wg_gemm_bias_act_stats_bf16": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_void_p, c_long,
                                    c_int, c_int, c_int, c_int, c_void_p, c_long, c_void_p],
    "wg_gemm_lnp_bias_act_bf16": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_long, c_float, c_void_p, c_long,
                                  c_int, c_int, c_int, c_int, c_void_p],
    "wg_layernorm_rows": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_int,
                          c_void_p],
    "wg_mha_bf16": [c_void_p, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_void_p, c_long, c_long,
                    c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "wg_mha_small_bf16": [c_void_p, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_void_p, c_long,
                          c_long, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "wg_sam_attn_relpos_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                c_float, c_void_p],
    "wg_sam_attn_relpos_mx_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "wg_patchify_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "wg_im2row3x3_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "wg_add_rows_bf16": [c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_long, c_long, c_int, c_void_p],
    "wg_tokens_to_nchw_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "wg_nchw_to_tokens_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "wg_dense_pe_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "wg_cast_f32_to_bf16": [c_void_p, c_void_p, c_long, c_void_p],
    "wg_hyper_mask_dot": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "wg_quantize_rows_fp8": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_int, c_int, c_void_p],
    "wg_layernorm_quantize_fp8": [c_void_p, c_long, c_void_p, c_void_p, c_float, c_void_p, c_long, c_void_p, c_int, c_int, c_void_p],
    "wg_gemm_fp8_bias_act": [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p, c_long,
                             c_int, c_int, c_int, c_int, c_void_p],
    "wg_gemm_mxfp8": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long,
                      c_float, c_void_p, c_long, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long,
                      c_int, c_int, c_int, c_int, c_void_p],
    "wg_quantize_mx_fp8": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_long, c_void_p],
    "wg_colsum_f32": [c_void_p, c_long, c_void_p, c_int, c_int, c_void_p],
    "wg_act_bf16": [c_void_p, c_void_p, c_long, c_int, c_void_p],
    "wg_act_bwd_bf16": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p],
    "wg_layernorm_bwd_bf16": [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "wg_layernorm_bwd_det_bf16": [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_void_p, c_long, c_int, c_int,
                                  c_float, c_void_p],
    "wg_l2norm_scale_bf16": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "wg_l2norm_scale_bwd_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "wg_attn_bwd_short_side": [c_int, c_int],
    "wg_attn_bwd_bf16": [c_void_p] * 12 + [c_long, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "wg_postprocess_masks_bwd_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "wg_mask_losses_bwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_long, c_float, c_float, c_float, c_float, c_void_p],
    "wg_avgpool_tokens_bwd_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "wg_mean_tokens_bwd_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "wg_sigmoid_gate_bwd_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p],
    "wg_resample_tokens_bwd_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "wg_splice_multimodal_bwd_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "wg_gemm_tn_bf16": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_void_p, c_long, c_int, c_int, c_int, c_void_p],
    "wg_gemm_nn_bf16": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p, c_long, c_int, c_int, c_int, c_void_p],
    "wg_topk_pool_bf16": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "wg_topk_pool_bwd_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "wg_pool_rows_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "wg_pool_rows_bwd_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "wg_nce_tail_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p],
    "wg_nce_tail_bwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                            c_int, c_void_p],
    "wg_attn_pipe_mode": [c_int],
    "wg_mask_losses_bwd_dev_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_long, c_void_p, c_float, c_float, c_void_p],
    "wg_nce_tail_bwd_dev_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                c_int, c_void_p],
    "wg_f32_gemm_bias_act": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p],
    "wg_f32_layernorm": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_float, c_void_p],
    "wg_f32_mha": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "wg_f32_mha_ex": [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                      c_void_p],
    "wg_debug_fill_lds_u32": [ctypes.c_uint, c_void_p, c_void_p],
    "wg_hyper_rows_f32": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "wg_hyper_rows_bwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p],
    "wg_colsum_det_f32": [c_void_p, c_long, c_void_p, c_int, c_void_p, c_long, c_int, c_int, c_void_p],
    "wg_upscale_mask_bf16": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "wg_dec_tokens_f32": [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                          c_int, c_float, c_void_p],
    "wg_dec_tokens_ctp_f32": [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p, c_int,
                              c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p],
    "wg_dec_attn_partial_f32": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_long, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p],
    "wg_dec_mlp_slices": [],
    "wg_dec_mlp_partial_f32": [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "wg_gemm_skinny_ln_supported": [c_int, c_int, c_int, c_long, c_long, c_long],
    "wg_gemm_skinny_ln_bias_act_bf16": [c_void_p, c_long, c_void_p, c_void_p, c_float, c_void_p, c_long, c_int, c_void_p, c_void_p, c_long, c_int,
                                        c_int, c_int, c_int, c_int, c_void_p],
    "wg_tile_weight_bf16": [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p],
    "wg_dec_heads_f32": [c_void_p, c_void_p, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "wg_dec_i2t_rows_bf16": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float,
                             c_void_p, c_int, c_int, c_void_p],
    "wg_postprocess_masks_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "wg_postprocess_masks_score_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p],
    "wg_postprocess_masks_score_fused_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                             c_int, c_void_p],
    "wg_mask_score_f32": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_long, c_void_p],
    "wg_mask_iou_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_long, c_float, c_void_p],
    "wg_mask_losses_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_long, c_float, c_float, c_void_p],
    "wg_splice_multimodal_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_void_p],
    "wg_preprocess_frames_u8": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "wg_match_cost_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "wg_row_inv_norm_bf16": [c_void_p, c_long, c_void_p, c_long, c_int, c_float, c_void_p],
    "wg_l2_normalize_rows_bf16": [c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_float, c_void_p],
    "wg_nce_attn_f32": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                        c_float, c_void_p],
    "wg_nce_loss_f32": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                        c_int, c_int, c_int, c_float, c_void_p],
    "wg_avgpool_tokens_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "wg_mean_tokens_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "wg_sigmoid_gate_bf16": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p],
    "wg_ctp_tail_bf16": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int,
                         c_float, c_void_p],
    "wg_resample_tokens_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
}
_SPECIAL = {"wg_last_error": (ctypes.c_char_p, []), "wg_version": (c_int, []),
            "wg_mask_score_workspace_floats": (c_long, [c_int, c_long]),
            "wg_postprocess_score_workspace_floats": (c_long, [c_int, c_int, c_int]),
            "wg_mask_stats_workspace_floats": (c_long, [c_int, c_long]),
            "wg_match_cost_workspace_floats": (c_long, [c_int, c_int, c_int]),
            "wg_gemm_bwd_splits": (c_int, [c_int, c_int, c_int]),
            "wg_gemm_bwd_workspace_floats": (c_long, [c_int, c_int, c_int, c_int]),
            "wg_layernorm_bwd_det_workspace_floats": (c_long, [c_int, c_int]),
            "wg_hyper_rows_bwd_workspace_floats": (c_long, [c_int, c_int, c_int]),
            "wg_colsum_det_workspace_floats": (c_long, [c_int, c_int]),
            "wg_attn_bwd_workspace_floats": (c_long, [c_int, c_int, c_int, c_int, c_int]),
            "wg_sam_attn_mx_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
            "wg_gemm_pick_tile": (c_int, [c_int, c_int]),
            "wg_gemm_ln_supported": (c_int, [c_int, c_int, c_int, c_long, c_long, c_long]),
            "wg_gemm_row_partials_supported": (c_int, [c_int, c_int, c_int, c_long, c_long, c_long]),
            "wg_gemm_pick_tile_ex": (c_int, [c_int, c_int, c_int]),
            "wg_gemm_pick_tile_mnk": (c_int, [c_int, c_int, c_int, c_int])}


class WalkgptHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises if the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WalkgptHipError(
            "libwalkgpt_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python walkgpt_amd/_build.py`. There is no CPU fallback." % LIB_PATH)
    # torch must load its HIP runtime first: libwalkgpt_hip.so's libamdhip64 dependency then resolves to that same
    # runtime instance (streams and device pointers are shared with torch).  Loaded the other way round the process
    # ends up with two runtimes and launches fail with "no ROCm-capable device".
    import torch  # noqa: F401
    h = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(h, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    for name, (res, argtypes) in _SPECIAL.items():
        fn = getattr(h, name)
        fn.argtypes = argtypes
        fn.restype = res
    _lib = h
    return h


def exported_symbols():
    return list(SIGNATURES) + list(_SPECIAL)


def check(rc, what):
    if rc != 0:
        msg = lib().wg_last_error()
        raise WalkgptHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
