"""Thin tensor-level wrappers over the C-ABI.  torch is plumbing only here: it owns the HBM buffers and the
stream; every arithmetic step is a wg_* call.  All wrappers raise on non-GPU tensors -- there is no CPU path."""
import torch

from . import _lib

ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_RELU = 0, 1, 2, 3
_BF16 = torch.bfloat16


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream's handle.  (torch.cuda.current_stream() builds a Stream object through three Python layers, ~4 us a call and one call per
    launch: a third of a millisecond of a 590-launch training step; the raw accessor is the same handle in ~0.3 us.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.WalkgptHipError("walkgpt_amd ops need GPU (HIP) tensors; got a %s tensor" % t.device)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _rows(t):
    """View a [..., D] tensor as rows: returns (rows, D, leading dimension)."""
    assert t.stride(-1) == 1, "innermost dimension must be contiguous"
    if t.dim() == 1:
        return 1, t.shape[0], t.shape[0]
    if t.dim() == 2:
        return t.shape[0], t.shape[1], t.stride(0)
    assert t.is_contiguous(), "tensors with more than two dimensions must be contiguous"
    return t.numel() // t.shape[-1], t.shape[-1], t.shape[-1]


import contextlib as _contextlib
import os as _os
import threading as _threading

_FORCE_TILE = int(_os.environ.get("WG_GEMM_TILE", "0"))  # experiments only: force one GEMM tile variant
_tls = _threading.local()  # per-thread measurement state (no module-level mutable switches: two models / threads never interact)


@_contextlib.contextmanager
def time_gemms():
    """Measurement only (bench.py): inside the block every GEMM launch of THIS thread is bracketed by a pair of HIP events on its
    launch stream.  Yields the list that collects (kernel id, M, N, K, start_event, end_event); 17 = the LayerNorm-folded instance."""
    records = []
    prev = getattr(_tls, "gemm_records", None)
    _tls.gemm_records = records
    try:
        yield records
    finally:
        _tls.gemm_records = prev


def _timed(kid, M, N, K):
    rec = getattr(_tls, "gemm_records", None)
    if rec is None:
        return None
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    rec.append((kid, M, N, K, ev[0], ev[1]))
    return ev


def gemm_tile_for(M, N, K, lda, ldw, ldc, ldr, tail_tiles=False):
    """Which kernel wg_gemm_bias_act_bf16 runs for this shape: 1 = 128x128 tiles, 2 = 256x256 tiles,
    3 = row-wave (asks the library's own selector, so host-side accounting and the library agree).
    tail_tiles: also consider the tail-absorbing 128x128 kernel (wg_gemm_pick_tile_ex in csrc/gemm.hip; single-stream callers)."""
    ok = K % 64 == 0 and N % 4 == 0 and N >= 16 and lda % 8 == 0 and ldw % 8 == 0 and ldc % 4 == 0 and ldr % 4 == 0
    if not ok:
        return 3
    if _FORCE_TILE:
        return _FORCE_TILE
    return _lib.lib().wg_gemm_pick_tile_mnk(M, N, K, 1 if tail_tiles else 0)


ROW_PARTIALS = _os.environ.get("WG_ROW_PARTIALS", "1") != "0"   # experiments: 0 = always take the row statistics in their own pass


def _forget_sidecars(t):
    """A caller-supplied `out=` tensor is about to be overwritten through its raw pointer (tensor._version does not move): whatever an
    earlier producer attached to it -- row partial sums, an e4m3 + block-scale copy -- no longer describes its contents."""
    if t is not None:
        for a in ("_wg_row_partials", "_wg_mx"):
            if hasattr(t, a):
                delattr(t, a)


def linear(x, weight, bias=None, act=ACT_NONE, residual=None, res_row_mod=0, out=None, out_f32=False, tile=0, tail_tiles=False,
           row_partials=False):
    """y = act(x @ weight.T + bias) (+ residual).  x [..., K] bf16, weight [N, K] bf16.
    row_partials: y feeds a LayerNorm that ln_linear folds into the next GEMM -- where the shape allows it the GEMM also leaves the
    per-row partial sums of y (wg_gemm_bias_act_stats_bf16) and ln_linear(y, ...) picks them up instead of running a statistics
    pass (they ride on the returned tensor object: a view or an in-place edit of y silently falls back to the pass)."""
    _need_gpu(x, weight, bias, residual, out)
    assert x.dtype == _BF16 and weight.dtype == _BF16
    M, K, lda = _rows(x)
    N, K2 = weight.shape
    assert K2 == K and weight.stride(1) == 1
    _forget_sidecars(out)
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32 if out_f32 else _BF16)
    Mo, No, ldc = _rows(out)
    assert Mo == M and No == N
    ldr = 0
    if residual is not None:
        assert residual.dtype == _BF16
        _, Nr, ldr = _rows(residual)
        assert Nr == N
    if bias is not None:
        assert bias.dtype == _BF16 and bias.numel() == N
    if (row_partials and ROW_PARTIALS and not out_f32 and tile == 0 and not _FORCE_TILE and not tail_tiles
            and _lib.lib().wg_gemm_row_partials_supported(M, N, K, lda, weight.stride(0), ldc)
            and (bias is None or bias.data_ptr() % 16 == 0) and (residual is None or (ldr % 8 == 0 and residual.data_ptr() % 16 == 0))):
        mpad = (M + 255) // 256 * 256
        part = torch.empty(N // 256, mpad, 2, device=x.device, dtype=torch.float32)
        ev = _timed(18, M, N, K)  # 18: the statistics-producing instance of kernel 16
        if ev is not None:
            ev[0].record()
        rc = _lib.lib().wg_gemm_bias_act_stats_bf16(x.data_ptr(), lda, weight.data_ptr(), weight.stride(0), _ptr(bias), _ptr(residual), ldr,
                                                    res_row_mod, out.data_ptr(), ldc, M, N, K, act, part.data_ptr(), mpad, _stream())
        if ev is not None:
            ev[1].record()
        _lib.check(rc, "wg_gemm_bias_act_stats_bf16")
        out._wg_row_partials = (part, mpad, out._version, M, N)
        return out
    if tile == 0:
        tile = gemm_tile_for(M, N, K, lda, weight.stride(0), ldc, ldr, tail_tiles)
    ev = _timed(tile, M, N, K)
    if ev is not None:
        ev[0].record()
    rc = _lib.lib().wg_gemm_bias_act_bf16(x.data_ptr(), lda, weight.data_ptr(), weight.stride(0), _ptr(bias),
                                          _ptr(residual), ldr, res_row_mod, out.data_ptr(), ldc, M, N, K, act,
                                          1 if out.dtype == torch.float32 else 0, tile, _stream())
    if ev is not None:
        ev[1].record()
    _lib.check(rc, "wg_gemm_bias_act_bf16")
    return out


def layernorm(x, gamma, beta, eps, act=ACT_NONE, out=None):
    _need_gpu(x, gamma, beta, out)
    assert x.dtype == _BF16 and gamma.dtype == _BF16 and beta.dtype == _BF16
    M, D, ldx = _rows(x)
    _forget_sidecars(out)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=_BF16)
    _, _, ldy = _rows(out)
    rc = _lib.lib().wg_layernorm_rows(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), ldy, M, D,
                                      float(eps), act, _stream())
    _lib.check(rc, "wg_layernorm_rows")
    return out


LN_FUSE = _os.environ.get("WG_LN_FUSE", "1") != "0"  # experiments: 0 = always run LayerNorm and GEMM as two kernels


def fold_layernorm(gamma, beta, weight, bias):
    """Operands of ln_linear: the LayerNorm's affine map folded into the linear layer that consumes it.
    y = LN(x) W^T + b = rstd (x Wg^T - mean s) + b'  with  Wg = bf16(W gamma), s = row sums of Wg (fp32, of the ROUNDED
    Wg, so the mean term cancels exactly what the MFMA accumulates), b' = b + W beta (fp32)."""
    wf = weight.float()
    wg = (wf * gamma.float()[None, :]).to(_BF16).contiguous()
    b = wf @ beta.float()
    if bias is not None:
        b = b + bias.float()
    return {"wg": wg, "colsum": wg.float().sum(1).contiguous(), "bias_f32": b.contiguous(),
            "gamma": gamma, "beta": beta, "weight": weight, "bias": bias}


def row_stats(x, eps):
    """(mean, 1/sqrt(var + eps)) per row of x [..., D] bf16 -> fp32 [rows rounded up to even, 2]."""
    _need_gpu(x)
    M, D, ldx = _rows(x)
    st = torch.empty((M + 1) // 2 * 2, 2, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().wg_row_stats_bf16(x.data_ptr(), ldx, st.data_ptr(), M, D, float(eps), _stream()), "wg_row_stats_bf16")
    return st


def ln_linear(x, fold, eps, act=ACT_NONE, tail_tiles=False):
    """act(LayerNorm(x) @ W.T + b) with `fold` from fold_layernorm.  Shapes the persistent 256x256 GEMM takes run as a
    row-statistics pass + one GEMM on the raw rows (the normalised rows never reach HBM); the rest as LayerNorm + GEMM."""
    _need_gpu(x)
    assert x.dtype == _BF16
    M, K, lda = _rows(x)
    N = fold["wg"].shape[0]
    L = _lib.lib()
    if not (LN_FUSE and not _FORCE_TILE and L.wg_gemm_ln_supported(M, N, K, lda, K, N)):
        return linear(layernorm(x, fold["gamma"], fold["beta"], eps), fold["weight"], fold["bias"], act=act, tail_tiles=tail_tiles)
    out = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=_BF16)
    rp = getattr(x, "_wg_row_partials", None)
    if rp is not None and rp[2] == x._version and rp[3] == M and rp[4] == K and ROW_PARTIALS:
        # the GEMM that wrote x left its rows' partial sums (linear(..., row_partials=True)): no statistics pass
        part, mpad = rp[0], rp[1]
        ev = _timed(17, M, N, K)
        if ev is not None:
            ev[0].record()
        rc = L.wg_gemm_lnp_bias_act_bf16(x.data_ptr(), lda, fold["wg"].data_ptr(), K, fold["bias_f32"].data_ptr(), fold["colsum"].data_ptr(),
                                         part.data_ptr(), part.shape[0], mpad, float(eps), out.data_ptr(), N, M, N, K, act, _stream())
        if ev is not None:
            ev[1].record()
        _lib.check(rc, "wg_gemm_lnp_bias_act_bf16")
        return out
    st = row_stats(x, eps)
    ev = _timed(17, M, N, K)  # 17: the LayerNorm-folded instance of kernel 16
    if ev is not None:
        ev[0].record()
    rc = L.wg_gemm_ln_bias_act_bf16(x.data_ptr(), lda, fold["wg"].data_ptr(), K, fold["bias_f32"].data_ptr(),
                                    fold["colsum"].data_ptr(), st.data_ptr(), out.data_ptr(), N, M, N, K, act, _stream())
    if ev is not None:
        ev[1].record()
    _lib.check(rc, "wg_gemm_ln_bias_act_bf16")
    return out


def layernorm_linear(x, gamma, beta, eps, weight, bias=None, act=ACT_NONE, out_f32=False, weight_tiled=None):
    """act(LayerNorm(x) @ weight.T + bias) (gamma None: no LayerNorm) for a handful of rows (M <= 128: the [SEG] hidden states into
    text_hidden_fcs) in one launch (wg_gemm_skinny_ln_bias_act_bf16); other shapes run LayerNorm and the GEMM as two kernels.
    weight_tiled: tile_weight(weight), read instead of `weight` by the one-launch form."""
    _need_gpu(x, gamma, beta, weight, bias, weight_tiled)
    assert x.dtype == _BF16 and weight.dtype == _BF16
    M, K, lda = _rows(x)
    N = weight.shape[0]
    L = _lib.lib()
    if _FORCE_TILE or not L.wg_gemm_skinny_ln_supported(M, N, K, lda, weight.stride(0), N):
        return linear(x if gamma is None else layernorm(x, gamma, beta, eps), weight, bias, act=act, out_f32=out_f32)
    assert weight.shape[1] == K and weight.stride(1) == 1
    assert gamma is None or (gamma.dtype == _BF16 and beta.dtype == _BF16 and gamma.numel() == K and beta.numel() == K)
    assert weight_tiled is None or _tiled_ok(weight_tiled, N, K)
    out = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32 if out_f32 else _BF16)
    ev = _timed(5, M, N, K)
    if ev is not None:
        ev[0].record()
    w = weight if weight_tiled is None else weight_tiled
    rc = L.wg_gemm_skinny_ln_bias_act_bf16(x.data_ptr(), lda, _ptr(gamma), _ptr(beta), float(eps), w.data_ptr(), weight.stride(0),
                                           0 if weight_tiled is None else 1, _ptr(bias), out.data_ptr(), N, M, N, K, act,
                                           1 if out_f32 else 0, _stream())
    if ev is not None:
        ev[1].record()
    _lib.check(rc, "wg_gemm_skinny_ln_bias_act_bf16")
    return out


def _rows_per_batch(t):
    """[B, L, D] (contiguous rows, possibly a column slice of a wider buffer) -> (B, L, ld, rows per batch)."""
    assert t.dim() == 3 and t.stride(2) == 1
    B, L, _ = t.shape
    ld = t.stride(1)
    assert t.stride(0) % ld == 0 or B == 1
    return B, L, ld, (t.stride(0) // ld if B > 1 else L)


def mha(q, k, v, heads, scale, key_bias=None, out=None, small=None):
    """softmax(scale * q k^T + key_bias) v per head.  q [B,Lq,D], k/v [B,Lk,D] (bf16, may be column slices of a
    packed buffer); key_bias [B,Lk] fp32 additive or None.  `small` forces / forbids the one-wave-per-query kernel."""
    _need_gpu(q, k, v, key_bias, out)
    B, Lq, ldq, qbs = _rows_per_batch(q)
    _, Lk, ldk, kbs = _rows_per_batch(k)
    _, _, ldv, vbs = _rows_per_batch(v)
    assert kbs == vbs
    D = q.shape[2]
    hd = D // heads
    _forget_sidecars(out)
    if out is None:
        out = torch.empty(B, Lq, D, device=q.device, dtype=_BF16)
    _, _, ldo, obs = _rows_per_batch(out)
    if small is None:
        small = hd < 32 or Lq <= 32 or Lk <= 32
    if small and key_bias is None and Lq <= 32 and Lk >= 256 and hd in (16, 32, 64, 128):
        small = False  # few queries, many keys: one MFMA wave per (batch, head) reads K/V once for all queries
    if small:
        assert key_bias is None
        rc = _lib.lib().wg_mha_small_bf16(q.data_ptr(), ldq, qbs, k.data_ptr(), ldk, v.data_ptr(), ldv, kbs,
                                          out.data_ptr(), ldo, obs, B, heads, hd, Lq, Lk, float(scale), _stream())
        _lib.check(rc, "wg_mha_small_bf16")
    else:
        if key_bias is not None:
            assert key_bias.dtype == torch.float32 and key_bias.is_contiguous() and key_bias.shape == (B, Lk)
        rc = _lib.lib().wg_mha_bf16(q.data_ptr(), ldq, qbs, k.data_ptr(), ldk, v.data_ptr(), ldv, kbs, out.data_ptr(),
                                    ldo, obs, _ptr(key_bias), B, heads, hd, Lq, Lk, float(scale), _stream())
        _lib.check(rc, "wg_mha_bf16")
    return out


def sam_attention(qkv, qkv_bias, rel_pos_h, rel_pos_w, B, grid, window, heads, out=None, mx_out=False):
    """SAM ViT attention over packed qkv rows [B*grid*grid, 3D]; window == grid means global attention.
    mx_out: the fp8 chain's form -- where the kernels carry it (head_dim 64; window 14 or the 64 x 64 global grid) the result is the tuple
    (e4m3 bytes, E8M0 block scales) quantize_mx_fp8 would make of the bf16 output, written by the attention kernel itself; elsewhere the bf16
    tensor as usual (the caller quantises)."""
    _need_gpu(qkv, qkv_bias, rel_pos_h, rel_pos_w, out)
    D = qkv.shape[-1] // 3
    hd = D // heads
    assert qkv.is_contiguous() and qkv.shape[0] == B * grid * grid and qkv.dtype == _BF16
    assert rel_pos_h.shape == (2 * window - 1, hd) and rel_pos_h.is_contiguous() and rel_pos_w.is_contiguous()
    if mx_out and out is None and _lib.lib().wg_sam_attn_mx_supported(B, grid, window, heads, hd):
        M = B * grid * grid
        q = torch.empty(M, D, device=qkv.device, dtype=torch.uint8)
        mx = torch.empty(D // 32, mx_pitch(M), device=qkv.device, dtype=torch.uint8)
        rc = _lib.lib().wg_sam_attn_relpos_mx_bf16(qkv.data_ptr(), qkv_bias.data_ptr(), rel_pos_h.data_ptr(), rel_pos_w.data_ptr(), q.data_ptr(), mx.data_ptr(),
                                                   mx.shape[1], B, grid, window, heads, hd, float(hd) ** -0.5, _stream())
        _lib.check(rc, "wg_sam_attn_relpos_mx_bf16")
        return q, mx
    _forget_sidecars(out)
    if out is None:
        out = torch.empty(B * grid * grid, D, device=qkv.device, dtype=_BF16)
    rc = _lib.lib().wg_sam_attn_relpos_bf16(qkv.data_ptr(), qkv_bias.data_ptr(), rel_pos_h.data_ptr(),
                                            rel_pos_w.data_ptr(), out.data_ptr(), B, grid, window, heads, hd,
                                            float(hd) ** -0.5, _stream())
    _lib.check(rc, "wg_sam_attn_relpos_bf16")
    return out


def patchify(images, patch, kpad=None):
    """NCHW bf16 images -> [B*gh*gw, kpad] rows in conv-weight column order, zero padded to kpad columns."""
    _need_gpu(images)
    assert images.dtype == _BF16 and images.is_contiguous()
    B, C, H, W = images.shape
    K = C * patch * patch
    kpad = kpad or ((K + 63) // 64) * 64
    rows = torch.empty(B * (H // patch) * (W // patch), kpad, device=images.device, dtype=_BF16)
    rc = _lib.lib().wg_patchify_bf16(images.data_ptr(), rows.data_ptr(), B, C, H, W, patch, kpad, _stream())
    _lib.check(rc, "wg_patchify_bf16")
    return rows


def im2row3x3(x, B, H, W):
    _need_gpu(x)
    C = x.shape[-1]
    assert x.is_contiguous() and x.numel() == B * H * W * C
    rows = torch.empty(B * H * W, 9 * C, device=x.device, dtype=_BF16)
    rc = _lib.lib().wg_im2row3x3_bf16(x.data_ptr(), rows.data_ptr(), B, H, W, C, _stream())
    _lib.check(rc, "wg_im2row3x3_bf16")
    return rows


def add_rows(a, b, out=None):
    """a [..., C] + b [rb, C] broadcast by row index modulo rb."""
    _need_gpu(a, b, out)
    rows, cols, lda = _rows(a)
    rb, cb, ldb = _rows(b)
    assert cb == cols
    _forget_sidecars(out)
    if out is None:
        out = torch.empty(a.shape, device=a.device, dtype=_BF16)
    _, _, ldo = _rows(out)
    rc = _lib.lib().wg_add_rows_bf16(a.data_ptr(), lda, b.data_ptr(), ldb, rb, out.data_ptr(), ldo, rows, cols, _stream())
    _lib.check(rc, "wg_add_rows_bf16")
    return out


def tokens_to_nchw(x, B, HW, C):
    _need_gpu(x)
    assert x.is_contiguous()
    y = torch.empty(B, C, HW, device=x.device, dtype=_BF16)
    _lib.check(_lib.lib().wg_tokens_to_nchw_bf16(x.data_ptr(), y.data_ptr(), B, HW, C, _stream()), "wg_tokens_to_nchw_bf16")
    return y


def nchw_to_tokens(x):
    _need_gpu(x)
    assert x.is_contiguous() and x.dtype == _BF16
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty(B, HW, C, device=x.device, dtype=_BF16)
    _lib.check(_lib.lib().wg_nchw_to_tokens_bf16(x.data_ptr(), y.data_ptr(), B, HW, C, _stream()), "wg_nchw_to_tokens_bf16")
    return y


def dense_pe_tokens(gaussian, h, w):
    """[h*w, 2F] fp32 positional encoding rows from the [2, F] fp32 gaussian matrix."""
    _need_gpu(gaussian)
    g = gaussian.float().contiguous()
    F_ = g.shape[1]
    pe = torch.empty(h * w, 2 * F_, device=g.device, dtype=torch.float32)
    _lib.check(_lib.lib().wg_dense_pe_f32(g.data_ptr(), pe.data_ptr(), h, w, F_, _stream()), "wg_dense_pe_f32")
    return pe


def cast_bf16(x):
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=_BF16)
    _lib.check(_lib.lib().wg_cast_f32_to_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "wg_cast_f32_to_bf16")
    return y


def hyper_mask_dot(up, hyper, T, h, w, first_mask, num_masks):
    """up: pixel-shuffled upscaled embedding rows [T*h*w*4, 4*32]; hyper [T, nmask, 32] -> fp32 [T, num_masks, 4h, 4w]."""
    _need_gpu(up, hyper)
    assert up.is_contiguous() and hyper.is_contiguous() and up.shape == (T * h * w * 4, 128)
    masks = torch.empty(T, num_masks, 4 * h, 4 * w, device=up.device, dtype=torch.float32)
    rc = _lib.lib().wg_hyper_mask_dot(up.data_ptr(), hyper.data_ptr(), masks.data_ptr(), T, h, w, 32, hyper.shape[1],
                                      first_mask, num_masks, _stream())
    _lib.check(rc, "wg_hyper_mask_dot")
    return masks


def quantize_rows_fp8(x, ln=None, eps=0.0):
    """Per-row e4m3 quantisation of x [..., K] bf16 (of LayerNorm(x) when ln = (gamma, beta)) -> (q uint8 [..., K], scale fp32 [rows])."""
    _need_gpu(x)
    assert x.dtype == _BF16
    M, K, ldx = _rows(x)
    q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    scale = torch.empty(M, device=x.device, dtype=torch.float32)
    L = _lib.lib()
    if ln is None:
        rc = L.wg_quantize_rows_fp8(x.data_ptr(), ldx, q.data_ptr(), K, scale.data_ptr(), M, K, _stream())
    else:
        g, b = ln
        _need_gpu(g, b)
        rc = L.wg_layernorm_quantize_fp8(x.data_ptr(), ldx, g.data_ptr(), b.data_ptr(), float(eps), q.data_ptr(), K, scale.data_ptr(), M, K, _stream())
    _lib.check(rc, "wg_quantize_rows_fp8")
    return q, scale


def quantize_weight_fp8(weight):
    """nn.Linear weight [N, K] bf16 -> (e4m3 bytes [N, K], per-output-channel scale fp32 [N]); once per weight set."""
    return quantize_rows_fp8(weight.contiguous())


def mx_pitch(M):
    """Row pitch (bytes) of an MX block-scale plane for M rows: whole 256-row tiles."""
    return (M + 255) // 256 * 256


def mx_scale_index(M, device=None, group=128):
    """Position of row m inside a block-scale plane (include/walkgpt_hip.h: wg_gemm_mxfp8): rows of a `group`-row group are
    permuted to (r % 16) * (group / 16) + r // 16 (128: activations, 64: weights).  For tests and tools; the kernels never materialise it."""
    m = torch.arange(M, device=device)
    r = m % group
    return (m - r) + (r % 16) * (group // 16) + r // 16


def quantize_mx_fp8(x, group=128, row_partials=False):
    """x [..., K] bf16 -> (e4m3 bytes uint8 [..., K], E8M0 block scales uint8 [K / 32, mx_pitch(rows)]): OCP-MX, one power-of-two scale per
    (row, 32 columns).  group=128: an activation operand of linear_mxfp8; group=64: its weight operand (once per weight).
    row_partials (K % 256 == 0): the same pass also attaches x's row partial sums (x._wg_row_partials), as a GEMM epilogue would have."""
    _need_gpu(x)
    assert x.dtype == _BF16
    M, K, ldx = _rows(x)
    q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    mx = torch.empty(K // 32, mx_pitch(M), device=x.device, dtype=torch.uint8)
    part = torch.empty(K // 256, mx_pitch(M), 2, device=x.device, dtype=torch.float32) if row_partials else None
    rc = _lib.lib().wg_quantize_mx_fp8(x.data_ptr(), ldx, q.data_ptr(), K, mx.data_ptr(), mx.shape[1], group, M, K, _ptr(part),
                                       part.shape[1] if row_partials else 0, _stream())
    _lib.check(rc, "wg_quantize_mx_fp8")
    if row_partials:
        x._wg_row_partials = (part, part.shape[1], x._version, M, K)
    return q, mx


def mx_dequantize(q, mx, group=128):
    """(e4m3 bytes [rows, K], block scales [K / 32, pitch]) -> fp32 values (host-side helper: weight folds, tests)."""
    rows, K = q.shape
    e = mx[:, mx_scale_index(rows, mx.device, group)].float() - 127.0          # [K/32, rows]
    return (q.view(torch.float8_e4m3fn).float().view(rows, K // 32, 32) * torch.exp2(e).t()[..., None]).view(rows, K)


def mx_weight(weight):
    """nn.Linear weight [N, K] bf16 -> {"q", "mx"}: the W operand of linear_mxfp8 (once per weight set)."""
    q, mx = quantize_mx_fp8(weight.contiguous(), group=64)
    return {"q": q, "mx": mx}


def fold_layernorm_mx(gamma, beta, weight, bias):
    """fold_layernorm for the fp8 path: Wg = W gamma quantised to e4m3 with MX block scales, s = row sums of the DEQUANTISED Wg (so the
    mean term cancels exactly what the block-scaled MFMA accumulates), b' = b + W beta (fp32, from the unquantised weight)."""
    wf = weight.float()
    q, mx = quantize_mx_fp8((wf * gamma.float()[None, :]).to(_BF16).contiguous(), group=64)
    b = wf @ beta.float()
    if bias is not None:
        b = b + bias.float()
    return {"q": q, "mx": mx, "colsum": mx_dequantize(q, mx, 64).sum(1).contiguous(), "bias_f32": b.contiguous()}


def mx_chain_ok(D, hidden):
    """Can a transformer block of width D / MLP width `hidden` run its four linears as one MX chain on the persistent fp8 GEMM?  (LayerNorm
    fold: the row statistics arrive as D / 256 partial planes, at most 5; fp8 copies need whole 32-column blocks.)"""
    return D % 256 == 0 and D <= 1280 and hidden % 32 == 0 and hidden % 128 == 0


def mx_prepare_rows(x):
    """Give a bf16 activation tensor [.., D] (D % 256 == 0) the by-products an MX-chain consumer (linear_mxfp8(ln_eps=...)) expects: its
    e4m3 + block-scale form and its rows' partial sums.  Inside the chain the producing GEMM's epilogue leaves both; at the chain's
    entry one quantisation pass makes what is missing."""
    rp = getattr(x, "_wg_row_partials", None)
    have_rp = rp is not None and rp[2] == x._version
    if getattr(x, "_wg_mx", None) is None or not have_rp:
        x._wg_mx = quantize_mx_fp8(x, row_partials=not have_rp)
    return True


def linear_mxfp8(x, w, bias=None, act=ACT_NONE, residual=None, res_row_mod=0, out=None, ln_eps=None, mx_out=False, bf16_out=True,
                 row_partials=False):
    """act(dequant(x) @ dequant(w).T + bias) (+ residual) on the persistent fp8 GEMM with MX block scales on both operands.
    x: (xq uint8 [..., K], x_mx) from quantize_mx_fp8 / an earlier call's mx_out -- or, with ln_eps set, the bf16 tensor such a call
       returned (it carries its fp8 form and its rows' partial sums): LayerNorm(x) is then folded in and w = fold_layernorm_mx(...);
    w: mx_weight(...) / fold_layernorm_mx(...).
    mx_out: the result also leaves as e4m3 + block scales (attached to the returned tensor as ._wg_mx; with bf16_out=False ONLY so, and
    the pair is what is returned); row_partials: its rows' {sum, sum of squares} partials are attached as ._wg_row_partials (N % 256 == 0)."""
    ln = None
    if ln_eps is not None:
        xt = x
        rp, xm = getattr(xt, "_wg_row_partials", None), getattr(xt, "_wg_mx", None)
        assert rp is not None and xm is not None and rp[2] == xt._version, "linear_mxfp8(ln_eps=...): x must come from a call with mx_out and row_partials"
        x = xm
        ln = (rp[0], rp[1])
    xq, x_mx = x
    wq, w_mx = w["q"], w["mx"]
    _need_gpu(xq, x_mx, wq, w_mx, bias, residual, out)
    assert xq.dtype == torch.uint8 and wq.dtype == torch.uint8 and xq.is_contiguous() and wq.is_contiguous()
    assert x_mx.dtype == torch.uint8 and w_mx.dtype == torch.uint8 and x_mx.is_contiguous() and w_mx.is_contiguous()
    K = xq.shape[-1]
    M = xq.numel() // K
    N = wq.shape[0]
    assert wq.shape[1] == K and x_mx.shape[0] == K // 32 and w_mx.shape[0] == K // 32
    dev = xq.device
    assert bf16_out or mx_out
    _forget_sidecars(out)
    if out is None and bf16_out:
        out = torch.empty(xq.shape[:-1] + (N,), device=dev, dtype=_BF16)
    ldc = _rows(out)[2] if out is not None else 0
    ldr = 0
    if residual is not None:
        assert residual.dtype == _BF16
        _, _, ldr = _rows(residual)
    cq = cmx = part = None
    if mx_out:
        cq = torch.empty(xq.shape[:-1] + (N,), device=dev, dtype=torch.uint8)
        cmx = torch.empty(N // 32, mx_pitch(M), device=dev, dtype=torch.uint8)
    if row_partials:
        assert N % 256 == 0 and ln is None
        part = torch.empty(N // 256, mx_pitch(M), 2, device=dev, dtype=torch.float32)
    ev = _timed(22 if ln is not None else (23 if row_partials else 21), M, N, K)   # (ids: bench.py KERNEL_NAMES)
    if ev is not None:
        ev[0].record()
    rc = _lib.lib().wg_gemm_mxfp8(xq.data_ptr(), K, x_mx.data_ptr(), x_mx.shape[1], wq.data_ptr(), K, w_mx.data_ptr(), w_mx.shape[1],
                                  None if ln is not None else _ptr(bias),
                                  w["colsum"].data_ptr() if ln is not None else None, w["bias_f32"].data_ptr() if ln is not None else None,
                                  ln[0].data_ptr() if ln is not None else None, ln[0].shape[0] if ln is not None else 0,
                                  ln[1] if ln is not None else 0, float(ln_eps) if ln is not None else 0.0,
                                  _ptr(residual), ldr, res_row_mod, _ptr(out), ldc, _ptr(cq), N, _ptr(cmx), cmx.shape[1] if mx_out else 0,
                                  _ptr(part), part.shape[1] if row_partials else 0, M, N, K, act, _stream())
    if ev is not None:
        ev[1].record()
    _lib.check(rc, "wg_gemm_mxfp8")
    if not bf16_out:
        return cq, cmx
    if mx_out:
        out._wg_mx = (cq, cmx)
    if row_partials:
        out._wg_row_partials = (part, part.shape[1], out._version, M, N)
    return out


def linear_fp8(xq, x_scale, wq, w_scale, bias=None, act=ACT_NONE, residual=None, res_row_mod=0, out=None):
    """act(dequant(xq) @ dequant(wq).T + bias) (+ residual) -> bf16.  xq [..., K] uint8 (e4m3) with per-row scales, wq [N, K] with
    per-output-channel scales (quantize_rows_fp8 / quantize_weight_fp8).  The form for widths the MX chain (linear_mxfp8) does not take."""
    _need_gpu(xq, x_scale, wq, w_scale, bias, residual, out)
    assert xq.dtype == torch.uint8 and wq.dtype == torch.uint8 and xq.is_contiguous() and wq.is_contiguous()
    K = xq.shape[-1]
    M = xq.numel() // K
    N = wq.shape[0]
    assert wq.shape[1] == K and x_scale.numel() == M and w_scale.numel() == N
    _forget_sidecars(out)
    if out is None:
        out = torch.empty(xq.shape[:-1] + (N,), device=xq.device, dtype=_BF16)
    _, _, ldc = _rows(out)
    ldr = 0
    if residual is not None:
        assert residual.dtype == _BF16
        _, _, ldr = _rows(residual)
    ev = _timed(20, M, N, K)
    if ev is not None:
        ev[0].record()
    rc = _lib.lib().wg_gemm_fp8_bias_act(xq.data_ptr(), K, x_scale.data_ptr(), wq.data_ptr(), K, w_scale.data_ptr(), _ptr(bias),
                                         _ptr(residual), ldr, res_row_mod, out.data_ptr(), ldc, M, N, K, act, _stream())
    if ev is not None:
        ev[1].record()
    _lib.check(rc, "wg_gemm_fp8_bias_act")
    return out


def tile_weight(w, k_slices=1):
    """Weight matrix [N, K] bf16 -> fragment order [k_slices, ceil(N / 16), K / (32 k_slices), 64, 8] for the token-side decoder kernels
    (wg_tile_weight_bf16; squeezed to 4-D when k_slices == 1).  k_slices > 1 tiles each K-slice on its own (mlp.lin2 of the two-way block)."""
    _need_gpu(w)
    assert w.dtype == _BF16 and w.dim() == 2 and w.stride(1) == 1
    N, K = w.shape
    assert K % (32 * k_slices) == 0
    ks = K // k_slices
    out = torch.empty(k_slices, (N + 15) // 16, ks // 32, 64, 8, device=w.device, dtype=_BF16)
    for s in range(k_slices):
        rc = _lib.lib().wg_tile_weight_bf16(w.data_ptr() + 2 * s * ks, w.stride(0), N, ks, out[s].data_ptr(), _stream())
        _lib.check(rc, "wg_tile_weight_bf16")
    return out[0] if k_slices == 1 else out


def _tiled_ok(t, N, K, k_slices=1):
    shape = ((N + 15) // 16, K // (32 * k_slices), 64, 8)
    return t.dtype == _BF16 and t.is_contiguous() and tuple(t.shape) == ((k_slices,) + shape if k_slices > 1 else shape)


TOK_SUM_MLP, TOK_SELF, TOK_Q_T2I, TOK_COMBINE, TOK_INIT = 1, 2, 4, 8, 16
_TOK_PART = 6 * 18          # floats of one attention partial (csrc/decoder.hip)


def _f32_tokens(t, P, last):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == (P, 6, last), (t.dtype, tuple(t.shape))


def dec_tokens(stages, skip_pe, queries, query_pe, weights, q_t2i=None, attn_partials=None, mlp_partials=None, k_i2t=None, v_i2t=None,
               eps=1e-5, init_tokens=None, init_prompt=None, prompt_tail=None):
    """Per-prompt stages of the token side of SAM's two-way transformer (csrc/decoder.hip: wg_dec_tokens_f32; stage bits TOK_*).
    queries / query_pe [P, 6, 256] fp32 (queries updated in place); weights: the 24-slot table the C-ABI documents (bf16 tensors or
    None for the slots of stages that do not run; the weight matrices in fragment order, ops.tile_weight).  TOK_INIT: queries and query_pe are OUTPUTS, both set to
    cat(init_tokens [5, 256] fp32, init_prompt [P, 256] bf16) before the other stages run; with prompt_tail = (gamma, beta, text_type, log_temp
    bf16 tensors, eps) init_prompt holds the text projector's rows BEFORE its tail and the launch applies the tail (wg_dec_tokens_ctp_f32: same
    bits as ops.ctp_tail followed by the plain launch)."""
    import ctypes
    _need_gpu(queries, query_pe, q_t2i, attn_partials, mlp_partials, k_i2t, v_i2t, init_tokens, init_prompt, *weights)
    P = queries.shape[0]
    _f32_tokens(queries, P, 256)
    _f32_tokens(query_pe, P, 256)
    assert len(weights) == 24 and all(w is None or (w.dtype == _BF16 and w.is_contiguous()) for w in weights)
    n_splits = 0
    if stages & TOK_INIT:
        assert init_tokens.dtype == torch.float32 and init_tokens.is_contiguous() and init_tokens.shape == (5, 256)
        assert init_prompt.dtype == _BF16 and init_prompt.is_contiguous() and init_prompt.numel() == P * 256
    if stages & TOK_Q_T2I:
        _f32_tokens(q_t2i, P, 128)
    if stages & TOK_COMBINE:
        assert attn_partials.dtype == torch.float32 and attn_partials.is_contiguous() and attn_partials.dim() == 4
        assert attn_partials.shape[0] == P and attn_partials.shape[1] == 8 and attn_partials.shape[3] == _TOK_PART
        n_splits = attn_partials.shape[2]
    if stages & TOK_SUM_MLP:
        assert mlp_partials.dtype == torch.float32 and mlp_partials.is_contiguous() and mlp_partials.shape == (P, dec_mlp_slices(), 6, 256)
        for t in (k_i2t, v_i2t):
            assert t.dtype == _BF16 and t.is_contiguous() and t.shape == (P, 6, 128)
    table = (ctypes.c_void_p * 24)(*[None if w is None else w.data_ptr() for w in weights])
    if prompt_tail is not None:
        assert stages & TOK_INIT
        g_, b_, tt_, lt_, teps = prompt_tail
        _need_gpu(g_, b_, tt_, lt_)
        assert all(t.dtype == _BF16 and t.is_contiguous() for t in (g_, b_, tt_, lt_)) and g_.numel() == b_.numel() == tt_.numel() == 256
        ttab = (ctypes.c_void_p * 4)(g_.data_ptr(), b_.data_ptr(), tt_.data_ptr(), lt_.data_ptr())
        rc = _lib.lib().wg_dec_tokens_ctp_f32(stages, 1 if skip_pe else 0, queries.data_ptr(), query_pe.data_ptr(), _ptr(init_tokens), _ptr(init_prompt),
                                              ttab, float(teps), table, 24, _ptr(q_t2i), _ptr(attn_partials), n_splits, _ptr(mlp_partials),
                                              _ptr(k_i2t), _ptr(v_i2t), P, float(eps), _stream())
        _lib.check(rc, "wg_dec_tokens_ctp_f32")
        return queries
    rc = _lib.lib().wg_dec_tokens_f32(stages, 1 if skip_pe else 0, queries.data_ptr(), query_pe.data_ptr(), _ptr(init_tokens), _ptr(init_prompt),
                                      table, 24, _ptr(q_t2i),
                                      _ptr(attn_partials), n_splits, _ptr(mlp_partials), _ptr(k_i2t), _ptr(v_i2t), P, float(eps), _stream())
    _lib.check(rc, "wg_dec_tokens_f32")
    return queries


def _prompt_image_ok(prompt_image, P, n_images):
    assert prompt_image.dtype == torch.int32 and prompt_image.is_contiguous() and prompt_image.numel() == P and n_images >= 1


def dec_attn_partial(q_t2i, kv_img, prompt_image=None):
    """Token->image attention partials (wg_dec_attn_partial_f32).  q_t2i [P, 6, 128] fp32; kv_img [1 | P, hw, 256] bf16: the projected image
    tokens with the columns ordered [K_h | V_h] for head h = 0..7 (16 + 16 each; a column slice of the fused image-side projection);
    with prompt_image (int32 [P]): kv_img [B, hw, 256] holds one block per IMAGE and prompt p attends to block prompt_image[p]
    -> fp32 [P, 8, ceil(hw / 1024), 108] for the COMBINE step."""
    _need_gpu(q_t2i, kv_img, prompt_image)
    P = q_t2i.shape[0]
    _f32_tokens(q_t2i, P, 128)
    hw = kv_img.shape[1]
    assert kv_img.dtype == _BF16 and kv_img.shape[-1] == 256 and kv_img.stride(-1) == 1
    if prompt_image is not None:
        _prompt_image_ok(prompt_image, P, kv_img.shape[0])
        img_bs = kv_img.stride(0) // kv_img.stride(1) if kv_img.shape[0] > 1 else hw
    else:
        assert kv_img.shape[0] in (1, P)
        img_bs = 0 if kv_img.shape[0] == 1 and P > 1 else (kv_img.stride(0) // kv_img.stride(1) if kv_img.shape[0] > 1 else hw)
    n_splits = (hw + 1023) // 1024
    part = torch.empty(P, 8, n_splits, _TOK_PART, device=q_t2i.device, dtype=torch.float32)
    rc = _lib.lib().wg_dec_attn_partial_f32(q_t2i.data_ptr(), kv_img.data_ptr(), kv_img.data_ptr() + 32, kv_img.stride(1), 32, img_bs,
                                            _ptr(prompt_image), hw, part.data_ptr(), n_splits, P, _stream())
    _lib.check(rc, "wg_dec_attn_partial_f32")
    return part


def _combine_table(combine, P):
    """(attention partials [P, 8, n_splits, 108] fp32, out_proj weight [256, 128], bias, LayerNorm gamma, beta) -> (ctypes table, n_splits)"""
    import ctypes
    if combine is None:
        return None, 0
    part, wo, bo, g, b = combine
    _need_gpu(part, wo, bo, g, b)
    assert part.dtype == torch.float32 and part.is_contiguous() and part.dim() == 4 and part.shape[0] == P and part.shape[1] == 8
    assert part.shape[3] == _TOK_PART and _tiled_ok(wo, 256, 128) and bo.numel() == 256 and g.numel() == 256 and b.numel() == 256
    assert all(t.dtype == _BF16 and t.is_contiguous() for t in (bo, g, b))
    return (ctypes.c_void_p * 5)(part.data_ptr(), wo.data_ptr(), bo.data_ptr(), g.data_ptr(), b.data_ptr()), part.shape[2]


def dec_mlp_slices():
    """Slices the library cuts the decoder MLP's 2048 hidden units into (wg_dec_mlp_slices): lin2 is tiled in that many K-slices."""
    return int(_lib.lib().wg_dec_mlp_slices())


def dec_mlp_partial(x, lin1_w, lin1_b, lin2_w, combine=None, eps=1e-5):
    """The S = dec_mlp_slices() slices of mlp(x) for x [P, 6, 256] fp32 (wg_dec_mlp_partial_f32) -> fp32 [P, S, 6, 256]; lin2's bias is added
    by dec_tokens(TOK_SUM_MLP).  With `combine` (see _combine_table) x holds the tokens before the COMBINE stage, which the launch runs
    itself: returns (partials, tokens after the LayerNorm -- a new buffer)."""
    _need_gpu(x, lin1_w, lin1_b, lin2_w)
    P = x.shape[0]
    _f32_tokens(x, P, 256)
    S = dec_mlp_slices()
    assert _tiled_ok(lin1_w, 2048, 256) and _tiled_ok(lin2_w, 256, 2048, S) and lin1_b.shape == (2048,)     # ops.tile_weight(w), (w, S)
    assert lin1_b.dtype == _BF16 and lin1_b.is_contiguous()
    out = torch.empty(P, S, 6, 256, device=x.device, dtype=torch.float32)
    table, n_splits = _combine_table(combine, P)
    x_out = torch.empty_like(x) if combine is not None else None
    rc = _lib.lib().wg_dec_mlp_partial_f32(x.data_ptr(), table, n_splits, float(eps), _ptr(x_out), lin1_w.data_ptr(), lin1_b.data_ptr(),
                                           lin2_w.data_ptr(), out.data_ptr(), P, _stream())
    _lib.check(rc, "wg_dec_mlp_partial_f32")
    return out if combine is None else (out, x_out)


def dec_heads(x, weights, combine=None, eps=1e-5):
    """The four hypernetwork MLPs and the IoU head on x [P, 6, 256] fp32 (wg_dec_heads_f32); weights: 30 bf16 tensors
    -> (hyper fp32 [P, 4, 32], iou fp32 [P, 4]).  With `combine` the launch first applies the final attention's COMBINE stage."""
    import ctypes
    _need_gpu(x, *weights)
    P = x.shape[0]
    _f32_tokens(x, P, 256)
    assert len(weights) == 30 and all(w.dtype == _BF16 and w.is_contiguous() for w in weights)
    for i in range(5):
        assert all(_tiled_ok(weights[(3 * i + j) * 2], n, 256) for j, n in enumerate((256, 256, 32 if i < 4 else 4))), i
    hyper = torch.empty(P, 4, 32, device=x.device, dtype=torch.float32)
    iou = torch.empty(P, 4, device=x.device, dtype=torch.float32)
    table = (ctypes.c_void_p * 30)(*[w.data_ptr() for w in weights])
    ctab, n_splits = _combine_table(combine, P)
    rc = _lib.lib().wg_dec_heads_f32(x.data_ptr(), ctab, n_splits, float(eps), table, 30, hyper.data_ptr(), iou.data_ptr(), P, _stream())
    _lib.check(rc, "wg_dec_heads_f32")
    return hyper, iou


def dec_i2t_rows(q_img, k_i2t, v_i2t, out_w, out_b, keys, ln_g, ln_b, eps, P, res_bias=None, prompt_image=None):
    """transformer.py:173-180 in one launch (wg_dec_i2t_rows_bf16): norm4(keys + out_proj(attention of every image token over the six
    prompt tokens)).  q_img [1 | P, hw, 128] bf16 (column slice of the image-side projection), k_i2t / v_i2t [P, 6, 128] bf16,
    keys [1 | P, hw, 256] bf16 -> bf16 [P, hw, 256].  res_bias [256] bf16: the residual is keys + res_bias (a constant row the caller
    never added to the image tokens).  prompt_image (int32 [P]): q_img / keys hold one block per IMAGE, prompt p reads block prompt_image[p]."""
    _need_gpu(q_img, k_i2t, v_i2t, out_w, out_b, keys, ln_g, ln_b, res_bias, prompt_image)
    assert res_bias is None or (res_bias.dtype == _BF16 and res_bias.numel() == 256 and res_bias.is_contiguous())
    hw = keys.shape[1]
    assert q_img.dtype == _BF16 and q_img.shape[-1] == 128 and q_img.stride(-1) == 1 and q_img.shape[:2] == keys.shape[:2]
    assert keys.dtype == _BF16 and keys.shape[-1] == 256 and keys.stride(-1) == 1 and (prompt_image is not None or keys.shape[0] in (1, P))
    if prompt_image is not None:
        _prompt_image_ok(prompt_image, P, keys.shape[0])
    assert k_i2t.shape == (P, 6, 128) and v_i2t.shape == (P, 6, 128) and k_i2t.is_contiguous() and v_i2t.is_contiguous()
    assert out_w.shape == (256, 128) and out_w.is_contiguous() and out_b.numel() == 256 and ln_g.numel() == 256 and ln_b.numel() == 256
    for t in (q_img, keys):          # rows of all prompts must be evenly spaced
        assert t.shape[0] == 1 or t.stride(0) == hw * t.stride(1)
    shared = prompt_image is None and keys.shape[0] == 1 and P > 1
    out = torch.empty(P, hw, 256, device=keys.device, dtype=_BF16)
    rc = _lib.lib().wg_dec_i2t_rows_bf16(q_img.data_ptr(), q_img.stride(1), k_i2t.data_ptr(), v_i2t.data_ptr(), out_w.data_ptr(),
                                         out_b.data_ptr(), keys.data_ptr(), keys.stride(1), _ptr(res_bias), hw if shared else 0, _ptr(prompt_image),
                                         ln_g.data_ptr(),
                                         ln_b.data_ptr(), float(eps), out.data_ptr(), P, hw, _stream())
    _lib.check(rc, "wg_dec_i2t_rows_bf16")
    return out


def upscale_mask(keys, up1_w, up1_b, ln_g, ln_b, eps, up2_w, up2_b, hyper, h, w, first_mask, num_masks):
    """mask_decoder.py:140-160 in one launch.  keys [P, h*w, 256] bf16 image tokens; up1_w [(dy,dx,64), 256] (in fragment order,
    ops.tile_weight), up2_w [(dy,dx,32), 64] (re-laid ConvTranspose2d weights); hyper [P, nmask, 32] fp32 -> fp32 [P, num_masks, 4h, 4w]."""
    _need_gpu(keys, up1_w, up1_b, ln_g, ln_b, up2_w, up2_b, hyper)
    P = keys.shape[0]
    assert keys.dtype == _BF16 and keys.is_contiguous() and keys.shape[1:] == (h * w, 256)
    assert _tiled_ok(up1_w, 256, 256) and up2_w.shape == (128, 64) and up2_w.is_contiguous()     # up1_w: ops.tile_weight of the re-laid ConvT
    assert hyper.dtype == torch.float32 and hyper.is_contiguous() and hyper.shape[0] == P and hyper.shape[2] == 32
    masks = torch.empty(P, num_masks, 4 * h, 4 * w, device=keys.device, dtype=torch.float32)
    rc = _lib.lib().wg_upscale_mask_bf16(keys.data_ptr(), 256, up1_w.data_ptr(), up1_b.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(),
                                         float(eps), up2_w.data_ptr(), up2_b.data_ptr(), hyper.data_ptr(), masks.data_ptr(), P, h, w,
                                         hyper.shape[1], first_mask, num_masks, _stream())
    _lib.check(rc, "wg_upscale_mask_bf16")
    return masks


def postprocess_masks(low_res, img_size, input_size, original_size):
    """fp32 [N, C, lh, lw] -> fp32 [N, C, H0, W0] (Sam.postprocess_masks, both resamples + crop in one pass)."""
    _need_gpu(low_res)
    assert low_res.dtype == torch.float32 and low_res.is_contiguous() and low_res.dim() == 4
    N, C, lh, lw = low_res.shape
    out = torch.empty(N, C, int(original_size[0]), int(original_size[1]), device=low_res.device, dtype=torch.float32)
    rc = _lib.lib().wg_postprocess_masks_f32(low_res.data_ptr(), out.data_ptr(), N * C, lh, lw, img_size,
                                             int(input_size[0]), int(input_size[1]), int(original_size[0]),
                                             int(original_size[1]), _stream())
    _lib.check(rc, "wg_postprocess_masks_f32")
    return out


def postprocess_masks_scored(low_res, img_size, input_size, original_size):
    """postprocess_masks for single-channel masks plus their mask scores in the same pass (wg_postprocess_masks_score_f32):
    fp32 [N, 1, lh, lw] -> (fp32 [N, H0, W0], fp32 [N])."""
    _need_gpu(low_res)
    assert low_res.dtype == torch.float32 and low_res.is_contiguous() and low_res.dim() == 4 and low_res.shape[1] == 1
    N, _, lh, lw = low_res.shape
    H0, W0 = int(original_size[0]), int(original_size[1])
    out = torch.empty(N, H0, W0, device=low_res.device, dtype=torch.float32)
    score = torch.empty(N, device=low_res.device, dtype=torch.float32)
    L = _lib.lib()
    nws = L.wg_postprocess_score_workspace_floats(N, H0, W0)
    ws = torch.empty(nws, device=low_res.device, dtype=torch.float32)
    if N <= _SCORE_FUSED_MASKS:
        # one launch: the workgroup that completes a mask folds its partials (tickets: zero words the kernel leaves zero, one set per stream).
        # For the few masks of one image's decode, where the chain is latency (-1.3 us of 200); with a batch's masks the fat workgroups of
        # the fused form cost more than the second launch (8 masks: 19 against 15 us), so those keep the two-launch form
        rc = L.wg_postprocess_masks_score_fused_f32(low_res.data_ptr(), out.data_ptr(), score.data_ptr(), ws.data_ptr(), nws,
                                                    _score_tickets(low_res.device).data_ptr(), N, lh, lw, img_size, int(input_size[0]),
                                                    int(input_size[1]), H0, W0, _stream())
        _lib.check(rc, "wg_postprocess_masks_score_fused_f32")
        return out, score
    rc = L.wg_postprocess_masks_score_f32(low_res.data_ptr(), out.data_ptr(), score.data_ptr(), ws.data_ptr(), nws, N, lh, lw, img_size,
                                          int(input_size[0]), int(input_size[1]), H0, W0, _stream())
    _lib.check(rc, "wg_postprocess_masks_score_f32")
    return out, score


_SCORE_TICKETS = 64
_SCORE_FUSED_MASKS = 4
_score_ticket_cache = {}
_score_ticket_pool = {}
_score_ticket_keep = []      # every set ever handed to a capture: a captured graph bakes in the raw address, so the storage lives as long as the process


def _score_tickets(device):
    """Zero words for wg_postprocess_masks_score_fused_f32 (the kernel leaves them zero).  Eager calls share one set per (device, stream): launches on
    one HIP stream are ordered, which is all that sharing needs.  A call made under stream CAPTURE gets a set of its own: the graph is replayed on
    whatever stream is current then, possibly beside eager calls on a stream whose pooled handle equals the capture stream's, and two launches adding
    to the same words misdetect the last arriver.  The sets for captures come from a small pool zeroed eagerly on the first eager call (no memset node
    in the latency-bound decode graph); a capture that finds the pool empty allocates inside the capture (a memset node, replayed with the graph).
    The caller only takes data_ptr(): every set handed to a capture is therefore parked in _score_ticket_keep (views keep their whole block alive), or the
    caching allocator would hand the words to a later tensor while replayed graphs still count arrivals in them."""
    if torch.cuda.is_current_stream_capturing():
        pool = _score_ticket_pool.get(device.index)
        t = pool.pop() if pool else torch.zeros(_SCORE_TICKETS, device=device, dtype=torch.int32)
        _score_ticket_keep.append(t)
        return t
    if device.index not in _score_ticket_pool:
        _score_ticket_pool[device.index] = list(torch.zeros(16, _SCORE_TICKETS, device=device, dtype=torch.int32).unbind(0))
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _score_ticket_cache.get(key)
    if t is None:
        t = _score_ticket_cache[key] = torch.zeros(_SCORE_TICKETS, device=device, dtype=torch.int32)
    return t


def mask_score(masks):
    """fp32 [N, H, W] -> fp32 [N]."""
    _need_gpu(masks)
    assert masks.dtype == torch.float32 and masks.is_contiguous()
    N = masks.shape[0]
    hw = masks.numel() // N
    score = torch.empty(N, device=masks.device, dtype=torch.float32)
    nws = _lib.lib().wg_mask_score_workspace_floats(N, hw)
    ws = torch.empty(nws, device=masks.device, dtype=torch.float32)
    _lib.check(_lib.lib().wg_mask_score_f32(masks.data_ptr(), score.data_ptr(), ws.data_ptr(), nws, N, hw, _stream()), "wg_mask_score_f32")
    return score


def avgpool_tokens(x, B, H, W, s):
    """channels-last [B,H,W,C] rows -> [B, (H/s)*(W/s), C]."""
    _need_gpu(x)
    C = x.shape[-1]
    assert x.is_contiguous() and x.dtype == _BF16 and x.numel() == B * H * W * C
    y = torch.empty(B, (H // s) * (W // s), C, device=x.device, dtype=_BF16)
    _lib.check(_lib.lib().wg_avgpool_tokens_bf16(x.data_ptr(), y.data_ptr(), B, H, W, C, s, _stream()), "wg_avgpool_tokens_bf16")
    return y


def mean_tokens(x):
    """[B, L, C] -> [B, 1, C]."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dtype == _BF16 and x.dim() == 3
    B, L, C = x.shape
    y = torch.empty(B, 1, C, device=x.device, dtype=_BF16)
    _lib.check(_lib.lib().wg_mean_tokens_bf16(x.data_ptr(), y.data_ptr(), B, L, C, _stream()), "wg_mean_tokens_bf16")
    return y


def sigmoid_gate(x, logit):
    """x [..., C] bf16 * sigmoid(logit [...]) fp32."""
    _need_gpu(x, logit)
    assert x.is_contiguous() and logit.is_contiguous() and logit.dtype == torch.float32
    C = x.shape[-1]
    rows = x.numel() // C
    assert logit.numel() == rows
    y = torch.empty(x.shape, device=x.device, dtype=_BF16)
    _lib.check(_lib.lib().wg_sigmoid_gate_bf16(x.data_ptr(), logit.data_ptr(), y.data_ptr(), rows, C, _stream()), "wg_sigmoid_gate_bf16")
    return y


def ctp_tail(x, gamma, beta, text_type, log_temp, eps):
    """normalize(LayerNorm(x) + text_type, dim=-1) * exp(log_temp)."""
    _need_gpu(x, gamma, beta, text_type, log_temp)
    M, C, ldx = _rows(x)
    y = torch.empty(x.shape, device=x.device, dtype=_BF16)
    _, _, ldy = _rows(y)
    rc = _lib.lib().wg_ctp_tail_bf16(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), text_type.data_ptr(),
                                     log_temp.data_ptr(), y.data_ptr(), ldy, M, C, float(eps), _stream())
    _lib.check(rc, "wg_ctp_tail_bf16")
    return y


def resample_tokens(x, target=16):
    """[n, p*p, C] -> bilinear (align_corners=False) -> [n, target*target, C]."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dtype == _BF16 and x.dim() == 3
    n, l, C = x.shape
    p = int(round(l ** 0.5))
    assert p * p == l, "Token count %d is not square." % l
    y = torch.empty(n, target * target, C, device=x.device, dtype=_BF16)
    _lib.check(_lib.lib().wg_resample_tokens_bf16(x.data_ptr(), y.data_ptr(), n, p, target, C, _stream()), "wg_resample_tokens_bf16")
    return y


def mask_iou(pred_logits, gt, ignore_index=255):
    """intersectionAndUnionGPU(pred > 0, gt, K=2, ignore_index) per mask, thresholding fused.  pred_logits, gt fp32 [N,H,W]
    (gt values 0 / 1 / ignore_index) -> (intersection [N,2], union [N,2], target_area [N,2]) fp32."""
    _need_gpu(pred_logits, gt)
    assert pred_logits.dtype == torch.float32 and gt.dtype == torch.float32 and pred_logits.shape == gt.shape
    assert pred_logits.is_contiguous() and gt.is_contiguous()
    N = pred_logits.shape[0]
    hw = pred_logits.numel() // N
    out = torch.empty(N, 6, device=pred_logits.device, dtype=torch.float32)
    nws = _lib.lib().wg_mask_stats_workspace_floats(N, hw)
    ws = torch.empty(nws, device=pred_logits.device, dtype=torch.float32)
    _lib.check(_lib.lib().wg_mask_iou_f32(pred_logits.data_ptr(), gt.data_ptr(), out.data_ptr(), ws.data_ptr(), nws, N, hw,
                                          float(ignore_index), _stream()), "wg_mask_iou_f32")
    return out[:, 0:2], out[:, 2:4], out[:, 4:6]


def mask_losses(pred_logits, targets, num_masks, dice_scale=1000.0, dice_eps=1e-6):
    """(sigmoid_ce_loss, dice_loss) forward of utils_walkgpt.py:76-120: per-mask reductions on the GPU, the final
    sum / (num_masks + 1e-8) over N scalars on the host side of the stream (torch, N values)."""
    _need_gpu(pred_logits, targets)
    assert pred_logits.dtype == torch.float32 and targets.dtype == torch.float32 and pred_logits.shape == targets.shape
    assert pred_logits.is_contiguous() and targets.is_contiguous()
    N = pred_logits.shape[0]
    hw = pred_logits.numel() // N
    out = torch.empty(N, 2, device=pred_logits.device, dtype=torch.float32)
    nws = _lib.lib().wg_mask_stats_workspace_floats(N, hw)
    ws = torch.empty(nws, device=pred_logits.device, dtype=torch.float32)
    _lib.check(_lib.lib().wg_mask_losses_f32(pred_logits.data_ptr(), targets.data_ptr(), out.data_ptr(), ws.data_ptr(), nws, N, hw,
                                             float(dice_scale), float(dice_eps), _stream()), "wg_mask_losses_f32")
    per = out.sum(0) / (num_masks + 1e-8)
    return per[0], per[1]


def nce_forward(z, sam_tokens, seg_row_ids, wq, bq, wk_t, temperature, top_k, exclude_same_row, want_logits=False):
    """Device part of infonce_loss() (utils_walkgpt.py:8-73) for normalize=True.

    z [M, D] bf16 [SEG] embeddings; sam_tokens [rows, N, D] bf16; seg_row_ids [M] int; wq/bq the query projection of
    TinyCrossAttn and wk_t = Wk^T (contiguous [D, D]): q.(Wk kv) = (Wk^T q).kv.  Returns a dict with the fp32 GEMM `st`,
    `inv_norm`, `zn`, `attn_w` [M, N], `vraw` [M, D] (top-k refined positive, or the attention-pooled raw token when the
    refinement is off) and a `finish(vpos)` closure producing (loss, loss_m, logits)."""
    _need_gpu(z, sam_tokens, seg_row_ids)
    assert z.dtype == _BF16 and sam_tokens.dtype == _BF16 and sam_tokens.is_contiguous() and z.dim() == 2
    rows, N, D = sam_tokens.shape
    M = z.shape[0]
    assert z.shape[1] == D and seg_row_ids.shape == (M,)
    dev = z.device
    L = _lib.lib()
    tok2 = sam_tokens.reshape(rows * N, D)
    seg = seg_row_ids.to(torch.int32).contiguous()
    zc = z.contiguous()
    qcat = torch.empty(2 * M, D, device=dev, dtype=_BF16)
    _lib.check(L.wg_l2_normalize_rows_bf16(zc.data_ptr(), D, qcat.data_ptr(), D, M, D, 1e-12, _stream()), "wg_l2_normalize_rows_bf16")
    q = linear(zc, wq, bq)                       # [M, D]   TinyCrossAttn.wq
    linear(q, wk_t, out=qcat[M:])                # rows M..2M-1: Wk^T q
    inv_norm = torch.empty(rows * N, device=dev, dtype=torch.float32)
    _lib.check(L.wg_row_inv_norm_bf16(tok2.data_ptr(), D, inv_norm.data_ptr(), rows * N, D, 1e-12, _stream()), "wg_row_inv_norm_bf16")
    st = linear(qcat, tok2, out_f32=True)        # [2M, rows*N] fp32
    attn_w = torch.empty(M, N, device=dev, dtype=torch.float32)
    vraw = torch.empty(M, D, device=dev, dtype=torch.float32)
    k = int(top_k) if (top_k is not None and 0 < top_k < N) else 0
    _lib.check(L.wg_nce_attn_f32(st.data_ptr(), st.stride(0), tok2.data_ptr(), D, seg.data_ptr(), attn_w.data_ptr(), vraw.data_ptr(),
                                 M, N, rows, D, k if k <= 32 else 0, float(D) ** -0.5, _stream()), "wg_nce_attn_f32")
    if k > 32:
        # beyond the attention kernel's in-register selection: pick the k tokens from its weights and pool them with the streaming kernel
        # (weights renormalised over the selection = the softmax of their scores, utils_walkgpt.py:37-40)
        idx = torch.topk(attn_w, k=k, dim=1).indices
        kt = torch.gather(sam_tokens.index_select(0, seg_row_ids.long()), 1, idx.unsqueeze(-1).expand(-1, -1, D)).contiguous()
        vb = torch.empty(M, D, device=dev, dtype=_BF16)
        _lib.check(L.wg_pool_rows_bf16(qcat[M:].data_ptr(), kt.data_ptr(), None, vb.data_ptr(), M, k, D, _stream()), "wg_pool_rows_bf16")
        vraw = vb.float()

    def finish(vpos):
        assert vpos.dtype == torch.float32 and vpos.is_contiguous() and vpos.shape == (M, D)
        loss_m = torch.empty(M, device=dev, dtype=torch.float32)
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        logits = torch.empty(M, 1 + rows * N, device=dev, dtype=torch.float32) if want_logits else None
        _lib.check(L.wg_nce_loss_f32(st.data_ptr(), st.stride(0), inv_norm.data_ptr(), qcat.data_ptr(), vpos.data_ptr(), seg.data_ptr(),
                                     loss_m.data_ptr(), loss.data_ptr(), _ptr(logits), M, N, rows, D, 1 if exclude_same_row else 0,
                                     float(temperature), _stream()), "wg_nce_loss_f32")
        return loss[0], loss_m, logits

    return {"st": st, "inv_norm": inv_norm, "zn": qcat[:M], "attn_w": attn_w, "vraw": vraw, "refined": k > 0, "finish": finish}


def match_cost(pred_logits, targets, point_coords):
    """[P, T] matching cost of utils/matcher.py:93-133 (point-sampled BCE + dice).  pred_logits [P,H,W], targets [T,H,W]
    fp32, point_coords [NP, 2] fp32 in [0,1]^2 as (x, y)."""
    _need_gpu(pred_logits, targets, point_coords)
    assert pred_logits.dtype == torch.float32 and targets.dtype == torch.float32 and point_coords.dtype == torch.float32
    assert pred_logits.is_contiguous() and targets.is_contiguous() and point_coords.is_contiguous()
    P, H, W = pred_logits.shape
    T = targets.shape[0]
    assert targets.shape[1:] == (H, W) and point_coords.dim() == 2 and point_coords.shape[1] == 2
    NP = point_coords.shape[0]
    cost = torch.empty(P, T, device=pred_logits.device, dtype=torch.float32)
    nws = _lib.lib().wg_match_cost_workspace_floats(P, T, NP)
    ws = torch.empty(nws, device=pred_logits.device, dtype=torch.float32)
    _lib.check(_lib.lib().wg_match_cost_f32(pred_logits.data_ptr(), targets.data_ptr(), point_coords.data_ptr(), cost.data_ptr(),
                                            ws.data_ptr(), nws, P, T, H, W, NP, _stream()), "wg_match_cost_f32")
    return cost
