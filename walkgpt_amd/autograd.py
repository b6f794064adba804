"""Differentiable forms of the HIP operators the TRAINABLE parts of the grounding head are built from (train_walkgpt.py:347-350 trains the
mask decoder, text_hidden_fcs (CTP) and the projector; the SAM / CLIP encoders and the prompt encoder stay frozen).

The reference gets these gradients from torch autograd over nn.Linear / nn.LayerNorm / nn.GELU / ...; here every forward AND backward is a
HIP kernel behind the C-ABI (gemm.hip for the three GEMMs of a Linear, csrc/backward.hip for the rest) wrapped in torch.autograd.Function,
so that `loss.backward()` on a loss computed through this module fills `.grad` of the parameters and of the incoming hidden states (which
the caller's LLM continues from).  Activations and gradients travel as bf16, parameter gradients are accumulated in fp32 and handed to
autograd in the parameter's dtype.  Inference keeps using the fused kernels (walkgpt_amd.ops, segment_anything.modeling); this module is the
per-operator path they decompose into when gradients are asked for.
"""
import torch

from . import _lib, ops

BF16 = torch.bfloat16


def _pad_rows(t, mult=64):
    """[M, C] -> [M rounded up to `mult`, C] with zero rows behind (GEMM reductions over M want whole MFMA slabs)."""
    M = t.shape[0]
    Mp = (M + mult - 1) // mult * mult
    if Mp == M:
        return t
    out = torch.zeros(Mp, t.shape[1], device=t.device, dtype=t.dtype)
    out[:M] = t
    return out


def _dense2d(t, C):
    """[..., C] -> [rows, C] with strides exactly (C, 1) (torch calls a [rows, 1] tensor contiguous whatever its last stride is; the C-ABI
    takes leading dimensions literally)."""
    t = t.reshape(-1, C)
    if not t.is_contiguous():
        t = t.contiguous()
    if t.stride() != (C, 1):
        t = t.as_strided(t.shape, (C, 1))
    return t


def transpose2d(t):
    """[R, C] bf16 contiguous -> [C, R] (wg_tokens_to_nchw_bf16 with one batch)."""
    R, C = t.shape
    return _dense2d(ops.tokens_to_nchw(_dense2d(t, C), 1, R, C).view(C, R), R)


def colsum(t):
    """[R, C] bf16 -> fp32 [C] column sums (wg_colsum_det_f32: per-workgroup partial rows folded in a fixed order, no atomics)."""
    R, C = t.shape
    if C % 8 or t.stride(0) % 8 or t.data_ptr() % 16:      # the kernel reads 16-byte pieces: a handful of columns (IoU head: 4) go padded
        Cp = (C + 7) // 8 * 8
        tp = torch.zeros(R, Cp, device=t.device, dtype=t.dtype)
        tp[:, :C] = t
        return colsum(tp)[:C]
    L = _lib.lib()
    out = torch.empty(C, device=t.device, dtype=torch.float32)
    nws = L.wg_colsum_det_workspace_floats(R, C)
    ws = torch.empty(nws, device=t.device, dtype=torch.float32)
    _lib.check(L.wg_colsum_det_f32(t.data_ptr(), t.stride(0), out.data_ptr(), 1, ws.data_ptr(), nws, R, C, ops._stream()), "wg_colsum_det_f32")
    return out


def _bwd_gemm_ok(*ts):
    """csrc/gemm_bwd.hip takes bf16 operands with 16-byte rows (every width a multiple of 8 elements); the few Linears that are not (the IoU
    head's 4 outputs) keep the transposed-copy path."""
    return all(t.dtype == BF16 and t.shape[-1] % 8 == 0 and t.shape[-1] >= 8 and t.data_ptr() % 16 == 0 for t in ts)


def gemm_nn(dy2, w):
    """dX[M, K] = dY[M, N] . W[N, K] on the operands as they lie (wg_gemm_nn_bf16) -> bf16 [M, K]."""
    M, N = dy2.shape
    K = w.shape[1]
    L = _lib.lib()
    dx = torch.empty(M, K, device=dy2.device, dtype=BF16)
    nws = L.wg_gemm_bwd_workspace_floats(M, K, N, 0)
    ws = torch.empty(nws, device=dy2.device, dtype=torch.float32) if nws else None
    _lib.check(L.wg_gemm_nn_bf16(dy2.data_ptr(), N, w.data_ptr(), K, dx.data_ptr(), 0, ops._ptr(ws) or None, nws, M, N, K, ops._stream()), "wg_gemm_nn_bf16")
    return dx


def gemm_tn(dy2, x2, want_colsum, out_dtype):
    """dW[N, K] = dY[M, N]^T . X[M, K] and (want_colsum) db[N] = column sums of dY, one pass over dY (wg_gemm_tn_bf16); partial sums of a split
    reduction are combined in a fixed order: the same bits every run."""
    M, N = dy2.shape
    K = x2.shape[1]
    L = _lib.lib()
    f32 = out_dtype == torch.float32
    dt = torch.float32 if f32 else BF16
    dw = torch.empty(N, K, device=dy2.device, dtype=dt)
    db = torch.empty(N, device=dy2.device, dtype=dt) if want_colsum else None
    nws = L.wg_gemm_bwd_workspace_floats(N, K, M, 1 if want_colsum else 0)
    ws = torch.empty(nws, device=dy2.device, dtype=torch.float32) if nws else None
    _lib.check(L.wg_gemm_tn_bf16(dy2.data_ptr(), N, x2.data_ptr(), K, dw.data_ptr(), ops._ptr(db) or None, 1 if f32 else 0, ops._ptr(ws) or None, nws, M, N, K,
                                 ops._stream()), "wg_gemm_tn_bf16")
    if dt != out_dtype:
        dw, db = dw.to(out_dtype), (None if db is None else db.to(out_dtype))
    return dw, db


class _Linear(torch.autograd.Function):
    """y = x W^T + b.  Backward: dX = dY W and dW = dY^T X on the operands as they lie (csrc/gemm_bwd.hip: reduction-major tiles through the
    hardware transpose read; db = column sums of dY from the same pass; a long reduction split and summed in a fixed order).  Operands without
    16-byte rows take the older path: transposed copies (wg_tokens_to_nchw_bf16) through the forward GEMM, db by wg_colsum_f32."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_f32=False):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        with torch.no_grad():
            y = ops.linear(_dense2d(x, weight.shape[1]), _dense2d(weight, weight.shape[1]), bias, out_f32=out_f32)
            return y.view(x.shape[:-1] + (weight.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, K = weight.shape
        dy2 = _dense2d(dy.to(BF16), N)
        x2 = _dense2d(x, K)
        dx = dw = db = None
        w2 = _dense2d(weight.detach(), K)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[0]:
            if _bwd_gemm_ok(dy2, w2):
                dx = gemm_nn(dy2, w2).view(x.shape)
            else:
                dx = ops.linear(dy2, transpose2d(w2)).view(x.shape)                              # [M, N] @ [N, K]
        if ctx.needs_input_grad[1]:
            if _bwd_gemm_ok(dy2, x2):
                dw, db_ = gemm_tn(dy2, x2, want_db, weight.dtype)
                if want_db:
                    db, want_db = db_, False
            else:
                dyt, xt = transpose2d(_pad_rows(dy2)), transpose2d(_pad_rows(x2))                # [N, Mp], [K, Mp]
                dw = ops.linear(dyt, xt, out_f32=True).to(weight.dtype)                          # [N, K]
        if want_db:
            db = colsum(dy2).to(weight.dtype)
        return dx, dw, db, None


class _Act(torch.autograd.Function):
    """GELU (erf) / quick-GELU / ReLU as a separate operator (the fused GEMM epilogue keeps no pre-activation): wg_act_bf16 / wg_act_bwd_bf16."""

    @staticmethod
    def forward(ctx, x, code):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.code = code
        y = torch.empty_like(x)
        _lib.check(_lib.lib().wg_act_bf16(x.data_ptr(), y.data_ptr(), x.numel(), code, ops._stream()), "wg_act_bf16")
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().wg_act_bwd_bf16(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), ctx.code, ops._stream()), "wg_act_bwd_bf16")
        return dx, None


class _LayerNorm(torch.autograd.Function):
    """Row LayerNorm (ops.layernorm); backward wg_layernorm_bwd_bf16 (dgamma / dbeta accumulated in fp32 over the rows)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        with torch.no_grad():
            return ops.layernorm(x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        C = x.shape[-1]
        M = x.numel() // C
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        L = _lib.lib()
        if C in (64, 128, 256, 512) and gamma.dtype in (BF16, torch.float32) and dy.dtype == BF16:
            # the head's widths: every lane busy, no atomics (fixed-order sums: the same bits every run), parameter gradients written in their dtype
            f32 = gamma.dtype == torch.float32
            dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
            nws = L.wg_layernorm_bwd_det_workspace_floats(M, C)
            wsd = torch.empty(nws, device=x.device, dtype=torch.float32)
            g16 = gamma if gamma.dtype == BF16 else gamma.to(BF16)
            rc = L.wg_layernorm_bwd_det_bf16(x.data_ptr(), C, g16.data_ptr(), dy.data_ptr(), C, dx.data_ptr(), C, dg.data_ptr(), db.data_ptr(), 1 if f32 else 0,
                                             wsd.data_ptr(), nws, M, C, float(ctx.eps), ops._stream())
            _lib.check(rc, "wg_layernorm_bwd_det_bf16")
            return dx, dg, db, None
        dg = torch.zeros(C, device=x.device, dtype=torch.float32)
        db = torch.zeros(C, device=x.device, dtype=torch.float32)
        ws = torch.empty(2 * M, device=x.device, dtype=torch.float32) if C > 4096 else None      # (wide rows: {mean, rstd} between the two kernels)
        rc = _lib.lib().wg_layernorm_bwd_bf16(x.data_ptr(), C, gamma.data_ptr(), dy.data_ptr(), C, dx.data_ptr(), C, dg.data_ptr(), db.data_ptr(),
                                              ops._ptr(ws) or None, M, C, float(ctx.eps), ops._stream())
        _lib.check(rc, "wg_layernorm_bwd_bf16")
        return dx, dg.to(gamma.dtype), db.to(gamma.dtype), None


def linear(x, weight, bias=None, act=ops.ACT_NONE, out_f32=False):
    """Differentiable ops.linear (+ a separate activation operator when one is asked for).  out_f32: fp32 result (no activation)."""
    ops._need_gpu(x, weight, bias)
    y = _Linear.apply(x, weight, bias, out_f32)
    return y if act == ops.ACT_NONE else _Act.apply(y, act)


def activation(x, act):
    return x if act == ops.ACT_NONE else _Act.apply(x, act)


def layernorm(x, gamma, beta, eps):
    ops._need_gpu(x, gamma, beta)
    return _LayerNorm.apply(x, gamma, beta, eps)


class _AddRow(torch.autograd.Function):
    """x [..., C] + vec [C] (a type / positional embedding broadcast over the rows; ops.add_rows).  d vec = column sums of dy."""

    @staticmethod
    def forward(ctx, x, vec):
        with torch.no_grad():
            return ops.add_rows(x.contiguous(), vec.reshape(1, -1).contiguous())

    @staticmethod
    def backward(ctx, dy):
        C = dy.shape[-1]
        dv = colsum(dy.contiguous().view(-1, C)).to(dy.dtype) if ctx.needs_input_grad[1] else None
        return dy, dv


class _L2NormScale(torch.autograd.Function):
    """y = x / max(|x|, eps) * exp(log_temp) (utils_walkgpt.py:325-327): wg_l2norm_scale_bf16 / _bwd."""

    @staticmethod
    def forward(ctx, x, log_temp, eps):
        x = x.contiguous()
        ctx.save_for_backward(x, log_temp)
        ctx.eps = eps
        C = x.shape[-1]
        y = torch.empty_like(x)
        _lib.check(_lib.lib().wg_l2norm_scale_bf16(x.data_ptr(), log_temp.data_ptr(), y.data_ptr(), x.numel() // C, C, float(eps), ops._stream()),
                   "wg_l2norm_scale_bf16")
        return y

    @staticmethod
    def backward(ctx, dy):
        x, log_temp = ctx.saved_tensors
        C = x.shape[-1]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dt = torch.zeros(1, device=x.device, dtype=torch.float32)
        rc = _lib.lib().wg_l2norm_scale_bwd_bf16(x.data_ptr(), dy.data_ptr(), log_temp.data_ptr(), dx.data_ptr(), dt.data_ptr(), x.numel() // C, C,
                                                 float(ctx.eps), ops._stream())
        _lib.check(rc, "wg_l2norm_scale_bwd_bf16")
        return dx, dt.to(log_temp.dtype).view(log_temp.shape), None


class _Attention(torch.autograd.Function):
    """o = softmax(scale q k^T) v per head (ops.mha); backward wg_attn_bwd_bf16 (one side of every trainable attention has <= 16 rows)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        if any(ctx.needs_input_grad[:3]):
            # what the backward kernel takes, checked where the operator is CALLED (not in the middle of loss.backward())
            hd = q.shape[-1] // heads
            if _lib.lib().wg_attn_bwd_short_side(q.shape[1], k.shape[1]) < 0:
                raise NotImplementedError("differentiable attention: one side must have at most 16 rows (Lq = %d, Lk = %d)" % (q.shape[1], k.shape[1]))
            if hd > 128 or hd % 8 != 0:
                raise NotImplementedError("differentiable attention: head_dim %d (needs a multiple of 8, at most 128)" % hd)
        with torch.no_grad():
            o = ops.mha(q, k, v, heads, scale)
        ctx.save_for_backward(q, k, v, o)
        ctx.heads, ctx.scale = heads, scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o = ctx.saved_tensors
        B, Lq, D = q.shape
        Lk = k.shape[1]
        H = ctx.heads
        L = _lib.lib()
        side = L.wg_attn_bwd_short_side(Lq, Lk)
        if side < 0:
            raise NotImplementedError("attention backward: one side must have at most 16 rows (Lq = %d, Lk = %d)" % (Lq, Lk))
        dev = q.device
        do = do.contiguous()
        nws = L.wg_attn_bwd_workspace_floats(B, H, D // H, Lq, Lk)
        ws = torch.empty(nws, device=dev, dtype=torch.float32)       # softmax statistics + the short side's partial planes (summed in a fixed order)
        if side == 1:      # few keys
            dq = torch.empty_like(q)
            dk32, dv32 = torch.empty(B, Lk, D, device=dev), torch.empty(B, Lk, D, device=dev)
            args = (dq.data_ptr(), None, None, None, dk32.data_ptr(), dv32.data_ptr())
        else:              # few queries
            dk, dv = torch.empty_like(k), torch.empty_like(v)
            dq32 = torch.empty(B, Lq, D, device=dev)
            args = (None, dk.data_ptr(), dv.data_ptr(), dq32.data_ptr(), None, None)
        rc = L.wg_attn_bwd_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), *args, ws.data_ptr(), nws, B, H, D // H, Lq, Lk,
                                float(ctx.scale), ops._stream())
        _lib.check(rc, "wg_attn_bwd_bf16")
        if side == 1:
            return dq, dk32.to(BF16), dv32.to(BF16), None, None
        return dq32.to(BF16), dk, dv, None, None


class _Postprocess(torch.autograd.Function):
    """Sam.postprocess_masks (ops.postprocess_masks); backward: its adjoint (wg_postprocess_masks_bwd_f32)."""

    @staticmethod
    def forward(ctx, low_res, img_size, input_size, original_size):
        ctx.geom = (tuple(low_res.shape), int(img_size), (int(input_size[0]), int(input_size[1])), (int(original_size[0]), int(original_size[1])))
        with torch.no_grad():
            return ops.postprocess_masks(low_res.contiguous(), img_size, input_size, original_size)

    @staticmethod
    def backward(ctx, dout):
        shape, img, (ih, iw), (oh, ow) = ctx.geom
        N, C, lh, lw = shape
        dout = dout.contiguous().float()
        dlow = torch.empty(shape, device=dout.device, dtype=torch.float32)      # (written, not accumulated: the gather form)
        rc = _lib.lib().wg_postprocess_masks_bwd_f32(dout.data_ptr(), dlow.data_ptr(), N * C, lh, lw, img, ih, iw, oh, ow, ops._stream())
        _lib.check(rc, "wg_postprocess_masks_bwd_f32")
        return dlow, None, None, None


class _MaskLosses(torch.autograd.Function):
    """(sigmoid_ce_loss, dice_loss) of utils_walkgpt.py:76-120 (ops.mask_losses); backward wg_mask_losses_bwd_f32."""

    @staticmethod
    def forward(ctx, pred_logits, targets, num_masks, dice_scale, dice_eps):
        pred_logits, targets = pred_logits.contiguous(), targets.contiguous()
        ctx.save_for_backward(pred_logits, targets)
        ctx.cfg = (float(num_masks), float(dice_scale), float(dice_eps))
        with torch.no_grad():
            return ops.mask_losses(pred_logits, targets, num_masks, dice_scale, dice_eps)

    @staticmethod
    def backward(ctx, g_bce, g_dice):
        pred, tgt = ctx.saved_tensors
        num_masks, scale, eps = ctx.cfg
        N = pred.shape[0]
        hw = pred.numel() // N
        L = _lib.lib()
        nws = L.wg_mask_stats_workspace_floats(N, hw) + 2 * N
        ws = torch.empty(nws, device=pred.device, dtype=torch.float32)
        dpred = torch.empty_like(pred)
        k = 1.0 / (num_masks + 1e-8)
        # the two upstream gradients stay on the device (no host read inside the backward pass)
        zero = pred.new_zeros(())
        g2 = (torch.stack([zero if g_bce is None else g_bce.reshape(()).float(), zero if g_dice is None else g_dice.reshape(()).float()]) * k).contiguous()
        rc = L.wg_mask_losses_bwd_dev_f32(pred.data_ptr(), tgt.data_ptr(), dpred.data_ptr(), ws.data_ptr(), nws, N, hw, g2.data_ptr(), scale, eps, ops._stream())
        _lib.check(rc, "wg_mask_losses_bwd_dev_f32")
        return dpred, None, None, None, None


def add_row(x, vec):
    return _AddRow.apply(x, vec)


def l2norm_scale(x, log_temp, eps=1e-12):
    return _L2NormScale.apply(x, log_temp, eps)


def attention(q, k, v, heads, scale):
    """Differentiable ops.mha: q [B, Lq, D], k / v [B, Lk, D] bf16."""
    return _Attention.apply(q, k, v, heads, scale)


def postprocess_masks(low_res, img_size, input_size, original_size):
    return _Postprocess.apply(low_res, img_size, input_size, original_size)


def mask_losses(pred_logits, targets, num_masks, dice_scale=1000.0, dice_eps=1e-6):
    """-> (sigmoid_ce_loss, dice_loss), differentiable in pred_logits (fp32 [N, H, W])."""
    return _MaskLosses.apply(pred_logits, targets, num_masks, dice_scale, dice_eps)


class _AvgPool(torch.autograd.Function):
    """MSQP's _pool_grid_tokens (utils_walkgpt.py:195-201; ops.avgpool_tokens) and its backward."""

    @staticmethod
    def forward(ctx, x, B, H, W, s):
        ctx.geom = (B, H, W, x.shape[-1], s)
        with torch.no_grad():
            return ops.avgpool_tokens(x.contiguous(), B, H, W, s)

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, s = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty(B, H * W, C, device=dy.device, dtype=BF16)
        _lib.check(_lib.lib().wg_avgpool_tokens_bwd_bf16(dy.data_ptr(), dx.data_ptr(), B, H, W, C, s, ops._stream()), "wg_avgpool_tokens_bwd_bf16")
        return dx, None, None, None, None


class _MeanTokens(torch.autograd.Function):
    """MSQP's _global_token (utils_walkgpt.py:256-257; ops.mean_tokens): [B, L, C] -> [B, 1, C]."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        with torch.no_grad():
            return ops.mean_tokens(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        B, L, C = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, L, C, device=dy.device, dtype=BF16)
        _lib.check(_lib.lib().wg_mean_tokens_bwd_bf16(dy.data_ptr(), dx.data_ptr(), B, L, C, ops._stream()), "wg_mean_tokens_bwd_bf16")
        return dx


class _Gate(torch.autograd.Function):
    """SegAwareGate's tail y = x * sigmoid(logit) (utils_walkgpt.py:213-217; ops.sigmoid_gate); logit fp32 [rows, 1]."""

    @staticmethod
    def forward(ctx, x, logit):
        x, logit = x.contiguous(), logit.contiguous()
        ctx.save_for_backward(x, logit)
        with torch.no_grad():
            return ops.sigmoid_gate(x, logit)

    @staticmethod
    def backward(ctx, dy):
        x, logit = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dl = torch.empty(logit.shape, device=x.device, dtype=torch.float32)
        rc = _lib.lib().wg_sigmoid_gate_bwd_bf16(x.data_ptr(), logit.data_ptr(), dy.data_ptr(), dx.data_ptr(), dl.data_ptr(), rows, C, ops._stream())
        _lib.check(rc, "wg_sigmoid_gate_bwd_bf16")
        return dx, dl


class _Resample(torch.autograd.Function):
    """The token resample in front of the splice (llava_arch.py:252-259; ops.resample_tokens) and its adjoint."""

    @staticmethod
    def forward(ctx, x, target):
        ctx.shape, ctx.target = tuple(x.shape), target
        with torch.no_grad():
            return ops.resample_tokens(x.contiguous(), target)

    @staticmethod
    def backward(ctx, dy):
        n, pp, C = ctx.shape
        p = int(round(pp ** 0.5))
        dy = dy.contiguous()
        dx = torch.zeros(n, pp, C, device=dy.device, dtype=torch.float32)
        _lib.check(_lib.lib().wg_resample_tokens_bwd_f32(dy.data_ptr(), dx.data_ptr(), n, p, ctx.target, C, ops._stream()), "wg_resample_tokens_bwd_f32")
        return dx.to(BF16), None


class _Splice(torch.autograd.Function):
    """prepare_inputs_labels_for_multimodal (llava_arch.py:265-518; llava_splice) as a function of the image features and of
    embed_tokens.weight: the gradient of the spliced embeddings goes back to both (wg_splice_multimodal_bwd_bf16)."""

    @staticmethod
    def forward(ctx, image_features, embed_weight, input_ids, attention_mask, labels, vit_attention_mask, seg_token_idx, aux):
        from . import llava_splice
        with torch.no_grad():
            mask, embeds, lab, seg, pos = llava_splice.prepare_inputs_labels_for_multimodal(
                input_ids, attention_mask, labels, image_features, embed_weight, vit_attention_mask, seg_token_idx, return_positions=True)
        aux.extend([mask, lab, seg])
        ctx.save_for_backward(input_ids.contiguous(), pos)
        ctx.geom = (tuple(image_features.shape), tuple(embed_weight.shape), embed_weight.dtype)
        return embeds

    @staticmethod
    def backward(ctx, dembeds):
        ids, pos = ctx.saved_tensors
        (rows, T, H), (V, _), wdtype = ctx.geom
        L = ids.shape[1]
        dembeds = dembeds.contiguous().to(BF16)
        dimg = torch.empty(rows, T, H, device=dembeds.device, dtype=BF16)
        dtab = torch.zeros(V, H, device=dembeds.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        rc = _lib.lib().wg_splice_multimodal_bwd_bf16(ids.data_ptr(), pos.data_ptr(), dembeds.data_ptr(), dimg.data_ptr(), ops._ptr(dtab) or None, rows, L,
                                                      T, H, V, ops._stream())
        _lib.check(rc, "wg_splice_multimodal_bwd_bf16")
        return dimg, (dtab.to(wdtype) if dtab is not None else None), None, None, None, None, None, None


def avgpool_tokens(x, B, H, W, s):
    return _AvgPool.apply(x, B, H, W, s)


def mean_tokens(x):
    return _MeanTokens.apply(x)


def sigmoid_gate(x, logit):
    return _Gate.apply(x, logit)


def resample_tokens(x, target=16):
    return _Resample.apply(x, target)


def splice(input_ids, attention_mask, labels, image_features, embed_weight, vit_attention_mask=None, seg_token_idx=None):
    """Differentiable llava_splice.prepare_inputs_labels_for_multimodal -> (attention_mask, embeds, labels, seg_mask)."""
    aux = []
    embeds = _Splice.apply(image_features, embed_weight, input_ids, attention_mask, labels, vit_attention_mask, seg_token_idx, aux)
    return aux[0], embeds, aux[1], aux[2]


class _TopkPool(torch.autograd.Function):
    """v_m = sum_k softmax_k(u_m . kt_mk / sqrt(D)) kt_mk over the top-k SAM tokens of [SEG] m (infonce_loss's refined positive,
    utils_walkgpt.py:33-40 with TinyCrossAttn's projections folded into u); kt is a constant.  wg_topk_pool_bf16 / _bwd."""

    @staticmethod
    def forward(ctx, u, kt):
        u, kt = u.contiguous(), kt.contiguous()
        ctx.save_for_backward(u, kt)
        M, Kt, D = kt.shape
        v = torch.empty(M, D, device=u.device, dtype=BF16)
        _lib.check(_lib.lib().wg_topk_pool_bf16(u.data_ptr(), kt.data_ptr(), v.data_ptr(), M, Kt, D, ops._stream()), "wg_topk_pool_bf16")
        return v

    @staticmethod
    def backward(ctx, dv):
        u, kt = ctx.saved_tensors
        M, Kt, D = kt.shape
        dv = dv.contiguous()
        du = torch.empty_like(u)
        _lib.check(_lib.lib().wg_topk_pool_bwd_bf16(u.data_ptr(), kt.data_ptr(), dv.data_ptr(), du.data_ptr(), M, Kt, D, ops._stream()), "wg_topk_pool_bwd_bf16")
        return du, None


class _NceTail(torch.autograd.Function):
    """InfoNCE's cross-entropy over [positive | all SAM tokens] (utils_walkgpt.py:42-73): wg_nce_tail_f32 / _bwd."""

    @staticmethod
    def forward(ctx, z, vp, sim, own_row, rows, N, temperature, exclude):
        z, vp, sim = z.contiguous(), vp.contiguous(), sim.contiguous()
        M, D = z.shape
        loss_m = torch.empty(M, device=z.device, dtype=torch.float32)
        lse = torch.empty(M, device=z.device, dtype=torch.float32)
        rc = _lib.lib().wg_nce_tail_f32(z.data_ptr(), vp.data_ptr(), sim.data_ptr(), own_row.data_ptr(), loss_m.data_ptr(), lse.data_ptr(), M, rows, N, D,
                                        float(temperature), 1 if exclude else 0, ops._stream())
        _lib.check(rc, "wg_nce_tail_f32")
        ctx.save_for_backward(z, vp, sim, own_row, lse)
        ctx.cfg = (rows, N, float(temperature), 1 if exclude else 0)
        return loss_m.mean()

    @staticmethod
    def backward(ctx, g):
        z, vp, sim, own_row, lse = ctx.saved_tensors
        rows, N, temperature, exclude = ctx.cfg
        M, D = z.shape
        dz, dvp, dsim = torch.empty_like(z), torch.empty_like(vp), torch.empty_like(sim)
        gd = g.reshape(1).float().contiguous()      # stays on the device: no host read inside the backward pass
        rc = _lib.lib().wg_nce_tail_bwd_dev_f32(z.data_ptr(), vp.data_ptr(), sim.data_ptr(), own_row.data_ptr(), lse.data_ptr(), gd.data_ptr(), dz.data_ptr(),
                                                dvp.data_ptr(), dsim.data_ptr(), M, rows, N, D, temperature, exclude, ops._stream())
        _lib.check(rc, "wg_nce_tail_bwd_dev_f32")
        return dz, dvp, dsim, None, None, None, None, None


class _PoolRows(torch.autograd.Function):
    """_TopkPool over any number of tokens: tokens [rows, Kt, D] (constants), query m pools row row_of[m] (None: row m).  wg_pool_rows_bf16 / _bwd."""

    @staticmethod
    def forward(ctx, u, tokens, row_of):
        u, tokens = u.contiguous(), tokens.contiguous()
        row_of = None if row_of is None else row_of.to(torch.int32).contiguous()
        ctx.save_for_backward(u, tokens, row_of)
        _, Kt, D = tokens.shape
        M = u.shape[0]
        v = torch.empty(M, D, device=u.device, dtype=BF16)
        _lib.check(_lib.lib().wg_pool_rows_bf16(u.data_ptr(), tokens.data_ptr(), None if row_of is None else row_of.data_ptr(), v.data_ptr(), M, Kt, D,
                                                ops._stream()), "wg_pool_rows_bf16")
        return v

    @staticmethod
    def backward(ctx, dv):
        u, tokens, row_of = ctx.saved_tensors
        _, Kt, D = tokens.shape
        M = u.shape[0]
        dv = dv.contiguous()
        du = torch.empty_like(u)
        _lib.check(_lib.lib().wg_pool_rows_bwd_bf16(u.data_ptr(), tokens.data_ptr(), None if row_of is None else row_of.data_ptr(), dv.data_ptr(),
                                                    du.data_ptr(), M, Kt, D, ops._stream()), "wg_pool_rows_bwd_bf16")
        return du, None, None


def topk_pool(u, kt):
    """kt [M, Kt, D]: the selected tokens of every query (Kt <= 16: the wave-per-query kernel; more: the streaming one)."""
    return _TopkPool.apply(u, kt) if kt.shape[1] <= 16 else _PoolRows.apply(u, kt, None)


def pool_rows(u, tokens, row_of):
    """softmax_k(u_m . tokens[row_of[m], k] / sqrt(D)) pooling of whole rows: u [M, D], tokens [rows, Kt, D] constants -> [M, D] bf16."""
    return _PoolRows.apply(u, tokens, row_of)


def nce_tail(z, vp, sim, own_row, rows, N, temperature, exclude_same_row):
    return _NceTail.apply(z, vp, sim, own_row, rows, N, temperature, exclude_same_row)


class _HyperRows(torch.autograd.Function):
    """masks = hyper_in @ upscaled (mask_decoder.py:150-160) for all prompts at once on channels-last rows: wg_hyper_rows_f32 / _bwd_f32."""

    @staticmethod
    def forward(ctx, up, hyper):
        up, hyper = up.contiguous(), hyper.contiguous()
        ctx.save_for_backward(up, hyper)
        P, HW, C = up.shape
        K = hyper.shape[1]
        masks = torch.empty(P, K, HW, device=up.device, dtype=torch.float32)
        _lib.check(_lib.lib().wg_hyper_rows_f32(up.data_ptr(), hyper.data_ptr(), masks.data_ptr(), P, HW, C, K, ops._stream()), "wg_hyper_rows_f32")
        return masks

    @staticmethod
    def backward(ctx, dm):
        up, hyper = ctx.saved_tensors
        P, HW, C = up.shape
        K = hyper.shape[1]
        dm = dm.contiguous().float()
        dup = torch.empty_like(up)
        dh = torch.empty(P, K, C, device=up.device, dtype=torch.float32)
        L = _lib.lib()
        nws = L.wg_hyper_rows_bwd_workspace_floats(P, HW, K)
        ws = torch.empty(nws, device=up.device, dtype=torch.float32)
        rc = L.wg_hyper_rows_bwd_f32(up.data_ptr(), hyper.data_ptr(), dm.data_ptr(), dup.data_ptr(), dh.data_ptr(), ws.data_ptr(), nws, P, HW, C, K, ops._stream())
        _lib.check(rc, "wg_hyper_rows_bwd_f32")
        return dup, dh.to(hyper.dtype)


def hyper_rows(up, hyper):
    """up [P, HW, 32] bf16, hyper [P, K <= 4, 32] bf16 -> masks [P, K, HW] fp32, differentiable in both."""
    return _HyperRows.apply(up, hyper)
