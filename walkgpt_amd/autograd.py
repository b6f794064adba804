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


def transpose2d(t):
    """[R, C] bf16 contiguous -> [C, R] (wg_tokens_to_nchw_bf16 with one batch)."""
    R, C = t.shape
    return ops.tokens_to_nchw(t.contiguous(), 1, R, C).view(C, R)


def colsum(t):
    """[R, C] bf16 -> fp32 [C] column sums (wg_colsum_f32)."""
    R, C = t.shape
    if C % 8 or t.stride(0) % 8 or t.data_ptr() % 16:      # the kernel reads 16-byte pieces: a handful of columns (IoU head: 4) go padded
        Cp = (C + 7) // 8 * 8
        tp = torch.zeros(R, Cp, device=t.device, dtype=t.dtype)
        tp[:, :C] = t
        return colsum(tp)[:C]
    out = torch.zeros(C, device=t.device, dtype=torch.float32)
    _lib.check(_lib.lib().wg_colsum_f32(t.data_ptr(), t.stride(0), out.data_ptr(), R, C, ops._stream()), "wg_colsum_f32")
    return out


class _Linear(torch.autograd.Function):
    """y = x W^T + b.  Backward: dX = dY W, dW = dY^T X (both on the bf16 MFMA GEMM, operands transposed by wg_tokens_to_nchw_bf16, the
    reduction dimension padded to whole slabs; dW accumulated and returned in fp32 -> parameter dtype), db = column sums of dY."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        with torch.no_grad():
            return ops.linear(x.contiguous(), weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, K = weight.shape
        dy2 = dy.contiguous().view(-1, N)
        x2 = x.contiguous().view(-1, K)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear(dy2, transpose2d(weight.detach())).view(x.shape)                     # [M, N] @ [N, K]
        if ctx.needs_input_grad[1]:
            dyt, xt = transpose2d(_pad_rows(dy2)), transpose2d(_pad_rows(x2))                    # [N, Mp], [K, Mp]
            dw = ops.linear(dyt, xt, out_f32=True).to(weight.dtype)                              # [N, K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy2).to(weight.dtype)
        return dx, dw, db


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
        dg = torch.zeros(C, device=x.device, dtype=torch.float32)
        db = torch.zeros(C, device=x.device, dtype=torch.float32)
        rc = _lib.lib().wg_layernorm_bwd_bf16(x.data_ptr(), C, gamma.data_ptr(), dy.data_ptr(), C, dx.data_ptr(), C, dg.data_ptr(), db.data_ptr(),
                                              M, C, float(ctx.eps), ops._stream())
        _lib.check(rc, "wg_layernorm_bwd_bf16")
        return dx, dg.to(gamma.dtype), db.to(gamma.dtype), None


def linear(x, weight, bias=None, act=ops.ACT_NONE):
    """Differentiable ops.linear (+ a separate activation operator when one is asked for)."""
    ops._need_gpu(x, weight, bias)
    y = _Linear.apply(x, weight, bias)
    return y if act == ops.ACT_NONE else _Act.apply(y, act)


def activation(x, act):
    return x if act == ops.ACT_NONE else _Act.apply(x, act)


def layernorm(x, gamma, beta, eps):
    ops._need_gpu(x, gamma, beta)
    return _LayerNorm.apply(x, gamma, beta, eps)
