"""Thin tensor-level wrappers over the C-ABI.  torch is plumbing only here: it owns the HBM buffers and the
stream; every arithmetic step is a wg_* call.  All wrappers raise on non-GPU tensors -- there is no CPU path."""
import torch

from . import _lib

ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_RELU = 0, 1, 2, 3
_BF16 = torch.bfloat16


def _stream():
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


def linear(x, weight, bias=None, act=ACT_NONE, residual=None, res_row_mod=0, out=None, out_f32=False, tile=0):
    """y = act(x @ weight.T + bias) (+ residual).  x [..., K] bf16, weight [N, K] bf16."""
    _need_gpu(x, weight, bias, residual, out)
    assert x.dtype == _BF16 and weight.dtype == _BF16
    M, K, lda = _rows(x)
    N, K2 = weight.shape
    assert K2 == K and weight.stride(1) == 1
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
    rc = _lib.lib().wg_gemm_bias_act_bf16(x.data_ptr(), lda, weight.data_ptr(), weight.stride(0), _ptr(bias),
                                          _ptr(residual), ldr, res_row_mod, out.data_ptr(), ldc, M, N, K, act,
                                          1 if out.dtype == torch.float32 else 0, tile, _stream())
    _lib.check(rc, "wg_gemm_bias_act_bf16")
    return out


def layernorm(x, gamma, beta, eps, act=ACT_NONE, out=None):
    _need_gpu(x, gamma, beta, out)
    assert x.dtype == _BF16 and gamma.dtype == _BF16 and beta.dtype == _BF16
    M, D, ldx = _rows(x)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=_BF16)
    _, _, ldy = _rows(out)
    rc = _lib.lib().wg_layernorm_rows(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), ldy, M, D,
                                      float(eps), act, _stream())
    _lib.check(rc, "wg_layernorm_rows")
    return out
