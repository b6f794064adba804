# fp8 GEMM timing on the encoders' shapes (SAM ViT-H / ViT-B, CLIP ViT-L): the persistent MX kernel against the row-scale tile kernel and
# the bf16 persistent kernel, plain and with the epilogues the MX chain uses (LayerNorm fold + GELU + e4m3-only output; residual + e4m3 copy
# + row partials).
import sys
import torch
sys.path.insert(0, '/root/repo')
from walkgpt_amd import ops
dev = torch.device('cuda:0')
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("plain (bias; residual on the square / down projections)")
for (M, N, K, res) in [(32768, 3840, 1280, 0), (32768, 1280, 1280, 1), (32768, 5120, 1280, 0), (32768, 1280, 5120, 1),
                       (32768, 2304, 768, 0), (32768, 768, 768, 1), (32768, 3072, 768, 0), (32768, 768, 3072, 1),
                       (8200, 3072, 1024, 0), (8200, 1024, 1024, 1), (8200, 4096, 1024, 0), (8200, 1024, 4096, 1)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev).bfloat16(); r = torch.randn(M, N, device=dev).bfloat16() if res else None
    xq, xs = ops.quantize_rows_fp8(x); wq, ws = ops.quantize_weight_fp8(w)
    xm = ops.quantize_mx_fp8(x); wd = ops.mx_weight(w)
    fl = 2.0 * M * N * K
    a = t(lambda: ops.linear_fp8(xq, xs, wq, ws, bias=b, residual=r))
    c = t(lambda: ops.linear_mxfp8(xm, wd, bias=b, residual=r))
    d = t(lambda: ops.linear(x, w, b, residual=r))
    print("M=%d N=%d K=%d: tile fp8 %.1f us (%.0f TF/s) | persistent mxfp8 %.1f us (%.0f TF/s) | bf16 %.1f us (%.0f)" % (M, N, K, a, fl / a / 1e6, c, fl / c / 1e6, d, fl / d / 1e6), flush=True)
print("MLP of a block as the MX chain runs it: lin1 = LayerNorm fold + GELU + e4m3-only output, lin2 = residual + e4m3 copy + row partials")
for (M, D, H) in [(32768, 1280, 5120), (32768, 768, 3072), (8200, 1024, 4096)]:
    x = torch.randn(M, D, device=dev).bfloat16()
    w1 = (torch.randn(H, D, device=dev) / D ** 0.5).bfloat16(); w2 = (torch.randn(D, H, device=dev) / H ** 0.5).bfloat16()
    b1 = torch.randn(H, device=dev).bfloat16(); b2 = torch.randn(D, device=dev).bfloat16()
    g = torch.ones(D, device=dev).bfloat16(); be = torch.zeros(D, device=dev).bfloat16()
    ops.mx_prepare_rows(x)
    f1 = ops.fold_layernorm_mx(g, be, w1, b1); wd2 = ops.mx_weight(w2); wd1 = ops.mx_weight(w1)
    h = ops.linear_mxfp8(x, f1, act=1, ln_eps=1e-6, mx_out=True, bf16_out=False)
    fl = 2.0 * M * D * H
    a = t(lambda: ops.linear_mxfp8(x._wg_mx, wd1, bias=b1))
    b = t(lambda: ops.linear_mxfp8(x._wg_mx, wd1, bias=b1, act=1))
    c = t(lambda: ops.linear_mxfp8(x, f1, act=1, ln_eps=1e-6, mx_out=True, bf16_out=False))
    d = t(lambda: ops.linear_mxfp8(h, wd2, bias=b2, residual=x))
    e = t(lambda: ops.linear_mxfp8(h, wd2, bias=b2, residual=x, mx_out=True, row_partials=True))
    print("M=%d D=%d H=%d: lin1 plain %.1f us (%.0f TF/s) | + GELU %.1f (%.0f) | LN fold + GELU + e4m3 out %.1f (%.0f) || lin2 + residual %.1f (%.0f) | + e4m3 copy + partials %.1f (%.0f)"
          % (M, D, H, a, fl / a / 1e6, b, fl / b / 1e6, c, fl / c / 1e6, d, fl / d / 1e6, e, fl / e / 1e6), flush=True)
