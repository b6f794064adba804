# fp8 GEMM timing: row scales vs MX operand vs MX output, SAM ViT-H / ViT-B / CLIP MLP shapes
import torch, sys
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
for (M, D, H) in [(32768, 1280, 5120), (32768, 768, 3072), (8200, 1024, 4096)]:
    x = torch.randn(M, D, device=dev).bfloat16()
    w1 = (torch.randn(H, D, device=dev) / D ** 0.5).bfloat16(); w2 = (torch.randn(D, H, device=dev) / H ** 0.5).bfloat16()
    b1 = torch.randn(H, device=dev).bfloat16(); b2 = torch.randn(D, device=dev).bfloat16()
    xq, xs = ops.quantize_rows_fp8(x); w1q, w1s = ops.quantize_weight_fp8(w1); w2q, w2s = ops.quantize_weight_fp8(w2)
    h = ops.linear_fp8(xq, xs, w1q, w1s, bias=b1, act=1)
    hq, hs = ops.quantize_rows_fp8(h)
    mq, ms = ops.linear_fp8(xq, xs, w1q, w1s, bias=b1, act=1, mx_out=True)
    f1, f2 = 2.0 * M * D * H, 2.0 * M * D * H
    a = t(lambda: ops.linear_fp8(xq, xs, w1q, w1s, bias=b1, act=1))
    b = t(lambda: ops.linear_fp8(xq, xs, w1q, w1s, bias=b1, act=1, mx_out=True))
    c = t(lambda: ops.quantize_rows_fp8(h))
    d = t(lambda: ops.linear_fp8(hq, hs, w2q, w2s, bias=b2, residual=x))
    e = t(lambda: ops.linear_fp8(mq, ms, w2q, w2s, bias=b2, residual=x))
    print("M=%d D=%d H=%d: lin1 bf16-out %.1f us (%.0f TF/s) | lin1 mx-out %.1f us (%.0f) | quantise %.1f us | lin2 row %.1f us (%.0f) | lin2 mx %.1f us (%.0f)"
          % (M, D, H, a, f1 / a / 1e6, b, f1 / b / 1e6, c, d, f2 / d / 1e6, e, f2 / e / 1e6), flush=True)

# persistent MX GEMM (block scales on both operands) against the tile kernel, ViT-H / ViT-B / CLIP shapes
print("persistent mxfp8 vs tile fp8 (no activation, residual on the square / down projections)")
for (M, N, K, res) in [(32768, 3840, 1280, 0), (32768, 1280, 1280, 1), (32768, 5120, 1280, 0), (32768, 1280, 5120, 1),
                       (32768, 2304, 768, 0), (32768, 768, 768, 1), (32768, 3072, 768, 0), (32768, 768, 3072, 1),
                       (8200, 3072, 1024, 0), (8200, 1024, 1024, 1), (8200, 4096, 1024, 0), (8200, 1024, 4096, 1)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev).bfloat16(); r = torch.randn(M, N, device=dev).bfloat16() if res else None
    xq, xs = ops.quantize_rows_fp8(x); wq, ws = ops.quantize_weight_fp8(w)
    xq2, xm = ops.quantize_mx_fp8(x); wd = ops.mx_weight(w)
    fl = 2.0 * M * N * K
    a = t(lambda: ops.linear_fp8(xq, xs, wq, ws, bias=b, residual=r))
    c = t(lambda: ops.linear_mxfp8((xq2, xm), wd, bias=b, residual=r))
    d = t(lambda: ops.linear(x, w, b, residual=r))
    print("M=%d N=%d K=%d: tile fp8 %.1f us (%.0f TF/s) | persistent mxfp8 %.1f us (%.0f TF/s) | bf16 %.1f us (%.0f)" % (M, N, K, a, fl / a / 1e6, c, fl / c / 1e6, d, fl / d / 1e6), flush=True)
