"""GEMM epilogue / tile ablation on hot-path shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(32768, 2304, 768), (32768, 3072, 768), (32768, 768, 3072), (32768, 768, 768), (8200, 3072, 1024), (8200, 4096, 1024), (8200, 1024, 4096), (8200, 1024, 1024)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for tile in (1, 2):
        res = []
        for name, kw in [("none", {}), ("bias", dict(bias=b)), ("gelu", dict(bias=b, act=ops.ACT_GELU)), ("qgelu", dict(bias=b, act=ops.ACT_QUICK_GELU)), ("resid", dict(bias=b, residual=r))]:
            ms = t(lambda: ops.linear(a, w, out=out, tile=tile, **kw))
            res.append("%s %.3fms %4.0fTF" % (name, ms, 2.0 * M * N * K / ms / 1e9))
        print("M=%d N=%d K=%d tile=%d | %s" % (M, N, K, 128 * tile, " | ".join(res)), flush=True)
