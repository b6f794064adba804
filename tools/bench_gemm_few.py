"""A few GEMM shapes on the persistent kernel (tile 16), for same-box A/B of library variants: WG_LIB=... python tools/bench_gemm_few.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [("sam qkv", 32768, 2304, 768), ("sam proj", 32768, 768, 768), ("sam lin2", 32768, 768, 3072), ("clip-like fc2", 8192, 1024, 4096), ("4k", 4096, 4096, 4096), ("8k", 8192, 8192, 8192)]
res = []
for (name, M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: ops.linear(a, w, b, out=out, tile=16))
    res.append("%s %.0f" % (name, 2.0 * M * N * K / ms / 1e9))
print(os.environ.get("WG_LIB", "product"), "|", " | ".join(res), flush=True)
