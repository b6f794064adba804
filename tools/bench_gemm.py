"""A/B of the GEMM tile/pipeline variants on the hot-path shapes with their real epilogues (random data)."""
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
NAMES = {1: "128", 11: "128persist", 2: "256", 14: "256pp", 16: "256pp-persist"}
shapes = [("sam qkv", 32768, 2304, 768, "bias"), ("sam proj", 32768, 768, 768, "resid"), ("sam lin1", 32768, 3072, 768, "gelu"),
          ("sam lin2", 32768, 768, 3072, "resid"), ("clip qkv", 8200, 3072, 1024, "bias"), ("clip out", 8200, 1024, 1024, "resid"),
          ("clip fc1", 8200, 4096, 1024, "qgelu"), ("clip fc2", 8200, 1024, 4096, "resid"), ("8k", 8192, 8192, 8192, "none"),
          ("4k", 4096, 4096, 4096, "none"), ("sam qkv", 32768, 2304, 768, "none"), ("sam lin1", 32768, 3072, 768, "none"),
          ("sam lin1", 32768, 3072, 768, "bias"), ("sam lin2", 32768, 768, 3072, "none")]
# correctness of every variant first
a = torch.randn(1777, 256, device=dev).to(torch.bfloat16); w = (torch.randn(520, 256, device=dev) / 16).to(torch.bfloat16)
b = torch.randn(520, device=dev).to(torch.bfloat16); r = torch.randn(1777, 520, device=dev).to(torch.bfloat16)
ref = torch.nn.functional.gelu(a.float() @ w.float().t() + b.float()) + r.float()
for tile in NAMES:
    out = ops.linear(a, w, b, act=ops.ACT_GELU, residual=r, tile=tile)
    print("tile", NAMES[tile], "max err", (out.float() - ref).abs().max().item(), flush=True)
for (name, M, N, K, epi) in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = {"none": {}, "bias": dict(bias=b), "gelu": dict(bias=b, act=ops.ACT_GELU), "qgelu": dict(bias=b, act=ops.ACT_QUICK_GELU), "resid": dict(bias=b, residual=r)}[epi]
    res = []
    for tile in NAMES:
        ms = t(lambda: ops.linear(a, w, out=out, tile=tile, **kw))
        res.append("%s %.0f" % (NAMES[tile], 2.0 * M * N * K / ms / 1e9))
    print("%-9s M=%d N=%d K=%d %-5s | %s" % (name, M, N, K, epi, " | ".join(res)), flush=True)
