"""The step's narrow-N GEMMs (one or three tile columns: every A panel comes from beyond L2 for one tile) on the persistent kernel, for same-box A/B of
library variants, round-robin inside one process per library:  WG_LIB=... python tools/bench_gemm_narrow.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
shapes = [("neck 3x3", 32768, 256, 2304), ("neck 1x1", 32768, 256, 768), ("patch embed", 32768, 768, 768), ("sam lin2", 32768, 768, 3072), ("clip proj-like", 8200, 1024, 1024)]
ops_ = []
for (name, M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops_.append((name, (lambda a=a, w=w, b=b, out=out: ops.linear(a, w, b, out=out, tile=16))))
best = {n: 1e9 for n, _ in ops_}
for r in range(5):
    for n, f in ops_:
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best[n] = min(best[n], e0.elapsed_time(e1) / 20 * 1e3)
print(os.environ.get("WG_LIB", "product"), "|", " | ".join("%s %.1f us" % (n, best[n]) for n, _ in ops_), flush=True)
