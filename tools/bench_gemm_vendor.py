"""Ceiling check (tools only, never the product path): vendor GEMM via torch (hipBLASLt/rocBLAS) vs wg_gemm on the
hot-path shapes, same process, same random data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
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
shapes = [("sam qkv", 32768, 2304, 768), ("sam proj", 32768, 768, 768), ("sam lin1", 32768, 3072, 768), ("sam lin2", 32768, 768, 3072),
          ("clip qkv", 8200, 3072, 1024), ("clip out", 8200, 1024, 1024), ("clip fc1", 8200, 4096, 1024), ("clip fc2", 8200, 1024, 4096),
          ("4k", 4096, 4096, 4096), ("8k", 8192, 8192, 8192)]
for (name, M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K / 1e9
    v1 = t(lambda: torch.matmul(a, w.t(), out=out))
    v2 = t(lambda: F.linear(a, w, b))
    m0 = t(lambda: ops.linear(a, w, out=out))
    m1 = t(lambda: ops.linear(a, w, b, out=out))
    print("%-9s M=%d N=%d K=%d | vendor matmul %.0f TF, linear+bias %.0f TF | wg none %.0f TF, bias %.0f TF" % (name, M, N, K, fl / v1, fl / v2, fl / m0, fl / m1), flush=True)
