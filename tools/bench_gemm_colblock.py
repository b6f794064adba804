"""Tile-order column blocks of the persistent GEMM on given shapes: one process per value (the override is read once):
   for cb in 0 2 3 4 5 8; do WG_GEMM_COLBLOCK=$cb python tools/bench_gemm_colblock.py vith; done"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
sets = {"vith": [("H qkv", 131072, 3840, 1280), ("H lin1", 131072, 5120, 1280), ("H proj", 131072, 1280, 1280), ("H lin2", 131072, 1280, 5120)],
        "vitb": [("B qkv", 32768, 2304, 768), ("B lin1", 32768, 3072, 768), ("B lin2", 32768, 768, 3072)],
        "clip": [("C qkv", 8200, 3072, 1024), ("C fc1", 8200, 4096, 1024), ("C fc2", 8200, 1024, 4096)]}
res = []
for (name, M, N, K) in sets[sys.argv[1] if len(sys.argv) > 1 else "vith"]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.linear(a, w, b, out=out, tile=16)
    best = 1e9
    for r in range(4):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    res.append("%s %.0f us (%.0f TF)" % (name, best, 2.0 * M * N * K / best / 1e6))
    del a, w, out
print("col_block", os.environ.get("WG_GEMM_COLBLOCK", "default"), "|", " | ".join(res), flush=True)
