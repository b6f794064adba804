"""LayerNorm + GEMM as two kernels vs the folded form (ops.ln_linear), on the workload's four LN->GEMM shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from walkgpt_amd import ops

dev = "cuda:0"
for name, M, N, K, act in [("sam qkv", 32768, 2304, 768, 0), ("sam lin1", 32768, 3072, 768, 1), ("clip qkv", 8200, 3072, 1024, 0),
                           ("clip fc1", 8200, 4096, 1024, 2)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    gm, bt = torch.randn(K, device=dev).bfloat16(), torch.randn(K, device=dev).bfloat16()
    w, b = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(), torch.randn(N, device=dev).bfloat16()
    fold = ops.fold_layernorm(gm, bt, w, b)
    res = {}
    for fuse in (False, True):
        ops.LN_FUSE = fuse
        for _ in range(5):
            ops.ln_linear(x, fold, 1e-6, act=act)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            ops.ln_linear(x, fold, 1e-6, act=act)
        e.record()
        torch.cuda.synchronize()
        res[fuse] = s.elapsed_time(e) / 50 * 1e3
    print("%-9s M=%5d N=%4d K=%4d  LN+GEMM %7.1f us   folded %7.1f us   (%.2fx)" % (name, M, N, K, res[False], res[True], res[False] / res[True]), flush=True)
