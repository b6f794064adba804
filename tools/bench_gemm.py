"""Micro-benchmark of wg_gemm_bias_act_bf16 on the hot-path shapes (random data), both tile configs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops

dev = torch.device("cuda:0")
shapes = [(32768, 2304, 768), (32768, 768, 768), (32768, 3072, 768), (32768, 768, 3072),
          (8200, 3072, 1024), (8200, 1024, 1024), (8200, 4096, 1024), (8200, 1024, 4096), (4096, 4096, 4096), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for tile in (1, 2):
        for _ in range(3):
            ops.linear(a, w, b, act=ops.ACT_GELU, out=out, tile=tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for _ in range(n):
            ops.linear(a, w, b, act=ops.ACT_GELU, out=out, tile=tile)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("M=%d N=%d K=%d tile=%d  %.3f ms  %.1f TFLOP/s" % (M, N, K, 128 * tile, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
