"""Calibration of rocprofv3 FETCH_SIZE for this kernel's LDS-DMA access pattern: a GEMM with ONE tile column reads A exactly once."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(32768, 256, 768), (65536, 256, 1024), (32768, 256, 3072)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    big = torch.empty(600 * 1024 * 1024, device=dev, dtype=torch.uint8)
    for _ in range(3):
        big.fill_(1)                       # flush the 256 MiB Infinity Cache between launches
        ops.linear(a, w, out=out, tile=14)
    torch.cuda.synchronize()
    print("M=%d N=%d K=%d: A = %.1f MB, W = %.2f MB" % (M, N, K, M * K * 2 / 1e6, N * K * 2 / 1e6), flush=True)
