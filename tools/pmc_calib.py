"""rocprofv3 FETCH_SIZE experiments for the GEMM's LDS-DMA access pattern (run under `rocprofv3 --pmc FETCH_SIZE`).
A GEMM with ONE tile column reads A exactly once: the calibration point (raw counter x 1024 = half the bytes on gfx950).
Wider N at the same M, K shows which operand is re-read as the tile grid grows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
big = torch.empty(600 * 1024 * 1024, device=dev, dtype=torch.uint8)
for (M, N, K) in [(32768, 256, 768), (32768, 512, 768), (32768, 1024, 768), (32768, 2304, 768), (8192, 2304, 768), (32768, 256, 3072)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(2):
        big.fill_(1)                       # flush the 256 MiB Infinity Cache between launches
        ops.linear(a, w, out=out, tile=14)
    torch.cuda.synchronize()
    print("M=%d N=%d K=%d: A = %.1f MB, W = %.2f MB" % (M, N, K, M * K * 2 / 1e6, N * K * 2 / 1e6), flush=True)
