import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
for (M, N, K, tile) in [(32768, 2304, 768, 14), (32768, 3072, 768, 14), (8200, 4096, 1024, 14), (8200, 3072, 1024, 14), (32768, 768, 3072, 14)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(a, w, out=out, tile=tile)
    torch.cuda.synchronize()
