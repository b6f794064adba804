import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
B, heads, hd, grid = 8, 12, 64, 64
D = heads * hd
qkv = torch.randn(B * grid * grid, 3 * D, device=dev).to(torch.bfloat16)
qb = torch.randn(3 * D, device=dev).to(torch.bfloat16)
out = torch.empty(B * grid * grid, D, device=dev, dtype=torch.bfloat16)
for win in (64, 14):
    rh = (torch.randn(2 * win - 1, hd, device=dev) * 0.1).to(torch.bfloat16); rw = (torch.randn(2 * win - 1, hd, device=dev) * 0.1).to(torch.bfloat16)
    for _ in range(3):
        ops.sam_attention(qkv, qb, rh, rw, B, grid, win, heads, out=out)
q = torch.randn(B, 1025, 3 * 1024, device=dev).to(torch.bfloat16)
o2 = torch.empty(B, 1025, 1024, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.mha(q[..., :1024], q[..., 1024:2048], q[..., 2048:], 16, 0.125, out=o2)
torch.cuda.synchronize()
