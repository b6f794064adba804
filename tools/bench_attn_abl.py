"""Diagnostic: time the SAM global attention kernel with an ablation build of the library (WG_ABL_LIB=path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import _lib
if os.environ.get("WG_ABL_LIB"):
    _lib.LIB_PATH = os.environ["WG_ABL_LIB"]
from walkgpt_amd import ops
dev = torch.device("cuda:0")
B, heads, hd, grid = 8, 12, 64, 64
D = heads * hd
qkv = torch.randn(B * grid * grid, 3 * D, device=dev).to(torch.bfloat16)
qb = torch.randn(3 * D, device=dev).to(torch.bfloat16)
out = torch.empty(B * grid * grid, D, device=dev, dtype=torch.bfloat16)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for win in (64, 14):
    rh = (torch.randn(2 * win - 1, hd, device=dev) * 0.1).to(torch.bfloat16); rw = (torch.randn(2 * win - 1, hd, device=dev) * 0.1).to(torch.bfloat16)
    print(os.environ.get("WG_ABL_LIB", "normal"), "window", win, "%.1f us" % (1e3 * t(lambda: ops.sam_attention(qkv, qb, rh, rw, B, grid, win, heads, out=out))), flush=True)
q = torch.randn(B, 1025, 3 * 1024, device=dev).to(torch.bfloat16)
o2 = torch.empty(B, 1025, 1024, device=dev, dtype=torch.bfloat16)
print(os.environ.get("WG_ABL_LIB", "normal"), "clip %.1f us" % (1e3 * t(lambda: ops.mha(q[..., :1024], q[..., 1024:2048], q[..., 2048:], 16, 0.125, out=o2))), flush=True)
