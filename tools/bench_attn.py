"""Attention kernels alone on the C2 shapes (random data): SAM global / windowed (rel-pos), CLIP 1025-key."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops, _lib
if os.environ.get("WG_LIB"):      # A/B against another build of the library in the same call (same box)
    _lib.LIB_PATH = os.path.abspath(os.environ["WG_LIB"])
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 8
for (name, heads, hd, grid, win) in [("sam-b global", 12, 64, 64, 64), ("sam-b window", 12, 64, 64, 14), ("sam-h global", 16, 80, 64, 64), ("sam-h window", 16, 80, 64, 14)]:
    D = heads * hd
    qkv = torch.randn(B * grid * grid, 3 * D, device=dev).to(torch.bfloat16)
    qb = torch.randn(3 * D, device=dev).to(torch.bfloat16)
    rh = (torch.randn(2 * win - 1, hd, device=dev) * 0.1).to(torch.bfloat16); rw = (torch.randn(2 * win - 1, hd, device=dev) * 0.1).to(torch.bfloat16)
    out = torch.empty(B * grid * grid, D, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: ops.sam_attention(qkv, qb, rh, rw, B, grid, win, heads, out=out))
    nw = (grid + win - 1) // win
    keys = win * win
    fl = 4.0 * B * heads * (nw * nw) * keys * keys * hd
    print("%-14s %.1f us  %.0f TF (useful flops)" % (name, ms * 1e3, fl / ms / 1e9), flush=True)
for (name, heads, hd, L) in [("clip 1025", 16, 64, 1025), ("plain 4096", 16, 64, 4096), ("plain 4096 hd128", 8, 128, 4096)]:
    D = heads * hd
    qkv = torch.randn(B, L, 3 * D, device=dev).to(torch.bfloat16)
    out = torch.empty(B, L, D, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: ops.mha(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], heads, hd ** -0.5, out=out))
    print("%-14s %.1f us  %.0f TF" % (name, ms * 1e3, 4.0 * B * heads * L * L * hd / ms / 1e9), flush=True)
    kb = torch.zeros(B, L, device=dev)
    ms = t(lambda: ops.mha(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], heads, hd ** -0.5, key_bias=kb, out=out))
    print("%-14s %.1f us  %.0f TF (key bias)" % (name, ms * 1e3, 4.0 * B * heads * L * L * hd / ms / 1e9), flush=True)
