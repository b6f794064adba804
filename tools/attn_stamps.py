"""Diagnostic: phase timeline of the attention main loop (build with WG_EXTRA_HIPCC_FLAGS=-DWG_ATTN_STAMP)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops, _lib
dev = torch.device("cuda:0")
B, heads, hd, grid, win = 8, 12, 64, 64, 64
D = heads * hd
qkv = torch.randn(B * grid * grid, 3 * D, device=dev).to(torch.bfloat16)
qb = torch.randn(3 * D, device=dev).to(torch.bfloat16)
rh = (torch.randn(2 * win - 1, hd, device=dev) * 0.1).to(torch.bfloat16); rw = (torch.randn(2 * win - 1, hd, device=dev) * 0.1).to(torch.bfloat16)
out = torch.empty(B * grid * grid, D, device=dev, dtype=torch.bfloat16)
buf = torch.zeros(8 * 12 * 8, device=dev, dtype=torch.int32)
lib = _lib.lib()
lib.wg_debug_attn_stamps.argtypes = [ctypes.c_void_p]
assert lib.wg_debug_attn_stamps(buf.data_ptr()) == 0
for _ in range(20):
    ops.sam_attention(qkv, qb, rh, rw, B, grid, win, heads, out=out)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype("int64").reshape(8, 12, 8) & 0xffffffff
base = s[0, 2, 0]
for w in (0, 4, 1, 7):
    print("wave", w)
    for t in range(2, 10):
        r = s[w, t, :5] - base
        print("  t=%d start %6d | stage+block1 %5d  block2 %5d  max/rescale %5d  vmcnt+barrier %5d | next %5d" % (
            t, r[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], (s[w, t + 1, 0] - base) - r[4]))
