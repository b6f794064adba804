"""Diagnostic: where a window-attention workgroup spends its life (build with WG_EXTRA_HIPCC_FLAGS=-DWG_ATTN_STAMP):
kernel entry -> operands requested / rel-pos tables built -> first tile landed -> loop done -> output stored, in s_memtime units (100 MHz)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.lib()
lib.wg_debug_attn_stamps.argtypes = [ctypes.c_void_p]
for (name, heads, hd, nwav) in (("vit-b windows", 12, 64, 4), ("vit-h windows", 16, 80, 7)):
    B, grid, win = 8, 64, 14
    D = heads * hd
    qkv = torch.randn(B * grid * grid, 3 * D, device=dev).to(torch.bfloat16)
    qb = torch.randn(3 * D, device=dev).to(torch.bfloat16)
    rh = (torch.randn(2 * win - 1, hd, device=dev) * 0.1).to(torch.bfloat16); rw = (torch.randn(2 * win - 1, hd, device=dev) * 0.1).to(torch.bfloat16)
    out = torch.empty(B * grid * grid, D, device=dev, dtype=torch.bfloat16)
    buf = torch.zeros(8 * 12 * 8, device=dev, dtype=torch.int32)
    assert lib.wg_debug_attn_stamps(buf.data_ptr()) == 0
    for _ in range(5):
        ops.sam_attention(qkv, qb, rh, rw, B, grid, win, heads, out=out)
    torch.cuda.synchronize()
    s = buf.cpu().numpy().astype("int64").reshape(8, 12, 8) & 0xffffffff
    print(name)
    for w in range(nwav):
        k = s[w, 11, :5] - s[w, 11, 0]
        tiles = [int(s[w, t, 4] - s[w, t, 0]) for t in range(4)]
        print("  wave %d: tables built +%d | first tile landed +%d | loop +%d (tiles %s) | stored +%d   [x10 ns]" % (w, k[1], k[2] - k[1], k[3] - k[2], tiles, k[4] - k[3]))
