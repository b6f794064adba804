"""Diagnostic: half-phase timeline of the ping-pong GEMM loop (build with WG_EXTRA_HIPCC_FLAGS=-DWG_GEMM_STAMP)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops, _lib
dev = torch.device("cuda:0")
M, N, K = 32768, 2304, 768
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
buf = torch.zeros(1024, device=dev, dtype=torch.int32)
lib = _lib.lib()
lib.wg_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
assert lib.wg_debug_gemm_stamps(buf.data_ptr()) == 0
for _ in range(20):
    ops.linear(a, w, b, out=out, tile=14)
torch.cuda.synchronize()
raw = buf.cpu().numpy().astype("int64") & 0xffffffff
s = raw[:512].reshape(2, 8, 4, 8)
for blk in (0, 1):
    for g in (0, 1):
        t = raw[512 + blk * 16 + g * 8: 512 + blk * 16 + g * 8 + 4]
        print("tile stamps block %s group %d: prologue %d  main loop %d  epilogue (incl. store drain) %d  total %d" % ([0, 1024][blk], g, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[3] - t[0]))
base = s[0, 0, 0, 0]
for g in (0, 1):
    print("group", g, "(wave %d)" % (4 * g))
    for kt in range(1, 6):
        for c in range(2):
            r = s[g, kt, c, :5] - base
            print("  slab %d c=%d  start %6d | M: issue+wait %4d  barrier %4d | C: mfma issue %4d  barrier %4d" % (kt + 2, c, r[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3]))
