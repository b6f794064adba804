"""Diagnostic: timeline of the pipelined attention loop (a library built with -DWG_ATTN_STAMP: tools/build_variant.py stamp -DWG_ATTN_STAMP,
run with WG_LIB=walkgpt_amd/_abl/lib_stamp.so).  Prints, per wave and iteration, the cycles (s_memtime) of: head (tile requests, LDS issue, maximum,
rescale decision) | main stream (16 MFMAs + exponentials) | wait for the requested tiles | barrier."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops, _lib
if os.environ.get("WG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["WG_LIB"])
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "global"
B, heads, hd, grid = 8, 12, 64, 64
D = heads * hd
buf = torch.zeros(8 * 14 * 8, device=dev, dtype=torch.int32)
lib = _lib.lib()
lib.wg_debug_attn_pipe_stamps.argtypes = [ctypes.c_void_p]
assert lib.wg_debug_attn_pipe_stamps(buf.data_ptr()) == 0
if mode == "global":
    qkv = torch.randn(B * grid * grid, 3 * D, device=dev).to(torch.bfloat16)
    qb = torch.randn(3 * D, device=dev).to(torch.bfloat16)
    rh = (torch.randn(127, hd, device=dev) * 0.1).to(torch.bfloat16); rw = (torch.randn(127, hd, device=dev) * 0.1).to(torch.bfloat16)
    out = torch.empty(B * grid * grid, D, device=dev, dtype=torch.bfloat16)
    run = lambda: ops.sam_attention(qkv, qb, rh, rw, B, grid, grid, heads, out=out)
else:
    heads = 16; D = heads * hd
    qkv = torch.randn(B, 1025, 3 * D, device=dev).to(torch.bfloat16)
    out = torch.empty(B, 1025, D, device=dev, dtype=torch.bfloat16)
    run = lambda: ops.mha(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], heads, hd ** -0.5, out=out)
import time
t0 = time.time()
while time.time() - t0 < 2.5:      # MI355X_MICROARCH.md: read the clock after >= 2 s of back-to-back launches on random data
    for _ in range(50):
        run()
torch.cuda.synchronize()
s = buf.cpu().numpy().astype("int64").reshape(-1)[: 8 * 8 * 4].reshape(8, 8, 4) & 0xffffffff
raw = buf.cpu().numpy().astype("int64") & 0xffffffff
for wg in range(8):
    c, r = raw[256 + 2 * wg], raw[257 + 2 * wg]
    if r:
        print("  wg %d: loop %d core cycles in %d ticks of 100 MHz -> %.2f GHz" % (wg, c, r, c / r * 0.1))
nt = 64 if mode == "global" else 16
print("cycles per iteration (s_memtime), summed over the loop / %d iterations; workgroups 0-7" % nt)
for wg in range(8):
    for w in (0, 1, 4, 7):
        r = s[wg, w] / nt
        print("  wg %d wave %d: head %5.0f  main %5.0f  tile wait %5.0f  barrier %5.0f | iteration %5.0f" % (wg, w, r[0], r[1], r[2], r[3], r.sum()))
