"""Does the row pitch of an operand matter?  The persistent GEMM on shapes of the workload with the rows of A / W / C at their natural pitch and
at padded pitches, configurations timed round-robin (four rounds; the first is warm-up):  WG_LIB=... python tools/bench_gemm_pitch.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [("sam lin2", 32768, 768, 3072), ("vit-h lin2", 32768, 1280, 5120), ("clip fc2", 8200, 1024, 4096), ("sam proj", 32768, 768, 768), ("sam qkv", 32768, 2304, 768)]
cfgs = ((0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (0, 0, 64), (8, 0, 0), (128, 0, 0))
for (name, M, N, K) in shapes:
    runs = []
    for pad, wpad, opad in cfgs:
        a = torch.randn(M, K + pad, device=dev).to(torch.bfloat16)[:, :K]
        w = (torch.randn(N, K + wpad, device=dev) / K ** 0.5).to(torch.bfloat16)[:, :K]
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N + opad, device=dev, dtype=torch.bfloat16)[:, :N]
        runs.append((a, w, b, out))
    times = [[] for _ in cfgs]
    for rnd in range(4):
        for i, (a, w, b, out) in enumerate(runs):
            ms = t(lambda: ops.linear(a, w, b, out=out, tile=16))
            if rnd > 0: times[i].append(ms * 1e3)
    print("%-10s M=%d N=%d K=%d | %s" % (name, M, N, K, " | ".join("A+%d W+%d C+%d: %s" % (c + ("/".join("%.1f" % x for x in ts),)) for c, ts in zip(cfgs, times))), flush=True)
