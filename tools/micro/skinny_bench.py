# GEMMs with a few dozen rows (MSQP's query tokens x batch: M = 32..96 at 1024 / 4096 columns): 128x128 tiles vs the skinny kernel
import sys
import torch
sys.path.insert(0, '/root/repo')
from walkgpt_amd import ops
dev = torch.device('cuda:0')
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (16, 32, 64, 96, 128):
    for (N, K) in ((1024, 1024), (4096, 1024), (1024, 4096), (2048, 1024)):
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
        r = torch.randn(M, N, device=dev).bfloat16()
        a = t(lambda: ops.linear(x, w, b, act=1, tile=1))
        c = t(lambda: ops.linear(x, w, b, act=1, tile=5))
        d = t(lambda: ops.linear(x, w, b, residual=r, tile=5))
        ref = ops.linear(x, w, b, residual=r, tile=1).float(); got = ops.linear(x, w, b, residual=r, tile=5).float()
        print("M=%3d N=%4d K=%4d: 128x128 tiles %.1f us | skinny %.1f us (with residual %.1f) | max diff %.3g" % (M, N, K, a, c, d, (ref - got).abs().max().item()), flush=True)
