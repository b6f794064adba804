"""MLP (lin1 + GELU, lin2 + residual) over the rows in one piece or in row chunks: does the hidden layer reach lin2 from nearer memory when it is
produced and consumed chunk by chunk?  Round-robin over the variants in one process.  python tools/bench_mlp_chunks.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
def run(M, D, H, chunks_list, rounds=6, n=20):
    x = torch.randn(M, D, device=dev).to(torch.bfloat16)
    w1 = (torch.randn(H, D, device=dev) / D ** 0.5).to(torch.bfloat16); b1 = torch.randn(H, device=dev).to(torch.bfloat16)
    w2 = (torch.randn(D, H, device=dev) / H ** 0.5).to(torch.bfloat16); b2 = torch.randn(D, device=dev).to(torch.bfloat16)
    h = torch.empty(M, H, device=dev, dtype=torch.bfloat16); y = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    def mk(c):
        step = (M // c + 255) // 256 * 256
        rng = [(i, min(i + step, M)) for i in range(0, M, step)]
        def f():
            for (a, b) in rng:
                ops.linear(x[a:b], w1, b1, act=ops.ACT_GELU, out=h[a:b], tile=16)
                ops.linear(h[a:b], w2, b2, residual=x[a:b], out=y[a:b], tile=16)
        return f
    fns = {c: mk(c) for c in chunks_list}
    best = {c: 1e9 for c in chunks_list}
    for r in range(rounds):
        for c in chunks_list:
            f = fns[c]
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): f()
            e1.record(); torch.cuda.synchronize()
            best[c] = min(best[c], e0.elapsed_time(e1) / n * 1e3)
    print("M=%d D=%d H=%d us per MLP (best of %d):" % (M, D, H, rounds), "  ".join("%d chunks %.1f" % (c, best[c]) for c in chunks_list), flush=True)
run(32768, 768, 3072, [1, 2, 4, 8])
run(8200, 1024, 4096, [1, 2, 4])
run(32768, 1280, 5120, [1, 2, 4, 8])
