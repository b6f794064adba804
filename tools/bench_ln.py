"""LayerNorm kernel alone on the C2 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, D) in [(32768, 768), (8200, 1024), (32768, 1280), (32768, 256)]:
    x = torch.randn(M, D, device=dev).to(torch.bfloat16)
    g = torch.randn(D, device=dev).to(torch.bfloat16); b = torch.randn(D, device=dev).to(torch.bfloat16)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), g.float(), b.float(), 1e-6)
    y = ops.layernorm(x, g, b, 1e-6)
    err = (y.float() - ref).abs().max().item()
    ms = t(lambda: ops.layernorm(x, g, b, 1e-6))
    print("M=%d D=%d  %.1f us  %.2f TB/s  max err %.3g" % (M, D, ms * 1e3, 4.0 * M * D / ms / 1e9, err), flush=True)
