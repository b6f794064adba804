"""wg_dec_tokens_f32 alone: how its time splits between the token-side Linear chain and the token->image attention (hw keys)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
from walkgpt_amd.segment_anything import modeling as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
sam = M.build_sam_vit_b().to(dev).bfloat16()
md = sam.mask_decoder
layer = md.transformer.layers[1]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for P in (1, 8, 112):
    for hw in (64, 1024, 4096):
        q = torch.randn(P, 6, 256, device=dev)
        pe = torch.randn(P, 6, 256, device=dev)
        proj = torch.randn(P, hw, 384, device=dev).to(torch.bfloat16)
        kq = torch.empty(P, 6, 128, device=dev, dtype=torch.bfloat16); vq = torch.empty_like(kq)
        w0 = layer.token_weights(); w1 = md.head_weights()
        hy = torch.empty(P, 4, 32, device=dev); io = torch.empty(P, 4, device=dev)
        a = t(lambda: ops.dec_tokens(0, False, q, pe, w0, proj[..., :128], proj[..., 128:256], hw, k_i2t=kq, v_i2t=vq))
        b = t(lambda: ops.dec_tokens(1, False, q, pe, w1, proj[..., :128], proj[..., 128:256], hw, hyper_out=hy, iou_out=io))
        print("P=%3d hw=%4d: block kernel %.1f us, tail+heads kernel %.1f us" % (P, hw, a, b), flush=True)
