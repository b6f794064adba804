"""Token-side kernels of the mask decoder on their own: per-launch time of each stage at P prompts and hw image tokens."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
from walkgpt_amd.segment_anything import modeling as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
sam = M.build_sam_vit_b().to(dev).bfloat16()
md = sam.mask_decoder
l0, l1 = md.transformer.layers
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
        qt = torch.empty(P, 6, 128, device=dev)
        tab = M.token_stage_table(self_blk=l1, t2i=l1.cross_attn_token_to_image, norm=l1.norm2, sum_blk=l0)
        part = ops.dec_attn_partial(qt, proj[..., :256])
        mp = l1.mlp_partials(q)
        r = {
            "sum|self|q": t(lambda: ops.dec_tokens(7, False, q, pe, tab, q_t2i=qt, mlp_partials=mp, k_i2t=kq, v_i2t=vq)),
            "attn": t(lambda: ops.dec_attn_partial(qt, proj[..., :256])),
            "combine": t(lambda: ops.dec_tokens(8, False, q, pe, tab, attn_partials=part)),
            "mlp": t(lambda: l1.mlp_partials(q)),
            "combine+mlp": t(lambda: l1.mlp_partials(q, (part, M._tiled(l1.cross_attn_token_to_image.out_proj.weight), l1.cross_attn_token_to_image.out_proj.bias, l1.norm2.weight, l1.norm2.bias))),
            "heads": t(lambda: ops.dec_heads(q, md.head_weights())),
        }
        print("P=%3d hw=%4d: " % (P, hw) + "  ".join("%s %.1f us" % kv for kv in r.items()), flush=True)
