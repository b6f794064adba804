"""Two identical forward + backward passes of the trainable grounding head: which parameter gradients differ in their bits (atomics left on the path)?"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from walkgpt_amd import autograd as ag, train_head
from walkgpt_amd.walkgpt import WalkGPTGrounding

dev = torch.device("cuda:0")
torch.manual_seed(0)
g = WalkGPTGrounding(sam="vit_b", llm_hidden=4096, with_clip=False).to(dev).bfloat16()
B, T = 8, 2
emb = torch.randn(B, 64 * 64, 256, device=dev).bfloat16()
hidden = [torch.randn(T, 4096, device=dev).bfloat16().requires_grad_(True) for _ in range(B)]
resize, orig = [(1024, 1024)] * B, [(448, 448)] * B
gt = torch.cat([(torch.rand(T, 448, 448, device=dev) > 0.5).float() for _ in range(B)], 0)
ctp = g.text_hidden_fcs[0]
named = [(n, p) for n, p in g.named_parameters() if p.requires_grad] + [("hidden%d" % i, h) for i, h in enumerate(hidden)]


def step():
    for _, p in named:
        p.grad = None
    pred = train_head.ctp_forward(ctp, torch.cat(hidden, 0))
    masks = train_head.decode(g, emb, list(torch.split(pred, T, 0)), resize, orig)
    bce, dice = ag.mask_losses(torch.cat(masks, 0).contiguous(), gt, T)
    (2.0 * bce + 0.5 * dice).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in named if p.grad is not None}


a, b = step(), step()
bad = [n for n in a if not torch.equal(a[n], b[n])]
print("%d tensors with gradients, %d differ between two passes" % (len(a), len(bad)))
for n in bad[:40]:
    d = (a[n].float() - b[n].float()).abs().max().item()
    print("  %-70s max |diff| %.3g (|g| max %.3g)" % (n, d, a[n].float().abs().max().item()))
