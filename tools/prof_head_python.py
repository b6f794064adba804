"""cProfile of the host side of a head training step (which Python frames the ~7 ms go to)."""
import cProfile, pstats, sys, io
import torch
sys.path.insert(0, "/root/repo")
from walkgpt_amd import autograd as ag, train_head
from walkgpt_amd.walkgpt import WalkGPTGrounding

dev = torch.device("cuda:0")
torch.manual_seed(0)
g = WalkGPTGrounding(sam="vit_b", llm_hidden=4096, with_clip=False).to(dev).bfloat16()
B, T = 8, 1
emb = torch.randn(B, 64 * 64, 256, device=dev).bfloat16()
hidden = [torch.randn(T, 4096, device=dev).bfloat16().requires_grad_(True) for _ in range(B)]
resize, orig = [(1024, 1024)] * B, [(448, 448)] * B
gt = torch.cat([(torch.rand(T, 448, 448, device=dev) > 0.5).float() for _ in range(B)], 0)
params = [p for p in g.parameters() if p.requires_grad]


def step():
    for p in params + hidden:
        p.grad = None
    pred = train_head.ctp_forward(g.text_hidden_fcs[0], torch.cat(hidden, 0))
    masks = train_head.decode(g, emb, list(torch.split(pred, T, 0)), resize, orig)
    bce, dice = ag.mask_losses(masks.stacked.contiguous(), gt, T)
    ((2.0 * bce + 0.5 * dice) * T / (B * T + 1e-8)).backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
