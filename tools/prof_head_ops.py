"""Which torch operators (and from which source lines) launch the elementwise kernels of a head training step: torch.profiler, grouped by stack."""
import sys
import torch
from torch.profiler import profile, ProfilerActivity
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
    bce, dice = ag.mask_losses(torch.cat(masks, 0).contiguous(), gt, T)
    ((2.0 * bce + 0.5 * dice) * T / (B * T + 1e-8)).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ka if e.key in ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::mul", "aten::cat", "aten::index_select", "aten::clone", "aten::contiguous", "aten::to")]
rows.sort(key=lambda e: -e.count)
for e in rows[:60]:
    print("%-16s x%-3d %s" % (e.key, e.count, str(e.input_shapes)[:150]))
