"""Forward + backward of the trainable grounding head alone on the GPU (SAM ViT-B decoder geometry, frozen embedding given): CTP -> prompt encoder ->
mask decoder -> postprocess -> sigmoid-CE + dice, `loss.backward()`.  Prints ms per step; under rocprofv3 the kernel table is the measurement
(tools/profile_r03.sh writes profiles/r03_train_head_*)."""
import argparse
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from walkgpt_amd import autograd as ag, train_head
from walkgpt_amd.walkgpt import WalkGPTGrounding


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seg-tokens", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--hidden", type=int, default=4096)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    g = WalkGPTGrounding(sam="vit_b", llm_hidden=a.hidden, with_clip=False).to(dev).bfloat16()
    B, T = a.batch, a.seg_tokens
    emb = torch.randn(B, 64 * 64, 256, device=dev).bfloat16()
    hidden = [torch.randn(T, a.hidden, device=dev).bfloat16().requires_grad_(True) for _ in range(B)]
    resize, orig = [(1024, 1024)] * B, [(448, 448)] * B
    gt = [(torch.rand(T, 448, 448, device=dev) > 0.5).float() for _ in range(B)]
    ctp = g.text_hidden_fcs[0]

    params = [p for p in g.parameters() if p.requires_grad]
    gt_all = torch.cat(gt, 0)

    def step():
        for p in params + hidden:          # optimizer.zero_grad(set_to_none=True), torch's default: gradients are stored, not added to old ones
            p.grad = None
        pred = train_head.ctp_forward(ctp, torch.cat(hidden, 0))
        masks = train_head.decode(g, emb, list(torch.split(pred, T, 0)), resize, orig)
        # equal mask counts and sizes: one reduction over all masks (what causal_lm.model_forward does for such a batch)
        bce, dice = ag.mask_losses(masks.stacked.contiguous(), gt_all, T)
        loss = (2.0 * bce + 0.5 * dice) * T / (B * T + 1e-8)
        loss.backward()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print("grounding head forward + backward: B=%d images x T=%d prompts: %.2f ms / step (loss %.4f)" % (B, T, ms, float(loss.detach())))


if __name__ == "__main__":
    main()
