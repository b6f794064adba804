"""Backward passes of the trainable grounding head (walkgpt_amd.autograd; train_walkgpt.py:347-350): every differentiable HIP operator against
torch autograd on the same bf16-rounded inputs in fp32, then the composed training paths (CTP, mask decoder, losses) against the oracle's
restatement differentiated by torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from walkgpt_amd import autograd as ag
from walkgpt_amd import ops


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _leaf(t, dev):
    return t.to(dev).detach().requires_grad_(True)


@pytest.mark.parametrize("M,K,N", [(300, 256, 512), (7, 256, 4), (4099, 768, 256), (64, 4096, 512)])
def test_linear_backward(dev, M, K, N):
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    dy = torch.randn(M, N, generator=g).to(torch.bfloat16)
    xh, wh, bh = _leaf(x, dev), _leaf(w, dev), _leaf(b, dev)
    y = ag.linear(xh, wh, bh)
    y.backward(dy.to(dev))
    xr, wr, br = (t.float().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(dy.float())
    assert rel(y, yr) < 4e-3
    assert rel(xh.grad, xr.grad) < 6e-3 and rel(wh.grad, wr.grad) < 6e-3 and rel(bh.grad, br.grad) < 6e-3


@pytest.mark.parametrize("act", [1, 2, 3])
def test_activation_backward(dev, act):
    g = torch.Generator().manual_seed(act)
    x = (torch.randn(1000, 264, generator=g) * 2).to(torch.bfloat16)
    dy = torch.randn(1000, 264, generator=g).to(torch.bfloat16)
    xh = _leaf(x, dev)
    y = ag.activation(xh, act)
    y.backward(dy.to(dev))
    xr = x.float().requires_grad_(True)
    fn = {1: torch.nn.functional.gelu, 2: lambda t: t * torch.sigmoid(1.702 * t), 3: torch.relu}[act]
    yr = fn(xr)
    yr.backward(dy.float())
    assert rel(y, yr) < 4e-3 and rel(xh.grad, xr.grad) < 5e-3


@pytest.mark.parametrize("M,C,eps", [(300, 256, 1e-5), (4099, 1280, 1e-6), (33, 4096, 1e-5)])
def test_layernorm_backward(dev, M, C, eps):
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(torch.bfloat16)
    gam = (1 + 0.3 * torch.randn(C, generator=g)).to(torch.bfloat16)
    bet = (0.3 * torch.randn(C, generator=g)).to(torch.bfloat16)
    dy = torch.randn(M, C, generator=g).to(torch.bfloat16)
    xh, gh, bh = _leaf(x, dev), _leaf(gam, dev), _leaf(bet, dev)
    y = ag.layernorm(xh, gh, bh, eps)
    y.backward(dy.to(dev))
    xr, gr, br = (t.float().requires_grad_(True) for t in (x, gam, bet))
    yr = torch.nn.functional.layer_norm(xr, (C,), gr, br, eps)
    yr.backward(dy.float())
    assert rel(y, yr) < 4e-3
    assert rel(xh.grad, xr.grad) < 6e-3 and rel(gh.grad, gr.grad) < 6e-3 and rel(bh.grad, br.grad) < 6e-3


@pytest.mark.parametrize("B,H,hd,Lq,Lk", [(3, 8, 16, 6, 4096), (3, 8, 32, 6, 6), (2, 8, 16, 4096, 6), (2, 8, 128, 12, 1024), (4, 1, 64, 1, 300),
                                          (2, 4, 32, 700, 16)])
def test_attention_backward(dev, B, H, hd, Lq, Lk):
    g = torch.Generator().manual_seed(Lq + Lk)
    D = H * hd
    q, k, v = (torch.randn(B, L, D, generator=g).to(torch.bfloat16) for L in (Lq, Lk, Lk))
    do = torch.randn(B, Lq, D, generator=g).to(torch.bfloat16)
    scale = hd ** -0.5
    qh, kh, vh = _leaf(q, dev), _leaf(k, dev), _leaf(v, dev)
    o = ag.attention(qh, kh, vh, H, scale)
    o.backward(do.to(dev))
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    sp = lambda t, L: t.view(B, L, H, hd).transpose(1, 2)
    att = torch.softmax(sp(qr, Lq) @ sp(kr, Lk).transpose(-1, -2) * scale, -1)
    orf = (att @ sp(vr, Lk)).transpose(1, 2).reshape(B, Lq, D)
    orf.backward(do.float())
    assert rel(o, orf) < 6e-3
    assert rel(qh.grad, qr.grad) < 1.5e-2 and rel(kh.grad, kr.grad) < 1.5e-2 and rel(vh.grad, vr.grad) < 1.5e-2, \
        (rel(qh.grad, qr.grad), rel(kh.grad, kr.grad), rel(vh.grad, vr.grad))


def test_ctp_tail_operators_backward(dev):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(37, 256, generator=g).to(torch.bfloat16)
    tt = (0.3 * torch.randn(256, generator=g)).to(torch.bfloat16)
    lt = torch.tensor([0.4]).to(torch.bfloat16)
    dy = torch.randn(37, 256, generator=g).to(torch.bfloat16)
    xh, th, lh = _leaf(x, dev), _leaf(tt, dev), _leaf(lt, dev)
    y = ag.l2norm_scale(ag.add_row(xh, th), lh)
    y.backward(dy.to(dev))
    xr, tr, lr = (t.float().requires_grad_(True) for t in (x, tt, lt))
    yr = torch.nn.functional.normalize(xr + tr, dim=-1, eps=1e-12) * lr.exp()
    yr.backward(dy.float())
    assert rel(y, yr) < 6e-3 and rel(xh.grad, xr.grad) < 1.2e-2 and rel(th.grad, tr.grad) < 1.5e-2 and rel(lh.grad, lr.grad) < 1.5e-2


@pytest.mark.parametrize("N,inp,orig", [(3, (1024, 1024), (448, 448)), (2, (683, 1024), (300, 450))])
def test_postprocess_and_mask_losses_backward(dev, N, inp, orig):
    """loss(postprocess(low_res)) differentiated through both HIP adjoints against torch autograd over F.interpolate and the reference's loss formulas."""
    g = torch.Generator().manual_seed(N)
    low = torch.randn(N, 1, 256, 256, generator=g) * 3
    tgt = (torch.rand(N, orig[0], orig[1], generator=g) > 0.6).float()
    lh = _leaf(low, dev)
    masks = ag.postprocess_masks(lh, 1024, inp, orig)
    bce, dice = ag.mask_losses(masks[:, 0].contiguous(), tgt.to(dev), N)
    (2.0 * bce + 0.5 * dice).backward()
    lr = low.clone().requires_grad_(True)
    F = torch.nn.functional
    m = F.interpolate(lr, (1024, 1024), mode="bilinear", align_corners=False)[..., :inp[0], :inp[1]]
    m = F.interpolate(m, orig, mode="bilinear", align_corners=False)[:, 0]
    bce_r = F.binary_cross_entropy_with_logits(m, tgt, reduction="none").flatten(1, 2).mean(1).sum() / (N + 1e-8)
    s = m.sigmoid().flatten(1, 2)
    t = tgt.flatten(1, 2)
    num = 2 * (s / 1000 * t).sum(-1)
    den = (s / 1000).sum(-1) + (t / 1000).sum(-1)
    dice_r = (1 - (num + 1e-6) / (den + 1e-6)).sum() / (N + 1e-8)
    (2.0 * bce_r + 0.5 * dice_r).backward()
    assert abs(float(bce) - float(bce_r)) < 1e-5 and abs(float(dice) - float(dice_r)) < 1e-5
    assert rel(lh.grad, lr.grad) < 1e-4, rel(lh.grad, lr.grad)
