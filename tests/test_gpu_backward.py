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
