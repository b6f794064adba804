"""GPU parity of the fused attention kernels (through the C-ABI) against fp32 references built from the oracle's
own pieces (oracle.sam.rel_pos_bias) on the same bf16-rounded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import sam as osam
from walkgpt_amd import ops


def _rand(shape, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(torch.bfloat16)


def _ref_mha(q, k, v, heads, scale, key_bias=None):
    B, Lq, D = q.shape
    hd = D // heads
    sp = lambda t: t.float().reshape(B, -1, heads, hd).transpose(1, 2)
    a = (sp(q) @ sp(k).transpose(2, 3)) * scale
    if key_bias is not None:
        a = a + key_bias[:, None, None, :]
    return (a.softmax(-1) @ sp(v)).transpose(1, 2).reshape(B, Lq, D)


def _ref_sam_attention(qkv, bias, rel_h, rel_w, B, grid, window, heads):
    """image_encoder.py:177-193 + 235-260 on an already-projected qkv buffer: zero-padded positions carry q=k=v=bias."""
    D = qkv.shape[-1] // 3
    hd = D // heads
    x = qkv.float().reshape(B, grid, grid, 3 * D)
    pad = (-grid) % window
    Hp = grid + pad
    full = bias.float().reshape(1, 1, 1, 3 * D).repeat(B, Hp, Hp, 1)
    full[:, :grid, :grid] = x
    nw = Hp // window
    w = full.reshape(B, nw, window, nw, window, 3 * D).permute(0, 1, 3, 2, 4, 5).reshape(-1, window * window, 3, heads, hd)
    w = w.permute(2, 0, 3, 1, 4)
    q, k, v = [t.reshape(-1, window * window, hd) for t in w]
    logits = (q * hd ** -0.5) @ k.transpose(1, 2) + osam.rel_pos_bias(q, rel_h.float(), rel_w.float(), window)
    o = logits.softmax(-1) @ v
    o = o.reshape(B * nw * nw, heads, window, window, hd).permute(0, 2, 3, 1, 4).reshape(B, nw, nw, window, window, D)
    o = o.permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Hp, D)[:, :grid, :grid]
    return o.reshape(B * grid * grid, D)


@pytest.mark.parametrize("B,grid,window,heads,hd", [
    (2, 64, 14, 3, 64),   # SAM windows: 64 -> 70 padding, bias-valued pad keys
    (1, 64, 64, 2, 64),   # SAM global attention (row-tile rel-pos path)
    (2, 32, 14, 2, 64),   # tiny golden geometry: 32 -> 42
    (2, 32, 32, 2, 64),
    (1, 28, 14, 2, 32),   # hd 32, no padding
    (1, 28, 28, 2, 32),
    (1, 64, 14, 2, 80),   # SAM ViT-H head_dim (padded to 96 inside LDS)
    (1, 64, 64, 2, 80),
])
def test_sam_attention_relpos(dev, B, grid, window, heads, hd):
    D = heads * hd
    qkv = _rand((B * grid * grid, 3 * D), 1)
    bias = _rand((3 * D,), 2, 0.5)
    rel_h = _rand((2 * window - 1, hd), 3, 0.2)
    rel_w = _rand((2 * window - 1, hd), 4, 0.2)
    ref = _ref_sam_attention(qkv, bias, rel_h, rel_w, B, grid, window, heads)
    out = ops.sam_attention(qkv.to(dev), bias.to(dev), rel_h.to(dev), rel_w.to(dev), B, grid, window, heads)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 0.03, err


@pytest.mark.parametrize("B,Lq,Lk,heads,hd,masked", [
    (2, 1025, 1025, 4, 64, True),    # CLIP ViT-L geometry (fewer heads), key-padding mask
    (1, 1025, 1025, 2, 64, False),
    (2, 65, 65, 2, 64, True),        # tiny CLIP golden geometry
    (2, 300, 77, 2, 128, False),
    (1, 200, 1000, 2, 32, False),
])
def test_mha_plain_and_keymask(dev, B, Lq, Lk, heads, hd, masked):
    D = heads * hd
    q, k, v = _rand((B, Lq, D), 5), _rand((B, Lk, D), 6), _rand((B, Lk, D), 7)
    kb = None
    if masked:
        keep = (torch.rand(B, Lk, generator=torch.Generator().manual_seed(8)) > 0.3).float()
        keep[:, 0] = 1
        kb = (1 - keep) * torch.finfo(torch.float32).min
    ref = _ref_mha(q, k, v, heads, hd ** -0.5, kb)
    out = ops.mha(q.to(dev), k.to(dev), v.to(dev), heads, hd ** -0.5, None if kb is None else kb.to(dev), small=False)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 0.03, err


@pytest.mark.parametrize("B,Lq,Lk,heads,hd", [(5, 7, 4096, 8, 16), (3, 6, 1000, 8, 16), (2, 12, 4096, 8, 128), (2, 8, 1024, 8, 128),
                                             (2, 20, 300, 4, 32), (2, 9, 257, 2, 64)])
def test_mha_few_queries_many_keys_auto_routing(dev, B, Lq, Lk, heads, hd):
    """Decoder token->image and MSQP shapes through the default routing (MFMA flash kernel, one wave per (batch, head))."""
    D = heads * hd
    q, k, v = _rand((B, Lq, D), 20), _rand((B, Lk, D), 21), _rand((B, Lk, D), 22)
    scale = 1.0 / math.sqrt(hd)
    ref = _ref_mha(q, k, v, heads, scale)
    out = ops.mha(q.to(dev), k.to(dev), v.to(dev), heads, scale)
    assert (out.float().cpu() - ref).abs().max().item() < 0.03
    # shared (zero batch stride) keys, as the decoder's first layer uses them
    k1, v1 = k[:1].to(dev), v[:1].to(dev)
    out1 = ops.mha(q.to(dev), k1.expand(B, -1, -1), v1.expand(B, -1, -1), heads, scale)
    ref1 = _ref_mha(q, k[:1].expand(B, -1, -1), v[:1].expand(B, -1, -1), heads, scale)
    assert (out1.float().cpu() - ref1).abs().max().item() < 0.03


def test_mha_packed_qkv_views(dev):
    # q/k/v as column slices of one packed [B, L, 3D] buffer (CLIP layout)
    B, L, heads, hd = 2, 130, 2, 64
    D = heads * hd
    qkv = _rand((B, L, 3 * D), 9)
    ref = _ref_mha(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], heads, 0.125)
    d = qkv.to(dev)
    out = ops.mha(d[..., :D], d[..., D:2 * D], d[..., 2 * D:], heads, 0.125, small=False)
    assert (out.float().cpu() - ref).abs().max().item() < 0.03


@pytest.mark.parametrize("B,Lq,Lk,heads,hd", [
    (3, 7, 4096, 8, 16),    # mask decoder token -> image
    (3, 4096, 7, 8, 16),    # image -> token
    (3, 7, 7, 8, 32),       # token self-attention
    (2, 12, 4096, 8, 128),  # MSQP x1 scale
    (2, 4, 1, 8, 128),      # MSQP global token
    (2, 8, 256, 8, 128),
    (1, 5, 100, 2, 64),
])
def test_mha_small(dev, B, Lq, Lk, heads, hd):
    D = heads * hd
    q, k, v = _rand((B, Lq, D), 10), _rand((B, Lk, D), 11), _rand((B, Lk, D), 12)
    scale = 1.0 / math.sqrt(hd)
    ref = _ref_mha(q, k, v, heads, scale)
    out = ops.mha(q.to(dev), k.to(dev), v.to(dev), heads, scale, small=True)
    assert (out.float().cpu() - ref).abs().max().item() < 0.03


@pytest.mark.parametrize("mode", ["global", "plain"])
def test_attention_large_uneven_scores(dev, mode):
    """Scores of +-100s with the dominant keys confined to ONE lane half of each key tile (slots 4..7, 12..15, ... of every 32-key
    block belong to lanes >= 32): the running maximum must be the true maximum over both halves of the key tile, or the
    probabilities overflow.  (Regression: hipcc's permlane32_swap builtin returned the lower half's maximum for both halves.)"""
    g = torch.Generator().manual_seed(21)
    heads, hd = 2, 64
    D = heads * hd
    if mode == "global":
        B, grid = 1, 64
        qkv = torch.randn(B * grid * grid, 3 * D, generator=g)
        k = qkv[:, D:2 * D].reshape(-1, heads, hd)
        slot = torch.arange(grid * grid) % 64
        upper = ((slot % 8) >= 4)                         # key slots held by the upper lane half
        k[upper] *= 6.0                                   # those keys get scores up to several hundred
        qkv = qkv.to(torch.bfloat16)
        bias = _rand((3 * D,), 2, 0.5)
        rel_h, rel_w = _rand((2 * grid - 1, hd), 3, 0.2), _rand((2 * grid - 1, hd), 4, 0.2)
        ref = _ref_sam_attention(qkv, bias, rel_h, rel_w, B, grid, grid, heads)
        out = ops.sam_attention(qkv.to(dev), bias.to(dev), rel_h.to(dev), rel_w.to(dev), B, grid, grid, heads)
    else:
        B, L = 2, 1025
        q, k, v = torch.randn(B, L, D, generator=g), torch.randn(B, L, D, generator=g), torch.randn(B, L, D, generator=g)
        upper = ((torch.arange(L) % 8) >= 4)
        k[:, upper] *= 6.0
        q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
        ref = _ref_mha(q, k, v, heads, hd ** -0.5)
        out = ops.mha(q.to(dev), k.to(dev), v.to(dev), heads, hd ** -0.5, small=False)
    o = out.float().cpu()
    assert torch.isfinite(o).all()
    assert (o - ref).abs().max().item() < 0.06


@pytest.fixture
def pipe_plain():
    """Route plain attention through the pipelined kernel too (by default only SAM's global attention takes it: wg_attn_pipe_mode)."""
    from walkgpt_amd import _lib
    prev = _lib.lib().wg_attn_pipe_mode(2)
    yield
    _lib.lib().wg_attn_pipe_mode(prev)


@pytest.mark.parametrize("B,Lq,Lk,heads", [(1, 4096, 4096, 2), (2, 1024, 1024, 2), (1, 300, 1025, 3), (1, 512, 256, 2)])
def test_mha_pipelined_loop(dev, B, Lq, Lk, heads, pipe_plain):
    """head_dim 64 without a key bias on whole 64-key tiles: the software-pipelined kernel (attn_pipe.hip; eight-wave and four-wave
    workgroups, with and without the lone 1025th key, the shortest loop it takes)."""
    hd = 64
    D = heads * hd
    q, k, v = _rand((B, Lq, D), 31), _rand((B, Lk, D), 32), _rand((B, Lk, D), 33)
    ref = _ref_mha(q, k, v, heads, hd ** -0.5)
    out = ops.mha(q.to(dev), k.to(dev), v.to(dev), heads, hd ** -0.5, small=False)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 0.03, err


@pytest.mark.parametrize("mode,ramp_f,tol", [("global", (2.0, 3.0, 4.5, 6.0), 0.03), ("plain", (2.0, 3.0, 4.5, 6.0), 0.03),
                                           ("global", (3.0, 5.0, 8.0, 12.0), 0.03), ("plain", (3.0, 5.0, 8.0, 12.0), 0.03)])
def test_pipelined_attention_rescales_a_pending_tile(dev, mode, ramp_f, tol, pipe_plain):
    """The pipelined loop decides tile t's rescale while the P.V product of tile t-1 is still pending: that tile's probabilities must be
    rescaled with the accumulator (cdna_hip_programming.md T13: a rare, data-dependent branch needs an input that FORCES it).  Keys are
    scaled so that the row maximum jumps by far more than the lazy-rescale threshold at chosen tiles late in the loop -- once, twice in
    consecutive tiles, and in the very last tile -- against an fp32 reference over the full tensor.  Two strengths: maxima that jump by
    5-20 nats, and scores of several hundred, where the softmax is one-hot up to ties.  The kernel keeps the fp32 scale-and-offset of the
    scores on the vector ALU (attn_pipe.hip; the variant that pre-multiplied the queries and rounded them to bf16 a second time was removed
    in round 4): measured 0.009 (global) and 0.015-0.016 (plain) at both strengths on the shipped kernel, bound 0.03 -- the tolerance of the
    other attention tests."""
    g = torch.Generator().manual_seed(5)
    heads, hd = 2, 64
    D = heads * hd
    L = 4096 if mode == "global" else 2048
    ramp = torch.ones(L)
    for t, f in zip((7, 8, 20, L // 64 - 1), ramp_f):
        ramp[t * 64:(t + 1) * 64] = f
    if mode == "global":
        qkv = torch.randn(L, 3 * D, generator=g)
        qkv[:, D:2 * D] *= ramp[:, None]
        qkv = qkv.to(torch.bfloat16)
        bias = _rand((3 * D,), 2, 0.5)
        rel_h, rel_w = _rand((127, hd), 3, 0.2), _rand((127, hd), 4, 0.2)
        ref = _ref_sam_attention(qkv, bias, rel_h, rel_w, 1, 64, 64, heads)
        out = ops.sam_attention(qkv.to(dev), bias.to(dev), rel_h.to(dev), rel_w.to(dev), 1, 64, 64, heads)
    else:
        q, k, v = torch.randn(1, L, D, generator=g), torch.randn(1, L, D, generator=g), torch.randn(1, L, D, generator=g)
        k *= ramp[None, :, None]
        q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
        ref = _ref_mha(q, k, v, heads, hd ** -0.5)
        out = ops.mha(q.to(dev), k.to(dev), v.to(dev), heads, hd ** -0.5, small=False)
    o = out.float().cpu()
    assert torch.isfinite(o).all()
    err = (o - ref).abs().max().item()
    print("pending-tile rescale (%s, key scales %s): max abs err %.4f" % (mode, ramp_f, err))
    assert err < tol
