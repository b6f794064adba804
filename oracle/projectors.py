"""ORACLE (test infrastructure, never the product path): CPU fp32 restatement of WalkGPT's two projectors and the
LLM-side token resample.

  msqp()            Multi-Scale Query Projector   /root/reference/utils/utils_walkgpt.py:220-300
                    (+ CrossAttnBlock :163-185, _pool_grid_tokens :195-201, SegAwareGate :204-217)
  ctp()             Calibrated Text Projector     /root/reference/utils/utils_walkgpt.py:302-327
  resample_tokens() 6x6 -> 16x16 bilinear         /root/reference/model/llava_walkgpt/model/llava_arch.py:252-259

Pinned by tests/golden/*.npz (outputs of the reference modules themselves on walkgpt_amd/synth.py weights).
`w` is a flat {reference state_dict key: fp32 tensor} dict relative to the module root.
"""
import math

import torch
import torch.nn.functional as F


def _ln(x, w, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + ".weight"], w[prefix + ".bias"], eps)


def _lin(x, w, prefix):
    return F.linear(x, w[prefix + ".weight"], w.get(prefix + ".bias"))


def _mha(w, prefix, q, kv, heads):
    """nn.MultiheadAttention(batch_first=True, dropout 0), packed in_proj, as CrossAttnBlock uses it (:175-180)."""
    D = q.shape[-1]
    Wi, bi = w[prefix + ".in_proj_weight"], w[prefix + ".in_proj_bias"]
    qp = F.linear(q, Wi[:D], bi[:D])
    kp = F.linear(kv, Wi[D:2 * D], bi[D:2 * D])
    vp = F.linear(kv, Wi[2 * D:], bi[2 * D:])
    B, Nq, _ = qp.shape
    hd = D // heads

    def split(t):
        return t.reshape(B, t.shape[1], heads, hd).transpose(1, 2)

    a = (split(qp) @ split(kp).transpose(2, 3)) / math.sqrt(hd)
    o = (a.softmax(-1) @ split(vp)).transpose(1, 2).reshape(B, Nq, D)
    return F.linear(o, w[prefix + ".out_proj.weight"], w[prefix + ".out_proj.bias"])


def _cross_block(w, prefix, queries, kv, heads):
    """CrossAttnBlock.forward :180-185: pre-norm cross attention + pre-norm GELU FFN, both residual."""
    out = queries + _mha(w, prefix + ".attn", _ln(queries, w, prefix + ".q_norm"), _ln(kv, w, prefix + ".kv_norm"), heads)
    h = F.gelu(_lin(_ln(out, w, prefix + ".ffn.0"), w, prefix + ".ffn.1"))
    return out + _lin(h, w, prefix + ".ffn.3")


def _gate(w, prefix, x):
    """SegAwareGate :213-217: x * sigmoid(W2 gelu(W1 LN(x)))."""
    logit = _lin(F.gelu(_lin(_ln(x, w, prefix + ".net.0"), w, prefix + ".net.1")), w, prefix + ".net.3")
    return x * torch.sigmoid(logit)


def msqp(w, sam_tokens, heads=8, side=6):
    """sam_tokens [B, L, sam_dim] (L a perfect square) -> [B, side*side, llama_dim]."""
    B, L, _ = sam_tokens.shape
    H = int(math.isqrt(L))
    assert H * H == L
    f = _lin(sam_tokens, w, "sam_to_proj")
    C = f.shape[-1]
    grid = f.reshape(B, H, H, C).permute(0, 3, 1, 2)

    def pool(s):  # _pool_grid_tokens :195-201
        return F.avg_pool2d(grid, s, s).permute(0, 2, 3, 1).reshape(B, -1, C)

    scales = [("q_x1", "cross_x1", f), ("q_x2", "cross_x2", pool(2)), ("q_x4", "cross_x4", pool(4)),
              ("q_global", "cross_glb", f.mean(1, keepdim=True))]
    outs = []
    for qname, cname, kv in scales:
        kv = _gate(w, "gate", kv)  # one shared gate for all four scales (:276)
        q = w[qname].expand(B, -1, -1)
        for layer in range(2):
            q = _cross_block(w, "%s.%d" % (cname, layer), q, kv, heads)
        outs.append(q)
    vis = torch.cat(outs, 1)  # [x1(12), x2(8), x4(8), glb(4)]  :290
    pad = side * side - vis.shape[1]
    if pad > 0:
        vis = torch.cat([vis, w["pad_token"].expand(B, pad, -1)], 1)
    return _lin(vis, w, "to_llama")


def ctp(w, x):
    """CalibratedTextProjector.forward :321-327 (use_residual False)."""
    y = _ln(x, w, "net.0")
    y = _lin(F.gelu(_lin(y, w, "net.1")), w, "net.3")
    y = _ln(y, w, "net.4")
    return F.normalize(y + w["text_type"], dim=-1) * w["log_temp"].exp()


def resample_tokens(tokens, target=16):
    """llava_arch.py:252-259: [n, p*p, C] -> bilinear (align_corners False) -> [n, target*target, C]."""
    n, l, c = tokens.shape
    p = int(math.isqrt(l))
    assert p * p == l
    g = tokens.permute(0, 2, 1).reshape(n, c, p, p).float()
    g = F.interpolate(g, size=(target, target), mode="bilinear", align_corners=False)
    return g.flatten(2).permute(0, 2, 1).to(tokens.dtype)
