"""ORACLE (test infrastructure, never the product path): CPU fp32 restatement of the SAM half of WalkGPT's
grounded-segmentation forward path -- image encoder, prompt encoder (text branch), two-way mask decoder and mask
post-processing.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Pinned by tests/golden/*.npz, which tests/golden/make_golden.py produced by running the reference's own modules
(imported from /root/reference in the build container) on the same synthetic weights (walkgpt_amd/synth.py).

Functional style: every function takes `w`, a flat {name: fp32 tensor} dict whose names are the reference
state_dict keys relative to the module root, and plain tensors.  File:line citations are relative to
/root/reference/model/segment_anything/modeling/.
"""
import math

import torch
import torch.nn.functional as F


def _ln(x, w, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + ".weight"], w[prefix + ".bias"], eps)


def _lin(x, w, prefix):
    return F.linear(x, w[prefix + ".weight"], w.get(prefix + ".bias"))


def _ln2d(x, w, prefix, eps=1e-6):
    """common.py:31-43 -- normalise over the channel axis of NCHW with biased variance, eps inside the sqrt."""
    mu = x.mean(1, keepdim=True)
    var = ((x - mu) ** 2).mean(1, keepdim=True)
    xn = (x - mu) / torch.sqrt(var + eps)
    return xn * w[prefix + ".weight"][None, :, None, None] + w[prefix + ".bias"][None, :, None, None]


# --------------------------------------------------------------------------------------------------------
# image encoder (image_encoder.py)
# --------------------------------------------------------------------------------------------------------
def rel_pos_bias(q, rel_h, rel_w, S):
    """image_encoder.py:321-392 for equal query/key grids of side S (no table interpolation, :335).
    q: [G, S*S, hd] UNSCALED queries (:247-249).  Returns [G, S*S, S*S]."""
    idx = torch.arange(S)[:, None] - torch.arange(S)[None, :] + (S - 1)  # :347-351  q - k + (S-1)
    Rh, Rw = rel_h[idx], rel_w[idx]  # [S(q), S(k), hd]
    G = q.shape[0]
    qg = q.reshape(G, S, S, -1)
    bh = torch.einsum("gyxc,ykc->gyxk", qg, Rh)  # [G, qy, qx, ky]
    bw = torch.einsum("gyxc,xkc->gyxk", qg, Rw)  # [G, qy, qx, kx]
    return (bh[:, :, :, :, None] + bw[:, :, :, None, :]).reshape(G, S * S, S * S)


def vit_attention(w, prefix, x, heads):
    """image_encoder.py:235-260.  x: [G, S, S, D] (a window batch or whole images)."""
    G, S, S2, D = x.shape
    assert S == S2
    hd = D // heads
    qkv = _lin(x, w, prefix + ".qkv").reshape(G, S * S, 3, heads, hd).permute(2, 0, 3, 1, 4)  # :238-242
    q, k, v = [t.reshape(G * heads, S * S, hd) for t in qkv]
    logits = (q * hd ** -0.5) @ k.transpose(1, 2)  # :244
    if (prefix + ".rel_pos_h") in w:
        logits = logits + rel_pos_bias(q, w[prefix + ".rel_pos_h"], w[prefix + ".rel_pos_w"], S)
    p = logits.softmax(-1)
    o = (p @ v).reshape(G, heads, S, S, hd).permute(0, 2, 3, 1, 4).reshape(G, S, S, D)
    return _lin(o, w, prefix + ".proj")


def vit_block(w, prefix, x, heads, window):
    """image_encoder.py:177-193 with window_partition/unpartition :263-318 folded in."""
    B, H, W_, D = x.shape
    y = _ln(x, w, prefix + ".norm1", 1e-6)
    if window > 0:
        ph, pw = (-H) % window, (-W_) % window
        y = F.pad(y, (0, 0, 0, pw, 0, ph))  # zeros AFTER norm1: pad tokens' q/k/v equal the qkv bias
        Hp, Wp = H + ph, W_ + pw
        y = y.reshape(B, Hp // window, window, Wp // window, window, D).permute(0, 1, 3, 2, 4, 5)
        y = y.reshape(-1, window, window, D)
        y = vit_attention(w, prefix + ".attn", y, heads)
        y = y.reshape(B, Hp // window, Wp // window, window, window, D).permute(0, 1, 3, 2, 4, 5)
        y = y.reshape(B, Hp, Wp, D)[:, :H, :W_]
    else:
        y = vit_attention(w, prefix + ".attn", y, heads)
    x = x + y
    h = F.gelu(_lin(_ln(x, w, prefix + ".norm2", 1e-6), w, prefix + ".mlp.lin1"))  # common.py:13-26, erf GELU
    return x + _lin(h, w, prefix + ".mlp.lin2")


def image_encoder(w, images, cfg, prefix="image_encoder"):
    """image_encoder.py:110-125.  images [B,3,S,S] -> [B,out,S/p,S/p].  cfg: embed_dim, depth, heads,
    global_idx, window, patch."""
    p = cfg["patch"]
    x = F.conv2d(images, w[prefix + ".patch_embed.proj.weight"], w[prefix + ".patch_embed.proj.bias"], stride=p)
    x = x.permute(0, 2, 3, 1)  # :422-426
    x = x + w[prefix + ".pos_embed"]  # :111-113
    for i in range(cfg["depth"]):
        win = 0 if i in cfg["global_idx"] else cfg["window"]
        x = vit_block(w, "%s.blocks.%d" % (prefix, i), x, cfg["heads"], win)
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, w[prefix + ".neck.0.weight"])  # :92-108
    x = _ln2d(x, w, prefix + ".neck.1")
    x = F.conv2d(x, w[prefix + ".neck.2.weight"], padding=1)
    return _ln2d(x, w, prefix + ".neck.3")


# --------------------------------------------------------------------------------------------------------
# prompt encoder, text branch only (prompt_encoder.py:140-186, 203-229)
# --------------------------------------------------------------------------------------------------------
def dense_pe(w, size, prefix="prompt_encoder"):
    """prompt_encoder.py:67-76 + 216-229: [1, 2*F, h, w] = [sin | cos](2 pi ((2c-1) @ G)), c = (i+0.5)/size, [x,y]."""
    Gm = w[prefix + ".pe_layer.positional_encoding_gaussian_matrix"]
    h, wd = size
    ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
    xs = (torch.arange(wd, dtype=torch.float32) + 0.5) / wd
    c = torch.stack([xs[None, :].expand(h, wd), ys[:, None].expand(h, wd)], -1)
    t = 2 * math.pi * ((2 * c - 1) @ Gm)
    return torch.cat([t.sin(), t.cos()], -1).permute(2, 0, 1)[None]


def prompt_encoder_text(w, text_embeds, size, prefix="prompt_encoder"):
    """text_embeds [T,1,C] -> sparse [T,1,C], dense [T,C,h,w] (no_mask_embed broadcast, :180-184)."""
    T = text_embeds.shape[0]
    dense = w[prefix + ".no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(T, -1, size[0], size[1])
    return text_embeds, dense


# --------------------------------------------------------------------------------------------------------
# two-way transformer + mask decoder (transformer.py, mask_decoder.py)
# --------------------------------------------------------------------------------------------------------
def _dec_attn(w, prefix, q, k, v, heads):
    """transformer.py:220-242: project, split heads, softmax(q k^T / sqrt(c_per_head)) v, merge, out_proj."""
    q, k, v = _lin(q, w, prefix + ".q_proj"), _lin(k, w, prefix + ".k_proj"), _lin(v, w, prefix + ".v_proj")
    B, Nq, C = q.shape
    c = C // heads

    def split(t):
        return t.reshape(B, t.shape[1], heads, c).transpose(1, 2)

    a = (split(q) @ split(k).transpose(2, 3)) / math.sqrt(c)
    o = (a.softmax(-1) @ split(v)).transpose(1, 2).reshape(B, Nq, C)
    return _lin(o, w, prefix + ".out_proj")


def two_way_transformer(w, prefix, src, pos, tokens, depth=2, heads=8):
    """transformer.py:62-106 / 151-182.  src,pos [B,C,h,w]; tokens [B,N,C] -> (queries [B,N,C], keys [B,hw,C])."""
    keys = src.flatten(2).permute(0, 2, 1)
    kpe = pos.flatten(2).permute(0, 2, 1)
    queries, qpe = tokens, tokens
    for i in range(depth):
        L = "%s.layers.%d" % (prefix, i)
        if i == 0:  # skip_first_layer_pe: self-attention output REPLACES the queries (:155-156)
            queries = _dec_attn(w, L + ".self_attn", queries, queries, queries, heads)
        else:
            qq = queries + qpe
            queries = queries + _dec_attn(w, L + ".self_attn", qq, qq, queries, heads)
        queries = _ln(queries, w, L + ".norm1", 1e-5)
        queries = queries + _dec_attn(w, L + ".cross_attn_token_to_image", queries + qpe, keys + kpe, keys, heads)
        queries = _ln(queries, w, L + ".norm2", 1e-5)
        queries = queries + _lin(F.relu(_lin(queries, w, L + ".mlp.lin1")), w, L + ".mlp.lin2")
        queries = _ln(queries, w, L + ".norm3", 1e-5)
        keys = keys + _dec_attn(w, L + ".cross_attn_image_to_token", keys + kpe, queries + qpe, queries, heads)
        keys = _ln(keys, w, L + ".norm4", 1e-5)
    queries = queries + _dec_attn(w, prefix + ".final_attn_token_to_image", queries + qpe, keys + kpe, keys, heads)
    return _ln(queries, w, prefix + ".norm_final_attn", 1e-5), keys


def _mlp3(w, prefix, x):
    """mask_decoder.py:169-191 with num_layers=3, ReLU between, none at the end."""
    x = F.relu(_lin(x, w, prefix + ".layers.0"))
    x = F.relu(_lin(x, w, prefix + ".layers.1"))
    return _lin(x, w, prefix + ".layers.2")


def mask_decoder(w, image_embedding, image_pe, sparse, dense, multimask_output=False, prefix="mask_decoder"):
    """mask_decoder.py:75-164.  image_embedding [1,C,h,w]; sparse [T,n,C]; dense [T,C,h,w] ->
    masks [T,1|3,4h,4w], iou [T,1|3]."""
    T = sparse.shape[0]
    out_tok = torch.cat([w[prefix + ".iou_token.weight"], w[prefix + ".mask_tokens.weight"]], 0)
    tokens = torch.cat([out_tok[None].expand(T, -1, -1), sparse], 1)  # order [iou, mask0..3, prompt]  :125-132
    src = image_embedding.repeat_interleave(T, 0) + dense
    pos = image_pe.repeat_interleave(T, 0)
    b, c, h, wd = src.shape
    hs, keys = two_way_transformer(w, prefix + ".transformer", src, pos, tokens)
    iou_tok, mask_toks = hs[:, 0], hs[:, 1:5]
    up = keys.transpose(1, 2).reshape(b, c, h, wd)
    up = F.conv_transpose2d(up, w[prefix + ".output_upscaling.0.weight"], w[prefix + ".output_upscaling.0.bias"], stride=2)
    up = F.gelu(_ln2d(up, w, prefix + ".output_upscaling.1"))
    up = F.gelu(F.conv_transpose2d(up, w[prefix + ".output_upscaling.3.weight"], w[prefix + ".output_upscaling.3.bias"], stride=2))
    hyper = torch.stack([_mlp3(w, "%s.output_hypernetworks_mlps.%d" % (prefix, i), mask_toks[:, i]) for i in range(4)], 1)
    masks = (hyper @ up.flatten(2)).reshape(b, 4, up.shape[2], up.shape[3])
    iou = _mlp3(w, prefix + ".iou_prediction_head", iou_tok)
    sl = slice(1, None) if multimask_output else slice(0, 1)  # :106-111
    return masks[:, sl], iou[:, sl]


def postprocess_masks(masks, img_size, input_size, original_size):
    """sam.py:137-172: fp32 bilinear to the padded square, crop to the resized extent, bilinear to the original."""
    m = F.interpolate(masks.float(), (img_size, img_size), mode="bilinear", align_corners=False)
    m = m[..., : input_size[0], : input_size[1]]
    return F.interpolate(m, tuple(original_size), mode="bilinear", align_corners=False)


def mask_score(pred_mask):
    """walkgpt.py:540-542 / :737: mean sigmoid over positive-logit pixels.  pred_mask [T,H,W]."""
    pos = (pred_mask > 0).flatten(1)
    return (pred_mask.sigmoid().flatten(1) * pos).sum(1) / (pos.sum(1) + 1e-6)
