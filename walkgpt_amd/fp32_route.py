"""fp32 VERIFICATION route of the text-logit path: CLIP vision tower -> mm_projector in fp32 storage and exact fp32 matrix math on the GPU
(csrc/fp32_ref.hip).  Not a product path and not benched: the deployed path is bf16 (clip_encoder.py), whose distance to the reference's fp32
CPU run is set by its bf16 weights (0.3-0.7 abs on the logits, the reference's own bf16 run likewise: tests/test_gpu_modules.py).  This route
shows that the arithmetic itself -- patch embedding, pre-LayerNorm, 24 x (LN, q|k|v, masked softmax attention, out_proj, LN, fc1, quick-GELU,
fc2), layer selection, projector -- holds north_star's 1e-4 when nothing is rounded to bf16 (tests/test_gpu_fp32_route.py).

Mirrors custom_clip.py:50-104 (hidden states incl. the key mask of :27-38), clip_encoder.py:61-69 (feature selection) and llava_arch.py:36-42
(projector).  Weights are read from the tower's modules and used as fp32 (a tower held in bf16 is up-cast: its weights are then the bf16
values, and the route reproduces the oracle run on those)."""
import torch

from . import _lib, ops

F32 = torch.float32


def _f32(t):
    return t.detach().to(F32).contiguous()


def linear(x, weight, bias=None, act=ops.ACT_NONE, residual=None, res_row_mod=0):
    """act(x W^T + b) (+ residual), fp32 [..., K] -> [..., N] on v_mfma_f32_16x16x4_f32."""
    assert x.is_cuda and x.dtype == F32 and weight.dtype == F32 and x.is_contiguous() and weight.is_contiguous()
    K = x.shape[-1]
    M = x.numel() // K
    N = weight.shape[0]
    out = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=F32)
    if residual is not None:
        assert residual.dtype == F32 and residual.is_contiguous() and residual.shape[-1] == N
    rc = _lib.lib().wg_f32_gemm_bias_act(x.data_ptr(), K, weight.data_ptr(), weight.stride(0), ops._ptr(bias), ops._ptr(residual), N, res_row_mod,
                                         out.data_ptr(), N, M, N, K, act, ops._stream())
    _lib.check(rc, "wg_f32_gemm_bias_act")
    return out


def layernorm(x, gamma, beta, eps):
    assert x.is_cuda and x.dtype == F32 and x.is_contiguous()
    C = x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(_lib.lib().wg_f32_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), x.numel() // C, C, float(eps), ops._stream()),
               "wg_f32_layernorm")
    return y


def mha(qkv, heads, scale, key_bias=None):
    """qkv [B, L, 3D] fp32 (q | k | v) -> [B, L, D]."""
    B, L, D3 = qkv.shape
    D = D3 // 3
    out = torch.empty(B, L, D, device=qkv.device, dtype=F32)
    base = qkv.data_ptr()
    rc = _lib.lib().wg_f32_mha(base, base + 4 * D, base + 8 * D, out.data_ptr(), ops._ptr(key_bias), D3, D, B, heads, D // heads, L, L, float(scale),
                               ops._stream())
    _lib.check(rc, "wg_f32_mha")
    return out


@torch.no_grad()
def clip_hidden_states(vision_model, pixel_values, attention_mask, want):
    """{index: hidden state [B, 1+P, D] fp32} of walkgpt_amd.clip_encoder._CLIPVisionTransformer `vision_model`, as its hidden_states() computes
    them (hidden_states[0] = pre_layrnorm output), with every tensor and every sum in fp32.  pixel_values fp32 [B, 3, S, S]."""
    e = vision_model.embeddings
    assert pixel_values.is_cuda and pixel_values.dtype == F32
    B, _, S, _ = pixel_values.shape
    ps, D = e.patch_size, e.embed_dim
    g = S // ps
    P = g * g
    # patch rows in conv-weight column order (c, ky, kx): a pure re-layout of the pixels
    rows = pixel_values.reshape(B, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * P, 3 * ps * ps).contiguous()
    K = rows.shape[1]
    kpad = (K + 3) // 4 * 4
    wp = _f32(e.patch_embedding.weight).reshape(D, K)
    if kpad != K:
        rows = torch.nn.functional.pad(rows, (0, kpad - K))
        wp = torch.nn.functional.pad(wp, (0, kpad - K))
    pos = _f32(e.position_embedding.weight)
    assert pos.shape[0] == P + 1, "position table has %d rows, the input needs %d" % (pos.shape[0], P + 1)
    x = torch.empty(B, P + 1, D, device=pixel_values.device, dtype=F32)
    x[:, 1:] = linear(rows.contiguous(), wp.contiguous(), residual=pos[1:].contiguous(), res_row_mod=P).reshape(B, P, D)
    x[:, 0] = _f32(e.class_embedding) + pos[0]
    x = layernorm(x, _f32(vision_model.pre_layrnorm.weight), _f32(vision_model.pre_layrnorm.bias), vision_model.pre_layrnorm.eps)
    key_bias = None
    if attention_mask is not None:      # custom_clip.py:27-38: (1 - mask) * finfo.min on padded keys
        key_bias = ((1.0 - attention_mask.to(x.device, F32)) * torch.finfo(F32).min).contiguous()
    layers = vision_model.encoder.layers
    n = len(layers)
    idx = {(i if i >= 0 else n + 1 + i) for i in want}
    keep = {0: x} if 0 in idx else {}
    for i in range(max(idx)):
        lyr = layers[i]
        a, m = lyr.self_attn, lyr.mlp
        heads = a.num_heads
        y = layernorm(x, _f32(lyr.layer_norm1.weight), _f32(lyr.layer_norm1.bias), lyr.layer_norm1.eps)
        wqkv = torch.cat([_f32(a.q_proj.weight), _f32(a.k_proj.weight), _f32(a.v_proj.weight)], 0)
        bqkv = torch.cat([_f32(a.q_proj.bias), _f32(a.k_proj.bias), _f32(a.v_proj.bias)], 0)
        qkv = linear(y, wqkv, bqkv)
        o = mha(qkv, heads, (D // heads) ** -0.5, key_bias)
        x = linear(o, _f32(a.out_proj.weight), _f32(a.out_proj.bias), residual=x)
        y = layernorm(x, _f32(lyr.layer_norm2.weight), _f32(lyr.layer_norm2.bias), lyr.layer_norm2.eps)
        h = linear(y, _f32(m.fc1.weight), _f32(m.fc1.bias), act=ops.ACT_QUICK_GELU)
        x = linear(h, _f32(m.fc2.weight), _f32(m.fc2.bias), residual=x)
        if i + 1 in idx:
            keep[i + 1] = x
    return {w_: keep[w_ if w_ >= 0 else n + 1 + w_] for w_ in want}


@torch.no_grad()
def clip_features(tower, images, attention_mask=None):
    """CLIPVisionTower.forward + feature_select (clip_encoder.py:61-98) in fp32: (features, [features of hidden_states[-11]])."""
    hs = clip_hidden_states(tower.vision_tower.vision_model, images, attention_mask, [tower.select_layer, -11])
    return hs[tower.select_layer][:, 1:].contiguous(), [hs[-11][:, 1:].contiguous()]


@torch.no_grad()
def mm_project(feats, projector):
    """llava_arch.py:36-42: nn.Linear or Linear -> GELU -> Linear, in fp32."""
    if isinstance(projector, torch.nn.Linear):
        return linear(feats.contiguous(), _f32(projector.weight), None if projector.bias is None else _f32(projector.bias))
    x = linear(feats.contiguous(), _f32(projector[0].weight), _f32(projector[0].bias), act=ops.ACT_GELU)
    return linear(x, _f32(projector[2].weight), _f32(projector[2].bias))


# ---------------------------------------------------------------------------------------------------------------------------------------------
# Path A of the text logits (model/walkgpt.py:313-330): SAM image encoder -> MSQP -> 6x6 -> 16x16 resample -> splice -> language model.
# Same idea as above: fp32 storage, every matrix product on the exact-fp32 MFMA kernels of csrc/fp32_ref.hip, LayerNorm / softmax in fp32;
# data movement (window partition, padding, pooling, bilinear resample, the rel-pos table gather and its two small contractions, the splice)
# in stock fp32 torch on the GPU.  Weights come as the reference's state_dict (flat {key: fp32 CUDA tensor} relative to the module root), which
# is also how the oracle takes them (oracle/sam.py, oracle/projectors.py).
# ---------------------------------------------------------------------------------------------------------------------------------------------
def mha_ex(q, k, v, ldq, ldkv, B, heads, hd, Lq, Lk, scale, attn_bias=None, key_bias=None):
    """q: tensor whose storage holds the query rows [B, Lq, ldq] starting at q's first element; k / v likewise with pitch ldkv.  -> [B, Lq, heads * hd]."""
    out = torch.empty(B, Lq, heads * hd, device=q.device, dtype=F32)
    if attn_bias is not None:
        assert attn_bias.dtype == F32 and attn_bias.is_contiguous() and attn_bias.shape == (B, heads, Lq, Lk)
    rc = _lib.lib().wg_f32_mha_ex(q.data_ptr(), ldq, k.data_ptr(), v.data_ptr(), ldkv, out.data_ptr(), heads * hd, ops._ptr(key_bias), ops._ptr(attn_bias),
                                  B, heads, hd, Lq, Lk, float(scale), ops._stream())
    _lib.check(rc, "wg_f32_mha_ex")
    return out


def _lin(w, prefix, x, act=ops.ACT_NONE, residual=None):
    wt = w[prefix + ".weight"]
    wt = wt.reshape(wt.shape[0], -1)
    K = wt.shape[1]
    x2 = x.reshape(-1, K)
    if K % 4:      # the MFMA kernel walks K in fours
        pad = 4 - K % 4
        x2, wt = torch.nn.functional.pad(x2, (0, pad)), torch.nn.functional.pad(wt, (0, pad))
    r2 = None if residual is None else residual.reshape(-1, wt.shape[0]).contiguous()
    y = linear(x2.contiguous(), wt.contiguous(), w.get(prefix + ".bias"), act=act, residual=r2)
    return y.reshape(x.shape[:-1] + (wt.shape[0],))


def _ln(w, prefix, x, eps):
    return layernorm(x.contiguous(), w[prefix + ".weight"], w[prefix + ".bias"], eps)


def _sam_attention(w, prefix, x, heads):
    """image_encoder.py:235-260 on a window batch or whole images x [G, S, S, D]."""
    G, S, _, D = x.shape
    hd = D // heads
    L = S * S
    qkv = _lin(w, prefix + ".qkv", x.reshape(G, L, D))      # [G, L, 3 D], columns (3, heads, hd) (:238-242)
    bias = None
    if (prefix + ".rel_pos_h") in w:      # decomposed relative position from the UNSCALED queries (:247-249, :321-392; equal grids: no table resize)
        idx = torch.arange(S, device=x.device)[:, None] - torch.arange(S, device=x.device)[None, :] + (S - 1)
        Rh, Rw = w[prefix + ".rel_pos_h"][idx], w[prefix + ".rel_pos_w"][idx]      # [S(q), S(k), hd]
        qg = qkv[:, :, :D].reshape(G, S, S, heads, hd)
        bh = torch.einsum("gyxhc,ykc->ghyxk", qg, Rh)
        bw = torch.einsum("gyxhc,xkc->ghyxk", qg, Rw)
        bias = (bh[..., :, None] + bw[..., None, :]).reshape(G, heads, L, L).contiguous()
    o = mha_ex(qkv, qkv[0, 0, D:], qkv[0, 0, 2 * D:], 3 * D, 3 * D, G, heads, hd, L, L, hd ** -0.5, attn_bias=bias)
    return _lin(w, prefix + ".proj", o).reshape(G, S, S, D)


@torch.no_grad()
def sam_image_encoder(w, images, cfg, prefix="image_encoder"):
    """image_encoder.py:110-125: images fp32 [B, 3, S, S] -> [B, out, S/p, S/p].  cfg: patch, depth, heads, global_idx, window."""
    Fn = torch.nn.functional
    p = cfg["patch"]
    B, _, S, _ = images.shape
    g = S // p
    D = w[prefix + ".pos_embed"].shape[-1]
    rows = images.reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * p * p)      # (c, ky, kx): the conv weight's own order (:422-426)
    x = _lin(w, prefix + ".patch_embed.proj", rows) + w[prefix + ".pos_embed"].reshape(1, g * g, D)       # :111-113
    x = x.reshape(B, g, g, D)
    for i in range(cfg["depth"]):
        pre = "%s.blocks.%d" % (prefix, i)
        win = 0 if i in cfg["global_idx"] else cfg["window"]
        y = _ln(w, pre + ".norm1", x, 1e-6)
        if win > 0:      # window_partition / unpartition, :263-318: zeros AFTER norm1, so pad tokens' q / k / v equal the qkv bias
            ph, pw = (-g) % win, (-g) % win
            y = Fn.pad(y, (0, 0, 0, pw, 0, ph))
            Hp, Wp = g + ph, g + pw
            y = y.reshape(B, Hp // win, win, Wp // win, win, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, win, win, D).contiguous()
            y = _sam_attention(w, pre + ".attn", y, cfg["heads"])
            y = y.reshape(B, Hp // win, Wp // win, win, win, D).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, D)[:, :g, :g]
        else:
            y = _sam_attention(w, pre + ".attn", y, cfg["heads"])
        x = x + y
        h = _lin(w, pre + ".mlp.lin1", _ln(w, pre + ".norm2", x, 1e-6), act=ops.ACT_GELU)      # common.py:13-26
        x = _lin(w, pre + ".mlp.lin2", h, residual=x)
    # neck (:92-108): 1x1 conv (no bias), LayerNorm2d = LayerNorm over the channels of every token (common.py:31-43), 3x3 conv (pad 1), LayerNorm2d
    t = _ln(w, prefix + ".neck.1", _lin(w, prefix + ".neck.0", x), 1e-6)
    C = t.shape[-1]
    cols = Fn.unfold(t.permute(0, 3, 1, 2), 3, padding=1)      # [B, C * 9, g * g] in (c, ky, kx) order = the conv weight's own order
    t = _lin(w, prefix + ".neck.2", cols.transpose(1, 2).contiguous())
    t = _ln(w, prefix + ".neck.3", t, 1e-6)
    return t.reshape(B, g, g, C).permute(0, 3, 1, 2).contiguous()


def _cross_block(w, prefix, q, kv, heads):
    """CrossAttnBlock.forward, utils_walkgpt.py:175-185 (nn.MultiheadAttention, packed in_proj, batch_first, dropout 0)."""
    D = q.shape[-1]
    hd = D // heads
    Wi, bi = w[prefix + ".attn.in_proj_weight"], w[prefix + ".attn.in_proj_bias"]
    qn, kn = _ln(w, prefix + ".q_norm", q, 1e-5), _ln(w, prefix + ".kv_norm", kv, 1e-5)
    B, Nq, Lk = q.shape[0], q.shape[1], kv.shape[1]
    qp = linear(qn.reshape(-1, D), Wi[:D].contiguous(), bi[:D].contiguous()).reshape(B, Nq, D)
    kvp = linear(kn.reshape(-1, D), Wi[D:].contiguous(), bi[D:].contiguous()).reshape(B, Lk, 2 * D)      # k | v
    o = mha_ex(qp, kvp, kvp[0, 0, D:], D, 2 * D, B, heads, hd, Nq, Lk, hd ** -0.5)
    out = _lin(w, prefix + ".attn.out_proj", o, residual=q.contiguous())
    h = _lin(w, prefix + ".ffn.1", _ln(w, prefix + ".ffn.0", out, 1e-5), act=ops.ACT_GELU)
    return _lin(w, prefix + ".ffn.3", h, residual=out)


@torch.no_grad()
def msqp(w, sam_tokens, heads=8, side=6):
    """MultiScaleQFormerProjector.forward, utils_walkgpt.py:259-300: sam_tokens fp32 [B, L, sam_dim] -> [B, side * side, llama_dim]."""
    Fn = torch.nn.functional
    B, L, _ = sam_tokens.shape
    H = int(round(L ** 0.5))
    assert H * H == L
    f = _lin(w, "sam_to_proj", sam_tokens)
    C = f.shape[-1]
    grid = f.reshape(B, H, H, C).permute(0, 3, 1, 2)

    def pool(s):      # _pool_grid_tokens :195-201
        return Fn.avg_pool2d(grid, s, s).permute(0, 2, 3, 1).reshape(B, -1, C).contiguous()

    outs = []
    for qname, cname, kv in (("q_x1", "cross_x1", f), ("q_x2", "cross_x2", pool(2)), ("q_x4", "cross_x4", pool(4)),
                             ("q_global", "cross_glb", f.mean(1, keepdim=True))):
        logit = _lin(w, "gate.net.3", _lin(w, "gate.net.1", _ln(w, "gate.net.0", kv, 1e-5), act=ops.ACT_GELU))      # SegAwareGate :213-217, one gate for all scales (:276)
        kv = kv * torch.sigmoid(logit)
        q = w[qname].expand(B, -1, -1).contiguous()
        for layer in range(2):
            q = _cross_block(w, "%s.%d" % (cname, layer), q, kv.contiguous(), heads)
        outs.append(q)
    vis = torch.cat(outs, 1)      # [x1 (12), x2 (8), x4 (8), glb (4)]  :290
    pad = side * side - vis.shape[1]
    if pad > 0:
        vis = torch.cat([vis, w["pad_token"].expand(B, pad, -1)], 1)
    return _lin(w, "to_llama", vis.contiguous())


@torch.no_grad()
def resample_tokens(tokens, target=16):
    """llava_arch.py:252-259: [n, p * p, C] -> bilinear (align_corners False) -> [n, target * target, C], fp32 throughout."""
    n, l, c = tokens.shape
    p = int(round(l ** 0.5))
    g = tokens.permute(0, 2, 1).reshape(n, c, p, p)
    g = torch.nn.functional.interpolate(g, size=(target, target), mode="bilinear", align_corners=False)
    return g.flatten(2).permute(0, 2, 1).contiguous()


@torch.no_grad()
def splice_rows(input_ids, image_features, embed_weight, image_token=-200):
    """prepare_inputs_labels_for_multimodal (llava_arch.py:265-518) for rows with exactly one image placeholder and no padding mask: the row's
    token embeddings with the placeholder replaced by its image's feature rows.  -> (attention mask [rows, L'], embeds [rows, L', H])."""
    rows = []
    for r in range(input_ids.shape[0]):
        ids = input_ids[r]
        pos = (ids == image_token).nonzero()
        assert pos.numel() == 1, "one image placeholder per row"
        p = int(pos[0, 0])
        rows.append(torch.cat([embed_weight[ids[:p]], image_features[r], embed_weight[ids[p + 1:]]], 0))
    emb = torch.stack(rows, 0)
    return torch.ones(emb.shape[:2], dtype=torch.bool, device=emb.device), emb
