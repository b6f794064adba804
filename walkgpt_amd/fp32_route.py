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
