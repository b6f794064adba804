"""ORACLE (test infrastructure, never the product path): CPU fp32 restatement of the CLIP ViT-L/14 vision tower as
WalkGPT drives it.

The transformer arithmetic is third-party: `transformers` CLIPVisionModel, pinned 4.31.0 by the reference
(/root/reference/requirements.txt:195) and NOT vendored under /root/reference.  Restated here from its published
algorithm (pre-LN ViT: bias-free patch conv, CLS token first, learned position table, `pre_layrnorm`, blocks of
LN(eps 1e-5) -> MHA (q scaled by hd^-0.5 after projection, additive mask before softmax) -> residual ->
LN -> fc1 -> quick-GELU x*sigmoid(1.702x) -> fc2 -> residual; hidden_states[0] is the pre_layrnorm output).
The reference's own additions are restated from its call sites:
  resize_position_table()  model/llava_walkgpt/model/multimodal_encoder/clip_encoder.py:38-55 (quirk kept as written)
  key-padding mask         model/llava_walkgpt/model/multimodal_encoder/custom_clip.py:27-38, 50-104
  patch-mask construction  model/llava_walkgpt/model/llava_arch.py:160-193
  layer selection          clip_encoder.py:61-69  (hidden_states[select_layer][:,1:], [hidden_states[-11][:,1:]])

PARITY PIN: the reference has no tests or fixtures at this boundary and its package does not import here (transformers 5.15
removed the class its wrapper subclasses).  tests/golden/clip_*.npz pins the block arithmetic against transformers 5.15
CLIPVisionModel (eager attention, quick_gelu) as a stand-in for the pinned 4.31.  The reference's own additions ARE pinned to its
source: tests/golden/clipwrap_*.npz holds the outputs of its encode_images (patch mask for the tower, token mask for the LLM),
_expand_mask (additive key mask) and CLIPVisionTower.load_model (position-table resize) run on synthetic inputs, each module loaded
outside its package (tests/golden/make_golden.py:make_clipwrap); tests/test_oracle_golden.py holds this file to them bit for bit.

Weight names follow transformers' CLIPVisionModel state_dict (`vision_model.*`).
"""
import torch
import torch.nn.functional as F


def _ln(x, w, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + ".weight"], w[prefix + ".bias"], eps)


def _lin(x, w, prefix):
    return F.linear(x, w[prefix + ".weight"], w.get(prefix + ".bias"))


def resize_position_table(table, new_side):
    """clip_encoder.py:38-55.  table [old_side^2 + 1, D] -> [new_side^2 + 1, D].
    As written in the reference the LAST row is treated as the class position and rows [:-1] as the patch grid
    (HF stores the class position first), and the result is [interpolated grid rows || old last row]."""
    n, D = table.shape
    old_side = int(round((n - 1) ** 0.5))
    grid = table[:-1].permute(1, 0).reshape(1, D, old_side, old_side)
    new = F.interpolate(grid, (new_side, new_side), mode="bilinear", align_corners=False)[0]
    return torch.cat([new.flatten(1).permute(1, 0), table[-1:]], 0)


def patch_key_mask(batch, image_hw, clip_resize_list, patch=14):
    """llava_arch.py:160-193: 1 for patches whose top-left pixel lies inside the un-padded (h,w), 0 for padding;
    nearest-neighbour down-sampling of the pixel mask; a leading 1 for the class token.  Returns [B, 1 + P*P]."""
    h, w = image_hw
    m = torch.zeros(batch, h, w)
    sizes = clip_resize_list if clip_resize_list is not None else [(h, w)] * batch
    for i, s in enumerate(sizes):
        m[i, : s[0], : s[1]] = 1
    pn = w // patch
    m = F.interpolate(m[:, None], size=(pn, pn), mode="nearest")[:, 0]
    return torch.cat([torch.ones(batch, 1), m.flatten(1)], 1)


def llm_token_mask(key_mask, side=16):
    """llava_arch.py:176-179: the patch mask (class column dropped) nearest-resampled to the side x side grid of the image tokens
    the LLM sees; spliced into the attention mask by prepare_inputs_labels_for_multimodal.  [B, 1 + P*P] -> [B, side*side]."""
    B = key_mask.shape[0]
    pn = int(round((key_mask.shape[1] - 1) ** 0.5))
    m = key_mask[:, 1:].reshape(B, 1, pn, pn)
    return F.interpolate(m, size=(side, side), mode="nearest")[:, 0].flatten(1)


def clip_hidden_states(w, images, key_mask, heads=16, layers=24, patch=14, prefix="vision_model"):
    """All hidden states [embeddings-after-pre_layrnorm, layer1, ..., layerN], each [B, 1+P, D]."""
    x = F.conv2d(images, w[prefix + ".embeddings.patch_embedding.weight"], stride=patch)
    B, D = x.shape[0], x.shape[1]
    x = x.flatten(2).transpose(1, 2)
    cls = w[prefix + ".embeddings.class_embedding"].reshape(1, 1, D).expand(B, -1, -1)
    x = torch.cat([cls, x], 1) + w[prefix + ".embeddings.position_embedding.weight"][None]
    x = _ln(x, w, prefix + ".pre_layrnorm")
    bias = None
    if key_mask is not None:  # custom_clip.py:27-38: (1 - mask) * finfo.min, broadcast over heads and queries
        bias = ((1.0 - key_mask) * torch.finfo(torch.float32).min)[:, None, None, :]
    hd = D // heads
    states = [x]
    for i in range(layers):
        L = "%s.encoder.layers.%d" % (prefix, i)
        y = _ln(x, w, L + ".layer_norm1")
        q = _lin(y, w, L + ".self_attn.q_proj") * hd ** -0.5
        k, v = _lin(y, w, L + ".self_attn.k_proj"), _lin(y, w, L + ".self_attn.v_proj")

        def split(t):
            return t.reshape(B, -1, heads, hd).transpose(1, 2)

        a = split(q) @ split(k).transpose(2, 3)
        if bias is not None:
            a = a + bias
        o = (a.softmax(-1) @ split(v)).transpose(1, 2).reshape(B, -1, D)
        x = x + _lin(o, w, L + ".self_attn.out_proj")
        y = _lin(_ln(x, w, L + ".layer_norm2"), w, L + ".mlp.fc1")
        x = x + _lin(y * torch.sigmoid(1.702 * y), w, L + ".mlp.fc2")
        states.append(x)
    return states


def clip_tower(w, images, key_mask=None, select_layer=-2, heads=16, layers=24, patch=14):
    """CLIPVisionTower.forward + feature_select (clip_encoder.py:61-98): (features, [features of hidden_states[-11]])."""
    hs = clip_hidden_states(w, images, key_mask, heads, layers, patch)
    return hs[select_layer][:, 1:], [hs[-11][:, 1:]]
