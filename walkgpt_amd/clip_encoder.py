"""MI355X-native CLIP ViT-L/14 vision tower behind the reference's CLIPVisionTower surface
(/root/reference/model/llava_walkgpt/model/multimodal_encoder/clip_encoder.py:7-135 and custom_clip.py:50-143).

The parameter tree reproduces transformers' CLIPVisionModel names (`vision_tower.vision_model.embeddings.*`,
`...encoder.layers.N.self_attn.{q,k,v,out}_proj`, `layer_norm1/2`, `mlp.fc1/fc2`, `pre_layrnorm`, `post_layernorm`)
so `model.vision_tower.vision_tower.vision_model.*` checkpoints load unchanged; the arithmetic is walkgpt_amd.ops (HIP):
patch gather + GEMM, fused q|k|v GEMM, flash attention with the additive key-padding mask, quick-GELU MLP.
There is no network in this environment, so `load_model()` builds the architecture from a config dict instead of
downloading `vision_tower_name`; weights then come from `load_state_dict`.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .segment_anything.modeling import _check_bf16_gpu, _Prepared

BF16 = torch.bfloat16

CLIP_VIT_L_14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                     image_size=336, patch_size=14, layer_norm_eps=1e-5)


def resize_position_table(table, new_side):
    """clip_encoder.py:38-55, quirk kept as written: rows [:-1] are taken as the patch grid and bilinearly resized
    (align_corners=False), the LAST row is carried over and appended.  table [old^2+1, D] bf16 GPU -> [new^2+1, D]."""
    n, D = table.shape
    grid = table[:-1].reshape(1, n - 1, D).to(BF16).contiguous()
    new = ops.resample_tokens(grid, new_side)[0]
    return torch.cat([new, table[-1:].to(BF16)], dim=0)


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        D, P = cfg.hidden_size, cfg.patch_size
        self.embed_dim, self.patch_size, self.image_size = D, P, cfg.image_size
        self.class_embedding = nn.Parameter(torch.randn(D))
        self.patch_embedding = nn.Conv2d(3, D, kernel_size=P, stride=P, bias=False)
        self.num_patches = (cfg.image_size // P) ** 2
        self.num_positions = self.num_patches + 1
        self.position_embedding = nn.Embedding(self.num_positions, D)
        self.register_buffer("position_ids", torch.arange(self.num_positions).expand((1, -1)), persistent=False)


class _SelfAttn(nn.Module):
    def __init__(self, D, heads):
        super().__init__()
        self.num_heads, self.head_dim = heads, D // heads
        self.scale = self.head_dim ** -0.5
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = nn.Linear(D, D), nn.Linear(D, D), nn.Linear(D, D), nn.Linear(D, D)


class _Mlp(nn.Module):
    def __init__(self, D, inner):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(D, inner), nn.Linear(inner, D)


class _EncoderLayer(nn.Module, _Prepared):
    def __init__(self, cfg):
        super().__init__()
        D = cfg.hidden_size
        self.self_attn = _SelfAttn(D, cfg.num_attention_heads)
        self.layer_norm1 = nn.LayerNorm(D, eps=cfg.layer_norm_eps)
        self.mlp = _Mlp(D, cfg.intermediate_size)
        self.layer_norm2 = nn.LayerNorm(D, eps=cfg.layer_norm_eps)

    def _build(self):
        a, n1, n2, m = self.self_attn, self.layer_norm1, self.layer_norm2, self.mlp
        w = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).contiguous()
        b = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0).contiguous()
        return {"qkv": ops.fold_layernorm(n1.weight, n1.bias, w, b),
                "fc1": ops.fold_layernorm(n2.weight, n2.bias, m.fc1.weight, m.fc1.bias)}

    gemm_dtype = "bf16"   # "fp8": q|k|v / out_proj / fc1 / fc2 on e4m3 operands (BASELINE config C5); set through WalkGPTGrounding.set_gemm_dtype

    def _build_fp8(self):
        a, m = self.self_attn, self.mlp
        wqkv = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).contiguous()
        return {"qkv": ops.quantize_weight_fp8(wqkv), "bqkv": torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0).contiguous(),
                "out": ops.quantize_weight_fp8(a.out_proj.weight), "fc1": ops.quantize_weight_fp8(m.fc1.weight),
                "fc2": ops.quantize_weight_fp8(m.fc2.weight)}

    def _build_mx(self):
        a, n1, n2, m = self.self_attn, self.layer_norm1, self.layer_norm2, self.mlp
        w = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).contiguous()
        b = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0).contiguous()
        return {"qkv": ops.fold_layernorm_mx(n1.weight, n1.bias, w, b), "out": ops.mx_weight(a.out_proj.weight),
                "fc1": ops.fold_layernorm_mx(n2.weight, n2.bias, m.fc1.weight, m.fc1.bias), "fc2": ops.mx_weight(m.fc2.weight)}

    def run_mx(self, x, key_bias):
        """The layer as one MX chain on the persistent fp8 GEMM (as Block.rows_mx of the SAM encoder)."""
        a, m = self.self_attn, self.mlp
        w = self._prep_get(self._build_mx, slot="_mx")
        D = x.shape[-1]
        qkv = ops.linear_mxfp8(x, w["qkv"], ln_eps=self.layer_norm1.eps)
        o = ops.mha(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], a.num_heads, a.scale, key_bias, small=False)
        x = ops.linear_mxfp8(ops.quantize_mx_fp8(o), w["out"], bias=a.out_proj.bias, residual=x, mx_out=True, row_partials=True)
        h = ops.linear_mxfp8(x, w["fc1"], act=ops.ACT_QUICK_GELU, ln_eps=self.layer_norm2.eps, mx_out=True, bf16_out=False)
        return ops.linear_mxfp8(h, w["fc2"], bias=m.fc2.bias, residual=x, mx_out=True, row_partials=True)

    def run_fp8(self, x, key_bias):
        """The layer on the fp8 GEMM path (as Block.rows_fp8 of the SAM encoder): the MX chain where the widths allow it, otherwise per-row
        activation scales with both LayerNorms fused into the quantisation of their output.  Attention, the residual stream and every
        statistic stay bf16 / fp32."""
        a, m = self.self_attn, self.mlp
        if ops.mx_chain_ok(x.shape[-1], m.fc1.weight.shape[0]) and ops.mx_prepare_rows(x):
            return self.run_mx(x, key_bias)
        key = tuple((p.data_ptr(), p._version) for p in (a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.out_proj.weight, m.fc1.weight,
                                                         m.fc2.weight, a.q_proj.bias, a.k_proj.bias, a.v_proj.bias))
        w = self.__dict__.get("_fp8_val")
        if w is None or self.__dict__.get("_fp8_key") != key:
            w = self._build_fp8()
            self.__dict__["_fp8_val"], self.__dict__["_fp8_key"] = w, key
        D = x.shape[-1]
        q, s = ops.quantize_rows_fp8(x, ln=(self.layer_norm1.weight, self.layer_norm1.bias), eps=self.layer_norm1.eps)
        qkv = ops.linear_fp8(q, s, *w["qkv"], bias=w["bqkv"])
        o = ops.mha(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], a.num_heads, a.scale, key_bias, small=False)
        q, s = ops.quantize_rows_fp8(o)
        x = ops.linear_fp8(q, s, *w["out"], bias=a.out_proj.bias, residual=x)
        q, s = ops.quantize_rows_fp8(x, ln=(self.layer_norm2.weight, self.layer_norm2.bias), eps=self.layer_norm2.eps)
        q, s = ops.quantize_rows_fp8(ops.linear_fp8(q, s, *w["fc1"], bias=m.fc1.bias, act=ops.ACT_QUICK_GELU))
        return ops.linear_fp8(q, s, *w["fc2"], bias=m.fc2.bias, residual=x)

    def run(self, x, key_bias, tail_tiles=False):
        """HF CLIPEncoderLayer: pre-LN attention + pre-LN quick-GELU MLP, both residual.  x [B, L, D].
        Both LayerNorms are folded into the GEMM behind them (ops.ln_linear).  tail_tiles: let the M = B*1025 GEMMs use the
        tail-absorbing tiles (faster when nothing else shares the GPU, slower under stream overlap)."""
        if self.gemm_dtype == "fp8":
            return self.run_fp8(x, key_bias)
        a = self.self_attn
        p = self._prep_get(self._build)
        D = x.shape[-1]
        qkv = ops.ln_linear(x, p["qkv"], self.layer_norm1.eps, tail_tiles=tail_tiles)
        o = ops.mha(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], a.num_heads, a.scale, key_bias, small=False)
        x = ops.linear(o, a.out_proj.weight, a.out_proj.bias, residual=x, tail_tiles=tail_tiles, row_partials=True)
        h = ops.ln_linear(x, p["fc1"], self.layer_norm2.eps, act=ops.ACT_QUICK_GELU, tail_tiles=tail_tiles)
        return ops.linear(h, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=x, tail_tiles=tail_tiles, row_partials=True)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(cfg) for _ in range(cfg.num_hidden_layers)])


class _CLIPVisionTransformer(nn.Module, _Prepared):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = _Embeddings(cfg)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.encoder = _Encoder(cfg)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def _build(self):
        e = self.embeddings
        K = 3 * e.patch_size * e.patch_size
        kpad = ((K + 63) // 64) * 64
        w = torch.zeros(e.embed_dim, kpad, device=e.patch_embedding.weight.device, dtype=e.patch_embedding.weight.dtype)
        w[:, :K] = e.patch_embedding.weight.reshape(e.embed_dim, K)
        return {"patch_w": w, "kpad": kpad}

    def hidden_states(self, pixel_values, attention_mask, want, run_all_layers=True, tail_tiles=False):
        """{index: hidden state [B, 1+P, D]} for the (python-style, possibly negative) indices in `want`;
        hidden_states[0] is the pre_layrnorm output (custom_clip.py:74-92)."""
        _check_bf16_gpu(pixel_values, "images_clip")
        _check_bf16_gpu(self.pre_layrnorm.weight, "CLIP weights")
        e = self.embeddings
        p = self._prep_get(self._build)
        B = pixel_values.shape[0]
        P = (pixel_values.shape[-1] // e.patch_size) ** 2
        pos = e.position_embedding.weight
        if pos.shape[0] != P + 1:
            raise RuntimeError("position table has %d rows, the input needs %d (resize_position_table first)" % (pos.shape[0], P + 1))
        D = e.embed_dim
        rows = ops.patchify(pixel_values.contiguous(), e.patch_size, p["kpad"])
        x = torch.empty(B, P + 1, D, device=pixel_values.device, dtype=BF16)
        for b in range(B):  # rows of image b land directly behind its class token
            ops.linear(rows[b * P:(b + 1) * P], p["patch_w"], residual=pos[1:], out=x[b, 1:])
        x[:, 0] = ops.add_rows(e.class_embedding.reshape(1, D), pos[:1])
        x = ops.layernorm(x, self.pre_layrnorm.weight, self.pre_layrnorm.bias, self.pre_layrnorm.eps)
        key_bias = None
        if attention_mask is not None:  # custom_clip.py:27-38: (1 - mask) * finfo.min on padded keys
            key_bias = torch.where(attention_mask > 0.5, 0.0, torch.finfo(torch.float32).min).float().contiguous()
        n = len(self.encoder.layers)
        idx = {(i if i >= 0 else n + 1 + i) for i in want}
        keep = {0: x} if 0 in idx else {}
        last = n if run_all_layers else max(idx)
        for i in range(last):
            x = self.encoder.layers[i].run(x, key_bias, tail_tiles)
            if i + 1 in idx:
                keep[i + 1] = x
        return {w_: keep[w_ if w_ >= 0 else n + 1 + w_] for w_ in want}


class _CLIPVisionModel(nn.Module):
    """Stand-in for transformers' CLIPVisionModel / the reference's _CLIPVisionModel (parameter names only)."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.vision_model = _CLIPVisionTransformer(cfg)
        self._register_load_state_dict_pre_hook(self._drop_position_ids)

    @staticmethod
    def _drop_position_ids(state_dict, prefix, *args):
        state_dict.pop(prefix + "vision_model.embeddings.position_ids", None)  # persistent only in transformers<=4.31

    @property
    def dtype(self):
        return self.vision_model.pre_layrnorm.weight.dtype

    @property
    def device(self):
        return self.vision_model.pre_layrnorm.weight.device


class CLIPVisionTower(nn.Module):
    """clip_encoder.py:7-135."""

    def __init__(self, vision_tower, args, delay_load=False, config=None):
        super().__init__()
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = args.mm_vision_select_layer
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        self.pad_vit = getattr(args, "pad_train_clip_images", False)
        self.resize_vision_tower = getattr(args, "resize_vision_tower", False)
        self.resize_vision_tower_size = getattr(args, "resize_vision_tower_size", 224)
        self.run_all_layers = getattr(args, "clip_run_all_layers", True)
        self._cfg_dict = dict(CLIP_VIT_L_14 if config is None else config)
        self.cfg_only = SimpleNamespace(**self._cfg_dict)
        if not delay_load:
            self.load_model()

    def load_model(self):
        cfg = dict(self._cfg_dict)
        if self.resize_vision_tower:
            cfg["image_size"] = self.resize_vision_tower_size  # the table is created at the run size (clip_encoder.py:38-55)
        self.vision_tower = _CLIPVisionModel(SimpleNamespace(**cfg))
        self.vision_tower.requires_grad_(False)
        if self.resize_vision_tower:
            self.vision_tower._register_load_state_dict_pre_hook(self._resize_loaded_position_table)
        self.is_loaded = True

    def _resize_loaded_position_table(self, state_dict, prefix, *args):
        """The reference loads the stock checkpoint (577-row table at 336 px) and THEN resizes it (clip_encoder.py:38-55); here the module
        is created at the run size, so a checkpoint that still carries the stock table is resized on its way in -- with the
        reference's own arithmetic (rows [:-1] taken as the grid, last row carried over; resize_position_table)."""
        key = prefix + "vision_model.embeddings.position_embedding.weight"
        t = state_dict.get(key)
        want = self.vision_tower.vision_model.embeddings.position_embedding.weight.shape[0]
        if t is None or t.shape[0] == want:
            return
        side = int(round((want - 1) ** 0.5))
        # One-time load step, the same arithmetic wherever the tensor sits: the reference interpolates the stock table in its own
        # precision and casts afterwards (clip_encoder.py:38-55), so fp32 bilinear here and ONE rounding at the end -- a bf16 round
        # trip through the HIP resample kernel on the GPU branch would give a checkpoint two different tables depending on its device.
        n, D = t.shape
        old = int(round((n - 1) ** 0.5))
        grid = t[:-1].float().reshape(1, old, old, D).permute(0, 3, 1, 2)
        new = torch.nn.functional.interpolate(grid, size=(side, side), mode="bilinear", align_corners=False)
        state_dict[key] = torch.cat([new.permute(0, 2, 3, 1).reshape(side * side, D), t[-1:].float()], 0).to(t.dtype)

    def feature_select(self, states):
        feats = states[self.select_layer]
        if self.select_feature == "patch":
            feats = feats[:, 1:]
        elif self.select_feature != "cls_patch":
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        return feats, [states[-11][:, 1:]]

    @torch.no_grad()
    def forward(self, images, attention_mask=None, tail_tiles=False):
        if type(images) is list:
            raise NotImplementedError("list-of-images input is not on the WalkGPT path (llava_arch.py:236-243 asserts it away)")
        states = self.vision_tower.vision_model.hidden_states(images, attention_mask, [self.select_layer, -11],
                                                              self.run_all_layers, tail_tiles)
        return self.feature_select(states)

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return self.vision_tower.dtype

    @property
    def device(self):
        return self.vision_tower.device

    @property
    def config(self):
        return self.vision_tower.config if self.is_loaded else self.cfg_only

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2


def llm_token_mask(key_mask, side=16):
    """llava_arch.py:176-179 (host side): the patch mask without its class column, nearest-resampled to the side x side grid of
    the image tokens the LLM sees -> [B, side*side]; the `vit_attention_mask` of llava_splice.prepare_inputs_labels_for_multimodal."""
    B = key_mask.shape[0]
    pn = int(round((key_mask.shape[1] - 1) ** 0.5))
    m = key_mask[:, 1:].reshape(B, 1, pn, pn).float()
    return torch.nn.functional.interpolate(m, size=(side, side), mode="nearest")[:, 0].flatten(1)


def patch_key_mask(images, clip_resize_list=None, patch_size=14):
    """llava_arch.py:160-193 (mask bookkeeping, host side): [B, 1 + P*P] float, 1 = real patch / class token."""
    B = images.shape[0]
    h, w = images.shape[-2:]
    pn = w // patch_size
    sizes = clip_resize_list if clip_resize_list is not None else [(h, w)] * B
    m = torch.zeros(B, h, w, dtype=torch.float32)
    for i, s in enumerate(sizes):
        m[i, : s[0], : s[1]] = 1
    m = torch.nn.functional.interpolate(m[:, None], size=(pn, pn), mode="nearest")[:, 0]
    return torch.cat([torch.ones(B, 1), m.flatten(1)], dim=-1).to(images.device)
