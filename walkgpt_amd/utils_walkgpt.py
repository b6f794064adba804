"""MI355X-native Multi-Scale Query Projector (MSQP) and Calibrated Text Projector (CTP) behind the reference's class
names, constructor arguments and state_dict keys (/root/reference/utils/utils_walkgpt.py:163-327).

The nn.* children are parameter containers only; the forward path is walkgpt_amd.ops (HIP).  Differences from the
reference that do not change results: images are processed as one batch (the reference loops per image,
model/walkgpt.py:364-378), and CTP can be applied to gathered [SEG] rows only (it is a per-token map).
"""
import math

import torch
import torch.nn as nn

from . import ops
from .segment_anything.modeling import _check_bf16_gpu, _Prepared

BF16 = torch.bfloat16


def _ln(x, n):
    return ops.layernorm(x, n.weight, n.bias, n.eps)


class CrossAttnBlock(nn.Module):
    """utils_walkgpt.py:163-185."""

    def __init__(self, d_model, nhead, mlp_ratio=4.0, dropout=0.0):
        super().__init__()
        self.q_norm = nn.LayerNorm(d_model)
        self.kv_norm = nn.LayerNorm(d_model)
        self.attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.proj_drop = nn.Dropout(dropout)
        self.ffn = nn.Sequential(
            nn.LayerNorm(d_model),
            nn.Linear(d_model, int(d_model * mlp_ratio)),
            nn.GELU(),
            nn.Linear(int(d_model * mlp_ratio), d_model),
            nn.Dropout(dropout),
        )
        self.nhead = nhead

    def project_kv(self, kv):
        """kv_norm + packed K/V in-projection of the key/value tokens: [B, L, D] -> [B, L, 2D]."""
        D = kv.shape[-1]
        a = self.attn
        return ops.linear(_ln(kv, self.kv_norm), a.in_proj_weight[D:], a.in_proj_bias[D:])

    def run(self, queries, kv):
        D = queries.shape[-1]
        a = self.attn
        q = ops.linear(_ln(queries, self.q_norm), a.in_proj_weight[:D], a.in_proj_bias[:D])
        kvp = self.project_kv(kv)
        o = ops.mha(q, kvp[..., :D], kvp[..., D:], self.nhead, 1.0 / math.sqrt(D // self.nhead))
        out = ops.linear(o, a.out_proj.weight, a.out_proj.bias, residual=queries)
        h = ops.linear(_ln(out, self.ffn[0]), self.ffn[1].weight, self.ffn[1].bias, act=ops.ACT_GELU)
        return ops.linear(h, self.ffn[3].weight, self.ffn[3].bias, residual=out)

    def forward(self, queries, kv):
        return self.run(queries, kv)


def _infer_hw_from_len(L):
    H = int(math.sqrt(L))
    if H * H != L:
        raise ValueError(f"Token length {L} is not a perfect square.")
    return H, H


class SegAwareGate(nn.Module):
    """utils_walkgpt.py:204-217."""

    def __init__(self, d_in, d_hidden=128):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(d_in), nn.Linear(d_in, d_hidden), nn.GELU(), nn.Linear(d_hidden, 1))

    def run(self, kv_tokens):
        h = ops.linear(_ln(kv_tokens, self.net[0]), self.net[1].weight, self.net[1].bias, act=ops.ACT_GELU)
        logit = ops.linear(h, self.net[3].weight, self.net[3].bias, out_f32=True)
        return ops.sigmoid_gate(kv_tokens, logit)

    def forward(self, kv_tokens):
        return self.run(kv_tokens)


class MultiScaleQFormerProjector(nn.Module):
    """utils_walkgpt.py:220-300."""

    def __init__(self, sam_dim, llama_dim, grid_size=None, num_heads=8, pad_to_square: bool = True,
                 target_square_side=None):
        super().__init__()
        self.grid_size = grid_size
        self.d_proj = 1024
        self.num_layers = 2
        self.pad_to_square = pad_to_square
        self.target_square_side = target_square_side
        self.pad_token = nn.Parameter(torch.zeros(1, 1, self.d_proj))
        nn.init.trunc_normal_(self.pad_token, std=0.02)
        self.sam_to_proj = nn.Linear(sam_dim, self.d_proj)
        self.q_x1 = nn.Parameter(torch.randn(1, 12, self.d_proj))
        self.q_x2 = nn.Parameter(torch.randn(1, 8, self.d_proj))
        self.q_x4 = nn.Parameter(torch.randn(1, 8, self.d_proj))
        self.q_global = nn.Parameter(torch.randn(1, 4, self.d_proj))
        self.cross_x1 = nn.ModuleList([CrossAttnBlock(self.d_proj, num_heads) for _ in range(self.num_layers)])
        self.cross_x2 = nn.ModuleList([CrossAttnBlock(self.d_proj, num_heads) for _ in range(self.num_layers)])
        self.cross_x4 = nn.ModuleList([CrossAttnBlock(self.d_proj, num_heads) for _ in range(self.num_layers)])
        self.cross_glb = nn.ModuleList([CrossAttnBlock(self.d_proj, num_heads) for _ in range(self.num_layers)])
        self.gate = SegAwareGate(self.d_proj)
        self.to_llama = nn.Linear(self.d_proj, llama_dim)
        for p in [self.q_x1, self.q_x2, self.q_x4, self.q_global]:
            nn.init.trunc_normal_(p, std=0.02)

    def forward(self, sam_feats, grid_size=None):
        """sam_feats [B, L, sam_dim] bf16 -> [B, s*s, llama_dim] (s*s = 36 with the reference's target side 6)."""
        _check_bf16_gpu(sam_feats, "sam_feats")
        B, L, _ = sam_feats.shape
        H, W = (grid_size or self.grid_size or _infer_hw_from_len(L))
        feats = ops.linear(sam_feats.contiguous(), self.sam_to_proj.weight, self.sam_to_proj.bias)  # [B, L, 1024]
        scales = [
            (self.q_x1, self.cross_x1, feats),
            (self.q_x2, self.cross_x2, ops.avgpool_tokens(feats, B, H, W, 2)),
            (self.q_x4, self.cross_x4, ops.avgpool_tokens(feats, B, H, W, 4)),
            (self.q_global, self.cross_glb, ops.mean_tokens(feats)),
        ]
        outs = []
        for q_param, layers, kv in scales:
            kv = self.gate.run(kv)  # the same gate module serves all four scales (:276)
            q = q_param.expand(B, -1, -1).contiguous()
            for blk in layers:
                q = blk.run(q, kv)
            outs.append(q)
        vis = torch.cat(outs, dim=1)
        if self.pad_to_square:
            Q = vis.shape[1]
            s = int(math.ceil(math.sqrt(Q))) if self.target_square_side is None else self.target_square_side
            assert s * s >= Q, "target_square_side too small"
            if s * s > Q:
                vis = torch.cat([vis, self.pad_token.expand(B, s * s - Q, -1)], dim=1)
        return ops.linear(vis.contiguous(), self.to_llama.weight, self.to_llama.bias)


class CalibratedTextProjector(nn.Module):
    """utils_walkgpt.py:302-327."""

    def __init__(self, in_dim: int, out_dim: int, widen: int = 2, use_residual: bool = False):
        super().__init__()
        mid = max(out_dim * widen, out_dim)
        self.net = nn.Sequential(nn.LayerNorm(in_dim), nn.Linear(in_dim, mid), nn.GELU(), nn.Linear(mid, out_dim),
                                 nn.LayerNorm(out_dim))
        self.use_residual = use_residual and (in_dim == out_dim)
        self.text_type = nn.Parameter(torch.zeros(1, 1, out_dim))
        self.log_temp = nn.Parameter(torch.zeros(1))
        nn.init.orthogonal_(self.net[3].weight, gain=0.5)
        if self.net[3].bias is not None:
            nn.init.zeros_(self.net[3].bias)

    def _tiled(self, weight):
        """ops.tile_weight(weight), cached per (storage, version)"""
        cache = self.__dict__.setdefault("_tiled_cache", {})
        key = (weight.data_ptr(), weight._version, weight.dtype, str(weight.device))
        ent = cache.get(id(weight))
        if ent is None or ent[0] != key:
            with torch.no_grad():
                ent = cache[id(weight)] = (key, ops.tile_weight(weight.detach().contiguous()))
        return ent[1]

    def pre_tail(self, x):
        """net[0..3] of :321-323: LayerNorm -> Linear -> GELU -> Linear, [.., in_dim] -> [.., out_dim] (before net[4], text_type, normalise)."""
        _check_bf16_gpu(x, "hidden states")
        if self.use_residual:
            raise NotImplementedError("use_residual=True is never configured by WalkGPT (walkgpt.py:115-123)")
        ln = self.net[0]
        x = x.contiguous()
        few = x.numel() // x.shape[-1] <= 128         # up to 128 [SEG] rows: one-launch skinny GEMMs on weights kept in fragment order
        t1, t3 = (self._tiled(self.net[1].weight), self._tiled(self.net[3].weight)) if few else (None, None)
        y = ops.layernorm_linear(x, ln.weight, ln.bias, ln.eps, self.net[1].weight, self.net[1].bias, act=ops.ACT_GELU, weight_tiled=t1)
        return ops.layernorm_linear(y, None, None, 0.0, self.net[3].weight, self.net[3].bias, weight_tiled=t3)

    def tail_operands(self):
        """(gamma, beta, text_type, log_temp, eps) of the tail (:324-327): ops.ctp_tail's operands, also taken by the mask decoder's first
        token launch, which applies the tail itself on the inference path (WalkGPTGrounding.decode_from_hidden)."""
        return (self.net[4].weight, self.net[4].bias, self.text_type.reshape(-1), self.log_temp, self.net[4].eps)

    def forward(self, x):
        return ops.ctp_tail(self.pre_tail(x), *self.tail_operands())


class TinyCrossAttn(nn.Module, _Prepared):
    """utils_walkgpt.py:330-357: single-head cross attention, Q = one [SEG] embedding, K/V = its row's SAM tokens."""

    def __init__(self, d=256, bias=False):
        super().__init__()
        self.wq = nn.Linear(d, d, bias=bias)
        self.wk = nn.Linear(d, d, bias=bias)
        self.wv = nn.Linear(d, d, bias=bias)
        self.out = nn.Linear(d, d, bias=bias)
        self.dropout = nn.Dropout(p=0.0)

    def _build(self):
        return {"wk_t": self.wk.weight.t().contiguous()}   # q.(Wk kv + bk) = (Wk^T q).kv + const; the constant cancels in the softmax

    def project_pooled(self, pooled):
        """out(wv(.)) applied to the attention-pooled raw token (fp32 [M, d]): sum_n a_n (Wv kv_n + bv) = Wv pooled + bv."""
        ctx = ops.linear(pooled.to(BF16), self.wv.weight, self.wv.bias)
        return ops.linear(ctx, self.out.weight, self.out.bias)

    def forward(self, q_vec, kv):
        """q_vec [M, d], kv [M, N, d] (bf16) -> (v_pos [M, d] bf16, attn [M, N] fp32)."""
        _check_bf16_gpu(q_vec, "q_vec")
        M = q_vec.shape[0]
        r = ops.nce_forward(q_vec, kv.contiguous(), torch.arange(M, device=q_vec.device), self.wq.weight, self.wq.bias,
                            self._prep_get(self._build)["wk_t"], 1.0, None, False)
        return self.project_pooled(r["vraw"]), r["attn_w"]


def infonce_loss(pred_embeddings, sam_tokens_256, seg_row_ids, tiny_xattn, *, temperature=0.07, top_k=None,
                 exclude_same_row=True, normalize=True, return_aux=False):
    """utils_walkgpt.py:8-73 (forward).  pred_embeddings [M, 256] bf16, sam_tokens_256 [rows, N, 256] bf16, seg_row_ids [M]."""
    if not normalize:
        raise NotImplementedError("WalkGPT always calls infonce_loss(normalize=True) (model/walkgpt.py:463-472)")
    _check_bf16_gpu(pred_embeddings, "pred_embeddings")
    assert sam_tokens_256.shape[-1] == pred_embeddings.shape[-1], "Vision/text feature dims must match for InfoNCE."
    r = ops.nce_forward(pred_embeddings, sam_tokens_256.contiguous(), seg_row_ids, tiny_xattn.wq.weight, tiny_xattn.wq.bias,
                        tiny_xattn._prep_get(tiny_xattn._build)["wk_t"], temperature, top_k, exclude_same_row, want_logits=return_aux)
    v_pos = r["vraw"] if r["refined"] else tiny_xattn.project_pooled(r["vraw"]).float().contiguous()
    loss, _, logits = r["finish"](v_pos)
    if return_aux:
        labels = torch.zeros(pred_embeddings.shape[0], dtype=torch.long, device=pred_embeddings.device)
        return loss, {"v_pos": v_pos, "attn_w": r["attn_w"], "logits": logits, "labels": labels}
    return loss
