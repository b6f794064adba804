"""MI355X-native SAM modules behind the reference's own module surface.

Same class names, constructor arguments, attribute tree and state_dict keys as
/root/reference/model/segment_anything/modeling/{image_encoder,prompt_encoder,transformer,mask_decoder,sam}.py, so a
reference checkpoint loads with strict=True and callers (`model/walkgpt.py:241-258, 713-737`) are unchanged.  The
nn.Linear / nn.Conv2d / nn.LayerNorm children are PARAMETER CONTAINERS ONLY: their forward is never called.  All
arithmetic goes through walkgpt_amd.ops (the C-ABI of libwalkgpt_hip.so); tensors must be bf16 on the GPU and there is
no CPU fallback.

Internal layout is channels-last token rows [B*h*w, C]; NCHW appears only at the module boundary where the
reference API returns / accepts it.
"""
import math
from functools import partial
from typing import Optional, Tuple

import os

import torch
import torch.nn as nn

from .. import ops

BF16 = torch.bfloat16


class _Prepared:
    """Derived (re-laid / concatenated / bf16) kernel operands, rebuilt after any parameter change."""

    def _prep_get(self, build, deps=None, slot=""):
        """deps: the tensors `build` reads (default: every parameter below this module -- pass them explicitly where the module
        tree is large, the key is recomputed on every forward).  slot: a second, independently keyed set on the same module."""
        if deps is None:
            deps = self.parameters(recurse=True)
        key = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in deps if p is not None)
        if getattr(self, "_prep_key" + slot, None) != key:
            object.__setattr__(self, "_prep_val" + slot, build())
            object.__setattr__(self, "_prep_key" + slot, key)
        return getattr(self, "_prep_val" + slot)


def _check_bf16_gpu(t, what):
    if not t.is_cuda or t.dtype != BF16:
        raise RuntimeError("%s must be a bf16 GPU tensor for the walkgpt_amd HIP path (got %s on %s); "
                           "cast the model and inputs with .bfloat16().cuda() -- there is no CPU fallback"
                           % (what, t.dtype, t.device))


class LayerNorm2d(nn.Module):
    """common.py:31-43 (parameter container; applied as a row LayerNorm on channels-last rows)."""

    def __init__(self, num_channels: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps


class MLPBlock(nn.Module):
    """common.py:13-26."""

    def __init__(self, embedding_dim: int, mlp_dim: int, act=nn.GELU) -> None:
        super().__init__()
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)
        self.act = act()
        self._act_code = ops.ACT_RELU if isinstance(self.act, nn.ReLU) else ops.ACT_GELU

    def rows(self, x, residual=None):
        h = ops.linear(x, self.lin1.weight, self.lin1.bias, act=self._act_code)
        return ops.linear(h, self.lin2.weight, self.lin2.bias, residual=residual)

    def forward(self, x):
        return self.rows(x)


# ------------------------------------------------------------------------------------------------------------------
# image encoder
# ------------------------------------------------------------------------------------------------------------------
class PatchEmbed(nn.Module):
    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=768) -> None:
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)


class Attention(nn.Module):
    """image_encoder.py:198-260 (container for qkv / proj / rel_pos_{h,w})."""

    def __init__(self, dim, num_heads=8, qkv_bias=True, use_rel_pos=False, rel_pos_zero_init=True, input_size=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        if self.use_rel_pos:
            assert input_size is not None, "Input size must be provided if using relative positional encoding."
            self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
            self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_dim))


class Block(nn.Module, _Prepared):
    """image_encoder.py:130-193.  norm1 / norm2 are folded into the qkv / lin1 GEMMs behind them (ops.ln_linear)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm, act_layer=nn.GELU,
                 use_rel_pos=False, rel_pos_zero_init=True, window_size=0, input_size=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, use_rel_pos=use_rel_pos,
                              rel_pos_zero_init=rel_pos_zero_init,
                              input_size=input_size if window_size == 0 else (window_size, window_size))
        self.norm2 = norm_layer(dim)
        self.mlp = MLPBlock(embedding_dim=dim, mlp_dim=int(dim * mlp_ratio), act=act_layer)
        self.window_size = window_size

    gemm_dtype = "bf16"   # "fp8": qkv / proj / lin1 / lin2 on e4m3 operands (BASELINE config C5); set through WalkGPTGrounding.set_gemm_dtype

    def _build_fp8(self):
        a, m = self.attn, self.mlp
        return {k: ops.quantize_weight_fp8(w) for k, w in (("qkv", a.qkv.weight), ("proj", a.proj.weight), ("lin1", m.lin1.weight),
                                                           ("lin2", m.lin2.weight))}

    def _build_mx(self):
        a, m = self.attn, self.mlp
        return {"qkv": ops.fold_layernorm_mx(self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias), "proj": ops.mx_weight(a.proj.weight),
                "lin1": ops.fold_layernorm_mx(self.norm2.weight, self.norm2.bias, m.lin1.weight, m.lin1.bias), "lin2": ops.mx_weight(m.lin2.weight)}

    # the derived operands of the three GEMM paths, each keyed on the parameters it reads
    def _prep_bf16(self):
        a, m = self.attn, self.mlp
        return self._prep_get(self._build, (self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, self.norm2.weight, self.norm2.bias,
                                            m.lin1.weight, m.lin1.bias, m.lin2.weight))

    def _prep_mx(self):
        a, m = self.attn, self.mlp
        return self._prep_get(self._build_mx, (self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, self.norm2.weight,
                                               self.norm2.bias, m.lin1.weight, m.lin1.bias, m.lin2.weight), slot="_mx")

    def _prep_fp8(self):
        a, m = self.attn, self.mlp
        return self._prep_get(self._build_fp8, (a.qkv.weight, a.proj.weight, m.lin1.weight, m.lin2.weight), slot="_fp8")

    def prepare(self):
        """Build (or refresh) the derived operands rows() will read, on the CURRENT stream: a caller that runs slices of a batch on side streams
        calls this before it forks, so that no slice builds them lazily on one stream while another reads them unordered."""
        if self.gemm_dtype != "fp8":
            return self._prep_bf16()
        m = self.mlp
        return self._prep_mx() if ops.mx_chain_ok(self.norm1.weight.shape[0], m.lin1.weight.shape[0]) else self._prep_fp8()

    def rows_mx(self, x, B, grid):
        """The block as one MX chain on the persistent fp8 GEMM (ops.linear_mxfp8): e4m3 operands with a power-of-two scale per 32
        values on both sides; both LayerNorms folded into the GEMM behind them from the partial sums the GEMM in front left; the
        residual GEMMs (proj, lin2) leave the stream as bf16 AND as the next GEMM's e4m3 operand; the MLP's hidden layer only ever
        exists as e4m3.  At head_dim 64 the attention kernels write their output as the proj GEMM's MX operand themselves (round 4); at ViT-H's 80 a
        32-column block would straddle two heads and the attention output keeps one quantisation pass per block.  Attention, the residual stream and every
        statistic stay bf16 / fp32."""
        a, m = self.attn, self.mlp
        w = self._prep_mx()
        qkv = ops.linear_mxfp8(x, w["qkv"], ln_eps=self.norm1.eps)
        window = self.window_size if self.window_size > 0 else grid
        o = ops.sam_attention(qkv, a.qkv.bias, a.rel_pos_h, a.rel_pos_w, B, grid, window, a.num_heads, mx_out=True)
        if not isinstance(o, tuple):       # (head_dim 80, other windows: the attention kernel writes bf16 and the quantisation is a pass of its own)
            o = ops.quantize_mx_fp8(o)
        x = ops.linear_mxfp8(o, w["proj"], bias=a.proj.bias, residual=x, mx_out=True, row_partials=True)
        h = ops.linear_mxfp8(x, w["lin1"], act=m._act_code, ln_eps=self.norm2.eps, mx_out=True, bf16_out=False)
        return ops.linear_mxfp8(h, w["lin2"], bias=m.lin2.bias, residual=x, mx_out=True, row_partials=True)

    def rows_fp8(self, x, B, grid):
        """The block on the fp8 GEMM path.  Widths the persistent MX kernel takes run as rows_mx; the rest
        with per-row activation scales: both LayerNorms fused with the per-row quantisation of their output."""
        a, m = self.attn, self.mlp
        if ops.mx_chain_ok(x.shape[-1], m.lin1.weight.shape[0]) and ops.mx_prepare_rows(x):
            return self.rows_mx(x, B, grid)
        w = self._prep_fp8()
        q, s = ops.quantize_rows_fp8(x, ln=(self.norm1.weight, self.norm1.bias), eps=self.norm1.eps)
        qkv = ops.linear_fp8(q, s, *w["qkv"], bias=a.qkv.bias)
        window = self.window_size if self.window_size > 0 else grid
        o = ops.sam_attention(qkv, a.qkv.bias, a.rel_pos_h, a.rel_pos_w, B, grid, window, a.num_heads)
        q, s = ops.quantize_rows_fp8(o)
        x = ops.linear_fp8(q, s, *w["proj"], bias=a.proj.bias, residual=x)
        q, s = ops.quantize_rows_fp8(x, ln=(self.norm2.weight, self.norm2.bias), eps=self.norm2.eps)
        q, s = ops.quantize_rows_fp8(ops.linear_fp8(q, s, *w["lin1"], bias=m.lin1.bias, act=m._act_code))
        return ops.linear_fp8(q, s, *w["lin2"], bias=m.lin2.bias, residual=x)

    def rows(self, x, B, grid):
        """x [B*grid*grid, D] -> same.  window partition / unpartition live inside the attention kernel."""
        a = self.attn
        if not a.use_rel_pos or a.qkv.bias is None:
            raise NotImplementedError("the HIP SAM attention is built for use_rel_pos=True, qkv_bias=True (build_sam.py:56-108)")
        if self.gemm_dtype == "fp8":
            return self.rows_fp8(x, B, grid)
        p = self._prep_bf16()
        qkv = ops.ln_linear(x, p["qkv"], self.norm1.eps)
        window = self.window_size if self.window_size > 0 else grid
        o = ops.sam_attention(qkv, a.qkv.bias, a.rel_pos_h, a.rel_pos_w, B, grid, window, a.num_heads)
        # proj and lin2 write the residual stream that norm2 / the next block's norm1 read: they leave the rows' partial sums next
        # to them and ln_linear picks those up (no statistics pass in between)
        x = ops.linear(o, a.proj.weight, a.proj.bias, residual=x, row_partials=True)
        h = ops.ln_linear(x, p["lin1"], self.norm2.eps, act=self.mlp._act_code)
        return ops.linear(h, p["lin2_w"], self.mlp.lin2.bias, residual=x, row_partials=True)

    def _build(self):
        a, m = self.attn, self.mlp
        w2 = m.lin2.weight
        if (w2.shape[1] * 2) % 10240 == 0 and w2.is_cuda and os.environ.get("WG_LIN2_PAD", "1") != "0":      # (=0: A/B runs)
            # ViT-H's lin2 (K = 5120): weight rows on a 10 240-byte pitch camp on a few memory channels -- the same rows 128 bytes further apart run the
            # GEMM 10 % faster (tools/bench_gemm_pitch.py, notes/r06_experiments.md section 7; K = 3072 / 4096 rows measured no difference)
            buf = torch.empty(w2.shape[0], w2.shape[1] + 64, device=w2.device, dtype=w2.dtype)
            buf[:, :w2.shape[1]].copy_(w2.detach())
            w2 = buf[:, :w2.shape[1]]
        return {"qkv": ops.fold_layernorm(self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias),
                "lin1": ops.fold_layernorm(self.norm2.weight, self.norm2.bias, m.lin1.weight, m.lin1.bias), "lin2_w": w2}


class ImageEncoderViT(nn.Module, _Prepared):
    """image_encoder.py:17-125."""

    def __init__(self, img_size=1024, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 out_chans=256, qkv_bias=True, norm_layer=nn.LayerNorm, act_layer=nn.GELU, use_abs_pos=True,
                 use_rel_pos=False, rel_pos_zero_init=True, window_size=0, global_attn_indexes=()):
        super().__init__()
        self.img_size = img_size
        self.embed_dim = embed_dim
        self.out_chans = out_chans
        self.patch_size = patch_size
        self.patch_embed = PatchEmbed(kernel_size=(patch_size, patch_size), stride=(patch_size, patch_size),
                                      in_chans=in_chans, embed_dim=embed_dim)
        self.pos_embed: Optional[nn.Parameter] = None
        if use_abs_pos:
            self.pos_embed = nn.Parameter(torch.zeros(1, img_size // patch_size, img_size // patch_size, embed_dim))
        self.blocks = nn.ModuleList()
        for i in range(depth):
            self.blocks.append(Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                     norm_layer=norm_layer, act_layer=act_layer, use_rel_pos=use_rel_pos,
                                     rel_pos_zero_init=rel_pos_zero_init,
                                     window_size=window_size if i not in global_attn_indexes else 0,
                                     input_size=(img_size // patch_size, img_size // patch_size)))
        self.neck = nn.Sequential(
            nn.Conv2d(embed_dim, out_chans, kernel_size=1, bias=False),
            LayerNorm2d(out_chans),
            nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False),
            LayerNorm2d(out_chans),
        )

    def _build_prepared(self):
        D = self.embed_dim
        return {
            "patch_w": self.patch_embed.proj.weight.reshape(D, -1),
            "neck0_w": self.neck[0].weight.reshape(self.out_chans, D),
            # [O, C, 3, 3] -> [O, (ky, kx, c)] to match wg_im2row3x3's column order
            "neck2_w": self.neck[2].weight.permute(0, 2, 3, 1).reshape(self.out_chans, -1).contiguous(),
            "pos": None if self.pos_embed is None else self.pos_embed.reshape(-1, D),
        }

    def _prep_own(self):
        return self._prep_get(self._build_prepared, (self.patch_embed.proj.weight, self.neck[0].weight, self.neck[2].weight, self.pos_embed))

    def prepare(self):
        """Every derived operand of forward_tokens, built on the current stream (see Block.prepare)."""
        self._prep_own()
        for blk in self.blocks:
            blk.prepare()

    def forward_tokens(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B,3,S,S] bf16 -> channels-last embedding rows [B*g*g, out_chans] (written to `out` when given)."""
        _check_bf16_gpu(x, "images")
        _check_bf16_gpu(self.patch_embed.proj.weight, "image encoder weights")
        B = x.shape[0]
        g = x.shape[-1] // self.patch_size
        if x.shape[-2] != x.shape[-1] or g != self.img_size // self.patch_size:
            raise RuntimeError("image encoder expects %dx%d inputs" % (self.img_size, self.img_size))
        p = self._prep_own()
        rows = ops.patchify(x.contiguous(), self.patch_size)
        t = ops.linear(rows, p["patch_w"], self.patch_embed.proj.bias, residual=p["pos"], res_row_mod=g * g if p["pos"] is not None else 0,
                       row_partials=True)      # block 0's norm1 reads these rows: the GEMM leaves their statistics with them
        for blk in self.blocks:
            t = blk.rows(t, B, g)
        t = ops.linear(t, p["neck0_w"])
        t = ops.layernorm(t, self.neck[1].weight, self.neck[1].bias, self.neck[1].eps)
        t = ops.linear(ops.im2row3x3(t, B, g, g), p["neck2_w"])
        return ops.layernorm(t, self.neck[3].weight, self.neck[3].bias, self.neck[3].eps, out=out)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        t = self.forward_tokens(x)
        B = x.shape[0]
        g = self.img_size // self.patch_size
        return ops.tokens_to_nchw(t, B, g * g, self.out_chans).view(B, self.out_chans, g, g)


# ------------------------------------------------------------------------------------------------------------------
# prompt encoder (text branch)
# ------------------------------------------------------------------------------------------------------------------
class PositionEmbeddingRandom(nn.Module):
    def __init__(self, num_pos_feats: int = 64, scale: Optional[float] = None) -> None:
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer("positional_encoding_gaussian_matrix", scale * torch.randn((2, num_pos_feats)))

    def tokens(self, size: Tuple[int, int]) -> torch.Tensor:
        """[h*w, 2F] fp32 rows (prompt_encoder.py:203-229), computed by wg_dense_pe_f32."""
        return ops.dense_pe_tokens(self.positional_encoding_gaussian_matrix, size[0], size[1])

    def forward(self, size: Tuple[int, int]) -> torch.Tensor:
        h, w = size
        return self.tokens(size).t().reshape(-1, h, w)


class PromptEncoder(nn.Module):
    """prompt_encoder.py:16-186.  Only the text branch (`text_embeds`) is on WalkGPT's path; point / box / mask prompts
    keep their parameters (checkpoint parity) but are out of scope."""

    def __init__(self, embed_dim, image_embedding_size, input_image_size, mask_in_chans, activation=nn.GELU):
        super().__init__()
        self.embed_dim = embed_dim
        self.input_image_size = input_image_size
        self.image_embedding_size = image_embedding_size
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.num_point_embeddings = 4
        self.point_embeddings = nn.ModuleList([nn.Embedding(1, embed_dim) for _ in range(self.num_point_embeddings)])
        self.not_a_point_embed = nn.Embedding(1, embed_dim)
        self.mask_input_size = (4 * image_embedding_size[0], 4 * image_embedding_size[1])
        self.mask_downscaling = nn.Sequential(
            nn.Conv2d(1, mask_in_chans // 4, kernel_size=2, stride=2),
            LayerNorm2d(mask_in_chans // 4),
            activation(),
            nn.Conv2d(mask_in_chans // 4, mask_in_chans, kernel_size=2, stride=2),
            LayerNorm2d(mask_in_chans),
            activation(),
            nn.Conv2d(mask_in_chans, embed_dim, kernel_size=1),
        )
        self.no_mask_embed = nn.Embedding(1, embed_dim)
        self._pe_cache = None

    def dense_pe_tokens(self) -> torch.Tensor:
        """bf16 [h*w, C] rows of get_dense_pe(); input independent, cached per (device, buffer version)."""
        G = self.pe_layer.positional_encoding_gaussian_matrix
        key = (G.data_ptr(), G._version, G.device)
        if self._pe_cache is None or self._pe_cache[0] != key:
            pe32 = self.pe_layer.tokens(self.image_embedding_size)
            self._pe_cache = (key, pe32, ops.cast_bf16(pe32))
        return self._pe_cache[2]

    def get_dense_pe(self) -> torch.Tensor:
        self.dense_pe_tokens()
        h, w = self.image_embedding_size
        return self._pe_cache[1].t().reshape(1, -1, h, w)

    def forward(self, points, boxes, masks, text_embeds):
        if points is not None or boxes is not None or masks is not None:
            raise NotImplementedError("walkgpt_amd: only the text-embedding prompt branch is on the WalkGPT hot path")
        if text_embeds is None:
            raise ValueError("text_embeds is required")
        bs = text_embeds.shape[0]
        sparse = text_embeds  # prompt_encoder.py:175-176: concatenation with an empty tensor
        dense = self.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(bs, -1, self.image_embedding_size[0], self.image_embedding_size[1])
        return sparse, dense


# ------------------------------------------------------------------------------------------------------------------
# two-way transformer + mask decoder
# ------------------------------------------------------------------------------------------------------------------
class DecoderAttention(nn.Module):
    """transformer.py:185-242 (named `Attention` there)."""

    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1) -> None:
        super().__init__()
        self.embedding_dim = embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        assert self.internal_dim % num_heads == 0, "num_heads must divide embedding_dim."
        self.q_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.k_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.v_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.out_proj = nn.Linear(self.internal_dim, embedding_dim)

    def run(self, q, k, v, P, residual=None, res_row_mod=0):
        """q [Pq, Nq, C], k/v [Pk, Nk, C] with Pq, Pk in {1, P}: a leading 1 means "shared by all P prompts" and is
        projected once, then broadcast with a zero batch stride.  Returns [P, Nq, C] (+ residual)."""
        qp = ops.linear(q, self.q_proj.weight, self.q_proj.bias)
        kp = ops.linear(k, self.k_proj.weight, self.k_proj.bias)
        vp = ops.linear(v, self.v_proj.weight, self.v_proj.bias)
        return self.run_projected(qp, kp, vp, P, residual, res_row_mod)

    def run_projected(self, qp, kp, vp, P, residual=None, res_row_mod=0):
        """Attention + out_proj on already projected q/k/v (possibly column slices of a fused projection buffer)."""
        ex = lambda t: t if t.shape[0] == P else t.expand(P, -1, -1)
        c = self.internal_dim // self.num_heads
        o = ops.mha(ex(qp), ex(kp), ex(vp), self.num_heads, 1.0 / math.sqrt(c))
        return ops.linear(o, self.out_proj.weight, self.out_proj.bias, residual=residual, res_row_mod=res_row_mod)


def _pe_folded_projection(pe_rows, parts, src_bias=None):
    """Fused image-side projection operands.  (x + pe) W^T + b == x W^T + (pe W^T + b), and pe is input independent, so
    every projection that reads the image tokens `x` of one layer becomes a column block of ONE GEMM `x @ Wcat^T + R[row %
    hw]`.  parts: [(Linear, adds_pe)]; src_bias [1, C]: a constant row the caller would otherwise add to every image token first
    (the dense no-mask embedding) -- it joins the table the same way.  Returns (Wcat [sum N, C], R [hw, sum N]) in bf16."""
    ws, rs = [], []
    hw = pe_rows.shape[0]
    with_pe = pe_rows if src_bias is None else (pe_rows.float() + src_bias.float()).to(pe_rows.dtype)
    for lin, adds_pe in parts:
        ws.append(lin.weight)
        if adds_pe:
            rs.append(ops.linear(with_pe, lin.weight, lin.bias))
        elif src_bias is not None:
            rs.append(ops.linear(src_bias.reshape(1, -1).to(pe_rows.dtype).contiguous(), lin.weight, lin.bias).expand(hw, -1))
        else:
            rs.append(lin.bias.unsqueeze(0).expand(hw, -1))
    return torch.cat(ws, 0).contiguous(), torch.cat(rs, 1).contiguous()


def _interleave_kv_heads(wcat, rtab, d, heads):
    """Reorder the first 2d output columns of a fused projection ([K (d) | V (d) | ...]) to [K_h | V_h] per head, so the token->image
    attention kernel finds everything it needs of (key, head) in one 64-byte piece."""
    c = d // heads
    old = torch.arange(2 * d, device=wcat.device).reshape(2, heads, c).permute(1, 0, 2).reshape(-1)     # new column -> old column
    idx = torch.cat([old, torch.arange(2 * d, wcat.shape[0], device=wcat.device)])
    return wcat[idx].contiguous(), rtab[:, idx].contiguous()


import weakref

_TILED = {}   # id(weight Parameter) -> (weak reference, identity key, fragment-order copy); tensors compare element-wise, so no WeakKeyDictionary


def _tiled(weight, k_slices=1):
    """ops.tile_weight(weight), cached until the parameter's storage, version, dtype or device changes (the token-side decoder kernels
    stream their weight matrices in MFMA fragment order; the copy is made once per checkpoint)."""
    key = (weight.data_ptr(), weight._version, weight.dtype, str(weight.device), k_slices)
    wid = id(weight)
    ent = _TILED.get(wid)
    if ent is None or ent[0]() is not weight or ent[1] != key:
        with torch.no_grad():
            tiled = ops.tile_weight(weight.detach().contiguous(), k_slices)
        ent = (weakref.ref(weight, lambda _r, wid=wid: _TILED.pop(wid, None)), key, tiled)
        _TILED[wid] = ent
    return ent[2]


def _lin_pair(lin):
    """[weight in fragment order, bias] of a Linear on the token side of the mask decoder"""
    return [_tiled(lin.weight), lin.bias]


class TwoWayAttentionBlock(nn.Module):
    def __init__(self, embedding_dim, num_heads, mlp_dim=2048, activation=nn.ReLU, attention_downsample_rate=2,
                 skip_first_layer_pe=False):
        super().__init__()
        self.self_attn = DecoderAttention(embedding_dim, num_heads)
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.cross_attn_token_to_image = DecoderAttention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.mlp = MLPBlock(embedding_dim, mlp_dim, activation)
        self.norm3 = nn.LayerNorm(embedding_dim)
        self.norm4 = nn.LayerNorm(embedding_dim)
        self.cross_attn_image_to_token = DecoderAttention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe

    def _image_side(self, key_pe, src_bias=None):
        """[(K_h | V_h of token->image, per head) | Q_i2t] of the image tokens as one GEMM (cached per positional-encoding tensor / weights /
        constant row)."""
        t2i, i2t = self.cross_attn_token_to_image, self.cross_attn_image_to_token
        key = (key_pe.data_ptr(), key_pe._version) + tuple((p.data_ptr(), p._version) for p in
                                                           (t2i.k_proj.weight, t2i.v_proj.weight, i2t.q_proj.weight,
                                                            t2i.k_proj.bias, t2i.v_proj.bias, i2t.q_proj.bias))
        if src_bias is not None:
            key += (src_bias.data_ptr(), src_bias._version)
        if getattr(self, "_img_key", None) != key:
            self._img_val = _interleave_kv_heads(*_pe_folded_projection(key_pe.reshape(-1, key_pe.shape[-1]),
                                                                        [(t2i.k_proj, True), (t2i.v_proj, False), (i2t.q_proj, True)], src_bias),
                                                 t2i.internal_dim, t2i.num_heads)
            self._img_key = key
        return self._img_val

    def image_side(self, keys, key_pe, src_bias=None):
        """[K_t2i | V_t2i | Q_i2t] of this block's image tokens: the three projections that read `keys` (the reference recomputes
        keys + key_pe for two of them) as one fused GEMM with the positional term folded into an additive table.  src_bias [1, C]:
        the block's image tokens are keys + src_bias (mask_decoder.py:136: the dense no-mask embedding), never materialised."""
        wcat, rtab = self._image_side(key_pe, src_bias)
        return ops.linear(keys, wcat, residual=rtab, res_row_mod=keys.shape[1])          # [1|P, hw, 3d]

    def fused_i2t_ok(self, keys, n_tokens):
        i2t = self.cross_attn_image_to_token
        return i2t.internal_dim == 128 and i2t.num_heads == 8 and keys.shape[-1] == 256 and n_tokens == 6 and keys.shape[1] % 16 == 0

    def image_to_token(self, proj, keys, kq, vq, P, src_bias=None, prompt_image=None):
        """transformer.py:173-180: image->token attention on the projected operands, out_proj + residual, norm4 -- one launch for SAM's
        geometry (ops.dec_i2t_rows), three otherwise."""
        i2t = self.cross_attn_image_to_token
        hw = keys.shape[1]
        d = i2t.internal_dim
        if self.fused_i2t_ok(keys, kq.shape[1]):
            return ops.dec_i2t_rows(proj[..., 2 * d:], kq, vq, i2t.out_proj.weight, i2t.out_proj.bias, keys, self.norm4.weight,
                                    self.norm4.bias, self.norm4.eps, P, res_bias=src_bias, prompt_image=prompt_image)
        if prompt_image is not None:           # (general geometries: one copy of the image tokens per prompt, as the reference makes)
            keys, proj = keys.index_select(0, prompt_image.long()), proj.index_select(0, prompt_image.long())
        if src_bias is not None:
            keys = ops.add_rows(keys, src_bias)
        keys = i2t.run_projected(proj[..., 2 * d:], kq, vq, P, residual=keys, res_row_mod=hw if keys.shape[0] == 1 and P > 1 else 0)
        return ops.layernorm(keys, self.norm4.weight, self.norm4.bias, self.norm4.eps)

    def mlp_partials(self, queries, combine=None):
        return ops.dec_mlp_partial(queries, _tiled(self.mlp.lin1.weight), self.mlp.lin1.bias, _tiled(self.mlp.lin2.weight, ops.dec_mlp_slices()), combine=combine,
                                   eps=self.norm2.eps)

    def check_fused(self):
        if self.norm1.eps != self.norm2.eps or self.norm1.eps != self.norm3.eps or self.mlp._act_code != ops.ACT_RELU:
            raise NotImplementedError("the fused token kernels are built for SAM's decoder block (one LayerNorm eps, ReLU MLP)")

    def fused_ok(self, embedding_dim, eps, d):
        """The fused token kernels take SAM's block: 256 channels, 8 heads, MLP 2048 with ReLU, one LayerNorm eps, one attention width."""
        return (embedding_dim == 256 and self.self_attn.num_heads == 8 and self.mlp.lin1.out_features == 2048 and self.mlp._act_code == ops.ACT_RELU
                and self.norm1.eps == self.norm2.eps == self.norm3.eps == eps and self.cross_attn_token_to_image.internal_dim == d
                and self.cross_attn_image_to_token.internal_dim == d)

    def run_general(self, queries, keys, query_pe, key_pe, P):
        """transformer.py:151-182 op by op (any width, head count, token count; bf16 token stream): the path for geometries the fused
        token kernels do not take.  queries / query_pe [P, N, C] bf16; keys [1|P, hw, C]; key_pe [1, hw, C]."""
        ln = lambda x, n: ops.layernorm(x, n.weight, n.bias, n.eps)   # noqa: E731
        hw = keys.shape[1]
        t2i, i2t = self.cross_attn_token_to_image, self.cross_attn_image_to_token
        d = t2i.internal_dim
        if self.skip_first_layer_pe:
            queries = self.self_attn.run(queries, queries, queries, P)
        else:
            q = ops.add_rows(queries, query_pe)
            queries = self.self_attn.run(q, q, queries, P, residual=queries)
        queries = ln(queries, self.norm1)
        wcat, rtab = _pe_folded_projection(key_pe.reshape(-1, key_pe.shape[-1]), [(t2i.k_proj, True), (t2i.v_proj, False), (i2t.q_proj, True)])
        proj = ops.linear(keys, wcat, residual=rtab, res_row_mod=hw)          # [1|P, hw, 2d + d_i2t]
        q = ops.add_rows(queries, query_pe)
        qp = ops.linear(q, t2i.q_proj.weight, t2i.q_proj.bias)
        queries = t2i.run_projected(qp, proj[..., :d], proj[..., d:2 * d], P, residual=queries)
        queries = ln(queries, self.norm2)
        queries = self.mlp.rows(queries, residual=queries)
        queries = ln(queries, self.norm3)
        q = ops.add_rows(queries, query_pe)
        kq = ops.linear(q, i2t.k_proj.weight, i2t.k_proj.bias)
        vq = ops.linear(queries, i2t.v_proj.weight, i2t.v_proj.bias)
        keys = i2t.run_projected(proj[..., 2 * d:], kq, vq, P, residual=keys, res_row_mod=hw if keys.shape[0] == 1 and P > 1 else 0)
        return queries, ln(keys, self.norm4)

    def run(self, queries, keys, query_pe, key_pe, P):
        """transformer.py:151-182, one block on its own (TwoWayTransformer.run_tokens merges the launch that closes a block with the one
        that opens the next).  queries / query_pe [P,6,C] fp32 (queries updated in place); keys [1|P, hw, C] bf16; key_pe [1, hw, C]."""
        self.check_fused()
        t2i = self.cross_attn_token_to_image
        d = t2i.internal_dim
        table = token_stage_table(self_blk=self, t2i=t2i, norm=self.norm2, sum_blk=self)
        q = torch.empty(P, queries.shape[1], d, device=keys.device, dtype=torch.float32)
        kq = torch.empty(P, queries.shape[1], d, device=keys.device, dtype=BF16)
        vq = torch.empty_like(kq)
        ops.dec_tokens(ops.TOK_SELF | ops.TOK_Q_T2I, self.skip_first_layer_pe, queries, query_pe, table, q_t2i=q, eps=self.norm1.eps)
        proj = self.image_side(keys, key_pe)
        part = ops.dec_attn_partial(q, proj[..., :2 * d])
        ops.dec_tokens(ops.TOK_COMBINE, False, queries, query_pe, table, attn_partials=part, eps=self.norm1.eps)
        ops.dec_tokens(ops.TOK_SUM_MLP, False, queries, query_pe, table, mlp_partials=self.mlp_partials(queries), k_i2t=kq, v_i2t=vq,
                       eps=self.norm1.eps)
        return queries, self.image_to_token(proj, keys, kq, vq, P)


def token_stage_table(self_blk=None, t2i=None, norm=None, sum_blk=None):
    """The 24-slot pointer table of wg_dec_tokens_f32 (include/walkgpt_hip.h): SELF from `self_blk`, Q_T2I / COMBINE from the attention
    module `t2i` and the LayerNorm `norm` that follows it, SUM_MLP from `sum_blk` (possibly the block BEFORE self_blk)."""
    w = [None] * 24
    if self_blk is not None:
        sa = self_blk.self_attn
        w[0:8] = _lin_pair(sa.q_proj) + _lin_pair(sa.k_proj) + _lin_pair(sa.v_proj) + _lin_pair(sa.out_proj)
        w[8:10] = [self_blk.norm1.weight, self_blk.norm1.bias]
    if t2i is not None:
        w[10:14] = _lin_pair(t2i.q_proj) + _lin_pair(t2i.out_proj)
        w[14:16] = [norm.weight, norm.bias]
    if sum_blk is not None:
        i2t = sum_blk.cross_attn_image_to_token
        w[16] = sum_blk.mlp.lin2.bias
        w[18:20] = [sum_blk.norm3.weight, sum_blk.norm3.bias]
        w[20:24] = _lin_pair(i2t.k_proj) + _lin_pair(i2t.v_proj)
    return w


class TwoWayTransformer(nn.Module):
    """transformer.py:16-106."""

    def __init__(self, depth, embedding_dim, num_heads, mlp_dim, activation=nn.ReLU, attention_downsample_rate=2):
        super().__init__()
        self.depth = depth
        self.embedding_dim = embedding_dim
        self.num_heads = num_heads
        self.mlp_dim = mlp_dim
        self.layers = nn.ModuleList()
        for i in range(depth):
            self.layers.append(TwoWayAttentionBlock(embedding_dim=embedding_dim, num_heads=num_heads, mlp_dim=mlp_dim,
                                                    activation=activation,
                                                    attention_downsample_rate=attention_downsample_rate,
                                                    skip_first_layer_pe=(i == 0)))
        self.final_attn_token_to_image = DecoderAttention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = nn.LayerNorm(embedding_dim)

    def run_tokens(self, src_tokens, pe_tokens, out_tokens, prompts, src_bias=None, prompt_image=None, prompt_tail=None):
        """transformer.py:62-106: the depth TwoWayAttentionBlocks and the final token->image attention + norm_final_attn.
        src_tokens [1|P, hw, C] bf16, pe_tokens [1, hw, C] bf16; out_tokens [5, C] fp32 (iou + mask tokens), prompts [P, 1, C] bf16: the
        point embedding of prompt p is cat(out_tokens, prompts[p]) (mask_decoder.py:125-132), built by the first launch; src_bias [1, C]:
        the image tokens are src_tokens + src_bias (the dense no-mask embedding, mask_decoder.py:136), folded into the first block;
        prompt_image (int32 [P]): src_tokens holds one block per IMAGE and prompt p belongs to image prompt_image[p] -- the reference
        repeats the image embedding per prompt (mask_decoder.py:135); the first block's image-side projection is the same for all prompts of
        an image, so it runs once per image and the first block's kernels read it through the map; prompt_tail (gamma, beta, text_type,
        log_temp, eps): `prompts` are the text projector's rows BEFORE its tail, applied by the first launch (ops.dec_tokens)
        -> (queries fp32 [P, 6, C] BEFORE the final attention's out_proj + norm_final_attn; the `combine` operands of that step, which
        ops.dec_heads applies in its own launch; keys [P, hw, C] bf16).
        Launch chain per block: tokens[close previous block | self attention | q] -> (previous block's image->token attention, norm4)
        -> image-side GEMM -> attention partials -> MLP partials (which first merge the partials: out_proj, norm2)."""
        if not self.fused_ok(out_tokens.shape[0], prompts.shape[1]):
            raise NotImplementedError("the fused token kernels are built for SAM's decoder geometry (256 channels, 8 heads, 5 + 1 tokens, "
                                      "MLP 2048); got %d / %d / %d + %d / %d -- MaskDecoder takes the per-op path for others"
                                      % (self.embedding_dim, self.num_heads, out_tokens.shape[0], prompts.shape[1], self.mlp_dim))
        P = prompts.shape[0]
        dev = prompts.device
        queries = torch.empty(P, 6, self.embedding_dim, device=dev, dtype=torch.float32)
        query_pe = torch.empty_like(queries)
        first = True
        fa = self.final_attn_token_to_image
        d = fa.internal_dim
        eps = self.norm_final_attn.eps
        q = torch.empty(P, 6, d, device=dev, dtype=torch.float32)
        kq = torch.empty(P, 6, d, device=dev, dtype=BF16)
        vq = torch.empty_like(kq)
        keys, prev, prev_proj, mlp_part = src_tokens, None, None, None
        stages = list(self.layers) + [None]                 # None = the tail
        for layer in stages:
            if layer is not None:
                layer.check_fused()
                if layer.norm1.eps != eps or layer.cross_attn_token_to_image.internal_dim != d:
                    raise NotImplementedError("the fused token kernels expect one LayerNorm eps and one attention width in the decoder")
                t2i, norm, bits = layer.cross_attn_token_to_image, layer.norm2, ops.TOK_SELF | ops.TOK_Q_T2I
            else:
                t2i, norm, bits = fa, self.norm_final_attn, ops.TOK_Q_T2I
            table = token_stage_table(self_blk=layer, t2i=t2i, norm=norm, sum_blk=prev)
            if prev is not None:
                bits |= ops.TOK_SUM_MLP
            if first:
                bits |= ops.TOK_INIT
            ops.dec_tokens(bits, layer is not None and layer.skip_first_layer_pe, queries, query_pe, table, q_t2i=q, mlp_partials=mlp_part,
                           k_i2t=kq, v_i2t=vq, eps=eps, init_tokens=out_tokens if first else None,
                           init_prompt=prompts.to(BF16).contiguous() if first else None, prompt_tail=prompt_tail if first else None)
            first = False
            if prev is not None:
                keys = prev.image_to_token(prev_proj, keys, kq, vq, P, src_bias, prompt_image)
                src_bias = prompt_image = None                    # (keys are per prompt and carry the bias from here on)
            proj = layer.image_side(keys, pe_tokens, src_bias) if layer is not None else self.final_image_side(keys, pe_tokens)
            part = ops.dec_attn_partial(q, proj[..., :2 * d], prompt_image)
            combine = (part, _tiled(t2i.out_proj.weight), t2i.out_proj.bias, norm.weight, norm.bias)
            if layer is not None:
                mlp_part, queries = layer.mlp_partials(queries, combine)
            prev, prev_proj = layer, proj
        return queries, combine, keys

    def final_image_side(self, keys, pe_tokens):
        """[K | V] of the final token->image attention (positional term folded into the additive table)."""
        fa = self.final_attn_token_to_image
        key = (pe_tokens.data_ptr(), pe_tokens._version) + tuple((p.data_ptr(), p._version) for p in
                                                                 (fa.k_proj.weight, fa.v_proj.weight, fa.k_proj.bias, fa.v_proj.bias))
        if getattr(self, "_fin_key", None) != key:
            self._fin_val = _interleave_kv_heads(*_pe_folded_projection(pe_tokens.reshape(-1, pe_tokens.shape[-1]),
                                                                        [(fa.k_proj, True), (fa.v_proj, False)]), fa.internal_dim, fa.num_heads)
            self._fin_key = key
        wcat, rtab = self._fin_val
        return ops.linear(keys, wcat, residual=rtab, res_row_mod=keys.shape[1])

    def fused_ok(self, n_out_tokens, n_prompt_tokens):
        """SAM's decoder geometry, the one the fused token kernels are built for (anything else runs op by op: run_general)."""
        fa = self.final_attn_token_to_image
        eps, d = self.norm_final_attn.eps, fa.internal_dim
        return (self.embedding_dim == 256 and self.num_heads == 8 and self.mlp_dim == 2048 and n_out_tokens == 5 and n_prompt_tokens == 1
                and d == 128 and all(layer.fused_ok(self.embedding_dim, eps, d) for layer in self.layers))

    def run_general(self, src_tokens, pe_tokens, point_embedding):
        """transformer.py:62-106 op by op: src_tokens [1|P, hw, C] rows, pe_tokens [1, hw, C], point_embedding [P, N, C] bf16
        -> (queries [P, N, C], keys [P, hw, C]) -- any width / head count / number of tokens."""
        P = point_embedding.shape[0]
        queries, keys = point_embedding, src_tokens
        for layer in self.layers:
            queries, keys = layer.run_general(queries, keys, point_embedding, pe_tokens, P)
        q = ops.add_rows(queries, point_embedding)
        fa = self.final_attn_token_to_image
        d = fa.internal_dim
        wcat, rtab = _pe_folded_projection(pe_tokens.reshape(-1, pe_tokens.shape[-1]), [(fa.k_proj, True), (fa.v_proj, False)])
        proj = ops.linear(keys, wcat, residual=rtab, res_row_mod=keys.shape[1])
        qp = ops.linear(q, fa.q_proj.weight, fa.q_proj.bias)
        queries = fa.run_projected(qp, proj[..., :d], proj[..., d:], P, residual=queries)
        return ops.layernorm(queries, self.norm_final_attn.weight, self.norm_final_attn.bias, self.norm_final_attn.eps), keys

    def forward(self, image_embedding, image_pe, point_embedding):
        """transformer.py:62-106 signature: image_embedding [B, C, h, w], image_pe the same shape, point_embedding [B, N, C]
        -> (processed point_embedding [B, N, C], processed image embedding [B, hw, C]).  (MaskDecoder drives the fused kernels through
        run_tokens for SAM's geometry; this entry runs op by op and takes any geometry.)"""
        _check_bf16_gpu(image_embedding, "image_embedding")
        src = ops.nchw_to_tokens(image_embedding.contiguous())
        pe = ops.nchw_to_tokens(image_pe.to(BF16).contiguous())[:1]
        return self.run_general(src, pe, point_embedding.to(BF16).contiguous())


class MLP(nn.Module):
    """mask_decoder.py:169-191."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, sigmoid_output=False):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        self.sigmoid_output = sigmoid_output

    def rows(self, x, out_f32=False):
        for i, layer in enumerate(self.layers):
            last = i == self.num_layers - 1
            x = ops.linear(x, layer.weight, layer.bias, act=ops.ACT_NONE if last else ops.ACT_RELU, out_f32=out_f32 and last)
        if self.sigmoid_output:
            raise NotImplementedError("sigmoid_output is not used on the WalkGPT path")
        return x


class MaskDecoder(nn.Module, _Prepared):
    """mask_decoder.py:16-164."""

    def __init__(self, *, transformer_dim, transformer, num_multimask_outputs=3, activation=nn.GELU, iou_head_depth=3,
                 iou_head_hidden_dim=256):
        super().__init__()
        self.transformer_dim = transformer_dim
        self.transformer = transformer
        self.num_multimask_outputs = num_multimask_outputs
        self.iou_token = nn.Embedding(1, transformer_dim)
        self.num_mask_tokens = num_multimask_outputs + 1
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, transformer_dim)
        self.output_upscaling = nn.Sequential(
            nn.ConvTranspose2d(transformer_dim, transformer_dim // 4, kernel_size=2, stride=2),
            LayerNorm2d(transformer_dim // 4),
            activation(),
            nn.ConvTranspose2d(transformer_dim // 4, transformer_dim // 8, kernel_size=2, stride=2),
            activation(),
        )
        self.output_hypernetworks_mlps = nn.ModuleList(
            [MLP(transformer_dim, transformer_dim, transformer_dim // 8, 3) for _ in range(self.num_mask_tokens)])
        self.iou_prediction_head = MLP(transformer_dim, iou_head_hidden_dim, self.num_mask_tokens, iou_head_depth)

    @staticmethod
    def _convt_as_gemm(weight):
        """ConvTranspose2d(k=2, s=2) == per-pixel GEMM whose output columns are (dy, dx, c_out): weight [Cin, Cout, 2, 2]
        -> [(dy, dx, Cout), Cin]; the bias repeats over the four sub-pixels."""
        return weight.permute(2, 3, 1, 0).reshape(-1, weight.shape[0]).contiguous()

    def _build_prepared(self):
        c1, c3 = self.output_upscaling[0], self.output_upscaling[3]
        return {
            "up1_w": ops.tile_weight(self._convt_as_gemm(c1.weight)),      # MFMA fragment order (wg_upscale_mask_bf16 keeps it in registers)
            "up2_w": self._convt_as_gemm(c3.weight),
            "out_tokens_f32": torch.cat([self.iou_token.weight, self.mask_tokens.weight], 0).float().contiguous(),
        }

    def head_weights(self):
        """Pointer-table order of wg_dec_heads_f32 (include/walkgpt_hip.h)."""
        w = []
        for mlp in list(self.output_hypernetworks_mlps) + [self.iou_prediction_head]:
            if mlp.num_layers != 3 or mlp.sigmoid_output:
                raise NotImplementedError("the fused head kernel is built for SAM's 3-layer hypernetwork / IoU MLPs")
            for layer in mlp.layers:
                w += _lin_pair(layer)
        return w

    def predict_masks_tokens(self, src_tokens, pe_tokens, sparse, h, w, mask_slice, src_bias=None, prompt_image=None, prompt_tail=None):
        """src_tokens [1|P, hw, C] (image embedding + dense prompt, channels-last rows; or the image embedding alone with the dense
        no-mask embedding as src_bias [1, C]; with prompt_image (int32 [P]) one block per image instead of per prompt), pe_tokens [1, hw, C],
        sparse [P, n, C] -> (masks fp32 [P, k, 4h, 4w], iou fp32 [P, k]).
        Launches: TwoWayTransformer.run_tokens, then one kernel for the hypernetwork / IoU heads and one for upscaling + the
        hypernetwork product.  prompt_tail (gamma, beta, text_type, log_temp, eps): `sparse` [P, 1, C] holds the text projector's rows
        before its tail (CalibratedTextProjector.pre_tail), which the first token launch applies."""
        p = self._prep_get(self._build_prepared, (self.output_upscaling[0].weight, self.output_upscaling[3].weight, self.iou_token.weight,
                                                  self.mask_tokens.weight))
        P = sparse.shape[0]
        tr = self.transformer
        if self.num_mask_tokens != 4 or self.transformer_dim != 256 or not tr.fused_ok(1 + self.num_mask_tokens, sparse.shape[1]) \
                or any(m.num_layers != 3 or m.sigmoid_output for m in list(self.output_hypernetworks_mlps) + [self.iou_prediction_head]):
            if prompt_tail is not None:
                g_, b_, tt_, lt_, teps = prompt_tail
                sparse = ops.ctp_tail(sparse.reshape(P, -1).contiguous(), g_, b_, tt_, lt_, teps).reshape(sparse.shape)
            return self._predict_masks_general(src_tokens, pe_tokens, sparse, h, w, mask_slice, src_bias, prompt_image)
        if len(tr.layers) == 0 or not tr.layers[0].fused_i2t_ok(src_tokens, 6):
            if prompt_image is not None:
                src_tokens, prompt_image = src_tokens.index_select(0, prompt_image.long()), None
            if src_bias is not None:
                src_tokens, src_bias = ops.add_rows(src_tokens, src_bias), None
        queries, final, keys = tr.run_tokens(src_tokens, pe_tokens, p["out_tokens_f32"], sparse, src_bias, prompt_image, prompt_tail)
        hyper, iou = ops.dec_heads(queries, self.head_weights(), combine=final, eps=tr.norm_final_attn.eps)
        ln1 = self.output_upscaling[1]
        k0, nk = mask_slice
        masks = ops.upscale_mask(keys, p["up1_w"], self.output_upscaling[0].bias, ln1.weight, ln1.bias, ln1.eps, p["up2_w"],
                                 self.output_upscaling[3].bias, hyper, h, w, k0, nk)
        return masks, iou[:, k0:k0 + nk]

    def _predict_masks_general(self, src_tokens, pe_tokens, sparse, h, w, mask_slice, src_bias=None, prompt_image=None):
        """mask_decoder.py:116-164 op by op, for what the fused kernels do not take: any number of sparse prompt tokens / mask tokens, other
        head counts or MLP widths (bf16 token stream; the upscaler's last width must be 32, what wg_hyper_mask_dot multiplies)."""
        P, C = sparse.shape[0], self.transformer_dim
        if C // 8 != 32:
            raise NotImplementedError("mask decoder widths other than 256 are not built (hyper_in @ upscaled runs on 32 channels)")
        if prompt_image is not None:
            src_tokens = src_tokens.index_select(0, prompt_image.long())
        if src_bias is not None:
            src_tokens = ops.add_rows(src_tokens, src_bias)
        out_tokens = torch.cat([self.iou_token.weight, self.mask_tokens.weight], 0).to(BF16)
        tokens = torch.cat([out_tokens.unsqueeze(0).expand(P, -1, -1), sparse.to(BF16)], dim=1).contiguous()      # :125-132
        hs, keys = self.transformer.run_general(src_tokens, pe_tokens, tokens)
        c1, ln1, c3 = self.output_upscaling[0], self.output_upscaling[1], self.output_upscaling[3]
        u = ops.linear(keys.reshape(P * h * w, C), self._convt_as_gemm(c1.weight), c1.bias.repeat(4).contiguous())     # [P*hw, 4 * C/4]
        u = ops.layernorm(u.view(P * h * w * 4, C // 4), ln1.weight, ln1.bias, ln1.eps, act=ops.ACT_GELU)
        u = ops.linear(u, self._convt_as_gemm(c3.weight), c3.bias.repeat(4).contiguous(), act=ops.ACT_GELU)             # [P*hw*4, 4 * C/8]
        hyper = torch.empty(P, self.num_mask_tokens, C // 8, device=u.device, dtype=BF16)
        for i in range(self.num_mask_tokens):
            hyper[:, i] = self.output_hypernetworks_mlps[i].rows(hs[:, 1 + i].contiguous())
        k0, nk = mask_slice
        masks = ops.hyper_mask_dot(u, hyper, P, h, w, k0, nk)
        iou = self.iou_prediction_head.rows(hs[:, 0].contiguous(), out_f32=True)
        return masks, iou[:, k0:k0 + nk]

    def forward(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output):
        """mask_decoder.py:75-114 signature.  image_embeddings [1,C,h,w]; dense [P,C,h,w] (normally the broadcast
        no_mask_embed); returns (masks [P,1|3,4h,4w] fp32, iou [P,1|3] fp32)."""
        _check_bf16_gpu(image_embeddings, "image_embeddings")
        b, c, h, w = image_embeddings.shape
        P = sparse_prompt_embeddings.shape[0]
        emb = ops.nchw_to_tokens(image_embeddings.contiguous())                 # [b, hw, C]
        d = dense_prompt_embeddings
        if d.stride(0) == 0 and d.stride(2) == 0 and d.stride(3) == 0 and b == 1:
            pe = ops.nchw_to_tokens(image_pe.to(BF16).contiguous())[:1]
            sl = (1, self.num_mask_tokens - 1) if multimask_output else (0, 1)
            return self.predict_masks_tokens(emb, pe, sparse_prompt_embeddings, h, w, sl,      # one image shared by all prompts
                                             src_bias=d[0, :, 0, 0].to(BF16).reshape(1, c).contiguous())
        else:
            src = ops.add_rows(emb.expand(P, -1, -1).contiguous() if b == 1 else emb,
                               ops.nchw_to_tokens(d.to(BF16).contiguous()).reshape(-1, c))
        pe = ops.nchw_to_tokens(image_pe.to(BF16).contiguous())[:1]
        sl = (1, self.num_mask_tokens - 1) if multimask_output else (0, 1)
        return self.predict_masks_tokens(src, pe, sparse_prompt_embeddings, h, w, sl)


class Sam(nn.Module):
    """sam.py:18-172 container + postprocess_masks (the only method WalkGPT calls on it)."""

    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, image_encoder, prompt_encoder, mask_decoder, pixel_mean=(123.675, 116.28, 103.53),
                 pixel_std=(58.395, 57.12, 57.375)):
        super().__init__()
        self.image_encoder = image_encoder
        self.prompt_encoder = prompt_encoder
        self.mask_decoder = mask_decoder
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)

    def postprocess_masks(self, masks, input_size, original_size):
        """sam.py:137-172 (fp32 out), fused into one pass by wg_postprocess_masks_f32."""
        return ops.postprocess_masks(masks.float().contiguous(), self.image_encoder.img_size, input_size, original_size)


def _build_sam(encoder_embed_dim, encoder_depth, encoder_num_heads, encoder_global_attn_indexes, checkpoint=None,
               image_size=1024):
    """build_sam.py:56-108."""
    prompt_embed_dim = 256
    vit_patch_size = 16
    image_embedding_size = image_size // vit_patch_size
    sam = Sam(
        image_encoder=ImageEncoderViT(depth=encoder_depth, embed_dim=encoder_embed_dim, img_size=image_size, mlp_ratio=4,
                                      norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=encoder_num_heads,
                                      patch_size=vit_patch_size, qkv_bias=True, use_rel_pos=True,
                                      global_attn_indexes=encoder_global_attn_indexes, window_size=14,
                                      out_chans=prompt_embed_dim),
        prompt_encoder=PromptEncoder(embed_dim=prompt_embed_dim, image_embedding_size=(image_embedding_size, image_embedding_size),
                                     input_image_size=(image_size, image_size), mask_in_chans=16),
        mask_decoder=MaskDecoder(num_multimask_outputs=3,
                                 transformer=TwoWayTransformer(depth=2, embedding_dim=prompt_embed_dim, mlp_dim=2048, num_heads=8),
                                 transformer_dim=prompt_embed_dim, iou_head_depth=3, iou_head_hidden_dim=256),
    )
    sam.eval()
    if checkpoint is not None:
        with open(checkpoint, "rb") as f:
            state_dict = torch.load(f)
        sam.load_state_dict(state_dict, strict=False)
    return sam


def build_sam_vit_h(checkpoint=None):
    return _build_sam(1280, 32, 16, [7, 15, 23, 31], checkpoint)


def build_sam_vit_l(checkpoint=None):
    return _build_sam(1024, 24, 16, [5, 11, 17, 23], checkpoint)


def build_sam_vit_b(checkpoint=None):
    return _build_sam(768, 12, 12, [2, 5, 8, 11], checkpoint)


build_sam = build_sam_vit_h
sam_model_registry = {"default": build_sam_vit_h, "vit_h": build_sam_vit_h, "vit_l": build_sam_vit_l, "vit_b": build_sam_vit_b}
