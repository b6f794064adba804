"""Training path of the grounding head: CalibratedTextProjector -> prompt encoder (text branch) -> MaskDecoder -> Sam.postprocess_masks ->
mask losses, composed from the differentiable HIP operators of walkgpt_amd.autograd so that `loss.backward()` reaches the parameters
train_walkgpt.py:347-350 leaves trainable (mask_decoder.*, text_hidden_fcs.*) and the LLM hidden states the [SEG] rows came from.

Reference structure restated operator by operator: utils_walkgpt.py:302-327 (CTP), prompt_encoder.py:149-186 (text branch),
mask_decoder.py:116-164, transformer.py:62-240, sam.py:137-172, utils_walkgpt.py:76-120; the per-image loop of model/walkgpt.py:716-737.
Inference uses the fused kernels (segment_anything.modeling); this is what they decompose into when gradients are asked for.  Tensor
re-arrangements (concatenation, broadcast, pixel shuffle, residual adds) are torch views / elementwise adds on the bf16 tensors; every
GEMM, LayerNorm, activation, attention, resample and loss -- forward and backward -- is a HIP kernel.
"""
import math

import torch

from . import autograd as ag
from . import ops

BF16 = torch.bfloat16


def _ln(x, n):
    return ag.layernorm(x, n.weight, n.bias, n.eps)


def _lin(x, layer, act=ops.ACT_NONE):
    return ag.linear(x, layer.weight, layer.bias, act)


def ctp_forward(ctp, x):
    """CalibratedTextProjector.forward (utils_walkgpt.py:316-327) on [M, H_llm] bf16 rows."""
    if ctp.use_residual:
        raise NotImplementedError("use_residual=True is never configured by WalkGPT (walkgpt.py:115-123)")
    net = ctp.net
    y = _lin(_ln(x.contiguous(), net[0]), net[1], ops.ACT_GELU)
    y = _lin(y, net[3])
    y = ag.add_row(_ln(y, net[4]), ctp.text_type.reshape(-1))
    return ag.l2norm_scale(y, ctp.log_temp, 1e-12)


def _attention(att, q, k, v):
    """transformer.py:185-240: projections, per-head softmax(q k^T / sqrt(d)) v, out_proj.  q [P, Nq, C], k / v [P, Nk, C]."""
    qp, kp, vp = _lin(q, att.q_proj), _lin(k, att.k_proj), _lin(v, att.v_proj)
    hd = att.internal_dim // att.num_heads
    o = ag.attention(qp, kp, vp, att.num_heads, 1.0 / math.sqrt(hd))
    return _lin(o, att.out_proj)


def _two_way_block(blk, queries, keys, query_pe, key_pe):
    """transformer.py:151-182."""
    if blk.skip_first_layer_pe:
        queries = _attention(blk.self_attn, queries, queries, queries)
    else:
        q = queries + query_pe
        queries = queries + _attention(blk.self_attn, q, q, queries)
    queries = _ln(queries, blk.norm1)
    k = keys + key_pe                      # (the image tokens do not change until the end of the block: one sum for both cross attentions)
    q = queries + query_pe
    queries = _ln(queries + _attention(blk.cross_attn_token_to_image, q, k, keys), blk.norm2)
    h = _lin(_lin(queries, blk.mlp.lin1, blk.mlp._act_code), blk.mlp.lin2)
    queries = _ln(queries + h, blk.norm3)
    q = queries + query_pe
    keys = _ln(keys + _attention(blk.cross_attn_image_to_token, k, q, queries), blk.norm4)
    return queries, keys


def _two_way_transformer(tr, src_tokens, pos_tokens, point_embedding):
    """transformer.py:62-106 on channels-last rows: src_tokens / pos_tokens [P, hw, C], point_embedding [P, Nt, C]."""
    queries, keys = point_embedding, src_tokens
    for blk in tr.layers:
        queries, keys = _two_way_block(blk, queries, keys, point_embedding, pos_tokens)
    q, k = queries + point_embedding, keys + pos_tokens
    queries = _ln(queries + _attention(tr.final_attn_token_to_image, q, k, keys), tr.norm_final_attn)
    return queries, keys


def _mlp3(mlp, x):
    """mask_decoder.py:167-188 (ReLU between layers, no sigmoid on WalkGPT's path)."""
    n = len(mlp.layers)
    for i, layer in enumerate(mlp.layers):
        x = _lin(x, layer, ops.ACT_RELU if i < n - 1 else ops.ACT_NONE)
    if mlp.sigmoid_output:
        raise NotImplementedError("sigmoid_output heads are not on WalkGPT's path")
    return x


def _convt2x2_rows(x_rows, conv, P, h, w):
    """ConvTranspose2d(k=2, s=2) on channels-last rows [P*h*w, Cin] -> [P*2h*2w, Cout]: a per-pixel GEMM whose output columns are
    (dy, dx, c_out), then the pixel shuffle (a view + one copy)."""
    cout = conv.weight.shape[1]
    wg = conv.weight.permute(2, 3, 1, 0).reshape(-1, conv.weight.shape[0])        # [(dy, dx, Cout), Cin]
    y = ag.linear(x_rows, wg.contiguous(), conv.bias.repeat(4))
    y = y.view(P, h, w, 2, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(P * 2 * h * 2 * w, cout)
    return y


def decoder_forward(dec, src_tokens, pos_tokens, sparse, dense_vec, h, w, mask_slice=(0, 1)):
    """MaskDecoder.predict_masks (mask_decoder.py:116-164) for P prompts.
    src_tokens [P, hw, C] (each prompt's image embedding, channels-last rows), pos_tokens [1 or P, hw, C], sparse [P, n, C],
    dense_vec [C] (the no-mask embedding, the same at every pixel: prompt_encoder.py:181-184).
    -> (low-res mask logits fp32 [P, k, 4h, 4w], iou predictions [P, k])."""
    P = sparse.shape[0]
    C = dec.transformer_dim
    out_tokens = torch.cat([dec.iou_token.weight, dec.mask_tokens.weight], 0).unsqueeze(0).expand(P, -1, -1)
    tokens = torch.cat([out_tokens, sparse], 1).contiguous()
    src = ag.add_row(src_tokens.contiguous(), dense_vec)
    pos = pos_tokens.expand(P, -1, -1)
    hs, keys = _two_way_transformer(dec.transformer, src, pos, tokens)
    nm = dec.num_mask_tokens
    iou_out, mask_out = hs[:, 0], hs[:, 1:1 + nm]
    up = _convt2x2_rows(keys.reshape(P * h * w, C), dec.output_upscaling[0], P, h, w)
    n1 = dec.output_upscaling[1]
    up = ag.activation(ag.layernorm(up, n1.weight, n1.bias, n1.eps), ops.ACT_GELU)
    up = ag.activation(_convt2x2_rows(up, dec.output_upscaling[3], P, 2 * h, 2 * w), ops.ACT_GELU)      # [P*4h*4w, C/8]
    hyper = torch.stack([_mlp3(dec.output_hypernetworks_mlps[i], mask_out[:, i].contiguous()) for i in range(nm)], 1)   # [P, nm, C/8]
    k0, k1 = mask_slice[0], mask_slice[0] + mask_slice[1]
    up = up.view(P, 16 * h * w, -1)
    if up.shape[-1] == 32 and k1 - k0 <= 4:      # every prompt's product in one launch (its weights are the prompt's own hypernetwork output)
        masks = ag.hyper_rows(up, hyper[:, k0:k1]).view(P, k1 - k0, 4 * h, 4 * w)
    else:                                        # other widths: a GEMM per prompt
        masks = torch.stack([ag.linear(up[p], hyper[p, k0:k1].contiguous(), None, out_f32=True).t().reshape(k1 - k0, 4 * h, 4 * w) for p in range(P)], 0)
    iou = _mlp3(dec.iou_prediction_head, iou_out.contiguous())
    return masks, iou[:, k0:k1]


class _MaskList(list):
    """pred_masks[i] as the reference returns them, plus `.stacked`: every image's masks as ONE tensor [sum T_i, H0, W0] when all images share
    a size (the list entries are then slices of it).  A caller that reduces over all masks at once (causal_lm.model_forward's batched mask
    losses) takes `.stacked`: concatenating the slices back costs a zero-fill, a copy and an add per image in the backward pass."""
    stacked = None


def decode(grounding, emb_tokens, pred_embeddings, resize_list, original_size_list, multimask_output=False):
    """WalkGPTGrounding.decode with gradients: emb_tokens [B, hw, 256] (frozen SAM embedding), pred_embeddings[i] [T_i, 256] (CTP output).
    -> pred_masks[i] fp32 [T_i, H0, W0] (logits), differentiable in pred_embeddings and the decoder's parameters (a _MaskList)."""
    vm = grounding.visual_model
    h, w = vm.prompt_encoder.image_embedding_size
    pe = vm.prompt_encoder.dense_pe_tokens().unsqueeze(0).to(BF16)
    no_mask = vm.prompt_encoder.no_mask_embed.weight.reshape(-1)
    dec = vm.mask_decoder
    sl = (1, dec.num_mask_tokens - 1) if multimask_output else (0, 1)
    counts = [int(p.shape[0]) for p in pred_embeddings]
    P = sum(counts)
    dev = emb_tokens.device
    out = _MaskList([None] * len(counts))
    if P > 0:
        # every prompt of every image in ONE decoder pass (prompt p attends to the embedding of its own image), as WalkGPTGrounding.decode does
        if hasattr(grounding, "_prompt_image_index"):                   # (cached per count tuple: no host-to-device copy per step)
            pimg = grounding._prompt_image_index(tuple(counts), dev)
        else:
            pimg = torch.tensor([i for i, c in enumerate(counts) for _ in range(c)], device=dev)
        src = emb_tokens.index_select(0, pimg)
        sparse = torch.cat([p for p in pred_embeddings if p.shape[0]], 0).unsqueeze(1)
        low_res, _ = decoder_forward(dec, src, pe, sparse, no_mask, h, w, sl)
        same = len(set(zip(map(tuple, resize_list), map(tuple, original_size_list)))) == 1
        if same:
            full = ag.postprocess_masks(low_res.contiguous(), vm.image_encoder.img_size, resize_list[0], original_size_list[0])[:, 0]
            out.stacked = full
        off = 0
        for i, c in enumerate(counts):
            if c:
                out[i] = full[off:off + c] if same else ag.postprocess_masks(low_res[off:off + c].contiguous(), vm.image_encoder.img_size, resize_list[i],
                                                                             original_size_list[i])[:, 0]
            off += c
    for i, c in enumerate(counts):
        if c == 0:
            H0, W0 = original_size_list[i]
            out[i] = torch.zeros(0, H0, W0, device=dev)
    return out


def _cross_block(blk, queries, kv):
    """CrossAttnBlock (utils_walkgpt.py:163-185): pre-LN cross attention of the query tokens to the image tokens + GELU MLP, both residual."""
    D = queries.shape[-1]
    a = blk.attn
    q = ag.linear(_ln(queries, blk.q_norm), a.in_proj_weight[:D], a.in_proj_bias[:D])
    kvp = ag.linear(_ln(kv, blk.kv_norm), a.in_proj_weight[D:], a.in_proj_bias[D:])
    o = ag.attention(q, kvp[..., :D], kvp[..., D:], blk.nhead, 1.0 / math.sqrt(D // blk.nhead))
    out = queries + ag.linear(o, a.out_proj.weight, a.out_proj.bias)
    h = _lin(_lin(_ln(out, blk.ffn[0]), blk.ffn[1], ops.ACT_GELU), blk.ffn[3])
    return out + h


def _gate(gate, kv):
    """SegAwareGate (utils_walkgpt.py:204-217)."""
    h = _lin(_ln(kv, gate.net[0]), gate.net[1], ops.ACT_GELU)
    logit = ag.linear(h, gate.net[3].weight, gate.net[3].bias, out_f32=True)
    return ag.sigmoid_gate(kv, logit)


def msqp_forward(proj, sam_feats, grid_size=None):
    """MultiScaleQFormerProjector.forward (utils_walkgpt.py:220-300): sam_feats [B, L, sam_dim] bf16 -> [B, s*s, llama_dim]."""
    B, L, _ = sam_feats.shape
    if grid_size is None and proj.grid_size is None:
        H = int(math.sqrt(L))
        if H * H != L:
            raise ValueError(f"Token length {L} is not a perfect square.")
        W = H
    else:
        H, W = grid_size or proj.grid_size
    feats = _lin(sam_feats.contiguous(), proj.sam_to_proj)
    scales = [(proj.q_x1, proj.cross_x1, feats),
              (proj.q_x2, proj.cross_x2, ag.avgpool_tokens(feats, B, H, W, 2)),
              (proj.q_x4, proj.cross_x4, ag.avgpool_tokens(feats, B, H, W, 4)),
              (proj.q_global, proj.cross_glb, ag.mean_tokens(feats))]
    outs = []
    for q_param, layers, kv in scales:
        kv = _gate(proj.gate, kv)
        q = q_param.expand(B, -1, -1)
        for blk in layers:
            q = _cross_block(blk, q, kv)
        outs.append(q)
    vis = torch.cat(outs, 1)
    if proj.pad_to_square:
        Q = vis.shape[1]
        s = int(math.ceil(math.sqrt(Q))) if proj.target_square_side is None else proj.target_square_side
        assert s * s >= Q, "target_square_side too small"
        if s * s > Q:
            vis = torch.cat([vis, proj.pad_token.expand(B, s * s - Q, -1)], 1)
    return _lin(vis.contiguous(), proj.to_llama)


def infonce_loss(pred, sam_tokens, seg_row_ids, tiny_xattn, temperature=0.07, top_k=8, exclude_same_row=True):
    """utils_walkgpt.py:8-73 (normalize=True) with gradients on `pred` ([SEG] embeddings out of CTP, [M, D] bf16) and on TinyCrossAttn: wq / wk in
    the top_k form model/walkgpt.py:459-473 calls (its wv / out do not enter the loss there), all four projections when top_k is None / <= 0 /
    >= N (the function's own default: the positive is TinyCrossAttn's output).  sam_tokens [rows, N, D] bf16: the frozen SAM embedding rows.
    Which top_k tokens are pooled is a selection (no gradient), made from the attention weights of the forward kernel; the pooled positive, both
    normalisations, the similarity GEMM and the cross-entropy are differentiable HIP operators."""
    M, D = pred.shape
    rows, N, _ = sam_tokens.shape
    refine = top_k is not None and 0 < top_k < N                                                     # utils_walkgpt.py:36
    tx = tiny_xattn
    with torch.no_grad():
        one = torch.zeros(1, device=pred.device, dtype=BF16)                                          # log_temp = 0: a plain F.normalize
        that = ag.l2norm_scale(sam_tokens.reshape(rows * N, D).contiguous(), one)
        if refine:
            r = ops.nce_forward(pred.detach().contiguous(), sam_tokens.contiguous(), seg_row_ids, tx.wq.weight, tx.wq.bias,
                                tx._prep_get(tx._build)["wk_t"], temperature, None, exclude_same_row)      # (its attention weights only)
            idx = torch.topk(r["attn_w"], k=top_k, dim=1).indices                                    # [M, top_k]
            kt = torch.gather(sam_tokens.index_select(0, seg_row_ids), 1, idx.unsqueeze(-1).expand(-1, -1, D)).contiguous()
    q = ag.linear(pred, tx.wq.weight, tx.wq.bias)
    u = ag.linear(q, tx.wk.weight.t().contiguous(), None)                                             # W_k^T q (b_k shifts all scores of a row alike)
    if refine:
        v_pos = ag.topk_pool(u, kt)                                                                   # raw tokens, renormalised weights (:37-40)
    else:
        # sum_n a_n (W_v kv_n + b_v) = W_v (sum_n a_n kv_n) + b_v: the row is pooled raw, then projected (utils_walkgpt.py:349-356)
        ctx = ag.pool_rows(u, sam_tokens.contiguous(), seg_row_ids)
        v_pos = ag.linear(ag.linear(ctx, tx.wv.weight, tx.wv.bias), tx.out.weight, tx.out.bias)
    z, vp = ag.l2norm_scale(pred, one), ag.l2norm_scale(v_pos, one)
    sim = ag.linear(z, that, None, out_f32=True)                                                      # [M, rows * N] cosines
    return ag.nce_tail(z, vp, sim, seg_row_ids.to(torch.int32).contiguous(), rows, N, temperature, exclude_same_row)
