"""ORACLE (test infrastructure, never the product path): CPU restatement of the evaluation metric and the mask losses'
forward.

  intersection_and_union()  /root/reference/utils/utils.py:192-204 (as called at evaluation_walkgpt.py:936-944)
  sigmoid_ce_loss()          /root/reference/utils/utils_walkgpt.py:103-120
  dice_loss()                /root/reference/utils/utils_walkgpt.py:76-99
  tiny_xattn()               /root/reference/utils/utils_walkgpt.py:330-357 (TinyCrossAttn.forward)
  infonce_loss()             /root/reference/utils/utils_walkgpt.py:8-73 (as called at model/walkgpt.py:459-473)
  match_cost()               /root/reference/utils/matcher.py:10-56,64-90,93-127 (match_pred's cost matrix for given points)
Pinned by tests/golden/metrics_*.npz, nce_*.npz and match_*.npz (outputs of the reference functions on synthetic inputs).
"""
import torch
import torch.nn.functional as F


def intersection_and_union(output, target, K=2, ignore_index=255):
    """output, target: integer class maps of equal shape -> (intersection[K], union[K], target_area[K]) as float32."""
    o = output.reshape(-1).clone()
    t = target.reshape(-1)
    o[t == ignore_index] = ignore_index
    hit = o[o == t]
    cnt = lambda v: torch.stack([(v == k).sum() for k in range(K)]).float()
    inter, ao, at = cnt(hit), cnt(o), cnt(t)
    return inter, ao + at - inter, at


def sigmoid_ce_loss(inputs, targets, num_masks):
    """inputs/targets [N,H,W]: per-mask mean BCE-with-logits, summed, / (num_masks + 1e-8)."""
    x, t = inputs.flatten(1), targets.flatten(1)
    bce = torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-x.abs()))
    return bce.mean(1).sum() / (num_masks + 1e-8)


def dice_loss(inputs, targets, num_masks, scale=1000, eps=1e-6):
    s, t = inputs.sigmoid().flatten(1), targets.flatten(1)
    num = 2 * (s / scale * t).sum(-1)
    den = (s / scale).sum(-1) + (t / scale).sum(-1)
    return (1 - (num + eps) / (den + eps)).sum() / (num_masks + 1e-8)


def tiny_xattn(w, q_vec, kv):
    """w: {wq,wk,wv,out}.weight (+ optional .bias); q_vec [M,d], kv [M,N,d] -> (v_pos [M,d], attn [M,N])."""
    lin = lambda x, n: F.linear(x, w[n + ".weight"], w.get(n + ".bias"))
    d = kv.shape[-1]
    q = lin(q_vec, "wq").unsqueeze(1)
    k, v = lin(kv, "wk"), lin(kv, "wv")
    attn = (q @ k.transpose(1, 2) / d ** 0.5).softmax(-1)
    return lin((attn @ v).squeeze(1), "out"), attn.squeeze(1)


def infonce_loss(w, pred, sam_tokens, seg_row_ids, temperature=0.07, top_k=None, exclude_same_row=True):
    """normalize=True form.  Returns (loss, dict(v_pos, attn_w, logits))."""
    M = pred.shape[0]
    rows, N, D = sam_tokens.shape
    KV = sam_tokens[seg_row_ids]
    v_pos, attn_w = tiny_xattn(w, pred, KV)
    if top_k is not None and 0 < top_k < N:
        vals, idx = torch.topk(attn_w, k=top_k, dim=1)
        alpha = vals / (vals.sum(1, keepdim=True) + 1e-12)
        v_pos = torch.einsum("mk,mkd->md", alpha, torch.gather(KV, 1, idx.unsqueeze(-1).expand(-1, -1, D)))
    Z, Vp = F.normalize(pred, dim=-1), F.normalize(v_pos, dim=-1)
    pos = (Z * Vp).sum(-1, keepdim=True)
    sim = Z @ F.normalize(sam_tokens.reshape(-1, D), dim=-1).T
    if exclude_same_row:
        own = torch.zeros(M, rows, dtype=torch.bool)
        own[torch.arange(M), seg_row_ids] = True
        sim = sim.masked_fill(own.unsqueeze(-1).expand(M, rows, N).reshape(M, rows * N), float("-inf"))
    logits = torch.cat([pos, sim], 1) / temperature
    loss = F.cross_entropy(logits, torch.zeros(M, dtype=torch.long))
    return loss, {"v_pos": v_pos, "attn_w": attn_w, "logits": logits}


def match_cost(out_mask, tgt_mask, point_coords):
    """out_mask [P,H,W] logits, tgt_mask [T,H,W], point_coords [NP,2] in [0,1]^2 (x, y) -> cost [P,T]."""
    def sample(m):
        g = (2.0 * point_coords - 1.0)[None, :, None, :].expand(m.shape[0], -1, -1, -1)
        return F.grid_sample(m[:, None].float(), g, align_corners=False)[:, 0, :, 0]
    x, y = sample(out_mask), sample(tgt_mask)
    pos = F.binary_cross_entropy_with_logits(x, torch.ones_like(x), reduction="none")
    neg = F.binary_cross_entropy_with_logits(x, torch.zeros_like(x), reduction="none")
    ce = (pos @ y.T + neg @ (1 - y).T) / x.shape[1]
    s = x.sigmoid()
    dice = 1 - (2 * s @ y.T + 1) / (s.sum(-1)[:, None] + y.sum(-1)[None, :] + 1)
    return ce + dice
