"""ORACLE (test infrastructure, never the product path): CPU restatement of the evaluation metric and the mask losses'
forward.

  intersection_and_union()  /root/reference/utils/utils.py:192-204 (as called at evaluation_walkgpt.py:936-944)
  sigmoid_ce_loss()          /root/reference/utils/utils_walkgpt.py:103-120
  dice_loss()                /root/reference/utils/utils_walkgpt.py:76-99
Pinned by tests/golden/metrics.npz (outputs of the reference functions on synthetic masks).
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
