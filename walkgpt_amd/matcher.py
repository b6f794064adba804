"""Prediction <-> ground-truth mask matching behind the reference's function name (/root/reference/utils/matcher.py:93-133).

The cost matrix (both mask sets sampled at 12544 shared random points, BCE + dice per pair) is computed on the GPU by
walkgpt_amd.ops.match_cost; the Hungarian assignment of the tiny [P, T] matrix stays on the host (scipy), as in the reference.
"""
import torch

from . import ops

NUM_POINTS = 12544   # matcher.py:96


def match_pred(out_mask, tgt_mask, point_coords=None):
    """out_mask [P,H,W] logits, tgt_mask [T,H,W] {0,1} (GPU tensors) -> (row_ind, col_ind) numpy arrays.

    `point_coords` ([NP,2] in [0,1]^2, x then y) may be passed for reproducibility; by default they are drawn exactly as the
    reference draws them (torch.rand(1, 12544, 2) on the masks' device, matcher.py:101)."""
    from scipy.optimize import linear_sum_assignment
    if point_coords is None:
        point_coords = torch.rand(1, NUM_POINTS, 2, device=out_mask.device)[0]
    cost = ops.match_cost(out_mask.detach().float().contiguous(), tgt_mask.detach().float().contiguous(),
                          point_coords.float().contiguous())
    return linear_sum_assignment(cost.cpu())
