"""`utils.utils_walkgpt` of the reference (/root/reference/utils/utils_walkgpt.py) -> walkgpt_amd.utils_walkgpt (forward paths)."""
from walkgpt_amd import ops as _ops
from walkgpt_amd.utils_walkgpt import (  # noqa: F401
    CalibratedTextProjector, CrossAttnBlock, MultiScaleQFormerProjector, SegAwareGate, TinyCrossAttn, infonce_loss)


def sigmoid_ce_loss(inputs, targets, num_masks):
    """utils_walkgpt.py:100-120 (forward)."""
    return _ops.mask_losses(inputs.float().contiguous(), targets.float().contiguous(), num_masks)[0]


def dice_loss(inputs, targets, num_masks, scale=1000, eps=1e-6):
    """utils_walkgpt.py:76-97 (forward)."""
    return _ops.mask_losses(inputs.float().contiguous(), targets.float().contiguous(), num_masks, dice_scale=scale, dice_eps=eps)[1]
