"""LLM-side splice of the visual tokens into the text embeddings, behind the reference's method name.

    /root/reference/model/llava_walkgpt/model/llava_arch.py:213-518   LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal
    /root/reference/model/walkgpt.py:293-306                           seg_token_mask (the "+255" shift)

The reference walks the rows in Python, slicing and concatenating embeddings, masks and labels; here one HIP launch gathers
everything (walkgpt_hip wg_splice_multimodal_bf16), fused with the embed_tokens lookup.  Scope = what WalkGPT feeds it: exactly
one IMAGE_TOKEN_INDEX per row, image features already resampled to T tokens (ops.resample_tokens) -- other inputs raise, as the
reference's `assert False` branches do.
"""
import torch

from . import _lib

IMAGE_TOKEN_INDEX = -200   # llava_walkgpt/constants.py
IGNORE_INDEX = -100


def prepare_inputs_labels_for_multimodal(input_ids, attention_mask, labels, image_features, embed_weight, vit_attention_mask=None,
                                         seg_token_idx=None, return_positions=False):
    """input_ids [rows,L] int64, attention_mask [rows,L] bool or None, labels [rows,L] int64 or None, image_features
    [rows,T,H] bf16, embed_weight [V,H] bf16 (the LLM's embed_tokens.weight), vit_attention_mask [rows,T] or None,
    seg_token_idx int / list of ints or None.
    Returns (attention_mask [rows,L+T-1] bool, new_input_embeds [rows,L+T-1,H] bf16, new_labels or None, seg_token_mask or None)."""
    for t, n in ((image_features, "image_features"), (embed_weight, "embed_tokens.weight")):
        if not t.is_cuda or t.dtype != torch.bfloat16:
            raise RuntimeError("%s must be a bf16 GPU tensor for the walkgpt_amd HIP path (got %s on %s); there is no CPU fallback"
                               % (n, t.dtype, t.device))
    assert input_ids.dtype == torch.int64 and input_ids.dim() == 2 and input_ids.is_cuda
    rows, L = input_ids.shape
    assert image_features.dim() == 3 and image_features.shape[0] == rows, "one image (T tokens) per row"
    T, H = image_features.shape[1], image_features.shape[2]
    V = embed_weight.shape[0]
    assert embed_weight.shape[1] == H
    dev = input_ids.device
    ids = input_ids.contiguous()
    Lo = L + T - 1
    embeds = torch.empty(rows, Lo, H, device=dev, dtype=torch.bfloat16)
    mask_in = attention_mask.to(torch.bool).contiguous() if attention_mask is not None else None   # None = ones (:246)
    vit = vit_attention_mask.to(torch.bool).contiguous() if vit_attention_mask is not None else None
    mask_out = torch.empty(rows, Lo, device=dev, dtype=torch.bool)
    lab_in = labels.contiguous() if labels is not None else None
    lab_out = torch.empty(rows, Lo, device=dev, dtype=torch.int64) if labels is not None else None
    seg_ids = seg_mask = None
    if seg_token_idx is not None:
        lst = list(seg_token_idx) if isinstance(seg_token_idx, (list, tuple)) else [int(seg_token_idx)]
        seg_ids = torch.tensor(lst, device=dev, dtype=torch.int64)
        seg_mask = torch.empty(rows, Lo, device=dev, dtype=torch.bool)
    pos = torch.empty(rows, device=dev, dtype=torch.int32)
    cnt = torch.empty(rows, device=dev, dtype=torch.int32)
    p = lambda t: t.data_ptr() if t is not None else None
    rc = _lib.lib().wg_splice_multimodal_bf16(ids.data_ptr(), embed_weight.contiguous().data_ptr(), image_features.contiguous().data_ptr(),
                                              p(mask_in), p(vit), p(lab_in), p(seg_ids), 0 if seg_ids is None else seg_ids.numel(),
                                              embeds.data_ptr(), mask_out.data_ptr(), p(lab_out), p(seg_mask), pos.data_ptr(), cnt.data_ptr(),
                                              rows, L, T, H, V, IMAGE_TOKEN_INDEX, IGNORE_INDEX, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "wg_splice_multimodal_bf16")
    if not bool((cnt == 1).all()):
        raise RuntimeError("every row must hold exactly one IMAGE_TOKEN_INDEX placeholder (found %s); other layouts are not on the "
                           "WalkGPT path (llava_arch.py:236-243,430-431 assert them away)" % cnt.tolist())
    text = ids[ids != IMAGE_TOKEN_INDEX]
    if bool(((text < 0) | (text >= V)).any()):
        raise IndexError("input_ids hold ids outside the embedding table")
    if return_positions:       # (+ the placeholder position of every row: what the splice's backward needs, walkgpt_amd.autograd)
        return mask_out, embeds, lab_out, seg_mask, pos
    return mask_out, embeds, lab_out, seg_mask
