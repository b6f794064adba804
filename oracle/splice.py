"""ORACLE (test infrastructure, never the product path): CPU restatement of the LLM-side multimodal splice.

  prepare_inputs_labels_for_multimodal()  /root/reference/model/llava_walkgpt/model/llava_arch.py:265-518, the branch WalkGPT takes
                                          (one image placeholder per row, `mm_use_im_start_end` False or True: both branches build
                                          the same sequence, :330-378)
  seg_token_mask()                        /root/reference/model/walkgpt.py:293-306

Pinned by tests/golden/splice_{r3,vit_mask}.npz: outputs of the reference's own method on synthetic rows (placeholder first / mid-row /
near the end, a masked text position, labels, a ViT patch mask).  llava_arch.py is loaded for that on its own, outside its package
(whose __init__ chain does not import under the installed transformers 5.x); how is written down in tests/golden/make_golden.py:make_splice.
The restatement follows the source line by line (per-row torch.cat of the same slices) and is bit-exact against those vectors
(tests/test_oracle_golden.py:test_splice_oracle_vs_reference); seg_token_mask has no reference fixture (walkgpt.py does not import here)
and is checked on a hand-built case.
"""
import torch

IMAGE_TOKEN_INDEX = -200
IGNORE_INDEX = -100


def prepare_inputs_labels_for_multimodal(input_ids, attention_mask, labels, image_features, embed_weight, vit_attention_mask=None):
    rows, L = input_ids.shape
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)                                                    # :246
    if vit_attention_mask is None:
        vit_attention_mask = torch.ones_like(image_features[..., 0])                                   # :262-263
    embeds, masks, labs = [], [], []
    for r in range(rows):
        cur = input_ids[r]
        where = torch.where(cur == IMAGE_TOKEN_INDEX)[0]
        assert where.numel() == 1
        s = int(where[0])
        embeds.append(torch.cat([embed_weight[cur[:s]], image_features[r], embed_weight[cur[s + 1:]]], 0))   # :357-365, :404-408
        masks.append(torch.cat([attention_mask[r][:s].bool(), vit_attention_mask[r].bool(), attention_mask[r][s + 1:].bool()], 0))
        if labels is not None:
            labs.append(torch.cat([labels[r][:s], torch.full((image_features.shape[1],), IGNORE_INDEX, dtype=labels.dtype),
                                   labels[r][s + 1:]], 0))                                            # :367-378
    return torch.stack(masks), torch.stack(embeds), (torch.stack(labs) if labels is not None else None)


def seg_token_mask(input_ids, seg_ids, n_image_tokens=256):
    m = torch.zeros_like(input_ids[:, 1:], dtype=torch.bool)
    for s in seg_ids:
        m = m | (input_ids[:, 1:] == s)
    m = torch.cat([m, torch.zeros(m.shape[0], 1, dtype=torch.bool)], 1)
    return torch.cat([torch.zeros(m.shape[0], n_image_tokens - 1, dtype=torch.bool), m], 1)
