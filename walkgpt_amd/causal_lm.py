"""Top-level surface of the reference model, `walkgptForCausalLM` (/root/reference/model/walkgpt.py:155-746), over the HIP modules.

What callers of the reference touch (train_walkgpt.py:19, evaluation_walkgpt.py:18,673,916) and what this adapter keeps:
  forward(**kw)          -> model_forward unless `past_key_values` is passed (walkgpt.py:262-265)
  model_forward(...)     the collate_fn dict of utils/dataset.py:180-197, same argument names; returns the reference's dicts
                         (inference: pred_masks / gt_masks / batch_seg_token_counts / mask_scores, walkgpt.py:549-555; training: the six
                         loss entries, :598-605 -- forward values only, the HIP ops carry no autograd)
  evaluate(...)          same signature and return tuple as walkgpt.py:607-746
  get_visual_embs(x)     walkgpt.py:241-258
  get_model(), .model.{visual_model, out_mm_projector, text_hidden_fcs, vision_tower, tiny_xattn}, get_vision_tower()

The language model is NOT part of this build (stock PyTorch / transformers): it is injected as a module that speaks the
transformers causal-LM protocol -- `get_input_embeddings()`, `forward(inputs_embeds=, attention_mask=, labels=, past_key_values=,
use_cache=, output_hidden_states=)` returning `.logits`, `.loss`, `.hidden_states`, `.past_key_values` -- which LlamaForCausalLM does.
Around it everything runs on the HIP path: SAM encoder, MSQP, token resample, the multimodal splice, CTP, prompt encoder, mask decoder,
postprocess, mask score, mask losses, InfoNCE.

One deliberate difference.  The released `model_forward` builds the [SEG] embeddings and then leaves `pred_masks` / `mask_scores`
empty (walkgpt.py:541-555: the decode loop is missing there; its training branch then indexes the empty list).  The wiring that does
decode is `evaluate()`'s (:713-737); `model_forward` here fills the lists with that wiring, so the returned dict has the reference's
keys with usable contents.  `decode_masks=False` reproduces the empty lists.
"""
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .llava_splice import IMAGE_TOKEN_INDEX, prepare_inputs_labels_for_multimodal
from .utils_walkgpt import TinyCrossAttn, infonce_loss
from .walkgpt import WalkGPTGrounding

BF16 = torch.bfloat16


class walkgptForCausalLM(nn.Module):  # noqa: N801  (the reference's class name)
    def __init__(self, llm: nn.Module, grounding: Optional[WalkGPTGrounding] = None, **kwargs):
        super().__init__()
        self.llm = llm
        hidden = llm.get_input_embeddings().weight.shape[1]
        self.model = grounding if grounding is not None else WalkGPTGrounding(
            sam=kwargs.get("sam", "vit_h"), llm_hidden=hidden, out_dim=kwargs.get("out_dim", 256), with_clip=kwargs.get("with_clip", True))
        if not hasattr(self.model, "tiny_xattn"):      # walkgpt.py:104-113; created where the other grounding modules already live
            ref = next(self.model.visual_model.mask_decoder.parameters())
            self.model.tiny_xattn = TinyCrossAttn(kwargs.get("out_dim", 256)).to(device=ref.device, dtype=ref.dtype)
        # walkgpt.py:163-186
        self.ce_loss_weight = kwargs.get("ce_loss_weight", 1.0)
        self.dice_loss_weight = kwargs.get("dice_loss_weight", 0.5)
        self.bce_loss_weight = kwargs.get("bce_loss_weight", 2.0)
        self.seg_token_idx = kwargs.get("seg_token_idx")
        self.seg_token_num = kwargs.get("seg_token_num", 1)
        self.image_feature_scale_num = kwargs.get("image_feature_scale_num", 1)
        self.nce_tau = kwargs.get("nce_tau", 0.07)
        self.nce_topk = kwargs.get("nce_topk", 8)
        self.eos_token_id = kwargs.get("eos_token_id", getattr(getattr(llm, "config", None), "eos_token_id", None))

    @classmethod
    def from_pretrained(cls, version, llm=None, **model_args):
        """The reference loads a LLaVA checkpoint from `version` (evaluation_walkgpt.py:204-225).  Checkpoints and the language model are
        outside this build: pass the already loaded causal LM as `llm=`; the grounding modules then load with `load_state_dict`."""
        if llm is None:
            raise RuntimeError("walkgpt_amd does not build or download the language model: call from_pretrained(version, llm=<causal LM>, ...)")
        return cls(llm, **model_args)

    # -- accessors the reference's scripts use (evaluation_walkgpt.py:244-248,314,331) ------------------------------------------------
    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return getattr(self.model, "vision_tower", None)

    def get_visual_embs(self, pixel_values):
        vm = getattr(self.model, "visual_model", None)
        if vm is None or not hasattr(vm, "image_encoder"):
            raise RuntimeError("visual_model not initialized. Make sure initialize_walkgpt_modules() was called during __init__ "
                               "(before DeepSpeed initialize).")
        return vm.image_encoder(pixel_values)

    def forward(self, **kwargs):
        if "past_key_values" in kwargs:
            return self.llm(**kwargs)
        return self.model_forward(**kwargs)

    # -- helpers -------------------------------------------------------------------------------------------------------------------------
    def _seg_ids(self):
        if self.seg_token_idx is None:
            raise RuntimeError("seg_token_idx is not set (the id of the [SEG] token in the tokenizer)")
        return list(self.seg_token_idx) if isinstance(self.seg_token_idx, (list, tuple)) else [int(self.seg_token_idx)]

    def _seg_token_mask(self, ids, pad_right):
        """walkgpt.py:284-306 / 645-659: marks the position BEFORE each [SEG] id, shifted by the 255 extra image positions."""
        m = torch.zeros_like(ids[:, 1:]).bool()
        for s in self._seg_ids():
            m = m | (ids[:, 1:] == s)
        parts = [torch.zeros((m.shape[0], 255), dtype=torch.bool, device=ids.device), m]
        if pad_right:
            parts.append(torch.zeros((m.shape[0], 1), dtype=torch.bool, device=ids.device))
        return torch.cat(parts, dim=1)

    def _pack_queries(self, batch_pred_embeddings):
        """walkgpt.py:431-447: [total, D] -> highest-resolution scale of every query, [Q * seg_token_num, D]."""
        n, f = self.seg_token_num, self.image_feature_scale_num
        total = batch_pred_embeddings.shape[0]
        assert total % (n * f) == 0, f"Bad pack: total={total}, seg_token_num={n}, feat_scale_num={f}"
        Q, D = total // (n * f), batch_pred_embeddings.shape[-1]
        return batch_pred_embeddings.view(Q, f, n, D)[:, -1].reshape(Q * n, D)

    def _llm_inputs(self, input_ids, attention_mask, labels, image_tokens):
        """LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal on already projected image tokens (llava_arch.py:252-259 resample to
        16x16, :265-518 splice)."""
        feats = ops.resample_tokens(image_tokens.contiguous(), 16)
        return prepare_inputs_labels_for_multimodal(input_ids, attention_mask, labels, feats, self.llm.get_input_embeddings().weight)

    # -- walkgpt.py:267-605 ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def model_forward(self, images, images_clip, input_ids, labels, attention_masks, offset, masks_list: List[torch.Tensor],
                      label_list: List[torch.Tensor], resize_list: List[tuple], inference: bool = False, clip_resize_list=None,
                      decode_masks: bool = True, **kwargs):
        batch_size = images.shape[0]
        assert batch_size == len(offset) - 1
        seg_token_mask = self._seg_token_mask(input_ids, pad_right=True)
        if inference:
            assert images_clip.shape[0] == 1, "inference branch assumes one image"
        off = [int(v) for v in offset]
        # SAM encoder once for the batch, MSQP on all images at once (the reference calls it image by image: :364-378)
        emb_tokens = self.model.get_visual_emb_tokens(images)                        # [B, hw, 256] channels-last rows
        tokens_proj = self.model.out_mm_projector(emb_tokens)                         # [B, 36, H]
        row_img = torch.tensor([i for i in range(batch_size) for _ in range(off[i + 1] - off[i])], device=images.device)
        if inference:
            row_img = torch.zeros(input_ids.shape[0], dtype=torch.long, device=images.device)
        sam_tokens = tokens_proj.index_select(0, row_img)                             # one row of image tokens per text row
        sam_tokens_256 = emb_tokens.index_select(0, row_img)
        attn, embeds, new_labels, _ = self._llm_inputs(input_ids, attention_masks, None if inference else labels, sam_tokens)
        output = self.llm(inputs_embeds=embeds, attention_mask=attn, labels=new_labels, output_hidden_states=True)
        last_hidden = output.hidden_states[-1]
        assert len(self.model.text_hidden_fcs) == 1
        # CTP is a per-token map: projecting the gathered [SEG] rows equals projecting the sequence and gathering (:406-409)
        assert seg_token_mask.shape[1] == last_hidden.shape[1], (seg_token_mask.shape, last_hidden.shape)
        seg_hidden = last_hidden[seg_token_mask]
        pred_embeddings = self.model.text_hidden_fcs[0](seg_hidden.to(BF16)) if seg_hidden.shape[0] else seg_hidden.new_zeros(0, 256, dtype=BF16)
        pred_embeddings_nce = pred_embeddings
        seg_token_counts = seg_token_mask.int().sum(-1)
        seg_token_offset = torch.cat([seg_token_counts.new_zeros(1), seg_token_counts.cumsum(-1)], 0)
        if inference:
            seg_off = [0, int(seg_token_offset[-1])]
        else:
            seg_off = [int(seg_token_offset[o]) for o in off]
        pred_list, batch_seg_token_counts = [], []
        for i in range(len(seg_off) - 1):
            e = self._pack_queries(pred_embeddings[seg_off[i]:seg_off[i + 1]])
            pred_list.append(e)
            batch_seg_token_counts.append(e.shape[0])
        # region-alignment InfoNCE (:449-473)
        seg_row_ids = torch.repeat_interleave(torch.arange(sam_tokens_256.size(0), device=images.device), seg_token_counts)
        loss_nce = torch.zeros((), device=images.device)
        if seg_row_ids.numel() > 0 and not inference:
            loss_nce = infonce_loss(pred_embeddings_nce, sam_tokens_256, seg_row_ids, self.model.tiny_xattn, temperature=self.nce_tau,
                                    top_k=self.nce_topk, exclude_same_row=sam_tokens_256.size(0) > 1, normalize=True)
        pred_masks, mask_scores = [], []
        if decode_masks:   # evaluate()'s wiring (:713-737); the released model_forward leaves both lists empty
            sizes = [tuple(l.shape[-2:]) for l in label_list]
            pred_masks, mask_scores = self.model.decode(emb_tokens[:1] if inference else emb_tokens, pred_list, resize_list, sizes)
        if inference:
            return {"pred_masks": pred_masks, "gt_masks": masks_list, "batch_seg_token_counts": batch_seg_token_counts,
                    "mask_scores": mask_scores}
        ce_loss = output.loss * self.ce_loss_weight
        mask_bce_loss = torch.zeros((), device=images.device)
        mask_dice_loss = torch.zeros((), device=images.device)
        num_masks = 0
        for gt_mask, pred_mask in zip(masks_list, pred_masks):
            assert gt_mask.shape[0] == pred_mask.shape[0], "gt_mask.shape: {}, pred_mask.shape: {}".format(gt_mask.shape, pred_mask.shape)
            if gt_mask.shape[0] == 0:
                continue
            bce, dice = ops.mask_losses(pred_mask.float().contiguous(), gt_mask.float().contiguous(), num_masks=gt_mask.shape[0])
            mask_bce_loss = mask_bce_loss + bce * gt_mask.shape[0]
            mask_dice_loss = mask_dice_loss + dice * gt_mask.shape[0]
            num_masks += gt_mask.shape[0]
        mask_bce_loss = self.bce_loss_weight * mask_bce_loss / (num_masks + 1e-8)
        mask_dice_loss = self.dice_loss_weight * mask_dice_loss / (num_masks + 1e-8)
        mask_loss = mask_bce_loss + mask_dice_loss
        nce_loss = 0.2 * loss_nce
        loss = ce_loss + mask_loss + nce_loss
        return {"loss": loss, "ce_loss": ce_loss, "mask_bce_loss": mask_bce_loss, "mask_dice_loss": mask_dice_loss,
                "nce_loss": nce_loss, "mask_loss": mask_loss}

    # -- walkgpt.py:607-746 ------------------------------------------------------------------------------------------------------------
    def _generate(self, embeds, attn, max_new_tokens):
        """Greedy decoding through the injected LM's KV cache (what `self.generate(..., num_beams=1, output_hidden_states=True,
        return_dict_in_generate=True)` does in the reference, :629-639).  Returns (new token ids [1, n], last-layer hidden states of the
        prompt and of every generated token that was fed back: [1, L0 + n - 1, H])."""
        table = self.llm.get_input_embeddings().weight
        out = self.llm(inputs_embeds=embeds, attention_mask=attn, use_cache=True, output_hidden_states=True)
        hs, new = [out.hidden_states[-1]], []
        for step in range(max_new_tokens):
            nxt = out.logits[:, -1].argmax(-1)
            new.append(nxt)
            if step + 1 == max_new_tokens or (self.eos_token_id is not None and int(nxt) == self.eos_token_id):
                break
            attn = torch.cat([attn, attn.new_ones(attn.shape[0], 1)], 1)
            out = self.llm(inputs_embeds=table[nxt][:, None].to(embeds.dtype), attention_mask=attn, past_key_values=out.past_key_values,
                           use_cache=True, output_hidden_states=True)
            hs.append(out.hidden_states[-1])
        return torch.stack(new, 1), torch.cat(hs, 1)

    @torch.no_grad()
    def evaluate(self, images_clip, images, input_ids, resize_list, clip_resize_list, original_size_list, max_new_tokens=32,
                 tokenizer=None):
        """One image, one or more prompt rows.  Returns (all_output_ids, pred_masks, batch_seg_token_counts, mask_scores)."""
        all_pred, all_output_ids, counts = [], [], []
        emb_tokens = self.model.get_visual_emb_tokens(images)                         # computed once (the reference: after generation, :711)
        tokens_proj = self.model.out_mm_projector(emb_tokens[:1])
        for input_id in input_ids:
            if bool((input_id == 0).any()):                                           # strip the right padding (:621-625)
                input_id = input_id[: int(torch.where(input_id == 0)[0].min())]
            ids = input_id[None]
            attn, embeds, _, _ = self._llm_inputs(ids, None, None, tokens_proj)
            new, hidden = self._generate(embeds, attn, max_new_tokens)
            output_ids = torch.cat([ids, new], 1)
            all_output_ids.append(output_ids)
            mask = self._seg_token_mask(output_ids, pad_right=False)
            assert mask.shape[1] == hidden.shape[1], (mask.shape, hidden.shape)
            seg_hidden = hidden[mask]
            if seg_hidden.shape[0] % (self.seg_token_num * self.image_feature_scale_num) != 0:   # :661-663
                seg_hidden = seg_hidden[:0]
            pred = self.model.text_hidden_fcs[0](seg_hidden.to(BF16)) if seg_hidden.shape[0] else seg_hidden.new_zeros(0, 256, dtype=BF16)
            e = self._pack_queries(pred)
            all_pred.append(e)
            counts.append(e.shape[0])
        batch_seg_token_counts = [torch.tensor(counts, device=images.device, dtype=torch.int32)]
        pred_embeddings = [torch.cat(all_pred)]
        pred_masks, mask_scores = self.model.decode(emb_tokens[:1], pred_embeddings, resize_list[:1], original_size_list[:1])
        return all_output_ids, pred_masks, batch_seg_token_counts, mask_scores
