"""Top-level surface of the reference model, `walkgptForCausalLM` (/root/reference/model/walkgpt.py:155-746), over the HIP modules.

What callers of the reference touch (train_walkgpt.py:19,244-245,304; evaluation_walkgpt.py:18,233-248,297,305,569,673,916) and what
this adapter keeps:
  from_pretrained(version, torch_dtype=, low_cpu_mem_usage=, **model_args)   the 20 keyword arguments of evaluation_walkgpt.py:204-225
  .config                                   the language model's config with the fields walkgpt.py:174-236 writes into it
  get_model()                               -> the grounding module (WalkGPTGrounding): .config, .initialize_vision_modules(cfg),
                                            .get_vision_tower(), .initialize_walkgpt_modules(cfg), .mm_projector, .out_mm_projector,
                                            .visual_model, .text_hidden_fcs, .tiny_xattn
  resize_token_embeddings / enable_input_require_grads / gradient_checkpointing_enable     forwarded to the language model
  state_dict() / load_state_dict()          the reference's key layout: `model.layers.*`, `model.embed_tokens.*`, `lm_head.*` next to
                                            `model.visual_model.*`, `model.out_mm_projector.*`, ...
  forward(**kw)          -> model_forward unless `past_key_values` is passed (walkgpt.py:262-265)
  model_forward(...)     the collate_fn dict of utils/dataset.py:180-197, same argument names; returns the reference's dicts
                         (inference: pred_masks / gt_masks / batch_seg_token_counts / mask_scores, walkgpt.py:549-555; training: the six
                         loss entries, :598-605 -- forward values only, the HIP ops carry no autograd)
  generate(images=<tokens [rows,N,H] | pixels [B,3,h,w]>, input_ids=, attention_mask=, max_new_tokens=, num_beams=1,
           output_hidden_states=, return_dict_in_generate=, clip_resize_list=)          evaluation_walkgpt.py:569-577, walkgpt.py:629-639
  evaluate(...)          same signature and return tuple as walkgpt.py:607-746
  get_visual_embs(x)     walkgpt.py:241-258

The language model itself is NOT rebuilt here (stock PyTorch / transformers): `from_pretrained` creates a transformers
`LlamaForCausalLM` from a config (random init = BASELINE config C1) or a local checkpoint directory, or takes an already built module
(`llm=`) that speaks the causal-LM protocol -- `get_input_embeddings()`, `forward(inputs_embeds=, attention_mask=, labels=,
past_key_values=, use_cache=, output_hidden_states=)` returning `.logits`, `.loss`, `.hidden_states`, `.past_key_values`.
Around it everything runs on the HIP path: CLIP tower, SAM encoder, MSQP, token resample, the multimodal splice, CTP, prompt encoder,
mask decoder, postprocess, mask score, mask losses, InfoNCE.

Where the adapter differs from the RELEASED reference, and why:
  * `model_forward`'s decode loop (walkgpt.py:511-543) feeds the mask decoder `image_embeddings` rebuilt from the LLM-space image tokens
    (`output.image_features`, [1, rows, H_llm, 6, 6]) and raises at `src + dense_prompt_embeddings`.  The adapter decodes from SAM's
    256-channel embedding, the wiring of `evaluate()` (:713-737).  `decode_masks=False` is an option of the adapter (skip the decode,
    both lists stay empty), not a behaviour of the reference.
  * `evaluate()` sends `images_clip` through `generate(images=<pixels>)` -> `encode_images` pixel path (llava_arch.py:160-193) ->
    CLIP tower; the `mm_projector` call behind it is commented out (:246-249), so the released code splices 1024-wide CLIP features
    into H_llm-wide text embeddings and raises in `torch.cat` unless H_llm == 1024.  The adapter follows the same route and applies
    `mm_projector` whenever the widths differ (LLaVA's original order); with equal widths it skips it exactly as released.  Without a
    vision tower, or with `evaluate_visual_input="sam"`, the language model sees the MSQP tokens of the SAM embedding instead (the
    visual input of `model_forward` and of `generate_predictions_from_questions`, evaluation_walkgpt.py:443-475,569).
"""
import json
import os
from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .clip_encoder import llm_token_mask, patch_key_mask
from .llava_splice import IMAGE_TOKEN_INDEX, prepare_inputs_labels_for_multimodal
from .utils_walkgpt import TinyCrossAttn, infonce_loss
from .walkgpt import WalkGPTGrounding

BF16 = torch.bfloat16

# attributes of the reference's `walkgptModel` that are NOT language-model weights (walkgpt.py:59-146, llava_arch.py:33-42)
_GROUNDING_CHILDREN = ("visual_model", "out_mm_projector", "text_hidden_fcs", "tiny_xattn", "vision_tower", "mm_projector")

# walkgptForCausalLM.__init__ overrides these whatever the caller passed (walkgpt.py:174-181)
_FORCED_KWARGS = {"image_feature_scale_num": 1, "pad_train_clip_images": True, "resize_vision_tower": True,
                  "resize_vision_tower_size": 448, "vision_tower_for_mask": False, "separate_mm_projector": True}


def _build_language_model(version, torch_dtype, low_cpu_mem_usage):
    """`version` -> a transformers causal LM.  A config object / dict: random-init LlamaForCausalLM (BASELINE config C1 is a
    random-init LLaVA-7B).  A local directory: its config.json (LLaVA checkpoints say model_type "llava", a LLaMA with extra fields)
    and, when weight files are present, its weights.  Anything else would be a hub download: there is no network here."""
    from transformers import LlamaConfig, LlamaForCausalLM, PretrainedConfig
    ckpt_dir = None
    if isinstance(version, PretrainedConfig):
        cfg = version
    elif isinstance(version, dict):
        cfg = LlamaConfig(**version)
    elif isinstance(version, (str, os.PathLike)) and os.path.isdir(version):
        with open(os.path.join(version, "config.json")) as f:
            d = json.load(f)
        d.pop("model_type", None)
        d.pop("architectures", None)
        cfg = LlamaConfig(**d)
        if any(n.endswith((".safetensors", ".bin")) for n in os.listdir(version)):
            ckpt_dir = os.fspath(version)
    else:
        raise RuntimeError(
            "walkgptForCausalLM.from_pretrained(%r): not a config, not a local directory.  Hub downloads are not available in this "
            "build -- pass a transformers config (random init), a local checkpoint directory, or an already built model as llm=" % (version,))
    if ckpt_dir is not None:
        kw = {"torch_dtype": torch_dtype} if torch_dtype is not None else {}
        llm = LlamaForCausalLM.from_pretrained(ckpt_dir, config=cfg, low_cpu_mem_usage=bool(low_cpu_mem_usage), **kw)
    else:
        prev = torch.get_default_dtype()
        try:
            if torch_dtype is not None and torch_dtype.is_floating_point:
                torch.set_default_dtype(torch_dtype)     # a 7B random init in fp32 would need 27 GB of host memory
            llm = LlamaForCausalLM(cfg)
        finally:
            torch.set_default_dtype(prev)
    return llm, ckpt_dir


def _load_grounding_from_dir(model, ckpt_dir):
    """Grounding weights stored next to the language model in a merged checkpoint (`model.visual_model.*`, `model.mm_projector.*`, ...;
    merge_lora_weights_and_save_hf_model.py:178-185 drops only `vision_tower`): loaded non-strictly, key names as in Appendix A."""
    found = {}
    for name in sorted(os.listdir(ckpt_dir)):
        path = os.path.join(ckpt_dir, name)
        if name.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(path, framework="pt") as f:
                for k in f.keys():
                    if k.startswith("model.") and k.split(".")[1] in _GROUNDING_CHILDREN:
                        found[k[len("model."):]] = f.get_tensor(k)
        elif name.endswith(".bin"):
            sd = torch.load(path, map_location="cpu")
            for k, v in sd.items():
                if k.startswith("model.") and k.split(".")[1] in _GROUNDING_CHILDREN:
                    found[k[len("model."):]] = v
    if found:
        model.model.load_state_dict(found, strict=False)
    return sorted(found)


class walkgptForCausalLM(nn.Module):  # noqa: N801  (the reference's class name)
    def __init__(self, llm: nn.Module, grounding: Optional[WalkGPTGrounding] = None, **kwargs):
        super().__init__()
        self.llm = llm
        hidden = llm.get_input_embeddings().weight.shape[1]
        cfg = getattr(llm, "config", None)
        self.config = cfg if cfg is not None else SimpleNamespace(hidden_size=hidden)
        self._write_config(self.config, hidden, kwargs)
        if grounding is None:
            ref_p = next(llm.parameters())
            grounding = WalkGPTGrounding(sam=kwargs.get("sam", "vit_h"), llm_hidden=hidden, out_dim=kwargs.get("out_dim", 256),
                                         with_clip=kwargs.get("with_clip", False), clip_config=kwargs.get("clip_config"),
                                         config=self.config, vision_pretrained=kwargs.get("vision_pretrained"))
            if ref_p.dtype != torch.float32:
                grounding.to(ref_p.dtype)
        else:
            grounding.config = self.config
        self.model = grounding
        if not hasattr(self.model, "tiny_xattn"):      # walkgpt.py:104-113; created where the other grounding modules already live
            ref = next(self.model.visual_model.mask_decoder.parameters())
            self.model.tiny_xattn = TinyCrossAttn(kwargs.get("out_dim", 256)).to(device=ref.device, dtype=ref.dtype)
        if getattr(self.config, "separate_mm_projector", False) and not hasattr(self.model, "mm_projector") \
                and kwargs.get("with_mm_projector", kwargs.get("_from_pretrained", False)):
            # LlavaMetaModel.__init__ (llava_arch.py:36-42) with mm_projector_hidden_dim = 2 (walkgpt.py:190): Linear -> GELU -> Linear
            mm = getattr(self.config, "mm_hidden_size", 1024)
            ref = next(self.model.visual_model.mask_decoder.parameters())
            self.model.mm_projector = nn.Sequential(nn.Linear(mm, hidden * 2), nn.GELU(), nn.Linear(hidden * 2, hidden)).to(
                device=ref.device, dtype=ref.dtype)
        # walkgpt.py:163-215
        self.ce_loss_weight = kwargs.get("ce_loss_weight", 1.0)
        self.dice_loss_weight = kwargs.get("dice_loss_weight", 0.5)
        self.bce_loss_weight = kwargs.get("bce_loss_weight", 2.0)
        self.seg_token_idx = kwargs.get("seg_token_idx")
        self.seg_token_num = kwargs.get("seg_token_num", 1)
        self.image_feature_scale_num = _FORCED_KWARGS["image_feature_scale_num"] if kwargs.get("_from_pretrained") \
            else kwargs.get("image_feature_scale_num", 1)
        self.nce_tau = kwargs.get("nce_tau", 0.07)
        self.nce_topk = kwargs.get("nce_topk", 8)
        self.tokenizer = kwargs.get("tokenizer")
        self.logger = kwargs.get("logger")
        self.local_rank = kwargs.get("local_rank", 1)
        self._eos_token_id = kwargs.get("eos_token_id")
        self.evaluate_visual_input = kwargs.get("evaluate_visual_input", "auto")     # "auto" | "clip" | "sam" (module docstring)
        if self.evaluate_visual_input not in ("auto", "clip", "sam"):
            raise ValueError("evaluate_visual_input must be 'auto', 'clip' or 'sam'")
        self._register_state_dict_hook(self._to_reference_keys)
        self._register_load_state_dict_pre_hook(self._from_reference_keys)

    # -- configuration (walkgpt.py:174-236, :147-157) ------------------------------------------------------------------------------
    @staticmethod
    def _write_config(config, hidden, kwargs):
        forced = _FORCED_KWARGS if kwargs.get("_from_pretrained") else {}
        get = lambda k, d: forced.get(k, kwargs.get(k, d))   # noqa: E731
        defaults = dict(
            hidden_size=hidden, out_dim=kwargs.get("out_dim", 256), train_mask_decoder=kwargs.get("train_mask_decoder", False),
            resize_vision_tower=get("resize_vision_tower", True), resize_vision_tower_size=get("resize_vision_tower_size", 448),
            pad_train_clip_images=get("pad_train_clip_images", True), vision_tower_for_mask=get("vision_tower_for_mask", False),
            separate_mm_projector=get("separate_mm_projector", True), mm_projector_hidden_dim=2, mm_projector_out_dim=1,
            image_feature_scale_num=get("image_feature_scale_num", 1), mm_hidden_size=1024, mm_vision_select_layer=-2)
        written = ("resize_vision_tower", "resize_vision_tower_size", "pad_train_clip_images", "vision_tower_for_mask",
                   "separate_mm_projector", "mm_projector_hidden_dim", "mm_projector_out_dim", "image_feature_scale_num")
        for k, v in defaults.items():
            if k in written or not hasattr(config, k):
                setattr(config, k, v)
        if getattr(config, "vision_tower_for_mask", False):
            raise NotImplementedError("vision_tower_for_mask=True is not on the WalkGPT path (walkgpt.py:178 forces False)")
        if not hasattr(config, "mm_vision_tower"):
            config.mm_use_im_start_end = kwargs.get("use_mm_start_end", True)
            config.mm_vision_tower = kwargs.get("vision_tower", "openai/clip-vit-large-patch14")
        # walkgptModel.__init__ (:147-157)
        config.use_cache = False
        config.vision_tower = config.mm_vision_tower
        config.mm_vision_select_feature = "patch"
        config.image_aspect_ratio = "square"
        config.image_grid_pinpoints = None
        config.tune_mm_mlp_adapter = False
        config.freeze_mm_mlp_adapter = True
        config.pretrain_mm_mlp_adapter = None
        config.mm_use_im_patch_token = False

    @classmethod
    def from_pretrained(cls, version, *, llm=None, torch_dtype=None, low_cpu_mem_usage=False, **model_args):
        """evaluation_walkgpt.py:233-235 / train_walkgpt.py:237-243.  `model_args` = the keyword arguments of the reference's build_model
        (:204-225) plus, optionally, `sam` ("vit_b" | "vit_l" | "vit_h" | a geometry dict; the reference hard-codes vit_h, walkgpt.py:128),
        `clip_config` (architecture of the CLIP tower; ViT-L/14 by default) and `llm` (an already built causal LM)."""
        ckpt_dir = None
        if llm is None:
            llm, ckpt_dir = _build_language_model(version, torch_dtype, low_cpu_mem_usage)
        model_args = dict(model_args)
        if "seg_token_idx" not in model_args:
            raise KeyError("seg_token_idx")                # walkgpt.py:207 pops it without a default
        model_args.setdefault("sam", "vit_h")
        model = cls(llm, _from_pretrained=True, **model_args)
        if torch_dtype is not None:
            model.to(torch_dtype)
        if ckpt_dir is not None:
            model.loaded_grounding_keys = _load_grounding_from_dir(model, ckpt_dir)
        return model

    # -- state_dict keys in the reference's layout ---------------------------------------------------------------------------------------
    def _hf_layout(self):
        return hasattr(self.llm, "lm_head") and hasattr(self.llm, "model")

    @staticmethod
    def _to_reference_keys(module, state_dict, prefix, local_metadata):
        if not module._hf_layout():
            return state_dict
        for k in [k for k in state_dict if k.startswith(prefix + "llm.")]:
            state_dict[prefix + k[len(prefix) + 4:]] = state_dict.pop(k)      # llm.model.X -> model.X, llm.lm_head.X -> lm_head.X
        return state_dict

    def _from_reference_keys(self, state_dict, prefix, *args):
        if not self._hf_layout():
            return
        for k in list(state_dict):
            rest = k[len(prefix):]
            if rest.startswith("lm_head.") or (rest.startswith("model.") and rest.split(".")[1] not in _GROUNDING_CHILDREN):
                state_dict[prefix + "llm." + rest] = state_dict.pop(k)

    # -- accessors the reference's scripts use (evaluation_walkgpt.py:244-248,297,314,331; train_walkgpt.py:244-245) ---------------------
    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return self.model.get_vision_tower()

    def get_input_embeddings(self):
        return self.llm.get_input_embeddings()

    def resize_token_embeddings(self, new_num_tokens=None, **kw):
        if not hasattr(self.llm, "resize_token_embeddings"):
            raise RuntimeError("the injected language model has no resize_token_embeddings()")
        return self.llm.resize_token_embeddings(new_num_tokens, **kw)

    def enable_input_require_grads(self):
        return self.llm.enable_input_require_grads()

    def gradient_checkpointing_enable(self, *a, **kw):
        return self.llm.gradient_checkpointing_enable(*a, **kw)

    @property
    def eos_token_id(self):
        if self._eos_token_id is not None:
            return self._eos_token_id
        return getattr(self.config, "eos_token_id", None)

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

    def _queries_per_image(self, pred_embeddings, seg_token_counts, off, inference):
        """walkgpt.py:413-447: the gathered [SEG] rows (row-major over the text rows) -> one block per image (`off`: image -> rows;
        inference: every row belongs to the one image), each packed by _pack_queries.  Returns (blocks, batch_seg_token_counts)."""
        seg_token_offset = torch.cat([seg_token_counts.new_zeros(1), seg_token_counts.cumsum(-1)], 0)
        seg_off = [0, int(seg_token_offset[-1])] if inference else [int(seg_token_offset[o]) for o in off]
        blocks, counts = [], []
        for i in range(len(seg_off) - 1):
            e = self._pack_queries(pred_embeddings[seg_off[i]:seg_off[i + 1]])
            blocks.append(e)
            counts.append(e.shape[0])
        return blocks, counts

    def _embed_table(self):
        w = self.llm.get_input_embeddings().weight
        if w.dtype != BF16:
            raise RuntimeError("the multimodal splice reads embed_tokens.weight as bf16 (got %s): build / cast the language model in "
                               "bf16, the precision the reference's callers use (evaluation_walkgpt.py:227-231)" % w.dtype)
        return w

    def _llm_inputs(self, input_ids, attention_mask, labels, image_tokens, vit_attention_mask=None, train=False):
        """LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal on image tokens already in language space (llava_arch.py:252-259
        resample to 16x16, :265-518 splice).  train: the differentiable forms (gradients reach the image tokens and embed_tokens)."""
        if train:
            from . import autograd as ag
            return ag.splice(input_ids, attention_mask, labels, ag.resample_tokens(image_tokens, 16), self._embed_table(),
                             vit_attention_mask=vit_attention_mask)
        feats = ops.resample_tokens(image_tokens.contiguous(), 16)
        return prepare_inputs_labels_for_multimodal(input_ids, attention_mask, labels, feats, self._embed_table(),
                                                    vit_attention_mask=vit_attention_mask)

    # -- llava_arch.py:133-210 ---------------------------------------------------------------------------------------------------------
    def encode_images(self, images, clip_resize_list=None, return_project=False):
        """-> (image_features [B, N, H_llm], vit_attention_mask_for_llm [B, 256] or None, pre_image_features).
        3-D input: tokens already in language space (the bypass of :142-154).  4-D input: pixels through the CLIP tower with the
        patch mask built from `clip_resize_list` (:160-193); `mm_projector` is applied when the tower's width differs from H_llm
        (module docstring)."""
        if images is not None and images.dim() == 3:
            return images, None, ([images] if return_project else [])
        tower = self.get_vision_tower()
        if tower is None:
            raise RuntimeError("encode_images: pixel input needs a vision tower (initialize_vision_modules)")
        h, w = images.shape[-2:]
        sizes = [tuple(s) for s in clip_resize_list] if clip_resize_list is not None else [(h, w)] * images.shape[0]
        key_mask = patch_key_mask(images, sizes, tower.config.patch_size)
        llm_mask = llm_token_mask(key_mask, 16)
        feats, pre = self.model.encode_images_clip(images, sizes)
        hidden = self.llm.get_input_embeddings().weight.shape[1]
        if feats.shape[-1] != hidden:
            feats = self._mm_project(feats)
            if return_project:
                pre = [self._mm_project(f) for f in pre]
        return feats, llm_mask, ([feats] + list(pre) if return_project else list(pre))

    def _mm_project(self, feats):
        mp = getattr(self.model, "mm_projector", None)
        if mp is None:
            raise RuntimeError("CLIP features are %d wide, the language model %d, and there is no mm_projector to map them"
                               % (feats.shape[-1], self.llm.get_input_embeddings().weight.shape[1]))
        if isinstance(mp, nn.Linear):
            return ops.linear(feats.contiguous(), mp.weight, mp.bias)
        x = ops.linear(feats.contiguous(), mp[0].weight, mp[0].bias, act=ops.ACT_GELU)     # Linear -> GELU -> Linear (llava_arch.py:41)
        return ops.linear(x, mp[2].weight, mp[2].bias)

    # -- walkgpt.py:267-605 ------------------------------------------------------------------------------------------------------------
    head_training = None      # None: decided per call (below); True / False: forced through enable_head_training()

    def enable_head_training(self, on: Optional[bool] = True):
        """Training of the grounding head (train_walkgpt.py:347-350).  By default `model_forward(inference=False)` decides by itself: called
        with gradients enabled while ANY parameter of text_hidden_fcs / visual_model.mask_decoder / out_mm_projector / tiny_xattn or of the
        language model requires a gradient -- the state train_walkgpt.py:347-357 leaves the model in -- it runs CTP, the mask decoder,
        postprocess and the mask losses through walkgpt_amd.train_head (differentiable HIP operators), so the reference loop's
        `model.backward(loss)` (:756) fills `.grad` of text_hidden_fcs.*, visual_model.mask_decoder.* and -- through the [SEG] hidden
        states and the language-model loss -- of whatever the caller left trainable in the language model, of out_mm_projector.* (MSQP)
        and of embed_tokens (through the splice), and -- through the region-alignment InfoNCE term -- of tiny_xattn.wq / wk: every entry
        of train_walkgpt.py's trainable_list.  Frozen here as in the reference: SAM's image and prompt encoders, the vision tower.
        `enable_head_training(False)` is the opt-out (always the no-gradient forward), `True` forces the training path, `None`
        restores the automatic choice."""
        self.head_training = None if on is None else bool(on)
        return self

    def _wants_head_training(self):
        if self.head_training is not None:
            return self.head_training
        # The automatic choice: does any parameter the head path can reach ask for a gradient?  Lazy and ordered: the generator stops at the first
        # trainable parameter, and the modules every training recipe of the reference leaves trainable (text_hidden_fcs, then MSQP: train_walkgpt.py's
        # trainable_list) come first -- in a training step this reads ONE flag.  The full walk over the language model only happens when nothing at all
        # is trainable and the caller still asked for gradients (model_forward does not get here under no_grad or with inference=True).
        m = self.model
        mods = [getattr(m, "text_hidden_fcs", None), getattr(m, "out_mm_projector", None), getattr(m, "tiny_xattn", None),
                getattr(getattr(m, "visual_model", None), "mask_decoder", None), self.llm]
        return any(p.requires_grad for mod in mods if mod is not None for p in mod.parameters())

    def model_forward(self, images, images_clip, input_ids, labels, attention_masks, offset, masks_list: List[torch.Tensor],
                      label_list: List[torch.Tensor], resize_list: List[tuple], inference: bool = False, clip_resize_list=None,
                      decode_masks: bool = True, **kwargs):
        train = bool(not inference and torch.is_grad_enabled() and self._wants_head_training())
        if not train:
            with torch.no_grad():
                return self._model_forward(images, images_clip, input_ids, labels, attention_masks, offset, masks_list, label_list, resize_list,
                                           inference, clip_resize_list, decode_masks, False)
        return self._model_forward(images, images_clip, input_ids, labels, attention_masks, offset, masks_list, label_list, resize_list, inference,
                                   clip_resize_list, decode_masks, True)

    def _model_forward(self, images, images_clip, input_ids, labels, attention_masks, offset, masks_list, label_list, resize_list, inference,
                       clip_resize_list, decode_masks, train):
        from . import autograd as ag
        from . import train_head
        batch_size = images.shape[0]
        assert batch_size == len(offset) - 1
        off = [int(v) for v in offset]
        with torch.no_grad():                  # the frozen SAM encoder and the bookkeeping: no gradients in either mode
            seg_token_mask = self._seg_token_mask(input_ids, pad_right=True)
            if inference:
                assert images_clip.shape[0] == 1, "inference branch assumes one image"
            emb_tokens = self.model.get_visual_emb_tokens(images)                        # [B, hw, 256] channels-last rows
            row_img = torch.tensor([i for i in range(batch_size) for _ in range(off[i + 1] - off[i])], device=images.device)
            if inference:
                row_img = torch.zeros(input_ids.shape[0], dtype=torch.long, device=images.device)
            sam_tokens_256 = emb_tokens.index_select(0, row_img)
        # MSQP on all images at once (the reference calls it image by image: :364-378), resample, splice.  Training: through the
        # differentiable operators, so that out_mm_projector.* and embed_tokens (train_walkgpt.py:347-350) get their gradients
        proj = self.model.out_mm_projector
        with torch.enable_grad() if train else torch.no_grad():
            tokens_proj = train_head.msqp_forward(proj, emb_tokens) if train else proj(emb_tokens)                  # [B, 36, H]
            sam_tokens = tokens_proj.index_select(0, row_img)                             # one row of image tokens per text row
            attn, embeds, new_labels, _ = self._llm_inputs(input_ids, attention_masks, None if inference else labels, sam_tokens, train=train)

        output = self.llm(inputs_embeds=embeds, attention_mask=attn, labels=new_labels, output_hidden_states=True)
        last_hidden = output.hidden_states[-1]
        assert len(self.model.text_hidden_fcs) == 1
        # CTP is a per-token map: projecting the gathered [SEG] rows equals projecting the sequence and gathering (:406-409)
        assert seg_token_mask.shape[1] == last_hidden.shape[1], (seg_token_mask.shape, last_hidden.shape)
        seg_hidden = last_hidden[seg_token_mask]
        ctp = self.model.text_hidden_fcs[0]
        if seg_hidden.shape[0] == 0:
            pred_embeddings = seg_hidden.new_zeros(0, 256, dtype=BF16)
        else:
            pred_embeddings = train_head.ctp_forward(ctp, seg_hidden.to(BF16)) if train else ctp(seg_hidden.to(BF16))
        pred_embeddings_nce = pred_embeddings
        seg_token_counts = seg_token_mask.int().sum(-1)
        pred_list, batch_seg_token_counts = self._queries_per_image(pred_embeddings, seg_token_counts, off, inference)
        # region-alignment InfoNCE (:449-473)
        seg_row_ids = torch.repeat_interleave(torch.arange(sam_tokens_256.size(0), device=images.device), seg_token_counts)
        loss_nce = torch.zeros((), device=images.device)
        if seg_row_ids.numel() > 0 and not inference:
            if train:
                loss_nce = train_head.infonce_loss(pred_embeddings_nce, sam_tokens_256, seg_row_ids, self.model.tiny_xattn, temperature=self.nce_tau,
                                                   top_k=self.nce_topk, exclude_same_row=sam_tokens_256.size(0) > 1)
            else:
                loss_nce = infonce_loss(pred_embeddings_nce, sam_tokens_256, seg_row_ids, self.model.tiny_xattn, temperature=self.nce_tau,
                                        top_k=self.nce_topk, exclude_same_row=sam_tokens_256.size(0) > 1, normalize=True)
        pred_masks, mask_scores = [], []
        if decode_masks:   # from SAM's embedding, evaluate()'s wiring (:713-737); the released loop (:511-543) decodes from LLM tokens and raises
            sizes = [tuple(l.shape[-2:]) for l in label_list]
            if train:
                pred_masks = train_head.decode(self.model, emb_tokens, pred_list, resize_list, sizes)
            else:
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
        losses = ag.mask_losses if train else ops.mask_losses
        live = [(g, p) for g, p in zip(masks_list, pred_masks) if g.shape[0] > 0]
        if len(live) > 1 and len({tuple(g.shape) for g, _ in live}) == 1:
            # every image has the same number of masks T at the same size (the usual batch): sum_i [S_i / (T + 1e-8)] * T is ONE reduction over
            # all masks -- the per-image loop of walkgpt.py:565-583 costs ~12 launches per image forward and backward
            T = live[0][0].shape[0]
            stacked = getattr(pred_masks, "stacked", None)          # the training decode's own [sum T, H0, W0] tensor, if every image has masks
            if stacked is None or stacked.shape[0] != T * len(live):
                stacked = torch.cat([p for _, p in live], 0)
            bce, dice = losses(stacked.float().contiguous(), torch.cat([g for g, _ in live], 0).float().contiguous(), num_masks=T)
            mask_bce_loss, mask_dice_loss, num_masks = bce * T, dice * T, T * len(live)
        else:
            for gt_mask, pred_mask in live:
                bce, dice = losses(pred_mask.float().contiguous(), gt_mask.float().contiguous(), num_masks=gt_mask.shape[0])
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

    # -- generation (walkgpt.py:629-639, evaluation_walkgpt.py:569-577) ---------------------------------------------------------------
    def _generate(self, embeds, attn, max_new_tokens):
        """Greedy decoding through the language model's KV cache.  The reference generates with `config.use_cache = False`
        (walkgpt.py:149), i.e. re-runs the whole sequence per new token, and reads `outputs.hidden_states[-1]` = the last step's
        last-layer states over the whole sequence; the cache gives the same states at one token of work per step.
        Returns (new token ids [rows, n], last-layer hidden states of the prompt and of every generated token that was fed back:
        [rows, L0 + n - 1, H]).  Rows that have emitted EOS keep emitting the pad id."""
        table = self.llm.get_input_embeddings().weight
        eos = self.eos_token_id
        pad = getattr(self.config, "pad_token_id", None)
        pad = eos if pad is None else pad
        out = self.llm(inputs_embeds=embeds, attention_mask=attn, use_cache=True, output_hidden_states=True)
        hs, new = [out.hidden_states[-1]], []
        finished = torch.zeros(embeds.shape[0], dtype=torch.bool, device=embeds.device)
        for step in range(max_new_tokens):
            nxt = out.logits[:, -1].argmax(-1)
            if eos is not None:
                nxt = torch.where(finished, torch.full_like(nxt, pad), nxt)
                finished = finished | (nxt == eos)
            new.append(nxt)
            if step + 1 == max_new_tokens or (eos is not None and bool(finished.all())):
                break
            attn = torch.cat([attn, attn.new_ones(attn.shape[0], 1)], 1)
            out = self.llm(inputs_embeds=table[nxt][:, None].to(embeds.dtype), attention_mask=attn, past_key_values=out.past_key_values,
                           use_cache=True, output_hidden_states=True)
            hs.append(out.hidden_states[-1])
        return torch.stack(new, 1), torch.cat(hs, 1)

    @torch.no_grad()
    def generate(self, images=None, input_ids=None, attention_mask=None, max_new_tokens=32, num_beams=1, output_hidden_states=False,
                 return_dict_in_generate=False, clip_resize_list=None, **kwargs):
        """The slice of transformers' `generate` the reference uses: greedy (`num_beams=1`), visual input either projected tokens
        [rows, N, H_llm] (evaluation_walkgpt.py:569-577) or CLIP pixels [B, 3, h, w] with `clip_resize_list` (walkgpt.py:629-639).
        `.sequences` = the prompt ids (placeholder -200 kept, as callers expect: evaluation_walkgpt.py:693,596) followed by the new
        ids; `.hidden_states[-1]` = last-layer states over the whole spliced sequence (see _generate)."""
        if num_beams != 1 or kwargs.get("do_sample"):
            raise NotImplementedError("walkgpt_amd.generate: greedy decoding only (the reference passes num_beams=1)")
        if input_ids is None:
            raise ValueError("generate: input_ids is required")
        rows = input_ids.shape[0]
        given_mask = kwargs.pop("_llm_mask", None)      # evaluate(): the patch mask that belongs to already encoded pixels
        if images is None or not bool((input_ids == IMAGE_TOKEN_INDEX).any()):
            embeds = self.llm.get_input_embeddings().weight[input_ids]
            attn = attention_mask.bool() if attention_mask is not None else torch.ones_like(input_ids, dtype=torch.bool)
        else:
            feats, llm_mask, _ = self.encode_images(images, clip_resize_list)
            llm_mask = given_mask if llm_mask is None else llm_mask
            if feats.shape[0] == 1 and rows > 1:
                feats = feats.expand(rows, -1, -1)
                llm_mask = llm_mask.expand(rows, -1) if llm_mask is not None else None
            attn, embeds, _, _ = self._llm_inputs(input_ids, attention_mask, None, feats, vit_attention_mask=llm_mask)
        new, hidden = self._generate(embeds, attn, max_new_tokens)
        sequences = torch.cat([input_ids, new], 1)
        if not return_dict_in_generate:
            return sequences
        return SimpleNamespace(sequences=sequences, hidden_states=(hidden,) if output_hidden_states else None)

    # -- walkgpt.py:607-746 ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, images_clip, images, input_ids, resize_list, clip_resize_list, original_size_list, max_new_tokens=32,
                 tokenizer=None):
        """One image, one or more prompt rows.  Returns (all_output_ids, pred_masks, batch_seg_token_counts, mask_scores).
        Visual input of the language model: `images_clip` through the CLIP tower as in the reference (:629-639) when a vision tower
        exists (`evaluate_visual_input` "auto" / "clip"), else the MSQP tokens of the SAM embedding ("sam"); module docstring."""
        all_pred, all_output_ids, counts = [], [], []
        emb_tokens = self.model.get_visual_emb_tokens(images)                         # computed once (the reference: after generation, :711)
        use_clip = self.evaluate_visual_input == "clip" or (
            self.evaluate_visual_input == "auto" and images_clip is not None and self.get_vision_tower() is not None)
        if use_clip:
            if images_clip is None:
                raise ValueError("evaluate: evaluate_visual_input='clip' needs images_clip")
            visual, llm_mask, _ = self.encode_images(images_clip, clip_resize_list)   # once per image, not once per prompt row
        else:
            visual, llm_mask = self.model.out_mm_projector(emb_tokens[:1]), None
        for input_id in input_ids:
            if bool((input_id == 0).any()):                                           # strip the right padding (:621-625)
                input_id = input_id[: int(torch.where(input_id == 0)[0].min())]
            ids = input_id[None]
            out = self.generate(images=visual, input_ids=ids, max_new_tokens=max_new_tokens, num_beams=1, output_hidden_states=True,
                                return_dict_in_generate=True, clip_resize_list=clip_resize_list, _llm_mask=llm_mask)
            output_ids, hidden = out.sequences, out.hidden_states[-1]
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
