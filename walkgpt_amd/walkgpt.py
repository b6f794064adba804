"""Grounded-segmentation forward path of WalkGPT on the HIP modules.

Mirrors the vision/grounding half of /root/reference/model/walkgpt.py:
  * attribute names of walkgptMetaModel (`visual_model`, `out_mm_projector`, `text_hidden_fcs`, `vision_tower`) so
    the `model.*` state_dict keys of SURVEY.md Appendix A line up;
  * `get_visual_embs` (:241-258);
  * the decode wiring of `evaluate()` (:713-737): image embedding i -> prompt encoder (text branch) -> mask decoder ->
    Sam.postprocess_masks -> pred_mask[:, 0] and the mask score.  (The released `model_forward` feeds LLM tokens into
    the mask decoder and raises; SURVEY.md fact 3.  The wiring here is the one that runs.)
The language model between MSQP and CTP stays stock PyTorch and is not part of this module: callers hand over the
LLM hidden states at the [SEG] positions (`decode_from_hidden`), or already-projected prompt embeddings (`decode`).
"""
from types import SimpleNamespace
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from .clip_encoder import CLIPVisionTower, patch_key_mask
from .segment_anything import modeling as sam_modeling
from .utils_walkgpt import CalibratedTextProjector, MultiScaleQFormerProjector

BF16 = torch.bfloat16


class WalkGPTGrounding(nn.Module):
    """Vision + grounding modules of walkgptMetaModel.initialize_walkgpt_modules (walkgpt.py:59-146).

    sam: "vit_b" | "vit_l" | "vit_h" (the reference hard-codes vit_h, :128) or a dict of ImageEncoderViT geometry.
    llm_hidden: H of the language model (4096 for 7B).  clip_config: None -> ViT-L/14.
    """

    def __init__(self, sam="vit_h", llm_hidden=4096, out_dim=256, clip_config=None, clip_image_size=448,
                 select_layer=-2, with_clip=True, with_projectors=True, config=None, vision_pretrained=None):
        super().__init__()
        self._sam_spec, self._clip_config, self.vision_pretrained = sam, clip_config, vision_pretrained
        # the configuration object the reference's scripts reach through `model.get_model().config` (evaluation_walkgpt.py:244-248);
        # walkgptForCausalLM shares the language model's config here
        self.config = config if config is not None else SimpleNamespace(
            hidden_size=llm_hidden, out_dim=out_dim, train_mask_decoder=False, vision_tower_for_mask=False,
            mm_vision_select_layer=select_layer, resize_vision_tower=True, resize_vision_tower_size=clip_image_size,
            pad_train_clip_images=True)
        self.visual_model = self._build_visual_model()
        if with_projectors:
            self.out_mm_projector = MultiScaleQFormerProjector(256, llm_hidden, target_square_side=6)  # walkgpt.py:99-102
            self.text_hidden_fcs = nn.ModuleList([CalibratedTextProjector(llm_hidden, out_dim)])        # :115-123
        if with_clip:
            args = SimpleNamespace(mm_vision_select_layer=select_layer, pad_train_clip_images=True,
                                   resize_vision_tower=True, resize_vision_tower_size=clip_image_size)
            self.vision_tower = CLIPVisionTower("openai/clip-vit-large-patch14-336", args, config=clip_config)
        self.eval()

    def _build_visual_model(self):
        sam = self._sam_spec
        if isinstance(sam, str):
            return sam_modeling.sam_model_registry[sam](self.vision_pretrained)
        return sam_modeling._build_sam(sam["embed_dim"], sam["depth"], sam["heads"], list(sam["global_idx"]),
                                       image_size=sam.get("img", 1024))

    # -- construction-time surface the reference's build_model drives (evaluation_walkgpt.py:244-248) -----------------------------
    def get_vision_tower(self):
        """llava_arch.py:43-47."""
        vt = getattr(self, "vision_tower", None)
        return vt[0] if type(vt) is list else vt

    def initialize_vision_modules(self, model_args, fsdp=None):
        """LlavaMetaModel.initialize_vision_modules (llava_arch.py:49-86): build the CLIP tower named by `model_args.vision_tower`
        (architecture from `model_args.clip_config` or ViT-L/14 -- nothing is downloaded here; weights arrive by load_state_dict),
        record its width in the config and create `mm_projector` if the constructor has not."""
        name = getattr(model_args, "vision_tower", None) or getattr(model_args, "mm_vision_tower", None)
        if name is None:
            raise ValueError("initialize_vision_modules: model_args.vision_tower is not set")
        if not (name.startswith("openai") or name.startswith("laion") or "clip" in name):   # multimodal_encoder/builder.py:10-17
            raise ValueError(f"Unknown vision tower: {name}")
        self.config.mm_vision_tower = name
        args = SimpleNamespace(
            mm_vision_select_layer=getattr(model_args, "mm_vision_select_layer", -2),
            mm_vision_select_feature=getattr(model_args, "mm_vision_select_feature", "patch"),
            pad_train_clip_images=getattr(model_args, "pad_train_clip_images", True),
            resize_vision_tower=getattr(model_args, "resize_vision_tower", True),
            resize_vision_tower_size=getattr(model_args, "resize_vision_tower_size", 448))
        tower = CLIPVisionTower(name, args, config=getattr(model_args, "clip_config", None) or self._clip_config)
        if fsdp is not None and len(fsdp) > 0:
            self.__dict__["vision_tower"] = [tower]          # a plain list, as the reference keeps it out of the module tree
            self._modules.pop("vision_tower", None)
        else:
            self.vision_tower = tower
        self.config.use_mm_proj = True
        self.config.mm_hidden_size = tower.hidden_size
        self.config.mm_vision_select_layer = args.mm_vision_select_layer
        self.config.mm_vision_select_feature = args.mm_vision_select_feature
        if not hasattr(self, "mm_projector"):
            self.mm_projector = nn.Linear(self.config.mm_hidden_size, self.config.hidden_size)
        adapter = getattr(model_args, "pretrain_mm_mlp_adapter", None)
        if adapter is not None:
            weights = torch.load(adapter, map_location="cpu")
            self.mm_projector.load_state_dict({k.split("mm_projector.")[1]: v for k, v in weights.items() if "mm_projector" in k})

    def initialize_walkgpt_modules(self, config):
        """walkgptMetaModel.initialize_walkgpt_modules, the SAM branch (walkgpt.py:94-146; `vision_tower_for_mask` is forced off at
        :178): (re)create TinyCrossAttn, MSQP, the SAM model (from `vision_pretrained` when given) and CTP, with the reference's
        requires_grad pattern."""
        from .utils_walkgpt import TinyCrossAttn
        if getattr(config, "vision_tower_for_mask", False):
            raise NotImplementedError("vision_tower_for_mask=True (MaskDecoderMultiScale) is not on the WalkGPT path: "
                                      "walkgptForCausalLM.__init__ forces it to False (walkgpt.py:178)")
        H = config.hidden_size
        self.tiny_xattn = TinyCrossAttn(256)
        self.out_mm_projector = MultiScaleQFormerProjector(256, H, pad_to_square=True, target_square_side=6)
        self.visual_model = self._build_visual_model()
        for p in self.visual_model.parameters():
            p.requires_grad = False
        if getattr(config, "train_mask_decoder", False):
            self.visual_model.mask_decoder.train()
            for p in self.visual_model.mask_decoder.parameters():
                p.requires_grad = True
        self.text_hidden_fcs = nn.ModuleList([CalibratedTextProjector(H, getattr(config, "out_dim", 256))])
        self.text_hidden_fcs.train()
        for p in self.text_hidden_fcs.parameters():
            p.requires_grad = True
        self.__dict__.pop("_decode_graphs", None)      # captured decode graphs hold the old modules' addresses

    def set_gemm_dtype(self, dtype, clip=False):
        """"bf16" (default) or "fp8": operand type of the qkv / proj / MLP GEMMs of the SAM encoder blocks (BASELINE config C5: "hi-res
        SAM encoder ... fp8 MFMA"; needs block widths that are multiples of 128: ViT-B / L / H are).  The CLIP tower feeds the language
        model, whose logits carry the path's tightest tolerance, so it stays bf16 unless clip=True is asked for (an opt-in experiment:
        its selected features move 8.2 % from fp32 on e4m3 operands against 1.0 % in bf16, tests/test_gpu_fullsize.py).  Attention,
        LayerNorm statistics and the residual stream stay bf16 / fp32 either way."""
        if dtype not in ("bf16", "fp8"):
            raise ValueError("gemm dtype must be 'bf16' or 'fp8'")
        for blk in self.visual_model.image_encoder.blocks:
            blk.gemm_dtype = dtype
        tower = self.get_vision_tower()
        if tower is not None:
            for layer in tower.vision_tower.vision_model.encoder.layers:
                layer.gemm_dtype = dtype if clip else "bf16"

    # -- walkgpt.py:241-258 -----------------------------------------------------------------------------------------
    def get_visual_embs(self, pixel_values):
        return self.visual_model.image_encoder(pixel_values)

    def get_visual_emb_tokens(self, pixel_values, sub_batches=None):
        """Channels-last form of get_visual_embs: [B, h*w, 256] rows (what MSQP and the mask decoder consume).

        sub_batches (default: self.sam_sub_batches, 1): the encoder runs as that many independent slices of the batch, each on a HIP
        stream of its own (slice 0 on the current stream), all writing their rows of ONE output; the current stream waits for
        them.  Images are independent in the encoder, so the rows are those of the single pass, bit for bit.  Why: the encoder's
        kernels are persistent 256-workgroup launches with a partly filled last round (proj / lin2: 384 tiles = 1.5 rounds);
        two half-batch chains side by side fill each other's (and the CLIP stream's) idle CUs -- bench.py C2: +2.1 ... +3.6 %
        images/s at 2 slices, a loss at 4 (notes/r05_experiments.md section 5)."""
        enc = self.visual_model.image_encoder
        B = pixel_values.shape[0]
        n = int(sub_batches if sub_batches is not None else getattr(self, "sam_sub_batches", 1))
        if n <= 1 or B % n != 0 or not pixel_values.is_cuda:
            t = enc.forward_tokens(pixel_values)
            return t.view(B, -1, t.shape[-1])
        g = enc.img_size // enc.patch_size
        per = B // n
        out = torch.empty(B * g * g, enc.out_chans, device=pixel_values.device, dtype=torch.bfloat16)
        cur = torch.cuda.current_stream()
        streams = self.__dict__.setdefault("_sam_streams", [])
        while len(streams) < n - 1:
            streams.append(torch.cuda.Stream())
        # the encoder's derived operands (LayerNorm folds, re-laid weights, fp8 copies) are built lazily on whatever stream is current: build them
        # HERE, on the caller's stream, before the fork -- a slice that built them on its side stream would leave the other slices reading them
        # unordered (and out of that stream's allocator pool)
        enc.prepare()
        for k in range(1, n):
            sk = streams[k - 1]
            sk.wait_stream(cur)
            with torch.cuda.stream(sk):
                enc.forward_tokens(pixel_values[k * per:(k + 1) * per], out=out[k * per * g * g:(k + 1) * per * g * g])
        enc.forward_tokens(pixel_values[:per], out=out[:per * g * g])
        for sk in streams[:n - 1]:
            cur.wait_stream(sk)
            out.record_stream(sk)
            pixel_values.record_stream(sk)
        return out.view(B, g * g, enc.out_chans)

    # -- llava_arch.py:160-193 + clip_encoder.py:71-98 ---------------------------------------------------------------------
    def encode_images_clip(self, images_clip, clip_resize_list=None, tail_tiles=False):
        h, w = images_clip.shape[-2:]
        if clip_resize_list is None or all(tuple(s) == (h, w) for s in clip_resize_list):
            # nothing is padded: the key mask is all ones and its additive bias all zeros, so it is not passed at all
            # (decided from the python size list; no device sync)
            return self.vision_tower(images_clip, attention_mask=None, tail_tiles=tail_tiles)
        mask = patch_key_mask(images_clip, clip_resize_list)
        return self.vision_tower(images_clip, attention_mask=mask, tail_tiles=tail_tiles)

    # -- walkgpt.py:316-318 / 364-378 (batched instead of one call per image) --------------------------------------------------
    def project_visual_tokens(self, emb_tokens):
        return self.out_mm_projector(emb_tokens)

    # -- walkgpt.py:713-737 --------------------------------------------------------------------------------------------------------
    def decode(self, emb_tokens, pred_embeddings: Sequence[torch.Tensor], resize_list, original_size_list,
               multimask_output=False, prompt_tail=None) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        """emb_tokens [B, hw, 256]; pred_embeddings[i] [T_i, 256] -> (pred_masks[i] fp32 [T_i, H0, W0], mask_scores[i] [T_i]).

        The reference decodes image by image (walkgpt.py:716-737); here every prompt of every image goes through the
        prompt encoder / two-way decoder as ONE batch of P = sum(T_i) prompts (prompt p attends to the embedding of
        its own image), and only the size-dependent postprocess runs per image.  Same arithmetic per prompt.
        prompt_tail (CalibratedTextProjector.tail_operands()): pred_embeddings are the projector's rows BEFORE its tail
        (CalibratedTextProjector.pre_tail), which the decoder's first token launch applies -- one launch less, same bits."""
        vm = self.visual_model
        h, w = vm.prompt_encoder.image_embedding_size
        dev = emb_tokens.device
        pe = vm.prompt_encoder.dense_pe_tokens().unsqueeze(0)
        no_mask = vm.prompt_encoder.no_mask_embed.weight.reshape(1, -1)
        sl = (1, vm.mask_decoder.num_mask_tokens - 1) if multimask_output else (0, 1)
        counts = [int(e.shape[0]) for e in pred_embeddings]
        P = sum(counts)
        pred_masks = [None] * len(counts)
        mask_scores = [None] * len(counts)
        if P > 0:
            text = self._cat_rows(pred_embeddings)
            sparse, _ = vm.prompt_encoder(points=None, boxes=None, masks=None, text_embeds=text.unsqueeze(1))
            pimg = None
            if len(counts) == emb_tokens.shape[0] and all(c == 1 for c in counts):
                src = emb_tokens                                                # one prompt per image: no gather
            elif len([c for c in counts if c > 0]) == 1:
                i = next(k for k, c in enumerate(counts) if c > 0)
                src = emb_tokens[i:i + 1]                                       # one image: shared by its prompts
            else:
                src = emb_tokens       # one block per image; the decoder's first block reads it through the prompt -> image map
                pimg = self._prompt_image_index(tuple(counts), dev)
            # (+ the dense no-mask embedding, mask_decoder.py:136: folded into the decoder's first block instead of a pass over src)
            low_res, _iou = vm.mask_decoder.predict_masks_tokens(src, pe, sparse, h, w, sl, src_bias=no_mask, prompt_image=pimg,
                                                                 prompt_tail=prompt_tail)
            off = 0
            same = len(set(zip(map(tuple, resize_list), map(tuple, original_size_list)))) == 1
            if same and sl[1] == 1:
                # Sam.postprocess_masks + the mask score in one pass over the output
                full, scores = ops.postprocess_masks_scored(low_res, vm.image_encoder.img_size, resize_list[0], original_size_list[0])
                for i, c in enumerate(counts):
                    pred_masks[i], mask_scores[i] = full[off:off + c], scores[off:off + c]
                    off += c
            else:
                for i, c in enumerate(counts):
                    if c == 0:
                        continue
                    full = vm.postprocess_masks(low_res[off:off + c], input_size=resize_list[i], original_size=original_size_list[i])
                    m = full[:, 0].contiguous()
                    pred_masks[i], mask_scores[i] = m, ops.mask_score(m)
                    off += c
        for i, c in enumerate(counts):
            if c == 0:
                H0, W0 = original_size_list[i]
                pred_masks[i] = torch.zeros(0, H0, W0, device=dev)
                mask_scores[i] = torch.zeros(0, device=dev)
        return pred_masks, mask_scores

    @staticmethod
    def _cat_rows(parts):
        """torch.cat(parts, 0) for [T_i, D] row blocks -- without the copy when they already sit back to back in one buffer (the
        pieces torch.split hands out, the graph path's static input buffer)."""
        parts = [t for t in parts if t.shape[0] > 0]
        if len(parts) == 1:
            return parts[0]
        first, rows = parts[0], 0
        for t in parts:
            if not (t.is_contiguous() and t.dtype == first.dtype and t.device == first.device and t.shape[1:] == first.shape[1:]
                    and t.untyped_storage().data_ptr() == first.untyped_storage().data_ptr()
                    and t.storage_offset() == first.storage_offset() + rows * first.stride(0)):
                return torch.cat(parts, 0)
            rows += t.shape[0]
        return first.as_strided((rows,) + tuple(first.shape[1:]), first.stride(), first.storage_offset())

    def _prompt_image_index(self, counts, dev):
        """image index of every prompt, cached per count pattern (a host->device copy per call would also forbid graph capture)"""
        cache = self.__dict__.setdefault("_pidx_cache", {})
        key = (counts, str(dev))
        if key not in cache:
            cache[key] = torch.tensor([i for i, c in enumerate(counts) for _ in range(c)], device=dev, dtype=torch.int32)
        return cache[key]

    def decode_from_hidden_graphed(self, emb_tokens, seg_hidden: Sequence[torch.Tensor], resize_list, original_size_list):
        """decode_from_hidden() replayed from a captured HIP graph: the chain is ~75 small launches whose run time is mostly
        launch gaps (1.4 -> 0.8 ms per batch of 8 prompts).  One graph per (shapes, sizes) signature; inputs are copied into
        the graph's static buffers, and the returned tensors are the graph's output buffers -- valid until the next call with
        the same signature (clone them to keep them)."""
        # The captured graph bakes in the addresses of the weights AND of every derived operand the modules cache (re-laid ConvT
        # weights, PE-folded K/V tables, the dense-PE rows): its key therefore carries the identity and version of every parameter
        # and buffer the chain reads, so load_state_dict / .to() / an in-place edit after the first call re-captures instead of
        # replaying stale (or freed) memory.
        vm = self.visual_model
        mods = [vm.prompt_encoder, vm.mask_decoder] + (list(self.text_hidden_fcs) if hasattr(self, "text_hidden_fcs") else [])
        wkey = tuple((t.data_ptr(), t._version, t.dtype) for m in mods for t in list(m.parameters()) + list(m.buffers()))
        key = (tuple(emb_tokens.shape), tuple(tuple(h.shape) for h in seg_hidden), tuple(map(tuple, resize_list)),
               tuple(map(tuple, original_size_list)), str(emb_tokens.device), seg_hidden[0].dtype if seg_hidden else None)
        graphs = self.__dict__.setdefault("_decode_graphs", {})
        ent = graphs.get(key)
        if ent is not None and ent[4] != wkey:
            ent = None          # weights changed since capture: drop the stale graph (its buffers are released with it)
        if ent is None:
            s_emb = torch.empty_like(emb_tokens)
            s_all = torch.cat(list(seg_hidden), 0)        # one buffer for every image's [SEG] rows: one copy per replay, no cat inside
            s_hid = list(torch.split(s_all, [int(h.shape[0]) for h in seg_hidden], 0))
            s_emb.copy_(emb_tokens)
            # warm-up and capture on ONE stream of the graph's own (kept with it): first-use work (attribute settings, caches, the
            # per-stream ticket words of ops.postprocess_masks_scored) must not fall inside the capture, where an allocation + fill
            # would become a node of every replay
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap), torch.no_grad():
                for _ in range(2):
                    self.decode_from_hidden(s_emb, s_hid, resize_list, original_size_list)
            torch.cuda.current_stream().wait_stream(cap)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap), torch.no_grad():
                out = self.decode_from_hidden(s_emb, s_hid, resize_list, original_size_list)
            ent = graphs[key] = (g, s_emb, s_all, out, wkey, cap)
        g, s_emb, s_all, out = ent[:4]
        # staging copies into the graph's static inputs -- skipped for an input that IS the static buffer (callers that keep their
        # data in the buffers `decode_graph_inputs` hands out: the two copies are 9 of a one-image decode's ~220 microseconds)
        if emb_tokens.data_ptr() != s_emb.data_ptr():
            s_emb.copy_(emb_tokens)
        off, in_place = 0, True
        for h in seg_hidden:
            in_place = in_place and h.is_contiguous() and h.data_ptr() == s_all.data_ptr() + off * s_all.stride(0) * s_all.element_size()
            off += int(h.shape[0])
        if not in_place:
            torch.cat(list(seg_hidden), 0, out=s_all)
        g.replay()
        return out

    def decode_graph_inputs(self, emb_tokens, seg_hidden, resize_list, original_size_list):
        """The static input buffers of the captured decode graph for this signature (capturing it if needed), as
        (emb_tokens-like, [seg_hidden-like ...]) views: a caller that writes its embedding / [SEG] states straight into them and passes
        them back to decode_from_hidden_graphed replays without staging copies."""
        self.decode_from_hidden_graphed(emb_tokens, seg_hidden, resize_list, original_size_list)
        key = (tuple(emb_tokens.shape), tuple(tuple(h.shape) for h in seg_hidden), tuple(map(tuple, resize_list)),
               tuple(map(tuple, original_size_list)), str(emb_tokens.device), seg_hidden[0].dtype if seg_hidden else None)
        s_emb, s_all = self._decode_graphs[key][1:3]
        return s_emb, list(torch.split(s_all, [int(h.shape[0]) for h in seg_hidden], 0))

    def decode_from_hidden(self, emb_tokens, seg_hidden: Sequence[torch.Tensor], resize_list, original_size_list):
        """seg_hidden[i] [T_i, H_llm]: last-layer LLM states at the positions preceding each [SEG] (walkgpt.py:406-447;
        CTP is per token, so projecting only the gathered rows equals projecting the sequence and gathering)."""
        ctp = self.text_hidden_fcs[0]
        return self.decode(emb_tokens, self._project_seg_hidden(seg_hidden, tail=False), resize_list, original_size_list,
                           prompt_tail=ctp.tail_operands())

    def _project_seg_hidden(self, seg_hidden, tail=True):
        """CTP over all images' [SEG] rows in one call, split back per image (tail=False: without the projector's tail, for decode(prompt_tail=))."""
        counts = [int(h.shape[0]) for h in seg_hidden]
        if sum(counts) == 0:
            return [h.new_zeros(0, 256) for h in seg_hidden]
        ctp = self.text_hidden_fcs[0]
        rows = self._cat_rows(seg_hidden)
        pred = ctp(rows) if tail else ctp.pre_tail(rows)
        return list(torch.split(pred, counts, 0))

    @torch.no_grad()
    def forward(self, images, images_clip, seg_hidden, resize_list, original_size_list, clip_resize_list=None,
                overlap_streams=True):
        """The fused vision + grounding step the bench times: CLIP tower, SAM encoder, MSQP, CTP, decode, postprocess.
        Returns dict(pred_masks, mask_scores, clip_features, visual_tokens).

        The CLIP tower and the SAM branch are independent until the LLM; with overlap_streams they run on two HIP
        streams so each one's partially filled launches (M = B*1025 rows never tile evenly) use the other's idle CUs."""
        out = {}
        side = None
        if hasattr(self, "vision_tower") and images_clip is not None:
            cur = torch.cuda.current_stream()
            if overlap_streams:
                if getattr(self, "_side_stream", None) is None:
                    self._side_stream = torch.cuda.Stream()
                side = self._side_stream
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    out["clip_features"], out["clip_pre_features"] = self.encode_images_clip(images_clip, clip_resize_list)
            else:   # nothing shares the GPU with the tower: its M = B*1025 GEMMs may use the tail-absorbing tiles
                out["clip_features"], out["clip_pre_features"] = self.encode_images_clip(images_clip, clip_resize_list, tail_tiles=True)
        emb_tokens = self.get_visual_emb_tokens(images)
        if hasattr(self, "out_mm_projector"):
            out["visual_tokens"] = self.project_visual_tokens(emb_tokens)
            pred, tail = self._project_seg_hidden(seg_hidden, tail=False), self.text_hidden_fcs[0].tail_operands()
        else:
            pred, tail = list(seg_hidden), None  # already 256-d prompt embeddings
        out["pred_masks"], out["mask_scores"] = self.decode(emb_tokens, pred, resize_list, original_size_list, prompt_tail=tail)
        if side is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(side)
            for t in [out["clip_features"]] + list(out["clip_pre_features"]):
                t.record_stream(cur)
        return out
