"""Top-level surface (walkgptForCausalLM adapter, model/ and utils/ alias packages): the reference's import lines resolve, and
model_forward / evaluate take the reference's arguments and return its structures, checked against the oracle composition on a
synthetic collate dict (ragged [SEG] counts, offsets, the -200 image placeholder)."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.golden import cases

SEG, EOS, V, H = 60, 2, 64, 64


def test_reference_import_lines_resolve():
    from model.walkgpt import walkgptForCausalLM                      # train_walkgpt.py:19, evaluation_walkgpt.py:18
    from model.segment_anything import build_sam_vit_h, sam_model_registry   # model/walkgpt.py:15 of the reference
    from utils.utils_walkgpt import CalibratedTextProjector, MultiScaleQFormerProjector, dice_loss, infonce_loss, sigmoid_ce_loss
    from utils.matcher import match_pred
    import walkgpt_amd.causal_lm as cl
    assert walkgptForCausalLM is cl.walkgptForCausalLM and callable(build_sam_vit_h) and "vit_h" in sam_model_registry
    for name in ("forward", "model_forward", "evaluate", "get_visual_embs", "get_model", "get_vision_tower", "from_pretrained"):
        assert hasattr(walkgptForCausalLM, name), name
    import inspect
    sig = inspect.signature(walkgptForCausalLM.model_forward)
    for arg in ("images", "images_clip", "input_ids", "labels", "attention_masks", "offset", "masks_list", "label_list", "resize_list",
                "inference", "clip_resize_list"):                  # the collate_fn dict of utils/dataset.py:180-197
        assert arg in sig.parameters, arg
    assert list(inspect.signature(walkgptForCausalLM.evaluate).parameters)[1:] == [
        "images_clip", "images", "input_ids", "resize_list", "clip_resize_list", "original_size_list", "max_new_tokens", "tokenizer"]
    with pytest.raises(RuntimeError):
        walkgptForCausalLM.from_pretrained("some/checkpoint")         # the language model is injected, never built here


def _bare_adapter(**attrs):
    """walkgptForCausalLM without its modules: the [SEG] bookkeeping helpers are plain torch and run on the CPU."""
    from walkgpt_amd.causal_lm import walkgptForCausalLM
    m = walkgptForCausalLM.__new__(walkgptForCausalLM)
    nn.Module.__init__(m)
    for k, v in {"seg_token_num": 1, "image_feature_scale_num": 1, **attrs}.items():
        setattr(m, k, v)
    return m


@pytest.mark.parametrize("name", ["train_int", "train_list"])
def test_seg_token_bookkeeping_vs_reference_model_forward(name):
    """model/walkgpt.py:284-306 (mask), :406-447 (gather + per-image packing) -- against what the reference's OWN model_forward did on
    these rows (tests/golden/make_golden.py:make_segmask runs it from its source on a harness with index-coded hidden states)."""
    from oracle import splice as osplice
    c = cases.SEGMASKS[name]
    gold = cases.load("segmask_" + name)
    ids = cases.segmask_inputs(c)
    m = _bare_adapter(seg_token_idx=c["seg"], seg_token_num=c.get("seg_token_num", 1))
    mask = m._seg_token_mask(ids, pad_right=True)
    assert np.array_equal(mask.numpy(), gold["seg_token_mask"])
    segs = c["seg"] if isinstance(c["seg"], list) else [c["seg"]]
    assert np.array_equal(osplice.seg_token_mask(ids, segs).numpy(), gold["seg_token_mask"])
    rows, L = ids.shape
    coded = (torch.arange(rows)[:, None, None] * 1000.0 + torch.arange(L + 255)[None, :, None]).expand(rows, L + 255, 4)
    gathered = coded[mask]                                            # what last_hidden_state[seg_token_mask] picks (:409)
    blocks, counts = m._queries_per_image(gathered, mask.int().sum(-1), c["offset"], inference=False)
    assert counts == gold["batch_seg_token_counts"].tolist() and [b.shape[0] for b in blocks] == gold["gathered_split"].tolist()
    assert np.array_equal(torch.cat(blocks)[:, 0].numpy(), gold["gathered"])


def test_seg_token_bookkeeping_vs_reference_evaluate():
    """model/walkgpt.py:621-625 (padding strip), :645-659 (mask without the right pad), :661-707 (per-row packing) against the reference's
    own evaluate() on a harness whose generate() appends the case's ids."""
    c = cases.SEGMASKS["eval"]
    gold = cases.load("segmask_eval")
    ids = cases.segmask_inputs(c)
    m = _bare_adapter(seg_token_idx=c["seg"])
    picked, counts, lens = [], [], []
    for r in range(ids.shape[0]):
        row = ids[r]
        if bool((row == 0).any()):
            row = row[: int(torch.where(row == 0)[0].min())]
        lens.append(int(row.shape[0]))
        out_ids = torch.cat([row, torch.tensor(c["new"][r])])[None]
        mask = m._seg_token_mask(out_ids, pad_right=False)
        coded = (1000.0 * r + torch.arange(out_ids.shape[1] - 1 + 255)[None, :, None]).expand(1, -1, 4)
        e = m._pack_queries(coded[mask])
        picked.append(e[:, 0])
        counts.append(e.shape[0])
    assert lens == gold["gen_input_ids_len"].tolist()                 # what the reference handed to generate() after the strip
    assert np.array_equal(mask.numpy(), gold["seg_token_mask"])       # the last row's, as in the reference's frame
    assert counts == gold["batch_seg_token_counts"].tolist()
    assert np.array_equal(torch.cat(picked).numpy(), gold["gathered"])


class TinyLM(nn.Module):
    """Stand-in for the injected causal LM (transformers protocol): two causal self-attention layers; `script` adds a large bias
    towards a fixed token at every position so that greedy decoding is reproducible across precisions."""

    def __init__(self, script=None):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.embed_tokens = nn.Embedding(V, H)      # HF LLaMA's names: train_walkgpt.py:347-357 picks its trainable_list by substring
        self.wq = nn.ParameterList([nn.Parameter(torch.randn(H, H, generator=g) / 8) for _ in range(2)])
        self.wv = nn.ParameterList([nn.Parameter(torch.randn(H, H, generator=g) / 8) for _ in range(2)])
        self.lm_head = nn.Linear(H, V, bias=False)
        with torch.no_grad():
            self.embed.weight.copy_(torch.randn(V, H, generator=g))
            self.head.weight.copy_(torch.randn(V, H, generator=g) / 8)
        self.script = script or {}
        self.config = SimpleNamespace(eos_token_id=EOS)

    @property
    def embed(self):
        return self.embed_tokens

    @property
    def head(self):
        return self.lm_head

    def get_input_embeddings(self):
        return self.embed_tokens

    def resize_token_embeddings(self, new_num_tokens=None, **kw):
        assert new_num_tokens in (None, V), "the toy LM keeps its vocabulary"
        return self.embed_tokens

    def forward(self, inputs_embeds=None, attention_mask=None, labels=None, past_key_values=None, use_cache=False,
                output_hidden_states=False, **kw):
        x_new = inputs_embeds.float()
        x = torch.cat([past_key_values, x_new], 1) if past_key_values is not None else x_new
        L = x.shape[1]
        keep = torch.ones(x.shape[0], L, dtype=torch.bool, device=x.device) if attention_mask is None else attention_mask.bool()
        causal = torch.tril(torch.ones(L, L, dtype=torch.bool, device=x.device))[None] & keep[:, None, :]
        h = x
        for wq, wv in zip(self.wq, self.wv):
            a = (h @ wq.float()) @ h.transpose(1, 2) / math.sqrt(H)
            a = a.masked_fill(~causal, -1e30).softmax(-1)
            h = h + torch.tanh(a @ (h @ wv.float()))
        logits = h @ self.head.weight.float().t()
        for pos, tok in self.script.items():
            if pos < L:
                logits[:, pos, tok] += 1e4
        n = x_new.shape[1]
        loss = None
        if labels is not None:
            loss = F.cross_entropy(logits[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), ignore_index=-100)
        return SimpleNamespace(logits=logits[:, -n:], loss=loss, hidden_states=(h[:, -n:],), past_key_values=x)


def _build(dev, script=None):
    from walkgpt_amd.causal_lm import walkgptForCausalLM
    from walkgpt_amd.walkgpt import WalkGPTGrounding
    c = cases.SAM_ENCODERS["tiny"]
    g = WalkGPTGrounding(sam=dict(embed_dim=c["embed_dim"], depth=c["depth"], heads=c["heads"], global_idx=c["global_idx"], img=c["img"]),
                         llm_hidden=H, with_clip=False)
    w_enc, w_dec = cases.sam_encoder_weights(c), cases.decoder_weights(5)
    wm, wt = cases.projector_weights(dict(cases.PROJECTORS["h64"]))

    def load(mod, weights, prefix, strict=True):
        mod.load_state_dict({k[len(prefix):]: v for k, v in weights.items() if k.startswith(prefix)}, strict=strict)

    load(g.visual_model.image_encoder, w_enc, "image_encoder.")
    load(g.visual_model.prompt_encoder, w_dec, "prompt_encoder.", strict=False)
    load(g.visual_model.mask_decoder, w_dec, "mask_decoder.")
    load(g.out_mm_projector, {"x." + k: v for k, v in wm.items()}, "x.")
    load(g.text_hidden_fcs[0], {"x." + k: v for k, v in wt.items()}, "x.")
    g.to(dev).bfloat16()
    g.visual_model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix.data = \
        w_dec["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].to(dev)
    lm = TinyLM(script).to(dev)
    lm.embed.weight.data = lm.embed.weight.data.to(torch.bfloat16)     # the splice reads embed_tokens.weight as bf16
    m = walkgptForCausalLM(lm, grounding=g, seg_token_idx=SEG, seg_token_num=1, eos_token_id=EOS)
    weights = dict(w_enc=w_enc, w_dec=w_dec, wm=wm, wt=wt, c=c)
    return m, lm, weights


def _oracle_masks(weights, lm, x, ids_rows, row_img, hidden_fn, resize, orig, visual=None):
    """Oracle composition: SAM encoder -> MSQP -> resample -> splice -> TinyLM (fp32, CPU) -> CTP at the [SEG]-1 positions -> decode.
    visual = (features [n, N, H] fp32, vit mask [n, 256] or None): what the language model sees instead of the MSQP tokens."""
    from oracle import projectors as oproj
    from oracle import sam as osam
    from oracle import splice as osplice
    c = weights["c"]
    g = c["img"] // c["patch"]
    w_all = dict(weights["w_enc"])
    w_all.update(weights["w_dec"])
    lm32 = TinyLM(lm.script)
    lm32.load_state_dict({k: v.float().cpu() for k, v in lm.state_dict().items()})
    with torch.no_grad():
        emb = osam.image_encoder(w_all, x, dict(patch=c["patch"], depth=c["depth"], heads=c["heads"], global_idx=c["global_idx"], window=c["window"]))
        vit = None
        if visual is None:
            vis = oproj.msqp(weights["wm"], emb.flatten(2).transpose(1, 2))
        else:
            vis, vit = visual
            vit = vit[row_img] if vit is not None else None
        feats = oproj.resample_tokens(vis)[row_img]
        amask, embeds, _ = osplice.prepare_inputs_labels_for_multimodal(ids_rows, None, None, feats, lm32.embed.weight.float(), vit)
        hidden = lm32(inputs_embeds=embeds, attention_mask=amask, output_hidden_states=True).hidden_states[-1]
        mask = hidden_fn(ids_rows)
        dpe = osam.dense_pe(w_all, (g, g))
        out = []
        for i in sorted(set(row_img.tolist())):
            rows = [r for r in range(len(row_img)) if int(row_img[r]) == i]
            hs = torch.cat([hidden[r][mask[r]] for r in rows], 0)
            if hs.shape[0] == 0:
                out.append(torch.zeros(0, *orig[i]))
                continue
            pe = oproj.ctp(weights["wt"], hs)
            sparse, dense = osam.prompt_encoder_text(w_all, pe.reshape(-1, 1, 256), (g, g))
            masks, _ = osam.mask_decoder(w_all, emb[i:i + 1], dpe, sparse, dense)
            out.append(osam.postprocess_masks(masks, c["img"], resize[i], orig[i])[:, 0])
    return out


def _iou(a, b):
    a, b = a > 0, b > 0
    return float((a & b).sum()) / max(1, int((a | b).sum()))


@pytest.mark.gpu
def test_model_forward_collate_dict_vs_oracle(dev):
    from oracle import splice as osplice
    m, lm, weights = _build(dev)
    c = weights["c"]
    x = cases.sam_encoder_input(c)                                   # two images
    L = 12
    ids = torch.randint(3, 50, (3, L), generator=torch.Generator().manual_seed(9))
    ids[:, 1] = -200                                                 # the image placeholder of every row
    ids[0, 5] = SEG; ids[0, 9] = SEG                                 # image 0: rows 0, 1 -> 2 + 1 prompts
    ids[1, 7] = SEG
    ids[2, 4] = SEG; ids[2, 6] = SEG; ids[2, 10] = SEG               # image 1: row 2 -> 3 prompts
    offset = torch.tensor([0, 2, 3])
    resize, orig = [(512, 384), (400, 512)], [(200, 150), (75, 96)]
    batch = dict(images=x.to(dev, torch.bfloat16), images_clip=torch.zeros(2, 3, 28, 28, device=dev, dtype=torch.bfloat16),
                 input_ids=ids.to(dev), labels=ids.to(dev), attention_masks=torch.ones(3, L, dtype=torch.bool, device=dev), offset=offset.to(dev),
                 masks_list=[torch.zeros(3, *orig[0], device=dev), torch.zeros(3, *orig[1], device=dev)],
                 label_list=[torch.zeros(orig[0], device=dev), torch.zeros(orig[1], device=dev)], resize_list=resize,
                 clip_resize_list=[(28, 28)] * 2, inference=False, image_paths=["a", "b"], questions_list=None)   # extra keys are absorbed
    out = m(**batch)
    assert set(out) == {"loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "nce_loss", "mask_loss"}      # walkgpt.py:598-605
    assert all(torch.isfinite(v).all() for v in out.values())
    # the same forward in inference layout returns the reference's inference dict (one image, its rows)
    inf = dict(batch, images=batch["images"][:1], images_clip=batch["images_clip"][:1], input_ids=ids[:2].to(dev), labels=ids[:2].to(dev),
               attention_masks=batch["attention_masks"][:2], offset=torch.tensor([0, 2], device=dev), masks_list=batch["masks_list"][:1],
               label_list=batch["label_list"][:1], resize_list=resize[:1], clip_resize_list=[(28, 28)], inference=True)
    res = m(**inf)
    assert set(res) == {"pred_masks", "gt_masks", "batch_seg_token_counts", "mask_scores"}                   # walkgpt.py:549-555
    assert res["batch_seg_token_counts"] == [3] and res["pred_masks"][0].shape == (3,) + orig[0]
    ref = _oracle_masks(weights, lm, x[:1], ids[:2], torch.tensor([0, 0]), lambda r: osplice.seg_token_mask(r, [SEG]), resize[:1], orig[:1])
    got = res["pred_masks"][0].float().cpu()
    e = float((got - ref[0]).norm() / ref[0].norm())
    assert e < 0.06 and _iou(got.numpy(), ref[0].numpy()) > 0.97, e
    # an option of the adapter, not of the reference (whose own decode loop, walkgpt.py:511-543, takes LLM-space tokens and raises): skip the decode
    assert m(**dict(inf, decode_masks=False))["pred_masks"] == []


@pytest.mark.gpu
def test_model_forward_trains_the_grounding_head(dev):
    """train_walkgpt.py's step on the adapter: `enable_head_training()`, forward(**collate dict) with gradients on, `loss.backward()`.  The
    loss dict equals the no-gradient forward's; text_hidden_fcs.*, mask_decoder.* and the language model's parameters receive finite
    gradients, so do out_mm_projector.* and embed_tokens (through the splice); the frozen SAM encoders receive none; a few descent steps on
    the head lower the mask loss."""
    m, lm, weights = _build(dev)
    c = weights["c"]
    x = cases.sam_encoder_input(c)
    L = 12
    ids = torch.randint(3, 50, (3, L), generator=torch.Generator().manual_seed(9))
    ids[:, 1] = -200
    ids[0, 5] = SEG; ids[0, 9] = SEG
    ids[1, 7] = SEG
    ids[2, 4] = SEG; ids[2, 6] = SEG; ids[2, 10] = SEG
    offset = torch.tensor([0, 2, 3])
    resize, orig = [(512, 384), (400, 512)], [(200, 150), (75, 96)]
    gt = [(torch.rand(3, *o, generator=torch.Generator().manual_seed(i)) > 0.5).float().to(dev) for i, o in enumerate(orig)]
    batch = dict(images=x.to(dev, torch.bfloat16), images_clip=torch.zeros(2, 3, 28, 28, device=dev, dtype=torch.bfloat16),
                 input_ids=ids.to(dev), labels=ids.to(dev), attention_masks=torch.ones(3, L, dtype=torch.bool, device=dev), offset=offset.to(dev),
                 masks_list=gt, label_list=[torch.zeros(orig[0], device=dev), torch.zeros(orig[1], device=dev)], resize_list=resize,
                 clip_resize_list=[(28, 28)] * 2, inference=False)
    with torch.no_grad():
        base = m(**batch)                                               # the no-gradient forward
    assert not base["loss"].requires_grad
    m.enable_head_training(False)                                       # the opt-out: no graph even with gradients enabled
    assert not m(**batch)["loss"].requires_grad
    for p in m.model.visual_model.parameters():                        # model/walkgpt.py:83-91: SAM frozen, its mask decoder trainable
        p.requires_grad_(False)
    for p in m.model.visual_model.mask_decoder.parameters():
        p.requires_grad_(True)
    m.enable_head_training()
    out = m(**batch)
    assert out["loss"].requires_grad
    for k in ("ce_loss", "mask_bce_loss", "mask_dice_loss"):
        assert abs(float(out[k].detach()) - float(base[k])) < 2e-2 * abs(float(base[k])) + 1e-4, (k, float(out[k].detach()), float(base[k]))
    out["loss"].backward()
    head = [(k, p) for k, p in m.model.named_parameters() if k.startswith("text_hidden_fcs.") or ".mask_decoder." in k]
    got = [k for k, p in head if p.grad is not None]
    assert len(got) > 0.8 * len(head), (len(got), len(head))            # (all but the IoU head and the hypernetworks of the masks not returned)
    assert all(torch.isfinite(p.grad).all() for _, p in head if p.grad is not None)
    zero = [k for k, p in head if p.grad is not None and "k_proj.bias" not in k and float(p.grad.float().abs().max()) == 0]
    # exactly zero, as in the reference: the hypernetworks of the masks multimask_output=False does not return
    assert all(".output_hypernetworks_mlps." in k and ".output_hypernetworks_mlps.0." not in k for k in zero), zero
    assert lm.head.weight.grad is not None and lm.wq[0].grad is not None and torch.isfinite(lm.wq[0].grad).all()   # the LLM learns from both losses
    frozen = [p for k, p in m.model.named_parameters() if ".image_encoder." in k or ".prompt_encoder." in k]
    assert frozen and all(p.grad is None for p in frozen)
    msqp = [(k, p) for k, p in m.model.named_parameters() if k.startswith("out_mm_projector.")]
    assert msqp and all(p.grad is not None and torch.isfinite(p.grad).all() for _, p in msqp)        # the projector learns from the LM loss
    assert lm.embed.weight.grad is not None and float(lm.embed.weight.grad.float().abs().max()) > 0    # embed_tokens through the splice
    tx = m.model.tiny_xattn                                                                            # the InfoNCE term's own parameters
    assert tx.wq.weight.grad is not None and float(tx.wq.weight.grad.float().abs().max()) > 0 and torch.isfinite(tx.wk.weight.grad).all()
    assert abs(float(out["nce_loss"].detach()) - float(base["nce_loss"])) < 2e-2 * abs(float(base["nce_loss"])) + 1e-4
    # a few descent steps on the head only, fp32 master weights behind the bf16 parameters (a bf16 parameter does not register a step of a
    # fraction of a percent): the mask loss goes down
    before = float(out["mask_loss"].detach())
    train_p = [p for _, p in head if p.grad is not None and float(p.grad.float().abs().mean()) > 0]
    master = [p.detach().float().clone() for p in train_p]
    for _ in range(6):
        with torch.no_grad():
            for p, w32 in zip(train_p, master):
                g = p.grad.float()
                w32 -= 0.002 * w32.abs().mean() / (g.abs().mean() + 1e-30) * g
                p.copy_(w32.to(p.dtype))
                p.grad = None
        step = m(**batch)
        step["mask_loss"].backward()
    after = float(step["mask_loss"].detach())
    print("head training, 6 steps: mask loss %.5f -> %.5f" % (before, after))
    assert after < before


class _PeftStandIn(nn.Module):
    """What train_walkgpt.py:304 / evaluation_walkgpt.py:294 get back from `peft.get_peft_model` (peft is not installed here), reduced to
    what the scripts touch: the wrapped model sits at `.base_model.model` (so every state_dict key gains the prefix `base_model.model.`,
    evaluation_walkgpt.py:295-305), every base parameter is frozen, calls and unknown attributes are forwarded."""

    class _Tuner(nn.Module):
        def __init__(self, model):
            super().__init__()
            self.model = model

        def forward(self, *a, **k):
            return self.model(*a, **k)

    def __init__(self, model):
        super().__init__()
        for p in model.parameters():
            p.requires_grad_(False)
        self.base_model = _PeftStandIn._Tuner(model)

    def forward(self, *a, **k):
        return self.base_model(*a, **k)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.base_model.model, name)

    def print_trainable_parameters(self):
        n = sum(p.numel() for p in self.parameters() if p.requires_grad)
        print("trainable params: %d" % n)


class _EngineStandIn:
    """`deepspeed.initialize(...)[0]` reduced to what train_walkgpt.py:739-757 calls: the engine is callable, `backward(loss)` runs
    autograd, `step()` applies the optimizer (bf16 engine: fp32 master weights behind the bf16 parameters) and clears the gradients."""

    def __init__(self, model, model_parameters, lr):
        self.module = model
        self.params = [p for p in model_parameters if p.requires_grad]
        self.master = [p.detach().float().clone() for p in self.params]
        self.lr = lr

    def train(self):
        self.module.train()

    def __call__(self, **kw):
        return self.module(**kw)

    def backward(self, loss):
        loss.backward()

    def step(self):
        with torch.no_grad():
            for p, w32 in zip(self.params, self.master):
                if p.grad is not None:
                    g = p.grad.float()
                    w32 -= self.lr * w32.abs().mean() / (g.abs().mean() + 1e-30) * g     # scale-free step (toy optimiser)
                    p.copy_(w32.to(p.dtype))
                p.grad = None


def _unwrap_model_with_attr(model, attribute):
    """train_walkgpt.py:32-44, restated."""
    cur = model
    for _ in range(10):
        if hasattr(cur, attribute):
            return cur
        if hasattr(cur, "module"):
            cur = cur.module
        elif hasattr(cur, "base_model"):
            cur = cur.base_model
        else:
            break
    return cur


def test_head_training_is_decided_from_requires_grad():
    """The switch train_walkgpt.py's loop relies on, without a GPU: with nothing forced, the training path is chosen exactly when one of
    the reference's trainable modules (or the language model) holds a parameter that requires a gradient; True / False force it."""
    m = _bare_adapter()
    m.llm = nn.Linear(4, 4)
    m.model = nn.Module()
    m.model.text_hidden_fcs = nn.ModuleList([nn.Linear(4, 4)])
    m.model.out_mm_projector = nn.Linear(4, 4)
    m.model.visual_model = nn.Module()
    m.model.visual_model.mask_decoder = nn.Linear(4, 4)
    m.model.visual_model.image_encoder = nn.Linear(4, 4)
    assert m.head_training is None and m._wants_head_training()
    for p in m.parameters():
        p.requires_grad_(False)
    assert not m._wants_head_training()
    m.model.visual_model.image_encoder.weight.requires_grad_(True)      # not a module the path trains: still the no-gradient forward
    assert not m._wants_head_training()
    m.model.text_hidden_fcs[0].bias.requires_grad_(True)
    assert m._wants_head_training()
    assert not m.enable_head_training(False)._wants_head_training()
    for p in m.parameters():
        p.requires_grad_(False)
    assert m.enable_head_training(True)._wants_head_training()
    assert m.enable_head_training(None).head_training is None and not m._wants_head_training()
    # the PEFT stand-in's layout is the one the reference's scripts expect
    w = _PeftStandIn(m)
    assert w.base_model.model is m and all(k.startswith("base_model.model.") for k in w.state_dict())
    assert _unwrap_model_with_attr(w, "get_model") is not None and w.seg_token_num == 1


@pytest.mark.gpu
def test_train_walkgpt_loop_replayed_unchanged(dev):
    """train_walkgpt.py, call by call, on the adapter WITHOUT any added line: get_peft_model (:304, stand-in: wraps at .base_model.model
    and freezes the base), resize_token_embeddings (:307), the trainable_list loop (:347-357) over named_parameters(), deepspeed.initialize
    (:545, stand-in engine), model.train(), `output_dict = model(**input_dict)` (:739), the .item() reads, `model.backward(loss)` (:756),
    `model.step()` (:757).  The loss carries a graph by itself, the listed modules receive gradients, nothing else does, and the steps
    lower the loss."""
    m, lm, weights = _build(dev)
    c = weights["c"]
    model = _PeftStandIn(m)                                              # :304
    model.print_trainable_parameters()
    model.resize_token_embeddings(V)                                     # :307 (same size: a no-op on the toy LM)
    trainable_list = ["lm_head", "embed_tokens", "mask_decoder", "text_hidden_fcs", "mm_projector", "out_mm_projector"]   # :347-350
    picked = []
    for n, p in model.named_parameters():                                # :351-357
        if any(x in n for x in trainable_list):
            p.requires_grad = True
            picked.append(n)
    assert any("lm_head" in n for n in picked) and any("embed_tokens" in n for n in picked) and any("mask_decoder" in n for n in picked)
    assert all(n.startswith("base_model.model.") for n in picked)
    engine = _EngineStandIn(model, model.parameters(), lr=0.002)         # :545
    x = cases.sam_encoder_input(c)
    L = 12
    ids = torch.randint(3, 50, (3, L), generator=torch.Generator().manual_seed(9))
    ids[:, 1] = -200
    ids[0, 5] = SEG; ids[0, 9] = SEG
    ids[1, 7] = SEG
    ids[2, 4] = SEG; ids[2, 6] = SEG; ids[2, 10] = SEG
    resize, orig = [(512, 384), (400, 512)], [(200, 150), (75, 96)]
    gt = [(torch.rand(3, *o, generator=torch.Generator().manual_seed(i)) > 0.5).float().to(dev) for i, o in enumerate(orig)]
    input_dict = dict(images=x.to(dev), images_clip=torch.zeros(2, 3, 28, 28, device=dev), input_ids=ids.to(dev), labels=ids.to(dev),
                      attention_masks=torch.ones(3, L, dtype=torch.bool, device=dev), offset=torch.tensor([0, 2, 3], device=dev), masks_list=gt,
                      label_list=[torch.zeros(orig[0], device=dev), torch.zeros(orig[1], device=dev)], resize_list=resize,
                      clip_resize_list=[(28, 28)] * 2, inference=False, image_paths=["a", "b"], questions_list=None)
    engine.train()                                                       # :716
    losses = []
    for _ in range(5):
        input_dict["images"] = input_dict["images"].bfloat16()           # :734-735 (precision == "bf16")
        input_dict["images_clip"] = input_dict["images_clip"].bfloat16()
        output_dict = engine(**input_dict)                               # :739
        loss = output_dict["loss"]
        for k in ("ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss"):
            assert math.isfinite(output_dict[k].item())                  # :741-752
        assert output_dict.get("nce_loss") is not None and loss.requires_grad
        losses.append(loss.item())
        engine.backward(loss)                                            # :756
        if len(losses) == 1:
            named = dict(model.named_parameters())
            with_grad = {n for n, p in named.items() if p.grad is not None}
            assert all(named[n].requires_grad for n in with_grad)
            for x_ in ("lm_head", "embed_tokens", "text_hidden_fcs", "out_mm_projector", "mask_decoder.transformer"):
                assert any(x_ in n for n in with_grad), x_
            assert not any(".image_encoder." in n or ".prompt_encoder." in n or n.endswith(".wq.0") for n in with_grad)
        engine.step()                                                    # :757
    print("train_walkgpt.py loop replayed: loss %s" % ["%.4f" % v for v in losses])
    assert losses[-1] < losses[0]


@pytest.mark.gpu
def test_evaluate_signature_and_masks_vs_oracle(dev):
    from oracle import splice as osplice
    L0 = 8
    # the scripted continuation: positions are in the spliced sequence (L0 + 255 prompt positions), greedy picks script[pos] for pos+1
    script = {L0 + 254: 7, L0 + 255: SEG, L0 + 256: 9, L0 + 257: SEG, L0 + 258: EOS}
    m, lm, weights = _build(dev, script)
    c = weights["c"]
    x = cases.sam_encoder_input(c)[:1]
    ids = torch.randint(3, 50, (1, L0 + 3), generator=torch.Generator().manual_seed(11))
    ids[0, 1] = -200
    ids[0, L0:] = 0                                                  # right padding that evaluate() strips (:621-625)
    resize, orig = [(512, 384)], [(200, 150)]
    all_ids, pred_masks, counts, scores = m.evaluate(torch.zeros(1, 3, 28, 28, device=dev, dtype=torch.bfloat16), x.to(dev, torch.bfloat16),
                                                     ids.to(dev), resize, [(28, 28)], orig, max_new_tokens=16)
    assert len(all_ids) == 1 and all_ids[0][0, L0:].tolist() == [7, SEG, 9, SEG, EOS]
    assert counts[0].tolist() == [2] and pred_masks[0].shape == (2,) + orig[0] and scores[0].shape == (2,)

    def ev_mask(r):                                                   # evaluate()'s mask has no right pad (:651-659) and the last
        return osplice.seg_token_mask(r, [SEG])[:, :-1]              # generated token is never fed back: one position fewer

    out_ids = all_ids[0].cpu()
    ref = _oracle_masks(weights, lm, x, out_ids[:, :-1], torch.tensor([0]), lambda r: ev_mask(out_ids)[:, : r.shape[1] + 255], resize, orig)
    got = pred_masks[0].float().cpu()
    e = float((got - ref[0]).norm() / ref[0].norm())
    assert e < 0.06 and _iou(got.numpy(), ref[0].numpy()) > 0.97, e


def _attach_tiny_clip(m, dev):
    """The CLIP tower of the 'tiny' golden case behind the adapter, built the way the reference's build_model does it
    (evaluation_walkgpt.py:244-248), plus an mm_projector 128 -> H."""
    c = cases.CLIPS["tiny"]
    cfg = dict(hidden_size=c["dim"], intermediate_size=4 * c["dim"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
               image_size=c["img"], patch_size=14, layer_norm_eps=1e-5)
    args = SimpleNamespace(vision_tower="openai/clip-vit-large-patch14-336", mm_vision_select_layer=c["select_layer"], clip_config=cfg,
                           resize_vision_tower=True, resize_vision_tower_size=c["img"], pad_train_clip_images=True, pretrain_mm_mlp_adapter=None)
    g = m.get_model()
    g.initialize_vision_modules(args)
    w = cases.clip_weights(c)
    g.get_vision_tower().vision_tower.load_state_dict(w, strict=True)
    gen = torch.Generator().manual_seed(21)
    with torch.no_grad():
        g.mm_projector.weight.copy_(torch.randn(H, c["dim"], generator=gen) / c["dim"] ** 0.5)
        g.mm_projector.bias.copy_(torch.randn(H, generator=gen) * 0.1)
    g.get_vision_tower().to(dev).bfloat16()
    g.mm_projector.to(dev).bfloat16()
    return c, w


@pytest.mark.gpu
def test_evaluate_sends_images_clip_through_the_clip_tower(dev):
    """walkgpt.py:629-639 -> llava_arch.py:160-193 -> clip_encoder.py:71-98: with a vision tower present, evaluate()'s language model
    sees the CLIP features of `images_clip` (patch mask from clip_resize_list, mm_projector because the widths differ), not MSQP tokens."""
    from oracle import clip as oclip
    from oracle import splice as osplice
    L0 = 8
    P = 256                                                          # image positions in the spliced sequence
    script = {L0 + P - 2: 7, L0 + P - 1: SEG, L0 + P: 9, L0 + P + 1: SEG, L0 + P + 2: EOS}
    m, lm, weights = _build(dev, script)
    c, wclip = _attach_tiny_clip(m, dev)
    assert m.get_vision_tower() is not None
    xs = cases.sam_encoder_input(weights["c"])[:1]
    imgs = cases.clip_inputs(c)[0][1:2]                              # the padded image of the case
    sizes = [tuple(c["clip_resize_list"][1])]
    ids = torch.randint(3, 50, (1, L0), generator=torch.Generator().manual_seed(12))
    ids[0, 1] = -200
    resize, orig = [(512, 384)], [(200, 150)]
    all_ids, pred_masks, counts, scores = m.evaluate(imgs.to(dev, torch.bfloat16), xs.to(dev, torch.bfloat16), ids.to(dev), resize, sizes,
                                                     orig, max_new_tokens=16)
    assert all_ids[0][0, L0:].tolist() == [7, SEG, 9, SEG, EOS] and counts[0].tolist() == [2]
    # oracle: CLIP tower (fp32) -> mm_projector -> 16x16 resample + patch mask -> splice -> TinyLM -> CTP -> decode
    with torch.no_grad():
        km = oclip.patch_key_mask(1, (c["img"], c["img"]), sizes)
        feats, _ = oclip.clip_tower(wclip, imgs.to(torch.bfloat16).float(), km, select_layer=c["select_layer"], heads=c["heads"], layers=c["layers"])
        mp = m.get_model().mm_projector
        vis = feats @ mp.weight.float().cpu().t() + mp.bias.float().cpu()
        vit = oclip.llm_token_mask(km, 16)
    assert float(vit.min()) == 0.0                                    # the case does mask image tokens out of the LLM's attention
    out_ids = all_ids[0].cpu()
    ev_mask = osplice.seg_token_mask(out_ids, [SEG])[:, :-1]
    ref = _oracle_masks(weights, lm, xs, out_ids[:, :-1], torch.tensor([0]), lambda r: ev_mask[:, : r.shape[1] + 255], resize, orig,
                        visual=(vis, vit))
    got = pred_masks[0].float().cpu()
    e = float((got - ref[0]).norm() / ref[0].norm())
    assert e < 0.06 and _iou(got.numpy(), ref[0].numpy()) > 0.97, e
    # the other visual input, on request: MSQP tokens of the SAM embedding (what model_forward feeds the LM)
    m.evaluate_visual_input = "sam"
    _, pm2, _, _ = m.evaluate(imgs.to(dev, torch.bfloat16), xs.to(dev, torch.bfloat16), ids.to(dev), resize, sizes, orig, max_new_tokens=16)
    assert float((pm2[0].float().cpu() - got).abs().max()) > 1e-3    # a different visual input gives different [SEG] states


@pytest.mark.gpu
def test_generate_returns_the_structures_the_reference_reads(dev):
    """evaluation_walkgpt.py:569-596: generate(images=<projected tokens>, input_ids=, attention_mask=, max_new_tokens=, num_beams=1,
    return_dict_in_generate=True, clip_resize_list=) -> `.sequences` = prompt ids (with the -200 placeholder) + new ids, one row per prompt."""
    L0 = 7
    script = {L0 + 254: 5, L0 + 255: 6, L0 + 256: EOS}
    m, lm, weights = _build(dev, script)
    ids = torch.randint(3, 50, (2, L0), generator=torch.Generator().manual_seed(13))
    ids[:, 1] = -200
    toks = torch.randn(2, 36, H, generator=torch.Generator().manual_seed(14)).to(dev, torch.bfloat16)
    out = m.generate(images=toks, input_ids=ids.to(dev), attention_mask=torch.ones(2, L0, dtype=torch.bool, device=dev), max_new_tokens=8,
                     num_beams=1, return_dict_in_generate=True, output_hidden_states=True, clip_resize_list=[(28, 28)] * 2)
    assert out.sequences.shape == (2, L0 + 3) and out.sequences[:, :L0].cpu().equal(ids)
    assert out.sequences[:, L0:].tolist() == [[5, 6, EOS]] * 2
    assert out.hidden_states[-1].shape == (2, L0 + 255 + 2, H)       # prompt + image positions + the new tokens that were fed back
    seq = m.generate(images=toks, input_ids=ids.to(dev), max_new_tokens=2)
    assert torch.is_tensor(seq) and seq.shape == (2, L0 + 2)
    with pytest.raises(NotImplementedError):
        m.generate(images=toks, input_ids=ids.to(dev), num_beams=4)


@pytest.mark.gpu
def test_c1_shapes_through_from_pretrained(dev):
    """BASELINE config C1 at its real widths: one 448x448 image, SAM ViT-B, CLIP ViT-L/14, H_llm = 4096 projectors, built by the
    reference's build_model sequence -- with a 2-layer random-init LLaMA so that it fits a test."""
    from transformers import LlamaConfig
    from model.walkgpt import walkgptForCausalLM
    Hc, Vc, SEGc = 4096, 320, 300
    cfg = LlamaConfig(vocab_size=Vc, hidden_size=Hc, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=32,
                      num_key_value_heads=32, max_position_embeddings=2048, mm_hidden_size=1024)
    torch.manual_seed(0)
    model = walkgptForCausalLM.from_pretrained(cfg, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True, sam="vit_b", train_mask_decoder=False,
                                               out_dim=256, ce_loss_weight=1.0, dice_loss_weight=0.5, bce_loss_weight=2.0, seg_token_idx=SEGc,
                                               vision_pretrained=None, vision_tower="openai/clip-vit-large-patch14-336", use_mm_start_end=False,
                                               seg_token_num=1, logger=None, tokenizer=None, local_rank=0)
    model.config.eos_token_id, model.config.bos_token_id, model.config.pad_token_id = 2, 1, 0
    model.get_model().initialize_vision_modules(model.get_model().config)
    model.get_model().get_vision_tower().to(dtype=torch.bfloat16, device=dev)
    model.get_model().initialize_walkgpt_modules(model.get_model().config)
    model.resize_token_embeddings(Vc + 2)
    model.to(device=dev, dtype=torch.bfloat16)
    model.eval()
    vm = model.get_model().visual_model
    with torch.no_grad():                                            # zero-initialised by default (image_encoder.py:71-74,232-233)
        vm.image_encoder.pos_embed.normal_(0, 0.02)
    assert model.get_model().mm_projector[0].weight.shape == (2 * Hc, 1024) and model.get_model().out_mm_projector.to_llama.weight.shape == (Hc, 1024)
    g = torch.Generator().manual_seed(3)
    images = torch.randn(1, 3, 1024, 1024, generator=g).to(dev, torch.bfloat16)
    images_clip = torch.randn(1, 3, 448, 448, generator=g).to(dev, torch.bfloat16)
    L = 24
    ids = torch.randint(3, Vc, (2, L), generator=g)
    ids[:, 1] = -200
    ids[0, 9] = SEGc; ids[0, 15] = SEGc; ids[1, 20] = SEGc
    orig = (448, 448)
    res = model(images=images, images_clip=images_clip, input_ids=ids.to(dev), labels=ids.to(dev),
                attention_masks=torch.ones(2, L, dtype=torch.bool, device=dev), offset=torch.tensor([0, 2], device=dev),
                masks_list=[torch.zeros(3, *orig, device=dev)], label_list=[torch.zeros(orig, device=dev)], resize_list=[(1024, 1024)],
                clip_resize_list=[(448, 448)], inference=True)
    assert res["batch_seg_token_counts"] == [3] and res["pred_masks"][0].shape == (3,) + orig and res["mask_scores"][0].shape == (3,)
    assert torch.isfinite(res["pred_masks"][0]).all() and float(res["pred_masks"][0].float().std()) > 0
    out_ids, pred_masks, counts, scores = model.evaluate(images_clip, images, ids[:1].to(dev), [(1024, 1024)], [(448, 448)], [orig],
                                                         max_new_tokens=4)
    assert out_ids[0].shape[0] == 1 and L < out_ids[0].shape[1] <= L + 4 and out_ids[0][0, :L].cpu().equal(ids[0])
    assert pred_masks[0].shape[1:] == orig and pred_masks[0].shape[0] == int(counts[0][0]) and torch.isfinite(pred_masks[0]).all()


class PresetLM(nn.Module):
    """A language model whose last-layer states are given: what model_forward reads at the [SEG]-1 positions is under the test's control."""

    def __init__(self, vocab, hidden):
        super().__init__()
        self.embed = nn.Embedding(vocab, hidden)
        self.states = None
        self.config = SimpleNamespace(eos_token_id=EOS)

    def get_input_embeddings(self):
        return self.embed

    def forward(self, inputs_embeds=None, attention_mask=None, labels=None, output_hidden_states=False, **kw):
        assert self.states.shape[:2] == inputs_embeds.shape[:2], (self.states.shape, inputs_embeds.shape)
        return SimpleNamespace(logits=None, loss=None, hidden_states=(self.states,), past_key_values=None)


@pytest.mark.gpu
def test_model_forward_confident_masks_vs_reference(dev):
    """The end-to-end confident-mask fixture (tests/golden/e2e_conf_tiny.npz: the reference's own encoder -> CTP -> prompt encoder ->
    mask decoder -> postprocess in fp32) through the TOP-LEVEL surface: walkgptForCausalLM.forward(**collate dict, inference=True)
    with the [SEG] states of the fixture at the positions the reference reads (walkgpt.py:293-306,406-409).  north_star: 1e-3 mIoU."""
    from tests.test_gpu_modules import _e2e_model, _iou_vs_band_gt, pixel_iou
    from walkgpt_amd.causal_lm import walkgptForCausalLM
    c = cases.E2ES["conf_tiny"]
    gold = cases.load("e2e_conf_tiny")
    g = _e2e_model(c, dev)
    x, hid = cases.e2e_inputs(c)                                     # [1,3,S,S], [3, 64]
    Hh = hid.shape[1]
    lm = PresetLM(V, Hh).to(dev).bfloat16()
    m = walkgptForCausalLM(lm, grounding=g, seg_token_idx=SEG, seg_token_num=1)
    L = 12
    ids = torch.randint(3, 50, (2, L), generator=torch.Generator().manual_seed(17))
    ids[:, 1] = -200
    seg_pos = [(0, 4), (0, 9), (1, 6)]                               # row 0 carries two [SEG]s, row 1 one: three prompts of the one image
    states = torch.zeros(2, L + 255, Hh)
    for k, (r, p) in enumerate(seg_pos):
        ids[r, p] = SEG
        states[r, 255 + p - 1] = hid[k]
    lm.states = states.to(dev, torch.bfloat16)
    res = m(images=x.to(dev, torch.bfloat16), images_clip=torch.zeros(1, 3, 28, 28, device=dev, dtype=torch.bfloat16), input_ids=ids.to(dev),
            labels=ids.to(dev), attention_masks=torch.ones(2, L, dtype=torch.bool, device=dev), offset=torch.tensor([0, 2], device=dev),
            masks_list=[torch.zeros(3, *c["original"], device=dev)], label_list=[torch.zeros(c["original"], device=dev)],
            resize_list=[c["resize"]], clip_resize_list=[(28, 28)], inference=True)
    assert res["batch_seg_token_counts"] == [3]
    post = res["pred_masks"][0]
    pix = pixel_iou(post.cpu().numpy(), gold["post"])
    iou_h, iou_r = _iou_vs_band_gt(post, gold["post"], dev)
    assert pix >= 1.0 - 1e-3 and pix >= pixel_iou(gold["post_bf16"], gold["post"]), pix
    assert np.abs(iou_h - iou_r).max() <= 1e-3 and abs(iou_h.mean() - iou_r.mean()) <= 1e-3, (iou_h, iou_r)


@pytest.mark.gpu
def test_reference_validate_loop_on_the_adapter(dev):
    """The body of the reference's `validate()` (evaluation_walkgpt.py:877-982) on the adapter: `model(**input_dict)` with the collate
    dict in inference layout -> `pred_masks[0] > 0` against `gt_masks[0]` -> per-mask intersection / union (utils/utils.py:192-204; here
    wg_mask_iou_f32, thresholding fused) -> the loop's gIoU / cIoU accumulators; plus the text side of the loop,
    `generate(images=<projected SAM tokens>, ...)` as `generate_predictions_from_questions` calls it (:443-475, :569-577).  The numbers
    are held to what the same loop gives on the REFERENCE's masks of the confident end-to-end fixture (its modules in fp32)."""
    from tests.test_gpu_modules import _e2e_model
    from walkgpt_amd import ops
    from walkgpt_amd.causal_lm import walkgptForCausalLM
    c = cases.E2ES["conf_tiny"]
    gold = cases.load("e2e_conf_tiny")
    g = _e2e_model(c, dev)
    x, hid = cases.e2e_inputs(c)
    Hh, T = hid.shape[1], hid.shape[0]
    lm = PresetLM(V, Hh).to(dev).bfloat16()
    model = walkgptForCausalLM(lm, grounding=g, seg_token_idx=SEG, seg_token_num=1).eval()
    L = 12
    ref_post = torch.from_numpy(gold["post"])
    gt = (ref_post > 0).float()
    gt[:, 60:90, :] = 1.0 - gt[:, 60:90, :]                           # a ground truth the masks do not match exactly
    gt[:, :, :4] = 255.0                                              # an ignore band (utils/utils.py:197-199)

    def batch():
        ids = torch.randint(3, 50, (T, L), generator=torch.Generator().manual_seed(23))
        ids[:, 1] = -200
        states = torch.zeros(T, L + 255, Hh)
        for r in range(T):                                            # one [SEG] per prompt row, as PAVEValDataset builds them
            ids[r, 5 + r] = SEG
            states[r, 255 + 5 + r - 1] = hid[r]
        lm.states = states.to(dev, torch.bfloat16)
        return dict(images=x.to(dev), images_clip=torch.zeros(1, 3, 28, 28, device=dev), input_ids=ids.to(dev), labels=ids.to(dev),
                    attention_masks=torch.ones(T, L, dtype=torch.bool, device=dev), offset=torch.tensor([0, T], device=dev),
                    masks_list=[gt.to(dev)], label_list=[torch.zeros(c["original"], device=dev)], resize_list=[c["resize"]],
                    clip_resize_list=[(28, 28)], inference=True, image_paths=["synthetic.png"], questions_list=[["q"] * T])

    num_classes = 2
    acc = {k: torch.zeros(num_classes, dtype=torch.float64) for k in ("inter", "union", "giou_sum", "giou_count")}
    ref = {k: torch.zeros(num_classes, dtype=torch.float64) for k in acc}

    def accumulate(a, inter, union):                                  # evaluation_walkgpt.py:944-955
        for inter_i, union_i in zip(inter.double().cpu(), union.double().cpu()):
            a["inter"] += inter_i
            a["union"] += union_i
            gs = inter_i / (union_i + 1e-5)
            gs[union_i == 0] += 1.0
            a["giou_sum"] += gs
            a["giou_count"] += 1.0

    for _ in range(2):                                                # two "batches" of the loader
        input_dict = batch()
        input_dict["images"] = input_dict["images"].bfloat16()        # :908-910
        input_dict["images_clip"] = input_dict["images_clip"].bfloat16()
        with torch.no_grad():
            output_dict = model(**input_dict)
        pred_masks = output_dict["pred_masks"]
        assert len(pred_masks) == 1
        masks_list = output_dict["gt_masks"][0]
        i_h, u_h, _ = ops.mask_iou(pred_masks[0].float().contiguous(), masks_list.float().contiguous())
        accumulate(acc, i_h, u_h)
        i_r, u_r, _ = ops.mask_iou(ref_post.to(dev).contiguous(), masks_list.float().contiguous())
        accumulate(ref, i_r, u_r)
        # the text side of the loop: projected SAM tokens as the visual input of generate()
        toks = model.get_model().out_mm_projector(model.get_visual_embs(input_dict["images"]).flatten(2).transpose(1, 2).contiguous())
        lm.states = None
        lm.forward = lambda inputs_embeds=None, **kw: SimpleNamespace(                      # a language model that always answers EOS
            logits=torch.nn.functional.one_hot(torch.full(inputs_embeds.shape[:2], EOS, device=dev), V).float(), loss=None,
            hidden_states=(inputs_embeds,), past_key_values=None)
        out = model.generate(images=toks.expand(T, -1, -1).contiguous(), input_ids=input_dict["input_ids"],
                             attention_mask=input_dict["attention_masks"], max_new_tokens=8, num_beams=1, return_dict_in_generate=True,
                             clip_resize_list=[(28, 28)] * T)
        assert out.sequences.shape == (T, L + 1) and bool((out.sequences[:, -1] == EOS).all())
        del lm.forward
    giou = (acc["giou_sum"] / (acc["giou_count"] + 1e-10))[1].item()
    ciou = (acc["inter"] / (acc["union"] + 1e-10))[1].item()
    giou_r = (ref["giou_sum"] / (ref["giou_count"] + 1e-10))[1].item()
    ciou_r = (ref["inter"] / (ref["union"] + 1e-10))[1].item()
    print("validate loop on the adapter: gIoU %.5f cIoU %.5f; on the reference's masks: %.5f %.5f" % (giou, ciou, giou_r, ciou_r))
    assert 0.5 < giou_r < 0.999 and abs(giou - giou_r) <= 1e-3 and abs(ciou - ciou_r) <= 1e-3
