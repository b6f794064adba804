"""Generate tests/golden/*.npz by running the REFERENCE's own modules (imported from /root/reference, this
container only) on synthetic weights/inputs from walkgpt_amd/synth.py.

Only outputs (or strided slices of them) are stored; weights and inputs are regenerated from (seed, key, shape).
Nothing from the reference -- source, bytecode or weights -- is written into the repo.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case ...]

The CLIP case uses transformers (5.15 here) CLIPVisionModel as a stand-in for the pinned 4.31 the reference
expects (its own wrapper no longer imports); see oracle/clip.py for what that does and does not pin.
"""
import os
import sys
import types
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"

from tests.golden import cases  # noqa: E402
from walkgpt_amd import synth  # noqa: E402


def _import_reference():
    import transformers  # noqa: F401  (must be imported before the torchvision stub is registered)
    for n in ["torchvision", "torchvision.ops", "torchvision.ops.boxes", "torchvision.transforms",
              "torchvision.transforms.functional"]:
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["torchvision.ops.boxes"].batched_nms = None
    sys.modules["torchvision.ops.boxes"].box_area = None
    sys.modules["torchvision.transforms.functional"].resize = None
    sys.modules["torchvision.transforms.functional"].to_pil_image = None
    # the reference's `model` / `utils` are namespace packages (no __init__.py): this repo's drop-in aliases of the same names are regular
    # packages and would win wherever they sit on sys.path, so the repo root leaves the path while the reference is imported
    saved = list(sys.path)
    sys.path[:] = [REF] + [q for q in saved if os.path.abspath(q or ".") != ROOT]
    try:
        from model.segment_anything import modeling as sam_modeling
        from utils import utils_walkgpt
    finally:
        sys.path[:] = saved
    assert sam_modeling.__file__.startswith(REF) and utils_walkgpt.__file__.startswith(REF)
    return sam_modeling, utils_walkgpt


def _load(module, seed, prefix):
    sd = module.state_dict()
    new = {k: torch.from_numpy(synth.param(seed, prefix + k, tuple(v.shape))) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    module.eval()
    return module


def make_sam_encoder(name):
    sam_modeling, _ = _import_reference()
    c = cases.SAM_ENCODERS[name]
    from functools import partial
    enc = sam_modeling.ImageEncoderViT(
        img_size=c["img"], patch_size=c["patch"], embed_dim=c["embed_dim"], depth=c["depth"], num_heads=c["heads"],
        mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), qkv_bias=True, use_rel_pos=True,
        global_attn_indexes=c["global_idx"], window_size=c["window"], out_chans=c["out"])
    _load(enc, c["seed"], "image_encoder.")
    x = cases.sam_encoder_input(c)
    taps = {}
    with torch.no_grad():
        h = enc.patch_embed(x) + enc.pos_embed
        for i, blk in enumerate(enc.blocks):
            h = blk(h)
            if i in c["tap_blocks"]:
                taps["block%d" % i] = cases.tap_tokens(h).numpy()
        out = enc.neck(h.permute(0, 3, 1, 2))
        assert torch.equal(out, enc(x))
    taps["out"] = cases.tap_embedding(out).numpy()
    taps["out_stats"] = np.array([out.mean().item(), out.std().item(), out.abs().max().item()], np.float64)
    np.savez_compressed(os.path.join(HERE, "sam_encoder_%s.npz" % name), **taps)


def make_decoder(name):
    """Prompt encoder (text branch) + mask decoder + Sam.postprocess_masks of the reference, three ways:
      * fp32 (the parity target), with stage taps: queries / keys after each TwoWayAttentionBlock and after the final attention,
        the upscaled embedding, hyper_in;
      * the same modules cast with .bfloat16() and run on CPU in bf16 -- the precision the reference's own callers use
        (evaluation_walkgpt.py:908-910) -- stored as `*_bf16`: the calibration of how far "bf16 storage" alone moves the masks."""
    sam_modeling, _ = _import_reference()
    c = cases.DECODERS[name]
    g = c["grid"]

    def build():
        pe = sam_modeling.PromptEncoder(embed_dim=256, image_embedding_size=(g, g), input_image_size=(g * 16, g * 16), mask_in_chans=16)
        dec = sam_modeling.MaskDecoder(num_multimask_outputs=3, transformer=sam_modeling.TwoWayTransformer(
            depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8), transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256)
        w = cases.decoder_case_weights(c)
        for mod, prefix in ((pe, "prompt_encoder."), (dec, "mask_decoder.")):
            sd = mod.state_dict()
            new = {k: (w[prefix + k] if prefix + k in w else torch.from_numpy(synth.param(c["seed"], prefix + k, tuple(v.shape))))
                   for k, v in sd.items()}
            mod.load_state_dict(new, strict=True)
            mod.eval()
        return pe, dec

    holder = types.SimpleNamespace(image_encoder=types.SimpleNamespace(img_size=g * 16))
    emb, text = cases.decoder_inputs(c)
    out = {}
    pe, dec = build()
    taps = {}
    hooks = []
    for i, layer in enumerate(dec.transformer.layers):
        hooks.append(layer.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__("layer%d" % i, o)))
    hooks.append(dec.transformer.register_forward_hook(lambda m, a, o: taps.__setitem__("final", o)))
    hooks.append(dec.output_upscaling.register_forward_hook(lambda m, a, o: taps.__setitem__("up", o)))
    with torch.no_grad():
        sparse, dense = pe(points=None, boxes=None, masks=None, text_embeds=text)
        dpe = pe.get_dense_pe()
        masks, iou = dec(image_embeddings=emb, image_pe=dpe, sparse_prompt_embeddings=sparse,
                         dense_prompt_embeddings=dense, multimask_output=False)
        # Sam.postprocess_masks (sam.py:137-172) needs a whole Sam object only for image_encoder.img_size
        post = sam_modeling.Sam.postprocess_masks(holder, masks, input_size=c["input_size"], original_size=c["original_size"])
        hs = taps["final"][0]
        hyper = torch.stack([dec.output_hypernetworks_mlps[i](hs[:, 1 + i]) for i in range(4)], 1)
    for h in hooks:
        h.remove()
    out["dense_pe"] = dpe[0, ::8].numpy()
    out["masks"] = masks.numpy()
    out["iou"] = iou.numpy()
    out["post"] = post.numpy()
    for i in range(2):
        out["queries%d" % i] = taps["layer%d" % i][0].numpy()
        out["keys%d" % i] = cases.tap_keys(taps["layer%d" % i][1], g).numpy()
    out["queries_final"] = hs.numpy()
    out["upscaled"] = cases.tap_upscaled(taps["up"]).numpy()
    out["hyper_in"] = hyper.numpy()
    # the reference's own bf16 run
    pe16, dec16 = build()
    pe16.bfloat16()
    dec16.bfloat16()
    with torch.no_grad():
        sparse, dense = pe16(points=None, boxes=None, masks=None, text_embeds=text.bfloat16())
        sparse = sparse.to(torch.bfloat16)   # model/walkgpt.py:521: `sparse_embeddings.to(pred_embeddings[i].dtype)`
        masks16, iou16 = dec16(image_embeddings=emb.bfloat16(), image_pe=pe16.get_dense_pe(), sparse_prompt_embeddings=sparse,
                           dense_prompt_embeddings=dense, multimask_output=False)
        post16 = sam_modeling.Sam.postprocess_masks(holder, masks16, input_size=c["input_size"], original_size=c["original_size"])
    out["masks_bf16"] = masks16.float().numpy()
    out["iou_bf16"] = iou16.float().numpy()
    out["post_bf16"] = post16.float().numpy()
    a, b = post.numpy() > 0, post16.float().numpy() > 0
    print("  %s: reference bf16 vs fp32: mask rel L2 %.4f, pixel IoU %.5f, positive fraction %.3f" % (
        name, float(np.linalg.norm(masks16.float().numpy() - masks.numpy()) / np.linalg.norm(masks.numpy())),
        (a & b).sum() / max(1, (a | b).sum()), a.mean()))
    np.savez_compressed(os.path.join(HERE, "decoder_%s.npz" % name), **out)


def make_e2e(name):
    """The whole grounding path of the reference's own modules on one image: ImageEncoderViT -> CalibratedTextProjector on the [SEG]
    states -> PromptEncoder (text branch) -> MaskDecoder -> Sam.postprocess_masks, the wiring of evaluate() (walkgpt.py:713-737) --
    in fp32 and with every module cast to bf16 (what the reference's callers run)."""
    sam_modeling, uw = _import_reference()
    c = cases.E2ES[name]
    e = cases.SAM_ENCODERS[c["enc"]]
    g = e["img"] // e["patch"]
    w_enc, w_dec, w_ctp = cases.e2e_weights(c)
    x, hid = cases.e2e_inputs(c)

    def build():
        enc = sam_modeling.ImageEncoderViT(
            depth=e["depth"], embed_dim=e["embed_dim"], img_size=e["img"], mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
            num_heads=e["heads"], patch_size=e["patch"], qkv_bias=True, use_rel_pos=True, global_attn_indexes=list(e["global_idx"]),
            window_size=e["window"], out_chans=e["out"])
        pe = sam_modeling.PromptEncoder(embed_dim=256, image_embedding_size=(g, g), input_image_size=(e["img"], e["img"]), mask_in_chans=16)
        dec = sam_modeling.MaskDecoder(num_multimask_outputs=3, transformer=sam_modeling.TwoWayTransformer(
            depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8), transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256)
        ctp = uw.CalibratedTextProjector(in_dim=hid.shape[-1], out_dim=256, widen=2, use_residual=False)
        enc.load_state_dict({k[len("image_encoder."):]: v for k, v in w_enc.items()}, strict=True)
        for mod, prefix in ((pe, "prompt_encoder."), (dec, "mask_decoder.")):
            mod.load_state_dict({k: (w_dec[prefix + k] if prefix + k in w_dec else torch.from_numpy(synth.param(c["dec_seed"], prefix + k, tuple(v.shape))))
                                 for k, v in mod.state_dict().items()}, strict=True)
        ctp.load_state_dict(w_ctp, strict=True)
        return [m.eval() for m in (enc, pe, dec, ctp)]

    holder = types.SimpleNamespace(image_encoder=types.SimpleNamespace(img_size=e["img"]))

    def run(mods, dtype):
        enc, pe, dec, ctp = mods
        with torch.no_grad():
            emb = enc(x.to(dtype))
            pred = ctp(hid.to(dtype)[None])[0]                                        # [T, 256]  (walkgpt.py:664)
            sparse, dense = pe(points=None, boxes=None, masks=None, text_embeds=pred.unsqueeze(1))
            sparse = sparse.to(pred.dtype)                                             # :726
            masks, iou = dec(image_embeddings=emb[0].unsqueeze(0), image_pe=pe.get_dense_pe(), sparse_prompt_embeddings=sparse,
                             dense_prompt_embeddings=dense, multimask_output=False)
            post = sam_modeling.Sam.postprocess_masks(holder, masks, input_size=c["resize"], original_size=c["original"])
        return emb.float(), pred.float(), masks.float(), post.float()[:, 0]

    emb, pred, masks, post = run(build(), torch.float32)
    m16 = build()
    for m in m16:
        m.bfloat16()
    emb16, pred16, masks16, post16 = run(m16, torch.bfloat16)
    a, b = post.numpy() > 0, post16.numpy() > 0
    rms = float(post.pow(2).mean().sqrt())
    print("  %s: logits rms %.3f, |logit| < 2%% rms on %.5f of the pixels, positive %s; reference bf16 vs fp32: embedding %.4f, masks %.4f, pixel IoU %.5f"
          % (name, rms, float((post.abs() < 0.02 * rms).float().mean()), np.round(a.mean(axis=(1, 2)), 3),
             float((emb16 - emb).norm() / emb.norm()), float((masks16 - masks).norm() / masks.norm()), (a & b).sum() / max(1, (a | b).sum())))
    np.savez_compressed(os.path.join(HERE, "e2e_%s.npz" % name), emb=cases.tap_embedding(emb).numpy(), pred=pred.numpy(), masks=masks.numpy(),
                        post=post.numpy(), emb_bf16=cases.tap_embedding(emb16).numpy(), masks_bf16=masks16.numpy(), post_bf16=post16.numpy())


def make_projectors(name):
    _, uw = _import_reference()
    c = cases.PROJECTORS[name]
    msqp = uw.MultiScaleQFormerProjector(sam_dim=256, llama_dim=c["llama_dim"], target_square_side=6)
    ctp = uw.CalibratedTextProjector(in_dim=c["llama_dim"], out_dim=256)
    _load(msqp, c["seed"], "out_mm_projector.")
    _load(ctp, c["seed"], "text_hidden_fcs.0.")
    toks, hid = cases.projector_inputs(c)
    with torch.no_grad():
        a = msqp(toks)
        b = ctp(hid)
    np.savez_compressed(os.path.join(HERE, "projectors_%s.npz" % name), msqp=a.numpy(), ctp=b.numpy())


def make_clip(name):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    c = cases.CLIPS[name]
    cfg = CLIPVisionConfig(hidden_size=c["dim"], intermediate_size=4 * c["dim"], num_hidden_layers=c["layers"],
                           num_attention_heads=c["heads"], image_size=c["img"], patch_size=14, hidden_act="quick_gelu")
    cfg._attn_implementation = "eager"
    model = CLIPVisionModel(cfg).eval()
    vm = model.vision_model if hasattr(model, "vision_model") else model
    w = cases.clip_weights(c)  # `vision_model.*` names, position table already at the run size
    sd = vm.state_dict()
    new = {}
    for k, v in sd.items():
        if k.endswith("position_ids"):
            new[k] = v
            continue
        new[k] = w["vision_model." + k].reshape(v.shape)
    vm.load_state_dict(new, strict=True)
    x, key_mask = cases.clip_inputs(c)
    bias = ((1.0 - key_mask) * torch.finfo(torch.float32).min)[:, None, None, :]
    with torch.no_grad():
        h = vm.pre_layrnorm(vm.embeddings(x))
        states = [h]
        for layer in vm.encoder.layers:
            h = layer(h, attention_mask=bias)
            if isinstance(h, tuple):
                h = h[0]
            states.append(h)
    np.savez_compressed(os.path.join(HERE, "clip_%s.npz" % name),
                        sel=states[c["select_layer"]][:, 1:].numpy(), pre=states[-11][:, 1:].numpy(),
                        emb=states[0].numpy())


def make_clip_calib(name):
    """The CLIP stand-in twice: fp32 with the full-precision synthetic weights, and bf16 the way the reference's callers run the
    tower (`.bfloat16()` module, bf16 pixels; eager attention: fp32 softmax cast back, every layer output rounded to bf16).  The bf16
    run is the calibration: how far the reference's OWN deployment precision sits from fp32, on the features and per layer."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    c = cases.CLIP_CALIBS[name]
    cfg = CLIPVisionConfig(hidden_size=c["dim"], intermediate_size=4 * c["dim"], num_hidden_layers=c["layers"],
                           num_attention_heads=c["heads"], image_size=c["img"], patch_size=14, hidden_act="quick_gelu")
    cfg._attn_implementation = "eager"
    model = CLIPVisionModel(cfg).eval()
    vm = model.vision_model if hasattr(model, "vision_model") else model
    w = cases.clip_weights(c)
    vm.load_state_dict({k: (v if k.endswith("position_ids") else w["vision_model." + k].reshape(v.shape)) for k, v in vm.state_dict().items()},
                       strict=True)
    x, key_mask = cases.clip_calib_inputs(c)

    def run(vm_, x_, dtype):
        bias = ((1.0 - key_mask) * torch.finfo(dtype).min).to(dtype)[:, None, None, :]      # custom_clip.py:27-38
        with torch.no_grad():
            h = vm_.pre_layrnorm(vm_.embeddings(x_))
            states = [h]
            for layer in vm_.encoder.layers:
                h = layer(h, attention_mask=bias)
                h = h[0] if isinstance(h, tuple) else h
                states.append(h)
        return states

    s32 = run(vm, x, torch.float32)
    vm16 = vm.bfloat16()
    s16 = run(vm16, x.bfloat16(), torch.bfloat16)
    st, ts = c["stride"], c["tap_stride"]
    out = {"sel": s32[c["select_layer"]][:, 1::st].numpy(), "pre": s32[-11][:, 1::st].numpy(),
           "sel_bf16": s16[c["select_layer"]][:, 1::st].float().numpy(), "pre_bf16": s16[-11][:, 1::st].float().numpy()}
    for t in c["taps"]:
        out["h%d" % t] = s32[t][:, ::ts].numpy()
        out["h%d_bf16" % t] = s16[t][:, ::ts].float().numpy()
    np.savez_compressed(os.path.join(HERE, "clipcal_%s.npz" % name), **out)
    e = lambda a, b: float((a.float() - b).norm() / b.norm())   # noqa: E731
    print("   reference-bf16 vs fp32: sel %.4f pre %.4f; taps %s" % (
        e(s16[c["select_layer"]], s32[c["select_layer"]]), e(s16[-11], s32[-11]),
        " ".join("h%d %.4f" % (t, e(s16[t], s32[t])) for t in c["taps"])), flush=True)


def make_metrics(name):
    """intersectionAndUnionGPU (utils/utils.py:192-204) and the mask losses (utils/utils_walkgpt.py:76-120)."""
    _, uw = _import_reference()
    from utils.utils import intersectionAndUnionGPU
    c = cases.METRICS[name]
    pred, gt = cases.metric_inputs(c)
    inter, union, tgt = [], [], []
    for i in range(c["n"]):
        # float carriers of the integral class ids: torch.histc has no CPU kernel for int tensors (the reference calls this on
        # the GPU only); the function's comparisons and histograms are dtype-agnostic
        a, b, t = intersectionAndUnionGPU((pred[i] > 0).float().contiguous().clone(), gt[i].contiguous().clone(), 2, ignore_index=255)
        inter.append(a); union.append(b); tgt.append(t)
    tg = (gt == 1).float()
    np.savez_compressed(os.path.join(HERE, "metrics_%s.npz" % name), inter=torch.stack(inter).numpy(), union=torch.stack(union).numpy(),
                        target=torch.stack(tgt).numpy(), bce=uw.sigmoid_ce_loss(pred, tg, num_masks=c["n"]).numpy(),
                        dice=uw.dice_loss(pred, tg, num_masks=c["n"]).numpy())


def make_nce(name):
    """infonce_loss + TinyCrossAttn of the reference (utils/utils_walkgpt.py:8-73, 330-357)."""
    _, uw = _import_reference()
    c = cases.NCES[name]
    xa = uw.TinyCrossAttn(d=c["D"], bias=c["bias"])
    xa.load_state_dict(cases.nce_weights(c), strict=True)
    pred, tok, seg = cases.nce_inputs(c)
    with torch.no_grad():
        loss, aux = uw.infonce_loss(pred, tok, seg, xa, temperature=0.07, top_k=c["top_k"], exclude_same_row=c["exclude"],
                                    normalize=True, return_aux=True)
        v_x, a_x = xa(pred, tok[seg])
    np.savez_compressed(os.path.join(HERE, "nce_%s.npz" % name), loss=loss.numpy(), v_pos=aux["v_pos"].numpy(),
                        attn_w=aux["attn_w"].numpy(), logits=aux["logits"].numpy(), xattn_out=v_x.numpy(), xattn_attn=a_x.numpy())


def make_match(name):
    """The cost matrix of match_pred (utils/matcher.py:93-127) from the reference's own point_sample / batch_* functions at
    fixed points (match_pred itself draws them with torch.rand), and scipy's assignment on it."""
    _import_reference()
    from utils import matcher as rm
    from scipy.optimize import linear_sum_assignment
    c = cases.MATCHES[name]
    pred, tgt, pts = cases.match_inputs(c)
    pc = pts[None]
    y = rm.point_sample(tgt[:, None], pc.repeat(tgt.shape[0], 1, 1), align_corners=False).squeeze(1)
    x = rm.point_sample(pred[:, None], pc.repeat(pred.shape[0], 1, 1), align_corners=False).squeeze(1)
    C = rm.batch_sigmoid_ce_loss(x.float(), y.float()) + rm.batch_dice_loss(x.float(), y.float())
    r, cidx = linear_sum_assignment(C)
    np.savez_compressed(os.path.join(HERE, "match_%s.npz" % name), cost=C.numpy(), rows=r, cols=cidx)


def make_prep(name):
    """ResizeLongestSide.apply_image of the reference (transforms.py:27-36).  torchvision is not installed: its two functions
    the method calls are stood in by what they are for a PIL input -- to_pil_image(ndarray) = Image.fromarray and
    resize(pil, (h, w)) = pil.resize((w, h), BILINEAR) -- so the arithmetic is the installed Pillow's.  The normalise / pad of
    PAVE_dataset.py:115-121 (its class needs cv2) is the three torch lines below."""
    from PIL import Image
    import torch.nn.functional as F
    _import_reference()
    tvf = sys.modules["torchvision.transforms.functional"]
    tvf.to_pil_image = lambda a: Image.fromarray(a)
    tvf.resize = lambda im, size: im.resize((size[1], size[0]), Image.BILINEAR)
    import importlib
    tr = importlib.import_module("model.segment_anything.utils.transforms")
    tr.to_pil_image, tr.resize = tvf.to_pil_image, tvf.resize
    c = cases.PREPS[name]
    frame = cases.prep_frame(c)
    resized = tr.ResizeLongestSide(c["target"]).apply_image(frame)
    mean = torch.Tensor([97.17, 105.73, 108.16]).view(-1, 1, 1)
    std = torch.Tensor([53.05, 56.40, 61.93]).view(-1, 1, 1)
    x = (torch.from_numpy(resized).permute(2, 0, 1).contiguous().float() - mean) / std
    x = F.pad(x, (0, c["target"] - x.shape[-1], 0, c["target"] - x.shape[-2]))
    np.savez_compressed(os.path.join(HERE, "prep_%s.npz" % name), resized=resized, image=x.numpy())


def make_splice(name):
    """LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal of the reference (llava_arch.py:213-518), run unmodified.
    Its package does not import here (`custom_clip.py:2` wants transformers' CLIPVisionTransformer, gone from the
    installed 5.x, and `llava_llama.py:176` re-registers the "llava" config name that 5.x already owns), so llava_arch.py is
    loaded on its own: as a module of a synthetic package whose search path is the reference's `model/llava_walkgpt/model/`
    directory (its relative imports then resolve to the reference's own files), with an empty placeholder class standing where
    the removed transformers name was -- the splice never touches the vision tower's classes.  `self` is a small harness that
    supplies what the method reads: `encode_images` (returns the case's projector output), `get_model().embed_tokens`,
    `get_vision_tower()`, `config`, `device`."""
    arch = _load_ref_llava_module("llava_arch")
    c = cases.SPLICES[name]
    ids, mask, labels, feats, table, vit = cases.splice_inputs(c)

    class Harness:
        device = torch.device("cpu")
        config = types.SimpleNamespace(mm_use_im_start_end=False, tune_mm_mlp_adapter=False)

        def __init__(self):
            self._model = types.SimpleNamespace(embed_tokens=torch.nn.Embedding.from_pretrained(table, freeze=True))

        def get_vision_tower(self):
            return object()

        def get_model(self):
            return self._model

        def encode_images(self, images, clip_resize_list):
            return feats, vit, []

    with torch.no_grad():
        out = arch.LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal(
            Harness(), ids, mask, None, labels, torch.zeros(c["rows"], 3, 4, 4), None)
    feats_out, none_ids, attn, _pkv, embeds, new_labels = out
    assert none_ids is None and feats_out[0] is feats
    save = dict(attention_mask=attn.numpy(), inputs_embeds=embeds.numpy())
    if new_labels is not None:
        save["labels"] = new_labels.numpy()
    np.savez_compressed(os.path.join(HERE, "splice_%s.npz" % name), **save)


def _load_ref_walkgpt():
    """The reference's model/walkgpt.py, loaded outside its package.  Its imports of the LLaVA language model do not resolve here
    (SURVEY.md 8c), so the two names it takes from there are placeholder classes in a synthetic package whose search path is the
    reference's `model/` directory -- segment_anything and utils resolve to the reference's own files.  The placeholder
    `LlavaLlamaForCausalLM.forward` / `.generate` hand back what the harness object carries (`_lm_output`, `_gen_outputs`): the
    language model is outside the path; everything walkgpt.py itself does around it runs unmodified."""
    import importlib
    _import_reference()
    if "_ref_model.walkgpt" in sys.modules:
        return sys.modules["_ref_model.walkgpt"]

    class LlavaLlamaModel:
        pass

    class LlavaLlamaForCausalLM:
        def forward(self, **kw):
            self._lm_calls.append(kw)
            return self._lm_output

        def generate(self, **kw):
            self._gen_calls.append(kw)
            return self._gen_outputs.pop(0)

    pkg = types.ModuleType("_ref_model")
    pkg.__path__ = [os.path.join(REF, "model")]
    sys.modules["_ref_model"] = pkg
    for name in ("llava_walkgpt", "llava_walkgpt.model", "llava_walkgpt.model.language_model"):
        m = types.ModuleType("_ref_model." + name)
        m.__path__ = []
        sys.modules["_ref_model." + name] = m
    stub = types.ModuleType("_ref_model.llava_walkgpt.model.language_model.llava_llama")
    stub.LlavaLlamaForCausalLM, stub.LlavaLlamaModel = LlavaLlamaForCausalLM, LlavaLlamaModel
    sys.modules[stub.__name__] = stub
    saved = list(sys.path)
    sys.path[:] = [REF] + [q for q in saved if os.path.abspath(q or ".") != ROOT]     # `from utils.utils_walkgpt import ...`
    try:
        mod = importlib.import_module("_ref_model.walkgpt")
    finally:
        sys.path[:] = saved
    assert mod.__file__.startswith(REF)
    return mod


class _StopHere(Exception):
    pass


def _locals_of(exc, func_name):
    tb = exc.__traceback__
    while tb is not None:
        if tb.tb_frame.f_code.co_name == func_name:
            return tb.tb_frame.f_locals
        tb = tb.tb_next
    raise RuntimeError("frame %s not on the traceback" % func_name)


def make_segmask(name):
    """[SEG] bookkeeping of the reference's own walkgptForCausalLM.model_forward (walkgpt.py:284-306 mask, :406-447 gather and
    per-image packing) and evaluate (:645-707), run from its source on a harness object: the language model and the vision modules
    are stand-ins that return index-coded tensors (hidden[r, p, :] = 1000 r + p), so the rows `last_hidden_state[seg_token_mask]`
    gathers say which positions the reference reads; the run is stopped at the first prompt-encoder call (model_forward) / at
    get_visual_embs (evaluate) and the method's locals are read from the frame.  `.cuda()` is the identity here (no GPU)."""
    ref = _load_ref_walkgpt()
    c = cases.SEGMASKS[name]
    ids = cases.segmask_inputs(c)
    rows, L = ids.shape
    Hh, g = 8, 4
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self

    def coded(n_rows, length):
        r = torch.arange(n_rows)[:, None, None].float() * 1000.0
        p = torch.arange(length)[None, :, None].float()
        return (r + p).expand(n_rows, length, 256).contiguous()

    def stop(*a, **k):
        raise _StopHere()

    obj = object.__new__(ref.walkgptForCausalLM)
    obj.seg_token_idx = c["seg"]
    obj.seg_token_num = c.get("seg_token_num", 1)
    obj.image_feature_scale_num = 1
    obj.config = types.SimpleNamespace()
    obj._lm_calls, obj._gen_calls = [], []
    ident = lambda t: t   # noqa: E731   text_hidden_fcs[0]: the coded states pass through
    obj.model = types.SimpleNamespace(
        text_hidden_fcs=[ident], out_mm_projector=lambda t: torch.zeros(t.shape[0], 36, Hh), mm_projector=None,
        tiny_xattn=_load(ref.TinyCrossAttn(d=256), c["seed"], "tiny_xattn."),
        visual_model=types.SimpleNamespace(prompt_encoder=stop, image_encoder=lambda x: torch.zeros(x.shape[0], 256, g, g)))
    obj.get_visual_embs = (lambda x: torch.zeros(x.shape[0], 256, g, g)) if c["mode"] == "train" else stop
    out = {}
    try:
        if c["mode"] == "train":
            B = len(c["offset"]) - 1
            obj._lm_output = types.SimpleNamespace(hidden_states=[coded(rows, L + 255)], image_features=[torch.zeros(rows, 36, Hh)],
                                                   logits=None, loss=torch.zeros(()))
            try:
                with torch.no_grad():
                    ref.walkgptForCausalLM.model_forward(
                        obj, images=torch.zeros(B, 3, 8, 8), images_clip=torch.zeros(B, 3, 8, 8), input_ids=ids, labels=ids,
                        attention_masks=torch.ones_like(ids).bool(), offset=torch.tensor(c["offset"]), masks_list=[], label_list=[],
                        resize_list=[(8, 8)] * B, inference=False, clip_resize_list=[(8, 8)] * B)
                raise RuntimeError("model_forward returned: the harness should have stopped it at the prompt encoder")
            except _StopHere as e:
                loc = _locals_of(e, "model_forward")
        else:
            new = c["new"]
            outs = []
            for r in range(rows):
                row = ids[r]
                row = row[: int(torch.where(row == 0)[0].min())] if bool((row == 0).any()) else row
                seq = torch.cat([row, torch.tensor(new[r])])[None]
                outs.append(types.SimpleNamespace(sequences=seq, hidden_states=[None, coded(1, seq.shape[1] - 1 + 255) + 1000.0 * r]))
            obj._gen_outputs = outs
            try:
                with torch.no_grad():
                    ref.walkgptForCausalLM.evaluate(obj, torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8), ids, [(8, 8)], [(8, 8)], [(8, 8)],
                                                    max_new_tokens=8)
                raise RuntimeError("evaluate returned: the harness should have stopped it at get_visual_embs")
            except _StopHere as e:
                loc = _locals_of(e, "evaluate")
            out["gen_input_ids_len"] = np.array([int(k["input_ids"].shape[1]) for k in obj._gen_calls])
            out["output_ids_len"] = np.array([int(o.shape[1]) for o in loc["all_output_ids"]])
    finally:
        torch.Tensor.cuda = cuda
    out["seg_token_mask"] = loc["seg_token_mask"].numpy()                      # model_forward: all rows; evaluate: the last row's
    pe = loc["pred_embeddings"]
    out["batch_seg_token_counts"] = np.array([int(v) for v in (loc["batch_seg_token_counts"][0].tolist()
                                                               if c["mode"] == "eval" else loc["batch_seg_token_counts"])])
    out["gathered"] = np.concatenate([p[:, 0].numpy() for p in pe]) if len(pe) else np.zeros(0, np.float32)   # 1000 row + position
    out["gathered_split"] = np.array([p.shape[0] for p in pe])
    np.savez_compressed(os.path.join(HERE, "segmask_%s.npz" % name), **out)
    print("   %s: mask %s, counts %s, gathered %s" % (name, out["seg_token_mask"].shape, out["batch_seg_token_counts"].tolist(),
                                                      out["gathered"].astype(int).tolist()), flush=True)


def _load_ref_llava_module(name):
    """A module of the reference's model/llava_walkgpt/model/ directory, loaded outside its package (see make_splice)."""
    import importlib
    import transformers.models.clip.modeling_clip as mc
    if not hasattr(mc, "CLIPVisionTransformer"):
        mc.CLIPVisionTransformer = type("CLIPVisionTransformer", (torch.nn.Module,), {})
    if "_ref_llava_model" not in sys.modules:
        pkg = types.ModuleType("_ref_llava_model")
        pkg.__path__ = [os.path.join(REF, "model", "llava_walkgpt", "model")]
        sys.modules["_ref_llava_model"] = pkg
    return importlib.import_module("_ref_llava_model." + name)


def make_clipwrap(name):
    """What the reference adds around transformers' CLIP, run from its own source on synthetic inputs:
      * patch-mask construction: LlavaMetaForCausalLM.encode_images (llava_arch.py:133-210, pixel path) with a harness tower that
        records the `attention_mask` it is called with -> `tower_mask` [B, 1 + P*P] and the returned `llm_mask` [B, 256];
      * key-padding mask: custom_clip._expand_mask (custom_clip.py:27-38) on that tower mask -> additive [B, 1, L, L] (row 0 kept);
      * position-table resize: CLIPVisionTower.load_model (clip_encoder.py:23-59) with `from_pretrained` of the two transformers
        names it calls replaced by a synthetic tower (no network, no checkpoint): lines 38-55 run unmodified on a synthetic table."""
    arch = _load_ref_llava_module("llava_arch")
    cc = _load_ref_llava_module("multimodal_encoder.custom_clip")
    ce = _load_ref_llava_module("multimodal_encoder.clip_encoder")
    c = cases.CLIPWRAPS[name]
    B, S = len(c["sizes"]), c["image"]
    seen = {}

    class Tower:
        vision_tower = object.__new__(cc._CLIPVisionModel)   # isinstance() only: never initialised, never called

        def __call__(self, images, attention_mask=None):
            seen["mask"] = attention_mask
            return torch.zeros(images.shape[0], 1, 1), []

    tower = Tower()

    class Harness:
        def get_model(self):
            return types.SimpleNamespace(get_vision_tower=lambda: tower)

    _f, llm_mask, _p = arch.LlavaMetaForCausalLM.encode_images(Harness(), torch.zeros(B, 3, S, S), [tuple(s) for s in c["sizes"]])
    key = cc._expand_mask(seen["mask"], torch.float32, seen["mask"].shape[1])

    table = cases.clipwrap_table(c)
    emb = torch.nn.Module()
    emb.num_patches, emb.embed_dim, emb.patch_size, emb.image_size = c["old_side"] ** 2, c["dim"], 14, c["old_side"] * 14
    emb.position_embedding = torch.nn.Embedding.from_pretrained(table.clone(), freeze=True)
    fake = torch.nn.Module()
    fake.vision_model = torch.nn.Module()
    fake.vision_model.embeddings = emb
    ce.CLIPImageProcessor = types.SimpleNamespace(from_pretrained=lambda *a, **k: None)
    ce.CLIPVisionModel = types.SimpleNamespace(from_pretrained=lambda *a, **k: fake)
    args = types.SimpleNamespace(mm_vision_select_layer=-2, resize_vision_tower=True, resize_vision_tower_size=c["new_side"] * 14)
    with torch.no_grad():
        vt = ce.CLIPVisionTower("synthetic", args, delay_load=False)
    new_table = vt.vision_tower.vision_model.embeddings.position_embedding.weight.detach()
    np.savez_compressed(os.path.join(HERE, "clipwrap_%s.npz" % name), tower_mask=seen["mask"].numpy(), llm_mask=llm_mask.numpy(),
                        key_mask_row0=key[:, 0, 0, :].numpy(), table=new_table.numpy())


def make_state_dict_shapes(_name):
    """Key -> shape tables of the reference modules (data, not source): the de-facto checkpoint ABI."""
    import json
    sam_modeling, uw = _import_reference()
    from model.segment_anything import build_sam_vit_b
    with torch.device("meta"):
        fix = {
            "sam_vit_b": {k: list(v.shape) for k, v in build_sam_vit_b().state_dict().items()},
            "msqp_4096": {k: list(v.shape) for k, v in uw.MultiScaleQFormerProjector(256, 4096, target_square_side=6).state_dict().items()},
            "ctp_4096": {k: list(v.shape) for k, v in uw.CalibratedTextProjector(4096, 256).state_dict().items()},
        }
    with open(os.path.join(HERE, "state_dict_shapes.json"), "w") as f:
        json.dump(fix, f, indent=0, sort_keys=True)


ALL = {
    "state_dict_shapes": (make_state_dict_shapes, {"all": None}),
    "metrics": (make_metrics, cases.METRICS),
    "nce": (make_nce, cases.NCES),
    "match": (make_match, cases.MATCHES),
    "prep": (make_prep, cases.PREPS),
    "splice": (make_splice, cases.SPLICES),
    "clipwrap": (make_clipwrap, cases.CLIPWRAPS),
    "sam_encoder": (make_sam_encoder, cases.SAM_ENCODERS),
    "decoder": (make_decoder, cases.DECODERS),
    "projectors": (make_projectors, cases.PROJECTORS),
    "clip": (make_clip, cases.CLIPS),
    "clipcal": (make_clip_calib, cases.CLIP_CALIBS),
    "segmask": (make_segmask, cases.SEGMASKS),
    "e2e": (make_e2e, cases.E2ES),
}

if __name__ == "__main__":
    want = sys.argv[1:]
    torch.set_num_threads(8)
    for kind, (fn, table) in ALL.items():
        for name in table:
            tag = "%s:%s" % (kind, name)
            if want and tag not in want and kind not in want:
                continue
            print("generating", tag, flush=True)
            fn(name)
