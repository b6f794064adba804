"""GPU parity of the HIP-backed modules (reference module surface) against the golden vectors the REFERENCE's own
modules produced on the same synthetic weights (tests/golden/make_golden.py).  bf16 storage / fp32 accumulation
on the GPU vs fp32 on the reference: tolerances are relative L2 error over the tapped tensor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.golden import cases
from walkgpt_amd import ops
from walkgpt_amd.clip_encoder import CLIPVisionTower, patch_key_mask
from walkgpt_amd.segment_anything import modeling as M
from walkgpt_amd.utils_walkgpt import CalibratedTextProjector, MultiScaleQFormerProjector
from walkgpt_amd.walkgpt import WalkGPTGrounding


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def load_into(module, weights, prefix, dev, strict=True):
    sd = {k[len(prefix):]: v for k, v in weights.items() if k.startswith(prefix)}
    res = module.load_state_dict(sd, strict=strict)
    module.to(dev).bfloat16().eval()
    return res


def build_encoder(c, dev):
    from functools import partial
    enc = M.ImageEncoderViT(img_size=c["img"], patch_size=c["patch"], embed_dim=c["embed_dim"], depth=c["depth"],
                            num_heads=c["heads"], mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                            qkv_bias=True, use_rel_pos=True, global_attn_indexes=c["global_idx"], window_size=c["window"],
                            out_chans=c["out"])
    load_into(enc, cases.sam_encoder_weights(c), "image_encoder.", dev)
    return enc


@pytest.mark.parametrize("name,tol", [("tiny", 0.02), ("tiny_hd32", 0.02), ("hd80", 0.02), ("vit_b", 0.03), ("vit_h3", 0.02)])
def test_sam_encoder_vs_reference_golden(dev, name, tol):
    c = cases.SAM_ENCODERS[name]
    gold = cases.load("sam_encoder_" + name)
    enc = build_encoder(c, dev)
    x = cases.sam_encoder_input(c).to(dev, torch.bfloat16)
    with torch.no_grad():
        out = enc(x)
    assert out.shape == (c["batch"], c["out"], c["img"] // c["patch"], c["img"] // c["patch"])
    e = rel_err(cases.tap_embedding(out.float().cpu()).numpy(), gold["out"])
    assert e < tol, e
    # block taps: run the same rows path block by block
    with torch.no_grad():
        p = enc._prep_get(enc._build_prepared)
        g = c["img"] // c["patch"]
        t = ops.linear(ops.patchify(x, c["patch"]), p["patch_w"], enc.patch_embed.proj.bias, residual=p["pos"], res_row_mod=g * g)
        for i, blk in enumerate(enc.blocks):
            t = blk.rows(t, c["batch"], g)
            if "block%d" % i in gold:
                tap = cases.tap_tokens(t.float().cpu().view(c["batch"], g, g, -1)).numpy()
                assert rel_err(tap, gold["block%d" % i]) < tol, ("block", i)


def pixel_iou(a, b):
    a, b = np.asarray(a) > 0, np.asarray(b) > 0
    return float((a & b).sum()) / max(1, int((a | b).sum()))


@pytest.mark.parametrize("name", ["g32", "g64", "conf_g64"])
def test_decoder_and_postprocess_vs_reference_golden(dev, name):
    """Prompt encoder + mask decoder + postprocess against the reference's fp32 run, calibrated by the reference's own bf16 run
    (both in the fixture).  north_star's bar -- masks within 1e-3 mIoU of the reference PyTorch CPU path -- is asserted on the
    confident-mask case; on the white-noise cases (every pixel a boundary pixel) the HIP path must be at least as close to the fp32
    reference as the reference is to itself when its callers run it in bf16 (evaluation_walkgpt.py:908-910)."""
    c = cases.DECODERS[name]
    gold = cases.load("decoder_" + name)
    g = c["grid"]
    sam = M._build_sam(128, 1, 2, [0], image_size=g * 16)   # encoder unused here
    w = cases.decoder_case_weights(c)
    res = load_into(sam.prompt_encoder, w, "prompt_encoder.", dev, strict=False)
    assert all(("point_embeddings" in k or "not_a_point" in k or "mask_downscaling" in k) for k in res.missing_keys)
    load_into(sam.mask_decoder, w, "mask_decoder.", dev)
    sam.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix.data = \
        w["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].to(dev)  # keep the buffer fp32
    sam.to(dev)
    emb, text = cases.decoder_inputs(c)
    emb, text = emb.to(dev, torch.bfloat16), text.to(dev, torch.bfloat16)
    with torch.no_grad():
        dpe = sam.prompt_encoder.get_dense_pe()
        assert np.abs(dpe[0, ::8].float().cpu().numpy() - gold["dense_pe"]).max() < 2e-4
        sparse, dense = sam.prompt_encoder(points=None, boxes=None, masks=None, text_embeds=text)
        masks, iou = sam.mask_decoder(image_embeddings=emb, image_pe=dpe, sparse_prompt_embeddings=sparse,
                                      dense_prompt_embeddings=dense, multimask_output=False)
        post = sam.postprocess_masks(masks, input_size=c["input_size"], original_size=c["original_size"])
        # stage taps of the two-way transformer (same code path, block by block)
        md = sam.mask_decoder
        p = md._prep_get(md._build_prepared)
        P = text.shape[0]
        src = ops.add_rows(ops.nchw_to_tokens(emb.contiguous()), sam.prompt_encoder.no_mask_embed.weight.reshape(1, -1))
        pe_tok = sam.prompt_encoder.dense_pe_tokens().unsqueeze(0)
        tokens = torch.cat([p["out_tokens_f32"].unsqueeze(0).expand(P, -1, -1), sparse.float()], 1).contiguous()
        q, keys = tokens.clone(), src
        taps = {}
        for i, layer in enumerate(md.transformer.layers):
            q, keys = layer.run(q, keys, tokens, pe_tok, P)
            taps["queries%d" % i] = q.clone().cpu().numpy()
            taps["keys%d" % i] = cases.tap_keys(keys.float().cpu(), g).numpy()
    assert masks.shape == gold["masks"].shape and post.shape == gold["post"].shape
    e_masks, e_post = rel_err(masks.cpu().numpy(), gold["masks"]), rel_err(post.cpu().numpy(), gold["post"])
    ref16_masks = rel_err(gold["masks_bf16"], gold["masks"])
    iou_hip, iou_ref16 = pixel_iou(post.cpu().numpy(), gold["post"]), pixel_iou(gold["post_bf16"], gold["post"])
    print("decoder %s: rel_err masks %.4f (reference bf16 run: %.4f) iou-head %.4f post %.4f | pixel IoU vs reference %.5f (reference bf16 run: %.5f)"
          % (name, e_masks, ref16_masks, rel_err(iou.cpu().numpy(), gold["iou"]), e_post, iou_hip, iou_ref16))
    for k, v in taps.items():
        print("   tap %-9s rel_err %.5f" % (k, rel_err(v, gold[k])))
    exact_weights = bool(c.get("bf16_weights"))
    # token stream is fp32 end to end: with bf16-representable weights it tracks the reference to fp32-rounding-of-inputs level
    for i in range(2):
        assert rel_err(taps["queries%d" % i], gold["queries%d" % i]) < (0.004 if exact_weights else 0.02), i
        assert rel_err(taps["keys%d" % i], gold["keys%d" % i]) < (0.004 if exact_weights else 0.02), i
    assert e_masks < (0.01 if exact_weights else 0.02), e_masks
    # IoU head: the reference's own bf16 run is 0.5 - 3 % off its fp32 run on these cases; stay well inside that
    e_iou, ref16_iou = rel_err(iou.cpu().numpy(), gold["iou"]), rel_err(gold["iou_bf16"], gold["iou"])
    assert e_iou < (0.01 if exact_weights else 0.02) and e_iou <= max(ref16_iou, 0.006), (e_iou, ref16_iou)
    assert e_post < (0.01 if exact_weights else 0.02), e_post
    # never further from the fp32 reference than the reference's own bf16 run
    assert e_masks <= ref16_masks, (e_masks, ref16_masks)
    assert iou_hip >= iou_ref16 - 2e-4, (iou_hip, iou_ref16)
    if name.startswith("conf"):
        # north_star: masks within 1e-3 mIoU of the reference.  (a) the thresholded masks themselves agree to 1e-3; (b) against a
        # ground truth (the reference's mask with a band flipped) both score the same IoU to 1e-3, per mask and on average
        assert iou_hip >= 1.0 - 1e-3, iou_hip
        ref_post = torch.from_numpy(gold["post"][:, 0]).to(dev)
        gt = (ref_post > 0).float()
        gt[:, 100:140, :] = 1.0 - gt[:, 100:140, :]
        i_h, u_h, _ = ops.mask_iou(post[:, 0].contiguous(), gt.contiguous())
        i_r, u_r, _ = ops.mask_iou(ref_post.contiguous(), gt.contiguous())
        iou_h, iou_r = (i_h[:, 1] / u_h[:, 1]).cpu().numpy(), (i_r[:, 1] / u_r[:, 1]).cpu().numpy()
        assert np.abs(iou_h - iou_r).max() <= 1e-3 and abs(iou_h.mean() - iou_r.mean()) <= 1e-3, (iou_h, iou_r)


def test_projectors_vs_reference_golden(dev):
    c = cases.PROJECTORS["h64"]
    gold = cases.load("projectors_h64")
    wm, wt = cases.projector_weights(c)
    msqp = MultiScaleQFormerProjector(256, c["llama_dim"], target_square_side=6)
    ctp = CalibratedTextProjector(c["llama_dim"], 256)
    load_into(msqp, {"x." + k: v for k, v in wm.items()}, "x.", dev)
    load_into(ctp, {"x." + k: v for k, v in wt.items()}, "x.", dev)
    toks, hid = cases.projector_inputs(c)
    with torch.no_grad():
        a = msqp(toks.to(dev, torch.bfloat16))
        b = ctp(hid.to(dev, torch.bfloat16))
    assert rel_err(a.float().cpu().numpy(), gold["msqp"]) < 0.03
    assert rel_err(b.float().cpu().numpy(), gold["ctp"]) < 0.02


def test_msqp_full_width_vs_oracle(dev):
    """MSQP at the deployed width (H_llm = 4096, 64x64 SAM tokens) against the oracle on the same synthetic weights."""
    from oracle import projectors as oproj
    c = dict(llama_dim=4096, grid=64, batch=2, seed=33, ctp_shape=(4, 3))
    wm, wt = cases.projector_weights(c)
    msqp = MultiScaleQFormerProjector(256, 4096, target_square_side=6)
    ctp = CalibratedTextProjector(4096, 256)
    load_into(msqp, {"x." + k: v for k, v in wm.items()}, "x.", dev)
    load_into(ctp, {"x." + k: v for k, v in wt.items()}, "x.", dev)
    toks, hid = cases.projector_inputs(c)
    with torch.no_grad():
        a = msqp(toks.to(dev, torch.bfloat16))
        b = ctp(hid.to(dev, torch.bfloat16))
        ra = oproj.msqp(wm, toks.to(torch.bfloat16).float())
        rb = oproj.ctp(wt, hid.to(torch.bfloat16).float())
    assert a.shape == (2, 36, 4096)
    assert rel_err(a.float().cpu().numpy(), ra.numpy()) < 0.03
    assert rel_err(b.float().cpu().numpy(), rb.numpy()) < 0.02
    r = ops.resample_tokens(a)   # llava_arch.py:252-259
    assert rel_err(r.float().cpu().numpy(), oproj.resample_tokens(a.float().cpu()).numpy()) < 0.01


def test_clip_tower_vs_standin_golden(dev):
    from types import SimpleNamespace
    c = cases.CLIPS["tiny"]
    gold = cases.load("clip_tiny")
    cfg = dict(hidden_size=c["dim"], intermediate_size=4 * c["dim"], num_hidden_layers=c["layers"],
               num_attention_heads=c["heads"], image_size=c["img"], patch_size=14, layer_norm_eps=1e-5)
    args = SimpleNamespace(mm_vision_select_layer=c["select_layer"], pad_train_clip_images=True,
                           resize_vision_tower=True, resize_vision_tower_size=c["img"])
    tower = CLIPVisionTower("synthetic", args, config=cfg)
    load_into(tower.vision_tower, cases.clip_weights(c), "", dev)
    x, key_mask = cases.clip_inputs(c)
    xd = x.to(dev, torch.bfloat16)
    assert torch.equal(patch_key_mask(xd, c["clip_resize_list"]).cpu(), key_mask)
    with torch.no_grad():
        sel, pre = tower(xd, attention_mask=key_mask.to(dev))
        emb = tower.vision_tower.vision_model.hidden_states(xd, key_mask.to(dev), [0])[0]
    assert rel_err(emb.float().cpu().numpy(), gold["emb"]) < 0.01
    assert rel_err(sel.float().cpu().numpy(), gold["sel"]) < 0.03
    assert rel_err(pre[0].float().cpu().numpy(), gold["pre"]) < 0.03


def clip_calibration(dev, name):
    """The HIP tower against the stand-in's fp32 run, next to the stand-in's OWN bf16 run (tests/golden/clipcal_*.npz:
    `.bfloat16()` module and pixels, the precision the reference's callers use, evaluation_walkgpt.py:227-231,908-910): selected
    features, the -11 features, hidden-state taps, and text logits through a projector (llava_arch.py:97-104) + a small fp32 LM.
    Returns {quantity: (error of the HIP tower, error of the reference's bf16 run)}, errors = relative L2 against fp32."""
    from types import SimpleNamespace
    from tests.test_toplevel import TinyLM, H
    c = cases.CLIP_CALIBS[name]
    gold = cases.load("clipcal_" + name)
    cfg = dict(hidden_size=c["dim"], intermediate_size=4 * c["dim"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
               image_size=c["img"], patch_size=14, layer_norm_eps=1e-5)
    args = SimpleNamespace(mm_vision_select_layer=c["select_layer"], pad_train_clip_images=True, resize_vision_tower=True,
                           resize_vision_tower_size=c["img"])
    tower = CLIPVisionTower("synthetic", args, config=cfg)
    load_into(tower.vision_tower, cases.clip_weights(c), "", dev)
    x, km = cases.clip_calib_inputs(c)
    st, ts = c["stride"], c["tap_stride"]
    with torch.no_grad():
        hs = tower.vision_tower.vision_model.hidden_states(x.to(dev, torch.bfloat16), km.to(dev), [c["select_layer"], -11] + list(c["taps"]))
    sel = hs[c["select_layer"]][:, 1::st].float().cpu().numpy()
    pre = hs[-11][:, 1::st].float().cpu().numpy()
    out = {"sel": (rel_err(sel, gold["sel"]), rel_err(gold["sel_bf16"], gold["sel"])),
           "pre": (rel_err(pre, gold["pre"]), rel_err(gold["pre_bf16"], gold["pre"]))}
    for t in c["taps"]:
        out["h%d" % t] = (rel_err(hs[t][:, ::ts].float().cpu().numpy(), gold["h%d" % t]), rel_err(gold["h%d_bf16" % t], gold["h%d" % t]))
    g = torch.Generator().manual_seed(77)
    proj = torch.randn(H, c["dim"], generator=g) / c["dim"] ** 0.5
    lm = TinyLM()

    def logits(f):
        with torch.no_grad():
            return lm(inputs_embeds=torch.from_numpy(np.asarray(f)).float() @ proj.t(), output_hidden_states=True).logits

    l32, lh, l16 = logits(gold["sel"]), logits(sel), logits(gold["sel_bf16"])
    l1 = logits(torch.from_numpy(gold["sel"]).bfloat16().float().numpy())
    out["logits"] = (float((lh - l32).norm() / l32.norm()), float((l16 - l32).norm() / l32.norm()))
    out["logits_max_abs"] = (float((lh - l32).abs().max()), float((l16 - l32).abs().max()))
    print("%s: text logits (std %.2f): max |HIP - fp32| %.3e, reference-bf16 %.3e, fp32 features rounded once to bf16 %.3e (north_star "
          "bar 1e-4); rel L2 %.4f / %.4f; features: %s" % (name, float(l32.std()), out["logits_max_abs"][0], out["logits_max_abs"][1],
                                                           float((l1 - l32).abs().max()), out["logits"][0], out["logits"][1],
                                                           " ".join("%s %.4f/%.4f" % (k, *v) for k, v in out.items() if k[0] in "sph")))
    return out


def test_clip_tower_and_text_logits_calibrated_against_reference_bf16(dev):
    """north_star asks for text logits within 1e-4 abs of the reference's fp32 CPU path.  A tower whose WEIGHTS are bf16 -- the
    deployed checkpoint -- sits 0.4 % from fp32 after the patch embedding alone (tap h0) and 1 % at the selected layer, whoever runs
    it: the reference's own bf16 run shows the same figures.  So the measurable statement is the calibrated one, asserted here at
    every tap and on the logits: the HIP tower is at least as close to fp32 as the reference's bf16 run (3 % margin for the different
    rounding points: fp32 accumulators and fused epilogues here, a rounding after every op there).
    The logits are a small LM's function of the selected features: two feature errors of the same norm but different direction move them
    by different amounts.  Measured on this tower with two GEMM tilings whose features are equally close to fp32 at every tap (0.96 % at
    the selected layer, reference-bf16 1.00 %): logits 1.18 % (128x128 tiles) and 1.49 % (skinny kernel) against reference-bf16's 1.20 %.
    The logit assertion therefore carries the spread of that draw (1.5 x); the per-tap ones, which are norms of the error itself, do not."""
    cal = clip_calibration(dev, "tiny")
    for k, (e_hip, e_ref16) in cal.items():
        assert e_hip <= (1.5 if k.startswith("logits") else 1.03) * e_ref16, (k, e_hip, e_ref16)
    assert cal["sel"][0] < 0.012 and cal["logits"][0] < 0.02


def test_grounding_pipeline_end_to_end_vs_oracle(dev):
    """Tiny SAM encoder -> CTP -> prompt encoder -> mask decoder -> postprocess, HIP vs the oracle on CPU."""
    from oracle import projectors as oproj
    from oracle import sam as osam
    c = cases.SAM_ENCODERS["tiny"]
    g = c["img"] // c["patch"]
    model = WalkGPTGrounding(sam=dict(embed_dim=c["embed_dim"], depth=c["depth"], heads=c["heads"],
                                      global_idx=c["global_idx"], img=c["img"]), llm_hidden=64, with_clip=False)
    w_enc = cases.sam_encoder_weights(c)
    w_dec = cases.decoder_weights(5)
    pc = dict(cases.PROJECTORS["h64"])
    wm, wt = cases.projector_weights(pc)
    load_into(model.visual_model.image_encoder, w_enc, "image_encoder.", dev)
    load_into(model.visual_model.prompt_encoder, w_dec, "prompt_encoder.", dev, strict=False)
    load_into(model.visual_model.mask_decoder, w_dec, "mask_decoder.", dev)
    load_into(model.out_mm_projector, {"x." + k: v for k, v in wm.items()}, "x.", dev)
    load_into(model.text_hidden_fcs[0], {"x." + k: v for k, v in wt.items()}, "x.", dev)
    model.visual_model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix.data = \
        w_dec["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].to(dev)
    x = cases.sam_encoder_input(c)
    hidden = [torch.randn(2, 64, generator=torch.Generator().manual_seed(3)), torch.randn(3, 64, generator=torch.Generator().manual_seed(4))]
    resize, orig = [(512, 384), (400, 512)], [(200, 150), (75, 96)]
    out = model(x.to(dev, torch.bfloat16), None, [h.to(dev, torch.bfloat16) for h in hidden], resize, orig)
    # oracle
    w_all = dict(w_enc)
    w_all.update(w_dec)
    with torch.no_grad():
        emb = osam.image_encoder(w_all, x, dict(patch=c["patch"], depth=c["depth"], heads=c["heads"], global_idx=c["global_idx"], window=c["window"]))
        dpe = osam.dense_pe(w_all, (g, g))
        vis = oproj.msqp(wm, emb.flatten(2).transpose(1, 2))
        assert rel_err(out["visual_tokens"].float().cpu().numpy(), vis.numpy()) < 0.04
        for i in range(2):
            pe = oproj.ctp(wt, hidden[i].to(torch.bfloat16).float())  # [T, 256]
            sparse, dense = osam.prompt_encoder_text(w_all, pe.reshape(-1, 1, 256), (g, g))
            masks, _ = osam.mask_decoder(w_all, emb[i:i + 1], dpe, sparse, dense)
            ref = osam.postprocess_masks(masks, c["img"], resize[i], orig[i])[:, 0]
            got = out["pred_masks"][i].cpu()
            assert got.shape == ref.shape
            assert rel_err(got.numpy(), ref.numpy()) < 0.05
            a, b = got.numpy() > 0, ref.numpy() > 0
            assert (a & b).sum() / max(1, (a | b).sum()) > 0.98
            assert torch.allclose(out["mask_scores"][i].cpu(), osam.mask_score(ref), atol=0.02)
    # the decode chain replayed from its captured HIP graph is the same kernels with the same arguments: bit-identical, also
    # on a second replay with fresh inputs copied into the graph's static buffers (ragged prompt counts 2 + 3, two image sizes)
    hid = [h.to(dev, torch.bfloat16) for h in hidden]
    emb_t = model.get_visual_emb_tokens(x.to(dev, torch.bfloat16))
    # the encoder as two slices of the batch on two streams writing one output: images are independent, same rows bit for bit
    emb_s = model.get_visual_emb_tokens(x.to(dev, torch.bfloat16), sub_batches=2)
    torch.cuda.synchronize()
    assert emb_s.shape == emb_t.shape and torch.equal(emb_s, emb_t)
    for rep in range(2):
        hid = [h * (1.0 + rep) for h in hid]
        em, es = model.decode_from_hidden(emb_t, hid, resize, orig)
        gm, gs = model.decode_from_hidden_graphed(emb_t, hid, resize, orig)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(em, gm)) and all(torch.equal(a, b) for a, b in zip(es, gs))
    # the projector's tail folded into the decoder's first token launch (decode_from_hidden) against the unfolded chain (CTP with its own
    # tail launch, then decode): the same bits -- both call wg_ctp_tail_row
    pm, ps = model.decode(emb_t, model._project_seg_hidden(hid), resize, orig)
    assert all(torch.equal(a, b) for a, b in zip(em, pm)) and all(torch.equal(a, b) for a, b in zip(es, ps))
    # a caller that keeps its data in the graph's own input buffers replays without staging copies, same result
    s_emb, s_hid = model.decode_graph_inputs(emb_t, hid, resize, orig)
    s_emb.copy_(emb_t)
    for d_, h_ in zip(s_hid, hid):
        d_.copy_(h_ * 0.5)
    em, es = model.decode_from_hidden(emb_t, [h_ * 0.5 for h_ in hid], resize, orig)
    gm, gs = model.decode_from_hidden_graphed(s_emb, s_hid, resize, orig)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(em, gm)) and all(torch.equal(a, b) for a, b in zip(es, gs))
    # new weights after the capture (load_state_dict copies in place; .to() / re-assignment would move the storage): the graph key
    # carries the identity and version of every parameter and buffer of the chain, so the stale graph is dropped and re-captured
    md = model.visual_model.mask_decoder
    sd = {k: v * 1.25 if "output_hypernetworks_mlps.0.layers.2" in k or "output_upscaling.0.weight" in k else v for k, v in md.state_dict().items()}
    md.load_state_dict(sd)
    ctp = model.text_hidden_fcs[0]
    ctp.load_state_dict({k: (v * 0.5 if k == "net.3.weight" else v) for k, v in ctp.state_dict().items()})
    em2, _ = model.decode_from_hidden(emb_t, hid, resize, orig)
    gm2, _ = model.decode_from_hidden_graphed(emb_t, hid, resize, orig)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(em2, gm2))
    assert not any(torch.equal(a, b) for a, b in zip(em2, em))


def test_sub_batched_encoder_as_first_call_and_after_weight_edit(dev):
    """get_visual_emb_tokens(sub_batches=2) on a model whose derived operands (LayerNorm folds, re-laid weights) do not exist yet, and again right
    after an in-place weight edit: the operands are built on the caller's stream before the slices fork, so both slices read finished tensors --
    the same rows bit for bit as the one-stream call."""
    c = cases.SAM_ENCODERS["tiny"]
    x = cases.sam_encoder_input(c).to(dev, torch.bfloat16)
    x = torch.cat([x, x.flip(0)], 0) if x.shape[0] % 2 else x

    def fresh():
        m = WalkGPTGrounding(sam=dict(embed_dim=c["embed_dim"], depth=c["depth"], heads=c["heads"], global_idx=c["global_idx"], img=c["img"]),
                             llm_hidden=64, with_clip=False)
        load_into(m.visual_model.image_encoder, cases.sam_encoder_weights(c), "image_encoder.", dev)
        return m
    a, b = fresh(), fresh()
    torch.cuda.synchronize()
    first = a.get_visual_emb_tokens(x, sub_batches=2)          # nothing prepared yet on `a`
    torch.cuda.synchronize()
    ref = b.get_visual_emb_tokens(x)
    torch.cuda.synchronize()
    assert torch.equal(first, ref)
    for m in (a, b):
        with torch.no_grad():
            blk = m.visual_model.image_encoder.blocks[0]
            blk.norm1.weight.mul_(1.25)
            blk.mlp.lin1.bias.add_(0.5)
            m.visual_model.image_encoder.neck[2].weight.mul_(0.5)
    again = a.get_visual_emb_tokens(x, sub_batches=2)          # every keyed operand of block 0 and the neck is stale here
    ref2 = b.get_visual_emb_tokens(x)
    torch.cuda.synchronize()
    assert torch.equal(again, ref2) and not torch.equal(again, first)


def test_decoder_first_block_reads_images_through_the_prompt_map(dev):
    """Token->image attention partials and the image->token rows kernel with one block of image tokens per IMAGE and a prompt -> image map
    (what the first decoder block uses when images have several prompts) == the same kernels on per-prompt copies, bit for bit."""
    g = torch.Generator().manual_seed(33)
    B, hw, P = 3, 1024, 7
    pimg = torch.tensor([0, 0, 1, 2, 2, 2, 1], dtype=torch.int32)
    proj = torch.randn(B, hw, 384, generator=g).to(torch.bfloat16).to(dev)
    keys = torch.randn(B, hw, 256, generator=g).to(torch.bfloat16).to(dev)
    q = torch.randn(P, 6, 128, generator=g).to(dev)
    kq, vq = (torch.randn(P, 6, 128, generator=g).to(torch.bfloat16).to(dev) for _ in range(2))
    wo = (torch.randn(256, 128, generator=g) / 128 ** 0.5).to(torch.bfloat16).to(dev)
    bo, gam, bet, rb = (torch.randn(256, generator=g).to(torch.bfloat16).to(dev) for _ in range(4))
    idx = pimg.long().to(dev)
    a = ops.dec_attn_partial(q, proj[..., :256], pimg.to(dev))
    b = ops.dec_attn_partial(q, proj.index_select(0, idx)[..., :256].contiguous())
    assert torch.equal(a, b)
    a = ops.dec_i2t_rows(proj[..., 256:], kq, vq, wo, bo, keys, gam, bet, 1e-5, P, res_bias=rb, prompt_image=pimg.to(dev))
    b = ops.dec_i2t_rows(proj.index_select(0, idx)[..., 256:], kq, vq, wo, bo, keys.index_select(0, idx), gam, bet, 1e-5, P, res_bias=rb)
    assert torch.equal(a, b)


@pytest.mark.parametrize("P,hw,shared", [(1, 4096, False), (3, 64, False), (5, 1024, True)])
def test_decoder_image_to_token_rows_vs_torch(dev, P, hw, shared):
    """wg_dec_i2t_rows_bf16 (transformer.py:173-180 in one launch) against fp32 torch on the same bf16 operands."""
    g = torch.Generator().manual_seed(21)
    rows = 1 if shared else P
    proj = torch.randn(rows, hw, 384, generator=g).to(torch.bfloat16)
    keys = torch.randn(rows, hw, 256, generator=g).to(torch.bfloat16)
    kq, vq = (torch.randn(P, 6, 128, generator=g).to(torch.bfloat16) for _ in range(2))
    wo = (torch.randn(256, 128, generator=g) / 128 ** 0.5).to(torch.bfloat16)
    bo, gam, bet = (torch.randn(256, generator=g).to(torch.bfloat16) for _ in range(3))
    q = proj[..., 256:].float().expand(P, -1, -1).reshape(P, hw, 8, 16).transpose(1, 2)
    k = kq.float().reshape(P, 6, 8, 16).transpose(1, 2)
    v = vq.float().reshape(P, 6, 8, 16).transpose(1, 2)
    att = torch.softmax(q @ k.transpose(-1, -2) / 4.0, -1) @ v
    x = keys.float().expand(P, -1, -1) + att.transpose(1, 2).reshape(P, hw, 128) @ wo.float().t() + bo.float()
    ref = torch.nn.functional.layer_norm(x, (256,), gam.float(), bet.float(), 1e-5)
    pd = proj.to(dev)
    out = ops.dec_i2t_rows(pd[..., 256:], kq.to(dev), vq.to(dev), wo.to(dev), bo.to(dev), keys.to(dev), gam.to(dev), bet.to(dev), 1e-5, P)
    assert out.shape == (P, hw, 256)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= 0.04, err                      # bf16 output rounding of values up to ~6
    assert rel_err(out.float().cpu().numpy(), ref.numpy()) < 3e-3


def test_decoder_multimask_branch_vs_oracle(dev):
    """mask_decoder.py:106-111 with multimask_output=True: masks 1..3 and their IoU predictions (WalkGPT itself asks for mask 0 only;
    the module keeps SAM's switch).  HIP against the oracle's fp32 decoder on the g32 case's weights and inputs."""
    from oracle import sam as osam
    c = cases.DECODERS["g32"]
    g = c["grid"]
    sam = M._build_sam(128, 1, 2, [0], image_size=g * 16)
    w = cases.decoder_case_weights(c)
    load_into(sam.prompt_encoder, w, "prompt_encoder.", dev, strict=False)
    load_into(sam.mask_decoder, w, "mask_decoder.", dev)
    sam.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix.data = \
        w["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].to(dev)
    sam.to(dev)
    emb, text = cases.decoder_inputs(c)
    with torch.no_grad():
        dpe = sam.prompt_encoder.get_dense_pe()
        sparse, dense = sam.prompt_encoder(points=None, boxes=None, masks=None, text_embeds=text.to(dev, torch.bfloat16))
        m3, i3 = sam.mask_decoder(image_embeddings=emb.to(dev, torch.bfloat16), image_pe=dpe, sparse_prompt_embeddings=sparse,
                                  dense_prompt_embeddings=dense, multimask_output=True)
        m1, i1 = sam.mask_decoder(image_embeddings=emb.to(dev, torch.bfloat16), image_pe=dpe, sparse_prompt_embeddings=sparse,
                                  dense_prompt_embeddings=dense, multimask_output=False)
        wf = {k: v.float() for k, v in w.items()}
        T = text.shape[0]
        o_dense = wf["prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(T, -1, g, g)
        o_pe = dpe.float().cpu()
        om3, oi3 = osam.mask_decoder(wf, emb.float(), o_pe, text.float(), o_dense, multimask_output=True)
    assert m3.shape == om3.shape == (T, 3, 4 * g, 4 * g) and i3.shape == (T, 3)
    assert rel_err(m3.cpu().numpy(), om3.numpy()) < 0.02
    assert rel_err(i3.cpu().numpy(), oi3.numpy()) < 0.02
    # the three masks are not the single-mask output shifted by one: mask 0 is a different token
    assert rel_err(m3[:, :1].cpu().numpy(), m1.cpu().numpy()) > 0.05


def _e2e_model(c, dev):
    e = cases.SAM_ENCODERS[c["enc"]]
    pc = cases.PROJECTORS[c["proj"]]
    model = WalkGPTGrounding(sam=dict(embed_dim=e["embed_dim"], depth=e["depth"], heads=e["heads"], global_idx=e["global_idx"], img=e["img"]),
                             llm_hidden=pc["llama_dim"], with_clip=False)
    w_enc, w_dec, w_ctp = cases.e2e_weights(c)
    load_into(model.visual_model.image_encoder, w_enc, "image_encoder.", dev)
    load_into(model.visual_model.prompt_encoder, w_dec, "prompt_encoder.", dev, strict=False)
    load_into(model.visual_model.mask_decoder, w_dec, "mask_decoder.", dev)
    load_into(model.text_hidden_fcs[0], {"x." + k: v for k, v in w_ctp.items()}, "x.", dev)
    model.to(dev).bfloat16()                 # (the MSQP of this model keeps its default init: it runs, its tokens are not looked at)
    model.visual_model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix.data = \
        w_dec["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].to(dev)
    return model


def _iou_vs_band_gt(post, ref_post, dev):
    """IoU of `post` and of the reference's masks against ONE ground truth: the reference's mask with a band of rows flipped."""
    ref_post = torch.as_tensor(ref_post).to(dev)
    gt = (ref_post > 0).float()
    h = gt.shape[1]
    gt[:, h // 3: h // 3 + h // 8, :] = 1.0 - gt[:, h // 3: h // 3 + h // 8, :]
    i_h, u_h, _ = ops.mask_iou(post.float().contiguous(), gt.contiguous())
    i_r, u_r, _ = ops.mask_iou(ref_post.contiguous(), gt.contiguous())
    return (i_h[:, 1] / u_h[:, 1]).cpu().numpy(), (i_r[:, 1] / u_r[:, 1]).cpu().numpy()


@pytest.mark.parametrize("name", ["conf_tiny"])
def test_end_to_end_confident_masks_vs_reference(dev, name):
    """north_star's mask bar -- within 1e-3 mIoU of the reference's PyTorch CPU path -- END TO END: image -> SAM encoder -> CTP on the
    [SEG] states -> prompt encoder -> mask decoder -> postprocess through WalkGPTGrounding.forward, against the reference's own modules
    run in fp32 on the same image (tests/golden/e2e_*.npz; its bf16 run alongside as the calibration).  Then the same case with the
    encoder's GEMMs on fp8 operands (BASELINE config C5): its measured distance, with the bound it is held to."""
    c = cases.E2ES[name]
    gold = cases.load("e2e_" + name)
    model = _e2e_model(c, dev)
    x, hid = cases.e2e_inputs(c)
    xd, hd = x.to(dev, torch.bfloat16), hid.to(dev, torch.bfloat16)
    res = {}
    for dt in ("bf16", "fp8"):
        model.set_gemm_dtype(dt)
        with torch.no_grad():
            out = model(xd, None, [hd], [c["resize"]], [c["original"]])
            emb = model.get_visual_embs(xd)
        post = out["pred_masks"][0]
        assert post.shape == gold["post"].shape
        iou_h, iou_r = _iou_vs_band_gt(post, gold["post"], dev)
        res[dt] = dict(emb=rel_err(cases.tap_embedding(emb.float().cpu()).numpy(), gold["emb"]), post=rel_err(post.cpu().numpy(), gold["post"]),
                       pix=pixel_iou(post.cpu().numpy(), gold["post"]), d_iou=float(np.abs(iou_h - iou_r).max()),
                       d_miou=float(abs(iou_h.mean() - iou_r.mean())))
    model.set_gemm_dtype("bf16")
    ref16 = dict(emb=rel_err(gold["emb_bf16"], gold["emb"]), post=rel_err(gold["post_bf16"], gold["post"]),
                 pix=pixel_iou(gold["post_bf16"], gold["post"]))
    print("e2e %s: embedding rel err HIP %.4f / fp8 %.4f (reference bf16 run %.4f); mask logits %.4f / %.4f (%.4f); pixel IoU vs reference %.5f / %.5f "
          "(%.5f); |IoU vs GT - reference's| max %.5f / %.5f, mean %.5f / %.5f"
          % (name, res["bf16"]["emb"], res["fp8"]["emb"], ref16["emb"], res["bf16"]["post"], res["fp8"]["post"], ref16["post"],
             res["bf16"]["pix"], res["fp8"]["pix"], ref16["pix"], res["bf16"]["d_iou"], res["fp8"]["d_iou"], res["bf16"]["d_miou"], res["fp8"]["d_miou"]))
    b = res["bf16"]
    assert b["emb"] <= ref16["emb"] and b["post"] <= ref16["post"], (b, ref16)            # never further from fp32 than the reference's bf16 run
    assert b["pix"] >= ref16["pix"] - 2e-4 and b["pix"] >= 1.0 - 1e-3, (b["pix"], ref16["pix"])
    assert b["d_iou"] <= 1e-3 and b["d_miou"] <= 1e-3, b                                    # north_star: 1e-3 mIoU
    f = res["fp8"]
    assert f["pix"] >= 0.997 and f["d_iou"] <= 1e-3 and f["d_miou"] <= 1e-3, f              # config C5's operand type, end to end: inside
    #   north_star's 1e-3 (measured, MX block scales in the MLP: pixel IoU 0.99818, |IoU - reference's| <= 8.9e-4 per mask, 5.1e-4 on
    #   average; with per-row scales throughout: 0.99810 / 2.8e-4 / 1.2e-4 -- two draws of the same e4m3 rounding noise)


def test_decoder_takes_geometries_the_fused_kernels_do_not(dev):
    """The reference's MaskDecoder accepts any number of sparse prompt tokens (mask_decoder.py:125-132) and TwoWayTransformer.forward is a
    public entry (transformer.py:62-106); the fused token kernels are built for 5 + 1 tokens.  Other geometries run op by op
    (MaskDecoder._predict_masks_general, TwoWayTransformer.run_general): two prompt tokens per query here, against the oracle."""
    from oracle import sam as osam
    c = cases.DECODERS["g32"]
    g = c["grid"]
    sam = M._build_sam(128, 1, 2, [0], image_size=g * 16)
    w = cases.decoder_case_weights(c)
    load_into(sam.prompt_encoder, w, "prompt_encoder.", dev, strict=False)
    load_into(sam.mask_decoder, w, "mask_decoder.", dev)
    sam.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix.data = w["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].to(dev)
    sam.to(dev)
    emb, text = cases.decoder_inputs(c)                                   # [1,256,g,g], [3,1,256]
    two = torch.cat([text, text.flip(0) * 0.5], 1).to(torch.bfloat16).float()   # [3, 2, 256]: two prompt tokens per query
    embq = emb.to(torch.bfloat16).float()
    with torch.no_grad():
        dpe = sam.prompt_encoder.get_dense_pe()
        dense = sam.prompt_encoder.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(3, -1, g, g)
        assert not sam.mask_decoder.transformer.fused_ok(5, 2) and sam.mask_decoder.transformer.fused_ok(5, 1)
        masks, iou = sam.mask_decoder(image_embeddings=embq.to(dev, torch.bfloat16), image_pe=dpe, sparse_prompt_embeddings=two.to(dev, torch.bfloat16),
                                      dense_prompt_embeddings=dense, multimask_output=True)
        wq = {k: v.float() for k, v in w.items()}
        rdpe = osam.dense_pe(wq, (g, g))
        rsp, rdense = osam.prompt_encoder_text(wq, two[:, :1], (g, g))
        rm, ri = osam.mask_decoder(wq, embq, rdpe, two, rdense, multimask_output=True)
        # the public transformer entry on its own
        toks = torch.randn(2, 7, 256, generator=torch.Generator().manual_seed(1)).to(dev, torch.bfloat16)
        q, k = sam.mask_decoder.transformer(embq.to(dev, torch.bfloat16).expand(2, -1, -1, -1), dpe.expand(2, -1, -1, -1), toks)
        rq, rk = osam.two_way_transformer(wq, "mask_decoder.transformer", embq.expand(2, -1, -1, -1), rdpe.expand(2, -1, -1, -1), toks.float().cpu())
    assert masks.shape == (3, 3, 4 * g, 4 * g) and iou.shape == (3, 3)
    assert rel_err(masks.cpu().numpy(), rm.numpy()) < 0.03 and rel_err(iou.cpu().numpy(), ri.numpy()) < 0.03
    assert q.shape == (2, 7, 256) and k.shape == (2, g * g, 256)
    assert rel_err(q.float().cpu().numpy(), rq.numpy()) < 0.03 and rel_err(k.float().cpu().numpy(), rk.numpy()) < 0.03
