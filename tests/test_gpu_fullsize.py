"""BASELINE config C2 at its full sizes (bs = 8, SAM ViT-B @ 1024, CLIP ViT-L/14 @ 448) through properties that do not need a full-size
reference run per image: an image's result must not depend on its batch (the bs = 8 launches take the persistent LayerNorm-folded GEMM
on M = 32768 / 8200 rows; one image alone tiles differently and, for CLIP, takes the unfused 128x128 path), two runs are bit-identical,
and ONE full-size CLIP image is held against the fp32 oracle (the full-size SAM image against the reference is test_gpu_modules'
`vit_b` case)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.golden import cases
from tests.test_gpu_modules import build_encoder, load_into, rel_err
from walkgpt_amd.clip_encoder import CLIP_VIT_L_14, CLIPVisionTower, patch_key_mask
from walkgpt_amd import synth


def test_sam_vit_b_full_size_batch_of_8(dev):
    c = dict(cases.SAM_ENCODERS["vit_b"])
    enc = build_encoder(c, dev)
    x = torch.from_numpy(synth.normal(7, "input.images8", (8, 3, 1024, 1024))).to(dev, torch.bfloat16)
    with torch.no_grad():
        out8 = enc(x)
        again = enc(x)
        one = [enc(x[i:i + 1]) for i in (0, 5)]
    assert out8.shape == (8, 256, 64, 64) and torch.isfinite(out8.float()).all()
    assert torch.equal(out8, again)
    for o, i in zip(one, (0, 5)):
        e = rel_err(out8[i].float().cpu().numpy(), o[0].float().cpu().numpy())
        assert e < 0.01, (i, e)
    # distinct images give distinct embeddings (a batch-index mix-up in the window / global addressing would not)
    assert rel_err(out8[0].float().cpu().numpy(), out8[5].float().cpu().numpy()) > 0.1


def test_clip_vit_l_448_full_size_batch_of_8(dev):
    from types import SimpleNamespace
    from oracle import clip as oclip
    cfg = dict(CLIP_VIT_L_14, image_size=448)
    c = dict(dim=1024, heads=16, layers=24, img=448, seed=43)
    args = SimpleNamespace(mm_vision_select_layer=-2, pad_train_clip_images=True, resize_vision_tower=True, resize_vision_tower_size=448)
    tower = CLIPVisionTower("synthetic", args, config=cfg)
    w = cases.clip_weights(c)
    load_into(tower.vision_tower, w, "", dev)
    x = torch.from_numpy(synth.normal(8, "input.images_clip8", (8, 3, 448, 448)))
    sizes = [(448, 448)] * 8
    sizes[3] = (300, 448)   # one padded image: its key mask rides along in the batch
    sizes[6] = (448, 210)
    xd = x.to(dev, torch.bfloat16)
    km = patch_key_mask(xd, sizes)
    with torch.no_grad():
        sel8, pre8 = tower(xd, attention_mask=km)
        sel8b, _ = tower(xd, attention_mask=km)
        sel1, pre1 = tower(xd[3:4], attention_mask=km[3:4])
    assert sel8.shape == (8, 1024, 1024) and torch.equal(sel8, sel8b)
    assert rel_err(sel8[3].float().cpu().numpy(), sel1[0].float().cpu().numpy()) < 0.02
    assert rel_err(pre8[0][3].float().cpu().numpy(), pre1[0][0].float().cpu().numpy()) < 0.02
    # the same image through the fp32 oracle (bf16-rounded weights and input, as the HIP tower holds them)
    wq = {k: v.bfloat16().float() for k, v in w.items()}
    with torch.no_grad():
        ref_sel, ref_pre = oclip.clip_tower(wq, x[3:4].bfloat16().float(), oclip.patch_key_mask(1, (448, 448), [sizes[3]]))
    assert rel_err(sel8[3].float().cpu().numpy(), ref_sel[0].numpy()) < 0.03
    assert rel_err(pre8[0][3].float().cpu().numpy(), ref_pre[0][0].numpy()) < 0.03


def test_clip_vit_l_448_fp8_mx_chain_full_size(dev):
    """The CLIP tower at its real size with the MX chain (D = 1024: four partial planes per LayerNorm fold, M = 2 x 1025 rows: ragged row tiles
    and key-masked attention in front of the quantisation pass): against the bf16 path on the same weights and, one image, the fp32 oracle."""
    from types import SimpleNamespace
    from oracle import clip as oclip
    cfg = dict(CLIP_VIT_L_14, image_size=448)
    c = dict(dim=1024, heads=16, layers=24, img=448, seed=43)
    args = SimpleNamespace(mm_vision_select_layer=-2, pad_train_clip_images=True, resize_vision_tower=True, resize_vision_tower_size=448)
    tower = CLIPVisionTower("synthetic", args, config=cfg)
    w = cases.clip_weights(c)
    load_into(tower.vision_tower, w, "", dev)
    x = torch.from_numpy(synth.normal(8, "input.images_clip8", (8, 3, 448, 448)))[:2]
    sizes = [(448, 448), (300, 448)]
    xd = x.to(dev, torch.bfloat16)
    km = patch_key_mask(xd, sizes)
    layers = tower.vision_tower.vision_model.encoder.layers
    with torch.no_grad():
        sel16, _ = tower(xd, attention_mask=km)
        for l in layers:
            l.gemm_dtype = "fp8"
        sel8, _ = tower(xd, attention_mask=km)
        sel8b, _ = tower(xd, attention_mask=km)
        for l in layers:
            l.gemm_dtype = "bf16"
    assert torch.equal(sel8, sel8b) and torch.isfinite(sel8.float()).all()
    wq = {k: v.bfloat16().float() for k, v in w.items()}
    with torch.no_grad():
        ref_sel, _ = oclip.clip_tower(wq, x[1:2].bfloat16().float(), oclip.patch_key_mask(1, (448, 448), [sizes[1]]))
    e16 = rel_err(sel16[1].float().cpu().numpy(), ref_sel[0].numpy())
    e8 = rel_err(sel8[1].float().cpu().numpy(), ref_sel[0].numpy())
    print("CLIP ViT-L/14 @ 448, selected features vs the fp32 oracle: bf16 GEMMs %.4f, fp8 MX chain %.4f" % (e16, e8))
    assert e16 < 0.03 and e8 < 0.12


def test_sam_vit_h_full_geometry_batch_of_4(dev):
    """Config C3's encoder at its full geometry (the reference's default, model/walkgpt.py:128: D = 1280, 32 blocks, 16 heads of 80,
    global attention at 7/15/23/31) on a batch of four: finite, deterministic, batch-independent, image-dependent.  The arithmetic of
    its blocks against the reference is the `vit_h3` golden (test_gpu_modules); this test is about the full depth and the batch."""
    c = dict(cases.SAM_ENCODERS["vit_b"])
    c.update(embed_dim=1280, depth=32, heads=16, global_idx=(7, 15, 23, 31), seed=16)
    enc = build_encoder(c, dev)
    x = torch.from_numpy(synth.normal(8, "input.images4", (4, 3, 1024, 1024))).to(dev, torch.bfloat16)
    with torch.no_grad():
        out4 = enc(x)
        again = enc(x)
        one = enc(x[2:3])
    assert out4.shape == (4, 256, 64, 64) and torch.isfinite(out4.float()).all()
    assert torch.equal(out4, again)
    e = rel_err(out4[2].float().cpu().numpy(), one[0].float().cpu().numpy())
    assert e < 0.02, e
    assert rel_err(out4[0].float().cpu().numpy(), out4[2].float().cpu().numpy()) > 0.1


def test_clip_vit_l_448_calibrated_against_reference_bf16(dev):
    """One full-size image (ViT-L/14 @ 448, padded to 300 x 448) against the stand-in's fp32 run, next to the stand-in's own bf16
    run: features, hidden-state taps 0 / 6 / 12 / 18 / 23 and text logits (tests/golden/clipcal_vit_l_448.npz)."""
    from tests.test_gpu_modules import clip_calibration
    cal = clip_calibration(dev, "vit_l_448")
    for k, (e_hip, e_ref16) in cal.items():
        assert e_hip <= 1.03 * e_ref16, (k, e_hip, e_ref16)
    assert cal["sel"][0] < 0.013 and cal["h0"][0] < 0.005


def test_c3_workload_at_its_own_batch(dev):
    """BASELINE config C3 on its own workload: bs = 32 at 448 x 448, SAM ViT-H (the reference's default, model/walkgpt.py:128) + MSQP at
    H_llm = 4096 + CTP + T = 14 [SEG] tokens per image -> masks.  One pass of the whole grounding step (what bench.py --config C3 times),
    checked through properties that need no full-size reference: shapes, finiteness, determinism, and image 5 of the batch against the
    same image run alone (other tilings: M = 131072 rows of the persistent GEMMs against 4096)."""
    from walkgpt_amd.walkgpt import WalkGPTGrounding
    B, T, Hl = 32, 14, 4096
    torch.manual_seed(0)
    model = WalkGPTGrounding(sam="vit_h", llm_hidden=Hl, with_clip=False).to(dev).bfloat16()
    enc = model.visual_model.image_encoder
    with torch.no_grad():                                            # zero-initialised by default (image_encoder.py:71-74,232-233)
        enc.pos_embed.normal_(0, 0.02)
        for blk in enc.blocks:
            blk.attn.rel_pos_h.normal_(0, 0.02)
            blk.attn.rel_pos_w.normal_(0, 0.02)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 3, 1024, 1024, generator=g).to(dev, torch.bfloat16)
    hid = [torch.randn(T, Hl, generator=g).to(dev, torch.bfloat16) for _ in range(B)]
    sizes, orig = [(1024, 1024)] * B, [(448, 448)] * B
    with torch.no_grad():
        out = model(x, None, hid, sizes, orig)
        again = model(x, None, hid, sizes, orig)
        one = model(x[5:6], None, hid[5:6], sizes[:1], orig[:1])
    assert len(out["pred_masks"]) == B and out["visual_tokens"].shape == (B, 36, Hl)
    for i in (0, 5, 31):
        m = out["pred_masks"][i]
        assert m.shape == (T, 448, 448) and torch.isfinite(m).all() and out["mask_scores"][i].shape == (T,)
        assert torch.equal(m, again["pred_masks"][i])
    e = rel_err(out["pred_masks"][5].float().cpu().numpy(), one["pred_masks"][0].float().cpu().numpy())
    assert e < 0.05, e
    assert rel_err(out["pred_masks"][5].float().cpu().numpy(), out["pred_masks"][0].float().cpu().numpy()) > 0.1   # distinct images, distinct masks


def _randomise_sam_tables(enc):
    with torch.no_grad():                                            # zero-initialised by default (image_encoder.py:71-74,232-233)
        enc.pos_embed.normal_(0, 0.02)
        for blk in enc.blocks:
            blk.attn.rel_pos_h.normal_(0, 0.02)
            blk.attn.rel_pos_w.normal_(0, 0.02)


def test_c5_workload_at_its_own_size(dev):
    """BASELINE config C5 on its own workload, one GPU's share: bs = 8, 1024 x 1024 originals, SAM ViT-H (32 blocks) with its qkv / proj /
    MLP GEMMs on fp8 MX operands, the CLIP ViT-L/14 tower beside it in bf16 (north_star asks fp8 of the hi-res SAM encoder only),
    MSQP + CTP + T = 14 [SEG] tokens per image -> masks at 1024 x 1024.  What bench.py --config C5 --dtype fp8 times, in one pass:
    finite, deterministic, an image of the batch against the same image alone, and the fp8 masks against the bf16 path on the SAME
    weights (thresholded agreement and logit distance, with the bounds they are held to)."""
    from tests.test_gpu_modules import pixel_iou
    from walkgpt_amd.walkgpt import WalkGPTGrounding
    B, T, Hl = 8, 14, 4096
    torch.manual_seed(0)
    model = WalkGPTGrounding(sam="vit_h", llm_hidden=Hl, with_clip=True).to(dev).bfloat16()
    pe = model.visual_model.prompt_encoder.pe_layer
    pe.positional_encoding_gaussian_matrix.data = pe.positional_encoding_gaussian_matrix.data.float()
    _randomise_sam_tables(model.visual_model.image_encoder)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, 1024, 1024, generator=g).to(dev, torch.bfloat16)
    xc = torch.randn(B, 3, 448, 448, generator=g).to(dev, torch.bfloat16)
    hid = [torch.randn(T, Hl, generator=g).to(dev, torch.bfloat16) for _ in range(B)]
    sizes, orig, csz = [(1024, 1024)] * B, [(1024, 1024)] * B, [(448, 448)] * B
    with torch.no_grad():
        ref16 = model(x, xc, hid, sizes, orig, csz)
        m16 = [m.clone() for m in ref16["pred_masks"]]
        f16 = ref16["clip_features"].clone()
        model.set_gemm_dtype("fp8")                                  # SAM encoder only: the CLIP tower stays bf16
        assert all(l.gemm_dtype == "bf16" for l in model.vision_tower.vision_tower.vision_model.encoder.layers)
        assert all(b.gemm_dtype == "fp8" for b in model.visual_model.image_encoder.blocks)
        out = model(x, xc, hid, sizes, orig, csz)
        m8 = [m.clone() for m in out["pred_masks"]]
        again = model(x, xc, hid, sizes, orig, csz)
        one = model(x[5:6], xc[5:6], hid[5:6], sizes[:1], orig[:1], csz[:1])
        model.set_gemm_dtype("bf16")
    torch.cuda.synchronize()
    assert len(m8) == B and out["visual_tokens"].shape == (B, 36, Hl) and out["clip_features"].shape == (B, 1024, 1024)
    assert torch.equal(out["clip_features"], f16)                    # the CLIP tower is untouched by the SAM encoder's operand type
    for i in range(B):
        assert m8[i].shape == (T, 1024, 1024) and torch.isfinite(m8[i]).all() and out["mask_scores"][i].shape == (T,)
        assert torch.equal(m8[i], again["pred_masks"][i])
    e_one = rel_err(m8[5].float().cpu().numpy(), one["pred_masks"][0].float().cpu().numpy())
    assert e_one < 0.05, e_one
    assert rel_err(m8[5].float().cpu().numpy(), m8[0].float().cpu().numpy()) > 0.1
    # fp8 against bf16 on the same weights.  White-noise images and random weights: the logits hover around zero, so every pixel is
    # a boundary pixel -- the worst case for thresholded agreement (confident masks: test_gpu_modules' e2e case stays within 3e-4)
    ious = [pixel_iou(m8[i].cpu().numpy(), m16[i].cpu().numpy()) for i in range(B)]
    errs = [rel_err(m8[i].float().cpu().numpy(), m16[i].float().cpu().numpy()) for i in range(B)]
    print("C5 at its own size (bs 8, ViT-H fp8 MX, T = 14, 1024^2): pixel IoU fp8 vs bf16 per image min %.4f mean %.4f; mask logits rel L2 "
          "max %.4f; image 5 in the batch vs alone %.4f" % (min(ious), float(np.mean(ious)), max(errs), e_one))
    assert min(ious) > 0.90 and max(errs) < 0.30


def test_c4_per_gpu_share_at_its_own_batch(dev):
    """BASELINE config C4's share of one GPU: bs = 32 of the C2 model (CLIP ViT-L/14 @ 448 + SAM ViT-B @ 1024 + CTP + decode, T = 1,
    448 x 448 originals, bf16) in one pass with the CLIP tower on its side stream -- what each of the eight ranks runs before the mask
    all-gather (tests/test_dist_gloo.py covers the exchange).  Shapes, finiteness, determinism, an image of the batch against the same
    image alone (M = 131072 / 32800 rows of the persistent GEMMs against 4096 / 1025)."""
    from walkgpt_amd.walkgpt import WalkGPTGrounding
    B, T, Hl = 32, 1, 4096
    torch.manual_seed(1)
    model = WalkGPTGrounding(sam="vit_b", llm_hidden=Hl, with_clip=True).to(dev).bfloat16()
    del model.out_mm_projector                                       # MSQP feeds the LLM, which C2 / C4 do not run (bench.py:build_model)
    pe = model.visual_model.prompt_encoder.pe_layer
    pe.positional_encoding_gaussian_matrix.data = pe.positional_encoding_gaussian_matrix.data.float()
    _randomise_sam_tables(model.visual_model.image_encoder)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, 3, 1024, 1024, generator=g).to(dev, torch.bfloat16)
    xc = torch.randn(B, 3, 448, 448, generator=g).to(dev, torch.bfloat16)
    hid = [torch.randn(T, Hl, generator=g).to(dev, torch.bfloat16) for _ in range(B)]
    ctp = model.text_hidden_fcs[0]
    sizes, orig, csz = [(1024, 1024)] * B, [(448, 448)] * B, [(448, 448)] * B

    def run(sl):
        emb = model.get_visual_emb_tokens(x[sl])
        feats, _ = model.encode_images_clip(xc[sl], csz[sl])
        masks, scores = model.decode_from_hidden(emb, hid[sl], sizes[sl], orig[sl])
        return feats, masks, scores

    with torch.no_grad():
        feats, masks, scores = run(slice(0, B))
        feats2, masks2, _ = run(slice(0, B))
        f1, m1, _ = run(slice(9, 10))
    torch.cuda.synchronize()
    assert feats.shape == (B, 1024, 1024) and torch.isfinite(feats.float()).all() and torch.equal(feats, feats2)
    assert len(masks) == B
    for i in range(B):
        assert masks[i].shape == (T, 448, 448) and torch.isfinite(masks[i]).all() and scores[i].shape == (T,)
        assert torch.equal(masks[i], masks2[i])
    e_m = rel_err(masks[9].float().cpu().numpy(), m1[0].float().cpu().numpy())
    e_f = rel_err(feats[9].float().cpu().numpy(), f1[0].float().cpu().numpy())
    print("C4 per-GPU share (bs 32): image 9 in the batch vs alone: masks %.4f, CLIP features %.4f" % (e_m, e_f))
    assert e_m < 0.05 and e_f < 0.02
    assert rel_err(masks[9].float().cpu().numpy(), masks[0].float().cpu().numpy()) > 0.1
    assert ctp is model.text_hidden_fcs[0]


def test_c2_step_on_three_streams_reproduces_itself_bit_for_bit(dev):
    """The fused C2 step as bench.py runs it (CLIP tower / SAM encoder in two slices / graph-replayed decode on HIP streams of their own, bs = 8) 25 times on the same
    inputs: CLIP features, SAM embedding, masks and scores of every pass equal the first pass's bit for bit.  The inference path has no atomics,
    so a difference would be an intermittent fault -- a missing wait state in front of a hand-placed MFMA (attn_pipe.hip), a counted wait that
    is one short (gemm.hip, attn_window_unit.hip), a buffer reused before its reader is done.  tools/soak_step.py is the long form (400 passes)."""
    from walkgpt_amd.walkgpt import WalkGPTGrounding
    B, T, Hl = 8, 1, 4096
    torch.manual_seed(2)
    model = WalkGPTGrounding(sam="vit_b", llm_hidden=Hl, with_clip=True).to(dev).bfloat16().eval()
    del model.out_mm_projector
    pe = model.visual_model.prompt_encoder.pe_layer
    pe.positional_encoding_gaussian_matrix.data = pe.positional_encoding_gaussian_matrix.data.float()
    _randomise_sam_tables(model.visual_model.image_encoder)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, 3, 1024, 1024, generator=g).to(dev, torch.bfloat16)
    xc = torch.randn(B, 3, 448, 448, generator=g).to(dev, torch.bfloat16)
    hid = [torch.randn(T, Hl, generator=g).to(dev, torch.bfloat16) for _ in range(B)]
    sizes, orig, csz = [(1024, 1024)] * B, [(448, 448)] * B, [(448, 448)] * B
    side, dec = torch.cuda.Stream(), torch.cuda.Stream()

    def step():
        with torch.no_grad():
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                feats, _ = model.encode_images_clip(xc, csz)
            emb = model.get_visual_emb_tokens(x, sub_batches=2)      # (bench.py --sam-split 2, the harder schedule: two slices on two streams; the bench's default is one slice)
            dec.wait_stream(cur)
            emb.record_stream(dec)
            with torch.cuda.stream(dec):
                masks, scores = model.decode_from_hidden_graphed(emb, hid, sizes, orig)
                masks, scores = [m.clone() for m in masks], [s.clone() for s in scores]
            cur.wait_stream(side)
            cur.wait_stream(dec)
        torch.cuda.synchronize()
        return [feats.clone(), emb.clone()] + masks + scores

    ref = step()
    assert all(torch.isfinite(t.float()).all() for t in ref)
    with torch.no_grad():
        whole = model.get_visual_emb_tokens(x, sub_batches=1)
    torch.cuda.synchronize()
    assert torch.equal(whole, ref[1]), "the encoder in two slices of the batch differs from the single pass"
    for it in range(25):
        out = step()
        bad = [i for i, (a, b) in enumerate(zip(ref, out)) if not torch.equal(a, b)]
        assert not bad, "pass %d: tensors %s differ from the first pass" % (it, bad)
