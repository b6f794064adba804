"""GPU parity of the layout / elementwise kernels against torch fp32 (exact where the op is data movement)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import sam as osam
from walkgpt_amd import ops


def _rand(shape, seed, std=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(dtype)


@pytest.mark.parametrize("B,C,H,P", [(2, 3, 64, 16), (1, 3, 1024, 16), (2, 3, 112, 14), (1, 3, 448, 14)])
def test_patchify_matches_unfold(dev, B, C, H, P):
    img = _rand((B, C, H, H), 1)
    K = C * P * P
    rows = ops.patchify(img.to(dev), P).cpu()
    ref = F.unfold(img.float(), P, stride=P).transpose(1, 2).reshape(-1, K).to(torch.bfloat16)
    assert torch.equal(rows[:, :K], ref)
    assert rows.shape[1] % 64 == 0 and (rows[:, K:] == 0).all()


def test_patchify_gemm_equals_conv(dev):
    img = _rand((2, 3, 64, 64), 2)
    w = _rand((128, 3, 16, 16), 3, 0.05)
    b = _rand((128,), 4, 0.1)
    ref = F.conv2d(img.float(), w.float(), b.float(), stride=16).permute(0, 2, 3, 1).reshape(-1, 128)
    out = ops.linear(ops.patchify(img.to(dev), 16), w.reshape(128, -1).to(dev), b.to(dev), out_f32=True)
    assert (out.cpu() - ref).abs().max().item() < 2e-3


def test_im2row3x3_gemm_equals_conv(dev):
    B, H, C, O = 2, 16, 64, 32
    x = _rand((B, H, H, C), 5)
    w = _rand((O, C, 3, 3), 6, 0.05)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, O)
    rows = ops.im2row3x3(x.to(dev), B, H, H)
    wr = w.permute(0, 2, 3, 1).reshape(O, 9 * C).contiguous()
    out = ops.linear(rows, wr.to(dev), out_f32=True)
    assert (out.cpu() - ref).abs().max().item() < 2e-3


def test_add_rows_broadcast(dev):
    a = _rand((3, 50, 256), 7)
    b = _rand((50, 256), 8)
    out = ops.add_rows(a.to(dev), b.to(dev)).cpu()
    assert torch.equal(out, (a.float() + b.float()).to(torch.bfloat16))


def test_layout_transposes_roundtrip(dev):
    x = _rand((2, 100, 48), 9)
    y = ops.tokens_to_nchw(x.to(dev), 2, 100, 48)
    assert torch.equal(y.cpu(), x.transpose(1, 2).contiguous())
    assert torch.equal(ops.nchw_to_tokens(y).cpu(), x)


def test_dense_pe_matches_oracle(dev):
    G = _rand((2, 128), 10, 1.0, torch.float32)
    ref = osam.dense_pe({"prompt_encoder.pe_layer.positional_encoding_gaussian_matrix": G}, (64, 64))
    pe = ops.dense_pe_tokens(G.to(dev), 64, 64).cpu()
    assert (pe - ref[0].flatten(1).t()).abs().max().item() < 2e-4


def test_hyper_mask_dot_pixel_shuffle(dev):
    # build `up` by running the two transposed convs in torch, then re-lay it the way the GEMMs emit it
    T, h, w = 2, 8, 8
    x = _rand((T, 64, 2 * h, 2 * w), 11)                       # after the first ConvT + LN + GELU (NCHW)
    w2 = _rand((64, 32, 2, 2), 12, 0.1)
    b2 = _rand((32,), 13, 0.1)
    up_ref = F.gelu(F.conv_transpose2d(x.float(), w2.float(), b2.float(), stride=2))   # [T,32,4h,4w]
    hyper = _rand((T, 4, 32), 14)
    ref = (hyper.float() @ up_ref.flatten(2)).reshape(T, 4, 4 * h, 4 * w)
    # rows = t, y, x, (dy,dx); cols = (dy2,dx2), c
    u = up_ref.reshape(T, 32, h, 2, 2, w, 2, 2)                # t c y dy dy2 x dx dx2
    u = u.permute(0, 2, 5, 3, 6, 4, 7, 1).reshape(T * h * w * 4, 4 * 32).to(torch.bfloat16)
    out = ops.hyper_mask_dot(u.to(dev), hyper.to(dev), T, h, w, 0, 4).cpu()
    assert (out - ref).abs().max().item() < 0.05 * ref.abs().max().item()
    one = ops.hyper_mask_dot(u.to(dev), hyper.to(dev), T, h, w, 2, 1).cpu()
    assert torch.equal(one[:, 0], out[:, 2])


@pytest.mark.parametrize("lh,img,inp,orig", [(128, 512, (384, 512), (75, 111)), (256, 1024, (1024, 683), (448, 299)),
                                             (256, 1024, (1024, 1024), (448, 448)), (256, 1024, (768, 1024), (1080, 1440))])
def test_postprocess_and_score(dev, lh, img, inp, orig):
    m = _rand((3, 1, lh, lh), 15, 4.0, torch.float32)
    ref = osam.postprocess_masks(m, img, inp, orig)
    out = ops.postprocess_masks(m.to(dev), img, inp, orig)
    # fp32 both sides; the tolerance covers a different (but equally valid) association of the 4-tap blends
    assert (out.cpu() - ref).abs().max().item() < 1e-4 * ref.abs().max().item() + 1e-5
    sc = ops.mask_score(out[:, 0].contiguous()).cpu()
    assert torch.allclose(sc, osam.mask_score(ref[:, 0]), atol=1e-5)
    # the one-pass form (postprocess + score): the same pixels bit for bit, the same score up to the summation order
    out2, sc2 = ops.postprocess_masks_scored(m.to(dev), img, inp, orig)
    assert torch.equal(out2, out[:, 0]) and torch.allclose(sc2.cpu(), sc, atol=1e-6)
    # ... which is ONE launch (the workgroup that completes a mask folds its partials): the same pixels as the two-launch entry, the same score
    # up to the summation order (its workgroups cover eight rows), and the same bits call after call (no race in the hand-over; the
    # tickets go back to zero)
    from walkgpt_amd import _lib
    L = _lib.lib()
    md = m.to(dev)
    N, H0, W0 = 3, orig[0], orig[1]
    nws = L.wg_postprocess_score_workspace_floats(N, H0, W0)
    ws, o3, s3 = torch.empty(nws, device=dev), torch.empty(N, H0, W0, device=dev), torch.empty(N, device=dev)
    rc = L.wg_postprocess_masks_score_f32(md.data_ptr(), o3.data_ptr(), s3.data_ptr(), ws.data_ptr(), nws, N, lh, lh, img, inp[0], inp[1], H0, W0,
                                          torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    first = None
    for _ in range(3):
        out4, sc4 = ops.postprocess_masks_scored(md, img, inp, orig)
        assert torch.equal(out4, o3) and torch.allclose(sc4, s3, rtol=0, atol=3e-6) and torch.allclose(sc4.cpu(), osam.mask_score(ref[:, 0]), atol=1e-5)
        first = sc4.clone() if first is None else first
        assert torch.equal(sc4, first)
    assert int(ops._score_tickets(torch.device(dev)).abs().sum()) == 0


def test_score_tickets_outlive_their_captured_graphs(dev):
    """More captured fused postprocess calls than the ticket pool holds (16), then eager allocations that would reuse any freed ticket words:
    every replay must still fold its scores (the graphs bake in the raw ticket addresses, so the sets have to stay allocated)."""
    lh, img, inp, orig = 256, 1024, (1024, 1024), (448, 448)
    ms = [(_rand((2, 1, lh, lh), 100 + i, 4.0, torch.float32)).to(dev) for i in range(20)]
    want = [ops.postprocess_masks_scored(m, img, inp, orig) for m in ms]
    torch.cuda.synchronize()
    graphs, outs = [], []
    side = torch.cuda.Stream()
    for m in ms:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                outs.append(ops.postprocess_masks_scored(m, img, inp, orig))
        graphs.append(g)
    torch.cuda.synchronize()
    # eager tensors of the ticket sets' size class, filled with ones: a freed set would be handed out again here
    junk = [torch.ones(ops._SCORE_TICKETS, device=dev, dtype=torch.int32) for _ in range(256)]
    for _ in range(2):
        for g in graphs:
            g.replay()
    torch.cuda.synchronize()
    for (o, s), (o0, s0) in zip(outs, want):
        assert torch.equal(o, o0) and torch.allclose(s, s0, rtol=0, atol=3e-6)
    assert all(int(j.sum()) == ops._SCORE_TICKETS for j in junk)
    assert len(ops._score_ticket_keep) >= 20


def test_mask_iou_and_losses_vs_reference_golden(dev):
    """SURVEY.md 8f rows 1-2 through the C-ABI against values the reference's own functions produced."""
    from tests.golden import cases
    c = cases.METRICS["m1"]
    gold = cases.load("metrics_m1")
    pred, gt = cases.metric_inputs(c)
    inter, union, tgt = ops.mask_iou(pred.to(dev), gt.to(dev), ignore_index=255)
    assert torch.equal(inter.cpu(), torch.from_numpy(gold["inter"]))        # integer counts: exact
    assert torch.equal(union.cpu(), torch.from_numpy(gold["union"]))
    assert torch.equal(tgt.cpu(), torch.from_numpy(gold["target"]))
    tg = (gt == 1).float()
    bce, dice = ops.mask_losses(pred.to(dev), tg.to(dev), num_masks=c["n"])
    assert abs(bce.item() - float(gold["bce"])) < 2e-5 * max(1.0, abs(float(gold["bce"])))
    assert abs(dice.item() - float(gold["dice"])) < 2e-5 * max(1.0, abs(float(gold["dice"])))
    # full-size property check: a mask compared with itself has union == intersection and dice -> ~0
    big = torch.randn(2, 1024, 1024, generator=torch.Generator().manual_seed(1)).to(dev)
    g = (big > 0).float()
    i2, u2, _ = ops.mask_iou(big, g)
    assert torch.equal(i2, u2) and float(i2.sum()) == 2 * 1024 * 1024


@pytest.mark.parametrize("name", ["k8", "pool", "one_row"])
def test_infonce_vs_reference_golden(dev, name):
    """walkgpt_amd.utils_walkgpt.infonce_loss / TinyCrossAttn (HIP) against the reference's outputs.
    Tolerances: the HIP path rounds the normalised [SEG] rows and the two query projections to bf16 (2^-9 relative) before
    the fp32-accumulating GEMM; logits are cosines / 0.07, so 1e-2 absolute there is 7e-4 in the cosine."""
    from tests.golden import cases
    from walkgpt_amd.utils_walkgpt import TinyCrossAttn, infonce_loss
    c = cases.NCES[name]
    gold = cases.load("nce_" + name)
    xa = TinyCrossAttn(d=c["D"], bias=c["bias"])
    xa.load_state_dict(cases.nce_weights(c), strict=True)
    xa = xa.to(dev).bfloat16()
    pred, tok, seg = cases.nce_inputs(c)
    loss, aux = infonce_loss(pred.to(dev).bfloat16(), tok.to(dev).bfloat16(), seg.to(dev), xa, temperature=0.07, top_k=c["top_k"],
                             exclude_same_row=c["exclude"], normalize=True, return_aux=True)
    torch.cuda.synchronize()
    gl = torch.from_numpy(gold["logits"])
    lg = aux["logits"].cpu()
    assert torch.equal(torch.isinf(lg), torch.isinf(gl))
    fin = ~torch.isinf(gl)
    assert (lg[fin] - gl[fin]).abs().max().item() < 2e-2
    assert abs(loss.item() - float(gold["loss"])) < 5e-3 * max(1.0, abs(float(gold["loss"])))
    aw = torch.from_numpy(gold["attn_w"])
    assert (aux["attn_w"].cpu() - aw).abs().max().item() < 2e-2 * aw.max().item()
    vp = torch.from_numpy(gold["v_pos"])
    assert ((aux["v_pos"].float().cpu() - vp).norm() / vp.norm()).item() < 2e-2
    # TinyCrossAttn.forward itself (attention-pooled path)
    v, a = xa(pred.to(dev).bfloat16(), tok[seg].to(dev).bfloat16())
    vx = torch.from_numpy(gold["xattn_out"])
    assert ((v.float().cpu() - vx).norm() / vx.norm()).item() < 2e-2
    assert (a.cpu() - torch.from_numpy(gold["xattn_attn"])).abs().max().item() < 2e-2 * float(gold["xattn_attn"].max())


def test_infonce_full_size_properties(dev):
    """C2-sized call (8 rows x 4096 tokens, 14 [SEG] per row): loss = mean of per-token cross entropies of the returned logits."""
    from walkgpt_amd.utils_walkgpt import TinyCrossAttn, infonce_loss
    g = torch.Generator().manual_seed(5)
    xa = TinyCrossAttn().to(dev).bfloat16()
    tok = torch.randn(8, 4096, 256, generator=g).to(dev).bfloat16()
    z = torch.randn(112, 256, generator=g).to(dev).bfloat16()
    seg = torch.arange(8).repeat_interleave(14).to(dev)
    loss, aux = infonce_loss(z, tok, seg, xa, top_k=8, return_aux=True)
    lg = aux["logits"]
    assert lg.shape == (112, 1 + 8 * 4096)
    ref = F.cross_entropy(lg, torch.zeros(112, dtype=torch.long, device=dev))
    assert abs(loss.item() - ref.item()) < 1e-4 * max(1.0, ref.item())
    own = lg[0, 1:1 + 4096]
    assert torch.isinf(own).all() and not torch.isinf(lg[0, 1 + 4096:]).any()
    assert (aux["attn_w"].sum(1) - 1).abs().max().item() < 1e-4


def test_infonce_forward_any_top_k_vs_oracle(dev):
    """infonce_loss forward with a top_k beyond the attention kernel's in-register selection (32): selection from its weights + the streaming
    pooling kernel, against the oracle's restatement of utils_walkgpt.py:8-73."""
    from oracle import metrics as om
    from walkgpt_amd.utils_walkgpt import TinyCrossAttn, infonce_loss
    g = torch.Generator().manual_seed(11)
    D, N, M, rows = 256, 1024, 6, 3
    xa = TinyCrossAttn(D)
    with torch.no_grad():
        for p_ in xa.parameters():
            p_.copy_((torch.randn(p_.shape, generator=g) * D ** -0.5).to(torch.bfloat16).float())
    w = {k: v.detach().clone() for k, v in xa.state_dict().items()}
    xa = xa.to(dev).bfloat16()
    z = torch.randn(M, D, generator=g).to(torch.bfloat16)
    tok = torch.randn(rows, N, D, generator=g).to(torch.bfloat16)
    seg = torch.randint(0, rows, (M,), generator=g)
    for top_k in (48, 200):
        loss = infonce_loss(z.to(dev), tok.to(dev), seg.to(dev), xa, temperature=0.07, top_k=top_k)
        ref, _ = om.infonce_loss(w, z.float(), tok.float(), seg, temperature=0.07, top_k=top_k)
        assert abs(loss.item() - ref.item()) < 2e-2 * abs(ref.item()), (top_k, loss.item(), ref.item())


def test_match_pred_vs_reference_golden(dev):
    """match_pred's cost matrix through the C-ABI against the reference's own functions; same assignment."""
    import numpy as np
    from tests.golden import cases
    from walkgpt_amd.matcher import match_pred
    c = cases.MATCHES["p5t4"]
    gold = cases.load("match_p5t4")
    pred, tgt, pts = cases.match_inputs(c)
    cost = ops.match_cost(pred.to(dev), tgt.to(dev), pts.to(dev)).cpu().numpy()
    assert np.allclose(cost, gold["cost"], rtol=2e-5, atol=2e-5)
    r, cidx = match_pred(pred.to(dev), tgt.to(dev), pts.to(dev))
    assert np.array_equal(r, gold["rows"]) and np.array_equal(cidx, gold["cols"])
    # full size, reference-style random points: every target is matched to the prediction it was derived from
    g = torch.Generator().manual_seed(3)
    big = torch.randn(6, 448, 448, generator=g) * 3
    t2 = (big[[4, 2, 0]] > 0).float()
    r2, c2 = match_pred(big.to(dev), t2.to(dev))
    assert sorted(zip(r2.tolist(), c2.tolist())) == [(0, 2), (2, 1), (4, 0)]


@pytest.mark.parametrize("name", ["down", "up", "square", "big"])
def test_preprocess_frames_bit_exact(dev, name):
    """the GPU input pipeline against the reference's ResizeLongestSide.apply_image (Pillow) output --
    uint8 resize bit-exact, normalised / padded fp32 image bit-exact, bf16 image = its rounding."""
    import numpy as np
    from tests.golden import cases
    from walkgpt_amd.preprocess import preprocess_frames, ResizeLongestSide
    c = cases.PREPS[name]
    gold = cases.load("prep_" + name)
    frame = torch.from_numpy(cases.prep_frame(c)).to(dev)
    img, resized, hw = preprocess_frames(torch.stack([frame, frame.flip(0)]), c["target"], out_dtype=torch.float32, want_resized=True)
    assert hw == gold["resized"].shape[:2]
    assert np.array_equal(resized[0].cpu().numpy(), gold["resized"])
    assert np.array_equal(img[0].cpu().numpy(), gold["image"])
    img16, _, _ = preprocess_frames(frame[None], c["target"])
    assert torch.equal(img16[0].cpu(), torch.from_numpy(gold["image"]).bfloat16())
    assert np.array_equal(ResizeLongestSide(c["target"]).apply_image(frame).cpu().numpy(), gold["resized"])


def test_preprocess_full_size_vs_oracle(dev):
    """1080p frame -> 1024 (SAM) and 448 (CLIP) against the oracle's numpy restatement, bit for bit."""
    import numpy as np
    from oracle import preprocess as op
    from walkgpt_amd.preprocess import preprocess_frames
    g = np.random.default_rng(7)
    frame = g.integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    for target in (1024, 448):
        img, resized, hw = preprocess_frames(torch.from_numpy(frame).to(dev)[None], target, out_dtype=torch.float32, want_resized=True)
        ref = op.resize_longest_side(frame, target)
        assert np.array_equal(resized[0].cpu().numpy(), ref)
        assert np.array_equal(img[0].cpu().numpy(), op.preprocess(ref, target, (97.17, 105.73, 108.16), (53.05, 56.40, 61.93)))


@pytest.mark.parametrize("name", ["w56", "w448"])
def test_clip_position_table_resize_vs_reference_golden(dev, name):
    """clip_encoder.resize_position_table (HIP bilinear on bf16 rows) against the reference's CLIPVisionTower.load_model run on the
    same synthetic table (tests/golden/clipwrap_*.npz): within bf16 rounding; the carried-over last row exactly."""
    from tests.golden import cases
    from walkgpt_amd.clip_encoder import resize_position_table
    c = cases.CLIPWRAPS[name]
    gold = torch.from_numpy(cases.load("clipwrap_" + name)["table"])
    table = cases.clipwrap_table(c).bfloat16()
    out = resize_position_table(table.to(dev), c["new_side"]).float().cpu()
    assert out.shape == gold.shape
    assert torch.equal(out[-1], table[-1].float())
    assert (out - gold).abs().max().item() <= 0.03 * max(1.0, gold.abs().max().item())


@pytest.mark.parametrize("name", ["r3", "vit_mask"])
def test_multimodal_splice_vs_reference_golden(dev, name):
    """The HIP resample + splice against the reference's own prepare_inputs_labels_for_multimodal (tests/golden/splice_*.npz):
    mask and labels bit-exact, text rows bit-exact after the bf16 cast, image rows within bf16 rounding of the fp32 resample."""
    from tests.golden import cases
    from walkgpt_amd import ops
    from walkgpt_amd.llava_splice import prepare_inputs_labels_for_multimodal
    c = cases.SPLICES[name]
    gold = cases.load("splice_" + name)
    ids, mask, labels, feats, table, vit = cases.splice_inputs(c)
    img = ops.resample_tokens(feats.bfloat16().to(dev).contiguous(), 16)
    m, e, l, _ = prepare_inputs_labels_for_multimodal(ids.to(dev), mask.to(dev), None if labels is None else labels.to(dev), img,
                                                      table.bfloat16().to(dev), None if vit is None else (vit > 0.5).to(dev))
    assert np.array_equal(m.cpu().numpy(), gold["attention_mask"])
    if labels is not None:
        assert np.array_equal(l.cpu().numpy(), gold["labels"])
    ge = torch.from_numpy(gold["inputs_embeds"])
    e = e.float().cpu()
    for r, s0 in enumerate(c["image_pos"]):
        text = [i for i in range(ge.shape[1]) if i < s0 or i >= s0 + 256]
        assert torch.equal(e[r, text], ge[r, text].bfloat16().float())
        assert (e[r, s0:s0 + 256] - ge[r, s0:s0 + 256]).abs().max().item() <= 0.03


def test_multimodal_splice_vs_oracle(dev):
    """the fused gather against the oracle's row-by-row restatement (bit-exact: pure data movement)."""
    from oracle import splice as osp
    from walkgpt_amd.llava_splice import prepare_inputs_labels_for_multimodal
    g = torch.Generator().manual_seed(11)
    rows, L, T, H, V = 5, 37, 16, 64, 101
    ids = torch.randint(0, V, (rows, L), generator=g)
    seg = [97, 99]
    for r, s in enumerate([0, 5, 36, 17, 20]):      # placeholder first / last / in the middle
        ids[r, s] = -200
        if s + 3 < L:
            ids[r, s + 3] = seg[r % 2]
    labels = torch.randint(-100, V, (rows, L), generator=g)
    mask = torch.rand(rows, L, generator=g) > 0.2
    vit = torch.rand(rows, T, generator=g) > 0.1
    table = torch.randn(V, H, generator=g).bfloat16()
    img = torch.randn(rows, T, H, generator=g).bfloat16()
    em, ee, el = osp.prepare_inputs_labels_for_multimodal(ids, mask, labels, img, table, vit)
    m, e, l, sm = prepare_inputs_labels_for_multimodal(ids.to(dev), mask.to(dev), labels.to(dev), img.to(dev), table.to(dev), vit.to(dev), seg)
    assert torch.equal(m.cpu(), em) and torch.equal(e.cpu(), ee) and torch.equal(l.cpu(), el)
    assert torch.equal(sm.cpu(), osp.seg_token_mask(ids, seg, T))
    # defaults (no masks, no labels) and WalkGPT's sizes: 256 image tokens, hidden 4096
    rows, L, T, H, V = 3, 130, 256, 4096, 32003
    ids = torch.randint(0, V, (rows, L), generator=g)
    ids[:, 35] = -200
    ids[:, 100] = 32000
    table = torch.randn(V, H, generator=g).bfloat16()
    img = torch.randn(rows, T, H, generator=g).bfloat16()
    em, ee, _ = osp.prepare_inputs_labels_for_multimodal(ids, None, None, img, table)
    m, e, l, sm = prepare_inputs_labels_for_multimodal(ids.to(dev), None, None, img.to(dev), table.to(dev), None, 32000)
    assert l is None and torch.equal(m.cpu(), em) and torch.equal(e.cpu(), ee)
    assert torch.equal(sm.cpu(), osp.seg_token_mask(ids, [32000], 256)) and int(sm.sum()) == rows and bool(sm[:, 99 + 255].all())
    bad = ids.clone(); bad[1, 36] = -200
    with pytest.raises(RuntimeError):
        prepare_inputs_labels_for_multimodal(bad.to(dev), None, None, img.to(dev), table.to(dev))
