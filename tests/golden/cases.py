"""Golden-vector case table, shared by make_golden.py (which runs the reference) and by the tests (which run the
oracle / the HIP path).  Inputs and weights are regenerated from walkgpt_amd/synth.py; only outputs are on disk."""
import os

import numpy as np
import torch

from walkgpt_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))

# --- SAM image encoder ---------------------------------------------------------------------------------------
# "tiny": 512 px -> 32x32 grid: windowed blocks pad 32 -> 42 (3x3 windows, bias-valued pad keys), global blocks
#         at S=32.  "vit_b": the real SAM-B geometry (1024 px, 64x64 grid, pad to 70 = 5x5 windows), one image.
SAM_ENCODERS = {
    "tiny": dict(img=512, patch=16, embed_dim=128, depth=4, heads=2, global_idx=(1, 3), window=14, out=256,
                 batch=2, seed=11, tap_blocks=(0, 1, 3)),
    "tiny_hd32": dict(img=448, patch=16, embed_dim=64, depth=2, heads=2, global_idx=(1,), window=14, out=256,
                      batch=1, seed=12, tap_blocks=(0, 1)),
    # head_dim 80 (SAM ViT-H's) on the real 64x64 grid: one windowed (64 -> 70 padding) and one global block
    "hd80": dict(img=1024, patch=16, embed_dim=160, depth=2, heads=2, global_idx=(1,), window=14, out=256,
                 batch=1, seed=14, tap_blocks=(0, 1)),
    "vit_b": dict(img=1024, patch=16, embed_dim=768, depth=12, heads=12, global_idx=(2, 5, 8, 11), window=14,
                  out=256, batch=1, seed=13, tap_blocks=(0, 2, 11)),
    # SAM ViT-H's real width (the reference's default encoder, model/walkgpt.py:128, build_sam.py:15-24): D = 1280, 16 heads of 80, the
    # 64x64 grid; three of its 32 blocks (two windowed, one global) -- what the CPU affords for a reference run
    "vit_h3": dict(img=1024, patch=16, embed_dim=1280, depth=3, heads=16, global_idx=(2,), window=14, out=256, batch=1, seed=15,
                   tap_blocks=(1, 2)),
}


def sam_encoder_input(c):
    return torch.from_numpy(synth.normal(c["seed"], "input.images", (c["batch"], 3, c["img"], c["img"])))


def sam_encoder_weights(c, prefix="image_encoder."):
    D, p, g = c["embed_dim"], c["patch"], c["img"] // c["patch"]
    hd = D // c["heads"]
    shapes = {"pos_embed": (1, g, g, D), "patch_embed.proj.weight": (D, 3, p, p), "patch_embed.proj.bias": (D,),
              "neck.0.weight": (c["out"], D, 1, 1), "neck.1.weight": (c["out"],), "neck.1.bias": (c["out"],),
              "neck.2.weight": (c["out"], c["out"], 3, 3), "neck.3.weight": (c["out"],), "neck.3.bias": (c["out"],)}
    for i in range(c["depth"]):
        S = g if i in c["global_idx"] else c["window"]
        b = "blocks.%d." % i
        shapes.update({b + "norm1.weight": (D,), b + "norm1.bias": (D,), b + "norm2.weight": (D,), b + "norm2.bias": (D,),
                       b + "attn.rel_pos_h": (2 * S - 1, hd), b + "attn.rel_pos_w": (2 * S - 1, hd),
                       b + "attn.qkv.weight": (3 * D, D), b + "attn.qkv.bias": (3 * D,),
                       b + "attn.proj.weight": (D, D), b + "attn.proj.bias": (D,),
                       b + "mlp.lin1.weight": (4 * D, D), b + "mlp.lin1.bias": (4 * D,),
                       b + "mlp.lin2.weight": (D, 4 * D), b + "mlp.lin2.bias": (D,)})
    return {prefix + k: torch.from_numpy(synth.param(c["seed"], prefix + k, s)) for k, s in shapes.items()}


def tap_tokens(h):
    """[B,H,W,D] block output -> strided slice."""
    return h[:, ::2, ::2, ::4].contiguous()


def tap_embedding(out):
    """[B,C,h,w] encoder output -> strided slice."""
    return out[:, ::4, ::2, ::2].contiguous()


# --- prompt encoder + mask decoder + postprocess -----------------------------------------------------------------
# "g32" / "g64": white-noise embedding, fp32 weights (every pixel of the mask is a boundary pixel: the hardest case for thresholded
# agreement; kept from round 1, now with the reference's own bf16 run stored next to its fp32 run as the calibration of "just bf16").
# "conf_g64": the confident-mask case.  The embedding is piecewise constant over nine regions (+2 % noise), as the embedding of an
# image of a few objects is, and the two transposed convolutions use the same weights for their 2x2 sub-pixels (a trained upscaler
# is coherent across sub-pixels; independent random ones paint a 4x4 texture whose amplitude exceeds the differences between regions),
# so the reference's logits are piecewise constant: four regions at >= +0.19, five at <= -0.19, 0.011 of spread inside a region (the
# seed was searched for that margin with the reference itself).  Weights and inputs are bf16-representable (the deployed checkpoint
# IS bf16: evaluation_walkgpt.py:908-910 casts the model), so the only difference between the reference's fp32 run and the HIP path
# is the arithmetic in between.
DECODERS = {
    "g32": dict(grid=32, tokens=3, seed=21, input_size=(384, 512), original_size=(75, 111)),
    "g64": dict(grid=64, tokens=2, seed=22, input_size=(1024, 683), original_size=(448, 299)),
    "conf_g64": dict(grid=64, tokens=3, seed=50, input_size=(1024, 1024), original_size=(448, 448), style="regions", regions=9,
                     bf16_weights=True, coherent_upscaler=True),
}


def _bf16_round(t):
    return t.to(torch.bfloat16).float()


def decoder_inputs(c):
    g = c["grid"]
    if c.get("style") == "regions":
        R = c["regions"]
        u = synth.uniform01(c["seed"], "input.region_sites", 2 * R).reshape(R, 2) * g
        yy, xx = np.mgrid[0:g, 0:g]
        d = (yy[None] + 0.5 - u[:, 0, None, None]) ** 2 + (xx[None] + 0.5 - u[:, 1, None, None]) ** 2
        region = torch.from_numpy(d.argmin(0))                                     # [g, g] nearest site
        vals = torch.from_numpy(synth.normal(c["seed"], "input.region_values", (R, 256)))
        emb = vals[region].permute(2, 0, 1)[None].contiguous()
        emb = emb + 0.02 * torch.from_numpy(synth.normal(c["seed"], "input.image_embedding", (1, 256, g, g)))
    else:
        emb = torch.from_numpy(synth.normal(c["seed"], "input.image_embedding", (1, 256, g, g)))
    text = torch.from_numpy(synth.normal(c["seed"], "input.text_embeds", (c["tokens"], 1, 256)))
    text = torch.nn.functional.normalize(text, dim=-1)
    if c.get("bf16_weights"):
        emb, text = _bf16_round(emb), _bf16_round(text)
    return emb, text


def decoder_case_weights(c):
    w = decoder_weights(c["seed"])
    if c.get("coherent_upscaler"):
        for k in ("mask_decoder.output_upscaling.0.weight", "mask_decoder.output_upscaling.3.weight"):
            w[k] = w[k][:, :, :1, :1].expand_as(w[k]).contiguous()
    if c.get("bf16_weights"):
        w = {k: (v if "gaussian_matrix" in k else _bf16_round(v)) for k, v in w.items()}   # the PE buffer stays fp32 in the build
    return w


def tap_keys(keys, g):
    """[T, g*g, C] image-token stream -> strided slice (the fixtures stay small)."""
    return keys.reshape(keys.shape[0], g, g, -1)[:, ::8, ::8, ::4].contiguous()


def tap_upscaled(up):
    """[T, 32, 4g, 4g] upscaled embedding -> strided slice."""
    return up[:, :, ::16, ::16].contiguous()


def _mlp3_shapes(prefix, din, dh, dout):
    return {prefix + "layers.0.weight": (dh, din), prefix + "layers.0.bias": (dh,),
            prefix + "layers.1.weight": (dh, dh), prefix + "layers.1.bias": (dh,),
            prefix + "layers.2.weight": (dout, dh), prefix + "layers.2.bias": (dout,)}


def decoder_weight_shapes():
    """Hot-path subset of the prompt-encoder + mask-decoder state_dict."""
    s = {"prompt_encoder.pe_layer.positional_encoding_gaussian_matrix": (2, 128),
         "prompt_encoder.no_mask_embed.weight": (1, 256),
         "mask_decoder.iou_token.weight": (1, 256), "mask_decoder.mask_tokens.weight": (4, 256)}
    T = "mask_decoder.transformer."

    def attn(p, inner):
        for n in ("q_proj", "k_proj", "v_proj"):
            s[p + n + ".weight"] = (inner, 256)
            s[p + n + ".bias"] = (inner,)
        s[p + "out_proj.weight"] = (256, inner)
        s[p + "out_proj.bias"] = (256,)

    for i in range(2):
        L = T + "layers.%d." % i
        attn(L + "self_attn.", 256)
        attn(L + "cross_attn_token_to_image.", 128)
        attn(L + "cross_attn_image_to_token.", 128)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            s[L + n + ".weight"] = (256,)
            s[L + n + ".bias"] = (256,)
        s[L + "mlp.lin1.weight"] = (2048, 256)
        s[L + "mlp.lin1.bias"] = (2048,)
        s[L + "mlp.lin2.weight"] = (256, 2048)
        s[L + "mlp.lin2.bias"] = (256,)
    attn(T + "final_attn_token_to_image.", 128)
    s[T + "norm_final_attn.weight"] = (256,)
    s[T + "norm_final_attn.bias"] = (256,)
    s["mask_decoder.output_upscaling.0.weight"] = (256, 64, 2, 2)
    s["mask_decoder.output_upscaling.0.bias"] = (64,)
    s["mask_decoder.output_upscaling.1.weight"] = (64,)
    s["mask_decoder.output_upscaling.1.bias"] = (64,)
    s["mask_decoder.output_upscaling.3.weight"] = (64, 32, 2, 2)
    s["mask_decoder.output_upscaling.3.bias"] = (32,)
    for i in range(4):
        s.update(_mlp3_shapes("mask_decoder.output_hypernetworks_mlps.%d." % i, 256, 256, 32))
    s.update(_mlp3_shapes("mask_decoder.iou_prediction_head.", 256, 256, 4))
    return s


def decoder_weights(seed):
    return {k: torch.from_numpy(synth.param(seed, k, sh)) for k, sh in decoder_weight_shapes().items()}


# --- MSQP + CTP ----------------------------------------------------------------------------------------------
PROJECTORS = {
    "h64": dict(llama_dim=64, grid=32, batch=2, seed=31, ctp_shape=(3, 5)),
}


def msqp_weight_shapes(llama_dim, d=1024, sam_dim=256):
    s = {"pad_token": (1, 1, d), "sam_to_proj.weight": (d, sam_dim), "sam_to_proj.bias": (d,),
         "q_x1": (1, 12, d), "q_x2": (1, 8, d), "q_x4": (1, 8, d), "q_global": (1, 4, d),
         "gate.net.0.weight": (d,), "gate.net.0.bias": (d,), "gate.net.1.weight": (128, d), "gate.net.1.bias": (128,),
         "gate.net.3.weight": (1, 128), "gate.net.3.bias": (1,),
         "to_llama.weight": (llama_dim, d), "to_llama.bias": (llama_dim,)}
    for c in ("cross_x1", "cross_x2", "cross_x4", "cross_glb"):
        for i in range(2):
            p = "%s.%d." % (c, i)
            s.update({p + "q_norm.weight": (d,), p + "q_norm.bias": (d,), p + "kv_norm.weight": (d,), p + "kv_norm.bias": (d,),
                      p + "attn.in_proj_weight": (3 * d, d), p + "attn.in_proj_bias": (3 * d,),
                      p + "attn.out_proj.weight": (d, d), p + "attn.out_proj.bias": (d,),
                      p + "ffn.0.weight": (d,), p + "ffn.0.bias": (d,), p + "ffn.1.weight": (4 * d, d),
                      p + "ffn.1.bias": (4 * d,), p + "ffn.3.weight": (d, 4 * d), p + "ffn.3.bias": (d,)})
    return s


def ctp_weight_shapes(in_dim, out_dim=256):
    mid = out_dim * 2
    return {"net.0.weight": (in_dim,), "net.0.bias": (in_dim,), "net.1.weight": (mid, in_dim), "net.1.bias": (mid,),
            "net.3.weight": (out_dim, mid), "net.3.bias": (out_dim,), "net.4.weight": (out_dim,), "net.4.bias": (out_dim,),
            "text_type": (1, 1, out_dim), "log_temp": (1,)}


def projector_weights(c):
    m = {k: torch.from_numpy(synth.param(c["seed"], "out_mm_projector." + k, s))
         for k, s in msqp_weight_shapes(c["llama_dim"]).items()}
    t = {k: torch.from_numpy(synth.param(c["seed"], "text_hidden_fcs.0." + k, s))
         for k, s in ctp_weight_shapes(c["llama_dim"]).items()}
    return m, t


def projector_inputs(c):
    g = c["grid"]
    toks = torch.from_numpy(synth.normal(c["seed"], "input.sam_tokens", (c["batch"], g * g, 256)))
    hid = torch.from_numpy(synth.normal(c["seed"], "input.hidden", c["ctp_shape"] + (c["llama_dim"],)))
    return toks, hid


# --- CLIP tower (stand-in pin) ---------------------------------------------------------------------------------
CLIPS = {
    "tiny": dict(dim=128, heads=2, layers=12, img=112, select_layer=-2, batch=2, seed=41,
                 clip_resize_list=[(112, 112), (70, 98)]),
}


# calibration of the tower's precision: the stand-in run in fp32 (full-precision weights) AND in bf16 (`.bfloat16()` module and input,
# the precision the reference's callers use: evaluation_walkgpt.py:227-231,908-910), with hidden-state taps.  `taps`: indices into
# hidden_states (0 = pre_layrnorm output); `stride`: token stride of what is stored (full-size outputs are stored as slices).
CLIP_CALIBS = {
    "tiny": dict(dim=128, heads=2, layers=12, img=112, select_layer=-2, batch=2, seed=41, clip_resize_list=[(112, 112), (70, 98)],
                 taps=(0, 3, 6, 9, 11), stride=1, tap_stride=1),
    # ONE full-size image: ViT-L/14 at 448 px, 1025 tokens, padded to (300, 448) -- image 3 of test_gpu_fullsize's batch
    "vit_l_448": dict(dim=1024, heads=16, layers=24, img=448, select_layer=-2, batch=1, seed=43, clip_resize_list=[(300, 448)],
                      taps=(0, 6, 12, 18, 23), stride=16, tap_stride=64, input=(8, "input.images_clip8", 8, 3)),
}


def clip_calib_inputs(c):
    from oracle.clip import patch_key_mask
    if "input" in c:
        seed, key, n, pick = c["input"]
        x = torch.from_numpy(synth.normal(seed, key, (n, 3, c["img"], c["img"])))[pick:pick + 1]
    else:
        x = torch.from_numpy(synth.normal(c["seed"], "input.images_clip", (c["batch"], 3, c["img"], c["img"])))
    return x, patch_key_mask(x.shape[0], (c["img"], c["img"]), c["clip_resize_list"])


def clip_weight_shapes(c):
    D, P = c["dim"], (c["img"] // 14) ** 2
    s = {"vision_model.embeddings.class_embedding": (D,),
         "vision_model.embeddings.patch_embedding.weight": (D, 3, 14, 14),
         "vision_model.embeddings.position_embedding.weight": (P + 1, D),
         "vision_model.pre_layrnorm.weight": (D,), "vision_model.pre_layrnorm.bias": (D,),
         "vision_model.post_layernorm.weight": (D,), "vision_model.post_layernorm.bias": (D,)}
    for i in range(c["layers"]):
        L = "vision_model.encoder.layers.%d." % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[L + "self_attn." + n + ".weight"] = (D, D)
            s[L + "self_attn." + n + ".bias"] = (D,)
        for n in ("layer_norm1", "layer_norm2"):
            s[L + n + ".weight"] = (D,)
            s[L + n + ".bias"] = (D,)
        s[L + "mlp.fc1.weight"] = (4 * D, D)
        s[L + "mlp.fc1.bias"] = (4 * D,)
        s[L + "mlp.fc2.weight"] = (D, 4 * D)
        s[L + "mlp.fc2.bias"] = (D,)
    return s


def clip_weights(c):
    return {k: torch.from_numpy(synth.param(c["seed"], k, sh)) for k, sh in clip_weight_shapes(c).items()}


def clip_inputs(c):
    from oracle.clip import patch_key_mask
    x = torch.from_numpy(synth.normal(c["seed"], "input.images_clip", (c["batch"], 3, c["img"], c["img"])))
    return x, patch_key_mask(c["batch"], (c["img"], c["img"]), c["clip_resize_list"])


# --- eval metric + mask losses ---------------------------------------------------------------
METRICS = {"m1": dict(n=5, h=61, w=83, seed=51)}


def metric_inputs(c):
    """pred logits [n,h,w] fp32; gt [n,h,w] fp32 in {0,1} with a band of 255 (ignore) pixels."""
    pred = torch.from_numpy(synth.normal(c["seed"], "input.pred_logits", (c["n"], c["h"], c["w"]), 3.0))
    u = torch.from_numpy(synth.uniform01(c["seed"], "input.gt", c["n"] * c["h"] * c["w"]).astype(np.float32)).reshape(c["n"], c["h"], c["w"])
    gt = (u > 0.6).float()
    gt[:, :3, :] = 255.0
    gt[0] = 0.0            # an empty ground truth (union only from the prediction)
    return pred, gt


def load(name):
    return dict(np.load(os.path.join(HERE, name + ".npz")))


# ---- region-alignment InfoNCE (utils_walkgpt.py:8-73, 330-357) ---------------------------------------------------------
NCES = {
    "k8": dict(rows=3, side=8, D=256, M=5, seg_rows=[0, 0, 1, 2, 2], top_k=8, exclude=True, bias=False, seed=71),
    "pool": dict(rows=2, side=8, D=256, M=3, seg_rows=[1, 0, 1], top_k=None, exclude=True, bias=True, seed=72),
    "one_row": dict(rows=1, side=8, D=256, M=2, seg_rows=[0, 0], top_k=8, exclude=False, bias=False, seed=73),
}


def nce_weights(c):
    names = ["wq", "wk", "wv", "out"]
    w = {}
    for n in names:
        w[n + ".weight"] = torch.from_numpy(synth.normal(c["seed"], "tiny_xattn.%s.weight" % n, (c["D"], c["D"]), c["D"] ** -0.5))
        if c["bias"]:
            w[n + ".bias"] = torch.from_numpy(synth.normal(c["seed"], "tiny_xattn.%s.bias" % n, (c["D"],), 0.1))
    return w


def nce_inputs(c):
    """(pred [M,D], sam_tokens [rows,N,D], seg_row_ids [M]) fp32; bf16-representable so that the HIP path sees the same numbers."""
    N = c["side"] ** 2
    pred = torch.from_numpy(synth.normal(c["seed"], "input.pred", (c["M"], c["D"]), 1.0)).bfloat16().float()
    tok = torch.from_numpy(synth.normal(c["seed"], "input.sam_tokens", (c["rows"], N, c["D"]), 1.0)).bfloat16().float()
    return pred, tok, torch.tensor(c["seg_rows"], dtype=torch.long)


# ---- match_pred cost matrix (utils/matcher.py:93-133) ---------------------------------------------------------------------
MATCHES = {"p5t4": dict(P=5, T=4, h=61, w=83, points=1024, seed=81)}


def match_inputs(c):
    """pred logits [P,h,w]; targets [T,h,w] in {0,1}; points [NP,2] in [0,1) (x, y).  Targets are noisy copies of thresholded
    predictions so that the assignment is well determined."""
    pred = torch.from_numpy(synth.normal(c["seed"], "input.pred", (c["P"], c["h"], c["w"]), 3.0))
    u = torch.from_numpy(synth.uniform01(c["seed"], "input.flip", c["T"] * c["h"] * c["w"]).astype(np.float32)).reshape(c["T"], c["h"], c["w"])
    tgt = ((pred[[3, 0, 4, 1][:c["T"]]] > 0).float() + (u > 0.9).float()).remainder(2.0)
    pts = torch.from_numpy(synth.uniform01(c["seed"], "input.points", c["points"] * 2).astype(np.float32)).reshape(c["points"], 2)
    return pred, tgt, pts


# ---- input pipeline (transforms.py:27-36, PAVE_dataset.py:115-121) ---------------------------------------------------------
PREPS = {
    "down": dict(h=37, w=53, target=32, seed=91),     # antialiased down-scale (5 taps), landscape
    "up": dict(h=31, w=20, target=48, seed=92),       # up-scale (3 taps), portrait
    "square": dict(h=40, w=40, target=40, seed=93),   # no resize at all
    "big": dict(h=270, w=480, target=128, seed=94),   # 3.75x down-scale (9 taps)
}


def prep_frame(c):
    u = synth.uniform01(c["seed"], "input.frame", c["h"] * c["w"] * 3).reshape(c["h"], c["w"], 3)
    yy, xx = np.mgrid[0:c["h"], 0:c["w"]]
    smooth = 0.5 + 0.5 * np.sin(yy[..., None] / 7.0 + xx[..., None] / 5.0 + np.arange(3))
    return np.clip((0.6 * smooth + 0.4 * u) * 255.0, 0, 255).astype(np.uint8)


# ---- LLM-side splice (llava_arch.py:213-518, the branch WalkGPT takes) -------------------------------------------------------
SPLICES = {
    "r3": dict(rows=3, L=11, hidden=32, vocab=50, image_pos=[2, 0, 9], seed=101),   # placeholder mid-row, first and near the end
    "vit_mask": dict(rows=2, L=8, hidden=16, vocab=40, image_pos=[3, 5], seed=102, vit_mask=True, no_labels=True),
}


# --- end to end, confident masks: image -> SAM encoder -> (LLM states ->) CTP -> prompt encoder -> mask decoder -> postprocess ----------
# The image is a few flat-coloured regions (+ a little noise), as a photograph of a few objects is; the decoder's transposed
# convolutions are coherent across sub-pixels and weights / inputs are bf16-representable (as in DECODERS["conf_g64"]).  Image and
# decoder seeds were searched with the oracle for margin: in the reference's fp32 run the logits of `conf_tiny` are farther than 2 % of
# their rms from zero on all but 0.13 % of the pixels, so thresholded masks say something about arithmetic rather than about coin flips.
E2ES = {
    "conf_tiny": dict(enc="tiny", img_seed=166, dec_seed=50, regions=5, amp=3.0, noise=0.02, proj="h64", tokens=3, hidden_seed=8,
                      resize=(512, 512), original=(224, 224)),
}


def e2e_inputs(c):
    """(image [1,3,S,S] bf16-representable fp32, LLM hidden states at the [SEG]-1 positions [T, H] bf16-representable)."""
    e = SAM_ENCODERS[c["enc"]]
    S, R = e["img"], c["regions"]
    u = synth.uniform01(c["img_seed"], "input.region_sites", 2 * R).reshape(R, 2) * S
    yy, xx = np.mgrid[0:S, 0:S]
    d = (yy[None] + 0.5 - u[:, 0, None, None]) ** 2 + (xx[None] + 0.5 - u[:, 1, None, None]) ** 2
    region = torch.from_numpy(d.argmin(0))
    vals = torch.from_numpy(synth.normal(c["img_seed"], "input.region_values", (R, 3))) * c["amp"]
    x = vals[region].permute(2, 0, 1)[None].contiguous()
    x = x + c["noise"] * torch.from_numpy(synth.normal(c["img_seed"], "input.images", (1, 3, S, S)))
    hid = torch.from_numpy(synth.normal(c["hidden_seed"], "input.seg_hidden", (c["tokens"], PROJECTORS[c["proj"]]["llama_dim"])))
    return _bf16_round(x), _bf16_round(hid)


def e2e_weights(c):
    """(encoder, decoder (conf style), CTP) weights, bf16-representable -- the deployed checkpoint is bf16."""
    w_enc = {k: _bf16_round(v) for k, v in sam_encoder_weights(SAM_ENCODERS[c["enc"]]).items()}
    w_dec = decoder_case_weights(dict(seed=c["dec_seed"], coherent_upscaler=True, bf16_weights=True))
    pc = PROJECTORS[c["proj"]]    # (the CTP half of projector_weights: the MSQP half is 105 M values nobody needs here)
    wt = {k: torch.from_numpy(synth.param(pc["seed"], "text_hidden_fcs.0." + k, sh)) for k, sh in ctp_weight_shapes(pc["llama_dim"]).items()}
    return w_enc, w_dec, {k: _bf16_round(v) for k, v in wt.items()}


# --- [SEG] bookkeeping of walkgptForCausalLM.model_forward / evaluate (model/walkgpt.py:284-306, 406-447, 645-707) -----------------
# rows of token ids with the image placeholder and [SEG] ids in them; `seg` an int or a list (the two forms of seg_token_idx, :284-292);
# `offset`: image -> rows (training layout); `new`: the ids generate() appends per row (evaluate layout)
SEGMASKS = {
    "train_int": dict(mode="train", rows=4, L=14, seg=61, offset=[0, 2, 3, 4], seg_pos=[[3, 9], [13], [], [5, 6, 7]], seed=111),
    "train_list": dict(mode="train", rows=3, L=12, seg=[61, 62], seg_token_num=2, offset=[0, 1, 3], seg_pos=[[4, 5], [2, 3, 9, 10], [6, 7]],
                       seed=112),
    "eval": dict(mode="eval", rows=3, L=10, seg=61, pad=[0, 3, 1], new=[[7, 61, 9, 61, 2], [61, 2], [8, 8, 2]], seed=113),
}


def segmask_inputs(c):
    """input_ids [rows, L] int64 in [3, 60] with -200 at position 1 and the case's [SEG] ids placed (list form: ids alternate);
    eval: rows right-padded with 0 (`pad` positions), as evaluate() strips them (:621-625)."""
    rows, L = c["rows"], c["L"]
    ids = (synth.uniform01(c["seed"], "input.ids", rows * L) * 57).astype(np.int64).reshape(rows, L) + 3
    ids[:, 1] = -200
    segs = c["seg"] if isinstance(c["seg"], list) else [c["seg"]]
    for r, pos in enumerate(c.get("seg_pos", [[]] * rows)):
        for j, q in enumerate(pos):
            ids[r, q] = segs[j % len(segs)]
    for r, n in enumerate(c.get("pad", [0] * rows)):
        if n:
            ids[r, L - n:] = 0
    return torch.from_numpy(ids)


def splice_inputs(c):
    """input_ids [rows, L] (one -200 placeholder per row), attention_mask (last position of row 0 masked), labels, image features
    [rows, 36, hidden] as the projector hands them over, embedding table [vocab, hidden], optional ViT patch mask [rows, 256]."""
    rows, L, H = c["rows"], c["L"], c["hidden"]
    ids = (synth.uniform01(c["seed"], "input.ids", rows * L) * (c["vocab"] - 1)).astype(np.int64).reshape(rows, L) + 1
    for r, pos in enumerate(c["image_pos"]):
        ids[r, pos] = -200
    ids = torch.from_numpy(ids)
    mask = torch.ones(rows, L, dtype=torch.bool)
    mask[0, L - 1] = False
    labels = None
    if not c.get("no_labels"):
        labels = ids.clone()
        labels[:, :2] = -100
    feats = torch.from_numpy(synth.normal(c["seed"], "input.image_features", (rows, 36, H)))
    table = torch.from_numpy(synth.normal(c["seed"], "embed_tokens.weight", (c["vocab"], H)))
    vit = None
    if c.get("vit_mask"):
        vit = torch.from_numpy((synth.uniform01(c["seed"], "input.vit_mask", rows * 256) > 0.3).astype(np.float32)).reshape(rows, 256)
    return ids, mask, labels, feats, table, vit


# ---- the reference's CLIP-wrapper additions (clip_encoder.py:38-55, custom_clip.py:27-38, llava_arch.py:160-193) --------------
CLIPWRAPS = {
    "w56": dict(image=56, sizes=[(56, 56), (30, 44), (14, 56), (41, 13)], old_side=4, new_side=6, dim=8, seed=111),
    "w448": dict(image=448, sizes=[(448, 448), (448, 300), (299, 448)], old_side=24, new_side=32, dim=8, seed=112),
}


def clipwrap_table(c):
    return torch.from_numpy(synth.normal(c["seed"], "position_embedding.weight", (c["old_side"] ** 2 + 1, c["dim"])))
