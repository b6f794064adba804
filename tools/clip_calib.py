"""Where the CLIP tower's error against fp32 comes from: HIP tower vs the stand-in's fp32 run, next to the stand-in's own bf16 run
(tests/golden/clipcal_*.npz), on the selected features, the -11 features and hidden-state taps; and on text logits through a
projector + a small fp32 LM.   python tools/clip_calib.py [tiny|vit_l_448]"""
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tests.golden import cases
from tests.test_toplevel import TinyLM, H
from walkgpt_amd.clip_encoder import CLIPVisionTower

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


for name in (sys.argv[1:] or ["tiny", "vit_l_448"]):
    c = cases.CLIP_CALIBS[name]
    gold = np.load(os.path.join(os.path.dirname(cases.__file__), "clipcal_%s.npz" % name))
    cfg = dict(hidden_size=c["dim"], intermediate_size=4 * c["dim"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
               image_size=c["img"], patch_size=14, layer_norm_eps=1e-5)
    args = SimpleNamespace(mm_vision_select_layer=c["select_layer"], pad_train_clip_images=True, resize_vision_tower=True,
                           resize_vision_tower_size=c["img"])
    tower = CLIPVisionTower("synthetic", args, config=cfg)
    w = cases.clip_weights(c)
    tower.vision_tower.load_state_dict(w, strict=True)
    tower.to(dev).bfloat16()
    x, km = cases.clip_calib_inputs(c)
    st, ts = c["stride"], c["tap_stride"]
    with torch.no_grad():
        want = [c["select_layer"], -11] + list(c["taps"])
        hs = tower.vision_tower.vision_model.hidden_states(x.to(dev, torch.bfloat16), km.to(dev), want)
    sel = hs[c["select_layer"]][:, 1::st].float().cpu().numpy()
    pre = hs[-11][:, 1::st].float().cpu().numpy()
    print("%s: sel  HIP %.4f  ref-bf16 %.4f | pre  HIP %.4f  ref-bf16 %.4f" % (
        name, rel(sel, gold["sel"]), rel(gold["sel_bf16"], gold["sel"]), rel(pre, gold["pre"]), rel(gold["pre_bf16"], gold["pre"])))
    for t in c["taps"]:
        h = hs[t][:, ::ts].float().cpu().numpy()
        print("   h%-2d HIP %.4f  ref-bf16 %.4f" % (t, rel(h, gold["h%d" % t]), rel(gold["h%d_bf16" % t], gold["h%d" % t])))
    g = torch.Generator().manual_seed(77)
    proj = torch.randn(H, c["dim"], generator=g) / c["dim"] ** 0.5
    lm = TinyLM()

    def logits(f):
        with torch.no_grad():
            return lm(inputs_embeds=torch.from_numpy(np.asarray(f)).float() @ proj.t(), output_hidden_states=True).logits

    l32 = logits(gold["sel"])
    for tag, f in (("HIP", sel), ("ref-bf16", gold["sel_bf16"]), ("fp32 rounded once", torch.from_numpy(gold["sel"]).bfloat16().float().numpy())):
        l = logits(f)
        print("   logits (std %.2f) %-18s max|d| %.3e  rel L2 %.4f" % (float(l32.std()), tag, float((l - l32).abs().max()), float((l - l32).norm() / l32.norm())))
