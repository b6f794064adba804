"""Soak: the same fused step (three streams, as bench.py runs it) N times on the same inputs; every pass must reproduce the first one's CLIP features,
SAM embedding, masks and scores BIT FOR BIT.  The inference path has no atomics, so any difference is an intermittent fault -- e.g. a missing wait
state in front of a hand-placed MFMA (tools/lint_asm_hazards.py checks the ISA statically; this checks the silicon).
   python tools/soak_step.py [--steps 300] [--dtype fp8] [--config C5]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

argv = [a for a in sys.argv[1:]]
n = 300
if "--steps" in argv:
    i = argv.index("--steps")
    n = int(argv[i + 1])
    del argv[i:i + 2]
args = bench.parse(argv + ["--steps", "1", "--warmup", "0"])
dev = torch.device("cuda:0")
model = bench.build_model(args, dev)
inp = bench.make_inputs(args, dev, 0)
side, dec = torch.cuda.Stream(), torch.cuda.Stream()


def step():
    with torch.no_grad():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            feats, _ = model.encode_images_clip(inp["images_clip"], inp["clip_resize_list"])
        emb = model.get_visual_emb_tokens(inp["images"])
        dec.wait_stream(cur)
        emb.record_stream(dec)
        with torch.cuda.stream(dec):
            masks, scores = model.decode_from_hidden_graphed(emb, inp["seg_hidden"], inp["resize_list"], inp["original_size_list"])
            masks, scores = [m.clone() for m in masks], [s.clone() for s in scores]
        cur.wait_stream(side)
        cur.wait_stream(dec)
    torch.cuda.synchronize()
    return [feats.clone(), emb.clone()] + masks + scores


ref = step()
bad = 0
for it in range(n):
    out = step()
    diff = [i for i, (a, b) in enumerate(zip(ref, out)) if not torch.equal(a, b)]
    if diff:
        bad += 1
        print("pass %d: tensors %s differ from the first pass" % (it, diff), flush=True)
    if (it + 1) % 100 == 0:
        print("%d passes, %d with differences" % (it + 1, bad), flush=True)
print("soak %s: %d passes of %s, %d with differences" % ("FAILED" if bad else "ok", n, bench.config_name(args, 1), bad))
sys.exit(1 if bad else 0)
