"""Experiment: let the CLIP tower of step k+1 start as soon as the side stream is free (no wait for the main stream at the start of a step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = bench.parse([])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
model = bench.build_model(args, dev)
inp = bench.make_inputs(args, dev, 0)
side, dec = torch.cuda.Stream(), torch.cuda.Stream()

def step(side_waits_main, main_waits_side_at_end):
    with torch.no_grad():
        cur = torch.cuda.current_stream()
        if side_waits_main:
            side.wait_stream(cur)
        with torch.cuda.stream(side):
            feats, _ = model.encode_images_clip(inp["images_clip"], inp["clip_resize_list"])
        emb = model.get_visual_emb_tokens(inp["images"])
        dec.wait_stream(cur)
        with torch.cuda.stream(dec):
            masks, scores = model.decode_from_hidden_graphed(emb, inp["seg_hidden"], inp["resize_list"], inp["original_size_list"])
        if main_waits_side_at_end:
            cur.wait_stream(side)
    return feats, masks, scores

def timeit(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for rep in range(2):
    print("as the bench does (side waits for main at step start, main for side at step end): %.3f ms/step" % timeit(lambda: step(True, True)), flush=True)
    print("CLIP runs ahead by up to one step (no wait at step start): %.3f ms/step" % timeit(lambda: step(False, True)), flush=True)
    print("CLIP free-running (joined only by the final synchronize): %.3f ms/step" % timeit(lambda: step(False, False)), flush=True)
