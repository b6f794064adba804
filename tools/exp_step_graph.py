"""Experiment: the whole C2 step (CLIP tower | SAM encoder | decode chain on three streams) captured in ONE HIP graph vs launched from Python."""
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

def step(join_dec):
    with torch.no_grad():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            feats, _ = model.encode_images_clip(inp["images_clip"], inp["clip_resize_list"])
        emb = model.get_visual_emb_tokens(inp["images"])
        dec.wait_stream(cur)
        with torch.cuda.stream(dec):
            masks, scores = model.decode_from_hidden(emb, inp["seg_hidden"], inp["resize_list"], inp["original_size_list"])
        cur.wait_stream(side)
        if join_dec:
            cur.wait_stream(dec)
    return feats, masks, scores

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("eager, decode overlapping the next step: %.3f ms/step" % timeit(lambda: step(False)), flush=True)
print("eager, decode joined at the end of its step: %.3f ms/step" % timeit(lambda: step(True)), flush=True)
for _ in range(2): step(True)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step(True)
print("one graph per step: %.3f ms/step" % timeit(g.replay), flush=True)
# two steps per graph: decode of the first overlaps the encoders of the second
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    step(False); out2 = step(True)
print("two steps per graph: %.3f ms/step" % (timeit(g2.replay) / 2), flush=True)
