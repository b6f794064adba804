"""Experiment: the prompt-encoder / mask-decoder / postprocess chain replayed from a captured HIP graph vs launched eagerly."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
class A: pass
args = A(); args.sam = "vit_b"; args.llm_hidden = 4096; args.batch = 8; args.seg_tokens = int(os.environ.get("T", "1")); args.with_msqp = False
dev = torch.device("cuda:0")
model = bench.build_model(args, dev)
inp = bench.make_inputs(args, dev, 0)
with torch.no_grad():
    emb = model.get_visual_emb_tokens(inp["images"])
    def run():
        return model.decode_from_hidden(emb, inp["seg_hidden"], inp["resize_list"], inp["original_size_list"])
    for _ in range(3): run()
    torch.cuda.synchronize()
    def timeit(fn, n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    print("eager  %.3f ms per batch" % timeit(run))
    def run_g():
        return model.decode_from_hidden_graphed(emb, inp["seg_hidden"], inp["resize_list"], inp["original_size_list"])
    masks, scores = run_g()
    torch.cuda.synchronize()
    ref_m, ref_s = run()
    run_g(); torch.cuda.synchronize()
    print("graph == eager:", all(torch.equal(a, b) for a, b in zip(masks, ref_m)), all(torch.equal(a, b) for a, b in zip(scores, ref_s)))
    print("graph  %.3f ms per batch (incl. the input copies)" % timeit(run_g))
