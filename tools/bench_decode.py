"""The decode chain alone (CTP -> prompt encoder -> mask decoder -> postprocess -> score) on synthetic inputs: eager and graph-replay
latency for B images x T prompts.  Under `rocprofv3 --kernel-trace --stats -- python3 tools/bench_decode.py` it gives the per-kernel
times of the chain."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--seg-tokens", type=int, default=1)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
from walkgpt_amd.walkgpt import WalkGPTGrounding
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = WalkGPTGrounding(sam="vit_b", llm_hidden=4096, with_clip=False).to(dev).bfloat16().eval()
pe = m.visual_model.prompt_encoder.pe_layer
pe.positional_encoding_gaussian_matrix.data = pe.positional_encoding_gaussian_matrix.data.float()
B, T = args.batch, args.seg_tokens
emb = torch.randn(B, 4096, 256, device=dev).to(torch.bfloat16)
hid = [torch.randn(T, 4096, device=dev).to(torch.bfloat16) for _ in range(B)]
rs, osz = [(1024, 1024)] * B, [(448, 448)] * B
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3
with torch.no_grad():
    s_emb, s_hid = m.decode_graph_inputs(emb, hid, rs, osz)      # inputs resident in the graph's own buffers: no staging copies
    s_emb.copy_(emb)
    for d_, h_ in zip(s_hid, hid):
        d_.copy_(h_)
    res = [t(lambda: m.decode_from_hidden_graphed(s_emb, s_hid, rs, osz)) for _ in range(3)]
    print("B=%d T=%d eager %.0f us  graph (staged inputs) %.0f us  graph (resident inputs) %s us" % (
        B, T, t(lambda: m.decode_from_hidden(emb, hid, rs, osz)), t(lambda: m.decode_from_hidden_graphed(emb, hid, rs, osz)),
        " / ".join("%.1f" % r for r in res)), flush=True)
