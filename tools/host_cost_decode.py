"""Host cost of one decode_from_hidden_graphed call (the graph key is rebuilt per call) against the replay's GPU time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd.walkgpt import WalkGPTGrounding
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = WalkGPTGrounding(sam="vit_b", llm_hidden=4096, with_clip=False).to(dev).bfloat16().eval()
pe = m.visual_model.prompt_encoder.pe_layer
pe.positional_encoding_gaussian_matrix.data = pe.positional_encoding_gaussian_matrix.data.float()
emb = torch.randn(1, 4096, 256, device=dev).to(torch.bfloat16)
hid = [torch.randn(1, 4096, device=dev).to(torch.bfloat16)]
rs, osz = [(1024, 1024)], [(448, 448)]
with torch.no_grad():
    s_emb, s_hid = m.decode_graph_inputs(emb, hid, rs, osz)
    for _ in range(20):
        m.decode_from_hidden_graphed(s_emb, s_hid, rs, osz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        m.decode_from_hidden_graphed(s_emb, s_hid, rs, osz)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("host per call %.1f us; with the GPU drained %.1f us per call" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
