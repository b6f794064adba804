import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops, _lib
if os.environ.get("WG_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["WG_LIB"])
dev = torch.device("cuda:0")
def ref_mha(q, k, v, heads, scale):
    B, Lq, D = q.shape
    hd = D // heads
    qh = q.float().reshape(B, Lq, heads, hd).transpose(1, 2)
    kh = k.float().reshape(B, -1, heads, hd).transpose(1, 2)
    vh = v.float().reshape(B, -1, heads, hd).transpose(1, 2)
    a = (qh * scale) @ kh.transpose(-1, -2)
    return (a.softmax(-1) @ vh).transpose(1, 2).reshape(B, Lq, D)
heads = 2
D = heads * 64
g = torch.Generator().manual_seed(5)
qkv = torch.randn(4096, 3 * D, generator=g).to(torch.bfloat16)
q, k, v = qkv[None, :, :D], qkv[None, :, D:2 * D], qkv[None, :, 2 * D:]
ref = ref_mha(q, k, v, heads, 0.125)[0]
z = torch.zeros(127, 64, dtype=torch.bfloat16)
bias = torch.zeros(3 * D, dtype=torch.bfloat16)
for rep in range(3):
    og = ops.sam_attention(qkv.to(dev), bias.to(dev), z.to(dev), z.to(dev), 1, 64, 64, heads).float().cpu()
    qd = qkv.to(dev)[None]
    op = ops.mha(qd[..., :D], qd[..., D:2 * D], qd[..., 2 * D:], heads, 0.125, small=False).float().cpu()[0]
    eg, ep = (og - ref).abs().amax(-1), (op - ref).abs().amax(-1)
    print("grid kernel (zero tables): max err %.4f bad %d | plain kernel: max err %.4f bad %d first bad %s" % (
        eg.max(), int((eg > 0.03).sum()), ep.max(), int((ep > 0.03).sum()), (ep > 0.03).nonzero().flatten()[:12].tolist()))
    bad = (ep > 0.03).nonzero().flatten()
    if len(bad):
        i = int(bad[0])
        print("  query", i, "plain out", op[i, :6].tolist(), "ref", ref[i, :6].tolist(), "ratio", (op[i, :6] / ref[i, :6]).tolist())
