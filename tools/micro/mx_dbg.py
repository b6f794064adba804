# Which lane's scale byte does v_mfma_scale_f32_16x16x128_f8f6f4 apply to which K range / row?  All-ones operands, scales that differ by
# plane (K block) or by position inside the 128-row group; prints what the GEMM returns.  (The run that fixed the fragment K order in
# gemm.hip: with 32 contiguous bytes per lane, "K block 0..3" printed 2.5 2.5 5 5 instead of 1 2 4 8.)
import sys
import torch
sys.path.insert(0, '/root/repo')
from walkgpt_amd import ops
dev = torch.device('cuda:0')
F8 = torch.float8_e4m3fn
def q(t): return t.to(F8).view(torch.uint8).contiguous()
M, N, K = 256, 256, 128
a = torch.ones(M, K)
pw = torch.full((K // 32, 256), 127, dtype=torch.uint8)
def run(planes, w):
    return ops.linear_mxfp8((q(a).to(dev), planes.to(dev)), {"q": q(w).to(dev), "mx": pw.to(dev)}).float().cpu()
w1 = torch.ones(N, K)
pl = torch.full((K // 32, 256), 128, dtype=torch.uint8)
print("uniform +1:", run(pl, w1).unique())
for kb in range(4):
    wk = torch.zeros(N, K); wk[:, kb * 32:(kb + 1) * 32] = 1
    pl = torch.stack([torch.full((256,), 127 + j, dtype=torch.uint8) for j in range(4)])
    print("K block", kb, "-> out/32 =", (run(pl, wk) / 32).unique())
pl = torch.zeros(4, 256, dtype=torch.uint8)
pos = torch.arange(256)
for code, f in (("pos%8", pos % 8), ("(pos//8)%16", (pos // 8) % 16), ("pos//128", pos // 128)):
    pl[:] = (127 + f).to(torch.uint8)[None]
    ex = torch.log2(run(pl, w1)[:, 0] / 128).round().int()
    print(code, "rows 0..40:", ex[:40].tolist()); print("   rows 120..136", ex[120:136].tolist())
