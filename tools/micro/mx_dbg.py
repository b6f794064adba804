import torch, sys
sys.path.insert(0, '/root/repo')
from walkgpt_amd import ops
dev = torch.device('cuda:0')
F8 = torch.float8_e4m3fn
def q(t): return t.to(F8).view(torch.uint8).contiguous()
M, N, K = 256, 256, 128
a = torch.ones(M, K); w = torch.zeros(N, K)
sw = torch.ones(N)
def run(planes, w):
    return ops.linear_fp8(q(a).to(dev), planes.to(dev), q(w).to(dev), sw.to(dev)).float().cpu()
# 1) uniform exponent +1: expect 2x
w1 = torch.ones(N, K)
pl = torch.full((K // 32, 256), 128, dtype=torch.uint8)
o = run(pl, w1); print("uniform +1:", o.unique())
# 2) per K block: only block kb has weights; scale differs per plane kb: 127+kb
for kb in range(4):
    wk = torch.zeros(N, K); wk[:, kb * 32:(kb + 1) * 32] = 1
    pl = torch.stack([torch.full((256,), 127 + j, dtype=torch.uint8) for j in range(4)])
    o = run(pl, wk); print("K block", kb, "-> out/32 =", (o / 32).unique())
# 3) per row: plane position p holds exponent p%8 (in-group) ; see which position each row reads
pl = torch.zeros(4, 256, dtype=torch.uint8)
pos = torch.arange(256)
for code, f in (("pos%8", pos % 8), ("(pos//8)%16", (pos // 8) % 16), ("pos//128", pos // 128)):
    pl[:] = (127 + f).to(torch.uint8)[None]
    o = run(pl, w1)
    ex = torch.log2(o[:, 0] / 128).round().int()
    print(code, "rows 0..40:", ex[:40].tolist()); print("   rows 120..136", ex[120:136].tolist())
