"""Free-running persistent GEMM (tile 17, tools/micro/gemm_fr.hip; needs a -DWG_GEMM_FR build: tools/build_variant.py fr -DWG_GEMM_FR, WG_LIB=walkgpt_amd/_abl/lib_fr.so) against the ping-pong persistent kernel (tile 16): bit-exactness, then timing on the hot-path shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops, _lib
if os.environ.get("WG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["WG_LIB"])
QUICK = os.environ.get("QUICK") == "1"
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(1)
checks = [] if QUICK else [(1777, 520, 256, "bias"), (256, 256, 128, "none"), (8200, 1024, 1024, "bias"), (4096, 2304, 768, "gelu"), (3000, 776, 192, "qgelu"), (70000, 256, 128, "bias"),
          (512, 4096, 4096, "bias")]
ok = True
for (M, N, K, epi) in checks:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    kw = {"none": {}, "bias": dict(bias=b), "gelu": dict(bias=b, act=ops.ACT_GELU), "qgelu": dict(bias=b, act=ops.ACT_QUICK_GELU)}[epi]
    o16 = ops.linear(a, w, tile=16, **kw)
    o17 = torch.full_like(o16, float("nan"))
    ops.linear(a, w, out=o17, tile=17, **kw)
    ref = a.float() @ w.float().t() + (b.float() if epi != "none" else 0)
    same = torch.equal(o16, o17)
    bad = (o16.float() - o17.float()).abs()
    print("check M=%d N=%d K=%d %-5s bit-exact %s  max diff %.4g  nan %d  | vs fp32 ref (pre-act) %.4g" % (M, N, K, epi, same, bad.nan_to_num(1e9).max().item(), int(torch.isnan(o17.float()).sum()),
          (o17.float() - ref).abs().max().item() if epi in ("none", "bias") else -1), flush=True)
    ok &= same
    if not same:
        idx = torch.nonzero(bad.nan_to_num(1e9) > 0)[:5]
        print("  first diffs at", idx.tolist(), flush=True)
print("ALL BIT-EXACT" if ok else "MISMATCH", flush=True)
shapes = [("sam qkv", 32768, 2304, 768, "bias"), ("sam lin1", 32768, 3072, 768, "gelu"), ("sam proj", 32768, 768, 768, "bias"), ("sam lin2", 32768, 768, 3072, "bias"),
          ("clip qkv", 8200, 3072, 1024, "bias"), ("clip fc1", 8200, 4096, 1024, "qgelu"), ("clip out", 8200, 1024, 1024, "bias"), ("clip fc2", 8200, 1024, 4096, "bias"),
          ("8k", 8192, 8192, 8192, "none"), ("4k", 4096, 4096, 4096, "none")]
for rnd in range(1 if QUICK else 2):
    for (name, M, N, K, epi) in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw = {"none": {}, "bias": dict(bias=b), "gelu": dict(bias=b, act=ops.ACT_GELU), "qgelu": dict(bias=b, act=ops.ACT_QUICK_GELU)}[epi]
        res = []
        for tile in (16, 17, 16, 17):
            ms = t(lambda: ops.linear(a, w, out=out, tile=tile, **kw))
            res.append("t%d %6.1f us %5.0f TF" % (tile, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
        print("%-9s M=%d N=%d K=%d %-5s | %s" % (name, M, N, K, epi, " | ".join(res)), flush=True)
