"""Where a slab of the free-running persistent GEMM (tools/micro/gemm_fr.hip) goes: needs a library built with -DWG_GEMM_FR -DWG_GEMM_STAMP
(python tools/build_variant.py gstamp -DWG_GEMM_STAMP), run with WG_LIB=walkgpt_amd/_abl/lib_gstamp.so."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops, _lib
if os.environ.get("WG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["WG_LIB"])
dev = torch.device("cuda:0")
lib = _lib.lib()
lib.wg_debug_gemm_fr_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(32 * 8 * 8, device=dev, dtype=torch.int32)
assert lib.wg_debug_gemm_fr_stamps(buf.data_ptr()) == 0
shapes = [("sam qkv", 32768, 2304, 768, "bias"), ("sam lin1", 32768, 3072, 768, "gelu"), ("sam proj", 32768, 768, 768, "bias"), ("sam lin2", 32768, 768, 3072, "bias"),
          ("clip qkv", 8200, 3072, 1024, "bias"), ("clip fc2", 8200, 1024, 4096, "bias"), ("8k", 8192, 8192, 8192, "none")]
print("| shape | us | TF/s | GHz | cycles / slab | load half-phase (issue + reads) grp 0 / 1 | wait at its barrier grp 0 / 1 | matrix half-phase grp 0 / 1 | wait at its barrier grp 0 / 1 | two stamps back to back (per slab) grp 0 / 1 |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = {"none": {}, "bias": dict(bias=b), "gelu": dict(bias=b, act=ops.ACT_GELU)}[epi]
    fn = lambda: ops.linear(a, w, out=out, tile=17, **kw)
    t0 = time.time()
    while time.time() - t0 < 0.5:
        for _ in range(20):
            fn()
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    r = (buf.cpu().numpy().astype("int64") & 0xffffffff).reshape(32, 8, 8).astype(float)
    r = r[r[:, 0, 5] > 0]
    slabs = r[:, :, 5]
    tiles = slabs / (K // 64)
    def g(col, per):      # mean per group of four waves
        x = (r[:, :, col] / per).mean(0)
        return "%.0f / %.0f" % (x[:4].mean(), x[4:].mean())
    tot = ((r[:, :, 0] + r[:, :, 1] + r[:, :, 2] + r[:, :, 3]) / slabs).mean()
    clk = (r[:, :, 6] / r[:, :, 7]).mean() * 0.1
    print("| %s %dx%dx%d | %.1f | %.0f | %.2f | %.0f | %s | %s | %s | %s | %s |" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6, clk, tot, g(0, slabs), g(1, slabs), g(2, slabs), g(3, slabs),
          g(4, slabs)), flush=True)
