"""Where a tile's life goes in the persistent 256 x 256 GEMM, per shape of the C2 step, and what its operand fetch achieves against the chip's
measured per-CU fill rates (MI355X_MICROARCH.md).  Needs a library built with -DWG_GEMM_STAMP (tools/build_variant.py gstamp -DWG_GEMM_STAMP),
run with WG_LIB=walkgpt_amd/_abl/lib_gstamp.so.  Phase sums are accumulated in scalar registers over all tiles of a workgroup (no store in
the loops); the stamp drains the LDS queue at the three phase boundaries only.

    python tools/gemm_phase_stamps.py > profiles/r04_gemm_phases.md"""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops, _lib
if os.environ.get("WG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["WG_LIB"])
dev = torch.device("cuda:0")
lib = _lib.lib()
lib.wg_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(1024, device=dev, dtype=torch.int32)
assert lib.wg_debug_gemm_stamps(buf.data_ptr()) == 0


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("SAM qkv (LN fold)", 32768, 2304, 768, "ln"), ("SAM lin1 (LN fold + GELU)", 32768, 3072, 768, "ln_gelu"),
          ("SAM proj (+ residual, row sums)", 32768, 768, 768, "res"), ("SAM lin2 (+ residual, row sums)", 32768, 768, 3072, "res"),
          ("CLIP q,k,v (LN fold)", 8200, 3072, 1024, "ln"), ("CLIP fc1 (LN fold + quick-GELU)", 8200, 4096, 1024, "ln_qgelu"),
          ("CLIP out_proj (+ residual, row sums)", 8200, 1024, 1024, "res"), ("CLIP fc2 (+ residual, row sums)", 8200, 1024, 4096, "res"),
          ("8192^3 (bias only)", 8192, 8192, 8192, "bias")]
if "--vit-h" in sys.argv:      # config C3 / C5's SAM ViT-H at bs = 32 (131 072 token rows)
    shapes = [("ViT-H qkv (LN fold)", 131072, 3840, 1280, "ln"), ("ViT-H lin1 (LN fold + GELU)", 131072, 5120, 1280, "ln_gelu"),
              ("ViT-H proj (+ residual, row sums)", 131072, 1280, 1280, "res"), ("ViT-H lin2 (+ residual, row sums)", 131072, 1280, 5120, "res")]
print("# Persistent 256 x 256 GEMM: phases of a tile's life and operand fetch per shape (round 4)\n")
print("`tools/gemm_phase_stamps.py` on a `-DWG_GEMM_STAMP` build: `s_memtime` sums over every tile of workgroups 0-31, waves 0 and 4 (the two halves of "
      "the ping-pong), in scalar registers; in-kernel clock = core cycles / 100 MHz ticks over the workgroup's life after >= 1 s of back-to-back "
      "launches on random data.  Fetch = the A and W slabs a 256 x 256 tile pulls into LDS by LDS-DMA ((256 + 256) x K x 2 bytes) over the tile's "
      "main-loop cycles.  Guide figures for comparison (MI355X_MICROARCH.md): one CU gathering L2-resident rows into LDS 66-73 GB/s (29-30 B/clk at "
      "2.4 GHz), Infinity-Cache-resident 33.5 GB/s, HBM 23-24 GB/s; a 256 x 256 bf16 tile needs 32 B/clk/CU of operand fetch to keep the matrix pipe "
      "busy (64 KiB per 2048 MFMA cycles).\n")
print("| shape | M x N x K | us (stamped build) | TFLOP/s | clock GHz | tiles / WG | first-slab wait | main loop | epilogue | cycles / slab (2048 = MFMA-bound) | "
      "fetch B/clk/CU in the loop | fetch GB/s/CU | epilogue cycles / tile: next-tile setup + first-slab requests, rows 0-63 compute + LDS writes, read-back + stores, rows 64-127 compute, stores |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16)
    if epi.startswith("ln"):
        gam, bet = (1 + 0.1 * torch.randn(K, device=dev)).to(torch.bfloat16), (0.1 * torch.randn(K, device=dev)).to(torch.bfloat16)
        fold = ops.fold_layernorm(gam, bet, w, b)
        # the producer's partial sums, as the step has them: x comes out of a GEMM that leaves its row sums
        x = ops.linear(torch.randn(M, K, device=dev).to(torch.bfloat16), (torch.randn(K, K, device=dev) / K ** 0.5).to(torch.bfloat16), row_partials=True)
        act = {"ln": ops.ACT_NONE, "ln_gelu": ops.ACT_GELU, "ln_qgelu": ops.ACT_QUICK_GELU}[epi]
        fn = lambda: ops.ln_linear(x, fold, 1e-6, act=act)
    elif epi == "res":
        fn = lambda: ops.linear(a, w, b, residual=r, row_partials=True)
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.linear(a, w, b, out=out)
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(20):
            fn()
    torch.cuda.synchronize()
    buf.zero_()
    us = t_us(fn)
    allb = buf.cpu().numpy().astype("int64") & 0xffffffff
    raw = allb[:512].reshape(32, 2, 8)
    epi = allb[512:1024].reshape(32, 2, 8)[raw[:, :, 3] > 0]
    raw = raw[raw[:, :, 3] > 0]
    ph = raw[:, :3].sum(0).astype(float)
    tiles = raw[:, 3].sum()
    clk = (raw[:, 4] / raw[:, 5]).mean() * 0.1
    tot = ph.sum()
    nk = K // 64
    per_slab = ph[1] / tiles / nk
    fetch = 512.0 * K * 2 / (ph[1] / tiles)
    ep = epi[:, :5].sum(0) / tiles
    print("| %s | %d x %d x %d | %.1f | %.0f | %.2f | %.1f | %.1f %% | %.1f %% | %.1f %% | %.0f | %.1f | %.1f | %s (of %.0f) |" % (
        name, M, N, K, us, 2.0 * M * N * K / us / 1e6, clk, tiles / len(raw), 100 * ph[0] / tot, 100 * ph[1] / tot, 100 * ph[2] / tot, per_slab, fetch,
        fetch * clk, " + ".join("%.0f" % v for v in ep), ph[2] / tiles))
