#!/usr/bin/env python3
"""bench.py -- images/sec of WalkGPT's grounded-segmentation forward path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input on each GPU:
  bs images/GPU of 448x448  ->  images_clip [bs,3,448,448] and SAM input [bs,3,1024,1024] (bf16, resident in HBM)
  CLIP ViT-L/14 tower (24 layers, 1025 tokens)  +  SAM image encoder (+ MSQP for C3/C5)
  + CTP on T [SEG] hidden states per image + prompt encoder + two-way mask decoder + fused postprocess
  (+ one RCCL all-gather of the mask logits when world_size > 1).
Configs (--config sets the defaults, explicit flags override, the label is derived from what actually ran):
  C2 (default)  bs=8/GPU, SAM ViT-B, T=1              -- the configuration the BASELINE metric is quoted on
  C3            bs=32/GPU, SAM ViT-H + MSQP, T=14     -- (the LLM between MSQP and CTP is stock PyTorch and is not run)
  C4            bs=32/GPU on 8 GPUs (global 256), model of C2, RCCL all-gather of the mask logits
  C5            bs=8/GPU, SAM ViT-H + MSQP, T=14, 1024x1024 originals (fp8 GEMMs: --dtype fp8)

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8            # spawns the 8 ranks itself (torch.distributed.run on 127.0.0.1) before touching a GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the dominant kernel (most GEMM time in a step): algorithmic FLOPs of all its launches in one step / the sum of
                their durations, timed with HIP events on the launch stream in an instrumented, serialised (single-stream) step
                that runs right after the timed region (same process, same buffers).  e2e_* = whole step.
  cpu_baseline  the CPU oracle (oracle/, fp32 PyTorch restatement pinned to the reference) timed on the host cores on a bounded
                sample (one image of the batch: 1 warm-up + 3 timed runs), rank 0 at N=1 only.
  secondary     (default C2 run at N=1 only) the other single-GPU configurations -- C3, C2 at T=14, C5 with fp8 -- as short passes in fresh
                processes after the headline is final: [{config, images_per_s, ms_per_step, e2e_frac, steps}].  --no-secondary skips them.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per image (2*MAC, matmul/conv only) -- BASELINE.md section 2, counted on the reference modules
GF_CLIP_L_448 = 723.6
GF_DECODE_PER_TOKEN = 3.61
GF_CTP_PER_TOKEN = 0.00446
GF_MSQP = 50.8
GF_SAM = {"vit_b": 972.1, "vit_l": 2985.7, "vit_h": 5961.1}
PEAK_TF = {"bf16": 2500.0, "fp8": 5000.0}  # MI355X_MICROARCH.md: ~2.5 PF dense bf16, ~5 PF dense fp8 (block-scaled MFMA)

CONFIGS = {
    "C2": dict(batch=8, sam="vit_b", seg_tokens=1, with_msqp=False, original=448),
    "C3": dict(batch=32, sam="vit_h", seg_tokens=14, with_msqp=True, original=448),
    "C4": dict(batch=32, sam="vit_b", seg_tokens=1, with_msqp=False, original=448),
    "C5": dict(batch=8, sam="vit_h", seg_tokens=14, with_msqp=True, original=1024),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step")
    ap.add_argument("--seg-tokens", type=int, default=None, help="[SEG] tokens per image (reference default --seg_token_num=1)")
    ap.add_argument("--sam", default=None, choices=sorted(GF_SAM))
    ap.add_argument("--original", type=int, default=None, help="side of the original image the masks are resampled to")
    ap.add_argument("--llm-hidden", type=int, default=4096)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"], help="GEMM operand type of the SAM encoder blocks (the CLIP tower stays bf16)")
    ap.add_argument("--fp8-clip", action="store_true", help="NOT config C5: with --dtype fp8 also run the CLIP tower's GEMMs on fp8 operands "
                                                            "(its features, which feed the text logits, move 8 %% from fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-b8", action="store_true", help="time the CPU oracle on a batch of 8 also for the ViT-L / ViT-H encoders (minutes)")
    ap.add_argument("--no-cpu-baseline-b8", action="store_true", help="skip the B = 8 pass of the CPU baseline (default: run it once with SAM ViT-B, ~45 s)")
    ap.add_argument("--with-msqp", action="store_true", default=None, help="also run the Multi-Scale Query Projector on the SAM tokens")
    ap.add_argument("--tail-tiles", action="store_true", help="allow the tail-absorbing 128x128 GEMM tiles (wins with --single-stream)")
    ap.add_argument("--side-priority", type=int, default=0, help="HIP priority of the CLIP stream (-1 = high)")
    ap.add_argument("--dec-priority", type=int, default=0, help="HIP priority of the decode stream (-1 = high)")
    ap.add_argument("--clip-skip-unused-layer", action="store_true",
                    help="NOT the headline: stop the CLIP tower after the last hidden state the path reads (hidden_states[-2]); the reference runs "
                         "layer 24 and discards it (clip_encoder.py:77-93).  Same outputs, 1/24 of the tower less; the FLOP count follows")
    ap.add_argument("--single-stream", action="store_true", help="run the CLIP tower and the SAM branch back to back")
    ap.add_argument("--main-stream-created", action="store_true", help="experiment: the SAM branch on a stream of its own instead of the default stream")
    ap.add_argument("--attn-pipe-mode", type=int, default=None, help="experiment: wg_attn_pipe_mode (0 never, 1 SAM global attention only = default, 2 also plain attention: CLIP)")
    ap.add_argument("--clip-split", type=int, default=1, help="experiment: the CLIP tower as this many independent sub-batches, each on a stream of its own")
    ap.add_argument("--sam-split", type=int, default=1, help="the SAM encoder as this many independent slices of the batch, each on a stream of its own (WalkGPT.get_visual_emb_tokens(sub_batches=))")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-decode-graph", action="store_true", help="launch the decode chain eagerly instead of replaying its captured HIP graph")
    ap.add_argument("--steps-only", action="store_true", help="profiling runs: warm-up + timed steps only (no latency / instrumented / CPU passes)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short secondary passes (C3, C2 at T=14, C5 fp8) the default N=1 run appends to its line")
    ap.add_argument("--probe-launch", action="store_true", help="self-test of the rank launch only: gloo rendezvous, no GPU work (tests/test_bench_launch.py)")
    args = ap.parse_args(argv)
    preset = CONFIGS[args.config]
    for k, v in preset.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


def config_name(args, world):
    """Name of what ran, derived from the arguments (never a fixed prefix)."""
    match = [k for k, p in CONFIGS.items() if all(getattr(args, f) == v for f, v in p.items())]
    if "C4" in match:
        return "C4" if world == 8 else "custom (C4's per-GPU share on %d GPU)" % world
    if "C5" in match:
        if args.dtype == "fp8" and args.fp8_clip:
            return "custom (C5 geometry, fp8 GEMMs in BOTH towers)"
        return "C5" if args.dtype == "fp8" else "C5 geometry with bf16 GEMMs"
    if match and args.dtype == "bf16":
        return match[0]
    return "custom"


def gemm_label(args):
    """Which towers run which operand type."""
    if args.dtype != "fp8":
        return "bf16"
    return "fp8 MX (SAM encoder and CLIP tower)" if args.fp8_clip else "fp8 MX in the SAM encoder, bf16 in the CLIP tower and everywhere else"


def workload_label(args, world):
    return ("%s: bs=%d/GPU x %d GPU, %dx%d source images (CLIP input 448^2, SAM input 1024^2), CLIP ViT-L/14 + SAM %s encoder%s + CTP + "
            "prompt encoder + mask decoder + postprocess to %dx%d, T=%d [SEG]/image, %s GEMMs, random-init weights"
            % (config_name(args, world), args.batch, world, args.original, args.original, args.sam, " + MSQP" if args.with_msqp else "", args.original,
               args.original, args.seg_tokens, gemm_label(args))
            + (" [NOT the headline workload: CLIP layer 24, whose output the path discards, is not run]" if args.clip_skip_unused_layer else ""))


def gflop_per_step(args):
    """Algorithmic GFLOP of one step on one GPU (2 * MAC, matmul / conv only, counted on the reference's modules)."""
    gf_clip = GF_CLIP_L_448 if not args.clip_skip_unused_layer else GF_CLIP_L_448 - (GF_CLIP_L_448 - 0.6) / 24.0   # (0.6 GF: patch embedding)
    return args.batch * (gf_clip + GF_SAM[args.sam] + (GF_MSQP if args.with_msqp else 0.0)
                         + args.seg_tokens * (GF_DECODE_PER_TOKEN + GF_CTP_PER_TOKEN))


# Secondary passes of the default single-GPU run: the other single-GPU configurations of BASELINE.json under the same clock as the headline.
# Each is a fresh process of this file in --steps-only form (its own model, its own warm-up, its own synchronise bracket around its timed
# steps); the headline `value` is never computed from them.
SECONDARY = [("C3", ["--config", "C3", "--steps", "3", "--warmup", "1"]),
             ("C2 with T=14 [SEG] tokens per image", ["--config", "C2", "--seg-tokens", "14", "--steps", "10", "--warmup", "3"]),
             ("C5 (fp8 MX GEMMs in the SAM encoder)", ["--config", "C5", "--dtype", "fp8", "--steps", "3", "--warmup", "1"])]


def run_secondary():
    out = []
    for name, flags in SECONDARY:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps-only", "--no-secondary"] + flags
        t0 = time.perf_counter()
        try:
            proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
            line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
            if proc.returncode != 0 or not line:
                out.append({"config": name, "error": "rc %d: %s" % (proc.returncode, proc.stderr.strip()[-300:])})
                continue
            d = json.loads(line[-1])
            out.append({"config": d["config"]["workload"], "images_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                        "warmup": d["warmup"], "dtype": d["dtype"], "e2e_frac": d["e2e_frac"],
                        "e2e_algorithmic_gflop_per_step": d["e2e_algorithmic_gflop_per_step"], "e2e_peak": d["e2e_peak"],
                        "process_s": round(time.perf_counter() - t0, 1)})
        except subprocess.TimeoutExpired:
            out.append({"config": name, "error": "timed out after 420 s"})
        note("secondary %s: %s" % (name, json.dumps(out[-1])[:200]))
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """--gpus N without a torchrun environment: start N fresh rank processes (one per GPU) through torch.distributed.run on
    127.0.0.1 and relay rank 0's JSON line.  Runs BEFORE anything in this process touches the GPU (no exec of a GPU process)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return proc.returncode if proc.returncode != 0 or line is not None else 1


def build_model(args, dev):
    import torch
    from walkgpt_amd.walkgpt import WalkGPTGrounding
    torch.manual_seed(1234)
    model = WalkGPTGrounding(sam=args.sam, llm_hidden=args.llm_hidden, with_clip=True, with_projectors=True)
    if not args.with_msqp:
        del model.out_mm_projector  # MSQP feeds the LLM, which is not part of config C2
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "rel_pos" in n or n.endswith("pos_embed"):
                p.normal_(0.0, 0.02)  # zero-initialised by default; would make the bias terms trivially cheap to get right
    model.to(dev).bfloat16().eval()
    pe = model.visual_model.prompt_encoder.pe_layer
    pe.positional_encoding_gaussian_matrix.data = pe.positional_encoding_gaussian_matrix.data.float()
    if args.dtype == "fp8":
        model.set_gemm_dtype("fp8", clip=args.fp8_clip)
    if args.clip_skip_unused_layer:
        model.vision_tower.run_all_layers = False
    return model


def make_inputs(args, dev, rank):
    import torch
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    B, T = args.batch, args.seg_tokens
    images = torch.randn(B, 3, 1024, 1024, generator=g).to(dev, torch.bfloat16)
    images_clip = torch.randn(B, 3, 448, 448, generator=g).to(dev, torch.bfloat16)
    seg_hidden = [torch.randn(T, args.llm_hidden, generator=g).to(dev, torch.bfloat16) for _ in range(B)]
    return dict(images=images, images_clip=images_clip, seg_hidden=seg_hidden, resize_list=[(1024, 1024)] * B,
                original_size_list=[(args.original, args.original)] * B, clip_resize_list=[(448, 448)] * B)


def usable_cpus():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a container on a big host sees every
    host core in its mask; running that many threads on its 16-core share is slower than 16 threads by an order of magnitude)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def note(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """The oracle's own forward on the host cores, fp32 (SURVEY.md 8d protocol: all cores the process may use, 1 warm-up + 3 timed
    runs at B=1, plus one B=8 pass with SAM ViT-B; for the larger encoders the B=8 pass is on request: --cpu-baseline-b8)."""
    import torch
    from oracle import clip as oclip
    from oracle import projectors as oproj
    from oracle import sam as osam
    from tests.golden import cases
    threads = args.cpu_threads or usable_cpus()
    torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(7)

    def rnd(shapes):
        out = {}
        for k, s in shapes.items():
            t = torch.randn(*s, generator=gen)
            if len(s) > 1:
                t = t / (float(torch.tensor(s[1:]).prod()) ** 0.5)
            elif k.endswith("weight"):
                t = 1 + 0.1 * t
            else:
                t = 0.1 * t
            out[k] = t
        return out

    c = dict(cases.SAM_ENCODERS["vit_b"])
    if args.sam != "vit_b":
        c.update({"vit_l": dict(embed_dim=1024, depth=24, heads=16, global_idx=(5, 11, 17, 23)),
                  "vit_h": dict(embed_dim=1280, depth=32, heads=16, global_idx=(7, 15, 23, 31))}[args.sam])
    w = rnd({k: tuple(v.shape) for k, v in _shape_only_encoder(c).items()})
    w.update(rnd(cases.decoder_weight_shapes()))
    clip_c = dict(dim=1024, heads=16, layers=24, img=448)
    wc = rnd(cases.clip_weight_shapes(clip_c))
    wt = rnd(cases.ctp_weight_shapes(args.llm_hidden))
    wt["text_type"] = wt["text_type"].reshape(1, 1, -1)
    cfg = dict(patch=16, depth=c["depth"], heads=c["heads"], global_idx=c["global_idx"], window=14)
    T = args.seg_tokens

    def run(B):
        x = torch.randn(B, 3, 1024, 1024, generator=gen)
        xc = torch.randn(B, 3, 448, 448, generator=gen)
        hid = torch.randn(B * T, args.llm_hidden, generator=gen)
        with torch.no_grad():
            t0 = time.perf_counter()
            key_mask = oclip.patch_key_mask(B, (448, 448), [(448, 448)] * B)
            oclip.clip_tower(wc, xc, key_mask, -2)
            t1 = time.perf_counter()
            emb = osam.image_encoder(w, x, cfg)
            t2 = time.perf_counter()
            pe = oproj.ctp(wt, hid).reshape(B, T, 1, 256)
            dpe = osam.dense_pe(w, (64, 64))
            for i in range(B):   # the reference decodes image by image (model/walkgpt.py:716-737)
                sparse, dense = osam.prompt_encoder_text(w, pe[i], (64, 64))
                masks, _ = osam.mask_decoder(w, emb[i:i + 1], dpe, sparse, dense)
                post = osam.postprocess_masks(masks, 1024, (1024, 1024), (args.original, args.original))
                osam.mask_score(post[:, 0])
            t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    run(1)                                   # warm-up (thread pool, allocator, oneDNN primitive caches)
    runs = []
    for i in range(3):
        runs.append(run(1))
        note("cpu baseline run %d: %.2f s" % (i + 1, sum(runs[-1])))
    tot = sorted(sum(r) for r in runs)
    med = tot[1]
    best = min(runs, key=sum)
    out = {"value": round(1.0 / med, 4), "unit": "images/s", "cores": threads, "kind": "port", "cpu": cpu_model_name(),
           "sample": "B=1 (one image of the batch), 1 warm-up + 3 timed runs, median %.2f s (min %.2f, max %.2f): CLIP ViT-L@448 %.2fs + SAM %s@1024 "
                     "%.2fs + decode T=%d %.2fs; fp32 oracle, torch CPU, %d threads"
                     % (med, tot[0], tot[2], best[0], args.sam, best[1], T, best[2], threads),
           "mask_decode_ms": round(sorted(r[2] for r in runs)[1] * 1e3, 1)}
    if args.cpu_baseline_b8 or (args.sam == "vit_b" and not args.no_cpu_baseline_b8):
        # SURVEY.md 8d asks for B = 1 and B = 8: with SAM ViT-B one B = 8 pass is about 45 s of host time (ViT-H: minutes, on request only)
        r8 = run(8)
        note("cpu baseline B=8: %.1f s" % sum(r8))
        out["b8"] = {"value": round(8.0 / sum(r8), 4), "unit": "images/s", "sample": "B=8, 1 run of %.1f s" % sum(r8)}
    return out


def _shape_only_encoder(c):
    """Key -> meta tensor of the right shape for a SAM encoder config (values are drawn by the caller)."""
    import torch
    D, p, g = c["embed_dim"], c["patch"], c["img"] // c["patch"]
    hd = D // c["heads"]
    pre = "image_encoder."
    s = {"pos_embed": (1, g, g, D), "patch_embed.proj.weight": (D, 3, p, p), "patch_embed.proj.bias": (D,),
         "neck.0.weight": (256, D, 1, 1), "neck.1.weight": (256,), "neck.1.bias": (256,),
         "neck.2.weight": (256, 256, 3, 3), "neck.3.weight": (256,), "neck.3.bias": (256,)}
    for i in range(c["depth"]):
        S = g if i in c["global_idx"] else c["window"]
        b = "blocks.%d." % i
        s.update({b + "norm1.weight": (D,), b + "norm1.bias": (D,), b + "norm2.weight": (D,), b + "norm2.bias": (D,),
                  b + "attn.rel_pos_h": (2 * S - 1, hd), b + "attn.rel_pos_w": (2 * S - 1, hd),
                  b + "attn.qkv.weight": (3 * D, D), b + "attn.qkv.bias": (3 * D,), b + "attn.proj.weight": (D, D),
                  b + "attn.proj.bias": (D,), b + "mlp.lin1.weight": (4 * D, D), b + "mlp.lin1.bias": (4 * D,),
                  b + "mlp.lin2.weight": (D, 4 * D), b + "mlp.lin2.bias": (D,)})
    return {pre + k: torch.empty(v, device="meta") for k, v in s.items()}


def probe_launch(args, rank, local_rank, world):
    """What the launcher hands each rank, checked without a GPU: env of a one-node torchrun job, a gloo rendezvous on 127.0.0.1 and
    one all-reduce; rank 0 prints a line shaped like the bench line."""
    import torch
    import torch.distributed as dist
    ok = world == args.gpus and 0 <= rank < world and local_rank == rank and os.environ.get("MASTER_ADDR") == "127.0.0.1"
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    ok = ok and float(t.item()) == world * (world + 1) / 2
    # shape bookkeeping of the step's one exchange, as the timed loop calls it: every rank's [B*T, O, O] block of logits, bf16 on the
    # wire, rank r's block at rows [r*B*T, (r+1)*B*T) of the preallocated buffer
    from walkgpt_amd.dist import all_gather_masks_uniform
    n, side = args.batch * args.seg_tokens, 4
    gathered = torch.empty(world * n, side, side, dtype=torch.bfloat16)
    mine = torch.full((n, side, side), rank + 0.5) * (torch.arange(n).view(n, 1, 1) % 2 * 2 - 1)     # +-(rank + 0.5)
    all_gather_masks_uniform(mine, out=gathered, wire_dtype=torch.bfloat16)
    for r in range(world):
        blk = gathered[r * n:(r + 1) * n].float()
        ok = ok and bool((blk.abs() == r + 0.5).all()) and bool(((blk > 0) == (torch.arange(n).view(n, 1, 1) % 2 == 1)).all())
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"metric": "launch-probe", "n_gpus": dist.get_world_size(), "ok": bool(flag.item()),
                          "config": {"workload": workload_label(args, world)}}), flush=True)
    dist.destroy_process_group()
    return 0 if flag.item() else 1


KERNEL_NAMES = {1: "wg_gemm_kernel<128,128,64,2,2,2>", 2: "wg_gemm_kernel<256,256,64,2,2,4>", 3: "wg_gemm_rowwave_kernel", 5: "wg_gemm_skinny_kernel",
                11: "wg_gemm_persist_kernel<128,128,2,2>", 12: "wg_gemm_kernel<128,128 tail (+16 rows)>",
                14: "wg_gemm_kernel<256,256,64,2,2,4,ping-pong>",
                16: "wg_gemm_pp_persist_kernel<0, false, false> (256x256 tiles, ping-pong, persistent)",
                17: "wg_gemm_pp_persist_kernel<2, false, false> (256x256 tiles, ping-pong, persistent, LayerNorm folded in, row statistics from the producing GEMM's partial sums)",
                18: "wg_gemm_pp_persist_kernel<0, true, false> (256x256 tiles, ping-pong, persistent, leaves the row sums of its output)",
                20: "wg_gemm_kernel<256, 256, 64, 2, 2, 4, true, 2, true> (256x256 tiles, fp8 MFMA, per-row / per-channel scales)",
                21: "wg_gemm_pp_persist_kernel<0, false, true> (256x256 tiles, ping-pong, persistent, fp8 with MX block scales on both operands)",
                22: "wg_gemm_pp_persist_kernel<2, false, true> (persistent fp8 MX, LayerNorm folded in from the producing GEMM's partial sums)",
                23: "wg_gemm_pp_persist_kernel<0, true, true> (persistent fp8 MX, leaves its output's row sums and e4m3 + block-scale copy)"}
# (prefixes: round 6 added a fourth template argument -- the tile-seam flow -- behind these three)
PMC_PREFIX = {16: "wg_gemm_pp_persist_kernel<0, false, false", 17: "wg_gemm_pp_persist_kernel<2, false, false", 18: "wg_gemm_pp_persist_kernel<0, true, false",
              20: "wg_gemm_kernel<256, 256, 64, 2, 2, 4, true, 2, true>", 21: "wg_gemm_pp_persist_kernel<0, false, true",
              22: "wg_gemm_pp_persist_kernel<2, false, true", 23: "wg_gemm_pp_persist_kernel<0, true, true"}
FP8_KERNELS = (20, 21, 22, 23)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.probe_launch:
        return probe_launch(args, rank, local_rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    from walkgpt_amd import ops
    model = build_model(args, dev)
    inp = make_inputs(args, dev, rank)
    B, T, O = args.batch, args.seg_tokens, args.original
    gathered = None
    if dist is not None:
        gathered = torch.empty(world * B * T, O, O, device=dev, dtype=torch.bfloat16)   # 2 bytes per logit on the wire

    decode_ev = []
    side = torch.cuda.Stream(priority=args.side_priority) if not args.single_stream else None
    dec = torch.cuda.Stream(priority=args.dec_priority) if not args.single_stream else None
    dec_fn = model.decode_from_hidden if args.no_decode_graph else model.decode_from_hidden_graphed
    clip_streams = [side] + [torch.cuda.Stream(priority=args.side_priority) for _ in range(args.clip_split - 1)] if side is not None else []

    def step(record_decode=False, serial=False):
        """One pass over one batch.  Three HIP streams in steady state: the CLIP tower (side), the SAM encoder (main) and
        the prompt-encoder / mask-decoder / postprocess chain (dec).  The decode chain is latency-bound at low occupancy; on its
        own stream it runs under the NEXT step's encoders instead of in front of them.
        All work of every step is inside the timed region (the closing fence synchronises the device)."""
        with torch.no_grad():
            cur = torch.cuda.current_stream()
            use_side = side is not None and not serial
            tails = bool(args.tail_tiles)
            if use_side and args.clip_split > 1:
                # experiment: independent sub-batches of the tower on streams of their own (finer interleaving of the persistent kernels)
                per = B // args.clip_split
                parts = []
                for k, sk in enumerate(clip_streams):
                    sk.wait_stream(cur)
                    with torch.cuda.stream(sk):
                        f, _pre = model.encode_images_clip(inp["images_clip"][k * per:(k + 1) * per], inp["clip_resize_list"][k * per:(k + 1) * per],
                                                           tail_tiles=tails)
                        parts.append(f)
                for sk in clip_streams[1:]:
                    side.wait_stream(sk)
                with torch.cuda.stream(side):
                    feats = torch.cat(parts, 0)
            elif use_side:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    feats, _pre = model.encode_images_clip(inp["images_clip"], inp["clip_resize_list"], tail_tiles=tails)
            else:
                feats, _pre = model.encode_images_clip(inp["images_clip"], inp["clip_resize_list"], tail_tiles=tails)
            emb = model.get_visual_emb_tokens(inp["images"], sub_batches=1 if serial or side is None else args.sam_split)
            if args.with_msqp:
                model.project_visual_tokens(emb)

            def decode_part():
                if record_decode:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                masks, scores = dec_fn(emb, inp["seg_hidden"], inp["resize_list"], inp["original_size_list"])
                if record_decode:
                    e1.record()
                    decode_ev.append((e0, e1))
                if dist is not None:  # the path's one exchange step: mask logits only
                    from walkgpt_amd.dist import all_gather_masks_uniform
                    all_gather_masks_uniform(model._cat_rows(masks), out=gathered, wire_dtype=torch.bfloat16)
                return masks, scores

            if dec is not None and not serial:
                dec.wait_stream(cur)          # the embedding is ready once the main stream reaches this point
                emb.record_stream(dec)
                with torch.cuda.stream(dec):
                    masks, scores = decode_part()
                cur.wait_stream(side)         # keep at most one CLIP pass in flight per step
            else:
                masks, scores = decode_part()
        return feats, masks, scores, emb

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.main_stream_created:
        torch.cuda.set_stream(torch.cuda.Stream())
    if args.attn_pipe_mode is not None:
        from walkgpt_amd import _lib
        _lib.lib().wg_attn_pipe_mode(args.attn_pipe_mode)
    note("model built; warm-up")
    for _ in range(args.warmup):
        step()
    fence()
    note("timed region: %d steps" % args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record_decode=True)
    t_enq = time.perf_counter() - t0   # host time to enqueue the launches of all steps (no synchronisation inside)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = world * B * args.steps / elapsed
    decode_batch_ms = sum(a.elapsed_time(b) for a, b in decode_ev) / max(1, len(decode_ev))

    if args.steps_only:
        if rank == 0:
            print(json.dumps({"metric": "images/sec", "value": round(images_per_s, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "dtype": args.dtype, "steps_only": True,
                              "e2e_algorithmic_gflop_per_step": round(gflop_per_step(args), 1), "e2e_peak": PEAK_TF["bf16"],
                              "e2e_frac": round(gflop_per_step(args) / ms_per_step / PEAK_TF["bf16"], 4),      # per GPU: GFLOP / ms == TFLOP/s
                              "config": {"workload": workload_label(args, world)}}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return 0

    # ---- mask-decode latency of ONE image (its T prompts) on an otherwise idle GPU: what BASELINE's "mask-decode ms" names --------
    note("%.2f images/s; single-image decode latency" % images_per_s)
    _f, _m, _s, emb = step(serial=True)
    torch.cuda.synchronize()
    one = (emb[:1].contiguous(), inp["seg_hidden"][:1], inp["resize_list"][:1], inp["original_size_list"][:1])
    lat = {}
    with torch.no_grad():
        # "graph": inputs resident where the chain reads them (the captured graph's own input buffers, filled once outside the timed
        # loop) -- what the contract asks of every timed region; "graph_staged": the same with the two staging copies a caller pays
        # who keeps its embedding / [SEG] states elsewhere
        s_emb, s_hid = model.decode_graph_inputs(*one)
        s_emb.copy_(one[0])
        for d, h in zip(s_hid, one[1]):
            d.copy_(h)
        resident = (s_emb, s_hid, one[2], one[3])
        for name, fn, arg in (("graph", model.decode_from_hidden_graphed, resident), ("graph_staged", model.decode_from_hidden_graphed, one),
                              ("eager", model.decode_from_hidden, one)):
            one_ = arg
            # 20 untimed calls, then the median of three timed runs of 50: ten calls right behind the step loop (round 3's form) sat 5-7 % above
            # what tools/bench_decode.py measures for the same chain on the same box -- the part's clocks are still settling from the full load
            for _ in range(20):
                fn(*one_)
            torch.cuda.synchronize()
            runs = []
            for _rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    fn(*one_)
                e1.record()
                torch.cuda.synchronize()
                runs.append(e0.elapsed_time(e1) / 50)
            lat[name] = sorted(runs)[1]

    # ---- instrumented step: per-launch HIP-event timing of every GEMM, grouped by the kernel that ran ---------------------------
    # serialised (one stream): the events then bracket each launch running alone on the chip, which is what a kernel roofline
    # describes; in the timed region above the streams overlap and per-launch times are not separable
    torch.cuda.synchronize()
    note("instrumented step")
    with ops.time_gemms() as records:
        step(serial=True)
    torch.cuda.synchronize()
    per_kernel = {}
    for kid, M, N, K, e0, e1 in records:
        d = per_kernel.setdefault(kid, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += 2.0 * M * N * K
        d[2] += e0.elapsed_time(e1) * 1e-3
        d[3] += (1.0 if kid in FP8_KERNELS else 2.0) * (M * K + N * K) + 2.0 * M * N   # operands (1 B fp8 / 2 B bf16) + bf16 output
    dom = max(per_kernel, key=lambda k: per_kernel[k][2])
    n_l, fl, sec, byt = per_kernel[dom]
    achieved_tf = fl / sec / 1e12
    peak = PEAK_TF["fp8" if dom in FP8_KERNELS else "bf16"]
    gf_step = gflop_per_step(args)
    roofline = {"bound": "mfma", "kernel": KERNEL_NAMES.get(dom, str(dom)),
                "achieved": round(achieved_tf, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved_tf / peak, 4), "traffic": None,
                "launches_per_step": n_l, "avg_launch_us": round(sec / n_l * 1e6, 1),
                "measured": "HIP events around every launch of this kernel in one serialised (single-stream) step run right after the timed region",
                "all_gemm_ms_serial": round(sum(v[2] for v in per_kernel.values()) * 1e3, 3),
                "e2e_algorithmic_gflop_per_step": round(gf_step, 1),
                "e2e_achieved": round(gf_step / ms_per_step, 1),  # GFLOP/ms == TFLOP/s
                "e2e_peak": PEAK_TF["bf16"], "e2e_frac": round(gf_step / ms_per_step / PEAK_TF["bf16"], 4),
                "algorithmic_bytes_per_launch": round(byt / n_l)}
    # HBM traffic of the dominant kernel.  PMC counters cannot be read from inside an un-profiled process: `traffic` is what the committed
    # rocprofv3 passes of the same command measured (tools/profile_r06.sh: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH
    # doubled per MI355X_MICROARCH.md), labelled with the commit the profile was taken at, and only when it profiled THIS configuration
    # and this kernel; null otherwise.
    for pmc_file in ("r06_pmc_traffic.json", "r06_c5_fp8_pmc_traffic.json", "r05_pmc_traffic.json", "r05_c5_fp8_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                pmc = json.load(f)
            cfg = dict(pmc.get("config") or {})
            cfg.setdefault("fp8_clip", args.dtype == "fp8" and pmc_file.startswith("r03"))   # round 3's C5 profile ran both towers in fp8
            same = cfg == {"batch": B, "sam": args.sam, "seg_tokens": T, "with_msqp": bool(args.with_msqp), "dtype": args.dtype,
                           "world": 1, "fp8_clip": bool(args.dtype == "fp8" and args.fp8_clip)}
            key = [k for k in pmc["kernels"] if dom in PMC_PREFIX and k.startswith(PMC_PREFIX[dom])]
            if same and key and "traffic_profiled" not in roofline:
                roofline["traffic_profiled"] = pmc["kernels"][key[0]]["hbm_bytes_per_launch"]
                # `traffic` carries the profiled figure (per launch, like `achieved`); the fields beside it say which commit and which passes it is from
                roofline["traffic"] = roofline["traffic_profiled"]
                roofline["traffic_profiled_head"] = pmc.get("head", "unknown (round-3 profile, taken before this field existed)")
                roofline["traffic_profiled_source"] = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE in separate passes of this "
                                                       "command, avg per launch; NOT measured in this run)" % pmc_file)
        except (OSError, KeyError, ValueError):
            pass

    out = {"metric": "images/sec", "value": round(images_per_s, 2), "unit": "images/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "mask_decode_ms": round(lat["graph"], 3),
           "mask_decode": {"one_image_latency_ms": round(lat["graph"], 3), "one_image_latency_staged_ms": round(lat["graph_staged"], 3),
                           "one_image_latency_eager_ms": round(lat["eager"], 3),
                           "amortised_ms_per_image": round(decode_batch_ms / B, 3), "batch_ms_overlapped": round(decode_batch_ms, 3),
                           "prompts_per_image": T,
                           "note": "latency (median of 3 runs of 50 calls after 20 untimed ones): prompt encoder + mask decoder + postprocess of one image's T prompts alone on the GPU (CTP included), graph "
                                   "replay with the embedding and [SEG] states resident in the graph's input buffers (staged: + the two copies into them); "
                                   "amortised: the batch's decode chain as timed inside the step, overlapped with the next step's encoders, / images"},
           "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
           "rccl_ranks": world if dist is not None else 0,
           "config": {"workload": workload_label(args, world), "global_batch": world * B, "batch_per_gpu": B, "seg_tokens_per_image": T,
                      "parallelism": "dp%d (images sharded, RCCL all-gather of mask logits)" % world if world > 1 else "single GPU"},
           "roofline": roofline}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("cpu baseline (oracle on %d host threads)" % (args.cpu_threads or usable_cpus()))
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0 and world == 1 and not args.no_secondary and config_name(args, world) == "C2":
        # the headline is measured and final; its model and buffers go before the secondary processes build theirs
        del model, inp, emb, one, resident, s_emb, s_hid
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        note("secondary passes (fresh processes, --steps-only)")
        out["secondary"] = run_secondary()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
