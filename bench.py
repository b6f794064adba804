#!/usr/bin/env python3
"""bench.py -- images/sec of WalkGPT's grounded-segmentation forward path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input on each GPU (config C2 of SURVEY.md §8d):
  bs=8 source images of 448x448  ->  images_clip [8,3,448,448] and SAM input [8,3,1024,1024] (bf16, resident in HBM)
  CLIP ViT-L/14 tower (24 layers, 1025 tokens, key-padding mask)  +  SAM ViT-B image encoder
  + CTP on T [SEG] hidden states per image + prompt encoder + two-way mask decoder + fused postprocess to 448x448
  (+ one RCCL all-gather of the mask logits when world_size > 1).
The language model between MSQP and CTP is not part of config C2 and is not run.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the dominant kernel (the 256x256-tile bf16 MFMA GEMM): algorithmic FLOPs of all its launches in one step
                / the sum of their durations, timed with HIP events on the launch stream in an instrumented, serialised
                (single-stream) step that runs right after the timed region (same process, same buffers);
                rocprofv3's average for the same kernel: profiles/r01_single_stream_summary.md.  e2e_* = whole step.
  cpu_baseline  the CPU oracle (oracle/, fp32 PyTorch restatement pinned to the reference) timed on the host cores on
                a bounded sample (one image of the batch), rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# algorithmic FLOPs per image (2*MAC, matmul/conv only) -- BASELINE.md §2, counted on the reference modules
GF_CLIP_L_448 = 723.6
GF_SAM_B = 972.1
GF_DECODE_PER_TOKEN = 3.61
GF_CTP_PER_TOKEN = 0.00446
GF_MSQP = 50.8
GF_SAM = {"vit_b": 972.1, "vit_l": 2985.7, "vit_h": 5961.1}
MFMA_BF16_DENSE_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--seg-tokens", type=int, default=1, help="[SEG] tokens per image (reference default --seg_token_num=1)")
    ap.add_argument("--sam", default="vit_b")
    ap.add_argument("--llm-hidden", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-msqp", action="store_true", help="also run the Multi-Scale Query Projector on the SAM tokens (config C3's projector)")
    ap.add_argument("--tail-tiles", action="store_true", help="allow the tail-absorbing 128x128 GEMM tiles (wins with --single-stream)")
    ap.add_argument("--sam-split", type=int, default=1, help="experiment: run the SAM encoder in this many batch chunks on separate streams")
    ap.add_argument("--side-priority", type=int, default=0, help="HIP priority of the CLIP stream (-1 = high)")
    ap.add_argument("--single-stream", action="store_true", help="run the CLIP tower and the SAM branch back to back")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-decode-graph", action="store_true", help="launch the decode chain eagerly instead of replaying its captured HIP graph")
    return ap.parse_args()


def build_model(args, dev):
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
    return model


def make_inputs(args, dev, rank):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    B, T = args.batch, args.seg_tokens
    images = torch.randn(B, 3, 1024, 1024, generator=g).to(dev, torch.bfloat16)
    images_clip = torch.randn(B, 3, 448, 448, generator=g).to(dev, torch.bfloat16)
    seg_hidden = [torch.randn(T, args.llm_hidden, generator=g).to(dev, torch.bfloat16) for _ in range(B)]
    return dict(images=images, images_clip=images_clip, seg_hidden=seg_hidden, resize_list=[(1024, 1024)] * B,
                original_size_list=[(448, 448)] * B, clip_resize_list=[(448, 448)] * B)


def cpu_baseline(args):
    """The oracle's own forward for ONE image of the workload, fp32, on the host cores."""
    from oracle import clip as oclip
    from oracle import projectors as oproj
    from oracle import sam as osam
    from tests.golden import cases
    threads = args.cpu_threads or min(16, len(os.sched_getaffinity(0)))  # 16 = the CPU share of a one-GPU box
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
    w = rnd({k: tuple(v.shape) for k, v in _shape_only_encoder(c).items()})
    w.update(rnd(cases.decoder_weight_shapes()))
    clip_c = dict(dim=1024, heads=16, layers=24, img=448)
    wc = rnd(cases.clip_weight_shapes(clip_c))
    wt = rnd(cases.ctp_weight_shapes(args.llm_hidden))
    wt["text_type"] = wt["text_type"].reshape(1, 1, -1)
    x = torch.randn(1, 3, 1024, 1024, generator=gen)
    xc = torch.randn(1, 3, 448, 448, generator=gen)
    hid = torch.randn(args.seg_tokens, args.llm_hidden, generator=gen)
    cfg = dict(patch=16, depth=c["depth"], heads=c["heads"], global_idx=c["global_idx"], window=14)
    with torch.no_grad():
        t0 = time.perf_counter()
        key_mask = oclip.patch_key_mask(1, (448, 448), [(448, 448)])
        oclip.clip_tower(wc, xc, key_mask, -2)
        t1 = time.perf_counter()
        emb = osam.image_encoder(w, x, cfg)
        t2 = time.perf_counter()
        pe = oproj.ctp(wt, hid).reshape(-1, 1, 256)
        dpe = osam.dense_pe(w, (64, 64))
        sparse, dense = osam.prompt_encoder_text(w, pe, (64, 64))
        masks, _ = osam.mask_decoder(w, emb, dpe, sparse, dense)
        post = osam.postprocess_masks(masks, 1024, (1024, 1024), (448, 448))
        osam.mask_score(post[:, 0])
        t3 = time.perf_counter()
    total = t3 - t0
    return {"value": round(1.0 / total, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "1 image of the batch (CLIP ViT-L@448 %.1fs + SAM ViT-B@1024 %.1fs + decode T=%d %.2fs), fp32 oracle, torch CPU"
                      % (t1 - t0, t2 - t1, args.seg_tokens, t3 - t2),
            "mask_decode_ms": round((t3 - t2) * 1e3, 1)}


def _shape_only_encoder(c):
    """Key -> zero tensor of the right shape for a SAM encoder config (values are drawn by the caller)."""
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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
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
    ops.ALLOW_TAIL_TILES = bool(args.tail_tiles)
    model = build_model(args, dev)
    inp = make_inputs(args, dev, rank)
    B, T = args.batch, args.seg_tokens
    gathered = None
    if dist is not None:
        gathered = torch.empty(world * B * T, 448, 448, device=dev, dtype=torch.float32)

    decode_ev = []

    side = torch.cuda.Stream(priority=args.side_priority) if not args.single_stream else None
    dec = torch.cuda.Stream() if not args.single_stream else None
    sam_streams = [torch.cuda.Stream() for _ in range(args.sam_split)] if args.sam_split > 1 else []

    def step(record_decode=False, serial=False):
        """One pass over one batch.  Three HIP streams in steady state: the CLIP tower (side), the SAM encoder (main) and
        the prompt-encoder / mask-decoder / postprocess chain (dec).  The decode chain is ~75 small latency-bound
        launches at low occupancy; on its own stream it runs under the NEXT step's encoders instead of in front of them.
        All work of every step is inside the timed region (the closing fence synchronises the device)."""
        with torch.no_grad():
            cur = torch.cuda.current_stream()
            use_side = side is not None and not serial
            if use_side:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    feats, _pre = model.encode_images_clip(inp["images_clip"], inp["clip_resize_list"])
            else:
                feats, _pre = model.encode_images_clip(inp["images_clip"], inp["clip_resize_list"])
            if args.sam_split > 1 and use_side:
                # experiment: the SAM batch in `sam_split` chunks on separate streams (MFMA-bound GEMMs of one chunk
                # next to the VALU-bound attention of another)
                chunks = list(torch.chunk(inp["images"], args.sam_split, 0))
                parts = [None] * len(chunks)
                for i, ch in enumerate(chunks):
                    st = sam_streams[i]
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        parts[i] = model.get_visual_emb_tokens(ch.contiguous())
                for i, st in enumerate(sam_streams[:len(chunks)]):
                    cur.wait_stream(st)
                    parts[i].record_stream(cur)
                emb = torch.cat(parts, 0)
            else:
                emb = model.get_visual_emb_tokens(inp["images"])
            if args.with_msqp:
                model.project_visual_tokens(emb)

            def decode_part():
                if record_decode:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                # the ~75 small launches of the chain are replayed from one captured HIP graph (same kernels, same arguments)
                dec_fn = model.decode_from_hidden if args.no_decode_graph else model.decode_from_hidden_graphed
                masks, scores = dec_fn(emb, inp["seg_hidden"], inp["resize_list"], inp["original_size_list"])
                if record_decode:
                    e1.record()
                    decode_ev.append((e0, e1))
                if dist is not None:  # the path's one exchange step: mask logits only
                    from walkgpt_amd.dist import all_gather_masks_uniform
                    all_gather_masks_uniform(torch.cat(masks, 0), out=gathered)
                return masks, scores

            if dec is not None and not serial:
                dec.wait_stream(cur)          # the embedding is ready once the main stream reaches this point
                emb.record_stream(dec)
                with torch.cuda.stream(dec):
                    masks, scores = decode_part()
                cur.wait_stream(side)         # keep at most one CLIP pass in flight per step
            else:
                masks, scores = decode_part()
        return feats, masks, scores

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
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
    decode_ms = sum(a.elapsed_time(b) for a, b in decode_ev) / max(1, len(decode_ev)) / B

    # ---- instrumented step: per-launch HIP-event timing of every GEMM, grouped by the tile kernel that ran ---------
    records = []

    def hook(M, N, K, tile):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        records.append((tile, 2.0 * M * N * K, e0, e1, 2.0 * (M * K + N * K + M * N)))
        return e0, e1

    # serialised (one stream): the events then bracket each launch running alone on the chip, which is what a kernel
    # roofline describes; in the timed region above the streams overlap and per-launch times are not separable
    torch.cuda.synchronize()
    ops.GEMM_EVENT_HOOK = hook
    step(serial=True)
    torch.cuda.synchronize()
    ops.GEMM_EVENT_HOOK = None
    per_tile = {}
    for tile, fl, e0, e1, nbytes in records:
        d = per_tile.setdefault(tile, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += fl
        d[2] += e0.elapsed_time(e1) * 1e-3
        d[3] += nbytes
    dom = max(per_tile, key=lambda k: per_tile[k][2])
    n_l, fl, sec, byt = per_tile[dom]
    achieved_tf = fl / sec / 1e12
    gf_step = B * (GF_CLIP_L_448 + GF_SAM[args.sam] + (GF_MSQP if args.with_msqp else 0.0) + T * (GF_DECODE_PER_TOKEN + GF_CTP_PER_TOKEN))
    roofline = {"bound": "mfma", "kernel": "wg_gemm_pp_persist_kernel<%s> (256x256 tiles, ping-pong, persistent%s)" % (("true", ", LayerNorm folded in") if dom == 17 else ("false", "")) if dom in (16, 17) else "wg_gemm_kernel<%s>" % {1: "128,128,64,2,2,2", 2: "256,256,64,2,2,4", 8: "256,256,64,2,2,4,pipe", 11: "persist 128,128,2,2", 12: "tail 128,128 (+16 rows)", 14: "256,256,64,2,2,4,ping-pong", 3: "rowwave"}.get(dom, str(dom)),
                "achieved": round(achieved_tf, 1), "peak": MFMA_BF16_DENSE_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(achieved_tf / MFMA_BF16_DENSE_PEAK_TF, 4), "traffic": None,
                "launches_per_step": n_l, "avg_launch_us": round(sec / n_l * 1e6, 1),
                "measured": "HIP events around every launch of this kernel in one serialised (single-stream) step run right after the timed region",
                "all_gemm_ms_serial": round(sum(v[2] for v in per_tile.values()) * 1e3, 3),
                "e2e_algorithmic_gflop_per_step": round(gf_step, 1),
                "e2e_achieved": round(gf_step / ms_per_step, 1),  # GFLOP/ms == TFLOP/s
                "e2e_frac": round(gf_step / ms_per_step / MFMA_BF16_DENSE_PEAK_TF, 4)}

    try:  # HBM traffic of the dominant kernel from the committed PMC passes of this same command (profiles/)
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pmc = json.load(f)["kernels"]
        key = [k for k in pmc if k.startswith("wg_gemm_pp_persist_kernel<true>" if dom == 17 else "wg_gemm_pp_persist_kernel<false>" if dom == 16 else "wg_gemm_kernel<%s" % ("256, 256" if dom in (2, 14) else "128, 128"))]
        if key:
            roofline["traffic"] = pmc[key[0]]["hbm_bytes_per_launch"]
            roofline["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, avg per launch)"
    except (OSError, KeyError, ValueError):
        pass
    roofline["algorithmic_bytes_per_launch"] = round(byt / n_l)

    out = {"metric": "images/sec", "value": round(images_per_s, 2), "unit": "images/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "mask_decode_ms": round(decode_ms, 3), "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
           "config": {"workload": "C2: bs=%d/GPU 448x448 source images (CLIP input 448^2, SAM input 1024^2), CLIP ViT-L/14 + SAM %s "
                                  "encoder%s + CTP + prompt encoder + mask decoder + postprocess, T=%d [SEG]/image, random-init weights"
                                  % (B, args.sam, " + MSQP" if args.with_msqp else "", T),
                      "global_batch": world * B, "batch_per_gpu": B, "seg_tokens_per_image": T,
                      "parallelism": "dp%d (images sharded, RCCL all-gather of mask logits)" % world if world > 1 else "single GPU"},
           "roofline": roofline}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
