"""Per-shape L2 hit rate of the persistent GEMMs from a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum pass of the single-stream step
(tools/profile_r05.sh: gpurun_out/p5_l2).  The kernel name does not carry the shape; the place in a transformer block does: in the serialised step
the launches of a SAM block are  <2>(qkv)  attention  <0,true>(proj)  <2>(lin1)  <0,true>(lin2)  and of a CLIP layer the same with the
plain attention kernel in between, so a GEMM launch is labelled by its kernel, the attention kernel last seen and its position behind it.
    python tools/l2_hit_by_shape.py gpurun_out/p5_l2 [cycles-per-slab table is in profiles/r04_gemm_phases.md]"""
import csv, sys, collections
d = sys.argv[1]
rows = collections.OrderedDict()
for r in csv.DictReader(open(d + "/s_counter_collection.csv")):
    k = int(r["Dispatch_Id"])
    e = rows.setdefault(k, {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
seq = [rows[k] for k in sorted(rows)]
# one steady-state step: between the last two SAM patch gathers
starts = [i for i, e in enumerate(seq) if e["name"].startswith("wg_patchify_kernel<true>") or "wg_patchify_kernel<true>" in e["name"]]
seq = seq[starts[-2]:starts[-1]] if len(starts) > 1 else seq      # one whole step (the CLIP tower runs behind the SAM encoder's first launch)
tower, pos, out = None, 0, collections.defaultdict(list)
for e in seq:
    n = e["name"]
    if "wg_attn_window" in n or "wg_attn_pipe" in n:
        tower, pos = "SAM", 0
    elif n.startswith("wg_attn_kernel") or "wg_attn_kernel<64, 0, 4" in n:
        tower, pos = "CLIP", 0
    elif "wg_gemm_pp_persist_kernel<0, true" in n and tower:
        pos += 1
        out[(tower, "proj / out_proj" if pos == 1 else "lin2 / fc2")].append(e)
    elif "wg_gemm_pp_persist_kernel<2, false" in n and tower:
        # behind an attention launch: proj(1) lin1(2) lin2(3) then the NEXT block's qkv(4)
        pos += 1
        out[(tower, "lin1 / fc1" if pos == 2 else "q,k,v")].append(e)
print("| tower | GEMM | launches | avg us | TCC_HIT | TCC_MISS | L2 hit rate |")
print("|---|---|---|---|---|---|---|")
for (tw, g), es in sorted(out.items()):
    hit = sum(e.get("TCC_HIT_sum", 0) for e in es) / len(es)
    miss = sum(e.get("TCC_MISS_sum", 0) for e in es) / len(es)
    us = sum(e["t1"] - e["t0"] for e in es) / len(es) / 1e3
    print("| %s | %s | %d | %.1f | %.3g | %.3g | %.1f %% |" % (tw, g, len(es), us, hit, miss, 100 * hit / max(1.0, hit + miss)))
