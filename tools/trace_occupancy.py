"""How full is the chip during a steady-state step?  From a rocprofv3 kernel trace (csv) of bench.py: for every instant of the last whole step, the number of
compute units the RUNNING kernels can occupy at most (workgroups of the dispatch / workgroups that fit a CU by LDS, registers and threads, capped at 256),
summed over the concurrent kernels.  A lower bound of the idle CU-time: persistent GEMM workgroups that run out of tiles early, and the tails of the other
kernels, are not seen (a dispatch counts as full from its start to its end).

    python tools/trace_occupancy.py gpurun_out/p5_stats/s_kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nwg = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(wg, 1)
    waves = (wg + 63) // 64
    lds = int(r["LDS_Block_Size"])
    regs = 2 * (int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))      # (this rocprofv3 reports half the per-lane register count of a wave64 kernel)
    alloc = max(8, (regs + 7) // 8 * 8)
    per_simd = min(8, 512 // alloc)
    by_regs = max(1, (per_simd * 4) // waves)
    by_lds = (160 * 1024) // lds if lds > 0 else 99      # (static LDS only: dynamic allocations are not in the trace)
    name = r["Kernel_Name"]
    if "wg_gemm_pp_persist_kernel" in name or "wg_attn_window_unit_kernel" in name or "wg_gemm_fr_kernel" in name:
        by_lds = 1                                         # 130-160 KiB of dynamic LDS
    elif "wg_gemm_persist_kernel<128, 128" in name or "wg_gemm_kernel<128, 128" in name:
        by_lds = 2
    by_thr = 32 // waves
    per_cu = max(1, min(by_regs, by_lds, by_thr))
    r["cus"] = min(256, (nwg + per_cu - 1) // per_cu)
    r["nwg"], r["per_cu"] = nwg, per_cu
# the last whole step: between the last two launches of the SAM patch gather
marks = [r["s"] for r in rows if r["Kernel_Name"].startswith("void wg_patchify_kernel<true>") or "wg_patchify_kernel<true>" in r["Kernel_Name"]]
assert len(marks) >= 2, "no step marker in the trace"
t0, t1 = marks[-2], marks[-1]
ev = []
for r in rows:
    if r["e"] <= t0 or r["s"] >= t1:
        continue
    ev.append((max(r["s"], t0), r["cus"], r))
    ev.append((min(r["e"], t1), -r["cus"], r))
ev.sort(key=lambda x: x[0])
cur, last = 0, t0
hist = collections.Counter()
busy_cu_time = 0
for t, d, r in ev:
    dt = t - last
    if dt > 0:
        c = min(cur, 256)
        busy_cu_time += c * dt
        hist[min(c // 32, 8)] += dt
    cur += d
    last = t
step = t1 - t0
print("step period %.3f ms; upper bound of the occupied CU-time %.1f %% of 256 CUs x step" % (step / 1e6, 100.0 * busy_cu_time / (256.0 * step)))
print("share of the step by the number of CUs the running dispatches can occupy (sum over concurrent kernels, capped at 256):")
for b in range(9):
    lab = "256" if b == 8 else "%3d-%3d" % (b * 32, b * 32 + 31)
    print("  %s CUs: %5.1f %%" % (lab, 100.0 * hist[b] / step))
# which kernels run alone below 256?
alone = collections.Counter()
cur_set = []
last = t0
for t, d, r in ev:
    dt = t - last
    if dt > 0 and sum(x["cus"] for x in cur_set) < 224:
        key = " + ".join(sorted(x["Kernel_Name"].split("(")[0].replace("void ", "")[:48] + "[%d]" % x["cus"] for x in cur_set)) or "(nothing)"
        alone[key] += dt
    if d > 0:
        cur_set.append(r)
    else:
        cur_set.remove(r)
    last = t
print("time with fewer than 224 CUs claimed, by what was running (top 15):")
for k, v in alone.most_common(15):
    print("  %6.1f us  %s" % (v / 1e3, k))
