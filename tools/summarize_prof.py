"""Turn rocprofv3 output directories (gpurun_out/...) into the small tracked summaries under profiles/.

    python tools/summarize_prof.py TAG --stats DIR [--fetch DIR --write DIR] [--sq DIR] --steps N --cmd "..." [--config JSON]

  --stats  rocprofv3 --kernel-trace --stats of `bench.py --steps K --warmup W --steps-only`; per-step launches and times come from the
           kernel trace (the window between the last two SAM patch-gather launches), one-time work is what ran before the first step
  --fetch / --write   separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (MI355X_MICROARCH.md: FETCH_SIZE x2 on gfx950) -> bytes / launch
  --sq     a --pmc pass with GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY ...:
           matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)   (SIMD-cycles; checked against the MFMA
           count of the attention kernel: 4.15e8 counted vs 4.02e8 from its FLOPs), VALU busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) over
           the same denominator, parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES.
Writes profiles/TAG_kernel_stats.csv, profiles/TAG_summary.md, profiles/TAG_pmc_traffic.json, profiles/TAG_mfma_busy.md."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(d, suffix):
    g = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not g:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return g[0]


def _git_head():
    import subprocess
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "unknown"
    except OSError:
        return "unknown"


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:80]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--stats", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--sq-cmd", default=None, help="the command of the --sq pass when it differs from --cmd")
    ap.add_argument("--steps", type=int, required=True, help="full steps the profiled command ran (timed + warm-up)")
    ap.add_argument("--cmd", required=True)
    ap.add_argument("--head", default=None, help="git commit the profiled tree was at (default: this checkout's HEAD)")
    ap.add_argument("--config", default=None, help='JSON the bench compares with its own run before it reports roofline.traffic')
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    rows = list(csv.DictReader(open(one(a.stats, "kernel_stats.csv"))))
    with open(os.path.join(out, a.tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows[:45])
    traffic = {}
    if a.fetch and a.write:
        pmc = {}
        for kind, d in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
            acc = defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(one(d, "counter_collection.csv"))):
                if r["Counter_Name"] == kind:
                    e = acc[short(r["Kernel_Name"])]
                    e[0] += 1
                    e[1] += float(r["Counter_Value"])
            for k, (n, v) in acc.items():
                pmc.setdefault(k, {})[kind] = {"launches": n, "avg_kb": v / n}
        for k, v in pmc.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                fetch = 2.0 * v["FETCH_SIZE"]["avg_kb"] * 1024.0  # gfx950: FETCH_SIZE reports half of a wide coalesced read
                write = v["WRITE_SIZE"]["avg_kb"] * 1024.0
                traffic[k] = {"launches": v["FETCH_SIZE"]["launches"], "read_bytes_per_launch": round(fetch),
                              "write_bytes_per_launch": round(write), "hbm_bytes_per_launch": round(fetch + write)}
        with open(os.path.join(out, a.tag + "_pmc_traffic.json"), "w") as f:
            json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = KB * 1024, FETCH_SIZE x2 "
                               "(MI355X_MICROARCH.md: gfx950 counts 128-B requests at 64 B)", "command": a.cmd,
                       "head": a.head or _git_head(), "config": json.loads(a.config) if a.config else None, "kernels": traffic}, f, indent=1,
                      sort_keys=True)
    # per-step work from the kernel trace itself: everything that starts between the last two SAM patch-gather launches (the first
    # kernel of a step on the main stream) is one steady-state step across all streams; what ran before the first step is one-time work
    trace = list(csv.DictReader(open(one(a.stats, "kernel_trace.csv"))))
    trace.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(trace) if "wg_patchify_kernel<true>" in r["Kernel_Name"]]
    if len(marks) < 3:
        raise SystemExit("need at least three steps in the trace")
    t0, t1 = int(trace[marks[-2]]["Start_Timestamp"]), int(trace[marks[-1]]["Start_Timestamp"])
    step = defaultdict(lambda: [0, 0.0])
    for r in trace:
        if t0 <= int(r["Start_Timestamp"]) < t1:
            e = step[short(r["Kernel_Name"])]
            e[0] += 1
            e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    first = int(trace[marks[0]]["Start_Timestamp"])
    once = defaultdict(lambda: [0, 0.0])
    for r in trace:
        if int(r["Start_Timestamp"]) < first:
            e = once[short(r["Kernel_Name"])]
            e[0] += 1
            e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    t_step = sum(v[1] for v in step.values())
    n_torch = sum(v[0] for k, v in step.items() if k.startswith("at::") or k.startswith("__amd_rocclr"))
    with open(os.path.join(out, a.tag + "_summary.md"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats of `%s` (%s)\n\n" % (a.cmd, a.tag))
        f.write("Per-step figures are taken from the kernel trace, not from a division: every kernel that STARTS between the last two "
                "launches of the SAM patch gather (the first kernel of a step on the main stream) belongs to one steady-state step, on "
                "whatever stream it ran (step period %.3f ms).  Kernel time summed over the streams: **%.2f ms / step** in %d launches, "
                "%d of them torch copies / concatenations (the decode graph's input staging).  One-time work before the first step "
                "(weight re-layouts, LayerNorm folds, fp8 weight quantisation, graph capture): %.1f ms, listed below the table.\n\n"
                % ((t1 - t0) / 1e6, t_step / 1e6, sum(v[0] for v in step.values()), n_torch, sum(v[1] for v in once.values()) / 1e6))
        f.write("| kernel | launches / step | avg us | ms / step | % of step kernel time | HBM bytes / launch (PMC) |\n|---|---|---|---|---|---|\n")
        for k, (n, ns) in sorted(step.items(), key=lambda kv: -kv[1][1])[:26]:
            t = traffic.get(k, {}).get("hbm_bytes_per_launch")
            f.write("| `%s` | %d | %.1f | %.3f | %.1f | %s |\n" % (k, n, ns / n / 1e3, ns / 1e6, 100.0 * ns / t_step, ("%.1f MB" % (t / 1e6)) if t else "-"))
        f.write("\nOne-time kernels (before the first step): ")
        f.write(", ".join("`%s` x%d (%.2f ms)" % (k[:50], n, ns / 1e6) for k, (n, ns) in sorted(once.items(), key=lambda kv: -kv[1][1])[:10]) + "\n")
    if a.sq:
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        rows = list(csv.DictReader(open(one(a.sq, "counter_collection.csv"))))
        # one steady-state step: the dispatches between the last two SAM patch gathers (model construction and the first step's one-time
        # preparation -- weight casts, folds, re-layouts -- are not part of the path and would otherwise rank among its kernels)
        marks = sorted({int(r["Dispatch_Id"]) for r in rows if "wg_patchify_kernel<true>" in r["Kernel_Name"]})
        lo, hi = (marks[-2], marks[-1]) if len(marks) >= 2 else (0, 1 << 62)
        for r in rows:
            if not lo <= int(r["Dispatch_Id"]) < hi:
                continue
            e = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
        ranked = sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [0, 0.0])[1])
        with open(os.path.join(out, a.tag + "_mfma_busy.md"), "w") as f:
            f.write("# Matrix-pipe and VALU utilisation per kernel (%s)\n\n`%s` under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES "
                    "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY` (kernels serialised by "
                    "the counter collection).\n\nmatrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); VALU busy = "
                    "4 x SQ_ACTIVE_INST_VALU over the same SIMD-cycles; parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves at s_waitcnt / s_barrier); "
                    "issue-stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.  Only ONE steady-state step is counted: the dispatches between the last two SAM patch gathers (model construction and first-step preparation excluded); kernels ranked by their share of GPU-active cycles.\n\n" % (a.tag, a.sq_cmd or a.cmd))
            f.write("| kernel | launches | share of active cycles | matrix pipe busy | VALU busy | waves parked | issue-stalled |\n|---|---|---|---|---|---|---|\n")
            tot = sum(v.get("GRBM_GUI_ACTIVE", [0, 0.0])[1] for _, v in ranked) or 1.0
            for k, v in ranked[:12]:
                gui = v.get("GRBM_GUI_ACTIVE", [0, 0.0])[1]
                simd = gui / 8.0 * 1024.0
                wave = v.get("SQ_WAVE_CYCLES", [0, 1.0])[1] or 1.0
                f.write("| `%s` | %d | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.1f %% |\n" % (
                    k, v["GRBM_GUI_ACTIVE"][0], 100.0 * gui / tot, 100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])[1] / simd,
                    100.0 * 4.0 * v.get("SQ_ACTIVE_INST_VALU", [0, 0.0])[1] / simd, 100.0 * v.get("SQ_WAIT_ANY", [0, 0.0])[1] / wave,
                    100.0 * v.get("SQ_WAIT_INST_ANY", [0, 0.0])[1] / wave))
    print(open(os.path.join(out, a.tag + "_summary.md")).read())


if __name__ == "__main__":
    main()
