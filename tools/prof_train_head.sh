#!/bin/bash
# per-kernel table of one training-head step (tools/bench_train_head.py under rocprofv3) -> gpurun_out/th/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/bench_train_head.py --steps 10 --warmup 3 "$@" 2>&1 | tail -1
rm -rf gpurun_out/th
rocprofv3 --kernel-trace --stats -d gpurun_out/th -o th --output-format csv -- python3 tools/bench_train_head.py --steps 4 --warmup 1 "$@" > gpurun_out/th.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/th/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r["Calls"]) for r in rows); t=sum(float(r["TotalDurationNs"]) for r in rows)
print("launches/step %.0f  kernel ms/step %.2f" % (tot/5, t/5e6))
for r in rows[:22]: print("%-70s %6d %8.1f us %5.1f%%" % (r["Name"][:70], int(r["Calls"])//5, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
