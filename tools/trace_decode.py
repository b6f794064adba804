"""Timeline of the last decode-graph replay out of a rocprofv3 kernel trace (rocpd database):
   rocprofv3 --kernel-trace -d gpurun_out/dec1 -o dec1 -- python3 tools/bench_decode.py --batch 1;  python tools/trace_decode.py gpurun_out/dec1/dec1_results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 22
last = rows[-n:]
t0, prev = last[0][1], None
for name, s, e in last:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-48s start %7.1f  dur %5.1f  gap %5.1f" % (short[:48], (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
print("span %.1f us" % ((last[-1][2] - t0) / 1e3))
