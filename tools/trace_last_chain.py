"""Kernel sequence of the last decode chain in a rocprofv3 --kernel-trace CSV (tools/bench_decode.py under the profiler)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wg_postprocess" in r["Kernel_Name"]]
t0 = prev = None
tot = 0.0
for r in rows[idx[-2] + 2:idx[-1] + 2]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    t0 = t0 or st
    print("%8.1f  gap %5.1f  dur %6.1f  %-62s wg %s" % ((st - t0) / 1e3, (st - prev) / 1e3 if prev else 0, (en - st) / 1e3, r["Kernel_Name"][:62],
                                                      int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    prev = en
    tot += (en - st) / 1e3
print("sum of durations %.1f us" % tot)
