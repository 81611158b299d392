#!/usr/bin/env python3
"""Print the launches of the LAST per-batch index build found in a rocprofv3 kernel trace (csv), with the gaps between them:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ixt -o s -- python3 bench.py --steady --steps 2 --warmup 1
   python tools/index_trace.py gpurun_out/ixt"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "ril_count_kernel" in n]
lo = starts[-1]
hi = next(i for i in range(lo, len(rows)) if "rows_transform" in names[i] or "rows_chain2" in names[i])
t0 = int(rows[lo]["Start_Timestamp"])
prev = None
busy = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%8.1f  gap %6.1f  run %6.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev = e
print("%d launches, span %.1f us, busy %.1f us" % (hi - lo, (prev - t0) / 1e3, busy / 1e3))
