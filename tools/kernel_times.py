#!/usr/bin/env python3
"""Per-kernel durations of one eager bench run (rocprofv3 --kernel-trace): python tools/kernel_times.py [bench args]"""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = "/tmp/dn_ktimes"
os.environ["TMPDIR"] = "/tmp"
args = sys.argv[1:] or ["--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-graph"]
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "k", "--", "python3",
                os.path.join(ROOT, "bench.py")] + args, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in ("wgrad", "rows_transform", "rows_selfsum", "rows_chain2", "gather_segsum_vec")):
        import re
        m = re.search(r"(rows_wgrad_dma_kernel|rows_wgrad_kernel|wgrad_reduce_kernel|rows_transform_kernel|rows_selfsum_kernel|rows_chain2_kernel|gather_segsum_vec_kernel)(<[^>]*>)?", n)
        key = (m.group(0) if m else n[:60]) + (" f32" if ("float" in n and "segsum" in n) else "")
        by[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    v2 = sorted(v)
    clusters = sorted(set(round(x, -1) for x in v2))
    print("%-72s n=%4d median %8.1f us   distinct(10us): %s" % (k, len(v), v2[len(v2) // 2], clusters[:14]))
