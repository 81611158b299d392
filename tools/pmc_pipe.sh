#!/bin/bash
# HBM traffic + L2 hit rate of the persistent launch on the benchmark batch (separate --pmc passes, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/pmc_$(echo $ctr | tr ' ' '_')
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -o p -- python3 $R/tools/pipe_stats.py $1 > /dev/null 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (r["Kernel_Name"][:60], r["Counter_Name"])
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    if "pipe" in k or "selfsum" in k or "rows_transform" in k or "gather_segsum" in k:
        print("%-62s %-14s launches %3d  per launch %.1f" % (k, c, n, v / n))
PY
done
