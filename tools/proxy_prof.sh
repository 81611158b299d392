# GPU box: per-kernel times of the 4096-graph shard step (steady = graph replay only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4096 -o k -- python3 $GRAFT_REPO_ROOT/bench.py --graphs 4096 --steady --no-cpu-baseline --no-proxy --steps 200 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/proxy_line.json 2> /dev/null
f=$(find /tmp/prof4096 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel time / 205 steps = %.1f us" % (tot / 205 / 1e3))
for r in rows[:40]:
    print("%-64s calls/step %5.2f avg %7.1f us  per step %6.1f us  %5.1f%%" % (r["Name"][:64], int(r["Calls"]) / 205.0, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 205 / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
python3 - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/proxy_line.json").read().strip().splitlines()[-1])
print("ms/step %.4f" % d["ms_per_step"])
PY
