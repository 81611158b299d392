# GPU box: default bench line (with the strong-scaling proxy) + per-kernel times of the 4096-graph shard step
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_proxy.json 2> gpurun_out/bench_proxy.err || { tail -5 gpurun_out/bench_proxy.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_proxy.json").read().strip().splitlines()[-1])
print("ms/step %.3f conv %.4f frac %.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"]))
print(json.dumps(d["config"]["strong_scaling_proxy"]))
PY
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4096 -o k -- python3 $GRAFT_REPO_ROOT/bench.py --graphs 4096 --no-cpu-baseline --no-proxy --steps 50 --warmup 5 > /dev/null 2>&1
f=$(find /tmp/prof4096 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-70s calls %5s avg %8.1f us  total %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
