# GPU box: the kernel / layer tests + the default bench line summary + per-kernel averages of a --steady run; fails on a GPU fault
cd $GRAFT_REPO_ROOT
o=gpurun_out/quick_check.txt
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers.py tests/test_gpu_fuzz.py -x -q > $o 2>&1; rc=$?
tail -3 $o
grep -q "Memory access fault" $o && exit 1
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err; rc=$?
grep -q "Memory access fault" gpurun_out/bench_line.err && exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step %.3f  G edges/s %.3f  conv_ms %.4f  frac %.4f  index_ms %.3f  overlapped %.3f  gin frac %.3f" % (
    d["ms_per_step"], d["value"] / 1e9, r["kernel_ms_per_step"], r["frac"], d["config"]["index_build_ms"],
    (d["config"]["edges_per_s_fresh_batch_overlapped"] or 0) / 1e9, d["secondary"]["gin_conv_gather"]["roofline"]["frac"]))
p = d["config"]["strong_scaling_proxy"]
print("proxy: shard %.4f ms, index %.3f ms, efficiency %.3f" % (p["shard_ms_per_step"], p["shard_index_build_ms"], p["efficiency_at_8"]))
PY
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/qc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qc -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steady --steps 20 > /dev/null 2>&1
python3 - $(find /tmp/qc -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-60s calls %4s avg %8.1f us min %8.1f" % (r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
exit $rc
