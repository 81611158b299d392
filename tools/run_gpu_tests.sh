# GPU box: the whole -m gpu suite + the default bench line; fails when the runtime reports a fault
cd $GRAFT_REPO_ROOT
o=gpurun_out/gpu_tests.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $o 2>&1; rc=$?
tail -5 $o
grep -q "Memory access fault" $o && exit 1
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err; rc=$?
grep -q "Memory access fault" gpurun_out/bench_line.err && exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step %.3f  G edges/s %.3f  conv_ms %.4f  frac %.4f  index_ms %.3f  overlapped %.3f  gin frac %.3f" % (
    d["ms_per_step"], d["value"] / 1e9, r["kernel_ms_per_step"], r["frac"], d["config"]["index_build_ms"],
    (d["config"]["edges_per_s_fresh_batch_overlapped"] or 0) / 1e9, d["secondary"]["gin_conv_gather"]["roofline"]["frac"]))
PY
exit $rc
