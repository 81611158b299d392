# GPU box: the kernel / layer tests + the default bench line summary; fails when the runtime reports a fault
cd $GRAFT_REPO_ROOT
o=gpurun_out/quick_check.txt
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers.py -x -q > $o 2>&1; rc=$?
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
print(json.dumps(d["config"]["strong_scaling_proxy"]))
PY
exit $rc
