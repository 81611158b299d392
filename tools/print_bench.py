import json
d = json.loads(open("gpurun_out/bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step %.3f  G edges/s %.3f  conv_ms %.4f  frac %.4f  index_ms %.3f  overlapped %.3f  gin frac %.3f" % (
    d["ms_per_step"], d["value"] / 1e9, r["kernel_ms_per_step"], r["frac"], d["config"]["index_build_ms"],
    (d["config"]["edges_per_s_fresh_batch_overlapped"] or 0) / 1e9, d["secondary"]["gin_conv_gather"]["roofline"]["frac"]))
p = d["config"]["strong_scaling_proxy"]
print("proxy: shard %.4f ms, index %.3f ms, efficiency %.3f" % (p["shard_ms_per_step"], p["shard_index_build_ms"], p["efficiency_at_8"]))
