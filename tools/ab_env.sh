#!/bin/bash
# GPU box: bench.py once per value of an environment variable, alternating, stderr kept.  usage: tools/ab_env.sh VAR v1 v2 ... [-- bench args]
VAR=$1; shift
VALS=()
while [ "$1" != "--" ] && [ -n "$1" ]; do VALS+=("$1"); shift; done
shift
k=0
for v in "${VALS[@]}"; do
  k=$((k + 1))
  env "$VAR=$v" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" > gpurun_out/ab_$k.json 2> gpurun_out/ab_$k.err || { echo "$VAR=$v failed"; tail -5 gpurun_out/ab_$k.err; exit 1; }
  python - gpurun_out/ab_$k.json "$VAR=$v" <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]; c = d["config"]
print(sys.argv[2], "ms/step %.3f  frac %.4f  conv_ms %.3f  index %.3f  shard %.4f  shard index %.3f  fresh-batch overlapped %.3f ms" % (
    d["ms_per_step"], r["frac"], r["kernel_ms_per_step"], c["index_build_ms"], c["strong_scaling_proxy"]["shard_ms_per_step"],
    c["strong_scaling_proxy"]["shard_index_build_ms"], c.get("fresh_batch_overlapped_ms_per_step") or 0))
PY
done
