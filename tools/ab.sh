#!/bin/bash
# usage: tools/ab.sh VAR v1 v2 ... -- [bench args]   : runs bench.py once per value of the env var VAR, prints ms/step etc.
# The knobs of the Python layer (DN_SELFSUM, DN_CHAIN2, DN_WGRAD_CHUNK, DN_F32_EXACT ...) work with the shipped
# library; the knobs INSIDE libdn_hip.so (DN_STRIDE, DN_NT, DN_TF_DEPTH, DN_TF_WGS, DN_WGRAD_DMA) need a tuning build first:
#   python -m dummynode4graphlearning_amd.csrc.build --tuning --force      (the default build has no environment access)
VAR=$1; shift
VALS=()
while [ "$1" != "--" ] && [ -n "$1" ]; do VALS+=("$1"); shift; done
shift
for v in "${VALS[@]}"; do
  env "$VAR=$v" python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('$VAR=$v', 'ms/step %.3f' % d['ms_per_step'], 'Medges/s %.1f' % (d['value'] / 1e6), 'frac %.4f' % r['frac'], 'conv_ms %.3f' % r['kernel_ms_per_step'], 'launches', r['launches_per_step'], 'parts', d['config']['sub_batches'])"
done
