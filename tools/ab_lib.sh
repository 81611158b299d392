#!/bin/bash
# usage: tools/ab_lib.sh LIB_A LIB_B [rounds] -- [bench args]
# Two builds of libdn_hip.so on ONE box, interleaved (A B A B ...): ms/step of `bench.py --steady` per run.  Box-to-box spread
# (3 %) is larger than most kernel changes, so only same-box pairs count.
A=$1; B=$2; shift 2
R=2
if [ "$1" != "--" ] && [ -n "$1" ]; then R=$1; shift; fi
shift
for i in $(seq $R); do
  for L in "$A" "$B"; do
    DN_HIP_LIB=$(realpath "$L") python bench.py --steady --steps 30 --warmup 5 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('$L', 'ms/step %.4f' % d['ms_per_step'], 'conv_ms %.3f' % r['kernel_ms_per_step'], 'frac %.4f' % r['frac'])"
  done
done
