#!/bin/bash
# The step time of the other configurations bench.py knows (BASELINE.md section 2's GPU column), one line each.
run() { echo -n "$* : "; python bench.py --steady "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'frac %s' % r.get('frac'))"; }
run --workload config3
run --workload config3 --act leaky_relu --regularizer bdd
run --workload config3 --dtype bf16
run --workload config3 --dtype bf16 --act leaky_relu --regularizer bdd
run --act leaky_relu --regularizer bdd
run --dtype f32 --steps 5 --warmup 2
python bench.py --workload config4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config4 : ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'])"
