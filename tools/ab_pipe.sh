run() { env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'step_ms', round(d['ms_per_step'],3), 'conv_ms', round(d['roofline']['kernel_ms_per_step'],3), 'frac', round(d['roofline']['frac'],3), 'idx_ms', round(d['config']['index_build_ms'],1))"; }
for v in "$@"; do run $v; done
