cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in 0 1; do
DN_SWEEP=$v timeout -k 10 200 python bench.py --steady --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('DN_SWEEP=$v ms/step %.3f conv_ms %.4f frac %.4f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac']))"
done; done
