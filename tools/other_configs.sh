cd $GRAFT_REPO_ROOT
for args in "--workload config3" "--workload config3 --act leaky_relu --regularizer bdd" "--workload config3 --dtype bf16" "--workload config3 --dtype bf16 --act leaky_relu --regularizer bdd" "--act leaky_relu --regularizer bdd --no-proxy" "--workload config4" "--dtype f32 --no-proxy"; do
  echo "== bench.py $args"
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 $args 2>gpurun_out/oc.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; c = d['config']
print('ms/step %.4f value %.4g %s scaling %s frac %.3f index_ms %s proxy %s' % (d['ms_per_step'], d['value'], d['unit'], d['scaling'], r['frac'], c.get('index_build_ms'), (c.get('strong_scaling_proxy') or {}).get('efficiency_at_8')))" || tail -5 gpurun_out/oc.err
done
