# GPU box: per-kernel averages of the replayed step at an eighth of config 5 next to an eighth of the full-size averages
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/shk_a /tmp/shk_b
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/shk_a -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steady --steps 100 --warmup 20 --graphs 4096 > /tmp/shk_a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/shk_b -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steady --steps 100 --warmup 20 > /tmp/shk_b.log 2>&1
python3 - <<'PY'
import csv, glob, re
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n)[:40]
        if int(r["Calls"]) >= 100:
            out[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return out
a, b = load("/tmp/shk_a"), load("/tmp/shk_b")
tot_a = tot_i = 0.0
for n in sorted(a, key=lambda k: -a[k][0] * a[k][1]):
    if n in b:
        ca, ua = a[n]; cb, ub = b[n]
        per_step = ca / 120.0
        print("%-42s x%.1f/step  shard %7.1f us   full/8 %7.1f us   excess %6.1f us/step" % (n, per_step, ua, ub / 8, (ua - ub / 8) * per_step))
        tot_a += ua * per_step; tot_i += ub / 8 * per_step
print("kernel time per step: shard %.1f us, full/8 %.1f us" % (tot_a, tot_i))
PY
grep -o '"ms_per_step": [0-9.]*' /tmp/shk_a.log /tmp/shk_b.log
