# usage (GPU box): bash tools/pmc_script.sh <tag> "<counters>" <script.py> [args] -- PMC counters per kernel (own pass, kernel-trace only)
tag=$1; ctrs=$2; shift; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/$@ > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:6]:
    print(k)
    for c, v in d.items():
        print("   %-28s %14.0f per launch" % (c, v / cnt[(k, c)]))
PY
