# usage (GPU box): bash tools/prof_script.sh <tag> <script.py> [args] -- per-kernel stats of any script under rocprofv3
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 $GRAFT_REPO_ROOT/$@ > $out.log 2>&1
grep -v "^W2026\|simple_timer" $out.log | tail -8
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -${TOPN:-24} "$f" | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    print(r[0][:110].ljust(110), r[1:4])
"; else echo "no stats file"; fi
