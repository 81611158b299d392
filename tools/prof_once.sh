# usage (on the GPU box): bash tools/prof_once.sh <tag> [ENV=VAL ...] -- per-kernel averages of one bench run under rocprofv3
tag=$1; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 $BENCH_ARGS > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -14 "$f" | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    print(r[0][:70].ljust(70), r[1:4])
"; else echo "no stats file"; tail -5 $out.log; fi
