# GPU box: the last N kernels of an eager bench run, in order (usage: bash tools/last_kernels.sh N [bench args])
N=$1; shift
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/lastk
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lastk -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-proxy --no-graph --steps 3 --warmup 2 "$@" > /dev/null 2>&1
f=$(find /tmp/lastk -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$N" <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-int(sys.argv[2]):]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%8.1f us  %s" % (d, re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:120]))
PY
