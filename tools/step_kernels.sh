# GPU box: kernels of ONE eager step, in order (usage: bash tools/step_kernels.sh [bench args])
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/stepk
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/stepk -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-proxy --no-graph --steps 3 --warmup 2 "$@" > /dev/null 2>&1
f=$(find /tmp/stepk -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last occurrence of the fwd transform kernel marks a step start; take the last full step: from the 3rd-last 'rows_chain2' fwd...
idx = [i for i, n in enumerate(names) if "multi_tensor_apply" in n]        # bucket.pack() closes a step
a, b = idx[-2] + 1, idx[-1] + 1
tot = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    print("%8.1f us  %s" % (d, n[:150]))
print("sum of kernel times %.1f us over %d launches" % (tot, b - a))
PY
