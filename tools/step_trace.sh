# usage (GPU box): bash tools/step_trace.sh -- the kernels of ONE eager step of bench.py in launch order (name, us)
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/steptrace
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $out.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# one step = from a "conv transform fwd" (first rows_transform after a fill/zero) to the next one two transforms later
idx = [i for i, n in enumerate(names) if "rows_transform_kernel" in n]
# steps make 2 transform launches each; take the span between the 3rd-from-last pair in the eager phase
a, b = idx[4], idx[6]
for r in rows[a:b]:
    n = r["Kernel_Name"]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    print("%8.1f us  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n[:110]))
print("span %.1f us, kernels %d, busy %.1f us" % ((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3, b - a,
      sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3))
PY
