# GPU box: the last N kernels of a bench run under HIP-graph replay, in order, with the gap to the previous kernel
# usage: bash tools/replay_kernels.sh N [bench args]
N=$1; shift
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/rpk
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rpk -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-proxy --no-secondary --steady --steps 6 --warmup 2 "$@" > /tmp/rpk.log 2>&1
f=$(find /tmp/rpk -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$N" <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
prev = None
cats = [i for i, r in enumerate(rows) if "CatArray" in r["Kernel_Name"] or "multi_tensor_apply" in r["Kernel_Name"]]
end = cats[-1] + 1 if cats else len(rows)          # (the step ends with bucket.pack(): the window ends there, not in the roofline leg)
for r in rows[max(0, end - int(sys.argv[2])):end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  gap %6.1f  %s" % ((e - s) / 1e3, ((s - prev) / 1e3) if prev else 0.0, re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:110]))
    prev = e
PY
tail -1 /tmp/rpk.log | cut -c1-200
