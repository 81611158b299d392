# GPU box: kernels of ONE fresh-batch index build (RowIndex + closing tables), in order.  usage: bash tools/index_kernels.sh config3|config5|proteins
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/idxk
cat > /tmp/idx_once.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from dummynode4graphlearning_amd import ops
w = sys.argv[1]
dev = torch.device("cuda:0")
g, raw, aug_ms = bench.build_batch(dev, {"config5": 5, "config3": 3, "proteins": 2}[w], {"config5": 32768, "config3": 512, "proteins": 16384}[w], w)
etype = g.edata["label"]; R = {"config5": 16, "config3": 8, "proteins": 16}[w]
for it in range(4):
    g._cache.clear(); torch.cuda.synchronize()
    marker = torch.zeros(7, device=dev) + 1          # marks the start of a build in the trace
    ix = g.row_index(etype, R, True, closing_hint=(64 if w == "config3" else 256, torch.bfloat16))
    for _, _, part in ix.parts:
        ops.prepare_closing(part, 64 if w == "config3" else 256, torch.bfloat16)       # what bench.py's index_build_ms covers
    torch.cuda.synchronize()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/idxk -o k -- python3 /tmp/idx_once.py $1 > /dev/null 2>&1
f=$(find /tmp/idxk -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "elementwise" in n and "add" in n.lower() or "AddFunctor" in n or "CUDAFunctorOnSelf_add" in n]
a = marks[-1] + 1
tot = 0.0
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    print("%8.1f us @%8.1f  %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, n[:120]))
print("kernel time %.1f us over %d launches; span %.1f us" % (tot, len(rows) - a, (int(rows[-1]["End_Timestamp"]) - t0) / 1e3))
PY
