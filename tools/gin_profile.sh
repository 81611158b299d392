# usage (GPU box): bash tools/gin_profile.sh  -- kernel stats + HBM traffic (separate --pmc passes) of the GIN gather leg
# (tools/gin_gather_bench.py: neighbor_sum fwd+bwd on 16384 PROTEINS-shaped dummy graphs, H = 128 fp32) -> gpurun_out/r02_gin_gather.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/gin_prof
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 $R/tools/gin_gather_bench.py > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 $R/tools/gin_gather_bench.py > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python3 $R/tools/gin_gather_bench.py > $out/write.log 2>&1
python3 - $out <<'PY' > $R/gpurun_out/r02_gin_gather.txt
import csv, glob, sys, collections
out = sys.argv[1]
print("# r02 -- GIN conv gather leg: ops.neighbor_sum forward + backward on 16384 PROTEINS-shaped dummy graphs (N = 656k, E = 3.66 M, H = 128 fp32)")
print("# commands: rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 tools/gin_gather_bench.py")
print("# bench line of the stats run:", [l for l in open(out + "/stats.log") if l.startswith("GIN gather")][-1].strip())
f = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]
print("# kernel stats (gather kernels only): name, calls, average ns")
for r in csv.DictReader(open(f)):
    if "gather_segsum" in r["Name"] or "indexFunc" in r["Name"] or "graph_tile_sum" in r["Name"] or "gather_rows_sum" in r["Name"]:
        print("%s,%s,%.0f" % (r["Name"][:90].replace(",", ";"), r["Calls"], float(r["AverageNs"])))
tot = {}
for ctr, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr and any(k in r["Kernel_Name"] for k in ("gather_segsum", "graph_tile_sum", "gather_rows_sum")):
            k = r["Kernel_Name"][:60]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    tot[ctr] = agg
print("# HBM traffic per launch (MB): hbm_read = 2 x FETCH_SIZE KB (gfx950 half-count correction, MI355X_MICROARCH.md), hbm_write = WRITE_SIZE KB")
print("kernel,launches,hbm_read_MB_per_launch,hbm_write_MB_per_launch")
for k in tot["FETCH_SIZE"]:
    fr, n = tot["FETCH_SIZE"][k]; wr, _ = tot["WRITE_SIZE"].get(k, [0.0, 1])
    print("%s,%d,%.1f,%.1f" % (k.replace(",", ";"), n, 2 * fr / n * 1024 / 1e6, wr / n * 1024 / 1e6))
print("# compulsory bytes per direction: x read 336 MB + out written 336 MB + int32 index 14.6 MB + bounds 2.6 MB = 0.69 GB")
PY
cat $R/gpurun_out/r02_gin_gather.txt
