#!/usr/bin/env python3
"""Collect the round's rocprofv3 evidence on the GPU box and write the summaries bench.py / DESIGN.md cite.

    python tools/collect_profiles.py --tag r01 [--out gpurun_out/profiles]

Runs (each as a child process, the profiled program directly after `--`):
  1. python bench.py                                             -> <tag>_bench_line.json
  2. rocprofv3 --kernel-trace --stats        -- python3 bench.py --steps 20 --warmup 5 --steady
                                                                 -> <tag>_kernel_stats.csv   (the replayed step and the roofline
                                                                    leg only: AverageNs of the conv kernels reproduces the line)
  3. rocprofv3 --kernel-trace --pmc FETCH_SIZE  -- python3 bench.py --steps 3 --warmup 1 --steady --no-graph
  4. rocprofv3 --kernel-trace --pmc WRITE_SIZE  -- (same)       -> <tag>_hbm_traffic_per_launch.csv, <tag>_traffic.json
  5. rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- (same)
                                                                 -> <tag>_lds_bank_conflicts.csv
PMC passes never share a run with --stats or any API trace (pool rule).  gfx950 correction (MI355X_MICROARCH.md, HBM
section): FETCH_SIZE (KB) counts half of a wide coalesced read, so hbm_read = 2 * FETCH_SIZE; WRITE_SIZE is exact.
Copy the files from the output directory into profiles/ and commit them.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dummynode4graphlearning_amd._lib import source_digest  # noqa: E402

# launches of ONE eager step of bench.py's config-5 workload, in order (kernel-name fragment, role)
STEP = [
    ("rows_transform_ring_kernel", "conv transform fwd, edge rows except the collapsed dummy relation (gathers x rows; weights read [k][n])"),
    ("rows_close_ring_kernel", "closing launch fwd (unit stream): self-loop transform + bias + per-dst row sums (selection MFMA) + per-graph column sums -> aux rows + AGG units (aux x W_agg added to the dummy nodes)"),
    ("rows_chain2_ring_kernel", "MLP forward: Linear+ReLU, Linear+ReLU in one pass (+ ReLU masks as bit tensors); LDS-DMA ring kernel"),
    ("rows_wgrad_dma_kernel", "MLP wgrad 2 (LDS-DMA ring, outer ReLU mask from bits)"),
    ("wgrad_reduce_kernel", "wgrad reduce"),
    ("rows_chain2_ring_kernel", "MLP input gradients: outer mask, dgrad 2, inner mask, dgrad 1 in one pass (masks from bits, by DMA)"),
    ("rows_wgrad_ls_kernel", "MLP wgrad 1 (LDS-DMA ring, loader waves + MFMA waves)"),
    ("wgrad_reduce_kernel", "wgrad reduce"),
    ("rows_transform_ring_kernel", "conv transform bwd, edge rows except the collapsed dummy relation (gathers g rows)"),
    ("rows_close_ring_kernel", "closing launch bwd (unit stream): self-loop transform + per-src row sums + per-graph column sums of g -> aux rows + AGG units"),
    ("rows_wgrad_ls_kernel", "conv wgrad (LDS-DMA ring, loader waves + MFMA waves; gathers x and g rows, interleaved pieces, + bias colsum of the self-loop rows)"),
    ("wgrad_reduce_kernel", "wgrad reduce"),
]
CONV_ROWS = (0, 1, 8, 9)
OURS = ("gather_segsum_vec_kernel", "overflow_rows_add_kernel", "rows_transform_ring_kernel", "rows_transform_kernel", "rows_close_ring_kernel",
        "rows_selfsum_kernel", "fold_tail_kernel", "conv_graphs_kernel", "rows_wgrad_multi_kernel", "rows_chain2_ring_kernel", "rows_chain2_kernel", "rows_wgrad_ls_kernel", "rows_wgrad_ix_kernel", "rows_wgrad_dma_kernel", "rows_wgrad_kernel",
        "rows_wgrad_f32s_multi_kernel", "wgrad_reduce_kernel")


def short(name):
    for k in OURS:
        if k in name:
            return k
    return None


def run(cmd, cwd=ROOT):
    print("+", " ".join(cmd), flush=True)
    return subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def one_csv(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no %s under %s" % (suffix, d))
    return hits[-1]


def counter_rows(path):
    """dispatch-ordered list of (kernel, {counter: value}, start_ns, end_ns)."""
    by = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            did = int(r["Dispatch_Id"])
            e = by.setdefault(did, [r["Kernel_Name"], {}, int(r.get("Start_Timestamp", 0) or 0), int(r.get("End_Timestamp", 0) or 0)])
            e[1][r["Counter_Name"]] = e[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [tuple(by[k]) for k in sorted(by)]


def last_step(rows):
    """index of the last full STEP pattern among our kernels (bf16 gather kernels only: the GIN leg is fp32)."""
    ours = [(i, short(r[0])) for i, r in enumerate(rows) if short(r[0]) and not ("gather_segsum" in r[0] and "float" in r[0])]
    names = [n for _, n in ours]
    want = [n for n, _ in STEP]
    for s in range(len(names) - len(want), -1, -1):
        if names[s:s + len(want)] == want:
            return [ours[s + j][0] for j in range(len(want))]
    raise SystemExit("step pattern not found in the trace")


def trace_rows(path):
    """kernel trace -> [(kernel name, start ns, end ns)] in start order."""
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out.sort(key=lambda t: t[1])
    return out


def write_window_stats(path, header, rows):
    """per-kernel Calls / TotalDurationNs / AverageNs of the launches in `rows` (the layout of rocprofv3's kernel_stats.csv)."""
    agg = {}
    for name, t0, t1 in rows:
        e = agg.setdefault(name, [0, 0])
        e[0] += 1
        e[1] += t1 - t0
    tot = sum(v[1] for v in agg.values()) or 1
    with open(path, "w") as f:
        for h in header:
            f.write("# " + h + "\n")
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage"\n')
        for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%d,%.1f,%.2f\n' % (name, c, d, d / c, 100.0 * d / tot))
    return tot


def split_step_and_leg(trace, steps, reps):
    """The timed regions of `bench.py --steady`: the last `steps` replays of the captured step (every launch between the first
    kernel of a STEP pattern and the pattern's last reduce, torch glue included) and, behind them, the roofline leg = the last
    reps x 4 conv launches (graph replays of ring, close, ring, close)."""
    conv = ("rows_transform_ring_kernel", "rows_close_ring_kernel")
    n_leg = reps * len(CONV_ROWS)
    tail = trace[-n_leg:]
    assert all(any(c in t[0] for c in conv) for t in tail), "the trace does not end with the roofline leg"
    body = trace[:-n_leg]
    want = [n for n, _ in STEP]
    ours = [(i, short(t[0])) for i, t in enumerate(body) if short(t[0])]
    names = [n for _, n in ours]
    found, s = [], len(names) - len(want)
    while s >= 0 and len(found) < steps:
        if names[s:s + len(want)] == want:
            found.append((ours[s][0], ours[s + len(want) - 1][0]))
            s -= len(want)
        else:
            s -= 1
    assert len(found) == steps, "found %d of %d step replays in the trace" % (len(found), steps)
    lo, hi = found[-1][0], found[0][1]
    return body[lo:hi + 1], tail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "profiles"))
    ap.add_argument("--skip-lds", action="store_true")
    ap.add_argument("--skip-others", action="store_true", help="config 5 only (no config-3 / PROTEINS / config-4 passes)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    tmp = os.path.join("/tmp", "dn_prof_" + a.tag)
    os.makedirs(tmp, exist_ok=True)
    env_note = "cd /tmp && export TMPDIR=/tmp"
    os.environ["TMPDIR"] = "/tmp"
    bench = os.path.join(ROOT, "bench.py")
    eager = ["python3", bench, "--steps", "3", "--warmup", "1", "--steady", "--no-graph"]

    # 2. kernel stats of the default (graph replay) run
    d = os.path.join(tmp, "stats")
    r = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "k", "--",
             "python3", bench, "--steps", "20", "--warmup", "5", "--steady"], cwd="/tmp")
    stats = one_csv(d, "kernel_stats.csv")
    with open(stats) as f:
        body = f.read()
    with open(os.path.join(a.out, a.tag + "_kernel_stats.csv"), "w") as f:
        f.write("# %s -- rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --steady\n" % a.tag)
        f.write("# (--steady: HIP-graph replay of the step + the roofline leg only -- no fresh-batch leg, no GIN leg, no proxy; the index\n")
        f.write("#  build of the one batch runs before the timed region under its own kernel names)\n")
        f.write("# (%s first) bench line of that run: %s\n" % (env_note, r.stdout.strip().splitlines()[-1][:400] if r.stdout.strip() else "n/a"))
        f.write(body)
    # the same run cut into its two timed regions (from the kernel trace): AverageNs x Calls / 20 of the first file reproduces
    # ms_per_step, of the second kernel_ms_per_step
    step_rows, leg_rows = split_step_and_leg(trace_rows(one_csv(d, "kernel_trace.csv")), 20, 20)
    t_step = write_window_stats(os.path.join(a.out, a.tag + "_kernel_stats_step.csv"),
                                ["%s -- the 20 timed replays of the captured step of the run above (kernel trace, all launches incl. torch glue)" % a.tag], step_rows)
    t_leg = write_window_stats(os.path.join(a.out, a.tag + "_kernel_stats_leg.csv"),
                               ["%s -- the 20 timed replays of the roofline leg of the run above (the conv's 4 launches: ring, close, ring, close)" % a.tag], leg_rows)
    span = lambda rows: (rows[-1][2] - rows[0][1]) / 1e6  # noqa: E731
    with open(os.path.join(a.out, a.tag + "_kernel_stats_step.csv"), "a") as f:
        f.write("# sum of kernel durations / 20 = %.4f ms per step; first start to last end / 20 = %.4f ms\n" % (t_step / 20e6, span(step_rows) / 20))
    with open(os.path.join(a.out, a.tag + "_kernel_stats_leg.csv"), "a") as f:
        f.write("# sum of kernel durations / 20 = %.4f ms per leg; first start to last end / 20 = %.4f ms\n" % (t_leg / 20e6, span(leg_rows) / 20))

    # 3./4. HBM traffic per launch
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(tmp, ctr)
        run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--"] + eager, cwd="/tmp")
        rows = counter_rows(one_csv(d, "counter_collection.csv"))
        idx = last_step(rows)
        vals[ctr] = [(rows[i][1][ctr], (rows[i][3] - rows[i][2]) / 1e3) for i in idx]
    lines, tot, conv = [], 0.0, 0.0
    for j, (kname, role) in enumerate(STEP):
        rd = 2.0 * vals["FETCH_SIZE"][j][0] * 1024 / 1e6          # KB -> MB, gfx950 half-count correction
        wr = vals["WRITE_SIZE"][j][0] * 1024 / 1e6
        us = 0.5 * (vals["FETCH_SIZE"][j][1] + vals["WRITE_SIZE"][j][1])
        lines.append("%s,%s,%.1f,%.1f,%.1f,%.1f,%.2f" % (kname, role, rd, wr, rd + wr, us, (rd + wr) / us if us else 0.0))
        tot += rd + wr
        if j in CONV_ROWS:
            conv += rd + wr
    N, E, H = 1015808, 3997696, 256
    alg = 2 * (E * H * 2 + N * H * 2 + 8 * E)
    with open(os.path.join(a.out, a.tag + "_hbm_traffic_per_launch.csv"), "w") as f:
        f.write("# %s -- HBM traffic per launch of one eager step (separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes)\n" % a.tag)
        f.write("# command: rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --steady --no-graph\n")
        f.write("# gfx950 correction (MI355X_MICROARCH.md, HBM): hbm_read = 2*FETCH_SIZE KB (a wide coalesced read is half-counted); WRITE_SIZE exact. durations under PMC collection (us)\n")
        f.write("kernel,role,hbm_read_MB,hbm_write_MB,hbm_total_MB,duration_us,TB_per_s\n")
        f.write("\n".join(lines) + "\n")
        f.write("# step total %.1f MB; conv gather-scatter launches (rows 1-2 and 9-10): %.1f MB vs %.1f MB algorithmic (SURVEY 8d) = %.2fx\n"
                % (tot, conv, alg / 1e6, conv * 1e6 / alg))
    with open(os.path.join(a.out, a.tag + "_traffic.json"), "w") as f:
        json.dump({"workload": "config5", "N": N, "E": E, "H": H, "dtype": "bf16",
                   "conv_gather_scatter_hbm_bytes_per_step": conv * 1e6, "step_hbm_bytes": tot * 1e6,
                   "source_sha16": source_digest(),
                   "source": "profiles/%s_hbm_traffic_per_launch.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per MI355X_MICROARCH.md)" % a.tag},
                  f, indent=1)

    # 5. LDS bank conflicts
    if not a.skip_lds:
        d = os.path.join(tmp, "lds")
        run(["rocprofv3", "--kernel-trace", "--pmc", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "--output-format", "csv",
             "-d", d, "-o", "p", "--"] + eager, cwd="/tmp")
        agg = {}
        for kname, ctrs, _, _ in counter_rows(one_csv(d, "counter_collection.csv")):
            k = short(kname)
            if k:
                e = agg.setdefault(k, [0.0, 0.0])
                e[0] += ctrs.get("SQ_LDS_BANK_CONFLICT", 0.0)
                e[1] += ctrs.get("SQ_LDS_IDX_ACTIVE", 0.0)
        with open(os.path.join(a.out, a.tag + "_lds_bank_conflicts.csv"), "w") as f:
            f.write("# %s -- LDS bank conflicts (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE), summed over all launches of the eager run\n" % a.tag)
            f.write("kernel,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,conflict_fraction\n")
            for k in sorted(agg):
                c, act = agg[k]
                f.write("%s,%d,%d,%.3f\n" % (k, c, act, c / act if act else 0.0))

    # 6. the other configurations bench.py knows (the driver times config 5 alone): kernel statistics of the replayed step and LDS
    #    conflicts of an eager run -- config 3 (the SI defaults' size: H = 64, 512 graphs of ~50 nodes) in bf16 and fp32, the
    #    PROTEINS-shaped RGIN workload (H = 256, graphs over 32 nodes: the non-absorbed fold), config 4 (GC model step)
    if not a.skip_others:
        for name, extra in (("config3_bf16", ["--workload", "config3", "--dtype", "bf16"]), ("config3_f32", ["--workload", "config3"]),
                            ("proteins", ["--workload", "proteins"]), ("config4", ["--workload", "config4"])):
            d = os.path.join(tmp, "stats_" + name)
            cmd = ["python3", bench] + extra + ["--steps", "20", "--warmup", "5", "--steady", "--no-cpu-baseline"]
            r = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "k", "--"] + cmd, cwd="/tmp")
            try:
                with open(one_csv(d, "kernel_stats.csv")) as f:
                    body = f.read()
            except SystemExit:
                print("no kernel stats for", name, r.stderr[-500:])
                continue
            with open(os.path.join(a.out, "%s_%s_kernel_stats.csv" % (a.tag, name)), "w") as f:
                f.write("# %s -- rocprofv3 --kernel-trace --stats --output-format csv -- %s\n" % (a.tag, " ".join(["python3", "bench.py"] + cmd[2:])))
                f.write("# (whole process: batch and index build, warm-up, 20 timed replays of the captured step)\n")
                f.write("# bench line of that run: %s\n" % (r.stdout.strip().splitlines()[-1][:600] if r.stdout.strip() else "n/a"))
                f.write(body)
            if a.skip_lds or name == "config4":
                continue
            d = os.path.join(tmp, "lds_" + name)
            run(["rocprofv3", "--kernel-trace", "--pmc", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "--output-format", "csv", "-d", d,
                 "-o", "p", "--", "python3", bench] + extra + ["--steps", "3", "--warmup", "1", "--steady", "--no-graph", "--no-cpu-baseline"],
                cwd="/tmp")
            agg = {}
            try:
                rows = counter_rows(one_csv(d, "counter_collection.csv"))
            except SystemExit:
                continue
            for kname, ctrs, _, _ in rows:
                k = kname.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
                e = agg.setdefault(k, [0.0, 0.0])
                e[0] += ctrs.get("SQ_LDS_BANK_CONFLICT", 0.0)
                e[1] += ctrs.get("SQ_LDS_IDX_ACTIVE", 0.0)
            with open(os.path.join(a.out, "%s_%s_lds_bank_conflicts.csv" % (a.tag, name)), "w") as f:
                f.write("# %s -- LDS bank conflicts of an eager run of bench.py %s (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE), all launches incl. the index build\n" % (a.tag, " ".join(extra)))
                f.write("kernel,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,conflict_fraction\n")
                for k in sorted(agg):
                    c, act = agg[k]
                    if act > 0:
                        f.write("%s,%d,%d,%.3f\n" % (k, c, act, c / act))

    # 1. the bench line itself, last, with the fresh traffic.json in place (bench.py reads profiles/<tag>_traffic.json)
    import shutil
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    shutil.copy(os.path.join(a.out, a.tag + "_traffic.json"), os.path.join(ROOT, "profiles", a.tag + "_traffic.json"))
    r = run(["python3", bench])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        sys.stderr.write(r.stderr[-2000:])
        raise SystemExit("bench.py printed no JSON line")
    with open(os.path.join(a.out, a.tag + "_bench_line.json"), "w") as f:
        f.write(line[-1] + "\n")
    print(line[-1][:600])


if __name__ == "__main__":
    main()
