#!/usr/bin/env python3
"""Experiment (tuning build: python -m dummynode4graphlearning_amd.csrc.build --tuning --force): where does the H = 256 closing
launch spend its time?  dn_rows_close_bf16 on the config-5 index with DN_CLOSE_ABL ablations, interleaved in one process.
bits: 1 entry rows from L2 (row & 1023), 2 x rows from L2, 4 no stores, 8 entry units fetched but not summed.
usage (GPU box): python tools/close_exp.py [rounds]
       python tools/close_exp.py --spread      (build with DN_BUILD_EXTRA=-DDN_CLOSE_TIMES: wall time of every workgroup; GRAPHS=4096
                                                for an eighth of the batch)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dummynode4graphlearning_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
graphs = int(os.environ.get("GRAPHS", "32768"))
workload = os.environ.get("WORKLOAD", "config5")          # WORKLOAD=proteins GRAPHS=16384: bench.py --workload proteins' batch
g, raw, _ = bench.build_batch(dev, 5, graphs, workload)
N, H, R = g.number_of_nodes(), 256, 16
ix = g.row_index(g.edata["label"], R, True).parts[0][2]
P = ix.num_edge_rows
x = torch.randn(N, H, device=dev).to(torch.bfloat16)
W = (torch.randn(H, H, device=dev) * 0.05).to(torch.bfloat16)
Y = torch.randn(P, H, device=dev).to(torch.bfloat16)
out = torch.empty_like(x)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
for d in ("f", "b"):
    cu = ix.close_units(d)
    fold = ops._row_index_fold(ix, d, "units")
    seg = agg = None
    if fold is not None and cu.agg:          # absorbed fold: AGG units at the end of every workgroup's stream
        aux = torch.empty((fold.n, H), dtype=x.dtype, device=dev)
        agg = (fold.graph_tiles[1], W, aux, fold.add_idx)
    elif fold is not None:
        seg = (fold.fold_info, torch.empty((fold.num_parts, H), dtype=torch.float32, device=dev))
    nu = int(cu.unit_ptr[-1])
    print("direction %s: %d units for %d tiles (%.2f per tile), fold %s" % (d, nu, cu.num_tiles, nu / cu.num_tiles, "absorbed" if agg else ("partial rows" if seg else "none")))

    def timed(abl, reps=20):
        os.environ["DN_CLOSE_ABL"] = str(abl)
        for _ in range(3):
            ops.rows_close(x, W, None, Y, cu, out=out, seg=seg, w_kn=True, agg=agg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.rows_close(x, W, None, Y, cu, out=out, seg=seg, w_kn=True, agg=agg)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    if "--spread" in sys.argv:              # -DDN_CLOSE_TIMES build: wall time of every workgroup's compute loop
        import ctypes
        import numpy as np
        from dummynode4graphlearning_amd import _lib
        h = ctypes.CDLL(_lib.LIB_PATH)
        t_all = timed(0)
        buf = (ctypes.c_ulonglong * (256 * 3))()
        torch.cuda.synchronize()
        assert h.dn_debug_close_times(buf) == 0
        a = np.array(buf, dtype=np.float64).reshape(256, 3)
        wall, start, nu_wg = (a[:, 1] - a[:, 0]) / 100.0, (a[:, 0] - a[:, 0].min()) / 100.0, a[:, 2]
        end = start + wall
        print("  launch %.1f us; workgroups: wall min %.1f median %.1f max %.1f us, start skew %.1f, last end %.1f; units per workgroup %d .. %d"
              % (t_all, wall.min(), np.median(wall), wall.max(), start.max(), end.max(), nu_wg.min(), nu_wg.max()))
        print("  per XCD median us: %s" % " ".join("%.0f" % np.median(wall[k::8]) for k in range(8)))
        order = np.argsort(-wall)
        print("  slowest: %s" % "  ".join("b%d:%.0fus/%du" % (b, wall[b], nu_wg[b]) for b in order[:10]))
        print("  fastest: %s" % "  ".join("b%d:%.0fus/%du" % (b, wall[b], nu_wg[b]) for b in order[-6:]))
        print("  us per unit: median %.3f, slowest ten %.3f" % (np.median(wall / nu_wg), np.median((wall / nu_wg)[order[:10]])))
        continue
    res = {}
    for _ in range(rounds):
        for abl in ([int(v) for v in os.environ["ABLS"].split(",")] if os.environ.get("ABLS") else (0, 16, 0, 16)):
            res.setdefault(abl, []).append(timed(abl))
    for abl, v in res.items():
        print("  DN_CLOSE_ABL=%2d: %s us (min %.1f)" % (abl, " ".join("%.1f" % t for t in v), min(v)))
