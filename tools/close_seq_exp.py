#!/usr/bin/env python3
"""Experiment: what would the H = 256 closing launch gain if the product rows lay in CONSUMPTION order (entry i reads row i:
16 KB contiguous per unit instead of ~2 KB runs in 16 relation segments)?  dn_rows_close_bf16 on the config-5 index with the
entry rows replaced by their positions (numerically meaningless, same bytes).  usage (GPU box): python tools/close_seq_exp.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dummynode4graphlearning_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g, raw, _ = bench.build_batch(dev, 5, int(os.environ.get("GRAPHS", "32768")), "config5")
N, H, R = g.number_of_nodes(), 256, 16
ix = g.row_index(g.edata["label"], R, True, closing_hint=(256, torch.bfloat16)).parts[0][2]
P = ix.num_edge_rows
x = torch.randn(N, H, device=dev).to(torch.bfloat16)
W = (torch.randn(H, H, device=dev) * 0.05).to(torch.bfloat16)
Y = torch.randn(P, H, device=dev).to(torch.bfloat16)
out = torch.empty_like(x)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for d in ("f", "b"):
    cu = ix.close_units(d)
    fold = ops._row_index_fold(ix, d, "units")
    aux = torch.empty((fold.n, H), dtype=x.dtype, device=dev)
    agg = (fold.graph_tiles[1], W, aux, fold.add_idx)
    real = cu.ent_row
    L = real.numel()
    seq = (torch.arange(L, device=dev, dtype=torch.int64) % P).to(torch.int32)
    # rows in consumption order but COMPACT (only the kept entries count): position among the covered entries
    res = {"as built": [], "entry i reads row i": []}
    for _ in range(3):
        for name, rows in (("as built", real), ("entry i reads row i", seq)):
            cu.ent_row = rows
            res[name].append(timed(lambda: ops.rows_close(x, W, None, Y, cu, out=out, w_kn=True, agg=agg)))
    cu.ent_row = real
    for k, v in res.items():
        print("direction %s, %-22s: %s us (min %.1f)" % (d, k, " ".join("%.1f" % t for t in v), min(v)), flush=True)
