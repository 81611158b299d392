#!/usr/bin/env python3
"""Experiment: do the conv launches care where their buffers lie relative to each other (HBM channel interleaving)?  Transform +
closing launch of config 5 (forward tables, absorbed fold) with the product rows Y, the output and the input shifted by a few KB.
usage (GPU box): python tools/align_exp.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dummynode4graphlearning_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g, raw, _ = bench.build_batch(dev, 5, 32768, "config5")
N, H, R = g.number_of_nodes(), 256, 16
ix = g.row_index(g.edata["label"], R, True, closing_hint=(256, torch.bfloat16)).parts[0][2]
P = ix.num_edge_rows
Wrel = (torch.randn(R, H, H, device=dev) * 0.05).to(torch.bfloat16)
Wl = (torch.randn(H, H, device=dev) * 0.05).to(torch.bfloat16)
fold = ops._row_index_fold(ix, "f", "units")
tiles = ops._conv_tiles_for(ix, fold, H, torch.bfloat16)
cu = ix.close_units("f")
PAD = 4 << 20                                            # elements of slack in front of every buffer
xbuf = torch.randn(N * H + PAD, device=dev).to(torch.bfloat16)
ybuf = torch.empty(ix.num_rows * H + PAD, device=dev, dtype=torch.bfloat16)
obuf = torch.empty(N * H + PAD, device=dev, dtype=torch.bfloat16)
aux = torch.empty((fold.n, H), dtype=torch.bfloat16, device=dev)
print("bases mod 2 MB (KB): x %d, y %d, out %d" % tuple((t.data_ptr() % (2 << 20)) // 1024 for t in (xbuf, ybuf, obuf)), flush=True)


def at(buf, off_bytes, rows):
    o = off_bytes // 2
    return buf[o:o + rows * H].view(rows, H)


def timed(x, Y, out, reps=20):
    def pair():
        ops.rows_transform(x, Wrel, tiles, P, idx=ix.row_in, tag="conv", out=Y, w_kn=True)
        ops.rows_close(x, Wl, None, Y[:P], cu, out=out, w_kn=True, agg=(fold.graph_tiles[1], Wrel[fold.rel], aux, fold.add_idx))
    for _ in range(3):
        pair()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pair()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


base = None
for name, (ox, oy, oo) in [("aligned", (0, 0, 0))] + [("y + %d KB" % k, (0, k * 1024, 0)) for k in (1, 4, 16, 64, 256, 1024)] + \
        [("out + %d KB" % k, (0, 0, k * 1024)) for k in (4, 64, 1024)] + [("x + %d KB" % k, (k * 1024, 0, 0)) for k in (4, 64, 1024)] + \
        [("aligned again", (0, 0, 0))]:
    x, Y, out = at(xbuf, ox, N), at(ybuf, oy, ix.num_rows), at(obuf, oo, N)
    t = [timed(x, Y, out) for _ in range(3)]
    print("%-16s transform + closing launch: %s us (min %.1f)" % (name, " ".join("%.1f" % v for v in t), min(t)), flush=True)
