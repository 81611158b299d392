#!/usr/bin/env python3
"""Experiment: does the closing launch gain when it STARTS where the transform launch ended?  The sweep order gives every XCD
(workgroups b = 8 j + x) an eighth of the node range, walked upwards; the closing launch walks the whole batch upwards as one
front (workgroup w takes tiles w, w + 256, ...), so what the transform wrote last -- still in the Infinity Cache / the XCD's L2 --
is read last.  Variant: workgroup w = 8 j + x takes the tiles of eighth x DOWNWARDS from its end (j + 32 n from the top).
Unit streams re-assembled on the host from the built one, without the AGG units (FOLD = 0: the same for both orders).
usage (GPU box): python tools/close_order_exp.py"""
import os
import sys

import numpy as np
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
Wrel = (torch.randn(R, H, H, device=dev) * 0.05).to(torch.bfloat16)
Wl = (torch.randn(H, H, device=dev) * 0.05).to(torch.bfloat16)
Y = torch.empty(ix.num_rows, H, device=dev, dtype=torch.bfloat16)
out = torch.empty_like(x)
fold = ops._row_index_fold(ix, "f", "units")
tiles = ops._conv_tiles_for(ix, fold, H, torch.bfloat16)
cu0 = ix.close_units("f")
G = cu0.num_wg
up = cu0.unit_ptr.cpu().numpy()
U = cu0.units.cpu().numpy()[:up[-1]]
per_tile = {}
for w in range(G):
    cur = None
    for u in U[up[w]:up[w + 1]]:
        if u[0] & (4 | 8):            # AGG / NOP
            continue
        if not (u[0] & 1):            # X unit: a new tile
            cur = int(u[3])
            per_tile[cur] = []
        per_tile[cur].append(u)
T = cu0.num_tiles
assert len(per_tile) == T


def stream(order_of_wg):
    ptr, rows = [0], []
    for w in range(G):
        for t in order_of_wg(w):
            rows.extend(per_tile[t])
        ptr.append(len(rows))
    cu = ops.CloseUnits()
    cu.num_wg, cu.num_nodes, cu.num_tiles, cu.agg = G, N, T, False
    cu.unit_ptr = torch.tensor(ptr, dtype=torch.int32, device=dev)
    cu.units = torch.from_numpy(np.stack(rows).astype(np.int32)).to(dev)
    cu.ent_row, cu.ent_mask = cu0.ent_row, cu0.ent_mask
    return cu


def upwards(w):
    return range(w, T, G)


def eighth_down(w):
    xcd, j = w % 8, w // 8
    lo, hi = xcd * T // 8, (xcd + 1) * T // 8
    return range(hi - 1 - j, lo - 1, -(G // 8))


def eighth_up(w):
    xcd, j = w % 8, w // 8
    lo, hi = xcd * T // 8, (xcd + 1) * T // 8
    return range(lo + j, hi, G // 8)


variants = {"one front, upwards (as built)": stream(upwards), "eighth of my XCD, downwards": stream(eighth_down),
            "eighth of my XCD, upwards": stream(eighth_up)}
ref = None
for name, cu in variants.items():
    ops.rows_close(x, Wl, None, Y[:P], cu, out=out, w_kn=True)
    torch.cuda.synchronize()
    if ref is None:
        ops.rows_transform(x, Wrel, tiles, P, idx=ix.row_in, tag="conv", out=Y, w_kn=True)
        ops.rows_close(x, Wl, None, Y[:P], cu, out=out, w_kn=True)
        ref = out.clone()
    else:
        ops.rows_close(x, Wl, None, Y[:P], cu, out=out, w_kn=True)
        assert torch.equal(out, ref), name


def timed(cu, reps=20):
    def pair():
        ops.rows_transform(x, Wrel, tiles, P, idx=ix.row_in, tag="conv", out=Y, w_kn=True)
        ops.rows_close(x, Wl, None, Y[:P], cu, out=out, w_kn=True)
    for _ in range(3):
        pair()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pair()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


res = {k: [] for k in variants}
for _ in range(4):
    for k, cu in variants.items():
        res[k].append(timed(cu))
for k, v in res.items():
    print("transform + closing launch, %-32s: %s us (min %.1f)" % (k, " ".join("%.1f" % t for t in v), min(v)), flush=True)
