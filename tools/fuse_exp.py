#!/usr/bin/env python3
"""Experiment (tuning build): is a role split worth building?  The conv's two launches of one direction -- ring transform (T)
and unit-stream closing launch (C) -- run (a) one after the other on 256 workgroups each, as shipped, and (b) SIDE BY SIDE on
two streams, T on 8 x wt workgroups (sweep order with wt workgroups per XCD group) and C on the other 256 - 8 wt.  In (b) C reads
the Y of the previous launch (the same values: the inputs do not change), so the traffic is that of a fused launch without
its hand-off.  usage (GPU box): python tools/fuse_exp.py [wt ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dummynode4graphlearning_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
graphs = int(os.environ.get("GRAPHS", "32768"))
g, raw, _ = bench.build_batch(dev, 5, graphs, "config5")
N, H, R = g.number_of_nodes(), 256, 16
ix = g.row_index(g.edata["label"], R, True).parts[0][2]
P = ix.num_edge_rows
x = torch.randn(N, H, device=dev).to(torch.bfloat16)
Wrel = (torch.randn(R, H, H, device=dev) * 0.05).to(torch.bfloat16)
W = (torch.randn(H, H, device=dev) * 0.05).to(torch.bfloat16)
Y = torch.zeros(P, H, device=dev).to(torch.bfloat16)
out = torch.empty_like(x)
wts = [int(a) for a in sys.argv[1:]] or [15, 16]
win = os.environ.get("WINDOW", "0") == "1"        # Y written into and read from a 64 MB window (what a perfect hand-off through the Infinity Cache could give)
if win:
    os.environ["DN_TF_ABL"], os.environ["DN_CLOSE_ABL"] = "64", "32"
if os.environ.get("PLAIN_Y", "0") == "1":
    os.environ["DN_TF_NT"] = "0"                  # plain (not streaming) stores of Y
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for d in ("f", "b"):
    fold = ops._row_index_fold(ix, d, "units")
    cu = ix.close_units(d)
    assert fold is not None and cu.agg
    idx_rows = ix.row_in if d == "f" else ix.row_out
    lists = (ix.dst_ptr, ix.dst_rows) if d == "f" else (ix.src_ptr, ix.src_rows)
    aux = torch.empty((fold.n, H), dtype=x.dtype, device=dev)
    agg = (fold.graph_tiles[1], W, aux, fold.add_idx)
    sweep = ops._conv_tiles_for(ix, fold, H, x.dtype)

    def T(tab, wgs=0):
        os.environ["DN_TF_RING_WGS"] = str(wgs)
        ops.rows_transform(x, Wrel, tab, P, idx=idx_rows, tag="conv", out=Y, w_kn=True)

    def C(units):
        ops.rows_close(x, W, None, Y, units, out=out, w_kn=True, agg=agg)

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    T(sweep); C(cu)
    torch.cuda.synchronize()
    ref = out.clone()
    print("direction %s: T 256 wgs %.1f us, C 256 wgs %.1f us, T;C %.1f us" % (
        d, timed(lambda: T(sweep)), timed(lambda: C(cu)), timed(lambda: (T(sweep), C(cu)))), flush=True)
    for wt in wts:
        tabw = ops.build_sweep_tables(ix.rel_ptr_dev, R, ix.row_in, ix.row_out, N, P, skip_mask=1 << fold.rel, wg_per_group=wt)
        ncw = 256 - 8 * wt
        cuw = ops.build_close_units(*lists, N, P, drop=(fold.beg, fold.end), num_wg=ncw, tile_ptr=fold.graph_tiles[0], agg=True)

        def both():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                T(tabw, 8 * wt)
            with torch.cuda.stream(s2):
                C(cuw)
            cur.wait_stream(s1); cur.wait_stream(s2)

        tT, tC, tB = timed(lambda: T(tabw, 8 * wt)), timed(lambda: C(cuw)), timed(both)
        both()
        torch.cuda.synchronize()
        err = (out.float() - ref.float()).abs().max().item()
        print("  wt %2d: T alone on %3d wgs %.1f us, C alone on %3d wgs %.1f us, side by side %.1f us   (max |diff| %.3g)" % (
            wt, 8 * wt, tT, ncw, tC, tB, err), flush=True)
