#!/usr/bin/env python3
"""Experiment: L2-blocked ("sweep") tile order for dn_rows_transform_bf16 on config 5.

Every workgroup keeps (mostly) ONE relation's weights, the workgroups of an XCD (blocks b, b+8, ...) walk the SAME eighth of
the batch in lock step, so a source row fetched for one relation is re-read from that XCD's L2 by the others.  The table is
built on the host here (numpy); timings of the transform launch alone (HIP events, back-to-back launches) and of the whole
conv leg.  usage (GPU box): python tools/sweep_exp.py [--graphs 32768] [--wgs 64] [--xcd-major]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sweep_ref import sweep_tables  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=32768)
    ap.add_argument("--wgs", type=int, nargs="*", default=[64])
    ap.add_argument("--xcd-major", action="store_true", help="control: wrong placement (workgroup b -> group b // W)")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="", choices=["", "baseline", "sweep"])
    ap.add_argument("--slow", type=int, default=0, help="tuning build with DN_RING_STATS: list the N slowest workgroups of a sweep launch")
    ap.add_argument("--ab", type=int, default=0, help="conv leg A/B only: plain and sweep tile order alternated this many times")
    a = ap.parse_args()
    global SLOW
    SLOW = a.slow
    import bench
    from dummynode4graphlearning_amd import ops
    dev = torch.device("cuda:0")
    g, raw, _ = bench.build_batch(dev, 5, a.graphs, os.environ.get("WORKLOAD", "config5"))    # WORKLOAD=proteins --graphs 16384
    N, E, H, R = g.number_of_nodes(), g.number_of_edges(), 256, 16
    etype = g.edata["label"]
    iset = g.row_index(etype, R, True)
    ix = iset.parts[0][2]
    ix.slots("f"), ix.slots("b")
    torch.manual_seed(0)
    x = torch.randn(N, H, device=dev).to(torch.bfloat16)
    W = (torch.randn(R + 1, H, H, device=dev) * 0.05).to(torch.bfloat16)
    P = ix.num_edge_rows
    rel_ptr = np.asarray(ix.rel_ptr_host[:R + 1], dtype=np.int64)
    row_in, row_out = ix.row_in.cpu().numpy().astype(np.int64), ix.row_out.cpu().numpy().astype(np.int64)
    key = np.where(row_out[:P] < N, row_out[:P], row_in[:P])

    def time_tf(idx_rows, table, reps=a.reps):
        Y = torch.zeros((P, H), dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            ops.rows_transform(x, W, table, P, idx=idx_rows, out=Y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            ops.rows_transform(x, W, table, P, idx=idx_rows, out=Y)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3, Y

    gout = torch.randn(N, H, device=dev).to(torch.bfloat16)
    bias = torch.zeros(H, device=dev, dtype=torch.bfloat16)
    alg = 2.0 * (E * H * 2 + N * H * 2 + 8.0 * E)
    if a.ab:
        plain = {d: ops.build_row_tables(ix.rel_ptr_dev, ix.num_rels, ix.num_edge_rows, 32, skip_mask=1 << ops._row_index_fold(ix, d).rel) for d in ("f", "b")}
        swept = {}
        for d in ("f", "b"):
            fold = ops._row_index_fold(ix, d)
            tab, S = sweep_tables(rel_ptr, key, N, a.wgs[0], skip_mask=1 << fold.rel)
            swept[d] = (torch.from_numpy(tab.reshape(-1, 4)).to(dev), int(tab.shape[0] * tab.shape[1]))
        res = {"plain": [], "sweep": []}
        for it in range(a.ab):
            for name, tabs in (("plain", plain), ("sweep", swept)) if it % 2 == 0 else (("sweep", swept), ("plain", plain)):
                for d in ("f", "b"):
                    ops._row_index_fold(ix, d).main_tiles = tabs[d]
                res[name].append(conv_leg(ops, ix, x, gout, W, bias, dev, reps=30))
        for name in ("plain", "sweep"):
            v = res[name]
            print("conv leg %s tile order: %s  mean %.4f ms (frac %.4f)" % (name, " ".join("%.4f" % t for t in v), sum(v) / len(v),
                                                                            alg / (sum(v) / len(v) * 1e-3) / 8e12), flush=True)
        return
    t_conv = conv_leg(ops, ix, x, gout, W, bias, dev)
    print("conv leg, plain tile order: %.4f ms  (roofline frac %.4f)" % (t_conv, alg / (t_conv * 1e-3) / 8e12), flush=True)
    for direction in ("f", "b"):
        fold = ops._row_index_fold(ix, direction)
        assert fold is not None
        idx_rows = ix.row_in if direction == "f" else ix.row_out
        skip = 1 << fold.rel if fold is not None else 0
        base = (fold.main_tiles or ops.build_row_tables(ix.rel_ptr_dev, ix.num_rels, ix.num_edge_rows, 32, skip_mask=skip)) if fold is not None else ix.edge_tile_table
        Y0 = None
        if a.only != "sweep":
            t0, Y0 = time_tf(idx_rows, base)
            print("dir %s baseline            %7.1f us   (tiles %d)  checksum %d"
                  % (direction, t0, base[1], int(Y0.view(torch.int16).long().sum())), flush=True)
            ring_stats("baseline")
        if a.only == "baseline":
            continue
        for Wg in a.wgs:
            for xm in ((False, True) if a.xcd_major else (False,)):
                tab, S = sweep_tables(rel_ptr, key, N, Wg, skip_mask=skip, xcd_major=xm)
                tt = (torch.from_numpy(tab.reshape(-1, 4)).to(dev), int(tab.shape[0] * tab.shape[1]))
                t1, Y1 = time_tf(idx_rows, tt)
                same = bool(torch.equal(Y0.view(torch.int16), Y1.view(torch.int16))) if Y0 is not None else None
                print("dir %s sweep W=%3d %s %7.1f us   steps %d  identical %s"
                      % (direction, Wg, "xcd-major(control)" if xm else "round-robin", t1, S, same), flush=True)
                ring_stats("sweep", tab if a.slow else None)
                if not xm:
                    fold.main_tiles = tt                  # (kept for the conv leg below)
    t_conv = conv_leg(ops, ix, x, gout, W, bias, dev)
    print("conv leg, sweep tile order: %.4f ms  (roofline frac %.4f)" % (t_conv, alg / (t_conv * 1e-3) / 8e12), flush=True)


def conv_leg(ops, ix, x, gout, W_all, bias, dev, reps=20):
    """both directions of ops.message_pass replayed from a HIP graph (what bench.py's roofline leg times)."""
    N, H = x.shape
    Wn = W_all.transpose(1, 2).contiguous()
    ybuf = torch.empty((ix.num_rows, H), dtype=x.dtype, device=dev)
    out = torch.empty_like(x)

    def leg():
        with torch.no_grad():
            ops.message_pass(x, ops.PassWeights(Wn[:-1], Wn[-1]), bias, ix, "f", ybuf, out)
            ops.message_pass(gout, ops.PassWeights(W_all[:-1], W_all[-1]), None, ix, "b", ybuf, out)
    leg()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        leg()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


SLOW = 0


def ring_stats(tag, table=None):
    """tuning build: cycle counters of the last ring launch (median over workgroups)."""
    import ctypes
    from dummynode4graphlearning_amd import _lib
    h = ctypes.CDLL(_lib.LIB_PATH)
    if not hasattr(h, "dn_debug_ring_stats"):
        return
    buf = (ctypes.c_ulonglong * (256 * 10))()
    torch.cuda.synchronize()
    h.dn_debug_ring_stats(buf)
    a = np.array(buf, dtype=np.float64).reshape(256, 10)
    m = np.median(a, 0)[:8]
    rt = a[:, 8] / 100.0                                    # us (100 MHz counter)
    start = (a[:, 9] - a[:, 9].min()) / 100.0
    print("   compute loop wall us: min %.1f median %.1f max %.1f; start skew max %.1f us; end max %.1f us; clock %.2f GHz; per XCD median us %s"
          % (rt.min(), np.median(rt), rt.max(), start.max(), (start + rt).max(), np.median(a[:, 0] / (rt * 1e3)),
             " ".join("%.0f" % np.median(rt[x::8]) for x in range(8))), flush=True)
    print("   %s cycles/WG (median): compute loop %.0f = barrier wait %.0f + reads/mfma %.0f + epilogue %.0f | loader loop %.0f = "
          "vm wait %.0f + barrier wait %.0f + body %.0f" % ((tag,) + tuple(m)), flush=True)
    if table is not None and SLOW:
        # the slowest workgroups: (block, XCD group x = b % 8, j = b // 8), loop us, tiles, relations in processing order with tile counts
        order = np.argsort(-rt)
        def describe(b):
            rows = table[b]
            live = rows[rows[:, 2] > rows[:, 1]]
            rels, cnt = [], []
            for r in live[:, 0]:
                if rels and rels[-1] == r:
                    cnt[-1] += 1
                else:
                    rels.append(int(r)); cnt.append(1)
            return "b %3d x %d j %2d  %.1f us  tiles %3d  cycles: bar %6.0f mfma %6.0f epi %6.0f  rels %s" % (
                b, b % 8, b // 8, rt[b], live.shape[0], a[b, 1], a[b, 2], a[b, 3], " ".join("%d:%d" % rc for rc in zip(rels, cnt)))
        for b in order[:SLOW]:
            print("   slow  " + describe(int(b)), flush=True)
        for b in order[-4:]:
            print("   fast  " + describe(int(b)), flush=True)
        mid = order[len(order) // 2]
        print("   mid   " + describe(int(mid)), flush=True)


if __name__ == "__main__":
    main()
