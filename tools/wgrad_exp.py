#!/usr/bin/env python3
"""Timing of the conv weight gradient alone (config 5 or a shard of it): python tools/wgrad_exp.py [--graphs 32768] [--sweep W]
(--sweep: additionally the L2-blocked order with a HOST-built table, tests/sweep_ref.py; the tuning build's DN_WGRAD_IX=0 selects the scalar-index kernel)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(fn, reps=20):
    for _ in range(3):
        out = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=32768)
    ap.add_argument("--sweep", type=int, default=0)
    a = ap.parse_args()
    import bench
    from dummynode4graphlearning_amd import ops
    dev = torch.device("cuda:0")
    g, raw, _ = bench.build_batch(dev, 5, a.graphs, "config5")
    N, H, R = g.number_of_nodes(), 256, 16
    iset = g.row_index(g.edata["label"], R, True)
    ix = iset.parts[0][2]
    torch.manual_seed(0)
    x = torch.randn(N, H, device=dev).to(torch.bfloat16)
    gout = torch.randn(N, H, device=dev).to(torch.bfloat16)
    W = (torch.randn(R + 1, H, H, device=dev) * 0.05).to(torch.bfloat16)
    Wn = W.transpose(1, 2).contiguous()
    ybuf = torch.empty((ix.num_rows, H), dtype=x.dtype, device=dev)
    out = torch.empty_like(x)
    with torch.no_grad():
        aux = ops.message_pass(x, Wn, None, ix, "f", ybuf, out)
        aux_b = ops.message_pass(gout, W, None, ix, "b", ybuf, out)
    Rt = ix.num_all_rels
    kw = dict(idx_a=ix.row_in, idx_g=ix.row_out, A2=aux, G2=aux_b, out_dtype=torch.float32, colsum_of=2)
    t0, (gw0, cs0) = timed(lambda: ops.rows_wgrad(x, gout, ix.chunk_table, Rt, **kw))
    # fp64 reference on a sample of relations
    xa = torch.cat([x, aux]).double() if aux is not None and aux.numel() else x.double()
    ga = torch.cat([gout, aux_b]).double() if aux_b is not None and aux_b.numel() else gout.double()
    rp = ix.rel_ptr_host
    worst = 0.0
    for r in (0, R // 2, Rt - 1):
        a_, b_ = int(rp[r]), int(rp[r + 1])
        if b_ > a_:
            ref = xa[ix.row_in[a_:b_].long()].t() @ ga[ix.row_out[a_:b_].long()]
            worst = max(worst, float((gw0[r].double() - ref).abs().max() / ref.abs().max()))
    print("plain chunks (%d): %.1f us incl. reduce   max-norm error vs fp64 on 3 relations %.2e" % (ix.chunk_table[2], t0, worst), flush=True)
    if a.sweep:
        from sweep_ref import sweep_tables
        rel_ptr = np.asarray(rp[:Rt + 1], dtype=np.int64)
        row_in, row_out = ix.row_in.cpu().numpy().astype(np.int64), ix.row_out.cpu().numpy().astype(np.int64)
        key = np.where(row_out < N, row_out, row_in)
        tab, S, slot_ptr = sweep_tables(rel_ptr, key, N, a.sweep, with_slots=True)
        st = (torch.from_numpy(tab.reshape(-1, 4)).to(dev), torch.from_numpy(slot_ptr).to(dev), tab.shape[0], S, int(slot_ptr[-1]))
        t1, (gw1, cs1) = timed(lambda: ops.rows_wgrad_sweep(x, gout, st, Rt, **kw))
        print("sweep order (%d x %d tiles, %d slots): %.1f us   difference gW %.2e, colsum %.2e" % (
            tab.shape[0], S, int(slot_ptr[-1]), t1, float((gw1 - gw0).abs().max() / gw0.abs().max()),
            float((cs1[-1] - cs0[-1]).abs().max() / cs0[-1].abs().max())), flush=True)
        _, (gw2, cs2) = timed(lambda: ops.rows_wgrad_sweep(x, gout, st, Rt, **kw), reps=2)
        print("bitwise reproducible:", bool(torch.equal(gw1, gw2) and torch.equal(cs1, cs2)), flush=True)


if __name__ == "__main__":
    main()
