#!/usr/bin/env python3
"""Timing of the conv weight gradient alone (config 5 or a shard of it): python tools/wgrad_exp.py [--graphs 32768]
(the tuning build's DN_WGRAD_IX=0 selects the scalar-index kernel)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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
        aux = ops.message_pass(x, ops.PassWeights(Wn[:-1], Wn[-1]), None, ix, "f", ybuf, out)
        aux_b = ops.message_pass(gout, ops.PassWeights(W[:-1], W[-1]), None, ix, "b", ybuf, out)
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


if __name__ == "__main__":
    main()
