#!/usr/bin/env python3
"""Experiment (tuning build): the closing launch on eight self-loading waves (dn_fuse.hip, DN_CLOSE8=1) against the twelve-wave
kernel (8 compute + 4 loader waves): bit-identical output, time.  usage (GPU box): python tools/close8_exp.py [rounds]"""
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
W = (torch.randn(H, H, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(H, device=dev).to(torch.bfloat16)
Y = torch.randn(P, H, device=dev).to(torch.bfloat16)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for d in ("f", "b"):
    cu = ix.close_units(d)
    fold = ops._row_index_fold(ix, d, "units")
    assert fold is not None and cu.agg
    outs, auxs = {}, {}

    def run(mode, out, aux):
        os.environ["DN_CLOSE8"] = str(mode)
        ops.rows_close(x, W, b, Y, cu, out=out, w_kn=True, agg=(fold.graph_tiles[1], W, aux, fold.add_idx))

    def timed(mode, reps=20):
        out, aux = torch.empty_like(x), torch.empty((fold.n, H), dtype=x.dtype, device=dev)
        for _ in range(3):
            run(mode, out, aux)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run(mode, out, aux)
        e1.record()
        torch.cuda.synchronize()
        outs[mode], auxs[mode] = out, aux
        return e0.elapsed_time(e1) / reps * 1e3

    res = {}
    for _ in range(rounds):
        for mode in (0, 1, 0, 1):
            res.setdefault(mode, []).append(timed(mode))
    same = torch.equal(outs[0], outs[1]) and torch.equal(auxs[0], auxs[1])
    diff = (outs[0].float() - outs[1].float()).abs().max().item()
    print("direction %s: bit-identical %s (max |diff| %.3g)" % (d, same, diff))
    for mode, v in res.items():
        print("  DN_CLOSE8=%d: %s us (min %.1f)" % (mode, " ".join("%.1f" % t for t in v), min(v)), flush=True)
