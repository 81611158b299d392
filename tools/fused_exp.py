#!/usr/bin/env python3
"""Experiment: dn_rows_fused_bf16 (one launch per conv direction, tables from tests/fuse_ref.py) against the two launches it
replaces (ring transform + unit-stream closing launch): bit-identical out / aux / product rows, time.
usage (GPU box): python tools/fused_exp.py [chunk_tiles ...]   env: GRAPHS, LEAD, ROUNDS"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import fuse_ref as fuse_tables  # noqa: E402
from dummynode4graphlearning_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
graphs = int(os.environ.get("GRAPHS", "32768"))
lead = int(os.environ.get("LEAD", "2"))
rounds = int(os.environ.get("ROUNDS", "2"))
g, raw, _ = bench.build_batch(dev, 5, graphs, "config5")
N, H, R = g.number_of_nodes(), 256, 16
ix = g.row_index(g.edata["label"], R, True).parts[0][2]
P = ix.num_edge_rows
x = torch.randn(N, H, device=dev).to(torch.bfloat16)
Wrel = (torch.randn(R, H, H, device=dev) * 0.05).to(torch.bfloat16)
W = (torch.randn(H, H, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(H, device=dev).to(torch.bfloat16)
chunks = [int(a) for a in sys.argv[1:]] or [1024]
for d in ("f", "b"):
    fold = ops._row_index_fold(ix, d, "units")
    cu = ix.close_units(d)
    idx_rows = ix.row_in if d == "f" else ix.row_out
    sweep = ops._conv_tiles_for(ix, fold, H, x.dtype)
    Y0, out0, aux0 = torch.zeros(P, H, device=dev).to(torch.bfloat16), torch.empty_like(x), torch.empty((fold.n, H), dtype=x.dtype, device=dev)

    def two():
        ops.rows_transform(x, Wrel, sweep, P, idx=idx_rows, tag="conv", out=Y0, w_kn=True)
        ops.rows_close(x, W, b, Y0, cu, out=out0, w_kn=True, agg=(fold.graph_tiles[1], Wrel[fold.rel], aux0, fold.add_idx))

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

    two()
    torch.cuda.synchronize()
    print("direction %s: two launches %s us" % (d, " ".join("%.1f" % timed(two) for _ in range(rounds))), flush=True)
    for ct in chunks:
        t0 = time.time()
        tabs = fuse_tables.build(ix, d, ops, chunk_tiles=ct, lead=lead)
        Y1, out1, aux1 = torch.zeros(P, H, device=dev).to(torch.bfloat16), torch.empty_like(x), torch.empty((fold.n, H), dtype=x.dtype, device=dev)
        state = [None]

        def fused():
            _, state[0] = ops.rows_fused(x, Wrel, W, b, Y1, idx_rows, tabs, cu, (fold.graph_tiles[1], Wrel[fold.rel], aux1, fold.add_idx),
                                         out=out1, w_kn=True, state=state[0])

        fused()
        torch.cuda.synchronize()
        err = int(state[0][1][0].item())
        same = (torch.equal(out0, out1), torch.equal(aux0, aux1), torch.equal(Y0, Y1))
        print("  chunk %5d tiles, lead %d: %s (host tables %.1f s); err %d; out / aux / S bit-identical: %s; max |out diff| %.3g" % (
            ct, lead, tabs["stats"], time.time() - t0, err, same, (out0.float() - out1.float()).abs().max().item()), flush=True)
        state[0][1].zero_()
        fused()
        torch.cuda.synchronize()
        print("     one launch: gates %d, slow %d (fetched value 0: %d, complete at the first poll: %d)" % (
            tabs["num_wg"] * tabs["num_chunks"], *state[0][1][1:].tolist()), flush=True)
        print("     fused %s us" % " ".join("%.1f" % timed(fused) for _ in range(rounds)), flush=True)
        if os.environ.get("NOGATE", "0") == "1":                    # (timing only: closing units may read rows that are not there yet)
            tg = fuse_tables.build(ix, d, ops, chunk_tiles=ct, lead=lead, gates=False)
            st4 = [None]

            def nog():
                _, st4[0] = ops.rows_fused(x, Wrel, W, b, Y1, idx_rows, tg, cu, (fold.graph_tiles[1], Wrel[fold.rel], aux1, fold.add_idx),
                                           out=out1, w_kn=True, state=st4[0])
            print("     without gates: %s us (out still identical: %s)" % (" ".join("%.1f" % timed(nog) for _ in range(rounds)), torch.equal(out0, out1)), flush=True)
        if os.environ.get("CMP", "0") == "1":
            tc = fuse_tables.build(ix, d, ops, chunk_tiles=ct, lead=lead, only="C")
            a_, b_ = tc["units"].cpu(), cu.units.cpu()[:tc["units"].shape[0]]
            print("     C-only table == the closing launch's table: units %s, unit_ptr %s, sizes %s %s" % (
                torch.equal(a_, b_), torch.equal(tc["unit_ptr"].cpu(), cu.unit_ptr.cpu()), tuple(a_.shape), int(cu.unit_ptr[-1])), flush=True)
        if os.environ.get("PARTS", "0") == "1":                     # the two halves of the stream on their own (same kernel)
            for only in ("T", "C"):
                tp = fuse_tables.build(ix, d, ops, chunk_tiles=ct, lead=lead, only=only)
                st2 = [None]

                def part():
                    _, st2[0] = ops.rows_fused(x, Wrel, W, b, Y1, idx_rows, tp, cu, (fold.graph_tiles[1], Wrel[fold.rel], aux1, fold.add_idx),
                                               out=out1, w_kn=True, state=st2[0])
                print("     %s units only: %s us" % (only, " ".join("%.1f" % timed(part) for _ in range(rounds))), flush=True)
            to = dict(unit_ptr=cu.unit_ptr, units=cu.units, num_wg=cu.num_wg, num_chunks=1)
            st3 = [None]

            def orig():
                _, st3[0] = ops.rows_fused(x, Wrel, W, b, Y1, idx_rows, to, cu, (fold.graph_tiles[1], Wrel[fold.rel], aux1, fold.add_idx),
                                           out=out1, w_kn=True, state=st3[0])
            print("     the closing launch's own tables through the fused kernel: %s us" % " ".join("%.1f" % timed(orig) for _ in range(rounds)), flush=True)
            os.environ["DN_CLOSE8"] = "1"
            print("     ... through the eight-wave kernel without the transform code: %.1f us" % timed(
                lambda: ops.rows_close(x, W, b, Y0, cu, out=out0, w_kn=True, agg=(fold.graph_tiles[1], Wrel[fold.rel], aux0, fold.add_idx))), flush=True)
            os.environ["DN_CLOSE8"] = "0"
            print("     ring transform alone %.1f us, 12-wave closing launch alone %.1f us" % (
                timed(lambda: ops.rows_transform(x, Wrel, sweep, P, idx=idx_rows, tag="conv", out=Y0, w_kn=True)),
                timed(lambda: ops.rows_close(x, W, b, Y0, cu, out=out0, w_kn=True, agg=(fold.graph_tiles[1], Wrel[fold.rel], aux0, fold.add_idx)))), flush=True)
        assert int(state[0][1][0].item()) == 0
