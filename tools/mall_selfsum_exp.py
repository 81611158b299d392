#!/usr/bin/env python3
"""Experiment: what would the closing launch (dn_rows_selfsum_bf16) gain if its Y rows came from the Infinity Cache?  The slot
table is folded into a window of W rows (W x 512 B), everything else unchanged.  usage (GPU box): python tools/mall_selfsum_exp.py"""
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
ix = g.row_index(g.edata["label"], R, True).parts[0][2]
slots, lists = ix.slots("f")
P = ix.num_edge_rows
x = torch.randn(N, H, device=dev).to(torch.bfloat16)
W = (torch.randn(H, H, device=dev) * 0.05).to(torch.bfloat16)
Y = torch.randn(P, H, device=dev).to(torch.bfloat16)
out = torch.empty_like(x)


def timed(sl, reps=20):
    for _ in range(3):
        ops.rows_selfsum(x, W, None, Y, None, sl, out=out, lists=lists)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.rows_selfsum(x, W, None, Y, None, sl, out=out, lists=lists)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("slots used per node: %.2f" % float((slots >= 0).float().sum() / N))
print("full Y (%.0f MB): %.1f us" % (P * H * 2 / 1e6, timed(slots)))
for wrows in (200000, 100000, 20000, 2000):
    sl = torch.where(slots >= 0, slots % wrows, slots).contiguous()
    print("Y window %7d rows (%.0f MB): %.1f us" % (wrows, wrows * H * 2 / 1e6, timed(sl)))
sl = torch.full_like(slots, -1)
print("no slot rows at all: %.1f us" % timed(sl))
