#!/usr/bin/env python3
"""Cycle breakdown of the conv weight gradient (rows_wgrad_ix_kernel) from a diagnostic build:
   DN_BUILD_EXTRA=-DDN_WG_STATS python -m dummynode4graphlearning_amd.csrc.build --force ; python tools/wgrad_stats.py [--graphs 32768]"""
import argparse, ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dummynode4graphlearning_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=32768)
a = ap.parse_args()
dev = torch.device("cuda:0")
g, raw, _ = bench.build_batch(dev, 5, a.graphs, "config5")
N, H, R = g.number_of_nodes(), 256, 16
ix = g.row_index(g.edata["label"], R, True).parts[0][2]
torch.manual_seed(0)
x = torch.randn(N, H, device=dev).to(torch.bfloat16)
gout = torch.randn(N, H, device=dev).to(torch.bfloat16)
W = (torch.randn(R + 1, H, H, device=dev) * 0.05).to(torch.bfloat16)
ybuf = torch.empty((ix.num_rows, H), dtype=x.dtype, device=dev)
out = torch.empty_like(x)
with torch.no_grad():
    aux = ops.message_pass(x, ops.PassWeights(W[:-1], W[-1], kn=True), None, ix, "f", ybuf, out)
    aux_b = ops.message_pass(gout, ops.PassWeights(W[:-1], W[-1]), None, ix, "b", ybuf, out)
kw = dict(idx_a=ix.row_in, idx_g=ix.row_out, A2=aux, G2=aux_b, out_dtype=torch.float32, colsum_of=2)
for _ in range(3):
    ops.rows_wgrad(x, gout, ix.chunk_table, ix.num_all_rels, **kw)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.rows_wgrad(x, gout, ix.chunk_table, ix.num_all_rels, **kw)
e1.record()
torch.cuda.synchronize()
print("rows_wgrad (kernel + reduce): %.1f us; rows %d = %d tiles" % (e0.elapsed_time(e1) / 10 * 1e3, ix.num_rows, ix.num_rows // 32))
L = _lib.lib()
buf = (ctypes.c_ulonglong * (256 * 2 * 5))()
fn = L.dn_debug_wgrad_stats
fn.restype = ctypes.c_int
assert fn(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 2, 5).astype(np.float64)
tiles = ix.num_rows / 32 / 256
for w, name in ((0, "wave 0 (DMAs first)"), (1, "wave 4 (MFMAs first)")):
    m = st[:, w].mean(0)
    print("%s: loop %.0f cycles = %.0f per tile; wait+barrier %.0f, DMA issue %.0f, fragments+MFMA+colsum %.0f per tile; wall %.1f us -> %.2f GHz"
          % (name, m[0], m[0] / tiles, m[1] / tiles, m[2] / tiles, m[3] / tiles, m[4] / 100.0, m[0] / (m[4] * 10.0) / 1e0 / 1e3 * 1e3 / 1e3))
