#!/usr/bin/env python3
"""Per-role cycle breakdown of rows_wgrad_ls_kernel from a diagnostic build (-DDN_WG_STATS):
   DN_BUILD_EXTRA=-DDN_WG_STATS python -m dummynode4graphlearning_amd.csrc.build --force ; python tools/wgrad_ls_stats.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dummynode4graphlearning_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
g, raw, _ = bench.build_batch(dev, 5, 32768, "config5")
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
L = _lib.lib()
fn = L.dn_debug_wgrad_stats
fn.restype = ctypes.c_int
tiles = ix.num_rows / 32 / 256


def run(ia, ig, label, cso=2):
    kw = dict(idx_a=ia, idx_g=ig, A2=aux, G2=aux_b, out_dtype=torch.float32, colsum_of=cso)
    for _ in range(3):
        ops.rows_wgrad(x, gout, ix.chunk_table, ix.num_all_rels, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.rows_wgrad(x, gout, ix.chunk_table, ix.num_all_rels, **kw)
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (256 * 2 * 5))()
    assert fn(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 2, 5).astype(np.float64)
    c, l = st[:, 0].mean(0), st[:, 1].mean(0)
    print("%s: %.1f us (kernel + reduce), %.0f tiles per workgroup" % (label, e0.elapsed_time(e1) / 10 * 1e3, tiles))
    print("   compute wave 0: loop %.0f cycles per tile, of which at the barrier %.0f; wall %.1f us -> %.2f GHz" %
          (c[0] / tiles, c[1] / tiles, c[4] / 100.0, c[0] / (c[4] * 10.0) / 1e3 * 1e3 / 1e3 if c[4] else 0))
    print("   loader 0: loop %.0f per tile = vmcnt wait %.0f + barrier %.0f + issue %.0f + column sums %.0f" %
          (l[0] / tiles, l[1] / tiles, l[2] / tiles, l[3] / tiles, l[4] / tiles), flush=True)


run(ix.row_in, ix.row_out, "real indices")
run(ix.row_in, ix.row_out, "real indices, no column sums", 0)
M = 2048
run((ix.row_in % M).contiguous(), (ix.row_out % M).contiguous(), "indices mod 2048 (1 MB per operand)")
run((ix.row_in % M).contiguous(), (ix.row_out % M).contiguous(), "indices mod 2048, no column sums", 0)
