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


TABLE = [ix.chunk_table]


def weighted_table(w_last, workgroups=256):
    """chunk table with the LAST relation's chunks holding w_last x the rows of the others' (host-built, as ops.wide_layer_chunks)"""
    ptr_ = [int(v) for v in ix.rel_ptr_host][:ix.num_all_rels + 1]
    sizes = [b - a for a, b in zip(ptr_[:-1], ptr_[1:])]
    last = lambda c: max(64, int(c * w_last) // 32 * 32)    # noqa: E731
    count = lambda c: sum(-(-m // c) for m in sizes[:-1] if m > 0) + (-(-sizes[-1] // last(c)))   # noqa: E731
    c = 256
    while count(c) > workgroups:
        c += 64
    chunks, cptr = [], [0]
    for r, m in enumerate(sizes):
        step = c if r < len(sizes) - 1 else last(c)
        a, b = ptr_[r], ptr_[r + 1]
        while a < b:
            chunks.append((r, a, min(a + step, b), 0))
            a += step
        cptr.append(len(chunks))
    print("   (table: %d chunks, %d rows a chunk, %d for the last relation)" % (len(chunks), c, last(c)))
    return (torch.tensor(chunks, dtype=torch.int32).to(dev), torch.tensor(cptr, dtype=torch.int32).to(dev), len(chunks))


def run(ia, ig, label, cso=2, rel_only=False):
    ix_chunk_table = TABLE[0]
    kw = dict(idx_a=ia, idx_g=ig, A2=aux, G2=aux_b, out_dtype=torch.float32, colsum_of=cso)
    if rel_only:
        kw["colsum_rel"] = ix.num_all_rels - 1
    for _ in range(3):
        ops.rows_wgrad(x, gout, ix_chunk_table, ix.num_all_rels, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.rows_wgrad(x, gout, ix_chunk_table, ix.num_all_rels, **kw)
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
    # spread over the workgroups (wall of the compute loop), by the relation of a workgroup's chunk
    ch = ix_chunk_table[0][:256].cpu().numpy()
    wall = st[:ch.shape[0], 0, 4] / 100.0
    nrow = ch[:, 2] - ch[:, 1]
    live = nrow > 0
    print("   workgroups: wall us min %.1f median %.1f max %.1f; rows per chunk %d .. %d" %
          (wall[live].min(), np.median(wall[live]), wall[live].max(), nrow[live].min(), nrow[live].max()))
    rels = ch[:, 0]
    for r in sorted(set(int(v) for v in rels[live])):
        m = live & (rels == r)
        print("     rel %2d: %2d chunks, rows %6d .. %6d, wall median %.1f max %.1f us, us per 1000 rows %.2f" %
              (r, int(m.sum()), nrow[m].min(), nrow[m].max(), np.median(wall[m]), wall[m].max(), float(np.median(wall[m] / nrow[m] * 1000))))


run(ix.row_in, ix.row_out, "real indices, column sums of the self-loop relation only (as the layer calls it)", 2, True)
for w in (0.85, 0.8, 0.75, 0.7):
    TABLE[0] = weighted_table(w)
    run(ix.row_in, ix.row_out, "self-loop chunks x %.2f rows, column sums of the self-loop relation only" % w, 2, True)
TABLE[0] = ix.chunk_table
run(ix.row_in, ix.row_out, "real indices")
run(ix.row_in, ix.row_out, "real indices, no column sums", 0)
if "--spread" in sys.argv:
    sys.exit(0)
M = 2048
run((ix.row_in % M).contiguous(), (ix.row_out % M).contiguous(), "indices mod 2048 (1 MB per operand)")
run((ix.row_in % M).contiguous(), (ix.row_out % M).contiguous(), "indices mod 2048, no column sums", 0)
