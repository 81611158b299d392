#!/usr/bin/env python3
"""What the conv weight gradient (rows_wgrad_ix_kernel) costs when its gathers come from the caches: the same chunk table,
   the row indices folded into the first M rows of each operand (M rows x 512 B: 1 MB fits an XCD's L2, 16-64 MB the Infinity Cache).
   The distance between the real indices' time and the folded ones' bounds what any re-ordering of the rows could gain."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dummynode4graphlearning_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
g, raw, _ = bench.build_batch(dev, 5, graphs, "config5")
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


def run(ia, ig, label):
    kw = dict(idx_a=ia, idx_g=ig, A2=aux, G2=aux_b, out_dtype=torch.float32, colsum_of=2)
    for _ in range(5):
        ops.rows_wgrad(x, gout, ix.chunk_table, ix.num_all_rels, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.rows_wgrad(x, gout, ix.chunk_table, ix.num_all_rels, **kw)
    e1.record()
    torch.cuda.synchronize()
    print("%-44s kernel + reduce %.1f us" % (label, e0.elapsed_time(e1) / 20 * 1e3), flush=True)


print("rows %d = %d tiles of 32" % (ix.num_rows, ix.num_rows // 32))
run(ix.row_in, ix.row_out, "real indices")
for M in (1 << 19, 1 << 17, 1 << 15, 1 << 13, 1 << 11):
    run((ix.row_in % M).contiguous(), (ix.row_out % M).contiguous(), "indices mod %d (%.0f MB per operand)" % (M, M * 512 / 2**20))
perm = torch.randperm(N, device=dev, dtype=torch.int32)
run(torch.where(ix.row_in < N, perm[ix.row_in.clamp(max=N - 1).long()], ix.row_in).contiguous(),
    torch.where(ix.row_out < N, perm[ix.row_out.clamp(max=N - 1).long()], ix.row_out).contiguous(), "node indices randomly permuted")
ar = torch.arange(ix.num_rows, device=dev, dtype=torch.int32)
run((ar % N).contiguous(), (ar % N).contiguous(), "consecutive rows (a dense product)")
