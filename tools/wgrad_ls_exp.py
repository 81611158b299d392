#!/usr/bin/env python3
"""rows_wgrad_ls_kernel (loaders and MFMA waves apart) against the 8-wave ring kernels, H = 256: time and a checksum of every output.
   Tuning build: DN_HIP_LIB=tools/_lib_tuning.so DN_WGRAD_LS=0|7 python tools/wgrad_ls_exp.py [graphs] -- the two runs' checksums
   must be equal (bit-identical partial sums)."""
import hashlib, os, sys
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
bits = (torch.rand(N, H // 8, device=dev) * 255).to(torch.uint8)
_, dense = ops._dense_table(N, dev)


def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        if t is not None:
            h.update(t.detach().float().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def run(label, fn, reps=20):
    res = fn()
    for _ in range(4):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    res = res if isinstance(res, tuple) else (res,)
    print("%-46s %8.1f us  sha %s" % (label, e0.elapsed_time(e1) / reps * 1e3, digest(res)), flush=True)


print("DN_WGRAD_LS=%s  N %d rows %d" % (os.environ.get("DN_WGRAD_LS", "(default)"), N, ix.num_rows))
for cso in (2, 1, 0):
    run("conv wgrad (gathered), colsum_of=%d" % cso, lambda: ops.rows_wgrad(x, gout, ix.chunk_table, ix.num_all_rels, idx_a=ix.row_in, idx_g=ix.row_out,
                                                                            A2=aux, G2=aux_b, out_dtype=torch.float32, colsum_of=cso))
run("conv wgrad (gathered), colsum_of=2, loop relation only", lambda: ops.rows_wgrad(x, gout, ix.chunk_table, ix.num_all_rels, idx_a=ix.row_in,
    idx_g=ix.row_out, A2=aux, G2=aux_b, out_dtype=torch.float32, colsum_of=2, colsum_rel=ix.num_all_rels - 1))
full = ops.rows_wgrad(x, gout, ix.chunk_table, ix.num_all_rels, idx_a=ix.row_in, idx_g=ix.row_out, A2=aux, G2=aux_b, out_dtype=torch.float32, colsum_of=2)
one = ops.rows_wgrad(x, gout, ix.chunk_table, ix.num_all_rels, idx_a=ix.row_in, idx_g=ix.row_out, A2=aux, G2=aux_b, out_dtype=torch.float32, colsum_of=2,
                     colsum_rel=ix.num_all_rels - 1)
assert torch.equal(full[0], one[0]) and torch.equal(full[1][-1], one[1][-1]) and not one[1][:-1].any(), "colsum_rel"
for cso in (1, 2, 0):
    run("dense wgrad, colsum_of=%d" % cso, lambda: ops.rows_wgrad(gout, x, dense, 1, out_dtype=torch.bfloat16, colsum_of=cso))
    run("dense wgrad + mask bits, colsum_of=%d" % cso, lambda: ops.rows_wgrad(gout, x, dense, 1, out_dtype=torch.bfloat16, colsum_of=cso,
                                                                              mask_a_bits=bits, slope=0.0))
run("dense wgrad + mask bits, slope 0.18", lambda: ops.rows_wgrad(gout, x, dense, 1, out_dtype=torch.float32, colsum_of=1, mask_a_bits=bits, slope=0.18))
# ragged ends: a row count that is no multiple of 32
M = N - 13
_, dense2 = ops._dense_table(M, dev)
run("dense wgrad, %d rows" % M, lambda: ops.rows_wgrad(gout[:M], x[:M], dense2, 1, out_dtype=torch.float32, colsum_of=1))
run("dense wgrad + mask bits, %d rows" % M, lambda: ops.rows_wgrad(gout[:M], x[:M], dense2, 1, out_dtype=torch.float32, colsum_of=1, mask_a_bits=bits[:M]))
