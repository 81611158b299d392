#!/usr/bin/env python3
"""Per-role statistics of one dn_rows_pipe_bf16 launch on the benchmark batch (a tuning aid).
usage: python tools/pipe_stats.py [graphs]   (env: DN_PIPE_* as in ops.py)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dummynode4graphlearning_amd import BatchedGraph, ops, synthetic, transforms  # noqa: E402

dev = torch.device("cuda:0")
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
raw = synthetic.config5(5, graphs)
t = {k: torch.from_numpy(v).to(dev) for k, v in raw.items() if isinstance(v, np.ndarray)}
aug = transforms.dummy_augment_si(t["node_ptr"], t["edge_ptr"], t["src"], t["dst"], t["node_id"], t["node_label"], t["edge_id"],
                                  t["edge_label"], raw["max_nv"], raw["max_nvl"], raw["max_ne"], raw["max_nel"])
N, H, R = int(aug["node_label"].numel()), 256, 16
bnn = (aug["node_ptr"][1:] - aug["node_ptr"][:-1]).long()
bne = (aug["edge_ptr"][1:] - aug["edge_ptr"][:-1]).long()
g = BatchedGraph(aug["src"], aug["dst"], N, bnn, bne)
iset = g.row_index(aug["edge_label"].long(), R, True)
ix = iset.parts[0][2]
pipe = ix.pipe
print("batches %d groups %d slot_rows %d tiles %d ring MB %.1f" % (pipe.num_batches, pipe.num_groups, pipe.slot_rows, pipe.num_tiles,
                                                                    pipe.ring_rows * H * 2 / 1e6))
x = torch.randn(N, H, device=dev).to(torch.bfloat16)
W = (torch.randn(R + 1, H, H, device=dev) / 16).to(torch.bfloat16)
out = torch.empty_like(x)
ybuf = iset.ybuf(H, x.dtype, dev)
for it in range(3):
    pipe.stats = torch.zeros((8 * pipe.roles_per_group, 8), dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.message_pass(x, W, None, ix, "f", ybuf, out)
    e1.record()
    torch.cuda.synchronize()
print("launch (incl. pre-aggregation) %.3f ms, abort %d" % (e0.elapsed_time(e1), pipe.aborted()))
st = pipe.stats.cpu().numpy()
for kind, name in ((0, "T"), (1, "S")):
    m = (st[:, 3] == kind) & (st[:, 2] > 0)
    tot, wait, tiles = st[m, 0] / 100.0, st[m, 1] / 100.0, st[m, 2]
    print("%s roles %d: total us mean %.0f max %.0f | waiting us mean %.0f max %.0f | tiles mean %.0f | busy us/tile mean %.2f"
          % (name, m.sum(), tot.mean(), tot.max(), wait.mean(), wait.max(), tiles.mean(), ((tot - wait) / tiles).mean()))
    print("   sections us/tile:", np.round((st[m, 4:8] / 100.0 / tiles[:, None]).mean(0), 2),
          "(T: load+mfma | stage next | wait+store | drain+barrier;  C: mfma+stage | wait | epilogue | prefetch+barrier)")
