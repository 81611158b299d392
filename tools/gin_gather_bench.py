"""GIN conv gather + segment-sum forward+backward on 16384 PROTEINS-shaped dummy graphs (H = 128, fp32): time under HIP-graph
replay and the compulsory-bytes roofline fraction (bench.py prints the same as its secondary line)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dummynode4graphlearning_amd import ops, synthetic as syn, transforms as tr
dev = torch.device("cuda:0")
r2 = syn.config2(graphs=16384)
t2 = {k: torch.from_numpy(v).to(dev) for k, v in r2.items()}
a2 = tr.dummy_augment_gc(t2["node_ptr"], t2["edge_ptr"], t2["src"], t2["dst"], t2["node_label"], t2["edge_label"])
N2, E2, H2 = int(a2["node_label"].numel()), int(a2["src"].numel()), 128
ei = ops.EdgeIndex(a2["src"], a2["dst"], N2, node_ptr=a2["node_ptr"])
x2 = torch.randn(N2, H2, device=dev, requires_grad=True); go2 = torch.randn(N2, H2, device=dev)
def fb():
    x2.grad = None
    ops.neighbor_sum(x2, ei, 1.0).backward(go2)
for _ in range(3): fb()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): fb()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
comp = 2.0 * (2 * N2 * H2 * 4 + 8.0 * E2)
plan = ei.tile_plan() if ops.TILE_SUM_ENABLED else None
path = ("matrix-core tiles for %d of %d rows + row lists" % (plan.covered, ei.num_nodes)) if plan is not None else \
       ("plain gather, hubs split: %s" % (ei.fwd.hub_ids is not None))
print("GIN gather fwd+bwd %.3f ms  compulsory %.2f GB -> %.0f GB/s = %.3f of 8 TB/s; %s" % (ms, comp / 1e9, comp / ms / 1e6, comp / ms / 1e6 / 8000, path))
