"""Per-launch times of the tile-path neighbour sum on bench.py's GIN-leg batch (tile kernel, short-row list, hub list)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dummynode4graphlearning_amd import ops, synthetic as syn, transforms as tr
dev = torch.device("cuda:0")
r2 = syn.config2(graphs=16384)
t2 = {k: torch.from_numpy(v).to(dev) for k, v in r2.items()}
a2 = tr.dummy_augment_gc(t2["node_ptr"], t2["edge_ptr"], t2["src"], t2["dst"], t2["node_label"], t2["edge_label"])
N, H = int(a2["node_label"].numel()), 128
ei = ops.EdgeIndex(a2["src"], a2["dst"], N, node_ptr=a2["node_ptr"])
plan = ei.tile_plan()
x = torch.randn(N, H, device=dev); out = torch.empty_like(x)
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for d in "fb":
    rec, short, long_, ptr_, idx = plan.dirs[d]
    rl = long_[:, 0].long(); rs = short[:, 0].long()
    print("%s: tiles %d (rows %d) | short rows %d (entries %d) | hub rows %d (entries %d)" % (d, rec.shape[0], plan.covered, short.shape[0],
          int((ptr_[rs + 1] - ptr_[rs]).sum()), long_.shape[0], int((ptr_[rl + 1] - ptr_[rl]).sum())))
    print("   tile kernel %.1f us | short list %.1f us | hub list %.1f us" % (
        timed(lambda: ops.graph_tile_sum(x, idx, ptr_, rec, 1.0, out, bad=plan.bad)),
        timed(lambda: ops.gather_rows_sum(x, idx, ptr_, short, False, 1.0, out)),
        timed(lambda: ops.gather_rows_sum(x, idx, ptr_, long_, True, 1.0, out))))
