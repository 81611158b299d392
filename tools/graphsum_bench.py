"""Matrix-core graph-local sum (dn_graph_tile_sum_f32) vs the plain gather on bench.py's GIN-leg batch
(16384 PROTEINS-shaped dummy graphs, H = 128 fp32): per-launch times, coverage of the <= 64-row tiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dummynode4graphlearning_amd import ops, synthetic as syn, transforms as tr
dev = torch.device("cuda:0")
r2 = syn.config2(graphs=16384)
t2 = {k: torch.from_numpy(v).to(dev) for k, v in r2.items()}
a2 = tr.dummy_augment_gc(t2["node_ptr"], t2["edge_ptr"], t2["src"], t2["dst"], t2["node_label"], t2["edge_label"])
N, E, H = int(a2["node_label"].numel()), int(a2["src"].numel()), 128
ei = ops.EdgeIndex(a2["src"], a2["dst"], N, node_ptr=a2["node_ptr"])
x = torch.randn(N, H, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for mr in (64,):
    tiles, covered, rest = ops.graph_tiles(a2["node_ptr"], mr)
    print("N %d E %d: %d tiles of <= %d rows cover %d rows (%.1f %%), %d rows in larger graphs" % (N, E, tiles.shape[0], mr, covered, 100.0 * covered / N, rest.numel()))
    out = torch.empty_like(x)
    for name, ptr_, idx in (("fwd", ei.in_ptr, ei.src_by_dst), ("bwd", ei.out_ptr, ei.dst_by_src)):
        rec = ops.graph_tile_records(tiles, ptr_)
        seg = torch.repeat_interleave(torch.arange(N, device=dev, dtype=torch.int32), (ptr_[1:] - ptr_[:-1]).long())
        us0 = timed(lambda: ops.graph_tile_sum(x, idx, ptr_, rec, self_coef=1.0, out=out))
        us = timed(lambda: ops.graph_tile_sum(x, idx, ptr_, rec, self_coef=1.0, out=out, seg=seg))
        print("  (without seg: %.1f us)" % us0)
        ent = int((ptr_[tiles[:, 1].long()] - ptr_[tiles[:, 0].long()]).sum().item())
        gb = (2.0 * covered * H * 4 + 4.0 * ent + 4.0 * covered) / 1e9
        print("  tile sum %s: %.1f us for %.3f GB compulsory -> %.2f TB/s" % (name, us, gb, gb / us * 1e3))
        us2 = timed(lambda: ops.gather_segsum(x, idx, ptr_, N, self_in=x, self_coef=1.0, out=out))
        gb2 = (2.0 * N * H * 4 + 4.0 * E + 4.0 * N) / 1e9
        print("  plain gather %s (all rows, unsplit hubs): %.1f us for %.3f GB -> %.2f TB/s" % (name, us2, gb2, gb2 / us2 * 1e3))
xg = x.clone().requires_grad_(True)
go = torch.randn(N, H, device=dev)
def fb():
    xg.grad = None
    ops.neighbor_sum(xg, ei, 1.0).backward(go)
print("neighbor_sum fwd+bwd (current path): %.1f us" % timed(fb))
