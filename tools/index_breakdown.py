"""Where a fresh batch's index build goes (wall time per stage, synchronised between stages): python tools/index_breakdown.py [config3|config5]"""
import os, sys, time, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dummynode4graphlearning_amd import ops
w = sys.argv[1] if len(sys.argv) > 1 else "config3"
dev = torch.device("cuda:0")
g, raw, aug_ms = bench.build_batch(dev, {"config5": 5, "config3": 3}[w], {"config5": 32768, "config3": 512}[w], w)
etype = g.edata["label"]; R = {"config5": 16, "config3": 8}[w]
def sync(): torch.cuda.synchronize()
for it in range(5):
    g._cache.clear(); sync(); t0 = time.perf_counter()
    npt, ept = g.node_ptr(), g.edge_ptr(); sync(); t1 = time.perf_counter()
    ix = ops.RowIndex(g._src, g._dst, etype, g.number_of_nodes(), R, self_loop=True, node_ptr=npt, edge_ptr=ept); sync(); t2 = time.perf_counter()
    ops._closing_tables(ix); sync(); t3 = time.perf_counter()
    print("%s build %d: ptrs %.3f  RowIndex(%s) %.3f  closing tables %.3f  total %.3f ms" % (w, it, (t1-t0)*1e3, ix.built_by, (t2-t1)*1e3, (t3-t2)*1e3, (t3-t0)*1e3))
pr = cProfile.Profile(); pr.enable()
for it in range(20):
    ix = ops.RowIndex(g._src, g._dst, etype, g.number_of_nodes(), R, self_loop=True, node_ptr=npt, edge_ptr=ept)
    ops._closing_tables(ix)
sync(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
