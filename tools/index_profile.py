"""Build the config-5 row index (RowIndex + slot tables + fold tables) a few times: wall time per build, and -- under
rocprofv3 --kernel-trace --stats -- the kernels it consists of."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
g, raw, aug_ms = bench.build_batch(dev, 5, 32768, "config5")
etype = g.edata["label"]
for it in range(6):
    g._cache.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix = g.row_index(etype, 16, True)
    t1 = time.perf_counter()
    for _, _, part in ix.parts:
        part.slots("f"), part.slots("b")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("build %d: row index %.2f ms (launch side), + slots/fold %.2f ms, total %.2f ms" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3))
