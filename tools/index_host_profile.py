#!/usr/bin/env python3
"""Where the HOST time of a per-batch index build goes (cProfile over repeated builds of one batch):
   python tools/index_host_profile.py [--graphs 4096] [--workload config5] [--reps 50]"""
import argparse, cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dummynode4graphlearning_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=4096)
ap.add_argument("--workload", default="config5")
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
seed = {"config5": 5, "config3": 3}[a.workload]
H, R, dtype = (256, 16, torch.bfloat16) if a.workload == "config5" else (64, 8, torch.float32)
g, raw, _ = bench.build_batch(dev, seed, a.graphs, a.workload)
et = g.edata["label"]


def build():
    g._cache.clear()
    ix = g.row_index(et, R, True, closing_hint=(H, dtype))
    if dtype == torch.bfloat16:
        for _, _, part in ix.parts:
            ops.prepare_closing(part, H, dtype)
    return ix


# time spent INSIDE the blocking library calls (they launch and wait for the read-back) vs around them
from dummynode4graphlearning_amd import _lib  # noqa: E402
L = _lib.lib()
inside = [0.0, 0]
for name in ("dn_conv_index_build_i32", "dn_row_index_build_local_i32"):
    fn = getattr(L, name)

    def wrapped(*a, _fn=fn):
        t = time.perf_counter()
        r = _fn(*a)
        inside[0] += time.perf_counter() - t
        inside[1] += 1
        return r
    setattr(L, name, wrapped)

for _ in range(5):
    build()
inside[:] = [0.0, 0]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    build()
torch.cuda.synchronize()
print("%.1f us per build (wall, %d builds back to back); %.1f us of it inside the blocking index call (%d calls)" % (
    (time.perf_counter() - t0) / a.reps * 1e6, a.reps, inside[0] / a.reps * 1e6, inside[1]))
pr = cProfile.Profile()
pr.enable()
for _ in range(a.reps):
    build()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
