import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_local_index as T
bad = 0
for seed in range(300):
    rng = np.random.default_rng(10_000 + seed)
    R = int(rng.integers(1, 20))
    src, dst, et, nptr, eptr = T._random_batch(rng, G=int(rng.integers(1, 150)), R=max(R, 3), nmin=0, nmax=int(rng.integers(1, 70)),
                                               dens=float(rng.uniform(0.1, 6.0)), dummy=bool(rng.integers(0, 2)), multi=True)
    for sl in (True, False):
        for ef in (0.75, float(rng.uniform(0, 2))):
            try:
                a, b = T._build(src, dst, et, int(nptr[-1]), max(R, 3), sl, nptr, eptr, edge_frac=ef)
                if b.built_by == "local":
                    T._same(a, b)
            except AssertionError as e:
                bad += 1
                print("MISMATCH seed", seed, sl, ef, str(e)[:200])
print("soak done, mismatches:", bad)
# round 6: the sort pass (graphs of 257 .. 1024 edges and small ones the bit sets do not take) and the listed-graph pass (1025 .. 8191)
bad2 = served = 0
for seed in range(120):
    rng = np.random.default_rng(50_000 + seed)
    R = int(rng.integers(3, 20))
    nmax = int(rng.choice([90, 200, 400, 900, 1800]))
    src, dst, et, nptr, eptr = T._random_batch(rng, G=int(rng.integers(1, 25)), R=R, nmin=0, nmax=nmax, dens=float(rng.uniform(0.3, 3.5)),
                                               dummy=bool(rng.integers(0, 2)), multi=True)
    if len(src) == 0 or max(np.diff(eptr)) >= 8192:
        continue
    for sl in (True, False):
        for ef in (0.75, float(rng.uniform(0, 2))):
            try:
                a, b = T._build(src, dst, et, int(nptr[-1]), R, sl, nptr, eptr, edge_frac=ef)
                assert b.built_by == "local", "fell back"
                T._same(a, b)
                served += 1
            except AssertionError as e:
                bad2 += 1
                print("MISMATCH (large) seed", seed, sl, ef, max(np.diff(eptr)), str(e)[:200])
print("large-graph soak done: %d builds, mismatches: %d" % (served, bad2))
