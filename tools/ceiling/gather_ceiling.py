"""Tuning aid: rate of a bare gather-add-write stream with the traffic shapes of the two conv launches (see the .hip).
build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ceiling/libceil.so tools/ceiling/gather_ceiling.hip"""
import ctypes, os, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libceil.so"))
lib.gather_sum.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                           ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
G, n = 32768, 31
N = G * n
rng = np.random.default_rng(0)
x = torch.randn(N, 256, device=dev, dtype=torch.bfloat16)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def run(idx, P, R, U, blocks, nt, y):
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.gather_sum(x.data_ptr(), idx.data_ptr(), P, R, U, blocks, nt, y.data_ptr(), st)
    assert rc == 0, rc


# shape of the transform launch: 14 relation blocks, each sweeping the graphs in order, source = a random node of the graph
rows = []
for r in range(14):
    k = rng.integers(3, 6, size=G)                         # rows of this relation per graph (~4.4 -> 62 per graph over 14)
    gid = np.repeat(np.arange(G), k)
    rows.append(gid * n + rng.integers(0, 30, size=gid.size))
idx1 = torch.from_numpy(np.concatenate(rows).astype(np.int32)).to(dev)
P1 = idx1.numel()
y1 = torch.empty(P1, 256, device=dev, dtype=torch.bfloat16)
print("transform shape: %d rows gathered from %d (%.2f GB in + %.2f GB out)" % (P1, N, P1 * 512 / 1e9, P1 * 512 / 1e9))
for U in (2, 4, 8):
    for blocks in (1024, 2048, 4096):
        for nt in (0, 1):
            t = timed(lambda: run(idx1, P1, 1, U, blocks, nt, y1))
            print("  R=1 U=%d blocks=%d nt=%d: %.1f us  %.2f TB/s" % (U, blocks, nt, t * 1e6, 2 * P1 * 512 / t / 1e12))
# shape of the closing launch: per node its own row (in order) + two rows of a 2 M-row table (14 ascending streams) -> one row
S = y1
P3 = N
own = np.arange(N)
a = rng.integers(0, P1 // 14, size=N) // 1 + (rng.integers(0, 14, size=N) * (P1 // 14))
b = rng.integers(0, 14, size=N) * (P1 // 14)
pos = (np.arange(N) * (P1 // 14) // N)
idx3 = np.stack([np.minimum(pos + a % 7, P1 - 1) + 0 * b, np.minimum(pos + b, P1 - 1), np.minimum(pos + (a % 14) * (P1 // 14), P1 - 1)], 1)
idx3 = torch.from_numpy(idx3.astype(np.int32).reshape(-1)).to(dev)
y3 = torch.empty(N, 256, device=dev, dtype=torch.bfloat16)
xs = x
x = S                                                       # gather from the 2 M-row table
print("closing shape: %d rows out, 3 gathered rows each from a %d-row table" % (N, P1))
for U in (1, 2, 4):
    for blocks in (1024, 2048, 4096):
        t = timed(lambda: run(idx3, P3, 3, U, blocks, 1, y3))
        print("  R=3 U=%d blocks=%d: %.1f us  %.2f TB/s (4 x 0.52 GB)" % (U, blocks, t * 1e6, 4 * N * 512 / t / 1e12))
