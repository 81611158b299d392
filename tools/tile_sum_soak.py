"""Random sweep: ops.neighbor_sum through the matrix-core tile path vs the plain gather and fp64 (forward and backward)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_graphsum as T
from dummynode4graphlearning_amd import ops
dev = torch.device("cuda:0")
worst, n_tile = 0.0, 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    rng = np.random.default_rng(7000 + seed)
    src, dst, nptr = T._batch(rng, G=int(rng.integers(80, 400)), nmin=0, nmax=int(rng.integers(30, 160)), deg=float(rng.uniform(0.3, 5.0)),
                              hub=bool(rng.integers(0, 2)))
    N = int(nptr[-1])
    if N < ops.TILE_SUM_MIN_ROWS:
        continue
    H = int(rng.choice([64, 128, 256]))
    s, d = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
    ei = ops.EdgeIndex(s, d, N, node_ptr=torch.from_numpy(nptr).to(dev))
    if ei.tile_plan() is None:
        continue
    n_tile += 1
    coef = float(rng.choice([0.0, 1.0, 1.37]))
    x = (torch.randn(N, H, device=dev) * torch.exp(torch.randn(N, 1, device=dev))).requires_grad_(True)
    go = torch.randn(N, H, device=dev)
    out = ops.neighbor_sum(x, ei, coef); out.backward(go)
    assert ei.tile_plan() is not None and int(ei.tile_plan().bad.item()) == 0
    xd = x.detach().double()
    ref = coef * xd; ref.index_add_(0, d, xd[s])
    gref = coef * go.double(); gref.index_add_(0, s, go.double()[d])
    for got, want in ((out, ref), (x.grad, gref)):
        scale = want.abs().max(dim=1, keepdim=True).values.clamp_min(1e-30)
        worst = max(worst, ((got.detach().double() - want).abs() / scale).max().item())
print("tile-path batches: %d, worst relative error vs fp64: %.2e" % (n_tile, worst))
assert worst < 3e-6
