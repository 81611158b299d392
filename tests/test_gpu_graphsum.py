"""dn_graph_tile_sum_f32 (graph-local neighbour sum as a dense product on the matrix cores) against the plain gather kernel and
fp64: same sums up to fp32 summation order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(rng, G, nmin, nmax, deg, hub=True):
    src, dst, nptr = [], [], [0]
    for g in range(G):
        n = int(rng.integers(nmin, nmax + 1))
        base = nptr[-1]
        m = int(deg * n)
        if n > 0 and m > 0:
            s, d = rng.integers(0, n, size=m), rng.integers(0, n, size=m)
            src += list(base + s); dst += list(base + d)
        if hub and n > 1:                                            # a dummy node wired to every node, both ways
            for u in range(n - 1):
                src += [base + u, base + n - 1]; dst += [base + n - 1, base + u]
        nptr.append(base + n)
    return np.array(src, np.int64), np.array(dst, np.int64), np.array(nptr, np.int64)


@pytest.mark.parametrize("H", [64, 128, 256])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_graph_tile_sum_equals_the_plain_gather(seed, H):
    from dummynode4graphlearning_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    src, dst, nptr = _batch(rng, G=int(rng.integers(3, 200)), nmin=0, nmax=int(rng.integers(2, 64)), deg=float(rng.uniform(0.5, 4)))
    N = int(nptr[-1])
    ei = ops.EdgeIndex(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), N)
    x = torch.randn(N, H, device=dev) * torch.exp(torch.randn(N, 1, device=dev) * 2)       # wide dynamic range across rows
    tiles, covered, rest = ops.graph_tiles(torch.from_numpy(nptr).to(dev), 64)
    assert covered == N and rest.numel() == 0
    for ptr_, idx in ((ei.in_ptr, ei.src_by_dst), (ei.out_ptr, ei.dst_by_src)):           # forward (CSR by dst) and its transpose
        want = ops.gather_segsum(x, idx, ptr_, N, self_in=x, self_coef=1.25)
        got, bad = ops.graph_tile_sum(x, idx, ptr_, tiles, self_coef=1.25)
        assert int(bad.item()) == 0
        seg = torch.repeat_interleave(torch.arange(N, device=dev, dtype=torch.int32), (ptr_[1:] - ptr_[:-1]).long())
        got2, bad2 = ops.graph_tile_sum(x, idx, ptr_, ops.graph_tile_records(tiles, ptr_), self_coef=1.25, seg=seg)
        assert int(bad2.item()) == 0 and torch.equal(got, got2)          # (the destination rows given instead of searched)
        ref = 1.25 * x.double()
        ref.index_add_(0, torch.repeat_interleave(torch.arange(N, device=dev), (ptr_[1:] - ptr_[:-1]).long()), x.double()[idx.long()])
        scale = ref.abs().max(dim=1, keepdim=True).values.clamp_min(1e-30)
        e_got, e_want = ((got.double() - ref).abs() / scale).max().item(), ((want.double() - ref).abs() / scale).max().item()
        assert e_got < 2e-6, (e_got, e_want)


def test_graph_tile_sum_flags_a_neighbour_outside_its_tile():
    from dummynode4graphlearning_amd import ops
    dev = torch.device("cuda:0")
    src, dst = torch.tensor([0, 1, 5], device=dev), torch.tensor([1, 0, 1], device=dev)     # 5 -> 1 crosses the tile boundary
    ei = ops.EdgeIndex(src, dst, 8)
    x = torch.randn(8, 64, device=dev)
    tiles = torch.tensor([[0, 4], [4, 8]], dtype=torch.int32, device=dev)
    _, bad = ops.graph_tile_sum(x, ei.src_by_dst, ei.in_ptr, tiles)
    assert int(bad.item()) == 1


def test_neighbor_sum_takes_the_tile_path_and_matches_the_plain_gather():
    """ops.neighbor_sum on an index that knows its graph boundaries: tiles for graphs of <= 64 rows, row lists (lane group per row,
    workgroup per hub) for the rest; forward and backward against the plain path (DN_TILE_SUM off) and fp64."""
    from dummynode4graphlearning_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    src, dst, nptr = _batch(rng, G=120, nmin=0, nmax=200, deg=2.5)                   # small and large graphs, hubs of up to 199 entries
    assert int(nptr[-1]) >= ops.TILE_SUM_MIN_ROWS                                     # (smaller batches keep the plain kernel)
    N = int(nptr[-1])
    ei = ops.EdgeIndex(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), N, node_ptr=torch.from_numpy(nptr).to(dev))
    plan = ei.tile_plan()
    assert plan is not None and 0 < plan.covered < N
    assert all(t[1].numel() > 0 and t[2].numel() > 0 for t in plan.dirs.values())       # both row lists are exercised
    for H in (64, 128):
        x = torch.randn(N, H, device=dev, requires_grad=True)
        go = torch.randn(N, H, device=dev)
        out = ops.neighbor_sum(x, ei, 1.5)
        out.backward(go)
        assert plan.checked and int(plan.bad.item()) == 0
        x2 = x.detach().clone().requires_grad_(True)
        try:
            ops.TILE_SUM_ENABLED = False
            out2 = ops.neighbor_sum(x2, ei, 1.5)
            out2.backward(go)
        finally:
            ops.TILE_SUM_ENABLED = True
        xd = x.detach().double()
        ref = 1.5 * xd
        ref.index_add_(0, torch.from_numpy(dst).to(dev), xd[torch.from_numpy(src).to(dev)])
        gref = 1.5 * go.double()
        gref.index_add_(0, torch.from_numpy(src).to(dev), go.double()[torch.from_numpy(dst).to(dev)])
        for got, plain, want in ((out, out2, ref), (x.grad, x2.grad, gref)):
            scale = want.abs().max(dim=1, keepdim=True).values.clamp_min(1e-30)
            assert ((got.detach().double() - want).abs() / scale).max().item() < 2e-6
            assert ((plain.detach().double() - want).abs() / scale).max().item() < 2e-6
