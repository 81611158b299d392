"""GPU parity of the segment kernels and CSR build against the oracle (CPU torch / numpy), through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import layers as OL
from oracle import transforms as OT

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from dummynode4graphlearning_amd import ops
    return ops


def _rand_csr(rng, S, M_avg, in_rows, skew=True):
    deg = rng.poisson(M_avg, size=S)
    deg[rng.integers(0, S, size=max(1, S // 10))] = 0          # empty segments
    if skew and S > 2:
        deg[rng.integers(0, S)] = 40 * max(M_avg, 1) + 3            # one dummy-like long segment
    ptr = np.zeros(S + 1, dtype=np.int64)
    np.cumsum(deg, out=ptr[1:])
    idx = rng.integers(0, in_rows, size=int(ptr[-1]))
    return ptr, idx


def _ref_gather_segsum(x, idx, ptr, scale=None, self_in=None, self_coef=0.0, mean=False):
    S = len(ptr) - 1
    seg = torch.repeat_interleave(torch.arange(S), torch.as_tensor(np.diff(ptr)))
    rows = x[torch.as_tensor(idx)] if idx is not None else x[: int(ptr[-1])]
    if scale is not None:
        rows = rows * scale.view(-1, 1)
    out = OL.segment_sum(rows.double(), seg, S)
    if mean:
        cnt = torch.as_tensor(np.diff(ptr)).clamp(min=1).double().view(-1, 1)
        out = out / cnt
    if self_in is not None:
        out = out + self_coef * self_in.double()
    return out


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("H", [2, 5, 8, 64, 128, 256, 260, 520])
def test_gather_segsum_matches_oracle(dtype, tol, H):
    ops = _ops()
    rng = np.random.default_rng(H)
    rows, S = 300, 257
    ptr, idx = _rand_csr(rng, S, 3, rows)
    x = torch.from_numpy(rng.standard_normal((rows, H)).astype(np.float32)).to(dtype)
    scale = torch.from_numpy(rng.uniform(0.2, 1.5, size=len(idx)).astype(np.float32))
    self_in = torch.from_numpy(rng.standard_normal((S, H)).astype(np.float32)).to(dtype)
    xd, idxd, ptrd = x.to(DEV), torch.from_numpy(idx).to(DEV, torch.int32), torch.from_numpy(ptr).to(DEV, torch.int32)
    for use_scale, use_self, mean in [(False, False, False), (True, False, False), (False, True, False),
                                      (True, True, True), (False, False, True)]:
        got = ops.gather_segsum(xd, idxd, ptrd, scale=scale.to(DEV) if use_scale else None,
                                self_in=self_in.to(DEV) if use_self else None, self_coef=1.25 if use_self else 0.0,
                                mean=mean)
        ref = _ref_gather_segsum(x.float(), idx, ptr, scale if use_scale else None,
                                 self_in.float() if use_self else None, 1.25, mean)
        # tolerance: fp32 accumulate; bf16 differs only by the final rounding of the stored row (2^-8 rel)
        torch.testing.assert_close(got.cpu().double(), ref, rtol=tol, atol=tol * max(1.0, float(ref.abs().max())) * 0.5)


def test_gather_rows_and_contiguous_segments():
    ops = _ops()
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((100, 64)).astype(np.float32))
    idx = torch.from_numpy(rng.integers(0, 100, size=333)).to(torch.int32)
    got = ops.gather_segsum(x.to(DEV), idx.to(DEV), None)                        # ptr == NULL: pure row gather
    torch.testing.assert_close(got.cpu(), x[idx.long()], rtol=0, atol=0)
    ptr = torch.tensor([0, 10, 10, 55, 100], dtype=torch.int32)
    got = ops.gather_segsum(x.to(DEV), None, ptr.to(DEV))                        # idx == NULL: contiguous rows
    ref = torch.stack([x[0:10].sum(0), torch.zeros(64), x[10:55].sum(0), x[55:100].sum(0)])
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-5)


def test_gather_segsum_is_bitwise_deterministic():
    ops = _ops()
    rng = np.random.default_rng(3)
    ptr, idx = _rand_csr(rng, 5000, 4, 4000)
    x = torch.from_numpy(rng.standard_normal((4000, 128)).astype(np.float32)).to(DEV)
    idxd, ptrd = torch.from_numpy(idx).to(DEV, torch.int32), torch.from_numpy(ptr).to(DEV, torch.int32)
    a = ops.gather_segsum(x, idxd, ptrd)
    for _ in range(3):
        b = ops.gather_segsum(x, idxd, ptrd)
        assert torch.equal(a, b)


def test_empty_inputs():
    ops = _ops()
    x = torch.zeros((0, 64), device=DEV)
    ptr = torch.zeros(1, dtype=torch.int32, device=DEV)
    assert ops.gather_segsum(x, torch.zeros(0, dtype=torch.int32, device=DEV), ptr).shape == (0, 64)
    ptr = torch.zeros(4, dtype=torch.int32, device=DEV)                          # 3 empty segments
    out = ops.gather_segsum(torch.ones((5, 8), device=DEV), torch.zeros(0, dtype=torch.int32, device=DEV), ptr)
    assert torch.equal(out.cpu(), torch.zeros(3, 8))


@pytest.mark.parametrize("M,K", [(0, 7), (1, 1), (1000, 37), (50000, 1200), (4096, 100000)])
def test_csr_build_bit_exact(M, K):
    ops = _ops()
    rng = np.random.default_rng(M + K)
    key = rng.integers(0, K, size=M)
    ptr, perm = ops.csr_build(torch.from_numpy(key).to(DEV, torch.int32), K)
    rptr, rperm = OT.csr_by_key(key, K)
    np.testing.assert_array_equal(ptr.cpu().numpy(), rptr)
    np.testing.assert_array_equal(perm.cpu().numpy(), rperm)


@pytest.mark.parametrize("kind", ["sum", "mean", "max"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H", [2, 7, 64])
def test_segment_readouts_forward_backward(kind, dtype, H):
    ops = _ops()
    rng = np.random.default_rng(H)
    sizes = np.array([3, 0, 17, 1, 40, 0, 9])
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    N, B = int(ptr[-1]), len(sizes)
    x = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32)).to(dtype)
    coef = torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)).to(dtype)
    xd = x.to(DEV).requires_grad_(True)
    out = ops.segment_reduce(xd, torch.from_numpy(ptr).to(DEV, torch.int32), kind)
    (out.float() * coef.to(DEV).float()).sum().backward()
    xr = x.float().requires_grad_(True)
    batch = torch.repeat_interleave(torch.arange(B), torch.from_numpy(sizes))
    ref = OL.global_pool(xr, batch, B, {"sum": "add"}.get(kind, kind))
    (ref * coef.float()).sum().backward()
    tol = 1e-5 if dtype == torch.float32 else 1.6e-2
    torch.testing.assert_close(out.detach().cpu().float(), ref.detach(), rtol=tol, atol=tol)
    torch.testing.assert_close(xd.grad.cpu().float(), xr.grad, rtol=tol, atol=tol)


def test_neighbor_sum_autograd_matches_oracle():
    ops = _ops()
    rng = np.random.default_rng(5)
    N, E, H = 200, 900, 64
    src, dst = rng.integers(0, N, size=E), rng.integers(0, N, size=E)
    dst[:150] = 7                                                                # dummy-like hub
    x = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
    coef = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
    index = ops.EdgeIndex(torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV), N)
    xd = x.to(DEV).requires_grad_(True)
    out = ops.neighbor_sum(xd, index, 1.5)
    (out * coef.to(DEV)).sum().backward()
    xr = x.clone().requires_grad_(True)
    ref = OL.gin_conv(xr, torch.from_numpy(src), torch.from_numpy(dst), 0.5, lambda t: t)
    (ref * coef).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4)


@pytest.fixture(params=["bf16", "f32-split", "f32-exact"])
def arith(request):
    """Storage type + fp32 arithmetic mode of the matrix kernels: bf16, fp32 on the 3-term bf16 split (default), exact fp32."""
    ops = _ops()
    old = ops.F32_EXACT
    ops.F32_EXACT = request.param == "f32-exact"
    yield (torch.bfloat16 if request.param == "bf16" else torch.float32), request.param
    ops.F32_EXACT = old


@pytest.mark.parametrize("H", [64, 128, 256])
def test_rows_wgrad_matches_reference(H, arith):
    """MFMA split-K weight gradient: out[r] = sum_p A[ia[p]]^T G[ig[p]] (asymmetric data: catches transposed tiles)."""
    dt, mode = arith
    # sums of up to 9001 products of N(0,1) values: |out| ~ 100; the split's products carry O(2^-16) relative error each
    watol = 2e-2 if mode == "f32-split" else 1e-3
    ops = _ops()
    rng = np.random.default_rng(H)
    R = 5
    sizes = [0, 37, 5000, 1, 9001]                      # empty relation, tiny, multi-chunk, single row, ragged tail
    rel_ptr = [0] + list(np.cumsum(sizes))
    P, NA, NG = rel_ptr[-1], 3000, 2500
    A = torch.from_numpy(rng.standard_normal((NA, H)).astype(np.float32)).to(dt)
    G = torch.from_numpy(rng.standard_normal((NG, H)).astype(np.float32)).to(dt)
    ia = torch.from_numpy(rng.integers(0, NA, size=P)).to(torch.int32)
    ig = torch.from_numpy(rng.integers(0, NG, size=P)).to(torch.int32)
    table = ops.make_row_chunks([int(v) for v in rel_ptr], DEV, chunk_rows=2048)
    got = ops.rows_wgrad(A.to(DEV), G.to(DEV), table, R, idx_a=ia.to(DEV), idx_g=ig.to(DEV), out_dtype=torch.float32)
    ref = torch.zeros(R, H, H, dtype=torch.float64)
    for r in range(R):
        a, b = rel_ptr[r], rel_ptr[r + 1]
        ref[r] = A[ia[a:b].long()].double().t() @ G[ig[a:b].long()].double()
    # bf16 products are exact in fp32; only the fp32 accumulation order differs
    torch.testing.assert_close(got.cpu().double(), ref, rtol=1e-4, atol=watol)
    got2, cs = ops.rows_wgrad(A.to(DEV), G.to(DEV), table, R, idx_a=ia.to(DEV), idx_g=ig.to(DEV), out_dtype=torch.float32,
                              colsum_of=2)
    assert torch.equal(got, got2)                        # deterministic
    cs_ref = torch.stack([G[ig[rel_ptr[r]:rel_ptr[r + 1]].long()].double().sum(0) for r in range(R)])
    torch.testing.assert_close(cs.cpu().double(), cs_ref, rtol=1e-4, atol=1e-3)
    _, cs_a = ops.rows_wgrad(A.to(DEV), G.to(DEV), table, R, idx_a=ia.to(DEV), idx_g=ig.to(DEV), colsum_of=1)
    cs_ref = torch.stack([A[ia[rel_ptr[r]:rel_ptr[r + 1]].long()].double().sum(0) for r in range(R)])
    torch.testing.assert_close(cs_a.cpu().double(), cs_ref, rtol=1e-4, atol=1e-3)
    # operands as virtual concatenations [A1; A2], [G1; G2] (the conv's rows + its pre-aggregated rows): the same bits
    got4 = ops.rows_wgrad(A[:2000].contiguous().to(DEV), G[:1700].contiguous().to(DEV), table, R, idx_a=ia.to(DEV), idx_g=ig.to(DEV),
                          out_dtype=torch.float32, A2=A[2000:].contiguous().to(DEV), G2=G[1700:].contiguous().to(DEV))
    assert torch.equal(got, got4)
    # contiguous rows (no index) + bf16 output
    P2 = min(P, NA, NG)
    t2 = ops.make_row_chunks([0, P2], DEV, chunk_rows=1024)
    got3 = ops.rows_wgrad(A[:P2].contiguous().to(DEV), G[:P2].contiguous().to(DEV), t2, 1)
    ref3 = A[:P2].double().t() @ G[:P2].double()
    torch.testing.assert_close(got3[0].cpu().double(), ref3, rtol=1e-2, atol=0.5 if dt == torch.bfloat16 else 1e-2)


@pytest.mark.parametrize("H", [64, 128, 256])
def test_rows_transform_matches_reference(H, arith):
    """Gathered-row MFMA transform: Y[p] = epi(Xcat[idx[p]] @ Wn[rel(p)]^T) with asymmetric weights."""
    dt, mode = arith
    ops = _ops()
    tol = {"bf16": dict(rtol=1e-2, atol=2e-2), "f32-exact": dict(rtol=1e-5, atol=1e-5), "f32-split": dict(rtol=1e-4, atol=1e-4)}[mode]
    rng = np.random.default_rng(10 + H)
    sizes = [0, 37, 1500, 1, 33, 64]
    rel_ptr = [0] + [int(v) for v in np.cumsum(sizes)]
    R, P, N1, N2 = len(sizes), rel_ptr[-1], 700, 90
    X = torch.from_numpy(rng.standard_normal((N1, H)).astype(np.float32)).to(dt)
    X2 = torch.from_numpy(rng.standard_normal((N2, H)).astype(np.float32)).to(dt)
    Wn = torch.from_numpy((rng.standard_normal((R, H, H)) / np.sqrt(H)).astype(np.float32)).to(dt)
    bias = torch.from_numpy(rng.standard_normal((R, H)).astype(np.float32)).to(dt)
    idx = torch.from_numpy(rng.integers(0, N1 + N2, size=P)).to(torch.int32)
    tiles = ops.make_row_tiles(rel_ptr, DEV)
    Xcat = torch.cat([X, X2]).double()
    rel_of_row = torch.repeat_interleave(torch.arange(R), torch.tensor(sizes))
    for use_bias, relu in [(False, False), (True, True)]:
        got = ops.rows_transform(X.to(DEV), Wn.to(DEV), tiles, P, idx=idx.to(DEV), X2=X2.to(DEV),
                                 bias=bias.to(DEV) if use_bias else None, relu=relu)
        ref = torch.einsum("pk,pnk->pn", Xcat[idx.long()], Wn.double()[rel_of_row])
        if use_bias:
            ref = ref + bias.double()[rel_of_row]
        if relu:
            ref = ref.clamp(min=0)
        # fp32 accumulate of exact products; bf16 adds one rounding of the stored result
        torch.testing.assert_close(got.cpu().double(), ref, **tol)
        mask = torch.from_numpy(rng.standard_normal((P, H)).astype(np.float32)).to(dt)
        gotm = ops.rows_transform(X.to(DEV), Wn.to(DEV), tiles, P, idx=idx.to(DEV), X2=X2.to(DEV),
                                  bias=bias.to(DEV) if use_bias else None, relu=relu, mask_pos=mask.to(DEV))
        torch.testing.assert_close(gotm.cpu().double(), ref * (mask.double() > 0), **tol)
    # identity rows, single source
    t1 = ops.make_row_tiles([0, N1], DEV)
    got = ops.rows_transform(X.to(DEV), Wn[2:3].contiguous().to(DEV), t1, N1)
    torch.testing.assert_close(got.cpu().double(), X.double() @ Wn[2].double().t(), **tol)
    # many tiles per workgroup (steady state of the gather pipelines / the LDS-DMA ring), relation changes inside a workgroup
    big = [70001, 3, 0, 90017, 40000]
    bptr = [0] + [int(v) for v in np.cumsum(big)]
    PB = bptr[-1]
    bidx = torch.from_numpy(rng.integers(0, N1 + N2, size=PB)).to(torch.int32)
    tb = ops.make_row_tiles(bptr, DEV)
    got = ops.rows_transform(X.to(DEV), Wn[:5].contiguous().to(DEV), tb, PB, idx=bidx.to(DEV), X2=X2.to(DEV))
    brel = torch.repeat_interleave(torch.arange(5), torch.tensor(big))
    for r in range(5):
        sel = (brel == r).nonzero().reshape(-1)
        if sel.numel():
            ref_r = Xcat[bidx[sel].long()] @ Wn[r].double().t()
            torch.testing.assert_close(got[sel.to(DEV)].cpu().double(), ref_r, **tol)


@pytest.mark.parametrize("H", [64, 128, 256])
@pytest.mark.parametrize("exact", [False, True])
def test_fp32_transform_on_the_parameters_where_they_lie(H, exact):
    """dn_rows_transform_f32 with weight [R, in, out] / loop_weight [in, out] / h_bias [out] read in place (w_kn, W_loop /
    loop_rel, bias_rel) gives the SAME BITS as the concatenated + transposed weight copy and the padded bias matrix it replaces."""
    ops = _ops()
    rng = np.random.default_rng(77 + H)
    sizes = [129, 0, 65, 700, 31, 260]                          # the last relation is the self loop
    rel_ptr = [0] + [int(v) for v in np.cumsum(sizes)]
    R, P, N1 = len(sizes) - 1, rel_ptr[-1], 500
    X = torch.from_numpy(rng.standard_normal((N1, H)).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rng.standard_normal((R, H, H)) / np.sqrt(H)).astype(np.float32)).to(DEV)          # [R, in, out]
    W_loop = torch.from_numpy((rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float32)).to(DEV)       # [in, out]
    bias = torch.from_numpy(rng.standard_normal(H).astype(np.float32)).to(DEV)
    idx = torch.from_numpy(rng.integers(0, N1, size=P)).to(torch.int32).to(DEV)
    tiles = ops.make_row_tiles(rel_ptr, DEV)
    W_all = torch.cat([W, W_loop.unsqueeze(0)], 0).transpose(1, 2).contiguous()                           # [R + 1, out, in]
    bias_all = torch.zeros((R + 1, H), device=DEV)
    bias_all[-1] = bias
    with ops.f32_exact(exact):
        want = ops.rows_transform(X, W_all, tiles, P, idx=idx, bias=bias_all, relu=True, slope=0.25)
        got = ops.rows_transform(X, W, tiles, P, idx=idx, bias=bias, relu=True, slope=0.25, w_kn=True, W_loop=W_loop, loop_rel=R,
                                 bias_rel=R)
        assert torch.equal(got, want)
        # [n][k] matrices with a separate loop matrix (the input-gradient pass), no bias
        want = ops.rows_transform(X, torch.cat([W, W_loop.unsqueeze(0)], 0), tiles, P, idx=idx)
        got = ops.rows_transform(X, W, tiles, P, idx=idx, W_loop=W_loop, loop_rel=R)
        assert torch.equal(got, want)
    ref = torch.einsum("pk,pkn->pn", X[idx.long()].double().cpu(),
                       torch.cat([W, W_loop.unsqueeze(0)], 0).double().cpu()[torch.repeat_interleave(torch.arange(R + 1), torch.tensor(sizes))])
    with ops.f32_exact(exact):
        got = ops.rows_transform(X, W, tiles, P, idx=idx, w_kn=True, W_loop=W_loop, loop_rel=R)
    torch.testing.assert_close(got.cpu().double(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("H", [8, 64, 128, 256, 520])
def test_gather_segsum_over_few_long_lists(dtype, tol, H):
    """The workgroup-per-segment form (few segments of >= 24 rows on average: the pre-aggregation of a collapsed dummy relation on
    a small batch), incl. empty and very long lists, scale / mean / self term, and bitwise reproducibility."""
    ops = _ops()
    rng = np.random.default_rng(900 + H)
    rows, S = 4000, 301
    deg = rng.integers(20, 80, size=S)
    deg[[0, 17, S - 1]] = 0
    deg[5] = 1500
    deg[6] = 1
    ptr = np.zeros(S + 1, dtype=np.int64)
    np.cumsum(deg, out=ptr[1:])
    idx = rng.integers(0, rows, size=int(ptr[-1]))
    x = torch.from_numpy(rng.standard_normal((rows, H)).astype(np.float32)).to(dtype)
    scale = torch.from_numpy(rng.uniform(0.2, 1.5, size=len(idx)).astype(np.float32))
    self_in = torch.from_numpy(rng.standard_normal((S, H)).astype(np.float32)).to(dtype)
    xd, idxd, ptrd = x.to(DEV), torch.from_numpy(idx).to(DEV, torch.int32), torch.from_numpy(ptr).to(DEV, torch.int32)
    for use_scale, use_self, mean in [(False, False, False), (True, True, True), (False, True, False), (True, False, False)]:
        kw = dict(scale=scale.to(DEV) if use_scale else None, self_in=self_in.to(DEV) if use_self else None,
                  self_coef=1.25 if use_self else 0.0, mean=mean)
        got = ops.gather_segsum(xd, idxd, ptrd, **kw)
        ref = _ref_gather_segsum(x.float(), idx, ptr, scale if use_scale else None, self_in.float() if use_self else None, 1.25, mean)
        torch.testing.assert_close(got.cpu().double(), ref, rtol=tol, atol=tol * max(1.0, float(ref.abs().max())) * 0.5)
        assert torch.equal(got, ops.gather_segsum(xd, idxd, ptrd, **kw))
    # contiguous lists (idx == None)
    n = int(ptr[-1])
    xc = torch.from_numpy(rng.standard_normal((n, H)).astype(np.float32)).to(dtype)
    got = ops.gather_segsum(xc.to(DEV), None, ptrd)
    ref = _ref_gather_segsum(xc.float(), None, ptr)
    torch.testing.assert_close(got.cpu().double(), ref, rtol=tol, atol=tol * max(1.0, float(ref.abs().max())) * 0.5)


@pytest.mark.parametrize("self_loop", [True, False])
def test_fused_row_factorisation_forward_backward(self_loop, arith):
    """Fused path (EDGE / AGG / TF relations + self loop), bf16 / fp32 split / exact-f32 MFMA, against the fp64 per-edge
    formulation."""
    dt, mode = arith
    ops = _ops()
    from dummynode4graphlearning_amd import synthetic
    raw = synthetic.config3(seed=7, graphs=24)
    aug = OT.dummy_augment_si(*(raw[k] for k in ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id",
                                                  "edge_label")), raw["max_nv"], raw["max_nvl"], raw["max_ne"], raw["max_nel"])
    src, dst, et = (torch.from_numpy(aug[k]) for k in ("src", "dst", "edge_label"))
    N, R, H = len(aug["node_label"]), raw["num_rels"], 64
    rng = np.random.default_rng(1)
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(dt)  # noqa: E731
    x = bf(rng.standard_normal((N, H)))
    W = bf(rng.standard_normal((R + (1 if self_loop else 0), H, H)) / np.sqrt(H))
    b = bf(rng.standard_normal(H)) if self_loop else None
    coef = bf(rng.standard_normal((N, H)))
    index = ops.RowIndex(src.to(DEV), dst.to(DEV), et.to(DEV), N, R, self_loop=self_loop)
    assert sorted(set(index.modes)) == [ops.RowIndex.EDGE, ops.RowIndex.AGG, ops.RowIndex.TF]
    xd, Wd = x.to(DEV).requires_grad_(True), W.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if self_loop else None
    out = ops.rel_transform_fused(xd, Wd, bd, index)
    out.backward(coef.to(DEV))
    xr, Wr = x.double().requires_grad_(True), W.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if self_loop else None
    msg = torch.bmm(xr[src].unsqueeze(1), Wr[et]).squeeze(1)
    ref = torch.zeros(N, H, dtype=torch.float64).index_add(0, dst, msg)
    if self_loop:
        ref = ref + xr @ Wr[R] + br
    ref.backward(coef.double())

    def rel(a, r):
        return float((a.detach().cpu().double() - r.detach()).abs().max() / r.detach().abs().max())
    # bf16 storage of every intermediate (2^-8 relative each): 2e-2 of the tensor range end to end; exact fp32: 1e-5;
    # 3-term bf16 split: 5e-5 (measured 4-9e-6)
    lim = {"bf16": 2e-2, "f32-exact": 1e-5, "f32-split": 5e-5}[mode]
    assert rel(out, ref) < lim
    assert rel(xd.grad, xr.grad) < lim
    assert rel(Wd.grad, Wr.grad) < lim
    if self_loop:
        assert rel(bd.grad, br.grad) < lim


def test_neighbor_sum_with_hub_splitting():
    """Dummy-node hubs (in- and out-degree ~600, as in PROTEINS-sized graphs) go through the split path; same result."""
    ops = _ops()
    rng = np.random.default_rng(6)
    N, H = 1500, 128
    src, dst = list(rng.integers(0, N, size=4000)), list(rng.integers(0, N, size=4000))
    for hub, n in ((7, 620), (900, 65), (1499, 300)):
        others = rng.choice(N, size=n, replace=False)
        src += list(others) + [hub] * n
        dst += [hub] * n + list(others)
    src, dst = np.array(src), np.array(dst)
    x = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(0.5, 1.5, size=len(src)).astype(np.float32))
    coef = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
    index = ops.EdgeIndex(torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV), N)
    assert index.fwd.hub_ids is not None and index.fwd.hub_ids.numel() == 3 and index.bwd.hub_ids.numel() == 3
    for scale in (None, w):
        xd = x.to(DEV).requires_grad_(True)
        out = ops.neighbor_sum(xd, index, 1.0, edge_scale=None if scale is None else scale.to(DEV))
        (out * coef.to(DEV)).sum().backward()
        xr = x.double().requires_grad_(True)
        rows = xr[torch.from_numpy(src)] * (1.0 if scale is None else scale.double().view(-1, 1))
        ref = xr + torch.zeros(N, H, dtype=torch.float64).index_add(0, torch.from_numpy(dst), rows)
        (ref * coef.double()).sum().backward()
        torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H", [3, 16, 64, 200])
def test_edge_dot_and_neighbor_max(dtype, H):
    ops = _ops()
    rng = np.random.default_rng(H)
    N, E = 300, 1500
    src, dst = rng.integers(0, N, size=E), rng.integers(0, N, size=E)
    dst[dst == 5] = 6                                            # node 5 has no in-edge -> max gives 0
    x = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32)).to(dtype)
    g = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32)).to(dtype)
    s, d = torch.from_numpy(src).to(DEV, torch.int32), torch.from_numpy(dst).to(DEV, torch.int32)
    got = ops.edge_dot(x.to(DEV), s, g.to(DEV), d)
    ref = (x.double()[src] * g.double()[dst]).sum(1)
    torch.testing.assert_close(got.cpu().double(), ref, rtol=1e-4, atol=1e-4)
    index = ops.EdgeIndex(s, d, N)
    xd = x.to(DEV).requires_grad_(True)
    out = ops.neighbor_max(xd, index)
    out.backward(g.to(DEV))
    xr = x.double().requires_grad_(True)
    want = OL.sage_conv(xr, torch.from_numpy(src), torch.from_numpy(dst), torch.eye(H, dtype=torch.float64), None,
                        torch.zeros(H, H, dtype=torch.float64), aggr="max")
    want.backward(g.double())
    assert float(out.detach()[5].abs().max()) == 0.0
    torch.testing.assert_close(out.detach().cpu().double(), want.detach(), rtol=0, atol=0)      # max is exact
    # ties (duplicate edges / equal bf16 values) may route the gradient to a different but equal-valued source; compare
    # the per-destination totals, which are unambiguous
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(xd.grad.cpu().double().sum(0), xr.grad.sum(0), rtol=tol, atol=tol * 10)
    if dtype == torch.float32:
        torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tiles_per_wg", [1, 2, 3, 4, 7, 8, 9, 17])
def test_ring_transform_at_every_short_pipeline_length(tiles_per_wg):
    """The persistent H = 256 transform (dn_rel_ring.hip) meets all twelve waves once per PAIR of tiles and runs eight stages
    ahead: tables that give a workgroup 1, 2, 3, ... tiles (shorter than the ring, odd counts, the last workgroup shorter than
    the others), with EMPTY tiles in the middle and at the end of a workgroup's range and a relation change at every tile."""
    ops = _ops()
    H = 256
    rng = np.random.default_rng(40 + tiles_per_wg)
    ntile = 256 * tiles_per_wg - (tiles_per_wg > 1) * 3           # dn_cdiv(ntile, 256) == tiles_per_wg, last workgroups short
    R, N1 = 7, 900
    recs, rows = [], 0
    for t in range(ntile):
        if t % 11 == 5 or t >= ntile - 2:
            recs.append((int(rng.integers(0, R)), rows, rows, 0))   # an empty tile: skipped, its relation not loaded
            continue
        n = 32 if t % 5 else int(rng.integers(1, 32))               # some partial tiles
        recs.append((t % R, rows, rows + n, 0))
        rows += n
    P = rows
    tiles = (torch.tensor(recs, dtype=torch.int32, device=DEV), ntile)
    rel_of_row = torch.cat([torch.full((e - b,), r) for r, b, e, _ in recs if e > b])
    X = torch.from_numpy(rng.standard_normal((N1, H)).astype(np.float32)).to(torch.bfloat16)
    Wn = torch.from_numpy((rng.standard_normal((R, H, H)) / np.sqrt(H)).astype(np.float32)).to(torch.bfloat16)
    idx = torch.from_numpy(rng.integers(0, N1, size=P)).to(torch.int32)
    got = ops.rows_transform(X.to(DEV), Wn.to(DEV), tiles, P, idx=idx.to(DEV))
    ref = torch.einsum("pk,pnk->pn", X.double()[idx.long()], Wn.double()[rel_of_row])
    torch.testing.assert_close(got.cpu().double(), ref, rtol=1e-2, atol=2e-2)
    again = ops.rows_transform(X.to(DEV), Wn.to(DEV), tiles, P, idx=idx.to(DEV))
    assert torch.equal(got, again)


@pytest.mark.parametrize("H", [64, 128])
def test_bf16_kernels_read_the_parameters_where_they_lie_at_the_default_widths(H):
    """w_kn at H = 64 / 128 (round 5): dn_rows_transform_bf16 and dn_rows_selfsum_bf16 given `weight` [R, in, out] / `loop_weight`
    [in, out] as the reference stores them (rgin.py:61-67) must produce, bit for bit, what they produce from the transposed copies
    -- the fragments are the same, only their way into the registers differs (relation changes inside a workgroup included)."""
    ops = _ops()
    rng = np.random.default_rng(40 + H)
    sizes = [0, 37, 1500, 1, 33, 64, 9000]
    rel_ptr = [0] + [int(v) for v in np.cumsum(sizes)]
    R, P, N = len(sizes), rel_ptr[-1], 700
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16).to(DEV)  # noqa: E731
    X, Wkn = bf(rng.standard_normal((N, H))), bf(rng.standard_normal((R, H, H)) / np.sqrt(H))       # Wkn[r] = [in = k][out = n]
    bias = bf(rng.standard_normal((R, H)))
    idx = torch.from_numpy(rng.integers(0, N, size=P)).to(torch.int32).to(DEV)
    tiles = ops.make_row_tiles(rel_ptr, DEV)
    Wnk = Wkn.transpose(1, 2).contiguous()
    for b, relu in ((None, False), (bias, True)):
        a = ops.rows_transform(X, Wkn, tiles, P, idx=idx, bias=b, relu=relu, w_kn=True)
        c = ops.rows_transform(X, Wnk, tiles, P, idx=idx, bias=b, relu=relu)
        assert torch.equal(a, c)
    ref = torch.einsum("pk,pkn->pn", X[idx.long()].double().cpu(),
                       Wkn.double().cpu()[torch.repeat_interleave(torch.arange(R), torch.tensor(sizes))])
    torch.testing.assert_close(ops.rows_transform(X, Wkn, tiles, P, idx=idx, w_kn=True).cpu().double(), ref, rtol=1e-2, atol=2e-2)
    # the slot kernel (self loop + bias + row sums)
    K = ops.SELFSUM_SLOTS
    lists = [np.append(rng.integers(0, P, size=c), P + v) for v, c in enumerate(rng.integers(0, K + 1, size=N))]
    ptr = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    lp, lr = torch.from_numpy(ptr).to(DEV).int(), torch.from_numpy(np.concatenate(lists)).to(DEV).int()
    slots, over = ops.build_slot_table(lp, lr, N, P)
    Y = bf(rng.standard_normal((P, H)))
    Wl = bf(rng.standard_normal((H, H)) / np.sqrt(H))                                               # [in][out]
    a = ops.rows_selfsum(X, Wl, bias[0], Y, None, slots, lists=(lp, lr, P, 0, 0, over), w_kn=True)
    c = ops.rows_selfsum(X, Wl.t().contiguous(), bias[0], Y, None, slots, lists=(lp, lr, P, 0, 0, over))
    assert torch.equal(a, c)


@pytest.mark.parametrize("H", [64, 128, 256])
def test_rows_selfsum_matches_reference(H):
    """dn_rows_selfsum_bf16: self-loop transform + bias + fixed-slot row sum (nodes with more rows than slots finished from
    their lists inside the launch) against fp64 on the same bf16 operands; slot tables built from ragged per-node lists by
    ops.build_slot_table."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(H + 1)
    N, P = 1000 + H // 64, 2500                       # N not a multiple of the 32-row tile
    x = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32)).to(torch.bfloat16)
    W = torch.from_numpy((rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float32)).to(torch.bfloat16)   # Wn [out][in]
    b = torch.from_numpy(rng.standard_normal(H).astype(np.float32)).to(torch.bfloat16)
    Y = torch.from_numpy(rng.standard_normal((P, H)).astype(np.float32)).to(torch.bfloat16)
    # ragged lists: most nodes 0-K rows, some up to 3K (overflow), every list ends with the node's self row id P + v
    K = ops.SELFSUM_SLOTS
    cnt = rng.integers(0, K + 1, size=N)
    cnt[rng.integers(0, N, size=40)] = rng.integers(K + 1, 3 * K, size=40)
    lists = [np.append(rng.integers(0, P, size=c), P + v) for v, c in enumerate(cnt)]
    ptr = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    rows = np.concatenate(lists)
    lp, lr = torch.from_numpy(ptr).to(DEV).int(), torch.from_numpy(rows).to(DEV).int()
    slots, over = ops.build_slot_table(lp, lr, N, P)
    sl = slots.cpu().numpy()
    assert slots.shape == (N, K) and int((sl[:, K - 1] == -2).sum()) == int((cnt > K).sum())
    assert np.array_equal(over.cpu().numpy()[:N] != 0, cnt > K)
    for v in (0, 1, int(np.argmax(cnt))):                       # kept rows in list order, -1 padded, -2 marks "walk the list"
        want = list(lists[v][:-1][:K - 1 if cnt[v] > K else K])
        assert list(sl[v][:len(want)]) == want and set(sl[v][len(want):]) <= {-1, -2}
    Yd = Y.to(DEV)
    limit = ops.OVERFLOW_INSIDE_MAX_ROWS
    try:
        for inside in (True, False):                  # the long lists finished inside the launch / by dn_overflow_rows_add_bf16
            ops.OVERFLOW_INSIDE_MAX_ROWS = 10 ** 9 if inside else 0
            for bias in (b.to(DEV), None):
                out = ops.rows_selfsum(x.to(DEV), W.to(DEV), bias, Yd, None, slots, lists=(lp, lr, P, 0, 0, over))
                ref = x.double() @ W.double().t() + (b.double() if bias is not None else 0.0)
                for v in range(N):
                    ref[v] += Y[lists[v][:-1]].double().sum(0)
                err = (out.cpu().double() - ref).abs().max() / ref.abs().max()
                assert float(err) < 1.2e-2, (inside, float(err))   # bf16 roundings: the self-loop tile, the output row, the overflow add
    finally:
        ops.OVERFLOW_INSIDE_MAX_ROWS = limit
    # a dropped row range (the relation a fold handles elsewhere) is left out by the table AND by the list walk
    d0, d1 = 700, 1100
    slots_d, over_d = ops.build_slot_table(lp, lr, N, P, drop=(d0, d1))
    out = ops.rows_selfsum(x.to(DEV), W.to(DEV), None, Yd, None, slots_d, lists=(lp, lr, P, d0, d1, over_d))
    ref = x.double() @ W.double().t()
    for v in range(N):
        keep = [r for r in lists[v][:-1] if not (d0 <= r < d1)]
        if keep:
            ref[v] += Y[keep].double().sum(0)
    assert float((out.cpu().double() - ref).abs().max() / ref.abs().max()) < 1.2e-2
    # no incoming rows at all: out = x W^T + b
    empty = torch.full((N, K), -1, dtype=torch.int32, device=DEV)
    out = ops.rows_selfsum(x.to(DEV), W.to(DEV), b.to(DEV), Yd[:0], None, empty)
    ref = x.double() @ W.double().t() + b.double()
    assert float((out.cpu().double() - ref).abs().max() / ref.abs().max()) < 6e-3


@pytest.mark.parametrize("H", [64, 128, 256])
def test_rows_wgrad_with_bit_mask_and_chain2(H):
    """ReLU masks as bit tensors: dn_rows_chain2_bf16 emits them (forward), dn_rows_wgrad_bf16(mask_a_bits) and the chain's
    mask0 / mask1 inputs consume them (backward) -- against fp64 math on the same bf16 operands and storage points."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(10 + H)
    N = 5000 + H // 64                                   # not a multiple of any tile size
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)  # noqa: E731
    x, g = bf(rng.standard_normal((N, H))), bf(rng.standard_normal((N, H)))
    W1, W2 = bf(rng.standard_normal((H, H)) / np.sqrt(H)), bf(rng.standard_normal((H, H)) / np.sqrt(H))
    b1, b2 = bf(rng.standard_normal(H) * 0.1), bf(rng.standard_normal(H) * 0.1)
    d = lambda t: t.to(DEV)  # noqa: E731
    h1, h2, bits1, bits2 = ops.rows_chain2(d(x), d(W1), d(b1), True, d(W2), d(b2), True, want_bits=True)
    r1 = torch.relu(x.double() @ W1.double().t() + b1.double())
    assert float((h1.cpu().double() - r1).abs().max() / r1.abs().max()) < 6e-3
    r2 = torch.relu(h1.cpu().double() @ W2.double().t() + b2.double())            # from the stored (bf16) hidden rows
    assert float((h2.cpu().double() - r2).abs().max() / r2.abs().max()) < 6e-3
    unpack = lambda b: ((b.cpu().unsqueeze(-1) >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(N, H).bool()  # noqa: E731
    assert torch.equal(unpack(bits1), h1.cpu() > 0) and torch.equal(unpack(bits2), h2.cpu() > 0)
    # backward: weight gradient of layer 2 with the outer mask from bits, then the input-gradient chain
    _, chunks = ops._dense_table(N, torch.device(DEV))
    gw2, cs2 = ops.rows_wgrad(d(g), h1, chunks, 1, out_dtype=torch.float32, colsum_of=1, mask_a_bits=bits2)
    gm = torch.where(h2.cpu() > 0, g, torch.zeros((), dtype=g.dtype)).double()
    ref_w2 = gm.t() @ h1.cpu().double()
    assert float((gw2[0].cpu().double() - ref_w2).abs().max() / ref_w2.abs().max()) < 2e-3
    assert float((cs2[0].cpu().double() - gm.sum(0)).abs().max() / gm.sum(0).abs().max()) < 2e-3
    g1, g0 = ops.rows_chain2(d(g), d(W2).t(), None, False, d(W1).t(), None, False, mask0_bits=bits2, mask1_bits=bits1)
    rg1 = (gm @ W2.double()) * (h1.cpu() > 0)
    assert float((g1.cpu().double() - rg1).abs().max() / rg1.abs().max()) < 6e-3
    rg0 = g1.cpu().double() @ W1.double()
    assert float((g0.cpu().double() - rg0).abs().max() / rg0.abs().max()) < 6e-3


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 95, 97, 1000, 8191, 70001])
def test_wgrad_ring_kernel_with_loader_waves_at_every_pipeline_length(rows):
    """rows_wgrad_ls_kernel (csrc/dn_rel.hip: four loader waves, eight MFMA waves; H = 256 bf16) from one row -- every look-ahead
    tile past the end -- to many tiles per workgroup with ragged ends, in its three forms: gathered operands with second sources
    (the conv's weight gradient; a relation's chunks taken as interleaved pieces, some of them EMPTY when a relation has fewer
    tiles than chunks, and a table with unused entries behind the last chunk), rows in order (the MLP's), rows in order with the
    A operand masked by bits.  Against fp64 sums of the same bf16 operands; column sums of every relation and of one relation
    only (dn_rows_wgrad_bf16's colsum_of | (relation + 1) << 8); run-to-run bitwise equal."""
    ops = _ops()
    H, R = 256, 4
    rng = np.random.default_rng(rows)
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)  # noqa: E731
    d = lambda t: t.to(DEV)  # noqa: E731
    sizes = [rows, 0, max(1, rows // 7), 3 * rows + 5]
    rel_ptr = [0] + [int(v) for v in np.cumsum(sizes)]
    P, NA, NG = rel_ptr[-1], 700, 900
    A, A2, G, G2 = bf(rng.standard_normal((NA, H))), bf(rng.standard_normal((300, H))), bf(rng.standard_normal((NG, H))), bf(rng.standard_normal((200, H)))
    ia = torch.from_numpy(rng.integers(0, NA + 300, size=P)).to(torch.int32)
    ig = torch.from_numpy(rng.integers(0, NG + 200, size=P)).to(torch.int32)
    Acat, Gcat = torch.cat([A, A2]).double(), torch.cat([G, G2]).double()
    ref = torch.stack([Acat[ia[a:b].long()].t() @ Gcat[ig[a:b].long()] for a, b in zip(rel_ptr[:-1], rel_ptr[1:])])
    cs_ref = torch.stack([Gcat[ig[a:b].long()].sum(0) for a, b in zip(rel_ptr[:-1], rel_ptr[1:])])
    scale = max(1.0, float(ref.abs().max()))
    # a device-built table (an upper bound of entries: empty pieces behind the last chunk) and a host-built one with small chunks
    rel_ptr_d = torch.tensor(rel_ptr, dtype=torch.int32, device=DEV)
    for table in (ops.build_row_tables(rel_ptr_d, R, P, 256, want_ptr=True), ops.make_row_chunks(rel_ptr, DEV, chunk_rows=64),
                  ops.make_row_chunks(rel_ptr, DEV, chunk_rows=1 << 20)):
        kw = dict(idx_a=d(ia), idx_g=d(ig), A2=d(A2), G2=d(G2), out_dtype=torch.float32)
        got, cs = ops.rows_wgrad(d(A), d(G), table, R, colsum_of=2, **kw)
        assert float((got.cpu().double() - ref).abs().max()) / scale < 1e-5
        assert float((cs.cpu().double() - cs_ref).abs().max()) < 1e-3 * max(1.0, float(cs_ref.abs().max()))
        got1, cs1 = ops.rows_wgrad(d(A), d(G), table, R, colsum_of=2, colsum_rel=3, **kw)
        assert torch.equal(got, got1) and torch.equal(cs1[3], cs[3]) and not cs1[:3].any()
        again, cs_again = ops.rows_wgrad(d(A), d(G), table, R, colsum_of=2, **kw)
        assert torch.equal(got, again) and torch.equal(cs, cs_again)
    # rows in order: plain, and with the A operand masked by bits (+ the masked operand's column sums)
    M = min(P, 650)
    bits = torch.from_numpy(rng.integers(0, 256, size=(M, H // 8)).astype(np.uint8))
    keep = ((bits.unsqueeze(-1) >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(M, H).bool()
    _, dense = ops._dense_table(M, torch.device(DEV))
    a, g = A[:M].contiguous(), G[:M].contiguous()
    got, cs = ops.rows_wgrad(d(a), d(g), dense, 1, out_dtype=torch.float32, colsum_of=1)
    ref2 = a.double().t() @ g.double()
    assert float((got[0].cpu().double() - ref2).abs().max()) / max(1.0, float(ref2.abs().max())) < 1e-5
    assert float((cs[0].cpu().double() - a.double().sum(0)).abs().max()) < 1e-3 * max(1.0, float(a.double().sum(0).abs().max()))
    for slope in (0.0, 0.25):
        am = torch.where(keep, a.double(), a.double() * slope)
        got, cs = ops.rows_wgrad(d(a), d(g), dense, 1, out_dtype=torch.float32, colsum_of=1, mask_a_bits=d(bits), slope=slope)
        am = torch.where(keep, a, (a.float() * slope).to(torch.bfloat16)).double()      # (the scaled rows are rounded to bf16 in the tile)
        ref3 = am.t() @ g.double()
        assert float((got[0].cpu().double() - ref3).abs().max()) / max(1.0, float(ref3.abs().max())) < 1e-5
        assert float((cs[0].cpu().double() - am.sum(0)).abs().max()) < 1e-3 * max(1.0, float(am.sum(0).abs().max()))


@pytest.mark.parametrize("H", [64, 128])
@pytest.mark.parametrize("N", [1, 31, 32, 33, 1000, 16385, 70001])
def test_chain2_f32_forward_and_backward_chain(H, N):
    """dn_rows_chain2_f32 (two dense layers in one pass over fp32 rows, 3-term split): the forward of the reference MLP
    (Linear-act-Linear-act, biases) and the backward's input-gradient chain (outer mask from the saved output, dgrad 2 on the weight as
    stored, inner mask from the saved hidden rows, dgrad 1) for ReLU and leaky ReLU, from one row to several tiles per workgroup with a
    ragged last tile -- against fp64 math at the goldens' tolerance; equal to the separate launches up to the split's rounding."""
    ops = _ops()
    rng = np.random.default_rng(N + H)
    f = lambda a: torch.from_numpy(a.astype(np.float32))  # noqa: E731
    d = lambda t: t.to(DEV)  # noqa: E731
    x, g = f(rng.standard_normal((N, H))), f(rng.standard_normal((N, H)))
    W1, W2 = f(rng.standard_normal((H, H)) / np.sqrt(H)), f(rng.standard_normal((H, H)) / np.sqrt(H))
    b1, b2 = f(rng.standard_normal(H) * 0.1), f(rng.standard_normal(H) * 0.1)
    for slope in (0.0, 1.0 / 5.5):
        act = lambda t: torch.where(t > 0, t, t * slope)  # noqa: E731
        h1, h2 = ops.rows_chain2_f32(d(x), d(W1), d(b1), True, d(W2), d(b2), True, slope=slope)
        r1 = act(x.double() @ W1.double().t() + b1.double())
        r2 = act(h1.cpu().double() @ W2.double().t() + b2.double())       # from the stored hidden rows
        assert float((h1.cpu().double() - r1).abs().max()) <= 1e-4 * max(1.0, float(r1.abs().max()))
        assert float((h2.cpu().double() - r2).abs().max()) <= 1e-4 * max(1.0, float(r2.abs().max()))
        tiles, _ = ops._dense_table(N, torch.device(DEV))
        s1 = ops.rows_transform(d(x), d(W1).unsqueeze(0), tiles, N, bias=d(b1).view(1, -1), relu=True, slope=slope)
        assert float((h1 - s1).abs().max()) <= 2e-5 * max(1.0, float(s1.abs().max()))
        # backward chain on the weights as stored ([out][in] read as [k][n])
        g1, g0 = ops.rows_chain2_f32(d(g), d(W2), None, False, d(W1), None, False, mask0=h2, mask1=h1, w_kn=(True, True), slope=slope)
        keep2, keep1 = (h2.cpu() > 0), (h1.cpu() > 0)
        gm = torch.where(keep2, g.double(), g.double() * slope)
        t = gm @ W2.double()
        rg1 = torch.where(keep1, t, t * slope)
        rg0 = g1.cpu().double() @ W1.double()
        assert float((g1.cpu().double() - rg1).abs().max()) <= 1e-4 * max(1.0, float(rg1.abs().max()))
        assert float((g0.cpu().double() - rg0).abs().max()) <= 1e-4 * max(1.0, float(rg0.abs().max()))
        again = ops.rows_chain2_f32(d(g), d(W2), None, False, d(W1), None, False, mask0=h2, mask1=h1, w_kn=(True, True), slope=slope)
        assert torch.equal(again[0], g1) and torch.equal(again[1], g0)


@pytest.mark.parametrize("N", [1, 31, 32, 33, 257, 8191, 8224, 300001])
@pytest.mark.parametrize("slope", [0.0, 1.0 / 5.5])
def test_chain2_ring_kernel_at_every_pipeline_length(N, slope):
    """dn_rows_chain2_bf16 at H = 256 (csrc/dn_chain2.hip: LDS-DMA ring, counted waits over loads AND stores, results from the
    accumulators): forward with and without the sign-bit outputs, the backward chain with both masks on the weights as stored,
    from one row (one workgroup, one tile, every look-ahead tile past the end) to many tiles per workgroup with a ragged last
    tile -- against fp64 math on the same bf16 operands and storage points; bits bit-exact; run-to-run bitwise equal."""
    from dummynode4graphlearning_amd import ops
    H = 256
    rng = np.random.default_rng(N)
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)  # noqa: E731
    x, g = bf(rng.standard_normal((N, H))), bf(rng.standard_normal((N, H)))
    W1, W2 = bf(rng.standard_normal((H, H)) / np.sqrt(H)), bf(rng.standard_normal((H, H)) / np.sqrt(H))
    b1, b2 = bf(rng.standard_normal(H) * 0.1), bf(rng.standard_normal(H) * 0.1)
    d = lambda t: t.to(DEV)  # noqa: E731
    act = lambda v: torch.where(v > 0, v, v * slope)  # noqa: E731
    rel = lambda a, r: float((a.cpu().double() - r).abs().max() / r.abs().max().clamp(min=1e-3))  # noqa: E731
    h1, h2, bits1, bits2 = ops.rows_chain2(d(x), d(W1), d(b1), True, d(W2), d(b2), True, want_bits=True, slope=slope)
    r1 = act(x.double() @ W1.double().t() + b1.double())
    assert rel(h1, r1) < 6e-3
    r2 = act(h1.cpu().double() @ W2.double().t() + b2.double())            # from the stored (bf16) hidden rows
    assert rel(h2, r2) < 6e-3
    unpack = lambda b: ((b.cpu().unsqueeze(-1) >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(N, H).bool()  # noqa: E731
    assert torch.equal(unpack(bits1), h1.cpu() > 0) and torch.equal(unpack(bits2), h2.cpu() > 0)
    # without the bit outputs, without bias, second layer linear: the same rows
    p1, p2 = ops.rows_chain2(d(x), d(W1), d(b1), True, d(W2), d(b2), True, slope=slope)
    assert torch.equal(p1, h1) and torch.equal(p2, h2)
    q1, q2 = ops.rows_chain2(d(x), d(W1), None, True, d(W2), None, False, slope=slope)
    s1 = act(x.double() @ W1.double().t())
    assert rel(q1, s1) < 6e-3 and rel(q2, q1.cpu().double() @ W2.double().t()) < 6e-3
    # the backward chain: outer mask, dgrad 2, inner mask, dgrad 1 -- weights [k][n] as the Linear stores them, and transposed copies
    keep2, keep1 = (h2.cpu() > 0), (h1.cpu() > 0)
    gm = torch.where(keep2, g.double(), g.double() * slope)
    rg1 = gm @ W2.double()
    rg1 = torch.where(keep1, rg1, rg1 * slope)
    g1, g0 = ops.rows_chain2(d(g), d(W2), None, False, d(W1), None, False, mask0_bits=bits2, mask1_bits=bits1, w_kn=(True, True),
                             slope=slope)
    assert rel(g1, rg1) < 6e-3
    assert rel(g0, g1.cpu().double() @ W1.double()) < 6e-3
    t1, t0 = ops.rows_chain2(d(g), d(W2).t().contiguous(), None, False, d(W1).t().contiguous(), None, False, mask0_bits=bits2,
                             mask1_bits=bits1, slope=slope)
    assert torch.equal(t1, g1) and torch.equal(t0, g0)
    again = ops.rows_chain2(d(g), d(W2), None, False, d(W1), None, False, mask0_bits=bits2, mask1_bits=bits1, w_kn=(True, True), slope=slope)
    assert torch.equal(again[0], g1) and torch.equal(again[1], g0)


def test_index_builds_reject_out_of_range_edge_types():
    """an edge type >= num_rels (easy to hit in the SI flow: the dummy labels extend the relation set) must raise,
    not be silently grouped into another relation."""
    ops = _ops()
    from dummynode4graphlearning_amd._lib import DnHipError
    src = torch.tensor([0, 1, 2], device=DEV)
    dst = torch.tensor([1, 2, 0], device=DEV)
    for bad in (torch.tensor([0, 3, 1], device=DEV), torch.tensor([0, -1, 1], device=DEV)):
        with pytest.raises(DnHipError, match="edge type out of"):
            ops.RowIndex(src, dst, bad, 3, 3)
        with pytest.raises(DnHipError, match="edge type out of"):
            ops.RelIndex(src, dst, bad, 3, 3)
    ops.RowIndex(src, dst, torch.tensor([0, 2, 1], device=DEV), 3, 3)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("K,N", [(8, 64), (38, 256), (32, 32), (100, 7), (64, 64)])
def test_any_width_grouped_products_match_per_relation_matmuls(K, N, dt):
    """dn_rows_gemm_* / dn_rows_wgrad_any_*: what PyG's RGCNConv does with a Python loop over relations (rgconv.py:17-18,96),
    as one launch each, for widths the matrix-core kernels do not take (first conv F -> H, H = 32, ...)."""
    ops = _ops()
    rng = np.random.default_rng(K * 1000 + N)
    sizes = [0, 37, 3000, 1, 700]
    rel_ptr = [0] + [int(v) for v in np.cumsum(sizes)]
    R, P = len(sizes), rel_ptr[-1]
    A = torch.from_numpy(rng.standard_normal((P, K)).astype(np.float32)).to(dt)
    G = torch.from_numpy(rng.standard_normal((P, N)).astype(np.float32)).to(dt)
    W = torch.from_numpy((rng.standard_normal((R, K, N)) / np.sqrt(K)).astype(np.float32)).to(dt)
    rp = torch.tensor(rel_ptr, dtype=torch.int32, device=DEV)
    tiles = ops.build_row_tables(rp, R, P, 64)
    chunks = ops.build_row_tables(rp, R, P, 512, want_ptr=True)
    tol = dict(rtol=2e-2, atol=3e-2) if dt == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    Y = ops.rows_gemm(A.to(DEV), W.to(DEV), tiles)
    Yt = ops.rows_gemm(G.to(DEV), W.to(DEV), tiles, transpose_w=True)
    gW = ops.rows_wgrad_any(A.to(DEV), G.to(DEV), chunks, R)
    for r in range(R):
        a, b = rel_ptr[r], rel_ptr[r + 1]
        torch.testing.assert_close(Y[a:b].cpu().double(), A[a:b].double() @ W[r].double(), **tol)
        torch.testing.assert_close(Yt[a:b].cpu().double(), G[a:b].double() @ W[r].double().t(), **tol)
        ref = A[a:b].double().t() @ G[a:b].double()
        wtol = dict(rtol=2e-2, atol=0.5) if dt == torch.bfloat16 else dict(rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(gW[r].cpu().double(), ref, **wtol)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C", [(2, 64), (63, 4), (5000, 128), (20181, 256), (300, 40), (129, 1024)])
def test_batch_norm_rows_matches_torch(N, C, dt):
    """dn_batchnorm_rows_* (training-mode BatchNorm over the nodes of a batch, gconv.py:187-194) against F.batch_norm in fp64:
    output, batch statistics, input / weight / bias gradients; a large column mean must not cancel the variance away."""
    ops = _ops()
    rng = np.random.default_rng(N + C)
    x = torch.from_numpy((rng.standard_normal((N, C)) * 2.0 + 30.0 * rng.standard_normal(C)).astype(np.float32)).to(dt)
    w = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(dt)
    b = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(dt)
    coef = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32)).to(dt)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y, mean, var = ops.batch_norm_rows(xd, wd, bd, 1e-5)
    y.backward(coef.to(DEV))
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.batch_norm(xr, None, None, wr, br, training=True, eps=1e-5)
    yr.backward(coef.double())
    tol = dict(rtol=2e-2, atol=6e-2) if dt == torch.bfloat16 else dict(rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(mean.cpu().double(), x.double().mean(0), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(var.cpu().double(), x.double().var(0, unbiased=False), rtol=2e-4, atol=1e-5)
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), **tol)
    if N > 1:
        gt = dict(rtol=3e-2, atol=0.15) if dt == torch.bfloat16 else dict(rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, **gt)
        torch.testing.assert_close(wd.grad.cpu().double(), wr.grad, rtol=3e-2 if dt == torch.bfloat16 else 1e-3,
                                   atol=(0.5 if dt == torch.bfloat16 else 1e-3) * max(1.0, N ** 0.5 / 10))
        torch.testing.assert_close(bd.grad.cpu().double(), br.grad, rtol=3e-2 if dt == torch.bfloat16 else 1e-4,
                                   atol=(0.5 if dt == torch.bfloat16 else 1e-3) * max(1.0, N ** 0.5 / 10))


@pytest.mark.parametrize("N,C", [(63, 4), (5000, 128), (20181, 256)])
def test_batch_norm_rows_with_fused_relu_and_running_statistics(N, C):
    """relu=True: y = ReLU(BatchNorm(x)) forward and backward (the mask is recomputed from x), and the running statistics updated
    by the statistics launch -- against torch's BatchNorm1d module + ReLU in fp64 over three steps."""
    ops = _ops()
    rng = np.random.default_rng(N * 7 + C)
    bn = torch.nn.BatchNorm1d(C, momentum=0.1).double()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.standard_normal(C)))
        bn.bias.copy_(torch.from_numpy(rng.standard_normal(C)))
    wd = bn.weight.detach().float().to(DEV).requires_grad_(True)
    bd = bn.bias.detach().float().to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    for step in range(3):
        x = torch.from_numpy((rng.standard_normal((N, C)) * 2.0 + 3.0).astype(np.float32))
        for _ in range(8):                          # no pre-activation within 1e-3 of the ReLU kink (fp32 / fp64 must agree on its side)
            xh = (x.double() - x.double().mean(0)) / (x.double().var(0, unbiased=False) + bn.eps).sqrt()
            near = (xh * bn.weight.detach() + bn.bias.detach()).abs() < 1e-3
            if not bool(near.any()):
                break
            x = torch.where(near, x + 0.05, x)
        assert not bool(near.any())
        coef = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32))
        xd = x.to(DEV).requires_grad_(True)
        wd.grad = bd.grad = None
        y, _, _ = ops.batch_norm_rows(xd, wd, bd, bn.eps, rm, rv, 0.1, relu=True)
        y.backward(coef.to(DEV))
        xr = x.double().requires_grad_(True)
        bn.zero_grad()
        yr = torch.relu(bn(xr))
        yr.backward(coef.double())
        torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), rtol=2e-4, atol=2e-4)
        assert float(y.min()) >= 0.0
        torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(wd.grad.cpu().double(), bn.weight.grad, rtol=1e-3, atol=1e-3 * max(1.0, N ** 0.5 / 10))
        torch.testing.assert_close(bd.grad.cpu().double(), bn.bias.grad, rtol=1e-3, atol=1e-3 * max(1.0, N ** 0.5 / 10))
        torch.testing.assert_close(rm.cpu().double(), bn.running_mean, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(rv.cpu().double(), bn.running_var, rtol=1e-4, atol=1e-5)
    # the module: BatchNorm1d's num_batches_tracked rides in the statistics launch (no launch of its own), state_dict parity
    from dummynode4graphlearning_amd.graph_classification.models import HipBatchNorm1d
    ref = torch.nn.BatchNorm1d(C)
    mod = HipBatchNorm1d(C, fuse_relu=False).to(DEV)
    mod.load_state_dict(ref.state_dict())
    xs = torch.from_numpy(rng.standard_normal((max(N, 2), C)).astype(np.float32))
    for _ in range(3):
        mod(xs.to(DEV))
        ref(xs)
    assert int(mod.num_batches_tracked) == int(ref.num_batches_tracked) == 3
    torch.testing.assert_close(mod.running_mean.cpu(), ref.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(mod.running_var.cpu(), ref.running_var, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("K,N_", [(5, 128), (38, 256), (128, 128), (64, 2), (256, 256)])
def test_linear_any_matches_torch(K, N_, dt):
    ops = _ops()
    rng = np.random.default_rng(K + N_)
    P = 3001
    x = torch.from_numpy(rng.standard_normal((P, K)).astype(np.float32)).to(dt)
    w = torch.from_numpy((rng.standard_normal((N_, K)) / np.sqrt(K)).astype(np.float32)).to(dt)
    b = torch.from_numpy(rng.standard_normal(N_).astype(np.float32)).to(dt)
    coef = torch.from_numpy(rng.standard_normal((P, N_)).astype(np.float32)).to(dt)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.linear_any(xd, wd, bd)
    y.backward(coef.to(DEV))
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(coef.double())
    tol = dict(rtol=2e-2, atol=5e-2) if dt == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), **tol)
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, **tol)
    torch.testing.assert_close(wd.grad.cpu().double(), wr.grad, rtol=2e-2 if dt == torch.bfloat16 else 1e-4, atol=1.0 if dt == torch.bfloat16 else 2e-2)
    torch.testing.assert_close(bd.grad.cpu().double(), br.grad, rtol=2e-2 if dt == torch.bfloat16 else 1e-4, atol=1.0 if dt == torch.bfloat16 else 2e-2)


def _dummy_batch(rng, sizes, R, deg=2):
    """Graphs of the given sizes, each followed by its dummy node (edges u -> dummy of type R, dummy -> u of type R + 1): the
    layout dn_dummy_augment produces (dummy node last in its graph).  Ordinary edges are permutations of a graph's nodes, so
    the ordinary relations have (almost) no shared endpoints and stay one-row-per-edge."""
    src, dst, et, base = [], [], [], 0
    for n in sizes:
        for _ in range(deg):
            src += list(base + np.arange(n)); dst += list(base + rng.permutation(n)); et += [int(rng.integers(0, R))] * n
        d = base + n
        src += list(range(base, d)) + [d] * n; dst += [d] * n + list(range(base, d)); et += [R] * n + [R + 1] * n
        base = d + 1
    return np.array(src), np.array(dst), np.array(et), base


@pytest.mark.parametrize("H", [64, 128, 256])
def test_folded_pre_aggregation_matches_the_separate_pass(H):
    """The closing launch's per-graph column sums (dn_rows_selfsum_bf16 with seg_of_node + combine / transform / add tail) against
    (a) the path with a separate dn_gather_segsum pass and (b) fp64 per-edge math; run twice: bitwise reproducible.  Graph sizes
    straddle the 32-row tiles in every way (1-node graphs, graphs longer than two tiles, a boundary exactly at a tile edge)."""
    ops = _ops()
    rng = np.random.default_rng(H)
    sizes = [31, 1, 1, 29, 70, 3, 31, 64, 2, 127] + list(rng.integers(1, 60, size=40))
    R = 5
    src, dst, et, N = _dummy_batch(rng, sizes, R)
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)  # noqa: E731
    x, coef = bf(rng.standard_normal((N, H))), bf(rng.standard_normal((N, H)))
    W = bf(rng.standard_normal((R + 3, H, H)) / np.sqrt(H))
    b = bf(rng.standard_normal(H))
    s, d, t = (torch.from_numpy(a).to(DEV) for a in (src, dst, et))

    def run(fold):
        old = ops.FOLD_ENABLED
        ops.FOLD_ENABLED = fold
        try:
            index = ops.RowIndex(s, d, t, N, R + 2, self_loop=True)
            xd, Wd, bd = (v.to(DEV).requires_grad_(True) for v in (x, W, b))
            out = ops.rel_transform_fused(xd, Wd, bd, index)
            out.backward(coef.to(DEV))
            folded = tuple(ops._row_index_fold(index, k) is not None for k in "fb")
            return index, folded, [v.detach().cpu() for v in (out, xd.grad, Wd.grad, bd.grad)]
        finally:
            ops.FOLD_ENABLED = old

    index, folded, got = run(True)
    assert folded == (True, True)
    assert index.modes[R] == ops.RowIndex.AGG and index.modes[R + 1] == ops.RowIndex.TF
    _, folded0, plain = run(False)
    assert folded0 == (False, False)
    _, _, again = run(True)
    for a, c in zip(got, again):
        assert torch.equal(a, c)
    xr, Wr, br = (v.double().requires_grad_(True) for v in (x, W, b))
    st, dt_, tt = (torch.from_numpy(a) for a in (src, dst, et))
    ref = torch.zeros(N, H, dtype=torch.float64).index_add(0, dt_, torch.bmm(xr[st].unsqueeze(1), Wr[tt]).squeeze(1))
    ref = ref + xr @ Wr[R + 2] + br
    ref.backward(coef.double())
    for name, a, p, r in zip(("out", "gx", "gW", "gb"), got, plain, (ref, xr.grad, Wr.grad, br.grad)):
        scale = float(r.abs().max())
        e_ref, e_plain = float((a.double() - r.detach()).abs().max()) / scale, float((p.double() - r.detach()).abs().max()) / scale
        assert e_ref < 2e-2, (name, e_ref)
        assert e_ref < 2.0 * e_plain + 1e-3, (name, e_ref, e_plain)          # as accurate as the path it replaces


def test_fold_is_refused_when_a_graph_is_not_a_contiguous_node_range():
    """Interleaved node numbering (the dummy node's sources are every other node): the separate pass stays."""
    ops = _ops()
    rng = np.random.default_rng(3)
    N, H, R = 400, 64, 2
    src, dst, et = list(rng.integers(0, N - 2, size=900)), list(rng.integers(0, N - 2, size=900)), list(rng.integers(0, R, size=900))
    for dummy, members in ((N - 2, range(0, N - 2, 2)), (N - 1, range(1, N - 2, 2))):
        src += list(members); dst += [dummy] * len(members); et += [R] * len(members)
    s, d, t = (torch.tensor(a, device=DEV) for a in (src, dst, et))
    index = ops.RowIndex(s, d, t, N, R + 1, self_loop=True)
    assert index.modes[R] == ops.RowIndex.AGG
    assert ops._row_index_fold(index, "f") is None
    x = torch.randn(N, H, device=DEV).to(torch.bfloat16)
    W = (torch.randn(R + 2, H, H, device=DEV) / 8).to(torch.bfloat16)
    out = ops.rel_transform_fused(x, W, None, index)
    ref = torch.zeros(N, H, dtype=torch.float64).index_add(
        0, torch.tensor(dst), torch.bmm(x.cpu().double()[torch.tensor(src)].unsqueeze(1), W.cpu().double()[torch.tensor(et)]).squeeze(1))
    ref = ref + x.cpu().double() @ W.cpu().double()[R + 1]
    assert float((out.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-2


@pytest.mark.parametrize("seed,R,W,skew,G", [(0, 16, 32, False, 3000), (1, 5, 32, True, 3000), (2, 40, 16, False, 3000),
                                             (3, 3, 64, True, 3000), (4, 17, 32, False, 3000), (5, 6, 16, False, 40000),
                                             (6, 16, 32, True, 40000)])
def test_sweep_tile_tables_match_the_host_restatement(seed, R, W, skew, G):
    """dn_sweep_tables_build_i32 (L2-blocked tile order of the persistent transform launch) against tests/sweep_ref.py: the same
    table bit for bit; every row of every kept relation covered exactly once; keys that are not monotone (any cut is valid);
    a table too small for a group falls back to the plain order (still a partition).  Both share rules: relations with workgroups
    of their own (64 or more tiles per workgroup: the 40,000-graph cases) and the plain cut of the group's tile line (fewer)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sweep_ref import check_partition, sweep_tables
    ops = _ops()
    rng = np.random.default_rng(100 + seed)
    sizes = rng.integers(1, 40, size=G)
    node_ptr = np.concatenate([[0], np.cumsum(sizes)])
    N = int(node_ptr[-1])
    rel_ptr, keys, rin = [0], [], []
    skip = 1 << int(rng.integers(0, R))
    for r in range(R):
        lam = (8.0 if (skew and r == 0) else 1.0) * rng.uniform(0.2, 2.5)
        cnt = rng.poisson(lam, G) if r % 7 != 6 else np.where(np.arange(G) % 50 == 0, 3, 0)       # a sparse relation too
        k = np.repeat(node_ptr[:-1], cnt) + rng.integers(0, np.repeat(sizes, cnt))
        k = np.sort(k)
        keys.append(k)
        rin.append(rng.integers(0, N, size=k.size))
        rel_ptr.append(rel_ptr[-1] + k.size)
    key = np.concatenate(keys).astype(np.int64)
    P = int(rel_ptr[-1])
    # the kernel's key: row_out when < N, else row_in -- give half the relations their key through row_in (row_out = N + something)
    row_out, row_in = key.copy(), np.concatenate(rin).astype(np.int64)
    for r in range(0, R, 2):
        a, b = rel_ptr[r], rel_ptr[r + 1]
        row_in[a:b] = key[a:b]
        row_out[a:b] = N + np.arange(b - a)
    rp = torch.tensor(rel_ptr, dtype=torch.int32, device=DEV)
    ri, ro = (torch.from_numpy(v.astype(np.int32)).to(DEV) for v in (row_in, row_out))
    (table, ntiles), info = ops.build_sweep_tables(rp, R, ri, ro, N, P, skip_mask=skip, wg_per_group=W, want_info=True)
    S = ntiles // (8 * W)
    plain, smax = (int(v) for v in info.cpu())
    assert plain == 0 and smax <= S
    from sweep_ref import PURE_MIN_S
    assert (smax >= PURE_MIN_S) == (G > 3000)                           # (which share rule the case exercises)
    got = table.cpu().numpy().reshape(8 * W, S, 4)
    want, _ = sweep_tables(rel_ptr, key, N, W, skip_mask=skip, s_cap=S)
    assert np.array_equal(got, want)
    check_partition(got, np.asarray(rel_ptr), skip)
    # non-monotone keys: still a partition
    ro2 = ro.clone()
    a, b = rel_ptr[1], rel_ptr[2]
    if b - a > 10 and 1 % 2 == 1:
        ro2[a:b] = ro2[a:b].flip(0)
    (t2, n2), _ = ops.build_sweep_tables(rp, R, ri, ro2, N, P, skip_mask=skip, wg_per_group=W, want_info=True)
    check_partition(t2.cpu().numpy().reshape(8 * W, -1, 4), np.asarray(rel_ptr), skip)
    # a table with too few slots per workgroup for the largest group: the plain order, flagged
    tiny = torch.empty((8 * W * (smax - 1), 4), dtype=torch.int32, device=DEV)
    info2 = torch.zeros(2, dtype=torch.int32, device=DEV)
    from dummynode4graphlearning_amd._lib import check, lib, ptr, stream_ptr
    if 8 * W * (smax - 1) >= P // 32 + R:
        check(lib().dn_sweep_tables_build_i32(R, ptr(rp), ptr(ri), ptr(ro), N, W, smax - 1, skip, ptr(tiny), ptr(info2), stream_ptr()), "sweep")
        assert int(info2[0]) == 1
        check_partition(tiny.cpu().numpy().reshape(8 * W, -1, 4), np.asarray(rel_ptr), skip)
