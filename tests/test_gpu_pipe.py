"""GPU parity of the persistent message-pass launch (dn_rows_pipe_bf16: transformed rows hand over through the XCD's L2)
against the two-launch path it replaces (dn_rows_transform_bf16 + dn_rows_selfsum_bf16) and against fp64 math on the same
bf16 operands; both directions, uniform and ragged batches, run-to-run bitwise reproducibility, and the abort -> fallback path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def _batch(rng, G, R, n_lo, n_hi, dummy=True):
    """Ragged graphs; with dummy=True every graph's last node is connected both ways to all others (types R-2, R-1), which
    the row index collapses into AGG / TF rows."""
    node_ptr, src, dst, et = [0], [], [], []
    for _ in range(G):
        n = int(rng.integers(n_lo, n_hi + 1))
        m = int(rng.integers(0, 3 * n + 1))
        base = node_ptr[-1]
        nr = n - 1 if dummy and n > 1 else n
        s, d = rng.integers(0, nr, size=m), rng.integers(0, nr, size=m)
        t = rng.integers(0, R - 2 if dummy else R, size=m)
        if dummy and n > 1:
            real = np.arange(n - 1)
            s = np.concatenate([s, real, np.full(n - 1, n - 1)])
            d = np.concatenate([d, np.full(n - 1, n - 1), real])
            t = np.concatenate([t, np.full(n - 1, R - 2), np.full(n - 1, R - 1)])
        src.extend((base + s).tolist()), dst.extend((base + d).tolist()), et.extend(t.tolist())
        node_ptr.append(base + n)
    return (torch.tensor(node_ptr, dtype=torch.int32), torch.tensor(src, dtype=torch.int64), torch.tensor(dst, dtype=torch.int64),
            torch.tensor(et, dtype=torch.int64))


def _index_with_pipe(src, dst, et, N, R, node_ptr):
    """The persistent launch is opt-in (DN_PIPE=1): build the index with it switched on."""
    from dummynode4graphlearning_amd import ops
    old = ops.PIPE_ENABLED
    ops.PIPE_ENABLED = True
    try:
        return ops.RowIndexSet(src.to(DEV), dst.to(DEV), et.to(DEV), N, R, True, node_ptr=node_ptr.to(DEV), edge_ptr=None)
    finally:
        ops.PIPE_ENABLED = old


def _run(iset, x, W_all, bias, use_pipe):
    from dummynode4graphlearning_amd import ops
    ix = iset.parts[0][2]
    old = ops.PIPE_ENABLED
    ops.PIPE_ENABLED = use_pipe
    try:
        N, H = x.shape
        Wn = W_all.transpose(1, 2).contiguous()
        ybuf = iset.ybuf(H, x.dtype, x.device)
        out_f, out_b = torch.empty_like(x), torch.empty_like(x)
        ops.message_pass(x, Wn, bias, ix, "f", ybuf, out_f)
        ops.message_pass(x, W_all.contiguous(), None, ix, "b", ybuf, out_b)
        torch.cuda.synchronize()
        return out_f, out_b
    finally:
        ops.PIPE_ENABLED = old


@pytest.mark.parametrize("H", [256, 128, 64])
@pytest.mark.parametrize("shape", ["uniform31", "ragged", "tiny"])
def test_pipe_matches_two_launch_path_and_fp64(H, shape):
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng({"uniform31": 1, "ragged": 2, "tiny": 3}[shape] + H)
    R = 8
    if shape == "uniform31":
        node_ptr, src, dst, et = _batch(rng, 700, R, 31, 31)
    elif shape == "ragged":
        node_ptr, src, dst, et = _batch(rng, 300, R, 2, 90)
    else:
        node_ptr, src, dst, et = _batch(rng, 5, R, 1, 6)
    N = int(node_ptr[-1])
    iset = _index_with_pipe(src, dst, et, N, R, node_ptr)
    ix = iset.parts[0][2]
    assert getattr(ix, "pipe", None) is not None
    gen = torch.Generator().manual_seed(H)
    x = torch.randn(N, H, generator=gen).to(torch.bfloat16).to(DEV)
    W_all = (torch.randn(R + 1, H, H, generator=gen) / H ** 0.5).to(torch.bfloat16).to(DEV)
    bias = torch.randn(H, generator=gen).to(torch.bfloat16).to(DEV)
    pf, pb = _run(iset, x, W_all, bias, True)
    assert not ix.pipe.disabled and ix.pipe.aborted() == 0
    of, ob = _run(iset, x, W_all, bias, False)
    # fp64 on the same bf16 operands
    xd, Wd = x.double().cpu(), W_all.double().cpu()
    want_f = xd @ Wd[R] + bias.double().cpu()
    want_f.index_add_(0, dst, torch.bmm(xd[src].unsqueeze(1), Wd[et]).squeeze(1))
    want_b = xd @ Wd[R].t()
    want_b.index_add_(0, src, torch.bmm(xd[dst].unsqueeze(1), Wd[et].transpose(1, 2)).squeeze(1))
    for got, old, want in ((pf, of, want_f), (pb, ob, want_b)):
        assert bool(torch.isfinite(got.float()).all())
        assert _rel_l2(got, want) < 6e-3, _rel_l2(got, want)          # bf16 storage of the products and of the output
        assert _rel_l2(got, old) < 6e-3


def test_pipe_is_bitwise_reproducible_and_used_by_the_layer():
    from dummynode4graphlearning_amd import BatchedGraph, ops, synthetic, transforms
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    raw = synthetic.config5(graphs=2048)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(DEV) for k in keys), raw["max_nv"], raw["max_nvl"],
                                      raw["max_ne"], raw["max_nel"])
    N, H, R = int(aug["node_label"].numel()), 256, raw["num_rels"]
    bnn = (aug["node_ptr"][1:] - aug["node_ptr"][:-1]).long()
    bne = (aug["edge_ptr"][1:] - aug["edge_ptr"][:-1]).long()
    g = BatchedGraph(aug["src"], aug["dst"], N, bnn, bne)
    et = aug["edge_label"].long()
    torch.manual_seed(5)
    layer = RGINLayer(H, H, num_rels=R, regularizer="basis", num_bases=-1, num_mlp_layers=2, act_func="relu").to(DEV).to(torch.bfloat16)
    gen = torch.Generator(device=DEV).manual_seed(9)
    x0 = torch.randn(N, H, device=DEV, generator=gen).to(torch.bfloat16)
    coef = torch.randn(N, H, device=DEV, generator=gen).to(torch.bfloat16)
    calls = []
    orig, old_flag = ops.rows_pipe, ops.PIPE_ENABLED
    ops.rows_pipe = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    ops.PIPE_ENABLED = True
    runs = []
    try:
        for _ in range(3):
            for p in layer.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            out, _ = layer(g, x, et)
            out.backward(coef)
            runs.append([out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters()])
    finally:
        ops.rows_pipe, ops.PIPE_ENABLED = orig, old_flag
    assert len(calls) == 6, "the layer did not take the persistent launch in both directions"
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)


def test_pipe_abort_falls_back():
    """A launch whose hand-offs cannot complete (a batch table that asks for more T signals than exist) must time out, raise
    the abort word, and the caller must transparently recompute on the two-launch path."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(0)
    node_ptr, src, dst, et = _batch(rng, 64, 6, 8, 20)
    N, H, R = int(node_ptr[-1]), 64, 6
    iset = _index_with_pipe(src, dst, et, N, R, node_ptr)
    ix = iset.parts[0][2]
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, generator=gen).to(torch.bfloat16).to(DEV)
    W_all = (torch.randn(R + 1, H, H, generator=gen) / 8).to(torch.bfloat16).to(DEV)
    good_f, _ = _run(iset, x, W_all, None, True)
    assert not ix.pipe.disabled
    ix.pipe.batches[:, 4] += 1                       # nobody will ever deliver the extra signal
    ix.pipe.checks_left = {"f": 2, "b": 2}
    old_to = ops.PIPE_TIMEOUT_MS
    ops.PIPE_TIMEOUT_MS = 5
    try:
        bad_f, _ = _run(iset, x, W_all, None, True)
    finally:
        ops.PIPE_TIMEOUT_MS = old_to
    assert ix.pipe.disabled
    assert _rel_l2(bad_f, good_f) < 6e-3             # recomputed by the fallback
