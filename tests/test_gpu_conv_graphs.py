"""dn_conv_graphs_bf16 (csrc/dn_conv_graph.hip): one launch per direction of the relational conv on batches of small graphs at the
reference's default width H = 64 (subgraph_isomorphism/config.py:456-461; BASELINE config 3) -- against fp64 math on the same bf16
operands, against the row-factorised launches it replaces, and bitwise against itself."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(rng, G, R, nmax, dens, dummy=True, empty=False):
    """graphs of 1 .. nmax real nodes (+ the SI dummy node with relations R-2 / R-1), multi-edges and self loops included"""
    src, dst, et, nptr, eptr = [], [], [], [0], [0]
    for g in range(G):
        n = int(rng.integers(0 if empty else 1, nmax + 1))
        base = nptr[-1]
        m = int(rng.integers(0, max(1, int(dens * n)) + 1)) if n else 0
        if m:
            src += list(base + rng.integers(0, n, size=m)); dst += list(base + rng.integers(0, n, size=m))
            et += list(rng.integers(0, max(1, R - 2), size=m))
        if dummy and n:
            src += list(base + np.arange(n)) + [base + n] * n
            dst += [base + n] * n + list(base + np.arange(n))
            et += [R - 2] * n + [R - 1] * n
            n += 1
        nptr.append(base + n); eptr.append(len(src))
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.int64), device=DEV)  # noqa: E731
    return t(src), t(dst), t(et), nptr, eptr


def _rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


@pytest.mark.parametrize("R,nmax,dens,G", [(8, 49, 2.1, 64), (16, 63, 3.0, 40), (3, 20, 6.0, 300), (5, 63, 12.0, 7), (1, 5, 1.0, 1)])
def test_conv_graphs_matches_fp64_in_both_directions(R, nmax, dens, G):
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(R * 100 + G)
    src, dst, et, nptr, eptr = _batch(rng, G, R, nmax, dens, dummy=R >= 3, empty=G > 100)
    N, H = nptr[-1], 64
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)  # noqa: E731
    ix = ops.RowIndex(src, dst, et, N, R, self_loop=True, node_ptr=i32(nptr), edge_ptr=i32(eptr))
    assert ix.built_by == "local" and ix.max_graph == (max(np.diff(nptr)), max(np.diff(eptr)))
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N, H, device=DEV, generator=gen).to(torch.bfloat16)
    W = (torch.randn(R, H, H, device=DEV, generator=gen) / 8).to(torch.bfloat16)
    Wl = (torch.randn(H, H, device=DEV, generator=gen) / 8).to(torch.bfloat16)
    b = torch.randn(H, device=DEV, generator=gen).to(torch.bfloat16)
    for direction, kn, bias in (("f", True, b), ("b", False, None), ("f", False, None)):
        pw = ops.PassWeights(W, Wl, kn=kn)
        assert ops.conv_graphs_ok(x, pw, ix)
        out = torch.full((N, H), float("nan"), dtype=torch.bfloat16, device=DEV)
        aux = ops.conv_graphs(x, pw, bias, ix, direction, out)
        out2 = torch.empty_like(out)
        aux2 = ops.conv_graphs(x, pw, bias, ix, direction, out2)
        assert int(ix._cg_err[0].item()) == 0
        assert torch.equal(out, out2) and (aux is None or torch.equal(aux, aux2))         # bitwise run to run
        a, o = (src, dst) if direction == "f" else (dst, src)
        Wd = W.double() if kn else W.double().transpose(1, 2)                            # kn False: the same memory read as [n][k]
        Wld = Wl.double() if kn else Wl.double().t()
        ref = x.double() @ Wld + (bias.double() if bias is not None else 0.0)
        if src.numel():
            msg = torch.bmm(x.double()[a].unsqueeze(1), Wd[et]).squeeze(1)
            ref.index_add_(0, o, msg)
        assert _rel_l2(out, ref) < 6e-3, (direction, kn, _rel_l2(out, ref))
        n_aux = ix.num_aux_f if direction == "f" else ix.num_aux_b
        if n_aux:
            ap, ai = (ix.aux_f_ptr, ix.aux_f_idx) if direction == "f" else (ix.aux_b_ptr, ix.aux_b_idx)
            want = ops.gather_segsum(x, ai, ap, n_aux)
            assert aux is not None and _rel_l2(aux, want) < 4e-3


@pytest.mark.parametrize("R,nmax,dens,G,slope", [(8, 49, 2.1, 64, 0.0), (16, 63, 3.0, 30, 1 / 5.5), (3, 20, 5.0, 200, 0.0)])
def test_layer_launches_match_fp64_stage_by_stage(R, nmax, dens, G, slope):
    """dn_layer_graphs_fwd_bf16 / _bwd_bf16: every stage against fp64 math on the launch's own stored operands (conv rows, layer-1 rows,
    their sign bits; masked gradient, dgrad 2, dgrad 1, the conv's input gradient, the column sums the weight gradient takes)."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(R + G)
    src, dst, et, nptr, eptr = _batch(rng, G, R, nmax, dens)
    N, H = nptr[-1], 64
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)  # noqa: E731
    ix = ops.RowIndex(src, dst, et, N, R, self_loop=True, node_ptr=i32(nptr), edge_ptr=i32(eptr))
    gen = torch.Generator(device=DEV).manual_seed(3)
    rnd = lambda *sh, sc=1.0: (torch.randn(*sh, device=DEV, generator=gen) * sc).to(torch.bfloat16)  # noqa: E731
    x, W, Wl, b = rnd(N, H), rnd(R, H, H, sc=1 / 8), rnd(H, H, sc=1 / 8), rnd(H)
    w1, b1, w2, b2 = rnd(H, H, sc=1 / 8), rnd(H), rnd(H, H, sc=1 / 8), rnd(H)
    act = lambda v: torch.where(v > 0, v, v * slope)  # noqa: E731
    h, h1, h2, bits1, bits2, aux = ops.layer_graphs_fwd(x, W, Wl, b, w1, b1, w2, b2, slope, ix)
    assert int(ix._cg_err[0].item()) == 0
    ref = x.double() @ Wl.double() + b.double()
    ref.index_add_(0, dst, torch.bmm(x.double()[src].unsqueeze(1), W.double()[et]).squeeze(1))
    assert _rel_l2(h, ref) < 6e-3
    assert _rel_l2(h1, act(h.double() @ w1.double().t() + b1.double())) < 4e-3
    assert _rel_l2(h2, act(h1.double() @ w2.double().t() + b2.double())) < 4e-3
    unpack = lambda bits: ((bits.unsqueeze(-1) >> torch.arange(8, device=DEV, dtype=torch.uint8)) & 1).reshape(N, H).bool()  # noqa: E731
    assert torch.equal(unpack(bits1), h1 > 0) and torch.equal(unpack(bits2), h2 > 0)
    assert _rel_l2(aux, ops.gather_segsum(x, ix.aux_f_idx, ix.aux_f_ptr, ix.num_aux_f)) < 4e-3
    g = rnd(N, H)
    g1, g0, gx, aux_b = ops.layer_graphs_bwd(g, W, Wl, w1, w2, slope, bits1, bits2, ix)
    assert int(ix._cg_err[0].item()) == 0
    gm = torch.where(unpack(bits2), g.double(), (g.double() * slope).to(torch.bfloat16).double())
    t = gm @ w2.double()
    assert _rel_l2(g1, torch.where(unpack(bits1), t, t * slope)) < 4e-3
    assert _rel_l2(g0, g1.double() @ w1.double()) < 4e-3
    refx = g0.double() @ Wl.double().t()
    refx.index_add_(0, src, torch.bmm(g0.double()[dst].unsqueeze(1), W.double().transpose(1, 2)[et]).squeeze(1))
    assert _rel_l2(gx, refx) < 6e-3
    assert _rel_l2(aux_b, ops.gather_segsum(g0, ix.aux_b_idx, ix.aux_b_ptr, ix.num_aux_b)) < 4e-3
    again = ops.layer_graphs_bwd(g, W, Wl, w1, w2, slope, bits1, bits2, ix)
    assert all(torch.equal(u, v) for u, v in zip((g1, g0, gx, aux_b), again))


def test_rgin_layer_at_the_default_width_takes_one_launch_per_direction():
    """RGINLayer(64, 64) bf16 on a config-3-shaped batch: the conv is ONE launch per direction (no transform / closing / tail
    launches), the step equals the row-factorised launches up to bf16 rounding and fp64 within bf16 noise, bitwise repeatable."""
    from dummynode4graphlearning_amd import BatchedGraph, ops, synthetic, transforms
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    import oracle.layers as OL
    raw = synthetic.config3(seed=3, graphs=96)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(DEV) for k in keys), raw["max_nv"], raw["max_nvl"], raw["max_ne"],
                                      raw["max_nel"])
    N, R, H = int(aug["node_label"].numel()), 8, 64
    bnn = (aug["node_ptr"][1:] - aug["node_ptr"][:-1]).long()
    bne = (aug["edge_ptr"][1:] - aug["edge_ptr"][:-1]).long()
    et = aug["edge_label"].long()
    torch.manual_seed(5)
    layer = RGINLayer(H, H, num_rels=R, regularizer="basis", num_bases=-1, num_mlp_layers=2, act_func="relu").to(DEV).to(torch.bfloat16)
    gen = torch.Generator(device=DEV).manual_seed(2)
    x0 = torch.randn(N, H, device=DEV, generator=gen).to(torch.bfloat16)
    coef = torch.randn(N, H, device=DEV, generator=gen).to(torch.bfloat16)

    def run(enabled):
        old = ops.CONV_GRAPHS_ENABLED
        ops.CONV_GRAPHS_ENABLED = enabled
        try:
            g = BatchedGraph(aug["src"], aug["dst"], N, bnn, bne, node_ptr=aug["node_ptr"], edge_ptr=aug["edge_ptr"])
            for p in layer.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            timer = ops.KernelTimer()
            ops.kernel_timer = timer
            try:
                out, _ = layer(g, x, et)
                out.backward(coef)
            finally:
                ops.kernel_timer = None
            return [r[0] for r in timer.records], [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
        finally:
            ops.CONV_GRAPHS_ENABLED = old

    tags, got = run(True)
    # the whole step in THREE library calls: the layer forward, its input gradients, ONE weight-gradient launch (+ its reduce)
    assert tags == ["layer_graphs_fwd", "layer_graphs_bwd", "rows_wgrad_multi"], tags
    old_lg, old_small = ops.LAYER_GRAPHS_ENABLED, ops.LAYER_SMALL_ENABLED
    try:
        ops.LAYER_GRAPHS_ENABLED = False                  # the conv launch + the separate MLP chain launches inside the same function:
        tags2, two = run(True)                            # the same arithmetic up to the activation's rounding point
        ops.LAYER_SMALL_ENABLED = False                   # ... and under the separate autograd functions (three weight-gradient launches)
        tags1, sep = run(True)
    finally:
        ops.LAYER_GRAPHS_ENABLED, ops.LAYER_SMALL_ENABLED = old_lg, old_small
    assert sorted(tags2) == ["conv_graphs", "conv_graphs", "rows_chain2", "rows_chain2", "rows_wgrad_multi"], tags2
    assert tags1.count("conv_graphs") == 2 and tags1.count("rows_wgrad") == 3 and "rows_wgrad_multi" not in tags1
    assert torch.equal(two[0], sep[0]) and torch.equal(two[1], sep[1])
    for a, b in zip(two[2:], sep[2:]):
        assert _rel_l2(a, b) < 2e-3
    assert _rel_l2(got[0], two[0]) < 4e-3
    for a, b in zip(got[1:], two[1:]):
        assert _rel_l2(a, b) < 0.08
    tags0, want = run(False)
    assert "conv_graphs" not in tags0 and len(tags0) >= len(tags) + 6
    assert _rel_l2(got[0], want[0]) < 1e-2
    # (gradients pass two ReLU masks: elements within bf16 noise of 0 flip between the two roundings of the conv's sums -- both paths
    #  sit 5-6 % from fp64 math on this batch, 6.5 % from each other; the conv's own backward is held to 6e-3 in the test above)
    for a, b in zip(got[1:], want[1:]):
        assert _rel_l2(a, b) < 0.12
    _, again = run(True)
    for a, b in zip(got, again):
        assert torch.equal(a, b)
    p64 = {k: v.detach().double().cpu() for k, v in layer.named_parameters()}
    ref = OL.rgin_layer(x0.double().cpu(), aug["src"].long().cpu(), aug["dst"].long().cpu(), et.cpu(), p64, regularizer="basis", num_rels=R,
                        num_bases=-1, num_mlp_layers=2, act="relu")
    assert _rel_l2(got[0], ref) < 3e-2


@pytest.mark.parametrize("H,act", [(64, "relu"), (64, "leaky_relu"), (128, "relu")])
def test_fp32_rgin_layer_takes_one_weight_gradient_launch(H, act):
    """RGINLayer(H, H) in the reference's own precision (fp32 on the bf16 split) on a config-3-shaped batch: ONE autograd function whose
    backward ends in ONE weight-gradient launch + reduce for the conv's R + 1 matrices and both Linears (dn_rows_wgrad_multi_f32) --
    the same forward launches and, up to the fp32 summation order of the weight gradients, the same numbers as the separate functions;
    against the fp64 oracle at the goldens' tolerance; bitwise repeatable; the exact-f32 mode keeps the separate functions."""
    from dummynode4graphlearning_amd import BatchedGraph, ops, synthetic, transforms
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    import oracle.layers as OL
    raw = synthetic.config3(seed=4, graphs=64)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(DEV) for k in keys), raw["max_nv"], raw["max_nvl"], raw["max_ne"],
                                      raw["max_nel"])
    N, R = int(aug["node_label"].numel()), 8
    bnn = (aug["node_ptr"][1:] - aug["node_ptr"][:-1]).long()
    bne = (aug["edge_ptr"][1:] - aug["edge_ptr"][:-1]).long()
    et = aug["edge_label"].long()
    torch.manual_seed(6)
    layer = RGINLayer(H, H, num_rels=R, regularizer="basis", num_bases=-1, num_mlp_layers=2, act_func=act).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    x0 = torch.randn(N, H, device=DEV, generator=gen)
    coef = torch.randn(N, H, device=DEV, generator=gen)

    def run():
        g = BatchedGraph(aug["src"], aug["dst"], N, bnn, bne, node_ptr=aug["node_ptr"], edge_ptr=aug["edge_ptr"])
        for p in layer.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        timer = ops.KernelTimer()
        ops.kernel_timer = timer
        try:
            out, _ = layer(g, x, et)
            out.backward(coef)
        finally:
            ops.kernel_timer = None
        return [r[0] for r in timer.records], [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]

    tags, got = run()
    assert tags.count("rows_wgrad_multi") == 1 and "rows_wgrad" not in tags, tags
    old = ops.LAYER_F32_ENABLED
    try:
        ops.LAYER_F32_ENABLED = False
        tags1, sep = run()
    finally:
        ops.LAYER_F32_ENABLED = old
    assert tags1.count("rows_wgrad") == 3 and "rows_wgrad_multi" not in tags1, tags1
    assert tags.count("rows_chain2") == 2 and tags1.count("rows_chain2") == 2   # the MLP each way as ONE launch (dn_rows_chain2_f32), in both
    assert len(tags) == len(tags1) - 2                                  # three weight-gradient calls -> one
    assert torch.equal(got[0], sep[0])                                  # the same forward launches
    for a, b in zip(got[1:], sep[1:]):                                  # (the same arithmetic; fp32 summation order of the weight gradients)
        assert _rel_l2(a, b) < 2e-5
    old_c = ops.CHAIN2_F32_ENABLED
    try:
        ops.CHAIN2_F32_ENABLED = False                                   # one launch per Linear inside the same function
        tags3, mid = run()
    finally:
        ops.CHAIN2_F32_ENABLED = old_c
    assert "rows_chain2" not in tags3 and tags3.count("rows_wgrad_multi") == 1 and len(tags3) == len(tags) + 2
    for a, b in zip(got, mid):
        assert _rel_l2(a, b) < 2e-5
    _, again = run()
    for a, b in zip(got, again):
        assert torch.equal(a, b)
    p64 = {k: v.detach().double().cpu() for k, v in layer.named_parameters()}
    ref = OL.rgin_layer(x0.double().cpu(), aug["src"].long().cpu(), aug["dst"].long().cpu(), et.cpu(), p64, regularizer="basis", num_rels=R,
                        num_bases=-1, num_mlp_layers=2, act=act)
    assert float((got[0].double().cpu() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    with ops.f32_exact(True):                                           # the exact-f32 checker mode: the separate functions
        tags2, _ = run()
    assert "rows_wgrad_multi" not in tags2


def test_fp32_rep_net_carries_the_residual_in_the_layer_launches():
    """RGINRepNet (3 layers, residual: the reference's default, config.py:345-347) in fp32 at H = 64: `outputs[-1] + layer(...)`
    (rgin.py:243-245) leaves the layer's MLP launch and the residual's gradient rides in the conv's closing gather -- no elementwise
    add launches; the same numbers as the separate functions with torch adds (forward bit-equal per layer chain, gradients to the
    split's rounding)."""
    from dummynode4graphlearning_amd import BatchedGraph, ops, synthetic, transforms
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINRepNet
    raw = synthetic.config3(seed=8, graphs=48)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(DEV) for k in keys), raw["max_nv"], raw["max_nvl"], raw["max_ne"],
                                      raw["max_nel"])
    N = int(aug["node_label"].numel())
    torch.manual_seed(9)
    net = RGINRepNet(64, 8, num_layers=3, regularizer="basis", act_func="relu").to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(4)
    x0 = torch.randn(N, 64, device=DEV, generator=gen)
    coef = torch.randn(N, 64, device=DEV, generator=gen)

    def run():
        g = BatchedGraph(aug["src"], aug["dst"], N, edata={"label": aug["edge_label"]})
        for p in net.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        timer = ops.KernelTimer()
        ops.kernel_timer = timer
        try:
            out = net.get_graph_rep(g, x)
            out.backward(coef)
        finally:
            ops.kernel_timer = None
        return [r[0] for r in timer.records], [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()]

    tags, got = run()
    assert tags.count("rows_chain2") == 6 and tags.count("rows_wgrad_multi") == 3, tags
    old = ops.LAYER_F32_ENABLED
    try:
        ops.LAYER_F32_ENABLED = False
        tags1, sep = run()
    finally:
        ops.LAYER_F32_ENABLED = old
    assert "rows_wgrad_multi" not in tags1
    for a, b in zip(got, sep):
        assert _rel_l2(a, b) < 2e-5
    _, again = run()
    for a, b in zip(got, again):
        assert torch.equal(a, b)


def test_batches_outside_the_limits_keep_the_row_factorised_launches():
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(9)
    src, dst, et, nptr, eptr = _batch(rng, 6, 4, 80, 2.0)              # a graph over 64 nodes is likely; force one
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)  # noqa: E731
    ix = ops.RowIndex(src, dst, et, nptr[-1], 4, self_loop=True, node_ptr=i32(nptr), edge_ptr=i32(eptr))
    x = torch.zeros(nptr[-1], 64, device=DEV, dtype=torch.bfloat16)
    pw = ops.PassWeights(torch.zeros(4, 64, 64, device=DEV, dtype=torch.bfloat16), torch.zeros(64, 64, device=DEV, dtype=torch.bfloat16), kn=True)
    assert ops.conv_graphs_ok(x, pw, ix) == (max(np.diff(nptr)) <= 64)
    ixg = ops.RowIndex(src, dst, et, nptr[-1], 4, self_loop=True)        # no graph boundaries: the general builder, no whole-graph launch
    assert not ops.conv_graphs_ok(x, pw, ixg)
    assert not ops.conv_graphs_ok(x.float(), pw, ix) and not ops.conv_graphs_ok(torch.zeros(nptr[-1], 128, device=DEV, dtype=torch.bfloat16), pw, ix)


@pytest.mark.parametrize("act,graphs", [("relu", 96), ("leaky_relu", 700)])
def test_wide_rgin_layer_takes_one_weight_gradient_launch(act, graphs):
    """RGINLayer(256, 256) in bf16 (BASELINE config 5's layer) as ONE autograd function whose backward ends in ONE weight-gradient
    launch + reduce for the conv's R + 1 matrices and both Linears (dn_rows_wgrad_multi_bf16 at H = 256: gathered rows, rows in row
    order and rows masked by bits side by side in one grid) -- the same forward and input-gradient launches as the separate functions
    (bitwise equal), the weight gradients equal up to the fp32 summation order of their split-K partials (other chunk boundaries);
    bitwise repeatable."""
    from dummynode4graphlearning_amd import BatchedGraph, ops, synthetic, transforms
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    raw = synthetic.config3(seed=9, graphs=graphs)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(DEV) for k in keys), raw["max_nv"], raw["max_nvl"], raw["max_ne"],
                                      raw["max_nel"])
    N, R, H = int(aug["node_label"].numel()), 8, 256
    bnn = (aug["node_ptr"][1:] - aug["node_ptr"][:-1]).long()
    bne = (aug["edge_ptr"][1:] - aug["edge_ptr"][:-1]).long()
    et = aug["edge_label"].long()
    torch.manual_seed(6)
    layer = RGINLayer(H, H, num_rels=R, regularizer="basis", num_bases=-1, num_mlp_layers=2, act_func=act).to(DEV).to(torch.bfloat16)
    gen = torch.Generator(device=DEV).manual_seed(3)
    x0 = torch.randn(N, H, device=DEV, generator=gen).to(torch.bfloat16)
    coef = torch.randn(N, H, device=DEV, generator=gen).to(torch.bfloat16)

    def run():
        g = BatchedGraph(aug["src"], aug["dst"], N, bnn, bne, node_ptr=aug["node_ptr"], edge_ptr=aug["edge_ptr"])
        for p in layer.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        timer = ops.KernelTimer()
        ops.kernel_timer = timer
        try:
            out, _ = layer(g, x, et)
            out.backward(coef)
        finally:
            ops.kernel_timer = None
        return ([r[0] for r in timer.records],
                [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters() if p.grad is not None])

    tags, got = run()
    assert tags.count("rows_wgrad_multi") == 1 and "rows_wgrad" not in tags, tags
    old = ops.LAYER_WIDE_ENABLED
    try:
        ops.LAYER_WIDE_ENABLED = False
        tags1, sep = run()
    finally:
        ops.LAYER_WIDE_ENABLED = old
    assert tags1.count("rows_wgrad") == 3 and "rows_wgrad_multi" not in tags1, tags1
    assert len(tags) == len(tags1) - 2 and len(got) == len(sep)
    assert torch.equal(got[0], sep[0]) and torch.equal(got[1], sep[1])      # the same launches for the output and the input gradient
    for a, b in zip(got[2:], sep[2:]):
        assert bool(torch.isfinite(a.float()).all()) and _rel_l2(a, b) < 4e-3   # (bf16 results of fp32 sums in another order)
    _, again = run()
    for a, b in zip(got, again):
        assert torch.equal(a, b)
