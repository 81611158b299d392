"""dn_row_index_build_local_i32 (one wavefront rank-sorts one graph in LDS, one scan) must produce the tables of the general
sort-based builder dn_row_index_build_i32 BIT FOR BIT -- the general builder is itself pinned against the oracle through the
layer goldens (tests/test_gpu_layers.py) -- on batches of graphs of every shape the modes can take."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FIELDS = ("row_in", "row_out", "aux_f_ptr", "aux_f_idx", "aux_b_ptr", "aux_b_idx", "dst_ptr", "dst_rows", "src_ptr", "src_rows")


def _build(src, dst, et, N, R, self_loop, nptr, eptr, edge_frac=0.75):
    from dummynode4graphlearning_amd import ops
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32, device=dev)  # noqa: E731
    a = ops.RowIndex(t(src), t(dst), t(et), N, R, self_loop=self_loop, edge_frac=edge_frac)
    b = ops.RowIndex(t(src), t(dst), t(et), N, R, self_loop=self_loop, edge_frac=edge_frac, node_ptr=t(nptr), edge_ptr=t(eptr))
    return a, b


def _same(a, b):
    assert a.built_by == "general" and b.built_by == "local"
    assert a.modes == b.modes and a.rel_ptr_host == b.rel_ptr_host
    assert (a.num_rows, a.num_edge_rows, a.num_aux_f, a.num_aux_b) == (b.num_rows, b.num_edge_rows, b.num_aux_f, b.num_aux_b)
    for f in FIELDS:
        x, y = getattr(a, f).cpu().numpy(), getattr(b, f).cpu().numpy()
        assert x.shape == y.shape, f
        assert np.array_equal(x, y), (f, np.flatnonzero(x != y)[:8], x[x != y][:8], y[x != y][:8])


def _random_batch(rng, G, R, nmin, nmax, dens, dummy=True, multi=True):
    """graphs with random sizes; real edges with relations < R - 2 (uniform), optional dummy node with relations R-2 / R-1
    (u -> dummy / dummy -> u: the collapsed AGG / TF relations), optional empty graphs and multi-edges"""
    src, dst, et, nptr, eptr = [], [], [], [0], [0]
    for g in range(G):
        n = int(rng.integers(nmin, nmax + 1))
        base = nptr[-1]
        m = int(rng.integers(0, max(1, int(dens * n)) + 1)) if n > 0 else 0
        if n > 0 and m > 0:
            s = rng.integers(0, n, size=m)
            d = rng.integers(0, n, size=m)
            r = rng.integers(0, max(1, R - 2), size=m)
            if multi and m > 3:                                       # repeat a few edges exactly (multi-edges)
                k = rng.integers(0, m, size=2)
                s[k[0]], d[k[0]], r[k[0]] = s[k[1]], d[k[1]], r[k[1]]
            src += list(base + s); dst += list(base + d); et += list(r)
        if dummy and n > 0:
            dn = base + n
            for u in range(n):
                src.append(base + u); dst.append(dn); et.append(R - 2)
            for u in range(n):
                src.append(dn); dst.append(base + u); et.append(R - 1)
            n += 1
        nptr.append(base + n)
        eptr.append(len(src))
    return np.array(src, np.int64), np.array(dst, np.int64), np.array(et, np.int64), np.array(nptr), np.array(eptr)


@pytest.mark.parametrize("self_loop", [True, False])
@pytest.mark.parametrize("seed", range(6))
def test_local_builder_equals_general_on_random_batches(seed, self_loop):
    rng = np.random.default_rng(100 + seed)
    R = int(rng.integers(3, 12))
    src, dst, et, nptr, eptr = _random_batch(rng, G=int(rng.integers(1, 90)), R=R, nmin=0, nmax=int(rng.integers(2, 40)),
                                             dens=float(rng.uniform(0.3, 4.0)), dummy=bool(seed % 2 == 0))
    a, b = _build(src, dst, et, int(nptr[-1]), R, self_loop, nptr, eptr)
    _same(a, b)


@pytest.mark.parametrize("edge_frac", [0.0, 0.5, 2.0])
def test_local_builder_every_mode_mix(edge_frac):
    """edge_frac 0 -> every relation EDGE; 2.0 -> every relation collapses to AGG or TF (several collapsed relations per node)"""
    rng = np.random.default_rng(7)
    src, dst, et, nptr, eptr = _random_batch(rng, G=60, R=6, nmin=1, nmax=25, dens=3.0, dummy=True)
    a, b = _build(src, dst, et, int(nptr[-1]), 6, True, nptr, eptr, edge_frac=edge_frac)
    _same(a, b)
    if edge_frac == 2.0:
        assert set(a.modes) <= {1, 2}
    if edge_frac == 0.0:
        assert set(a.modes) == {0}


def test_local_builder_config3_shape_and_layer():
    """SI-style dummy augmentation of a config-3 batch through the device pipeline; the layer on the local index equals the
    layer on the general index bit for bit (same tables -> same launches)."""
    from dummynode4graphlearning_amd import BatchedGraph, ops, synthetic, transforms
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    dev = torch.device("cuda:0")
    raw = synthetic.config3(seed=3, graphs=64)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    vocab = (raw["max_nv"], raw["max_nvl"], raw["max_ne"], raw["max_nel"])
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(dev) for k in keys), *vocab)
    N = int(aug["node_label"].numel())
    a, b = _build(aug["src"].cpu().numpy(), aug["dst"].cpu().numpy(), aug["edge_label"].cpu().numpy(), N, 8, True,
                  aug["node_ptr"].cpu().numpy(), aug["edge_ptr"].cpu().numpy())
    _same(a, b)
    assert sorted(a.modes) == [0] * 6 + [1, 2]


def test_local_builder_declines_and_falls_back():
    """a graph over the limit of the listed-graph pass, an endpoint outside its graph, ranges that do not tile: the general builder runs"""
    rng = np.random.default_rng(3)
    # (1) one graph with 8192 edges (one more than a workgroup's arrays take)
    n = 40
    s, d = rng.integers(0, n, size=8192), rng.integers(0, n, size=8192)
    src = np.concatenate([s, n + rng.integers(0, 5, size=10)]); dst = np.concatenate([d, n + rng.integers(0, 5, size=10)])
    et = rng.integers(0, 4, size=8202)
    a, b = _build(src, dst, et, n + 5, 4, True, [0, n, n + 5], [0, 8192, 8202])
    assert b.built_by == "general"
    for f in FIELDS:
        assert torch.equal(getattr(a, f), getattr(b, f))
    # (2) an edge that leaves its graph
    src2, dst2, et2 = np.array([0, 1, 2, 5]), np.array([1, 2, 0, 1]), np.array([0, 1, 0, 1])
    a, b = _build(src2, dst2, et2, 8, 2, True, [0, 4, 8], [0, 3, 4])
    assert b.built_by == "general"
    # (3) ranges that do not cover every node
    a, b = _build(src2[:3], dst2[:3], et2[:3], 8, 2, True, [0, 4, 6], [0, 3, 3])
    assert b.built_by == "general"
    # (4) ranges that point past the arrays (never dereferenced: the builder checks them against N and E first)
    a, b = _build(src2[:3], dst2[:3], et2[:3], 8, 2, True, [0, 4, 800], [0, 3, 300])
    assert b.built_by == "general"
    a, b = _build(src2[:3], dst2[:3], et2[:3], 8, 2, True, [0, -4, 8], [0, -3, 3])
    assert b.built_by == "general"
    for f in FIELDS:
        assert torch.equal(getattr(a, f), getattr(b, f))


@pytest.mark.parametrize("edge_frac", [0.0, 0.75, 2.0])
@pytest.mark.parametrize("seed", range(3))
def test_local_builder_sort_pass_on_mid_size_graphs(seed, edge_frac):
    """graphs of 257 .. 1024 edges (and smaller ones that do not fit the bit sets): ranks from sorts in one wavefront's LDS slice;
    edge_frac 2.0 collapses EVERY relation (hundreds of collapsed rows per graph, several collapsed relations per node)"""
    rng = np.random.default_rng(500 + seed)
    R = int(rng.integers(3, 17))
    src, dst, et, nptr, eptr = _random_batch(rng, G=40, R=R, nmin=0, nmax=int(rng.integers(150, 200)), dens=float(rng.uniform(1.5, 2.5)),
                                             dummy=bool(seed % 2 == 0))
    m = np.diff(eptr)
    assert m.max() <= 1024 and (m > 256).any()
    a, b = _build(src, dst, et, int(nptr[-1]), R, True, nptr, eptr, edge_frac=edge_frac)
    _same(a, b)


@pytest.mark.parametrize("edge_frac", [0.75, 2.0])
def test_local_builder_takes_graphs_over_one_wavefronts_slice_one_by_one(edge_frac):
    """a PER-GRAPH fallback: graphs of 1025 .. 8191 edges go to the listed-graph launches (one workgroup each), the small graphs
    around them stay on the one-wavefront kernels -- and the tables equal the general builder's"""
    rng = np.random.default_rng(77)
    parts = []
    for n, m_real, dummy in ((30, 50, True), (700, 2400, True), (12, 0, True), (1500, 5100, True), (3, 8185, False), (200, 900, True),
                             (2600, 2980, True), (40, 8191, False), (31, 62, True)):
        s = rng.integers(0, n, size=m_real); d = rng.integers(0, n, size=m_real); r = rng.integers(0, 5, size=m_real)
        if dummy:
            s = np.concatenate([s, np.arange(n), np.full(n, n)]); d = np.concatenate([d, np.full(n, n), np.arange(n)])
            r = np.concatenate([r, np.full(n, 5), np.full(n, 6)])
            n += 1
        parts.append((n, s, d, r))
    nptr = np.concatenate([[0], np.cumsum([p[0] for p in parts])])
    eptr = np.concatenate([[0], np.cumsum([len(p[1]) for p in parts])])
    src = np.concatenate([nptr[k] + p[1] for k, p in enumerate(parts)]); dst = np.concatenate([nptr[k] + p[2] for k, p in enumerate(parts)])
    et = np.concatenate([p[3] for p in parts])
    m = np.diff(eptr)
    assert (m > 1024).sum() >= 5 and m.max() == 8191
    a, b = _build(src, dst, et, int(nptr[-1]), 7, True, nptr, eptr, edge_frac=edge_frac)
    _same(a, b)


def test_local_builder_proteins_shaped_batch():
    """the shape bench.py --workload proteins builds (graphs of 4 .. 620 nodes, R = 16, SI dummy augmentation): every size class
    of the builder in one batch"""
    from dummynode4graphlearning_amd import synthetic, transforms
    dev = torch.device("cuda:0")
    raw = synthetic.proteins_si(2, 2048)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(dev) for k in keys), raw["max_nv"], raw["max_nvl"], raw["max_ne"],
                                      raw["max_nel"])
    m = (aug["edge_ptr"][1:] - aug["edge_ptr"][:-1]).cpu().numpy()
    assert (m > 1024).any() and (m <= 256).any() and ((m > 256) & (m <= 1024)).any()
    a, b = _build(aug["src"].cpu().numpy(), aug["dst"].cpu().numpy(), aug["edge_label"].cpu().numpy(), int(aug["node_label"].numel()), 16,
                  True, aug["node_ptr"].cpu().numpy(), aug["edge_ptr"].cpu().numpy())
    _same(a, b)


def test_local_builder_empty_batch_and_edgeless_graphs():
    a, b = _build(np.zeros(0), np.zeros(0), np.zeros(0), 7, 3, True, [0, 3, 3, 7], [0, 0, 0, 0])
    _same(a, b)
    a, b = _build(np.zeros(0), np.zeros(0), np.zeros(0), 0, 3, False, [0], [0])
    _same(a, b)


def test_local_builder_config5_full_size():
    """BASELINE config 5 at full size (32,768 graphs, N = 1,015,808, E = 3,997,696, R = 16): every table equal to the general
    builder's, compared on the device."""
    from dummynode4graphlearning_amd import ops, synthetic, transforms
    dev = torch.device("cuda:0")
    raw = synthetic.config5()
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    vocab = (raw["max_nv"], raw["max_nvl"], raw["max_ne"], raw["max_nel"])
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(dev) for k in keys), *vocab)
    N = int(aug["node_label"].numel())
    assert (N, int(aug["src"].numel())) == (1015808, 3997696)
    a = ops.RowIndex(aug["src"], aug["dst"], aug["edge_label"], N, 16, self_loop=True)
    b = ops.RowIndex(aug["src"], aug["dst"], aug["edge_label"], N, 16, self_loop=True, node_ptr=aug["node_ptr"],
                     edge_ptr=aug["edge_ptr"])
    assert (a.built_by, b.built_by) == ("general", "local")
    assert a.modes == b.modes == [0] * 14 + [1, 2] and a.rel_ptr_host == b.rel_ptr_host
    for f in FIELDS:
        assert torch.equal(getattr(a, f), getattr(b, f)), f


def test_pyg_style_batches_reach_the_local_builder():
    """graph.row_index_of derives the edge ranges of a collated PyG-style batch (edges grouped by graph) and the local builder
    takes it; a batch whose edges are NOT grouped by graph is declined on the device and the general builder runs -- same tables."""
    from dummynode4graphlearning_amd import graph as G
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    src, dst, et, nptr, eptr = _random_batch(rng, G=40, R=4, nmin=2, nmax=30, dens=2.5, dummy=True)
    N = int(nptr[-1])
    batch = torch.from_numpy(np.repeat(np.arange(len(nptr) - 1), np.diff(nptr))).to(dev)
    mk = lambda s, d: G.GraphBatch(torch.zeros(N, 4, device=dev), torch.stack([torch.from_numpy(s), torch.from_numpy(d)]).to(dev),  # noqa: E731
                                   batch=batch)
    etype = torch.from_numpy(et).to(dev)
    ix = G.row_index_of(mk(src, dst), etype, 4, True).parts[0][2]
    assert ix.built_by == "local"
    perm = rng.permutation(len(src))                                     # shuffle the edge list across graphs
    etype_p = torch.from_numpy(et[perm]).to(dev)
    ix_p = G.row_index_of(mk(src[perm], dst[perm]), etype_p, 4, True).parts[0][2]
    assert ix_p.built_by == "general"
    assert ix.num_rows == ix_p.num_rows and ix.modes == ix_p.modes       # (same multiset of rows; the order follows the edge ids)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_local_builder_graphs_of_several_hundred_edges(seed):
    """graphs with 257 .. 1024 edges take the pair-loop statistics (the hash table serves up to 256) and several 64-edge chunks per
    lane in the fill pass; mixed with small graphs in one batch"""
    rng = np.random.default_rng(40 + seed)
    parts = [_random_batch(rng, G=6, R=7, nmin=60, nmax=150, dens=float(rng.uniform(2.5, 4.5)), dummy=bool(seed % 2)),
             _random_batch(rng, G=20, R=7, nmin=0, nmax=20, dens=2.0, dummy=True)]
    src, dst, et, nptr, eptr = parts[0]
    s2, d2, e2, n2, p2 = parts[1]
    src = np.concatenate([src, s2 + nptr[-1]]); dst = np.concatenate([dst, d2 + nptr[-1]]); et = np.concatenate([et, e2])
    eptr = np.concatenate([eptr, p2[1:] + eptr[-1]]); nptr = np.concatenate([nptr, n2[1:] + nptr[-1]])
    sizes = np.diff(eptr)
    assert sizes.max() > 256 and sizes.max() <= 1024, sizes.max()
    a, b = _build(src, dst, et, int(nptr[-1]), 7, True, nptr, eptr)
    _same(a, b)


# ---------------------------------------------------------------------------------------------------------------------------------
# dn_conv_index_build_i32: row index + unit streams + sweep orders + chunk table in ONE call must leave what the separate calls do
def _index_pair(src, dst, et, N, R, nptr, eptr):
    """(one-call index, index by the separate builders + prepare_closing) of the same batch."""
    from dummynode4graphlearning_amd import ops
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32, device=dev)  # noqa: E731
    a = ops.RowIndex(t(src), t(dst), t(et), N, R, self_loop=True, node_ptr=t(nptr), edge_ptr=t(eptr),
                     closing_hint=(256, torch.bfloat16))
    b = ops.RowIndex(t(src), t(dst), t(et), N, R, self_loop=True, node_ptr=t(nptr), edge_ptr=t(eptr))
    return a, b


def _entry_positions(units):
    """positions of ent_row / ent_mask that the ENTRY units of a stream cover"""
    u = units[(units[:, 0] & 1) != 0]
    if u.shape[0] == 0:
        return np.zeros(0, np.int64)
    return np.concatenate([np.arange(b, e) for b, e in zip(u[:, 1], u[:, 2])])


def _tiles_per_workgroup(table, count):
    """the non-empty tiles of every workgroup of a sweep table, in order"""
    t = table.cpu().numpy().reshape(256, -1, 4)
    assert t.shape[1] * 256 == count
    return [[tuple(e[:3]) for e in w if e[2] > e[1]] for w in t]


def _same_closing(a, b, served):
    from dummynode4graphlearning_amd import ops
    for f in FIELDS + ("rel_ptr_dev",):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    assert a.rel_ptr_host == b.rel_ptr_host and a.modes == b.modes and a.built_by == b.built_by
    (ta, pa, ma), (tb, pb, mb) = a.chunk_table, b.chunk_table
    assert ma == mb and torch.equal(ta[:ma], tb[:mb]) and torch.equal(pa[:a.num_all_rels + 1], pb[:b.num_all_rels + 1])
    ops.prepare_closing(b, 256, torch.bfloat16)
    absorbed = b.built_by == "local" and all(b._units[d].agg for d in "fb")
    if served is not None:                                            # served: the one call left nothing to build for the first step
        assert bool(a._units) == served and (absorbed or not served)  # (the separate builders may absorb where the call does not serve:
    else:                                                             #  empty graphs between the others -- fewer segments than graphs)
        assert bool(a._units) == absorbed
    ops.prepare_closing(a, 256, torch.bfloat16)
    for d in "fb":
        fa, fb_ = a._fold[d], b._fold[d]
        assert (fa is None) == (fb_ is None)
        if fa is not None:
            assert (fa.rel, fa.beg, fa.end, fa.n) == (fb_.rel, fb_.beg, fb_.end, fb_.n) and torch.equal(fa.add_idx, fb_.add_idx)
            assert (fa.graph_tiles is None) == (fb_.graph_tiles is None)
            assert (fa.multi is None) == (fb_.multi is None)
            if fa.multi is not None:                                  # the chunked form: T tiles exist (on the device), the tables hold more
                assert torch.equal(fa.multi[0], fb_.multi[0]) and torch.equal(fa.multi[1], fb_.multi[1]) and fa.multi[2:] == fb_.multi[2:]
                T = int(fa.multi[0][-1])
                assert torch.equal(fa.graph_tiles[0][:T + 1], fb_.graph_tiles[0][:T + 1])
                assert torch.equal(fa.graph_tiles[1][:T], fb_.graph_tiles[1][:T])
            elif fa.graph_tiles is not None:
                for x, y in zip(fa.graph_tiles, fb_.graph_tiles):
                    assert torch.equal(x[:fa.n], y[:fa.n])
            assert (fa.sweep_tiles is None) == (fb_.sweep_tiles is None)
            if fa.sweep_tiles is not None:
                assert _tiles_per_workgroup(*fa.sweep_tiles) == _tiles_per_workgroup(*fb_.sweep_tiles)
        ua, ub = a._units[d], b._units[d]
        assert (ua.num_wg, ua.num_nodes, ua.num_tiles, ua.agg, ua.order, ua.num_segments) == (ub.num_wg, ub.num_nodes, ub.num_tiles, ub.agg,
                                                                                            ub.order, ub.num_segments)
        assert torch.equal(ua.unit_ptr, ub.unit_ptr)
        n = int(ua.unit_ptr[-1])
        xa, xb = ua.units[:n].cpu().numpy(), ub.units[:n].cpu().numpy()
        assert np.array_equal(xa, xb)
        pos = _entry_positions(xa)
        for f in ("ent_row", "ent_mask"):
            assert np.array_equal(getattr(ua, f).cpu().numpy()[pos], getattr(ub, f).cpu().numpy()[pos]), (d, f)


@pytest.mark.parametrize("graphs", [1, 37, 300, 4096])
def test_one_call_index_leaves_what_the_separate_builders_leave(graphs):
    """config-5-shaped batches (every graph + its dummy node inside 32 nodes): everything is served by the one call -- 4,096 graphs
    are enough rows for the sweep orders to be built too."""
    from dummynode4graphlearning_amd import synthetic, transforms
    raw = synthetic.config5(seed=graphs, graphs=graphs)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).cuda() for k in keys), raw["max_nv"], raw["max_nvl"], raw["max_ne"],
                                      raw["max_nel"])
    N = int(aug["node_label"].numel())
    a, b = _index_pair(aug["src"].cpu().numpy(), aug["dst"].cpu().numpy(), aug["edge_label"].cpu().numpy(), N, raw["num_rels"],
                       aug["node_ptr"].cpu().numpy(), aug["edge_ptr"].cpu().numpy())
    _same_closing(a, b, served=True if graphs > 1 else None)   # (one graph: every relation collapses, no single fold)
    if graphs == 4096:
        assert a._fold["f"].sweep_tiles is not None


@pytest.mark.parametrize("case", ["large graphs", "tu sizes", "one large graph"])
def test_one_call_index_serves_graphs_over_32_nodes_with_the_chunked_form(case):
    """A graph over 32 nodes no longer sends the batch to the separate builders: the same queued launches build the CHUNKED tiles
    and their unit streams (decided on the device); tables equal to what the separate entry points build."""
    rng = np.random.default_rng(78)
    if case == "large graphs":
        src, dst, et, nptr, eptr = _random_batch(rng, G=60, R=6, nmin=40, nmax=70, dens=2.0)
    elif case == "tu sizes":
        src, dst, et, nptr, eptr = _random_batch(rng, G=300, R=6, nmin=1, nmax=200, dens=1.5)
    else:
        src, dst, et, nptr, eptr = _random_batch(rng, G=90, R=6, nmin=3, nmax=20, dens=2.0)
        n0 = 500
        s0 = np.concatenate([rng.integers(0, n0, size=700), np.arange(n0), np.full(n0, n0)])
        d0 = np.concatenate([rng.integers(0, n0, size=700), np.full(n0, n0), np.arange(n0)])
        e0 = np.concatenate([rng.integers(0, 4, size=700), np.full(n0, 4), np.full(n0, 5)])
        src, dst, et = np.concatenate([s0, src + n0 + 1]), np.concatenate([d0, dst + n0 + 1]), np.concatenate([e0, et])
        nptr, eptr = np.concatenate([[0], nptr + n0 + 1]), np.concatenate([[0], eptr + len(s0)])
    a, b = _index_pair(src, dst, et, int(nptr[-1]), 6, nptr, eptr)
    assert a.built_by == b.built_by == "local"
    _same_closing(a, b, served=True)
    assert all(a._units[d].order >= 2 and b._units[d].order >= 2 for d in "fb")


@pytest.mark.parametrize("case", ["no dummy", "random sizes", "over the LDS limit"])
def test_one_call_index_where_it_cannot_serve(case):
    """Batches the queued table builders do not serve -- no collapsed relation, empty graphs between the others, a graph the local
    builder rejects: the call must leave the row index (or hand over to the general builder) and touch nothing else; the tables
    are then built on first use, as without the hint."""
    rng = np.random.default_rng(77)
    if case == "no dummy":
        src, dst, et, nptr, eptr = _random_batch(rng, G=200, R=6, nmin=3, nmax=20, dens=2.0, dummy=False)
    elif case == "random sizes":
        src, dst, et, nptr, eptr = _random_batch(rng, G=150, R=6, nmin=0, nmax=45, dens=3.0)
    else:                                                             # one graph of 9,000 edges in front of 40 small ones: the local
        src, dst, et, nptr, eptr = _random_batch(rng, G=40, R=6, nmin=3, nmax=20, dens=2.0)     # builder is tried and raises its flag
        n0, m0 = 60, 9000
        src = np.concatenate([rng.integers(0, n0, size=m0), src + n0])
        dst = np.concatenate([rng.integers(0, n0, size=m0), dst + n0])
        et = np.concatenate([rng.integers(0, 4, size=m0), et])
        nptr, eptr = np.concatenate([[0], nptr + n0]), np.concatenate([[0], eptr + m0])
    a, b = _index_pair(src, dst, et, int(nptr[-1]), 6, nptr, eptr)
    if case == "over the LDS limit":
        assert a.built_by == b.built_by == "general"
    _same_closing(a, b, served=False)


def test_layer_on_the_one_call_index_is_bitwise_the_layer_on_the_separate_builders():
    from dummynode4graphlearning_amd import ops, synthetic, transforms
    raw = synthetic.config5(seed=4, graphs=3000)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).cuda() for k in keys), raw["max_nv"], raw["max_nvl"], raw["max_ne"],
                                      raw["max_nel"])
    N, R, H = int(aug["node_label"].numel()), raw["num_rels"], 256
    gen = torch.Generator(device="cuda").manual_seed(3)
    W = (torch.randn(R, H, H, device="cuda", generator=gen) / 16).to(torch.bfloat16).requires_grad_(True)
    Wl = (torch.randn(H, H, device="cuda", generator=gen) / 16).to(torch.bfloat16).requires_grad_(True)
    bias = torch.randn(H, device="cuda", generator=gen).to(torch.bfloat16).requires_grad_(True)
    x0 = torch.randn(N, H, device="cuda", generator=gen).to(torch.bfloat16)
    gout = torch.randn(N, H, device="cuda", generator=gen).to(torch.bfloat16)
    res = []
    for hint in ((256, torch.bfloat16), None):
        iset = ops.RowIndexSet(aug["src"], aug["dst"], aug["edge_label"], N, R, True, node_ptr=aug["node_ptr"],
                               edge_ptr=aug["edge_ptr"], closing_hint=hint)
        assert bool(iset.parts[0][2]._units) == (hint is not None)
        x = x0.clone().requires_grad_(True)
        out = ops.rel_transform_fused(x, W, bias, iset, W_loop=Wl)
        gx, gW, gWl, gb = torch.autograd.grad(out, (x, W, Wl, bias), gout)
        res.append((out.detach(), gx, gW, gWl, gb))
    for u, v in zip(*res):
        assert torch.equal(u, v)


def test_one_call_index_checks_its_workspace():
    import ctypes
    from dummynode4graphlearning_amd import _lib
    L = _lib.lib()
    G, N, R, E, wg = 4, 40, 6, 100, 256
    need = L.dn_conv_index_workspace_bytes(G, N, R, E, wg, 0)
    assert need >= L.dn_row_index_local_workspace_bytes(G, N, R, E) + 2 * L.dn_close_units_workspace_bytes(G, wg)
    assert L.dn_conv_index_workspace_bytes(G, N, R, E, wg, 5000) >= (L.dn_row_index_local_workspace_bytes(G, N, R, E)
                                                                    + 2 * L.dn_close_units_workspace_bytes(5000, wg))
    cap = L.dn_close_units_capacity(G, E + N, wg)
    counts, rel, modes, st = (ctypes.c_int64 * 5)(), (ctypes.c_int32 * (R + 1))(), (ctypes.c_int32 * R)(), ctypes.c_int32(0)
    absorb, plan = (ctypes.c_int32 * 4)(), (ctypes.c_int32 * 6)()
    P256 = ctypes.c_void_p(256)
    for ws, nbytes, msg in ((P256, need - 1, b"workspace too small"), (ctypes.c_void_p(16), need, b"unaligned workspace")):
        rc = L.dn_conv_index_build_i32(G, N, R, E, P256, P256, P256, P256, P256, 1, 0.75, *([P256] * 10), counts, rel, modes,
                                       ctypes.byref(st), P256, P256, P256, P256, P256, absorb, wg, 1, cap, *([P256] * 8), 0, 0, *([None] * 8),
                                       32, 8, P256, P256, 256, 4096, 64, P256, P256, plan, ws, nbytes, None)
        assert rc == -1 and msg in L.dn_last_error(), L.dn_last_error()


def test_layer_asks_for_the_one_call_index():
    """An H = 256 bf16 RGINLayer on a fresh BatchedGraph: the layer's closing hint makes the batch's index ONE library call -- none of
    the separate table builders runs during the step."""
    from dummynode4graphlearning_amd import BatchedGraph, _lib, synthetic, transforms
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    raw = synthetic.config5(seed=12, graphs=500)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).cuda() for k in keys), raw["max_nv"], raw["max_nvl"], raw["max_ne"],
                                      raw["max_nel"])
    N, R, H = int(aug["node_label"].numel()), raw["num_rels"], 256
    bnn = (aug["node_ptr"][1:] - aug["node_ptr"][:-1]).long()
    bne = (aug["edge_ptr"][1:] - aug["edge_ptr"][:-1]).long()
    g = BatchedGraph(aug["src"], aug["dst"], N, bnn, bne, node_ptr=aug["node_ptr"], edge_ptr=aug["edge_ptr"])
    torch.manual_seed(0)
    layer = RGINLayer(H, H, num_rels=R, regularizer="basis", num_bases=-1, num_mlp_layers=2, act_func="relu").cuda().to(torch.bfloat16)
    x = torch.randn(N, H, device="cuda").to(torch.bfloat16).requires_grad_(True)
    L = _lib.lib()
    calls = {}
    names = ("dn_conv_index_build_i32", "dn_row_index_build_local_i32", "dn_row_index_build_i32", "dn_close_units_build_i32",
             "dn_sweep_tables_build_i32", "dn_fold_graph_tiles_build_i32", "dn_fold_tables_build_async_i32", "dn_slot_table_build_i32")
    saved = {n: getattr(L, n) for n in names}
    try:
        for n in names:
            def spy(*a, _n=n):
                calls[_n] = calls.get(_n, 0) + 1
                return saved[_n](*a)
            setattr(L, n, spy)
        out, _ = layer(g, x, aug["edge_label"].long())
        out.backward(torch.ones_like(out))
    finally:
        for n in names:
            setattr(L, n, saved[n])
    assert calls == {"dn_conv_index_build_i32": 1}, calls
    assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(x.grad.float()).all())
