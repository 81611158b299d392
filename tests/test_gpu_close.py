"""The H = 256 closing launch as a unit stream (csrc/dn_close.hip): dn_close_units_build_i32 bit-exact against its host
restatement (tests/close_ref.py), dn_rows_close_bf16 against fp64 math on the same bf16 operands and against the slot kernel it
replaces (dn_rows_selfsum_bf16 + dn_overflow_rows_add_bf16).  Reference semantics: the fn.sum reduce + self loop + bias of
subgraph_isomorphism/models/rgin.py:137-146."""
import numpy as np
import pytest
import torch

from close_ref import close_units_ref, graph_tiles_multi_ref, graph_tiles_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lists(rng, N, P, kind):
    """Per-node row lists (row-ordered, each ending with the node's self row P + v as the row index builders emit them)."""
    lists = []
    shared = np.sort(rng.choice(P, size=max(N // 20, 2), replace=False))        # rows fanned out to runs of consecutive nodes
    for v in range(N):
        c = int(rng.poisson(2.0))
        rows = list(rng.integers(0, P, size=c))
        if kind != "plain":
            rows.append(int(shared[(v // 29) % len(shared)]))                    # one row shared by ~29 consecutive nodes
        if kind == "repeat" and v % 7 == 0 and rows:
            rows.append(rows[0])                                                 # the same row twice in ONE list (parallel edges)
        if kind == "hub" and v in (3, 40, N - 1):
            rows += list(rng.integers(0, P, size=400 if v == 40 else 90))        # a tile over the de-duplication capacity / long lists
        lists.append(np.append(np.sort(np.array(rows, dtype=np.int64)), P + v))
    if kind == "unsorted":
        lists[5] = np.array([7, 9, 7, P + 5])                                    # a repeat that is not adjacent: the tile is listed as is
    ptr = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    return lists, ptr, np.concatenate(lists)


@pytest.mark.parametrize("kind", ["plain", "shared", "repeat", "hub", "unsorted"])
@pytest.mark.parametrize("N,G,order", [(1000, 256, 0), (1000, 256, 1), (33, 4, 0), (64, 1, 0), (2500, 7, 0), (2500, 8, 1), (90, 16, 1),
                                       (5000, 64, 1)])
def test_close_unit_tables_match_the_host_restatement(kind, N, G, order):
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(N + G)
    P = 3 * N
    lists, ptr, rows = _lists(rng, N, P, kind)
    lp, lr = torch.from_numpy(ptr).to(DEV).int(), torch.from_numpy(rows).to(DEV).int()
    for drop in ((0, 0), (P // 3, P // 2)):
        cu = ops.build_close_units(lp, lr, N, P, drop=drop, num_wg=G, order=order)
        up, un, tiles = close_units_ref(ptr, rows, N, P, G, drop, order=order)
        got_up = cu.unit_ptr.cpu().numpy()
        assert np.array_equal(got_up, up), (got_up[:8], up[:8])
        assert np.array_equal(cu.units.cpu().numpy()[:len(un)], un)
        er, em = cu.ent_row.cpu().numpy(), cu.ent_mask.cpu().numpy().view(np.uint32)
        for t, (e0, r, m) in tiles.items():
            assert list(er[e0:e0 + len(r)]) == r, t
            assert list(em[e0:e0 + len(r)]) == m, t
        # the lists are reproduced exactly: node v's kept rows (with multiplicity) = the entries of its tile that carry its bit
        for v in rng.integers(0, N, size=50):
            e0, r, m = tiles[v // 32]
            mine = sorted(rr for rr, mm in zip(r, m) if (mm >> (v % 32)) & 1)
            want = sorted(int(x) for x in lists[v] if x < P and not (drop[0] <= x < drop[1]))
            assert mine == want


def _close_case(rng, N, P, kind):
    H = 256
    lists, ptr, rows = _lists(rng, N, P, kind)
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)  # noqa: E731
    x, Y = bf(rng.standard_normal((N, H))), bf(rng.standard_normal((P, H)))
    W = bf(rng.standard_normal((H, H)) / np.sqrt(H))                           # [in][out], as the parameter stores it
    b = bf(rng.standard_normal(H))
    return lists, ptr, rows, x, Y, W, b


def _ref_close(x, W, b, Y, lists, P, drop=(0, 0)):
    ref = x.double() @ W.double() + (b.double() if b is not None else 0.0)
    for v, l in enumerate(lists):
        keep = [int(r) for r in l if r < P and not (drop[0] <= r < drop[1])]
        if keep:
            ref[v] += Y[keep].double().sum(0)
    return ref


@pytest.mark.parametrize("kind", ["shared", "repeat", "hub"])
@pytest.mark.parametrize("N,G", [(1003, 256), (1003, 3), (37, 256), (32, 1), (4100, 256)])
def test_rows_close_matches_reference(kind, N, G):
    """Every pipeline length (workgroups with 0, 1, 2, ... units up to hundreds), both weight layouts, with / without bias, a
    dropped row range; against fp64 on the same bf16 operands (one rounding of the fp32 sums: 2^-8 relative to the row)."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(N * 7 + G)
    P = 3 * N
    lists, ptr, rows, x, Y, W, b = _close_case(rng, N, P, kind)
    lp, lr = torch.from_numpy(ptr).to(DEV).int(), torch.from_numpy(rows).to(DEV).int()
    xd, Yd, Wd, bd = x.to(DEV), Y.to(DEV), W.to(DEV), b.to(DEV)
    Wn = Wd.t().contiguous()
    cu = ops.build_close_units(lp, lr, N, P, num_wg=G)
    outs = []
    for bias, w_kn in ((bd, True), (bd, False), (None, True)):
        out = ops.rows_close(xd, Wd if w_kn else Wn, bias, Yd, cu, w_kn=w_kn)
        ref = _ref_close(x, W, b if bias is not None else None, Y, lists, P)
        err = (out.cpu().double() - ref).abs() / (ref.abs() + 1.0)
        assert float(err.max()) < 6e-3, (w_kn, float(err.max()), int(err.max(1).values.argmax()))
        outs.append(out)
    assert torch.equal(outs[0], outs[1])                                       # the two weight layouts: same arithmetic
    assert torch.equal(outs[0], ops.rows_close(xd, Wd, bd, Yd, cu, w_kn=True))  # run to run
    d0, d1 = P // 4, P // 2
    cud = ops.build_close_units(lp, lr, N, P, drop=(d0, d1), num_wg=G)
    out = ops.rows_close(xd, Wd, None, Yd, cud, w_kn=True)
    ref = _ref_close(x, W, None, Y, lists, P, (d0, d1))
    assert float(((out.cpu().double() - ref).abs() / (ref.abs() + 1.0)).max()) < 6e-3


def test_rows_close_against_the_slot_kernel_and_without_any_rows():
    """Same inputs through dn_rows_selfsum_bf16 + dn_overflow_rows_add_bf16: the two closing launches agree to the slot kernel's
    extra roundings; a batch without a single list row is x W + b."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(11)
    N, P = 2000, 5000
    lists, ptr, rows, x, Y, W, b = _close_case(rng, N, P, "hub")
    lp, lr = torch.from_numpy(ptr).to(DEV).int(), torch.from_numpy(rows).to(DEV).int()
    xd, Yd, Wd, bd = x.to(DEV), Y.to(DEV), W.to(DEV), b.to(DEV)
    cu = ops.build_close_units(lp, lr, N, P)
    new = ops.rows_close(xd, Wd, bd, Yd, cu, w_kn=True)
    slots, over = ops.build_slot_table(lp, lr, N, P)
    old = ops.rows_selfsum(xd, Wd.t().contiguous(), bd, Yd, None, slots, lists=(lp, lr, P, 0, 0, over))
    ref = _ref_close(x, W, b, Y, lists, P)
    e_new = float(((new.cpu().double() - ref).abs() / (ref.abs() + 1.0)).max())
    e_old = float(((old.cpu().double() - ref).abs() / (ref.abs() + 1.0)).max())
    assert e_new < 6e-3 and e_new <= e_old + 1e-6, (e_new, e_old)
    only_self = [np.array([P + v]) for v in range(N)]
    p2 = torch.arange(N + 1, device=DEV, dtype=torch.int32)
    r2 = torch.arange(P, P + N, device=DEV, dtype=torch.int32)
    cu0 = ops.build_close_units(p2, r2, N, P)
    out = ops.rows_close(xd, Wd, bd, None, cu0, w_kn=True)
    ref = _ref_close(x, W, b, Y, only_self, P)
    assert float(((out.cpu().double() - ref).abs() / (ref.abs() + 1.0)).max()) < 6e-3


def test_non_finite_product_row_poisons_its_column_of_the_tile_and_nothing_else():
    """The documented difference to the reference's index_add_ ("non-finite values"): the unit-stream launch sums
    through a 0/1 selection-matrix MFMA, and 0 x NaN = NaN, so ONE non-finite element of a product row turns that COLUMN NaN for
    every node of the 32-node tile whose entry unit holds the row -- and for nothing else: other columns of the tile, other tiles
    and the nodes of other units stay finite and correct.  The slot kernel (DN_CLOSE_RING=0) confines it to the nodes that sum
    the row, as the reference does."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(23)
    N, P = 640, 1500
    lists, ptr, rows, x, Y, W, b = _close_case(rng, N, P, "plain")
    v, col = 100, 77                                                           # node 100 (tile 3) sums the poisoned row
    bad = int(lists[v][0]) if len(lists[v]) > 1 else None
    if bad is None:
        lists[v] = np.array([5, P + v])
        ptr = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
        rows, bad = np.concatenate(lists), 5
    Y = Y.clone()
    Y[bad, col] = float("nan")
    users = [u for u in range(N) if bad in lists[u][:-1]]
    lp, lr = torch.from_numpy(ptr).to(DEV).int(), torch.from_numpy(rows).to(DEV).int()
    xd, Yd, Wd, bd = x.to(DEV), Y.to(DEV), W.to(DEV), b.to(DEV)
    out = ops.rows_close(xd, Wd, bd, Yd, ops.build_close_units(lp, lr, N, P), w_kn=True).float().cpu()
    nan = torch.isnan(out)
    tiles = sorted({u // 32 for u in users})
    want = torch.zeros_like(nan)
    for t in tiles:
        want[32 * t:32 * t + 32, col] = True
    assert bool((nan & ~want).sum() == 0)                                      # nothing outside those tiles' column `col`
    assert all(bool(nan[u, col]) for u in users)                               # the nodes that sum the row: NaN, as in the reference
    Yc = Y.clone()
    Yc[bad, col] = 0.0
    ref = _ref_close(x, W, b, Yc, lists, P)
    ok = ~want
    assert float((((out.double() - ref).abs() / (ref.abs() + 1.0))[ok]).max()) < 6e-3
    slots, over = ops.build_slot_table(lp, lr, N, P)
    old = ops.rows_selfsum(xd, Wd.t().contiguous(), bd, Yd, None, slots, lists=(lp, lr, P, 0, 0, over)).float().cpu()
    nan_old = torch.isnan(old)
    assert sorted(int(u) for u in nan_old.any(1).nonzero().reshape(-1)) == sorted(users) and bool(nan_old[:, col].sum() == nan_old.sum())


def test_parameter_layout_weights_give_the_same_bits_as_the_transposed_copy():
    """w_kn = 1 (weights read as the reference stores them, [in][out]) against w_kn = 0 on an explicit transposed copy: the ring
    transform (several relation switches per workgroup) and the fold tail, bit for bit; unsupported widths refuse."""
    from dummynode4graphlearning_amd import ops
    from dummynode4graphlearning_amd._lib import DnHipError
    rng = np.random.default_rng(5)
    H, R, N = 256, 7, 3000
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16).to(DEV)  # noqa: E731
    X = bf(rng.standard_normal((N, H)))
    W = bf(rng.standard_normal((R, H, H)) / np.sqrt(H))                        # [R][in][out]
    Wn = W.transpose(1, 2).contiguous()
    cnt = rng.integers(0, 900, size=R)
    cnt[3] = 0                                                                 # an empty relation
    rel_ptr = np.concatenate([[0], np.cumsum(cnt)])
    P = int(rel_ptr[-1])
    idx = torch.from_numpy(rng.integers(0, N, size=P)).to(DEV).int()
    tiles = ops.make_row_tiles([int(v) for v in rel_ptr], torch.device(DEV))
    a = ops.rows_transform(X, W, tiles, P, idx=idx, w_kn=True)
    b = ops.rows_transform(X, Wn, tiles, P, idx=idx)
    assert torch.equal(a, b)
    ref = torch.cat([X[idx[rel_ptr[r]:rel_ptr[r + 1]].long()].double() @ W[r].double() for r in range(R)])
    assert float((a.double() - ref).abs().max() / ref.abs().max()) < 8e-3
    # fold tail
    nseg = 70
    part_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(rng.integers(1, 4, size=nseg))])).to(DEV).int()
    part = torch.from_numpy(rng.standard_normal((int(part_ptr[-1]), H)).astype(np.float32)).to(DEV)
    tgt = torch.from_numpy(rng.permutation(N)[:nseg]).to(DEV).int()
    o1, o2 = X.clone(), X.clone()
    x1 = ops.fold_tail(part, part_ptr, nseg, W[2], tgt, o1, w_kn=True)
    x2 = ops.fold_tail(part, part_ptr, nseg, Wn[2].contiguous(), tgt, o2)
    assert torch.equal(o1, o2) and torch.equal(x1, x2)
    # (H = 64 / 128 take w_kn too since round 5: tests/test_gpu_kernels.py); a width no MFMA kernel serves still refuses
    W64 = bf(rng.standard_normal((1, 64, 64)))
    t64 = ops.make_row_tiles([0, 100], torch.device(DEV))
    x64 = X[:, :64].contiguous()
    assert torch.equal(ops.rows_transform(x64, W64, t64, 100, w_kn=True), ops.rows_transform(x64, W64.transpose(1, 2).contiguous(), t64, 100))
    with pytest.raises((DnHipError, AssertionError)):
        ops.rows_transform(X[:, :96].contiguous(), bf(rng.standard_normal((1, 96, 96))), t64, 100, w_kn=True)


def _graph_batch(rng, sizes, P):
    """Graphs of `size` real nodes + one dummy node each; lists as the row index emits them; the segment of a graph = its real nodes."""
    lists, seg_ptr, seg_nodes, dummies, base = [], [0], [], [], 0
    for n in sizes:
        for v in range(n + 1):
            c = int(rng.poisson(2.0)) if v < n else 1
            lists.append(np.append(np.sort(rng.integers(0, P, size=c)), P + base + v))
        seg_nodes += list(range(base, base + n))
        seg_ptr.append(len(seg_nodes))
        dummies.append(base + n)
        base += n + 1
    ptr = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    return lists, ptr, np.concatenate(lists), np.array(seg_ptr), np.array(seg_nodes), np.array(dummies), base


@pytest.mark.parametrize("G,order", [(256, 0), (256, 1), (5, 0), (1, 0), (8, 1), (48, 1)])
@pytest.mark.parametrize("sizes_kind", ["config5", "mixed"])
def test_absorbed_fold_tables_and_launch(sizes_kind, G, order):
    """Every graph inside one tile: dn_fold_graph_tiles_build_i32 + the unit tables with AGG units against the host restatement,
    and the launch -- out = x W_loop + b + list rows, aux[j] = bf16 column sum of graph j's real nodes, out[dummy_j] += aux[j] W_agg
    -- against fp64 (what dn_rows_close_bf16 + dn_fold_tail_bf16 compute in two launches)."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(G + len(sizes_kind))
    sizes = [30] * 300 if sizes_kind == "config5" else list(rng.integers(1, 32, size=260)) + [31, 31, 1, 1]
    H, P = 256, 4000
    lists, ptr, rows, seg_ptr, seg_nodes, dummies, N = _graph_batch(rng, sizes, P)
    S = len(sizes)
    sp, sn = torch.from_numpy(seg_ptr).to(DEV).int(), torch.from_numpy(seg_nodes).to(DEV).int()
    tgt = torch.from_numpy(dummies).to(DEV).int()
    tile_ptr, info, ok = ops.build_graph_tiles(sp, sn, N, add_idx=tgt)
    okr, tpr, infor = graph_tiles_ref(seg_ptr, seg_nodes, N, add_idx=dummies)
    assert okr and int(ok.item()) != 0
    assert np.array_equal(tile_ptr.cpu().numpy(), tpr) and np.array_equal(info.cpu().numpy(), infor)
    lp, lr = torch.from_numpy(ptr).to(DEV).int(), torch.from_numpy(rows).to(DEV).int()
    cu = ops.build_close_units(lp, lr, N, P, num_wg=G, tile_ptr=tile_ptr, agg=True, order=order)
    up, un, tiles = close_units_ref(ptr, rows, N, P, G, tile_ptr=tpr, agg=True, order=order)
    assert np.array_equal(cu.unit_ptr.cpu().numpy(), up)
    assert np.array_equal(cu.units.cpu().numpy()[:len(un)], un)
    er, em = cu.ent_row.cpu().numpy(), cu.ent_mask.cpu().numpy().view(np.uint32)
    for t, (e0, r, m) in tiles.items():
        assert list(er[e0:e0 + len(r)]) == r and list(em[e0:e0 + len(r)]) == m, t
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)  # noqa: E731
    x, Y = bf(rng.standard_normal((N, H))), bf(rng.standard_normal((P, H)))
    W, Wa = bf(rng.standard_normal((H, H)) / np.sqrt(H)), bf(rng.standard_normal((H, H)) / np.sqrt(H))
    b = bf(rng.standard_normal(H))
    for w_kn in (True, False):
        aux = torch.empty((S, H), dtype=torch.bfloat16, device=DEV)
        Wd, Wad = (W.to(DEV), Wa.to(DEV)) if w_kn else (W.t().contiguous().to(DEV), Wa.t().contiguous().to(DEV))
        out = ops.rows_close(x.to(DEV), Wd, b.to(DEV), Y.to(DEV), cu, w_kn=w_kn, agg=(info, Wad, aux, tgt))
        aux_ref = torch.stack([x[seg_nodes[seg_ptr[j]:seg_ptr[j + 1]]].double().sum(0) for j in range(S)])
        assert float((aux.cpu().double() - aux_ref).abs().max() / aux_ref.abs().max()) < 5e-3
        ref = _ref_close(x, W, b, Y, lists, P)
        ref[dummies] += aux.cpu().double() @ Wa.double()                         # (the launch multiplies its own rounded aux rows)
        err = (out.cpu().double() - ref).abs() / (ref.abs() + 1.0)
        real = np.setdiff1d(np.arange(N), dummies)
        assert float(err[real].max()) < 8e-3, (w_kn, float(err[real].max()))
        # a dummy node's row is rounded to bf16 twice (stored, then re-read by its AGG unit: as dn_fold_tail_bf16 does): the error
        # is relative to the two terms, not to their (possibly cancelling) sum
        prod = (aux.cpu().double() @ Wa.double()).abs()
        errd = (out.cpu().double() - ref)[dummies].abs() / (ref[dummies].abs() + prod + 1.0)
        assert float(errd.max()) < 8e-3, (w_kn, float(errd.max()))
    # a graph over 32 nodes: the verdict is "no" and the caller keeps the partial rows + tail
    big_ptr, big_nodes = torch.tensor([0, 40], device=DEV, dtype=torch.int32), torch.arange(40, device=DEV, dtype=torch.int32)
    assert int(ops.build_graph_tiles(big_ptr, big_nodes, 41)[2].item()) == 0
    # a target row outside its segment's own block (the dummy node of graph j stored by another tile's workgroup): "no" as well --
    # the AGG unit is a read-modify-write of that row by the workgroup that owns the tile
    for bad in (np.roll(dummies, 1), np.full_like(dummies, N - 1), np.maximum(dummies - 32, 0)):
        if S > 1 and not all(tpr[j] <= bad[j] < tpr[j + 1] for j in range(S)):
            assert not graph_tiles_ref(seg_ptr, seg_nodes, N, add_idx=bad)[0]
            assert int(ops.build_graph_tiles(sp, sn, N, add_idx=torch.from_numpy(bad).to(DEV).int())[2].item()) == 0


@pytest.mark.parametrize("G,K", [(256, 1), (8, 3), (5, 2), (1, 1), (48, 4), (16, 1)])
@pytest.mark.parametrize("sizes_kind", ["tu", "huge", "small"])
def test_absorbed_fold_over_graphs_that_span_several_tiles(sizes_kind, G, K):
    """TU-shaped batches (graphs of 1 .. 700 nodes; tu_data_processing.py:179-218 keeps whatever sizes a dataset has): every
    workgroup takes a CHUNK of graphs whose nodes are cut into consecutive 32-node tiles (order 2).  Tables against the host restatement, the
    launch -- out = x W_loop + b + list rows, aux[j] = bf16 column sum of graph j's real nodes accumulated ACROSS its tiles,
    out[dummy_j] += aux[j] W_agg -- against fp64, bitwise run to run, and, on graphs within one tile, bit-identical to the
    single-tile tables of round 4 up to summation order."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(G + len(sizes_kind))
    if sizes_kind == "tu":
        sizes = [int(np.clip(rng.lognormal(3.4, 0.7), 1, 700)) for _ in range(150)] + [700, 1, 31, 32, 33, 63, 64, 65]
    elif sizes_kind == "huge":
        sizes = [1500, 3, 900]
    else:
        sizes = list(rng.integers(1, 32, size=120))
    H, P = 256, 4000
    lists, ptr, rows, seg_ptr, seg_nodes, dummies, N = _graph_batch(rng, sizes, P)
    S = len(sizes)
    sp, sn = torch.from_numpy(seg_ptr).to(DEV).int(), torch.from_numpy(seg_nodes).to(DEV).int()
    tgt = torch.from_numpy(dummies).to(DEV).int()
    tile_ptr, info, chunk_tile, chunk_graph, cap, ok = ops.build_graph_tiles_multi(sp, sn, N, add_idx=tgt, num_chunks=G * K)
    okr, ctr, cgr, tpr, infor = graph_tiles_multi_ref(seg_ptr, seg_nodes, N, G * K, add_idx=dummies)
    assert okr and int(ok.item()) != 0
    T = int(ctr[-1])
    assert T <= cap and T <= N // 32 + G * K and np.array_equal(chunk_tile.cpu().numpy(), ctr) and np.array_equal(chunk_graph.cpu().numpy(), cgr)
    assert np.array_equal(tile_ptr.cpu().numpy()[:T + 1], tpr) and np.array_equal(info.cpu().numpy()[:T], infor)
    lp, lr = torch.from_numpy(ptr).to(DEV).int(), torch.from_numpy(rows).to(DEV).int()
    cu = ops.build_close_units(lp, lr, N, P, num_wg=G, tile_ptr=tile_ptr, agg=True, multi=(chunk_tile, chunk_graph, cap, S))
    assert cu.order == (3 if G % 8 == 0 else 2) and cu.num_segments == S
    up, un, tiles = close_units_ref(ptr, rows, N, P, G, tile_ptr=tpr, agg=True, order=cu.order, multi=(ctr, cgr))
    assert np.array_equal(cu.unit_ptr.cpu().numpy(), up)
    assert np.array_equal(cu.units.cpu().numpy()[:len(un)], un)
    er, em = cu.ent_row.cpu().numpy(), cu.ent_mask.cpu().numpy().view(np.uint32)
    for t, (e0, r, m) in tiles.items():
        assert list(er[e0:e0 + len(r)]) == r and list(em[e0:e0 + len(r)]) == m, t
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)  # noqa: E731
    x, Y = bf(rng.standard_normal((N, H))), bf(rng.standard_normal((P, H)))
    W, Wa = bf(rng.standard_normal((H, H)) / np.sqrt(H)), bf(rng.standard_normal((H, H)) / np.sqrt(H))
    b = bf(rng.standard_normal(H))
    aux_ref = torch.stack([x[seg_nodes[seg_ptr[j]:seg_ptr[j + 1]]].double().sum(0) for j in range(S)])
    outs = []
    for w_kn in (True, False, True):
        aux = torch.full((S, H), float("nan"), dtype=torch.bfloat16, device=DEV)
        Wd, Wad = (W.to(DEV), Wa.to(DEV)) if w_kn else (W.t().contiguous().to(DEV), Wa.t().contiguous().to(DEV))
        out = ops.rows_close(x.to(DEV), Wd, b.to(DEV), Y.to(DEV), cu, w_kn=w_kn, agg=(info, Wad, aux, tgt))
        outs.append((out.clone(), aux.clone()))
        # (a 700-node sum of N(0,1) rows is ~26 wide: one bf16 rounding of it is 2^-9 relative)
        assert float((aux.cpu().double() - aux_ref).abs().max() / aux_ref.abs().max()) < 5e-3
        ref = _ref_close(x, W, b, Y, lists, P)
        ref[dummies] += aux.cpu().double() @ Wa.double()
        err = (out.cpu().double() - ref).abs() / (ref.abs() + 1.0)
        real = np.setdiff1d(np.arange(N), dummies)
        assert float(err[real].max()) < 8e-3, (w_kn, float(err[real].max()))
        prod = (aux.cpu().double() @ Wa.double()).abs()
        errd = (out.cpu().double() - ref)[dummies].abs() / (ref[dummies].abs() + prod + 1.0)
        assert float(errd.max()) < 8e-3, (w_kn, float(errd.max()))
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])       # bitwise run to run
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])       # ... and in both weight layouts
    if sizes_kind == "small":                              # every graph inside one tile: the single-tile tables of round 4 agree
        tp1, info1, ok1 = ops.build_graph_tiles(sp, sn, N, add_idx=tgt)        # (other tiles: same sums up to fp32 summation order)
        assert int(ok1.item()) != 0
        cu1 = ops.build_close_units(lp, lr, N, P, num_wg=G, tile_ptr=tp1, agg=True, order=0)
        aux1 = torch.empty((S, H), dtype=torch.bfloat16, device=DEV)
        out1 = ops.rows_close(x.to(DEV), W.to(DEV), b.to(DEV), Y.to(DEV), cu1, w_kn=True, agg=(info1, Wa.to(DEV), aux1, tgt))
        assert float((aux1.double() - outs[0][1].double()).abs().max() / aux_ref.abs().max()) < 4e-3
        assert float(((out1.double() - outs[0][0].double()).abs() / (out1.double().abs() + 1.0)).max()) < 2e-2
    # a target row outside its segment's own block: "no", as for the single-tile tables
    if S > 1:
        bad = np.roll(dummies, 1)
        assert not graph_tiles_multi_ref(seg_ptr, seg_nodes, N, G * K, add_idx=bad)[0]
        assert int(ops.build_graph_tiles_multi(sp, sn, N, add_idx=torch.from_numpy(bad).to(DEV).int(), num_chunks=G * K)[5].item()) == 0
