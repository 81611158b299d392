"""GPU parity (bit-exact) of the device-side index builds against the golden fixtures captured from the
reference and against the oracle on larger random batches.  All calls go through the C ABI."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import transforms as OT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _T():
    from dummynode4graphlearning_amd import transforms
    return transforms


def _d(a):
    return torch.as_tensor(np.asarray(a, dtype=np.int64)).to(DEV)


def _eq(got, ref, msg=""):
    np.testing.assert_array_equal(got.cpu().numpy().astype(np.int64), np.asarray(ref, dtype=np.int64), err_msg=msg)


def _check_conj(cj, ref, msg):
    for k in ("cnode_ptr", "cedge_ptr", "csrc", "cdst", "rep_edge", "shared_node"):
        _eq(cj[k], ref[k], msg + " " + k)


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _batch_from_dumps(dumps):
    node_ptr, edge_ptr, src, dst = [0], [0], [], []
    for d in dumps:
        base = node_ptr[-1]
        src.extend(base + u for u, _ in d["edges"])
        dst.extend(base + v for _, v in d["edges"])
        node_ptr.append(base + d["vcount"])
        edge_ptr.append(len(src))
    return np.array(node_ptr), np.array(edge_ptr), np.array(src, dtype=np.int64), np.array(dst, dtype=np.int64)


def test_gc_pipeline_matches_reference_goldens(golden_dir):
    T = _T()
    for case in _load(golden_dir, "gc_transforms.json"):
        raw = OT.tu_raw_to_batch(case["A"], case["graph_indicator"], case["node_labels"], case["edge_labels"])
        aug = T.dummy_augment_gc(*(_d(raw[k]) for k in ("node_ptr", "edge_ptr", "src", "dst", "node_label", "edge_label")))
        # against the reference's dumped igraph objects
        rn, re, rs, rd = _batch_from_dumps(case["dummy"])
        _eq(aug["node_ptr"], rn), _eq(aug["edge_ptr"], re), _eq(aug["src"], rs), _eq(aug["dst"], rd)
        _eq(aug["node_label"], sum((d["v_LABEL"] for d in case["dummy"]), []))
        _eq(aug["edge_label"], sum((d["e_LABEL"] for d in case["dummy"]), []))
        _eq(aug["is_dummy_edge"], sum((d["e_IS_DUMMY"] for d in case["dummy"]), []))
        _eq(aug["is_dummy_node"], sum((d["v_IS_DUMMY"] for d in case["dummy"]), []))
        _eq(aug["edge_id"], sum((d["e_ID"] for d in case["dummy"]), []))
        _eq(aug["node_id"], sum((d["v_ID"] for d in case["dummy"]), []))
        for tag, b, mode in (("plain", {k: _d(v) for k, v in raw.items()}, "line"), ("dummy", aug, "gc")):
            cj = T.conjugate(b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], b["node_label"],
                             is_dummy_edge=b.get("is_dummy_edge"), mode=mode)
            rn, re, rs, rd = _batch_from_dumps(case[tag + "_conj"])
            msg = "%s %s" % (case["name"], tag)
            _eq(cj["cnode_ptr"], rn, msg), _eq(cj["cedge_ptr"], re, msg), _eq(cj["csrc"], rs, msg), _eq(cj["cdst"], rd, msg)
            el = b["edge_label"].cpu().numpy()
            nl = b["node_label"].cpu().numpy()
            np.testing.assert_array_equal(el[cj["rep_edge"].cpu().numpy()],
                                          sum((d["v_LABEL"] for d in case[tag + "_conj"]), []), err_msg=msg)
            np.testing.assert_array_equal(nl[cj["shared_node"].cpu().numpy()],
                                          sum((d.get("e_LABEL", []) for d in case[tag + "_conj"]), []), err_msg=msg)


def _dgl_batch(items, key):
    node_ptr, edge_ptr, src, dst, nd, ed = [0], [0], [], [], {}, {}
    for x in items:
        d = x[key]
        base = node_ptr[-1]
        src.extend(base + u for u in d["u"])
        dst.extend(base + v for v in d["v"])
        node_ptr.append(base + d["num_nodes"])
        edge_ptr.append(len(src))
        for k, v in d.items():
            if k.startswith("n_"):
                nd.setdefault(k[2:], []).extend(v)
            elif k.startswith("e_"):
                ed.setdefault(k[2:], []).extend(v)
    return dict(node_ptr=node_ptr, edge_ptr=edge_ptr, src=src, dst=dst, n=nd, e=ed)


def test_si_pipeline_matches_reference_goldens(golden_dir):
    T = _T()
    g = _load(golden_dir, "si_transforms.json")
    vocab = g["vocab"]
    for key, mv, mvl, me, mel in (("graph", "max_ngv", "max_ngvl", "max_nge", "max_ngel"),
                                  ("pattern", "max_npv", "max_npvl", "max_npe", "max_npel")):
        b, a = _dgl_batch(g["before"], key), _dgl_batch(g["after"], key)
        aug = T.dummy_augment_si(_d(b["node_ptr"]), _d(b["edge_ptr"]), _d(b["src"]), _d(b["dst"]), _d(b["n"]["id"]),
                                 _d(b["n"]["label"]), _d(b["e"].get("id", [])), _d(b["e"].get("label", [])),
                                 vocab[mv], vocab[mvl], vocab[me], vocab[mel])
        for k in ("node_ptr", "edge_ptr", "src", "dst"):
            _eq(aug[k], a[k], key + " " + k)
        _eq(aug["node_id"], a["n"]["id"]), _eq(aug["node_label"], a["n"]["label"])
        _eq(aug["is_dummy_node"], a["n"]["is_dummy"]), _eq(aug["edge_id"], a["e"]["id"])
        _eq(aug["edge_label"], a["e"]["label"]), _eq(aug["is_dummy_edge"], a["e"]["is_dummy"])
        _eq(aug["is_reversed"], a["e"]["is_reversed"])
        cj = T.conjugate(aug["node_ptr"], aug["edge_ptr"], aug["src"], aug["dst"], aug["node_label"],
                         edge_id=aug["edge_id"], mode="si")
        rn, re, rs, rd = _batch_from_dumps([row[key] for row in g["conj"]])
        _eq(cj["cnode_ptr"], rn, key), _eq(cj["cedge_ptr"], re, key), _eq(cj["csrc"], rs, key), _eq(cj["cdst"], rd, key)
        eid = aug["edge_id"].cpu().numpy()
        np.testing.assert_array_equal(eid[cj["rep_edge"].cpu().numpy()], sum((row[key]["v_id"] for row in g["conj"]), []))
        nl = aug["node_label"].cpu().numpy()
        np.testing.assert_array_equal(nl[cj["shared_node"].cpu().numpy()],
                                      sum((row[key].get("e_label", []) for row in g["conj"]), []))
    # KAT-2 literal
    i, o = g["kat2"]["in"], g["kat2"]["out"]
    cj = T.conjugate(_d([0, 5]), _d([0, len(i["edges"])]), _d([e[0] for e in i["edges"]]), _d([e[1] for e in i["edges"]]),
                     _d(i["v_label"]), edge_id=_d(i["e_id"]), mode="si")
    assert [list(x) for x in zip(cj["csrc"].tolist(), cj["cdst"].tolist())] == o["edges"]


def _random_batch(rng, G, max_n, multi=True, empty_frac=0.1):
    node_ptr, edge_ptr, src, dst = [0], [0], [], []
    for g in range(G):
        n = int(rng.integers(0 if rng.random() < 0.05 else 1, max_n + 1))
        m = 0 if (n == 0 or rng.random() < empty_frac) else int(rng.integers(1, 3 * n + 1))
        base = node_ptr[-1]
        if m > 0:
            src.extend((base + rng.integers(0, n, size=m)).tolist())
            dst.extend((base + rng.integers(0, n, size=m)).tolist())
        node_ptr.append(base + n)
        edge_ptr.append(len(src))
    N, E = node_ptr[-1], len(src)
    return dict(node_ptr=np.array(node_ptr), edge_ptr=np.array(edge_ptr), src=np.array(src, dtype=np.int64),
                dst=np.array(dst, dtype=np.int64), node_label=rng.integers(1, 5, size=N),
                edge_label=rng.integers(1, 4, size=E))


@pytest.mark.parametrize("seed,G,max_n", [(0, 1, 6), (1, 64, 14), (2, 400, 30)])
def test_gc_random_batches_match_oracle(seed, G, max_n):
    T = _T()
    rng = np.random.default_rng(seed)
    b = _random_batch(rng, G, max_n)
    ref = OT.dummy_augment_gc(b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], b["node_label"], b["edge_label"])
    got = T.dummy_augment_gc(*(_d(b[k]) for k in ("node_ptr", "edge_ptr", "src", "dst", "node_label", "edge_label")))
    for k in ref:
        _eq(got[k], ref[k], k)
    for mode, bb, dflag in (("gc", ref, ref["is_dummy_edge"]), ("line", b, None)):
        rcj = OT.conjugate(bb["node_ptr"], bb["edge_ptr"], bb["src"], bb["dst"], bb["node_label"], is_dummy_edge=dflag, mode=mode)
        gcj = T.conjugate(_d(bb["node_ptr"]), _d(bb["edge_ptr"]), _d(bb["src"]), _d(bb["dst"]), _d(bb["node_label"]),
                          is_dummy_edge=None if dflag is None else _d(dflag), mode=mode)
        _check_conj(gcj, rcj, "seed %d mode %s" % (seed, mode))


@pytest.mark.parametrize("seed,G,max_n", [(3, 1, 5), (4, 50, 12), (5, 300, 20)])
def test_si_random_batches_match_oracle(seed, G, max_n):
    T = _T()
    rng = np.random.default_rng(seed)
    b = _random_batch(rng, G, max_n)
    N, E = b["node_ptr"][-1], len(b["src"])
    nid = np.arange(N) - np.repeat(b["node_ptr"][:-1], np.diff(b["node_ptr"]))
    eid = np.arange(E) - np.repeat(b["edge_ptr"][:-1], np.diff(b["edge_ptr"]))
    # duplicate some edge ids inside graphs (reversed-edge style sharing) to exercise the vertex merge
    dup = rng.random(E) < 0.15
    eid = np.where(dup, np.maximum(eid - 1, 0), eid)
    rev = (rng.random(E) < 0.3).astype(np.int64)
    args = (b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], nid, b["node_label"], eid, b["edge_label"])
    ref = OT.dummy_augment_si(*args, 99, 7, 500, 9, is_reversed=rev)
    got = T.dummy_augment_si(*(_d(a) for a in args), 99, 7, 500, 9, is_reversed=_d(rev))
    for k in ref:
        _eq(got[k], ref[k], k)
    rcj = OT.conjugate(ref["node_ptr"], ref["edge_ptr"], ref["src"], ref["dst"], ref["node_label"],
                       edge_id=ref["edge_id"], mode="si")
    gcj = T.conjugate(got["node_ptr"], got["edge_ptr"], got["src"], got["dst"], got["node_label"],
                      edge_id=got["edge_id"], mode="si")
    _check_conj(gcj, rcj, "seed %d" % seed)


def test_conjugate_of_empty_batch():
    T = _T()
    z = _d([])
    cj = T.conjugate(_d([0, 3, 3]), _d([0, 0, 0]), z, z, _d([1, 1, 1]), mode="line")
    assert cj["cnode_ptr"].tolist() == [0, 0, 0] and cj["csrc"].numel() == 0


def test_full_size_config5_dummy_augmentation_and_conjugate():
    """BASELINE config 5 at full size (32,768 graphs -> N = 1,015,808, E = 3,997,696): device build == C oracle,
    bit for bit; the conjugate transform of the augmented batch on a quarter of it (9.6 M raw 2-paths)."""
    import time
    from dummynode4graphlearning_amd import synthetic
    from oracle import c_oracle as OC
    T = _T()
    raw = synthetic.config5()
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    vocab = (raw["max_nv"], raw["max_nvl"], raw["max_ne"], raw["max_nel"])
    ref = OC.dummy_augment_si(*(raw[k] for k in keys), *vocab)
    dev_in = [_d(raw[k]) for k in keys]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = T.dummy_augment_si(*dev_in, *vocab)
    torch.cuda.synchronize()
    print("dummy augmentation of config 5 on the device: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    assert got["src"].numel() == 3997696 and got["node_label"].numel() == 1015808
    for k in ref:
        _eq(got[k], ref[k], k)
    # conjugate (SI rule: dummy edges share two ids) of the first 8192 graphs
    G = 8192
    n1, e1 = int(ref["node_ptr"][G]), int(ref["edge_ptr"][G])
    sub = dict(node_ptr=ref["node_ptr"][:G + 1], edge_ptr=ref["edge_ptr"][:G + 1], src=ref["src"][:e1], dst=ref["dst"][:e1],
               node_label=ref["node_label"][:n1], edge_id=ref["edge_id"][:e1])
    rcj = OC.conjugate(sub["node_ptr"], sub["edge_ptr"], sub["src"], sub["dst"], sub["node_label"], edge_id=sub["edge_id"], mode="si")
    dsub = {k: _d(v) for k, v in sub.items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gcj = T.conjugate(dsub["node_ptr"], dsub["edge_ptr"], dsub["src"], dsub["dst"], dsub["node_label"], edge_id=dsub["edge_id"],
                      mode="si")
    torch.cuda.synchronize()
    print("conjugate of %d graphs (%d raw 2-paths -> %d conj edges) on the device: %.2f ms"
          % (G, gcj["num_raw"], gcj["csrc"].numel(), (time.perf_counter() - t0) * 1e3))
    assert gcj["num_raw"] == rcj["num_raw"]
    _check_conj(gcj, rcj, "config5 si")


def test_full_size_proteins_shaped_gc_pipeline():
    """Config 2 shape (512 PROTEINS-like graphs, dummy in-degree up to several hundred): GC augmentation + L_Phi."""
    from dummynode4graphlearning_amd import synthetic
    from oracle import c_oracle as OC
    T = _T()
    raw = synthetic.config2()
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_label", "edge_label")
    ref = OC.dummy_augment_gc(*(raw[k] for k in keys))
    got = T.dummy_augment_gc(*(_d(raw[k]) for k in keys))
    for k in ref:
        _eq(got[k], ref[k], k)
    rcj = OC.conjugate(ref["node_ptr"], ref["edge_ptr"], ref["src"], ref["dst"], ref["node_label"],
                       is_dummy_edge=ref["is_dummy_edge"], mode="gc")
    gcj = T.conjugate(got["node_ptr"], got["edge_ptr"], got["src"], got["dst"], got["node_label"],
                      is_dummy_edge=got["is_dummy_edge"], mode="gc")
    _check_conj(gcj, rcj, "config2 gc")
    # idempotence-style property at full size: every real edge becomes exactly one conj vertex, plus one Phi per graph
    nreal = int((ref["is_dummy_edge"] == 0).sum())
    ngraphs_with_dummy = int((np.diff(ref["node_ptr"]) > 1).sum())
    assert gcj["rep_edge"].numel() == nreal + ngraphs_with_dummy


# ------------------------------------------------------------------------------------------------ f-3: TU files
def _materialise_tu(tmp_path, case):
    raw = os.path.join(str(tmp_path), case["name"], "raw")
    os.makedirs(raw)
    for fn, text in case["inputs"].items():
        with open(os.path.join(raw, fn), "w") as f:
            f.write(text)
    return raw


def test_process_dataset_writes_the_reference_files(golden_dir, tmp_path):
    """raw TU text -> device batch -> HIP dummy augmentation / conjugate -> DUMMY_/LINE_/CONJ_ files, byte for byte the files
    the reference's tu_data_processing.py wrote (tests/golden/tu_files.json)."""
    from dummynode4graphlearning_amd import tu_io
    with open(os.path.join(golden_dir, "tu_files.json")) as f:
        cases = json.load(f)
    for case in cases:
        raw = _materialise_tu(tmp_path, case)
        dirs = tu_io.process_dataset(raw, case["name"])
        assert sorted(dirs) == ["CONJ_", "DUMMY_", "LINE_"]
        for rel, text in case["outputs"].items():
            with open(os.path.join(str(tmp_path), rel)) as f:
                assert f.read() == text, rel


def test_read_tu_data_and_dataset_match_the_oracle(golden_dir, tmp_path):
    from dummynode4graphlearning_amd import tu_io
    from oracle import tu_format as TF
    with open(os.path.join(golden_dir, "tu_files.json")) as f:
        cases = json.load(f)
    for case in cases:
        raw = _materialise_tu(tmp_path, case)
        tu_io.process_dataset(raw, case["name"])
        for pre in ("DUMMY_", "CONJ_", "LINE_", ""):
            d = raw.replace(case["name"], pre + case["name"])
            data, slices = tu_io.read_tu_data(d, pre + case["name"])
            rdata, rslices = TF.read_tu_data(d, pre + case["name"])
            for k in ("x", "edge_index", "edge_attr", "y"):
                a, b = getattr(data, k), rdata[k]
                assert (a is None) == (b is None), (pre, k)
                if a is not None:
                    assert np.array_equal(a.cpu().numpy(), b), (pre, k)
            assert sorted(slices) == sorted(rslices)
            for k in slices:
                assert np.array_equal(slices[k].cpu().numpy(), rslices[k]), (pre, k)


def test_pyg_dataset_mirror_feeds_the_models(tmp_path):
    """PYGDataset over process_dataset output: dummy flags, label widths, and a GIN forward on a collated mini-batch."""
    from types import SimpleNamespace
    from dummynode4graphlearning_amd import graph_classification as GC, tu_io
    from oracle import tu_format as TF
    rng = np.random.default_rng(5)
    name, G = "TOYS", 12
    raw = os.path.join(str(tmp_path), name, "raw")
    os.makedirs(raw)
    A, gi, nl, base = [], [], [], 0
    for g in range(G):
        n = int(rng.integers(3, 9))
        pairs = {(int(u), int(v)) for u, v in rng.integers(0, n, size=(3 * n, 2)) if u != v}
        for u, v in sorted(pairs):
            A.append((base + u + 1, base + v + 1))
        gi += [g + 1] * n
        nl += [int(x) for x in rng.integers(0, 3, size=n)]
        base += n
    nl[0] = 0
    wr = lambda fn, rows: open(os.path.join(raw, name + "_" + fn + ".txt"), "w").write("".join(r + "\n" for r in rows))  # noqa: E731
    wr("A", ["%d, %d" % e for e in A]); wr("graph_indicator", map(str, gi)); wr("node_labels", map(str, nl))
    wr("graph_labels", [str(int(x)) for x in rng.integers(0, 2, size=G)])
    tu_io.process_dataset(raw, name)
    ds = tu_io.PYGDataset(str(tmp_path), name, add_dummy=True)
    assert len(ds) == G and ds.num_node_labels == 4 and ds.num_edge_labels == 2 and ds.num_features == 4
    b = TF.load_graph_data(TF.parse_tu_dir(raw), with_dummy=True)
    assert np.array_equal(ds.data.is_dummy_node.cpu().numpy(), b["is_dummy_node"].astype(bool))
    assert int(ds.data.is_dummy_edge.sum()) == int(b["is_dummy_edge"].sum())          # simple graphs: nothing coalesced away
    batch = ds.batch(range(5))
    assert batch.num_graphs == 5 and batch.x.is_cuda and batch.batch.is_cuda
    args = SimpleNamespace(num_features=ds.num_features, hidden_dim=64, num_classes=ds.num_classes, dropout_ratio=0.0,
                           additional={"num_layers": 2}, epochs=1, device=DEV, dummy_weight=0)
    torch.manual_seed(0)
    out = GC.GIN(args).to(DEV)(batch)
    assert out.shape == (5, ds.num_classes) and bool(torch.isfinite(out).all())


# ------------------------------------------------------------------------------------------------ f-2: SI bookkeeping
def test_si_bookkeeping_matches_reference_goldens(golden_dir):
    from dummynode4graphlearning_amd.subgraph_isomorphism import bookkeeping as BK
    with open(os.path.join(golden_dir, "si_bookkeeping.json")) as f:
        gold = json.load(f)
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int64)).to(DEV)  # noqa: E731
    for c in gold["cases"]:
        a = [t(c[k]) for k in ("p_u", "p_v", "p_el", "g_u", "g_v", "g_el")] + [t(c["subisomorphisms"])]
        assert BK.get_conjugate_subisomorphisms(*a).cpu().tolist() == c["conj_subisomorphisms"]
        assert BK.compute_edgeseq_subisoweights(*a).cpu().tolist() == c["edgeseq_subisoweights"]
        assert BK.compute_nodeseq_subisoweights(c["num_nodes"], a[6]).cpu().tolist() == c["nodeseq_subisoweights"]
        for sl in (True, False):
            nn_, en_ = BK.compute_norm(a[3], a[4], c["num_nodes"], sl)
            assert np.array_equal(nn_.view(-1).cpu().numpy(), np.asarray(c["node_norm_%d" % sl], dtype=np.float32))
            assert np.array_equal(en_.view(-1).cpu().numpy(), np.asarray(c["edge_norm_%d" % sl], dtype=np.float32))
        ne, ee = BK.compute_largest_eigenvalues(a[3], a[4], c["num_nodes"])
        assert (float(ne), float(ee)) == (c["node_eigenv"], c["edge_eigenv"])
    # add_reversed_edges: the four samples as ONE batch per side, compared with the per-sample reference output
    voc = gold["reversed"]["vocab"]
    for side, mne, mnel in (("pattern", voc["max_npe"], voc["max_npel"]), ("graph", voc["max_nge"], voc["max_ngel"])):
        bef = [x[side] for x in gold["reversed"]["before"]]
        aft = [x[side] for x in gold["reversed"]["after"]]
        ep = np.concatenate([[0], np.cumsum([len(b["u"]) for b in bef])])
        cat = lambda key, rows: t(np.concatenate([np.asarray(r[key], dtype=np.int64) for r in rows]))  # noqa: E731
        r = BK.add_reversed_edges(t(ep), cat("u", bef), cat("v", bef), cat("e_id", bef), cat("e_label", bef), mne, mnel)
        assert r["edge_ptr"].cpu().tolist() == (2 * ep).tolist()
        for key, gk in (("src", "u"), ("dst", "v"), ("edge_id", "e_id"), ("edge_label", "e_label"), ("is_reversed", "e_is_reversed")):
            assert r[key].cpu().tolist() == cat(gk, aft).cpu().tolist(), (side, key)


def test_si_bookkeeping_large_random_against_oracle():
    """A bigger pair (pattern 6 edges with repeated keys, graph 4k edges, 512 subisomorphisms) against the oracle."""
    from dummynode4graphlearning_amd.subgraph_isomorphism import bookkeeping as BK
    from oracle import si_bookkeeping as OB
    rng = np.random.default_rng(77)
    pn, gn, S_ = 5, 300, 512
    p_u = np.array([0, 0, 1, 1, 3, 0]); p_v = np.array([1, 1, 2, 2, 4, 1]); p_el = np.array([0, 1, 2, 2, 1, 3])
    sub = np.stack([rng.permutation(gn)[:pn] for _ in range(S_)])
    g_u, g_v = rng.integers(0, gn, size=3000), rng.integers(0, gn, size=3000)
    g_el = rng.integers(0, 4, size=3000)
    g_u = np.concatenate([g_u, sub[:, p_u].reshape(-1)]); g_v = np.concatenate([g_v, sub[:, p_v].reshape(-1)])
    g_el = np.concatenate([g_el, np.tile(p_el, S_)])
    o = np.lexsort((g_v, g_u))
    g_u, g_v, g_el = g_u[o], g_v[o], g_el[o]
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int64)).to(DEV)  # noqa: E731
    a = [t(x) for x in (p_u, p_v, p_el, g_u, g_v, g_el, sub)]
    assert np.array_equal(BK.get_conjugate_subisomorphisms(*a).cpu().numpy(),
                          OB.conjugate_subisomorphisms(p_u, p_v, p_el, g_u, g_v, g_el, sub))
    assert np.array_equal(BK.compute_edgeseq_subisoweights(*a).cpu().numpy(),
                          OB.edgeseq_subisoweights(p_u, p_v, p_el, g_u, g_v, g_el, sub))
