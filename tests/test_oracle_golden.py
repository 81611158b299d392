"""Pin the oracle (oracle/*.py) against golden vectors captured from the reference's own code
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch as th

from oracle import gc_models as OG
from oracle import layers as OL
from oracle import transforms as OT


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _batch_from_dumps(dumps, vkeys=(), ekeys=()):
    """Concatenate per-graph igraph dumps into the batched layout the oracle uses."""
    node_ptr, edge_ptr, src, dst = [0], [0], [], []
    vattr = {k: [] for k in vkeys}
    eattr = {k: [] for k in ekeys}
    for d in dumps:
        base = node_ptr[-1]
        for u, v in d["edges"]:
            src.append(base + u)
            dst.append(base + v)
        node_ptr.append(base + d["vcount"])
        edge_ptr.append(len(src))
        for k in vkeys:
            vattr[k].extend(d.get("v_" + k, []))
        for k in ekeys:
            eattr[k].extend(d.get("e_" + k, []))
    return dict(node_ptr=np.array(node_ptr), edge_ptr=np.array(edge_ptr), src=np.array(src, dtype=np.int64),
                dst=np.array(dst, dtype=np.int64), v=vattr, e=eattr)


def test_gc_dummy_and_conjugate_match_reference(golden_dir):
    cases = _load(golden_dir, "gc_transforms.json")
    assert cases[0]["name"] == "KAT1_figure"
    for case in cases:
        raw = OT.tu_raw_to_batch(case["A"], case["graph_indicator"], case["node_labels"], case["edge_labels"])
        # plain graphs (with_dummy=False)
        ref = _batch_from_dumps(case["plain"], ("LABEL", "ID"), ("LABEL", "ID"))
        for k in ("node_ptr", "edge_ptr", "src", "dst"):
            np.testing.assert_array_equal(raw[k], ref[k], err_msg=case["name"] + " plain " + k)
        np.testing.assert_array_equal(raw["node_label"], ref["v"]["LABEL"])
        np.testing.assert_array_equal(raw["edge_label"], ref["e"]["LABEL"])
        # a-1 dummy augmentation
        aug = OT.dummy_augment_gc(raw["node_ptr"], raw["edge_ptr"], raw["src"], raw["dst"],
                                  raw["node_label"], raw["edge_label"])
        ref = _batch_from_dumps(case["dummy"], ("LABEL", "ID", "IS_DUMMY"), ("LABEL", "ID", "IS_DUMMY"))
        for k in ("node_ptr", "edge_ptr", "src", "dst"):
            np.testing.assert_array_equal(aug[k], ref[k], err_msg=case["name"] + " dummy " + k)
        np.testing.assert_array_equal(aug["node_label"], ref["v"]["LABEL"])
        np.testing.assert_array_equal(aug["is_dummy_node"], ref["v"]["IS_DUMMY"])
        np.testing.assert_array_equal(aug["node_id"], ref["v"]["ID"])
        np.testing.assert_array_equal(aug["edge_label"], ref["e"]["LABEL"])
        np.testing.assert_array_equal(aug["is_dummy_edge"], ref["e"]["IS_DUMMY"])
        np.testing.assert_array_equal(aug["edge_id"], ref["e"]["ID"])
        # a-2 conjugate of the plain graphs ("line") and of the dummy graphs ("gc")
        for tag, b, mode, dflag in (("plain", raw, "line", None), ("dummy", aug, "gc", aug["is_dummy_edge"])):
            cj = OT.conjugate(b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], b["node_label"],
                              is_dummy_edge=dflag, mode=mode)
            ref = _batch_from_dumps(case[tag + "_conj"], ("LABEL", "ID", "IS_DUMMY"), ("LABEL", "ID", "IS_DUMMY"))
            msg = case["name"] + " conj " + tag
            np.testing.assert_array_equal(cj["cnode_ptr"], ref["node_ptr"], err_msg=msg)
            np.testing.assert_array_equal(cj["cedge_ptr"], ref["edge_ptr"], err_msg=msg)
            np.testing.assert_array_equal(cj["csrc"], ref["src"], err_msg=msg)
            np.testing.assert_array_equal(cj["cdst"], ref["dst"], err_msg=msg)
            # conj-vertex attrs = attrs of the representative edge; conj-edge attrs = attrs of the shared vertex
            np.testing.assert_array_equal(b["edge_label"][cj["rep_edge"]], ref["v"]["LABEL"], err_msg=msg)
            local_eid = np.arange(len(b["src"])) - np.repeat(b["edge_ptr"][:-1], np.diff(b["edge_ptr"]))
            np.testing.assert_array_equal(local_eid[cj["rep_edge"]], ref["v"]["ID"], err_msg=msg)
            np.testing.assert_array_equal(b["node_label"][cj["shared_node"]], ref["e"]["LABEL"], err_msg=msg)
            local_nid = np.arange(b["node_ptr"][-1]) - np.repeat(b["node_ptr"][:-1], np.diff(b["node_ptr"]))
            np.testing.assert_array_equal(local_nid[cj["shared_node"]], ref["e"]["ID"], err_msg=msg)
            if tag == "dummy":
                np.testing.assert_array_equal(aug["is_dummy_edge"][cj["rep_edge"]], ref["v"]["IS_DUMMY"], err_msg=msg)
                np.testing.assert_array_equal(aug["is_dummy_node"][cj["shared_node"]], ref["e"]["IS_DUMMY"], err_msg=msg)


def test_kat1_figure_literal(golden_dir):
    """The paper's figure, spelled out."""
    raw = OT.tu_raw_to_batch([(2, 1), (1, 3), (1, 4)], [1, 1, 1, 1], [1, 2, 3, 4], [1, 2, 3])
    aug = OT.dummy_augment_gc(raw["node_ptr"], raw["edge_ptr"], raw["src"], raw["dst"], raw["node_label"], raw["edge_label"])
    assert list(zip(aug["src"], aug["dst"])) == [(1, 0), (0, 2), (0, 3), (4, 0), (0, 4), (4, 1), (1, 4), (4, 2), (2, 4), (4, 3), (3, 4)]
    cj = OT.conjugate(aug["node_ptr"], aug["edge_ptr"], aug["src"], aug["dst"], aug["node_label"],
                      is_dummy_edge=aug["is_dummy_edge"], mode="gc")
    assert cj["cnode_ptr"].tolist() == [0, 4]
    assert list(zip(cj["csrc"], cj["cdst"])) == [(3, 0), (0, 1), (3, 1), (0, 2), (3, 2), (0, 3), (1, 3), (2, 3)]
    assert aug["node_label"][cj["shared_node"]].tolist() == [2, 1, 1, 1, 1, 1, 3, 4]
    assert aug["edge_label"][cj["rep_edge"]].tolist() == [1, 2, 3, 0]


def _dgl_batch(items, key):
    node_ptr, edge_ptr, src, dst = [0], [0], [], []
    nd = {}
    ed = {}
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
    return dict(node_ptr=np.array(node_ptr), edge_ptr=np.array(edge_ptr), src=np.array(src, dtype=np.int64),
                dst=np.array(dst, dtype=np.int64), n=nd, e=ed)


def test_si_dummy_and_conjugate_match_reference(golden_dir):
    g = _load(golden_dir, "si_transforms.json")
    vocab = g["vocab"]
    for key, mv, mvl, me, mel in (("graph", "max_ngv", "max_ngvl", "max_nge", "max_ngel"),
                                  ("pattern", "max_npv", "max_npvl", "max_npe", "max_npel")):
        b = _dgl_batch(g["before"], key)
        a = _dgl_batch(g["after"], key)
        aug = OT.dummy_augment_si(b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], b["n"]["id"], b["n"]["label"],
                                  b["e"].get("id", []), b["e"].get("label", []),
                                  vocab[mv], vocab[mvl], vocab[me], vocab[mel])
        for k in ("node_ptr", "edge_ptr", "src", "dst"):
            np.testing.assert_array_equal(aug[k], a[k], err_msg=key + " " + k)
        np.testing.assert_array_equal(aug["node_id"], a["n"]["id"])
        np.testing.assert_array_equal(aug["node_label"], a["n"]["label"])
        np.testing.assert_array_equal(aug["is_dummy_node"], a["n"]["is_dummy"])
        np.testing.assert_array_equal(aug["edge_id"], a["e"]["id"])
        np.testing.assert_array_equal(aug["edge_label"], a["e"]["label"])
        np.testing.assert_array_equal(aug["is_dummy_edge"], a["e"]["is_dummy"])
        np.testing.assert_array_equal(aug["is_reversed"], a["e"]["is_reversed"])
        # a-5 conjugate (igraph branch) of augmented and plain graphs
        for tag, bb, eid, nl in (("conj", aug, aug["edge_id"], aug["node_label"]),
                                 ("conj_plain", b, np.array(b["e"].get("id", []), dtype=np.int64),
                                  np.array(b["n"]["label"]))):
            cj = OT.conjugate(bb["node_ptr"], bb["edge_ptr"], bb["src"], bb["dst"], nl, edge_id=eid, mode="si")
            ref = _batch_from_dumps([row[key] for row in g[tag]], ("id", "label"), ("id", "label"))
            msg = "%s %s" % (key, tag)
            np.testing.assert_array_equal(cj["cnode_ptr"], ref["node_ptr"], err_msg=msg)
            np.testing.assert_array_equal(cj["cedge_ptr"], ref["edge_ptr"], err_msg=msg)
            np.testing.assert_array_equal(cj["csrc"], ref["src"], err_msg=msg)
            np.testing.assert_array_equal(cj["cdst"], ref["dst"], err_msg=msg)
            el = aug["edge_label"] if tag == "conj" else np.array(b["e"].get("label", []), dtype=np.int64)
            nid = aug["node_id"] if tag == "conj" else np.array(b["n"]["id"])
            np.testing.assert_array_equal(eid[cj["rep_edge"]], ref["v"]["id"], err_msg=msg)
            np.testing.assert_array_equal(el[cj["rep_edge"]], ref["v"]["label"], err_msg=msg)
            np.testing.assert_array_equal(nid[cj["shared_node"]], ref["e"]["id"], err_msg=msg)
            np.testing.assert_array_equal(nl[cj["shared_node"]], ref["e"]["label"], err_msg=msg)


def test_kat2_literal(golden_dir):
    g = _load(golden_dir, "si_transforms.json")["kat2"]
    i, o = g["in"], g["out"]
    u = np.array([e[0] for e in i["edges"]])
    v = np.array([e[1] for e in i["edges"]])
    cj = OT.conjugate([0, 5], [0, len(u)], u, v, i["v_label"], edge_id=i["e_id"], mode="si")
    assert cj["cnode_ptr"].tolist() == [0, 5]
    assert [list(x) for x in zip(cj["csrc"].tolist(), cj["cdst"].tolist())] == o["edges"]
    assert len(o["edges"]) == 13
    assert np.array(i["e_id"])[cj["rep_edge"]].tolist() == [0, 1, 2, 20, 21]
    assert np.array(i["v_label"])[cj["shared_node"]].tolist() == o["e_label"]


def _layer_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "si_layers.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def test_si_layers_match_reference_outputs_and_grads(golden_dir):
    z, meta = _layer_cases(golden_dir)
    assert len(meta) >= 40
    for m in meta:
        tag = m["tag"]
        p = {}
        for k in z.files:
            if k.startswith(tag + "/param/"):
                p[k[len(tag) + 7:]] = th.from_numpy(z[k]).clone().requires_grad_(True)
        x = th.from_numpy(z[tag + "/x"]).clone().requires_grad_(True)
        u, v, t = (th.from_numpy(z[tag + "/" + k]) for k in ("u", "v", "t"))
        if m["kind"] == "rgin":
            out = OL.rgin_layer(x, u, v, t, p, regularizer=m["regularizer"], num_rels=m["num_rels"],
                                num_bases=m["num_bases"], num_mlp_layers=m["num_mlp_layers"], act=m["act_func"])
        else:
            out = OL.rgcn_layer(x, u, v, t, p, regularizer=m["regularizer"], num_rels=m["num_rels"],
                                num_bases=m["num_bases"], edge_norm=m["edge_norm"], act=m["act_func"])
        (out * th.from_numpy(z[tag + "/coef"])).sum().backward()
        # same ops in the same order as the reference -> agreement to fp32 rounding
        th.testing.assert_close(out.detach(), th.from_numpy(z[tag + "/out"]), rtol=1e-5, atol=1e-5, msg=tag)
        th.testing.assert_close(x.grad, th.from_numpy(z[tag + "/grad_x"]), rtol=1e-5, atol=1e-5, msg=tag)
        for k, t_ in p.items():
            ref = z[tag + "/grad/" + k]
            if ref.size == 0:
                continue
            th.testing.assert_close(t_.grad, th.from_numpy(ref), rtol=1e-4, atol=1e-5, msg=tag + " " + k)


def test_si_layers_with_batch_norm_and_other_activations_match_reference(golden_dir):
    """si_layers_bn.npz (round 6): RGINLayer with batch_norm=True in training mode (models/rgin.py:50-57, --rep_rgin_batch_norm) and the
    activations gelu / selu / elu at a matrix-core width -- outputs, every gradient, the BatchNorm buffers after the step."""
    z = np.load(os.path.join(golden_dir, "si_layers_bn.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    assert sum(m["batch_norm"] for m in meta) >= 5 and {m["act_func"] for m in meta} >= {"relu", "leaky_relu", "tanh", "gelu", "selu", "elu"}
    for m in meta:
        tag = m["tag"]
        p = {k[len(tag) + 7:]: th.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith(tag + "/param/")}
        x = th.from_numpy(z[tag + "/x"]).clone().requires_grad_(True)
        u, v, t = (th.from_numpy(z[tag + "/" + k]) for k in ("u", "v", "t"))
        bname = "mlp.1" if m["kind"] == "rgin" else "bn"
        bn = [OL.batch_norm_train(p[bname + ".weight"], p[bname + ".bias"])] if m["batch_norm"] else None
        if m["kind"] == "rgin":
            out = OL.rgin_layer(x, u, v, t, p, regularizer=m["regularizer"], num_rels=m["num_rels"], num_bases=m["num_bases"],
                                num_mlp_layers=m["num_mlp_layers"], act=m["act_func"], mlp_bn=bn)
        else:
            out = OL.rgcn_layer(x, u, v, t, p, regularizer=m["regularizer"], num_rels=m["num_rels"], num_bases=m["num_bases"],
                                edge_norm=m["edge_norm"], act=m["act_func"], bn=bn[0])
        (out * th.from_numpy(z[tag + "/coef"])).sum().backward()
        th.testing.assert_close(out.detach(), th.from_numpy(z[tag + "/out"]), rtol=1e-4, atol=2e-5, msg=tag)
        th.testing.assert_close(x.grad, th.from_numpy(z[tag + "/grad_x"]), rtol=1e-4, atol=2e-5, msg=tag)
        for k, t_ in p.items():
            ref = z[tag + "/grad/" + k]
            if ref.size:
                th.testing.assert_close(t_.grad, th.from_numpy(ref), rtol=2e-4, atol=2e-5, msg=tag + " " + k)
        if m["batch_norm"]:                                             # buffers: momentum 0.1 from (0, 1), one step
            mean, uvar = bn[0].stats
            th.testing.assert_close(0.1 * mean, th.from_numpy(z[tag + "/buffer/%s.running_mean" % bname]), rtol=1e-4, atol=1e-6, msg=tag)
            th.testing.assert_close(0.9 + 0.1 * uvar, th.from_numpy(z[tag + "/buffer/%s.running_var" % bname]), rtol=1e-4, atol=1e-6, msg=tag)
            assert int(z[tag + "/buffer/%s.num_batches_tracked" % bname]) == 1


def test_agg_first_form_equals_reference_formulation(golden_dir):
    """aggregate-then-transform == the reference's per-edge transform (basis, full)."""
    z, meta = _layer_cases(golden_dir)
    done = 0
    for m in meta:
        if m["kind"] != "rgin" or m["regularizer"] != "basis" or m["num_bases"] != -1 or not m["self_loop"]:
            continue
        tag = m["tag"]
        p = {k[len(tag) + 7:]: th.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/param/")}
        x = th.from_numpy(z[tag + "/x"])
        u, v, t = (th.from_numpy(z[tag + "/" + k]) for k in ("u", "v", "t"))
        out = OL.rgin_layer_agg_first(x, u, v, t, p, m["num_rels"], act=m["act_func"], num_mlp_layers=m["num_mlp_layers"])
        th.testing.assert_close(out, th.from_numpy(z[tag + "/out"]), rtol=1e-4, atol=1e-5, msg=tag)
        done += 1
    assert done >= 3


def test_init_bound_matches_reference_weights(golden_dir):
    """Custom Xavier-uniform bound (SI utils/init.py:52-75): the golden initial weights must lie inside
    (-a, a) and fill most of it."""
    z, meta = _layer_cases(golden_dir)
    for m in meta:
        w = z[m["tag"] + "/param/weight"]
        a = OL.xavier_uniform_bound(w.shape, m["act_func"])
        assert np.abs(w).max() <= a * (1 + 1e-6), m["tag"]
        if w.size >= 512:
            assert np.abs(w).max() >= 0.95 * a, m["tag"]


def _random_batch(rng, G, max_n):
    node_ptr, edge_ptr, src, dst = [0], [0], [], []
    for _ in range(G):
        n = int(rng.integers(0 if rng.random() < 0.05 else 1, max_n + 1))
        m = 0 if (n == 0 or rng.random() < 0.1) else int(rng.integers(1, 3 * n + 1))
        base = node_ptr[-1]
        if m:
            src.extend((base + rng.integers(0, n, size=m)).tolist())
            dst.extend((base + rng.integers(0, n, size=m)).tolist())
        node_ptr.append(base + n)
        edge_ptr.append(len(src))
    N, E = node_ptr[-1], len(src)
    return dict(node_ptr=np.array(node_ptr), edge_ptr=np.array(edge_ptr), src=np.array(src, dtype=np.int64),
                dst=np.array(dst, dtype=np.int64), node_label=rng.integers(1, 5, size=N), edge_label=rng.integers(1, 4, size=E))


def test_c_oracle_equals_python_oracle_and_goldens(golden_dir):
    """oracle/dn_oracle.c (used for full-size checks) == oracle/transforms.py (pinned to the reference above)."""
    from oracle import c_oracle as OC
    rng = np.random.default_rng(42)
    for G, max_n in ((1, 5), (40, 12), (150, 25)):
        b = _random_batch(rng, G, max_n)
        args = (b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], b["node_label"], b["edge_label"])
        ref, got = OT.dummy_augment_gc(*args), OC.dummy_augment_gc(*args)
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg="gc " + k)
        for mode, bb, flag in (("gc", ref, ref["is_dummy_edge"]), ("line", b, None)):
            r = OT.conjugate(bb["node_ptr"], bb["edge_ptr"], bb["src"], bb["dst"], bb["node_label"], is_dummy_edge=flag, mode=mode)
            c = OC.conjugate(bb["node_ptr"], bb["edge_ptr"], bb["src"], bb["dst"], bb["node_label"], is_dummy_edge=flag, mode=mode)
            for k in r:
                np.testing.assert_array_equal(c[k], r[k], err_msg="%s %s" % (mode, k))
        N, E = b["node_ptr"][-1], len(b["src"])
        nid = np.arange(N) - np.repeat(b["node_ptr"][:-1], np.diff(b["node_ptr"]))
        eid = np.arange(E) - np.repeat(b["edge_ptr"][:-1], np.diff(b["edge_ptr"]))
        eid = np.where(rng.random(E) < 0.15, np.maximum(eid - 1, 0), eid)
        rev = (rng.random(E) < 0.3).astype(np.int64)
        sargs = (b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], nid, b["node_label"], eid, b["edge_label"], 99, 7, 500, 9)
        ref, got = OT.dummy_augment_si(*sargs, is_reversed=rev), OC.dummy_augment_si(*sargs, is_reversed=rev)
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg="si " + k)
        r = OT.conjugate(ref["node_ptr"], ref["edge_ptr"], ref["src"], ref["dst"], ref["node_label"], edge_id=ref["edge_id"], mode="si")
        c = OC.conjugate(ref["node_ptr"], ref["edge_ptr"], ref["src"], ref["dst"], ref["node_label"], edge_id=ref["edge_id"], mode="si")
        for k in r:
            np.testing.assert_array_equal(c[k], r[k], err_msg="si conj " + k)
    # KAT-1 through the C oracle
    raw = OT.tu_raw_to_batch([(2, 1), (1, 3), (1, 4)], [1, 1, 1, 1], [1, 2, 3, 4], [1, 2, 3])
    aug = OC.dummy_augment_gc(raw["node_ptr"], raw["edge_ptr"], raw["src"], raw["dst"], raw["node_label"], raw["edge_label"])
    cj = OC.conjugate(aug["node_ptr"], aug["edge_ptr"], aug["src"], aug["dst"], aug["node_label"],
                      is_dummy_edge=aug["is_dummy_edge"], mode="gc")
    assert list(zip(cj["csrc"], cj["cdst"])) == [(3, 0), (0, 1), (3, 1), (0, 2), (3, 2), (0, 3), (1, 3), (2, 3)]


def _materialise(tmp_path, case):
    raw = os.path.join(str(tmp_path), case["name"], "raw")
    os.makedirs(raw)
    for fn, text in case["inputs"].items():
        with open(os.path.join(raw, fn), "w") as f:
            f.write(text)
    return raw


def test_oracle_writes_the_reference_dataset_files(golden_dir, tmp_path):
    """f-3: DUMMY_/LINE_/CONJ_ datasets, byte for byte the files the reference's save_graph_data / save_graph_labels wrote
    (tests/golden/tu_files.json, produced by running tu_data_processing.py's own functions)."""
    from oracle import tu_format as TF
    with open(os.path.join(golden_dir, "tu_files.json")) as f:
        cases = json.load(f)
    assert len(cases) == 3
    for case in cases:
        raw = _materialise(tmp_path, case)
        TF.process_dataset(raw, case["name"])
        for rel, text in case["outputs"].items():
            with open(os.path.join(str(tmp_path), rel)) as f:
                assert f.read() == text, rel
        produced = sorted(os.path.relpath(os.path.join(dp, fn), str(tmp_path))
                          for dp, _, fns in os.walk(str(tmp_path)) for fn in fns if dp != raw)
        assert [p for p in produced if "_" + case["name"] + "/" in p] == sorted(case["outputs"])


def _without_multi_edges(case):
    name = case["name"]
    lines = case["inputs"][name + "_A.txt"].splitlines()
    seen, keep = set(), []
    for i, ln in enumerate(lines):
        k = ln.replace(" ", "")
        if k not in seen:
            seen.add(k)
            keep.append(i)
    inputs = dict(case["inputs"])
    for fn in (name + "_A.txt", name + "_edge_labels.txt", name + "_edge_attributes.txt"):
        if fn in inputs:
            rows = inputs[fn].splitlines()
            inputs[fn] = "".join(rows[i] + "\n" for i in keep)
    return {"name": name, "inputs": inputs}


def test_oracle_read_tu_data_on_the_written_datasets(golden_dir, tmp_path):
    """PyG 2.0.2 read_tu_data restated (oracle/tu_format.py header): structural properties on the reference-written
    DUMMY_ files + the dummy flags of PYGDataset.set_dummy_flags agree with the IS_DUMMY the writer knew."""
    from oracle import tu_format as TF
    with open(os.path.join(golden_dir, "tu_files.json")) as f:
        cases = json.load(f)
    for case in cases:
        case = _without_multi_edges(case)      # coalesce ADDS the one-hot rows of duplicates: PyG's label-width probe needs simple graphs
        raw = _materialise(tmp_path, case)
        TF.process_dataset(raw, case["name"])
        d = raw.replace(case["name"], "DUMMY_" + case["name"])
        data, slices = TF.read_tu_data(d, "DUMMY_" + case["name"])
        ei = data["edge_index"]
        G = len(slices["x"]) - 1
        assert slices["edge_index"][-1] == ei.shape[1] and len(data["y"]) >= G
        for g in range(G):
            e0, e1 = slices["edge_index"][g], slices["edge_index"][g + 1]
            n = slices["x"][g + 1] - slices["x"][g]
            sub = ei[:, e0:e1]
            assert sub.size == 0 or (sub.min() >= 0 and sub.max() < n)
            assert np.all(sub[0] != sub[1])                                   # self loops removed
            key = sub[0] * n + sub[1]
            assert np.all(np.diff(key) > 0)                                   # sorted and duplicate-free
        is_dn, is_de = TF.set_dummy_flags(data, add_dummy=True)
        b = TF.load_graph_data(TF.parse_tu_dir(raw), with_dummy=True)
        assert np.array_equal(is_dn, b["is_dummy_node"].astype(bool))
        # every kept edge that touches the dummy vertex is flagged, no other
        node_off = slices["x"][np.searchsorted(slices["edge_index"], np.arange(ei.shape[1]), side="right") - 1]
        touches = is_dn[ei[0] + node_off] | is_dn[ei[1] + node_off]
        assert np.array_equal(is_de, touches)


def test_oracle_si_bookkeeping_matches_reference(golden_dir):
    """f-2: conjugate sub-isomorphisms, match weights, norms, eigenvalue bounds and reversed edges against the outputs of the
    reference's own (numba-free) functions."""
    from oracle import si_bookkeeping as OB
    with open(os.path.join(golden_dir, "si_bookkeeping.json")) as f:
        gold = json.load(f)
    hits = 0
    for c in gold["cases"]:
        a = [c[k] for k in ("p_u", "p_v", "p_el", "g_u", "g_v", "g_el", "subisomorphisms")]
        conj = OB.conjugate_subisomorphisms(*a)
        assert conj.tolist() == c["conj_subisomorphisms"]
        hits += int((conj > 0).sum())
        assert OB.edgeseq_subisoweights(*a).tolist() == c["edgeseq_subisoweights"]
        assert OB.nodeseq_subisoweights(c["num_nodes"], c["subisomorphisms"]).tolist() == c["nodeseq_subisoweights"]
        for sl in (True, False):
            nn_, en_ = OB.compute_norm(c["g_u"], c["g_v"], c["num_nodes"], sl)
            assert np.array_equal(nn_, np.asarray(c["node_norm_%d" % sl], dtype=np.float32))
            assert np.array_equal(en_, np.asarray(c["edge_norm_%d" % sl], dtype=np.float32))
        assert OB.largest_eigenvalues(c["g_u"], c["g_v"], c["num_nodes"]) == (c["node_eigenv"], c["edge_eigenv"])
    assert hits > 20                                   # the fixtures embed the patterns: real matches are exercised
    voc = gold["reversed"]["vocab"]
    for before, after in zip(gold["reversed"]["before"], gold["reversed"]["after"]):
        for k, mne, mnel in (("pattern", voc["max_npe"], voc["max_npel"]), ("graph", voc["max_nge"], voc["max_ngel"])):
            r = OB.add_reversed_edges(before[k]["u"], before[k]["v"], before[k]["e_label"], mne, mnel)
            assert r["src"].tolist() == after[k]["u"] and r["dst"].tolist() == after[k]["v"]
            assert r["edge_id"].tolist() == after[k]["e_id"] and r["edge_label"].tolist() == after[k]["e_label"]
            assert r["is_reversed"].tolist() == after[k]["e_is_reversed"]


def test_oracle_dual_layers_match_reference(golden_dir):
    """f-4: CompGCNLayer / DMPLayer restatements against the reference's outputs and gradients (fp32, 1e-5 rel-max)."""
    import torch
    from oracle import layers as OL
    z = np.load(os.path.join(golden_dir, "si_dual_layers.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    worst = 0.0
    for m in meta:
        tag = m["tag"]
        p = {k[len(tag) + 7:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith(tag + "/param/")}
        u, v = torch.from_numpy(z[tag + "/u"]), torch.from_numpy(z[tag + "/v"])
        rev = torch.from_numpy(z[tag + "/rev"]) if m["rev"] else None
        x = torch.from_numpy(z[tag + "/x"]).clone().requires_grad_(True)
        ef = torch.from_numpy(z[tag + "/ef"]).clone().requires_grad_(True)
        if m["kind"] == "compgcn":
            no, eo = OL.compgcn_layer(x, ef, u, v, rev, p, comp_opt=m["comp_opt"], edge_norm=m["edge_norm"], act=m["act_func"])
        else:
            no, eo = OL.dmp_layer(x, ef, u, v, rev, p, num_mlp_layers=m["num_mlp_layers"], act=m["act_func"])
        ((no * torch.from_numpy(z[tag + "/c1"])).sum() + (eo * torch.from_numpy(z[tag + "/c2"])).sum()).backward()
        pairs = [(no, z[tag + "/node_out"]), (eo, z[tag + "/edge_out"]), (x.grad, z[tag + "/grad_x"]), (ef.grad, z[tag + "/grad_ef"])]
        for k, t in p.items():
            ref = z[tag + "/grad/" + k]
            if ref.size and np.abs(ref).max() > 0:
                pairs.append((t.grad, ref))
        for a, b in pairs:
            b = torch.from_numpy(np.asarray(b))
            err = float((a.detach() - b).abs().max() / b.abs().max().clamp(min=1e-12))
            assert err < 2e-5, (tag, err)
            worst = max(worst, err)
    assert len(meta) == 30


def gc_case(z, m, dtype=th.float32):
    """(params by reference state_dict name, data dict) of one gc_models.npz case."""
    tag = m["tag"]
    p = {k[len(tag) + 6:]: th.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/init/")}
    p = {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v) for k, v in p.items()}
    for k in list(p):                              # GIN: convs.i.nn IS nns.i (gconv.py:195-197) -- one tensor, two names
        if k.startswith("convs.") and ".nn." in k:
            i, rest = k.split(".")[1], k.split(".nn.", 1)[1]
            p[k] = p["nns.%s.%s" % (i, rest)]
    data = dict(x=th.from_numpy(z[tag + "/x"]).to(dtype), edge_index=th.from_numpy(z[tag + "/edge_index"]),
                edge_type=th.from_numpy(z[tag + "/edge_type"]), batch=th.from_numpy(z[tag + "/batch"]),
                y=th.from_numpy(z[tag + "/y"]), num_graphs=m["num_graphs"])
    return p, data


def test_oracle_gc_models_match_reference(golden_dir):
    """a-6 / a-7 / f-1: oracle/gc_models.py against the reference's gconv.py / rgconv.py run on the torch_geometric.nn
    stand-ins: log-probs, loss, every parameter gradient and the scalar dummy-edge-weight gradient."""
    z = np.load(os.path.join(golden_dir, "gc_models.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    assert len(meta) >= 17 and {m["kind"] for m in meta} == {"GIN", "RGIN", "RGCN", "GCN", "GCN_concat_readout", "GraphSAGE"}
    for m in meta:
        tag = m["tag"]
        p, data = gc_case(z, m)
        dw = th.tensor(m["dummy_weight"], requires_grad=True) if m["dummy_weight"] > 0 else None
        logp = OG.forward(m["kind"], p, data, m["additional"], dw)
        loss = th.nn.functional.nll_loss(logp, data["y"])
        loss.backward()
        th.testing.assert_close(logp.detach(), th.from_numpy(z[tag + "/logp"]), rtol=2e-5, atol=2e-5, msg=tag)
        assert abs(float(loss.detach()) - float(z[tag + "/loss"])) < 1e-5, tag
        for k, t_ in p.items():
            if not (t_.is_floating_point() and t_.requires_grad):
                continue
            if tag + "/grad/" + k not in z.files:          # alias name of a shared tensor
                continue
            ref = z[tag + "/grad/" + k]
            if ref.size == 0:
                assert t_.grad is None or float(t_.grad.abs().max()) == 0, (tag, k)
                continue
            scale = max(float(np.abs(ref).max()), 1e-3)
            assert float((t_.grad - th.from_numpy(ref)).abs().max()) < 2e-4 * scale + 2e-6, (tag, k)
        if dw is not None:
            assert abs(float(dw.grad) - float(z[tag + "/grad_dummy_weight"])) < 1e-5, tag


def test_dgl_branch_of_convert_conjugate_graph_equals_igraph_branch(golden_dir):
    """a-5: the reference's convert_conjugate_graph has a DGL branch (SI utils/graph.py:77-175) and an igraph branch
    (:177-267).  Both were run on the same items (si_conj_dgl.json / si_transforms.json): they agree on every vertex id,
    vertex label, edge, edge id and edge label, so the one restatement (and the device build pinned to it) covers both."""
    dgl, ig = _load(golden_dir, "si_conj_dgl.json"), _load(golden_dir, "si_transforms.json")
    n = 0
    for tag in ("conj", "conj_plain"):
        assert len(dgl[tag]) == len(ig[tag]) == 8
        for a, b in zip(dgl[tag], ig[tag]):
            for k in ("pattern", "graph"):
                for f in ("vcount", "edges", "v_id", "v_label", "e_id", "e_label"):
                    assert a[k].get(f, []) == b[k].get(f, []), (tag, k, f)
                n += 1
    for f in ("vcount", "edges", "v_id", "v_label", "e_id", "e_label"):
        assert dgl["kat2"].get(f, []) == ig["kat2"]["out"].get(f, []), f
    assert n == 32


def rep_net_oracle(z, m, p, x, dtype=th.float32):
    """The reference's stack semantics (rgin.py:214-260 / rgcn.py:254-300) on the oracle layers, parameters by state_dict
    name (rgin.<name>_rgin_(i).*), for a si_rep_nets.npz case."""
    tag = m["tag"]
    u, v, t = (th.from_numpy(z[tag + "/" + k]) for k in ("u", "v", "t"))
    mask, gate = th.from_numpy(z[tag + "/mask"]), th.from_numpy(z[tag + "/gate"]).to(dtype)
    mode = m["mode"]

    def layer(i, h):
        pre = "%s.%s_%s_(%d)." % (m["kind"], m["name"], m["kind"], i)
        lp = {k[len(pre):]: w for k, w in p.items() if k.startswith(pre)}
        if m["kind"] == "rgin":
            return OL.rgin_layer(h, u, v, t, lp, regularizer="basis", num_rels=m["R"], num_bases=-1, act=m["act_func"])
        return OL.rgcn_layer(h, u, v, t, lp, regularizer="basis", num_rels=m["R"], num_bases=-1, edge_norm=m["edge_norm"],
                             act=m["act_func"])

    if mode == "pattern_mask":
        h = x.masked_fill(~mask, 0.0)
        for i in range(m["num_layers"]):
            h = layer(i, h).masked_fill(~mask, 0.0)
        return h
    g = None
    if mode == "graph_gate":
        g = gate
    elif mode == "graph_mask":
        g = mask.to(dtype)
    elif mode == "graph_mask_gate":
        g = mask.to(dtype) * gate
    h = x if g is None else x * g
    for i in range(m["num_layers"]):
        o = layer(i, h)
        if g is not None:
            o = o * g
        h = h + o if m["rep_residual"] else o
    return h


def test_oracle_rep_nets_match_reference(golden_dir):
    """a-11: residual / pattern zero-mask / graph mask and gate paths of both stacks against the reference's own methods."""
    z = np.load(os.path.join(golden_dir, "si_rep_nets.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    assert len(meta) == 14
    for m in meta:
        tag = m["tag"]
        p = {k[len(tag) + 7:]: th.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith(tag + "/param/")}
        x = th.from_numpy(z[tag + "/x"]).clone().requires_grad_(True)
        out = rep_net_oracle(z, m, p, x)
        (out * th.from_numpy(z[tag + "/coef"])).sum().backward()
        th.testing.assert_close(out.detach(), th.from_numpy(z[tag + "/out"]), rtol=1e-4, atol=1e-5, msg=tag)
        th.testing.assert_close(x.grad, th.from_numpy(z[tag + "/grad_x"]), rtol=1e-4, atol=1e-5, msg=tag)
        for k, w in p.items():
            ref = z[tag + "/grad/" + k]
            if ref.size:
                th.testing.assert_close(w.grad, th.from_numpy(ref), rtol=2e-4, atol=2e-5, msg=tag + " " + k)
