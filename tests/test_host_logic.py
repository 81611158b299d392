"""Host-side logic on the CPU: module surface (names, shapes, initial weights), containers, partitioner,
and the rule that the product path refuses to run without the GPU (no CPU fallback)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import layers as OL


def _cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "si_layers.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def _build(m):
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGCNLayer, RGINLayer
    kw = dict(num_rels=m["num_rels"], regularizer=m["regularizer"], num_bases=m["num_bases"],
              self_loop=m["self_loop"], act_func=m["act_func"])
    if m["kind"] == "rgin":
        return RGINLayer(m["input_dim"], m["hidden_dim"], num_mlp_layers=m["num_mlp_layers"], **kw)
    return RGCNLayer(m["input_dim"], m["hidden_dim"], edge_norm=m["edge_norm"], **kw)


def test_initial_weights_are_bit_identical_to_the_reference(golden_dir):
    """Same torch.manual_seed => same RNG stream => same initial parameters as the reference modules
    (custom Xavier of SI utils/init.py:52-75, creation order of rgin.py:42-88 / rgcn.py:44-86)."""
    z, meta = _cases(golden_dir)
    for m in meta:
        torch.manual_seed(m["seed"])
        layer = _build(m)
        sd = layer.state_dict()
        ref_keys = sorted(k[len(m["tag"]) + 7:] for k in z.files if k.startswith(m["tag"] + "/param/"))
        assert sorted(sd.keys()) == ref_keys, m["tag"]
        for k in ref_keys:
            ref = z["%s/param/%s" % (m["tag"], k)]
            assert tuple(sd[k].shape) == ref.shape, (m["tag"], k)
            assert np.array_equal(sd[k].numpy(), ref), (m["tag"], k)


def test_gc_initial_state_dicts_are_bit_identical_to_the_reference(golden_dir):
    """Same torch.manual_seed => the build's GC models start from the reference's parameters bit for bit: creation order of
    gconv.py / rgconv.py plus what the torch-geometric 2.0.2 constructors do to the RNG stream (every conv re-initialises
    itself, GINConv re-initialises its nn, glorot / kaiming bounds), and the same state_dict names, shapes and buffers."""
    from dummynode4graphlearning_amd import graph_classification as GC
    z = np.load(os.path.join(golden_dir, "gc_models.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    for m in meta:
        args = SimpleNamespace(num_features=m["num_features"], hidden_dim=m["hidden_dim"], num_classes=m["num_classes"],
                               dropout_ratio=0.0, num_relations=m["num_relations"], additional=m["additional"],
                               epochs=m["epochs"], device="cpu", dummy_weight=m["dummy_weight"])
        torch.manual_seed(m["seed"])
        sd = getattr(GC, m["kind"])(args).state_dict()
        ref_keys = sorted(k[len(m["tag"]) + 6:] for k in z.files if k.startswith(m["tag"] + "/init/"))
        assert sorted(sd.keys()) == ref_keys, m["tag"]
        for k in ref_keys:
            ref = z["%s/init/%s" % (m["tag"], k)]
            assert tuple(sd[k].shape) == ref.shape, (m["tag"], k)
            assert np.array_equal(sd[k].numpy(), ref), (m["tag"], m["kind"], k)


def test_rep_net_state_dicts_are_bit_identical_to_the_reference(golden_dir):
    """RGINRepNet / RGCNRepNet built under the reference's seed: same module names (rgin.graph_rgin_(0) ...), same weights."""
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGCNRepNet, RGINRepNet
    z = np.load(os.path.join(golden_dir, "si_rep_nets.npz"))
    for m in json.loads(bytes(z["meta"]).decode()):
        torch.manual_seed(m["seed"])
        # the reference builds the graph net (3 layers) first, then the pattern net (2 layers): same RNG order here
        nets = {}
        for name, nl in (("graph", 3), ("pattern", 2)):
            if m["kind"] == "rgin":
                nets[name] = RGINRepNet(m["H"], m["R"], num_layers=nl, act_func=m["act_func"], name=name)
            else:
                nets[name] = RGCNRepNet(m["H"], m["R"], num_layers=nl, act_func=m["act_func"], edge_norm=m["edge_norm"], name=name)
        sd = nets[m["name"]].state_dict()
        ref_keys = sorted(k[len(m["tag"]) + 7:] for k in z.files if k.startswith(m["tag"] + "/param/"))
        assert sorted(sd.keys()) == ref_keys, (m["tag"], sorted(sd.keys())[:3], ref_keys[:3])
        for k in ref_keys:
            assert np.array_equal(sd[k].numpy(), z["%s/param/%s" % (m["tag"], k)]), (m["tag"], k)


def test_gc_model_state_dict_names():
    from dummynode4graphlearning_amd.graph_classification import GIN, RGCN, RGIN
    args = SimpleNamespace(num_features=7, hidden_dim=16, num_classes=3, dropout_ratio=0.5, num_relations=4,
                           additional={"num_layers": 3}, epochs=5, device="cpu", dummy_weight=0)
    gin = GIN(args)
    keys = set(gin.state_dict().keys())
    for k in ("first_h.0.weight", "first_h.1.running_mean", "first_h.3.bias", "first_h.4.weight", "nns.0.0.weight",
              "convs.0.nn.0.weight", "convs.0.eps", "convs.1.nn.4.bias", "linears.2.weight"):
        assert k in keys, k
    assert gin.convs[0].nn is gin.nns[0]                       # shared module, as in gconv.py:195-197
    assert gin.convs[0].train_eps is True                      # gconv.py:179 quirk: falls back to args.epochs (truthy)
    rgin = RGIN(args)
    assert tuple(rgin.state_dict()["convs.0.weight"].shape) == (4, 16, 16)
    assert tuple(rgin.state_dict()["convs.1.root"].shape) == (16, 16)
    rgcn = RGCN(args)
    assert {"conv1.weight", "conv2.root", "conv1.bias", "lin3.weight"} <= set(rgcn.state_dict().keys())
    assert tuple(rgcn.state_dict()["conv1.weight"].shape) == (4, 7, 16)


def test_layer_validation_matches_reference():
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    with pytest.raises(ValueError):
        RGINLayer(64, 64, num_rels=3, regularizer="bdd", num_bases=4)      # 64 % 3 != 0
    with pytest.raises(AssertionError):
        RGINLayer(8, 8, regularizer="diag")
    layer = RGINLayer(8, 8, num_rels=3, regularizer="basis", num_bases=7)
    assert layer.num_bases == 3 and layer.w_comp is None
    repr(layer)                                                              # the reference's extra_repr raises


def test_product_path_refuses_cpu_tensors():
    from dummynode4graphlearning_amd import BatchedGraph, ops
    from dummynode4graphlearning_amd._lib import DnHipError
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    x = torch.randn(4, 8)
    with pytest.raises(DnHipError):
        ops.gather_segsum(x, torch.zeros(2, dtype=torch.int32), torch.tensor([0, 1, 2], dtype=torch.int32))
    g = BatchedGraph(torch.tensor([0, 1]), torch.tensor([1, 2]), 4)
    with pytest.raises(DnHipError):
        RGINLayer(8, 8, num_rels=2)(g, x, torch.tensor([0, 1]))


def test_batched_graph_container():
    from dummynode4graphlearning_amd import BatchedGraph
    g1 = BatchedGraph(torch.tensor([0, 1]), torch.tensor([1, 2]), 3, ndata={"id": torch.arange(3)},
                      edata={"label": torch.tensor([0, 1])})
    g2 = BatchedGraph(torch.tensor([1]), torch.tensor([0]), 2, ndata={"id": torch.arange(2)},
                      edata={"label": torch.tensor([2])})
    b = BatchedGraph.batch([g1, g2])
    assert b.batch_size == 2 and b.number_of_nodes() == 5 and b.number_of_edges() == 3
    assert b.batch_num_nodes().tolist() == [3, 2] and b.batch_num_edges().tolist() == [2, 1]
    u, v = b.all_edges()
    assert u.tolist() == [0, 1, 4] and v.tolist() == [1, 2, 3]
    assert b.ndata["id"].tolist() == [0, 1, 2, 0, 1] and b.edata["label"].tolist() == [0, 1, 2]
    assert b.in_degrees().tolist() == [0, 1, 1, 1, 0] and b.out_degrees().tolist() == [1, 1, 0, 0, 1]
    assert b.node_ptr().tolist() == [0, 3, 5] and b.edge_ptr().tolist() == [0, 2, 3]


def test_graph_batch_collate():
    from dummynode4graphlearning_amd import GraphBatch
    items = [SimpleNamespace(x=torch.ones(2, 3), edge_index=torch.tensor([[0, 1], [1, 0]]), edge_attr=torch.eye(2),
                             y=torch.tensor([1]), is_dummy_node=torch.tensor([False, True]),
                             is_dummy_edge=torch.tensor([True, True])),
             SimpleNamespace(x=torch.zeros(3, 3), edge_index=torch.tensor([[2], [0]]), edge_attr=torch.tensor([[0., 1.]]),
                             y=torch.tensor([0]), is_dummy_node=torch.tensor([False, False, True]),
                             is_dummy_edge=torch.tensor([False]))]
    b = GraphBatch.collate(items)
    assert b.num_graphs == 2 and b.x.shape == (5, 3)
    assert b.edge_index.tolist() == [[0, 1, 4], [1, 0, 2]]
    assert b.batch.tolist() == [0, 0, 1, 1, 1] and b.ptr.tolist() == [0, 2, 5] and b.y.tolist() == [1, 0]


def test_cpu_baseline_port_equals_reference_formulation(golden_dir):
    """bench.py's CPU baseline (relation-grouped messages) is the same math as the reference's per-edge bmm."""
    z, meta = _cases(golden_dir)
    done = 0
    for m in meta:
        if m["kind"] != "rgin" or m["regularizer"] != "basis" or m["num_bases"] != -1 or not m["self_loop"]:
            continue
        tag = m["tag"]
        p = {k[len(tag) + 7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/param/")}
        u, v, t = (torch.from_numpy(z[tag + "/" + k]) for k in ("u", "v", "t"))
        out = OL.rgin_layer_rel_grouped(torch.from_numpy(z[tag + "/x"]), u, v, t, p, m["num_rels"], act=m["act_func"],
                                        num_mlp_layers=m["num_mlp_layers"])
        torch.testing.assert_close(out, torch.from_numpy(z[tag + "/out"]), rtol=1e-4, atol=1e-5, msg=tag)
        done += 1
    assert done >= 3


def test_synthetic_configs_have_the_survey_shapes():
    from dummynode4graphlearning_amd import synthetic
    from oracle import transforms as OT
    c3 = synthetic.config3()
    aug = OT.dummy_augment_si(*(c3[k] for k in ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id",
                                                 "edge_label")), c3["max_nv"], c3["max_nvl"], c3["max_ne"], c3["max_nel"])
    assert len(aug["node_label"]) == 25600 and len(aug["src"]) == 102400 and aug["edge_label"].max() == 7
    c5 = synthetic.config5(graphs=64)
    assert len(c5["src"]) == 64 * 62 and c5["num_rels"] == 16
    c1 = synthetic.config1()
    assert len(c1["node_ptr"]) == 33 and 400 < c1["node_ptr"][-1] < 800


def test_tu_reader_rejects_multi_column_attribute_files(tmp_path):
    """the reference parses one value per line (float(line.strip()), tu_data_processing.py:149-152) and raises on an
    ENZYMES-style multi-column attribute file; the reader must not flatten such a file into a longer 1-D array."""
    from dummynode4graphlearning_amd import tu_io
    d = tmp_path / "raw"
    d.mkdir()
    (d / "X_A.txt").write_text("1, 2\n2, 1\n")
    (d / "X_graph_indicator.txt").write_text("1\n1\n")
    (d / "X_node_labels.txt").write_text("0\n1\n")
    (d / "X_node_attributes.txt").write_text("0.5\n1.5\n")
    raw = tu_io.read_raw(str(d))
    assert raw["node_attributes"].tolist() == [0.5, 1.5] and raw["A"].tolist() == [[1, 2], [2, 1]]
    (d / "X_node_attributes.txt").write_text("0.5, 2.0\n1.5, 3.0\n")
    with pytest.raises(ValueError, match="one value per line"):
        tu_io.read_raw(str(d))




def test_weight_gradient_chunks_fit_one_round_of_workgroups():
    """ops.wgrad_chunk_rows / ops.close_chunks (host arithmetic behind the chunk tables of dn_rows_wgrad_* and the chunked closing
    launch): the smallest multiple of 64 rows (>= 256) for which every relation's chunks -- each relation ends in a partial one --
    fit ONE round of 256 workgroups, up to the cap; config 5's relation sizes give 245 chunks where 4,096-row chunks gave 775."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(0)
    for trial in range(50):
        R = int(rng.integers(1, 20))
        sizes = [int(rng.integers(0, 400000)) for _ in range(R)]
        ptr = [0] + list(np.cumsum(sizes))
        c = ops.wgrad_chunk_rows(ptr)
        n = sum(-(-s // c) for s in sizes if s > 0)
        assert c % 64 == 0 and 256 <= c <= ops.WGRAD_CHUNK_CAP
        assert n <= 256 or c == ops.WGRAD_CHUNK_CAP
        if c > 256 and n <= 256:                                       # minimal: 64 rows fewer would need a second round
            total = sum(sizes)
            lower = max(256, -(-total // 256 // 64) * 64)
            assert c == lower or sum(-(-s // (c - 64)) for s in sizes if s > 0) > 256
    sizes5 = [131072] * 16 + [1015808]                                  # config 5's shape: 16 relations of ~equal size + the self loop
    c5 = ops.wgrad_chunk_rows([0] + list(np.cumsum(sizes5)))
    assert sum(-(-s // c5) for s in sizes5) <= 256 < sum(-(-s // 4096) for s in sizes5)
    assert ops.wgrad_chunk_rows([0, 0, 0]) == 256
    for N, wg in ((1015808, 256), (645196, 256), (31, 256), (25600, 128)):
        C = ops.close_chunks(N, wg)                                     # chunks of the batch: a whole number per workgroup, < 16,384,
        assert C % wg == 0 and wg <= C < 16384                          # about CLOSE_CHUNK_TILES tiles each once the batch is that large
        assert C == wg or abs(N / 32.0 / C - ops.CLOSE_CHUNK_TILES) <= ops.CLOSE_CHUNK_TILES / 2


def test_wide_layer_chunks_cover_the_virtual_rows_in_one_round():
    """ops.wide_layer_chunks (the chunk table of the H = 256 layer's ONE weight-gradient launch): the conv's relations and the two
    Linears' rows end to end, every virtual row in exactly one chunk of its relation, chunk_ptr the running count, one round of 256
    workgroups up to the cap, and the Linears' chunks 1 / dense_weight the size of the conv's."""
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(1)
    dev = torch.device("cpu")
    for trial in range(30):
        R = int(rng.integers(1, 18))
        sizes = [int(rng.integers(0, 60000)) for _ in range(R)]
        N = int(rng.integers(1, 120000))
        ptr = [0] + [int(v) for v in np.cumsum(sizes + [N])]            # the self loop: one row per node, the last relation
        ch, cp, n = ops.wide_layer_chunks(ptr, N, dev)
        ch, cp = ch.numpy(), cp.numpy()
        vptr = ptr + [ptr[-1] + N, ptr[-1] + 2 * N]
        assert cp.shape[0] == len(vptr) and cp[0] == 0 and cp[-1] == n == ch.shape[0]
        assert n <= 256 or int((ch[:, 2] - ch[:, 1]).max()) == ops.WGRAD_CHUNK_CAP
        for r in range(len(vptr) - 1):
            mine = ch[cp[r]:cp[r + 1]]
            assert (mine[:, 0] == r).all()
            if vptr[r + 1] == vptr[r]:
                assert mine.shape[0] == 0
                continue
            assert mine[0, 1] == vptr[r] and mine[-1, 2] == vptr[r + 1]
            assert (mine[1:, 1] == mine[:-1, 2]).all() and (mine[:, 2] > mine[:, 1]).all()
        conv_step = int((ch[:cp[R + 1], 2] - ch[:cp[R + 1], 1]).max())
        dense_step = int((ch[cp[R + 1]:, 2] - ch[cp[R + 1]:, 1]).max())
        assert dense_step % 32 == 0 and dense_step <= max(64, conv_step // 2)
