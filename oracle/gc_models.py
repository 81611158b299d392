"""ORACLE (test infrastructure, never imported by the product package): functional CPU restatement of the
reference's graph classifiers, one function per model, parameters addressed by the reference's own state_dict names.

reference: graph_classification/graph_neural_networks/models/gconv.py:20-215 (GCN, GCN_concat_readout, GraphSAGE, GIN),
           graph_classification/graph_neural_networks/models/rgconv.py:6-126 (RGCN, RGIN).
The conv / pool arithmetic is torch-geometric 2.0.2's (README.md:26, third party, not under /root/reference) as
restated in oracle/layers.py.

Parity: PINNED for the model wiring by tests/golden/gc_models.npz -- the reference's gconv.py / rgconv.py imported unmodified
and run in the authoring container with torch_geometric.nn replaced by stand-ins of the published layer definitions
(tests/golden/make_golden.py:make_gc_models, tests/test_oracle_golden.py::test_oracle_gc_models_match_reference).  The PyG
arithmetic itself has no PyG run behind it (PyG is absent from the image): for those formulae parity stays "unpinned".

Everything is train-mode (BatchNorm uses batch statistics, dropout_ratio is 0 in every fixture), dtype follows the inputs
(fp32 against the goldens, fp64 as the checker of full-size GPU runs).
"""
import torch as th
import torch.nn.functional as F

from . import layers as OL


def _bn(x, p, prefix):
    return F.batch_norm(x, None, None, p[prefix + ".weight"], p[prefix + ".bias"], training=True, eps=1e-5)


def _mlp(x, p, prefix):
    """Sequential(Linear, BatchNorm1d, ReLU, Linear, BatchNorm1d, ReLU): gconv.py:187-188,193-194."""
    x = th.relu(_bn(F.linear(x, p[prefix + ".0.weight"], p[prefix + ".0.bias"]), p, prefix + ".1"))
    return th.relu(_bn(F.linear(x, p[prefix + ".3.weight"], p[prefix + ".3.bias"]), p, prefix + ".4"))


def _lin(x, p, name):
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def _head3(x, p):
    """lin1-relu-(dropout)-lin2-relu-(dropout)-lin3-log_softmax: gconv.py:55-60, rgconv.py:44-49."""
    x = th.relu(_lin(x, p, "lin1"))
    x = th.relu(_lin(x, p, "lin2"))
    return th.log_softmax(_lin(x, p, "lin3"), dim=-1)


def _num_layers(p):
    n = 0
    while "linears.%d.weight" % n in p:
        n += 1
    return n


def gin(p, x, src, dst, batch, num_graphs, aggregation="sum"):
    """gconv.py:203-215.  eps comes from the state_dict entry convs.i.eps (Parameter or buffer)."""
    pool = "add" if aggregation == "sum" else "mean"
    out = 0
    for layer in range(_num_layers(p)):
        if layer == 0:
            x = _mlp(x, p, "first_h")
            out = out + OL.global_pool(_lin(x, p, "linears.0"), batch, num_graphs, pool)
        else:
            eps = p["convs.%d.eps" % (layer - 1)]
            agg = (1.0 + eps) * x + OL.segment_sum(x[src], dst, x.shape[0])
            x = _mlp(agg, p, "convs.%d.nn" % (layer - 1))
            out = out + _lin(OL.global_pool(x, batch, num_graphs, pool), p, "linears.%d" % layer)
    return th.log_softmax(out, dim=-1)


def rgin(p, x, src, dst, etype, batch, num_graphs, aggregation="sum"):
    """rgconv.py:106-126: RGCNConv(aggr='add') then the layer's nn."""
    pool = "add" if aggregation == "sum" else "mean"
    out = 0
    for layer in range(_num_layers(p)):
        if layer == 0:
            x = _mlp(x, p, "first_h")
            out = out + OL.global_pool(_lin(x, p, "linears.0"), batch, num_graphs, pool)
        else:
            c = "convs.%d." % (layer - 1)
            x = OL.rgcn_conv(x, src, dst, etype, p[c + "weight"], p[c + "root"], p[c + "bias"], aggr="add")
            x = _mlp(x, p, "nns.%d" % (layer - 1))
            out = out + _lin(OL.global_pool(x, batch, num_graphs, pool), p, "linears.%d" % layer)
    return th.log_softmax(out, dim=-1)


def rgcn(p, x, src, dst, etype, batch, num_graphs):
    """rgconv.py:33-49: two RGCNConv (default aggr = mean) + mean readout + 3-layer head."""
    for c in ("conv1.", "conv2."):
        x = th.relu(OL.rgcn_conv(x, src, dst, etype, p[c + "weight"], p[c + "root"], p[c + "bias"], aggr="mean"))
    return _head3(OL.global_pool(x, batch, num_graphs, "mean"), p)


def gcn(p, x, src, dst, batch, num_graphs, edge_weight=None, concat_readout=False):
    """gconv.py:43-60 (GCN) and :86-104 (GCN_concat_readout); edge_weight = ones with the dummy weight on dummy edges."""
    for c in ("conv1.", "conv2."):
        x = th.relu(OL.gcn_conv(x, src, dst, edge_weight, p[c + "lin.weight"], p[c + "bias"]))
    if concat_readout:
        x = th.cat([OL.global_pool(x, batch, num_graphs, "max"), OL.global_pool(x, batch, num_graphs, "mean")], dim=1)
    else:
        x = OL.global_pool(x, batch, num_graphs, "mean")
    return _head3(x, p)


def graphsage(p, x, src, dst, batch, num_graphs, aggregation="mean"):
    """gconv.py:138-152."""
    xs = []
    i = 0
    while "layers.%d.lin_l.weight" % i in p:
        c = "layers.%d." % i
        x = OL.sage_conv(x, src, dst, p[c + "lin_l.weight"], p[c + "lin_l.bias"], p[c + "lin_r.weight"], aggr=aggregation)
        if aggregation == "max":
            x = th.relu(_lin(x, p, "fc_max"))
        xs.append(x)
        i += 1
    x = OL.global_pool(th.cat(xs, dim=1), batch, num_graphs, "max")
    return th.log_softmax(_lin(th.relu(_lin(x, p, "fc1")), p, "fc2"), dim=-1)


def forward(kind, p, data, additional=None, dummy_weight=None):
    """Dispatch on the reference's model name.  `data`: dict with x, edge_index [2,E], edge_type [E], batch, num_graphs."""
    cfg = additional or {}
    x, (src, dst), batch, B = data["x"], data["edge_index"], data["batch"], data["num_graphs"]
    if kind == "GIN":
        return gin(p, x, src, dst, batch, B, cfg.get("aggregation", "sum"))
    if kind == "RGIN":
        return rgin(p, x, src, dst, data["edge_type"], batch, B, cfg.get("aggregation", "sum"))
    if kind == "RGCN":
        return rgcn(p, x, src, dst, data["edge_type"], batch, B)
    if kind in ("GCN", "GCN_concat_readout"):
        w = None
        if dummy_weight is not None:
            w = th.where(data["edge_type"] == 0, dummy_weight, th.ones(src.numel(), dtype=x.dtype))
        return gcn(p, x, src, dst, batch, B, w, concat_readout=(kind == "GCN_concat_readout"))
    if kind == "GraphSAGE":
        return graphsage(p, x, src, dst, batch, B, cfg.get("aggregation", "mean"))
    raise ValueError(kind)
