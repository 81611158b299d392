"""GIN / RGCN / RGIN graph classifiers with the reference's constructor (``Model(args)``), attribute and
state_dict names and ``forward(data) -> log-probs`` surface, running their message passing and readouts on
the MI355X HIP path.

reference: graph_classification/graph_neural_networks/models/gconv.py:154-215 (GIN) and
rgconv.py:6-126 (RGCN, RGIN; not exported there and reading ``args.nhid`` -- here ``nhid`` falls back to
``args.hidden_dim``).
"""
import torch
import torch.nn.functional as F
from torch.nn import BatchNorm1d, Identity, Linear, ReLU, Sequential

from .. import ops
from .conv import GCNConv, GINConv, HipLinear, RGCNConv, SAGEConv, global_add_pool, global_max_pool, global_mean_pool


def _pooling(config):
    kind = config.get("aggregation", "sum")
    if kind == "sum":
        return global_add_pool
    if kind == "mean":
        return global_mean_pool
    raise ValueError("aggregation must be 'sum' or 'mean'")


class HipBatchNorm1d(BatchNorm1d):
    """torch.nn.BatchNorm1d (same parameters, buffers, names) whose training-mode statistics, normalisation and backward run
    as the row-streaming kernels of dn_norm.hip; eval mode and unsupported shapes fall through to torch.  fuse_relu: the module
    also applies the ReLU that follows it in the reference's Sequential (the ReLU's slot then holds an Identity, so the
    state_dict keys and module indices stay those of `Sequential(Linear, BatchNorm1d, ReLU, ...)`)."""

    def __init__(self, *args, fuse_relu=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.fuse_relu = bool(fuse_relu)

    def forward(self, x):
        return hip_batch_norm_forward(self, x, self.fuse_relu)


def hip_batch_norm_forward(mod, x, relu):
    """forward of a BatchNorm1d-like module `mod` (torch's parameters / buffers) on the HIP kernels, optionally with the ReLU behind
    it; shared by HipBatchNorm1d and by parallel.SyncBatchNorm1d when no process group is active."""
    if not (mod.training and ops.batch_norm_rows_supported(x) and x.shape[0] > 1):          # (one row: torch raises, as it should)
        y = BatchNorm1d.forward(mod, x)
        return F.relu(y) if relu else y
    w, b = (mod.weight, mod.bias) if mod.affine else (None, None)
    if mod.track_running_stats:
        nbt = mod.num_batches_tracked
        if mod.momentum is not None and mod.running_mean.dtype == torch.float32 and nbt.is_cuda and nbt.dtype == torch.int64:
            # the running-statistics update and the batch counter ride in the statistics launch (six tiny launches otherwise)
            return ops.batch_norm_rows(x, w, b, mod.eps, mod.running_mean, mod.running_var, mod.momentum, relu=relu,
                                       batches_tracked=nbt)[0]
        with torch.no_grad():
            mod.num_batches_tracked += 1
        if mod.momentum is not None and mod.running_mean.dtype == torch.float32:
            return ops.batch_norm_rows(x, w, b, mod.eps, mod.running_mean, mod.running_var, mod.momentum, relu=relu)[0]
    y, mean, var = ops.batch_norm_rows(x, w, b, mod.eps, relu=relu)
    if mod.track_running_stats:
        with torch.no_grad():
            mom = mod.momentum if mod.momentum is not None else 1.0 / float(mod.num_batches_tracked)
            n = x.shape[0]
            mod.running_mean.mul_(1.0 - mom).add_(mean.to(mod.running_mean.dtype), alpha=mom)
            mod.running_var.mul_(1.0 - mom).add_((var * (n / max(n - 1, 1))).to(mod.running_var.dtype), alpha=mom)
    return y


def _mlp(in_dim, out_dim):
    """gconv.py:187-194 / rgconv.py:85-93: Sequential(Linear, BatchNorm1d, ReLU, Linear, BatchNorm1d, ReLU) -- same indices and
    parameter names; each ReLU runs inside the BatchNorm launches before it."""
    return Sequential(HipLinear(in_dim, out_dim), HipBatchNorm1d(out_dim, fuse_relu=True), Identity(),
                      HipLinear(out_dim, out_dim), HipBatchNorm1d(out_dim, fuse_relu=True), Identity())


def _edge_type(data, x):
    """rgconv.py:35-38,110-113.  The derived tensor is kept on the batch object (keyed on the edge_attr tensor and its
    version) so that every conv of every forward over the same batch presents the SAME edge_type tensor to the index cache
    of graph.py (a fresh tensor per forward would rebuild the relation index -- sorts and stream syncs -- each time)."""
    edge_attr = getattr(data, "edge_attr", None)
    hit = getattr(data, "_dn_edge_type", None)
    key = (None, 0, data.edge_index.size(1)) if edge_attr is None else (id(edge_attr), edge_attr._version, edge_attr.shape[0])
    if hit is not None and hit[0] == key and hit[1].device == x.device:
        return hit[1]
    if edge_attr is not None:
        et = edge_attr.max(dim=1)[1]
    else:
        et = torch.zeros(data.edge_index.size(1), dtype=torch.long, device=x.device)
    try:
        data._dn_edge_type = (key, et, edge_attr)          # edge_attr kept alive so its id cannot be recycled
    except Exception:
        pass
    return et


def _dummy_edge_weight(model, data, device):
    """gconv.py:46-49: ones, with the (gradient-carrying) dummy weight on the dummy edges."""
    if not model.use_edge_weight:
        return None
    flag = data.is_dummy_edge.to(device)
    edge_attr = torch.ones(flag.size(), device=device)
    return torch.where(flag, model.dummy_weight.to(device), edge_attr)


class _GCNBase(torch.nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.num_features, self.hidden_dim = args.num_features, args.hidden_dim
        self.num_classes, self.dropout_ratio = args.num_classes, args.dropout_ratio
        # gconv.py:29-34: a plain tensor with requires_grad (NOT an nn.Parameter: the reference's optimiser never sees it;
        # its gradient is still produced, through dn_edge_dot_*)
        if getattr(args, "dummy_weight", 0) > 0:
            self.dummy_weight = torch.tensor(float(args.dummy_weight), requires_grad=True, device=args.device)
            self.use_edge_weight = True
        else:
            self.use_edge_weight = False
        self.conv1 = GCNConv(self.num_features, self.hidden_dim)
        self.conv2 = GCNConv(self.hidden_dim, self.hidden_dim)

    def _convs(self, data):
        x = data.x
        w = _dummy_edge_weight(self, data, x.device)
        x = F.relu(self.conv1(x, data, w))
        return F.relu(self.conv2(x, data, w))

    def _head(self, x):
        x = F.relu(self.lin1(x))
        x = F.dropout(x, p=self.dropout_ratio, training=self.training)
        x = F.relu(self.lin2(x))
        x = F.dropout(x, p=self.dropout_ratio, training=self.training)
        return F.log_softmax(self.lin3(x), dim=-1)


class GCN(_GCNBase):
    """reference: gconv.py:20-60."""

    def __init__(self, args):
        super().__init__(args)
        self.lin1 = Linear(self.hidden_dim, self.hidden_dim)
        self.lin2 = Linear(self.hidden_dim, self.hidden_dim // 2)
        self.lin3 = Linear(self.hidden_dim // 2, self.num_classes)

    def forward(self, data):
        return self._head(global_mean_pool(self._convs(data), data))


class GCN_concat_readout(_GCNBase):
    """reference: gconv.py:62-104 (max and mean readouts concatenated)."""

    def __init__(self, args):
        super().__init__(args)
        self.lin1 = Linear(self.hidden_dim * 2, self.hidden_dim)
        self.lin2 = Linear(self.hidden_dim, self.hidden_dim // 2)
        self.lin3 = Linear(self.hidden_dim // 2, self.num_classes)

    def forward(self, data):
        x = self._convs(data)
        return self._head(torch.cat([global_max_pool(x, data), global_mean_pool(x, data)], dim=1))


class GraphSAGE(torch.nn.Module):
    """reference: gconv.py:106-152."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.num_features, self.hidden_dim = args.num_features, args.hidden_dim
        self.num_classes, self.dropout_ratio = args.num_classes, args.dropout_ratio
        config = args.additional if getattr(args, "additional", None) else {"num_layers": 2, "aggregation": "mean"}
        if config.get("aggregation", "mean") == "max":
            self.fc_max = Linear(self.hidden_dim, self.hidden_dim)
        num_layers = config.get("num_layers", 2)
        self.aggregation = config.get("aggregation", "mean")
        self.layers = torch.nn.ModuleList([])
        for i in range(num_layers):
            conv = SAGEConv(self.num_features if i == 0 else self.hidden_dim, self.hidden_dim)
            conv.aggr = self.aggregation
            self.layers.append(conv)
        self.fc1 = Linear(num_layers * self.hidden_dim, self.hidden_dim)
        self.fc2 = Linear(self.hidden_dim, self.num_classes)

    def forward(self, data):
        x = data.x
        x_all = []
        for layer in self.layers:
            x = layer(x, data)
            if self.aggregation == "max":
                x = torch.relu(self.fc_max(x))
            x_all.append(x)
        x = global_max_pool(torch.cat(x_all, dim=1), data)
        x = F.relu(self.fc1(x))
        return F.log_softmax(self.fc2(x), dim=-1)


class GIN(torch.nn.Module):
    def __init__(self, args):
        super(GIN, self).__init__()
        self.args = args
        self.num_features = args.num_features
        self.hidden_dim = args.hidden_dim
        self.num_classes = args.num_classes
        self.dropout = args.dropout_ratio

        if getattr(args, "additional", None):
            config = args.additional
        else:
            config = {"train_eps": False, "num_layers": 2, "aggregation": "sum"}
        self.pooling = _pooling(config)
        train_eps = config.get("train_eps", getattr(args, "epochs", False))      # gconv.py:179 (sic)

        self.embeddings_dim = [self.hidden_dim for _ in range(config.get("num_layers", 2))]
        self.no_layers = len(self.embeddings_dim)
        nns, convs, linears = [], [], []
        for layer, out_emb_dim in enumerate(self.embeddings_dim):
            if layer == 0:
                self.first_h = _mlp(self.num_features, out_emb_dim)
                linears.append(HipLinear(out_emb_dim, self.num_classes))
            else:
                nns.append(_mlp(self.embeddings_dim[layer - 1], out_emb_dim))
                convs.append(GINConv(nns[-1], train_eps=bool(train_eps)))
                linears.append(HipLinear(out_emb_dim, self.num_classes))
        self.nns = torch.nn.ModuleList(nns)
        self.convs = torch.nn.ModuleList(convs)
        self.linears = torch.nn.ModuleList(linears)

    def forward(self, data):
        x = data.x
        out = 0
        for layer in range(self.no_layers):
            if layer == 0:
                x = self.first_h(x)
                out = out + F.dropout(self.pooling(self.linears[layer](x), data), p=self.dropout)   # always on (sic)
            else:
                x = self.convs[layer - 1](x, data)
                out = out + F.dropout(self.linears[layer](self.pooling(x, data)), p=self.dropout, training=self.training)
        return F.log_softmax(out, dim=-1)


class RGCN(torch.nn.Module):
    def __init__(self, args):
        super(RGCN, self).__init__()
        self.args = args
        self.num_features = args.num_features
        self.nhid = getattr(args, "nhid", None) or args.hidden_dim
        self.num_classes = args.num_classes
        self.dropout_ratio = args.dropout_ratio
        self.num_relations = args.num_relations

        self.conv1 = RGCNConv(self.num_features, self.nhid, self.num_relations)
        self.conv2 = RGCNConv(self.nhid, self.nhid, self.num_relations)
        config = getattr(args, "additional", None)
        if config and "weight_reg" in config and config["weight_reg"] > 1.1:
            with torch.no_grad():
                self.conv1.weight.div_(config["weight_reg"])
                self.conv2.weight.div_(config["weight_reg"])
        self.lin1 = Linear(self.nhid, self.nhid)
        self.lin2 = Linear(self.nhid, self.nhid // 2)
        self.lin3 = Linear(self.nhid // 2, self.num_classes)

    def forward(self, data):
        x = data.x
        edge_type = _edge_type(data, x)
        x = F.relu(self.conv1(x, data, edge_type))
        x = F.relu(self.conv2(x, data, edge_type))
        x = global_mean_pool(x, data)
        x = F.relu(self.lin1(x))
        x = F.dropout(x, p=self.dropout_ratio, training=self.training)
        x = F.relu(self.lin2(x))
        x = F.dropout(x, p=self.dropout_ratio, training=self.training)
        return F.log_softmax(self.lin3(x), dim=-1)


class RGIN(torch.nn.Module):
    def __init__(self, args):
        super(RGIN, self).__init__()
        self.args = args
        self.num_features = args.num_features
        self.nhid = getattr(args, "nhid", None) or args.hidden_dim
        self.num_classes = args.num_classes
        self.dropout = args.dropout_ratio
        self.num_relations = args.num_relations

        config = args.additional if getattr(args, "additional", None) else {"num_layers": 2}
        self.pooling = _pooling(config)
        self.embeddings_dim = [self.nhid for _ in range(config.get("num_layers", 2))]
        self.no_layers = len(self.embeddings_dim)
        nns, convs, linears = [], [], []
        for layer, out_emb_dim in enumerate(self.embeddings_dim):
            if layer == 0:
                self.first_h = _mlp(self.num_features, out_emb_dim)
                linears.append(HipLinear(out_emb_dim, self.num_classes))
            else:
                nns.append(_mlp(self.embeddings_dim[layer - 1], out_emb_dim))
                convs.append(RGCNConv(self.nhid, self.nhid, self.num_relations, aggr="add"))      # rgconv.py:96
                linears.append(HipLinear(out_emb_dim, self.num_classes))
        if "weight_reg" in config and config["weight_reg"] > 1.1:
            with torch.no_grad():
                for conv in convs:
                    conv.weight.div_(config["weight_reg"])
        self.nns = torch.nn.ModuleList(nns)
        self.convs = torch.nn.ModuleList(convs)
        self.linears = torch.nn.ModuleList(linears)

    def forward(self, data):
        x = data.x
        edge_type = _edge_type(data, x)
        out = 0
        for layer in range(self.no_layers):
            if layer == 0:
                x = self.first_h(x)
                out = out + F.dropout(self.pooling(self.linears[layer](x), data), p=self.dropout)
            else:
                x = self.convs[layer - 1](x, data, edge_type)
                x = self.nns[layer - 1](x)
                out = out + F.dropout(self.linears[layer](self.pooling(x, data)), p=self.dropout, training=self.training)
        return F.log_softmax(out, dim=-1)
