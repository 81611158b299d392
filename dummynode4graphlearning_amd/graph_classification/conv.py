"""Conv / readout building blocks the GC models take from torch-geometric 2.0.2, re-implemented on the
MI355X hot path (torch-geometric is a third-party dependency that is not part of the reference tree:
README.md:26; layer definitions restated from its published documentation -- see oracle/layers.py).

  GINConv   x_i' = nn((1 + eps) x_i + sum_{j->i} x_j)                    call site gconv.py:197,212
  RGCNConv  x_i' = sum_r aggr_{j in N_r(i)} x_j W_r + x_i root + bias      call sites rgconv.py:17-18,96
  global_add_pool / global_mean_pool / global_max_pool                    call sites gconv.py:53,95,148,210,213
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ..graph import edge_index_of, gcn_edge_index_of, graph_ptr_i32, rel_index_of, row_index_of


def _reset(module):
    """torch_geometric.nn.inits.reset: re-initialise every child that has reset_parameters (GINConv does this to
    its nn at construction, which consumes RNG a second time -- mimicked for initial-weight parity)."""
    children = list(module.children()) if hasattr(module, "children") else []
    for item in (children if children else [module]):
        if hasattr(item, "reset_parameters"):
            item.reset_parameters()


class HipLinear(nn.Linear):
    """torch.nn.Linear (same parameters, names and initialisation) whose forward / backward run on the HIP path for 2-D GPU
    inputs: matrix cores for the square hidden layers, the any-width kernels for F -> H and H -> classes."""

    def forward(self, x):
        # exact fp32 products: at these batch sizes (10-20 k rows) the dense layers are a few microseconds either way, and the
        # BatchNorm / Adam steps behind them amplify the 1e-5 noise of the bf16 split into visible trajectory differences
        return ops.linear_any(x, self.weight, self.bias, exact="fwd")


class GINConv(nn.Module):
    def __init__(self, nn_module, eps=0.0, train_eps=False):
        super().__init__()
        self.nn = nn_module
        self.initial_eps = float(eps)
        if train_eps:
            self.eps = nn.Parameter(torch.tensor([float(eps)]))
        else:
            self.register_buffer("eps", torch.tensor([float(eps)]))
        self.train_eps = bool(train_eps)
        self._eps_host = None            # (tensor id, version, value): host copy of the eps BUFFER, refreshed when it changes
        self.reset_parameters()

    def reset_parameters(self):
        _reset(self.nn)
        with torch.no_grad():
            self.eps.fill_(self.initial_eps)

    def _eps_value(self):
        """The registered buffer's value as a host float (the self coefficient is a kernel argument).  Read back only when
        the buffer object or its version changed (load_state_dict, in-place edits, .to()): no per-forward sync."""
        key = (id(self.eps), self.eps._version)
        if self._eps_host is None or self._eps_host[0] != key:
            self._eps_host = (key, float(self.eps))
        return self._eps_host[1]

    def forward(self, x, data):
        index = edge_index_of(data)
        if self.train_eps:
            # gradient w.r.t. eps flows through the torch add; neighbours through the HIP gather
            out = ops.neighbor_sum(x, index, 0.0) + (1.0 + self.eps.to(x.dtype)) * x
        else:
            out = ops.neighbor_sum(x, index, 1.0 + self._eps_value())
        return self.nn(out)


class RGCNConv(nn.Module):
    def __init__(self, in_channels, out_channels, num_relations, aggr="mean", root_weight=True, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.num_relations, self.aggr = in_channels, out_channels, num_relations, aggr
        self.weight = nn.Parameter(torch.empty(num_relations, in_channels, out_channels))
        if root_weight:
            self.root = nn.Parameter(torch.empty(in_channels, out_channels))
        else:
            self.register_parameter("root", None)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        for w in (self.weight, self.root):  # glorot: U(-a, a), a = sqrt(6 / (size(-2) + size(-1)))
            if w is not None:
                a = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
                nn.init.uniform_(w, -a, a)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x, data, edge_type):
        if self.aggr == "add" and self.root is not None and ops.fused_path_supported(x, self.weight):
            # bf16: relation transform + root weight + bias in the row-factorised MFMA pipeline (as SI RGINLayer)
            index = row_index_of(data, edge_type, self.num_relations, True, closing_hint=(x.shape[1], x.dtype))
            return ops.rel_transform_fused(x, self.weight, self.bias, index, W_loop=self.root)
        index = rel_index_of(data, edge_type, self.num_relations)
        scale = None
        if self.aggr == "mean":
            # mean over the (rel, dst) segment == constant per-edge weight 1 / |segment|
            cnt = (index.seg_ptr[1:] - index.seg_ptr[:-1]).to(torch.float32)
            seg_of_edge = torch.empty(index.num_edges, dtype=torch.long, device=x.device)
            seg_of_edge[index.operm.long()] = index.seg_by_src.long()
            scale = (1.0 / cnt).index_select(0, seg_of_edge)
        out = ops.rel_agg_transform(x, self.weight, index, edge_scale=scale)
        if self.root is not None:
            out = out + x @ self.root
        if self.bias is not None:
            out = out + self.bias
        return out


class GCNConv(nn.Module):
    """x_i' = sum_j norm_ij (x_j W) + b with the symmetric GCN normalisation over weighted edges + self loops
    (torch_geometric GCNConv 2.0.2 defaults; call sites gconv.py:36-37,51-52,78-79).  `edge_weight` may require grad (the
    trainable dummy-edge weight): the normalisation is [E]-sized scalar arithmetic in torch, the feature traffic and the
    per-edge gradient <x_j W, g_i> run on the HIP kernels."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = HipLinear(in_channels, out_channels, bias=False)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.lin.weight.size(-2) + self.lin.weight.size(-1)))     # glorot
        nn.init.uniform_(self.lin.weight, -a, a)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x, data, edge_weight=None):
        index, keep = gcn_edge_index_of(data)
        N = x.shape[0]
        if edge_weight is None:
            w = torch.ones(index.num_edges, dtype=torch.float32, device=x.device)
        else:
            ew = edge_weight.to(torch.float32)
            loop_w = torch.ones(N, dtype=torch.float32, device=x.device)
            if index.gcn_keep_idx is None:                      # no self loops in the batch (every dummy-augmented TU batch): nothing dropped
                w = torch.cat([ew, loop_w])
            else:                                               # (positions cached with the index: no boolean indexing per call)
                loop_w = loop_w.index_put((index.gcn_drop_src,), ew.index_select(0, index.gcn_drop_idx))
                w = torch.cat([ew.index_select(0, index.gcn_keep_idx), loop_w])
        # weighted in-degree through the CSR gather (ops.edge_sum: fixed summation order per node, no float atomics -- the
        # trainable dummy-edge weight flows through it, so torch's index_add would make the step non-reproducible)
        deg = ops.edge_sum(w.view(-1, 1), index).view(-1)
        dis = deg.pow(-0.5)
        dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis).view(-1, 1)
        # dis[src] * w * dis[dst] with the gathers' backward as segment sums over the index's own CSR / CSC (torch's indexing backward
        # sorts the 10^5 indices on every call -- 19 merge launches -- and the whole step stops being capturable)
        norm = (ops.gather_rows(dis, index.src, (index.out_ptr, index.out_perm)).view(-1) * w
                * ops.gather_rows(dis, index.dst, (index.in_ptr, index.in_perm)).view(-1))
        out = ops.neighbor_sum(self.lin(x), index, 0.0, edge_scale=norm)
        return out + self.bias if self.bias is not None else out


class SAGEConv(nn.Module):
    """x_i' = lin_l(aggr_{j->i} x_j) + lin_r(x_i), aggr = mean | max | add (settable after construction through `.aggr`,
    as gconv.py:131 does).  torch_geometric SAGEConv 2.0.2 defaults (root_weight, no normalisation)."""

    def __init__(self, in_channels, out_channels, aggr="mean"):
        super().__init__()
        self.in_channels, self.out_channels, self.aggr = in_channels, out_channels, aggr
        self.lin_l = HipLinear(in_channels, out_channels, bias=True)
        self.lin_r = HipLinear(in_channels, out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        # the PyG constructor re-initialises both Linears (same U(-1/sqrt(in), 1/sqrt(in)) law as nn.Linear): a second draw
        # from the RNG stream, kept for initial-weight parity under the same seed
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x, data):
        index = edge_index_of(data)
        if self.aggr == "max":
            agg = ops.neighbor_max(x, index)
        else:
            scale = None
            if self.aggr == "mean":
                deg = (index.in_ptr[1:] - index.in_ptr[:-1]).to(torch.float32).clamp(min=1.0)
                scale = (1.0 / deg).index_select(0, index.dst.long())
            agg = ops.neighbor_sum(x, index, 0.0, edge_scale=scale)
        return self.lin_l(agg) + self.lin_r(x)


def global_add_pool(x, data):
    return ops.segment_reduce(x, graph_ptr_i32(data), "sum")


def global_mean_pool(x, data):
    return ops.segment_reduce(x, graph_ptr_i32(data), "mean")


def global_max_pool(x, data):
    return ops.segment_reduce(x, graph_ptr_i32(data), "max")
