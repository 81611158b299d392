"""RGCN layer / stack of the subgraph-isomorphism models on the MI355X hot path.

Mirror of subgraph_isomorphism/models/rgcn.py:16-300 (constructor, parameter names, forward surface, the
in/out-degree normalisation and its side effects on the graph object); message passing runs through
the row-factorised matrix-core pipeline (square H in {64,128,256}, fp32/bf16; the edge norm is separable into a
per-source and a per-destination factor) or ops.rel_agg_transform with the per-edge norm folded into the first gather.
GPU only.
"""
import torch as th
import torch.nn as nn

from .. import ops
from .act import map_activation_str_to_layer
from .init import init_weight
from .rgin import dense_relation_weights

NODEFEAT, EDGETYPE = "node_feat", "edge_type"
NORM, INDEGREE, INNORM, OUTDEGREE, OUTNORM = "norm", "in_deg", "in_norm", "out_deg", "out_norm"


class RGCNLayer(nn.Module):
    def __init__(
        self,
        input_dim,
        hidden_dim,
        num_rels=1,
        regularizer="basis",
        num_bases=-1,
        edge_norm="in",
        self_loop=True,
        bias=True,
        batch_norm=False,
        act_func="relu",
        dropout=0.0,
    ):
        super(RGCNLayer, self).__init__()
        assert regularizer in ["none", "basis", "bdd"]
        assert edge_norm in ["none", "in", "both"]

        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_rels = num_rels
        self.regularizer = regularizer
        if regularizer == "none" or num_bases is None or num_bases > num_rels or num_bases <= 0:
            self.num_bases = num_rels
        else:
            self.num_bases = num_bases
        self.edge_norm = edge_norm
        if self_loop:
            self.loop_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        else:
            self.register_parameter("loop_weight", None)
        if bias:
            self.bias = nn.Parameter(th.empty(hidden_dim))
        else:
            self.register_parameter("bias", None)
        self.bn = nn.BatchNorm1d(hidden_dim) if batch_norm else None
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)

        if regularizer == "none" or regularizer == "basis":
            self.weight = nn.Parameter(th.empty(self.num_bases, self.input_dim, self.hidden_dim))
            if self.num_bases < self.num_rels:
                self.w_comp = nn.Parameter(th.empty(self.num_rels, self.num_bases))
            else:
                self.register_parameter("w_comp", None)
        else:
            if input_dim % self.num_bases != 0 or hidden_dim % self.num_bases != 0:
                raise ValueError("Feature size must be a multiplier of num_bases (%d)." % self.num_bases)
            submat_in = input_dim // self.num_bases
            submat_out = hidden_dim // self.num_bases
            self.weight = nn.Parameter(th.empty(self.num_rels, self.num_bases * submat_in * submat_out))
            self.register_parameter("w_comp", None)

        init_weight(self.weight, activation=act_func, init="uniform")
        if self.w_comp is not None:
            init_weight(self.w_comp, activation=act_func, init="uniform")
        if self_loop:
            init_weight(self.loop_weight, activation=act_func, init="uniform")
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    @property
    def self_loop(self):
        return hasattr(self, "loop_weight") and self.loop_weight is not None

    def _norms(self, g):
        """rgcn.py:132-165: fills g.ndata[in_deg/in_norm/out_deg/out_norm] and g.edata[norm] (cached on the graph)."""
        if self.edge_norm == "none":
            return None
        if NORM in g.edata and INNORM in g.ndata and (self.edge_norm == "in" or OUTNORM in g.ndata):
            return g.edata[NORM]
        src, dst = g.all_edges()
        src, dst = src.to(th.int32).contiguous(), dst.to(th.int32).contiguous()
        if INDEGREE not in g.ndata or (self.edge_norm == "both" and OUTDEGREE not in g.ndata):
            ind, outd = ops.degrees(src, dst, g.number_of_nodes())
            g.ndata.setdefault(INDEGREE, ind.long())
            if self.edge_norm == "both":
                g.ndata.setdefault(OUTDEGREE, outd.long())
        in_deg = g.ndata[INDEGREE].to(th.int32).contiguous()
        out_deg = g.ndata[OUTDEGREE].to(th.int32).contiguous() if self.edge_norm == "both" else in_deg
        in_norm, out_norm, en = ops.edge_norm(self.edge_norm, self.self_loop, src, dst, in_deg, out_deg)
        g.ndata[INNORM] = in_norm
        if out_norm is not None:
            g.ndata[OUTNORM] = out_norm
        g.edata[NORM] = en.view(-1, 1)
        return g.edata[NORM]

    def forward(self, g, node_feat, edge_type):
        g.ndata[NODEFEAT] = node_feat
        g.edata[EDGETYPE] = edge_type
        norm = self._norms(g)
        W = dense_relation_weights(self)
        if ops.fused_path_supported(node_feat, W):
            # the reference's edge norm is separable (rgcn.py:148-165): 'in' = in_norm[dst]; 'both' =
            # sqrt(out_norm[src]) * sqrt(in_norm[dst]), and the self-loop message carries the same two factors at u = v
            # (rgcn.py:174-179) -- so the whole layer is  s_in * rowpipeline(s_out * x, [W; W_loop])  on the matrix cores.
            s_in = s_out = None
            if self.edge_norm == "in":
                s_in = g.ndata[INNORM]
            elif self.edge_norm == "both":
                s_in, s_out = g.ndata[INNORM].sqrt(), g.ndata[OUTNORM].sqrt()
            xin = node_feat if s_out is None else node_feat * s_out.to(node_feat.dtype)
            index = g.row_index(edge_type, self.num_rels, self.self_loop, closing_hint=(xin.shape[1], xin.dtype))
            out = ops.rel_transform_fused(xin, W, None, index, W_loop=self.loop_weight if self.self_loop else None)
            if s_in is not None:
                out = out * s_in.to(out.dtype)
        else:
            index = g.rel_index(edge_type, self.num_rels)
            out = ops.rel_agg_transform(node_feat, W, index, edge_scale=None if norm is None else norm.view(-1))
            if self.self_loop:
                loop_msg = th.matmul(node_feat, self.loop_weight)
                if self.edge_norm == "in":
                    out = out + loop_msg * g.ndata[INNORM].to(loop_msg.dtype)                       # rgcn.py:174-175
                elif self.edge_norm == "both":
                    out = out + loop_msg * ((g.ndata[INNORM] * g.ndata[OUTNORM]) ** 0.5).to(loop_msg.dtype)
                else:
                    out = out + loop_msg
        if self.bias is not None:
            out = out + self.bias
        if self.bn is not None:
            if self.bn.training and ops.batch_norm_rows_supported(out) and out.shape[0] > 1:
                # --rep_rgcn_batch_norm (rgcn.py:52-53, 185-187): statistics, buffers, normalisation and a ReLU behind it on the HIP kernels
                from ..graph_classification.models import hip_batch_norm_forward
                fuse = isinstance(self.act, nn.ReLU)
                out = hip_batch_norm_forward(self.bn, out, fuse)
                if fuse:
                    return self.drop(out), edge_type
            else:
                out = self.bn(out)
        out = self.act(out)
        out = self.drop(out)
        return out, edge_type

    def get_output_dim(self):
        return self.hidden_dim

    def extra_repr(self):
        return "\n".join([
            "in=%d, out=%d," % (self.input_dim, self.hidden_dim),
            "num_rels=%d, regularizer=%s, num_bases=%d," % (self.num_rels, self.regularizer, self.num_bases),
            "edge_norm=%s, self_loop=%s, bias=%s," % (self.edge_norm, self.self_loop, self.bias is not None),
        ])


class RGCNRepNet(nn.Module):
    """ModuleList of RGCNLayer with the residual / mask / gate handling of rgcn.py:219-300."""

    def __init__(self, hid_dim, num_rels, num_layers=1, rep_residual=True, regularizer="basis", num_bases=-1,
                 edge_norm="in", batch_norm=False, act_func="relu", dropout=0.0, name="graph"):
        super().__init__()
        self.rep_residual = rep_residual
        layers = nn.ModuleList()
        for i in range(num_layers):
            layers.add_module("%s_rgcn_(%d)" % (name, i), RGCNLayer(
                hid_dim, hid_dim, num_rels=num_rels, regularizer=regularizer, num_bases=num_bases, edge_norm=edge_norm,
                batch_norm=batch_norm, act_func=act_func, dropout=dropout))
        self.rgcn = layers

    def get_pattern_rep(self, pattern, p_emb, mask=None):
        """rgcn.py:254-272: with a mask the pattern side zero-fills masked rows before and after every layer (no residual)."""
        if mask is not None:
            p_zero_mask = ~mask
            outputs = [p_emb.masked_fill(p_zero_mask, 0.0)]
            etype = pattern.edata["label"]
            for layer in self.rgcn:
                o, etype = layer(pattern, outputs[-1], etype)
                outputs.append(o.masked_fill(p_zero_mask, 0.0))
            return outputs[-1]
        return self.get_graph_rep(pattern, p_emb)

    def get_graph_rep(self, graph, g_emb, mask=None, gate=None):
        etype = graph.edata["label"]
        if mask is not None or gate is not None:
            if gate is None:
                gate = mask.to(g_emb.dtype)
            elif mask is not None:
                gate = mask.to(g_emb.dtype) * gate
        outputs = [g_emb if gate is None else g_emb * gate]
        for layer in self.rgcn:
            o, etype = layer(graph, outputs[-1], etype)
            if gate is not None:
                o = o * gate
            outputs.append(outputs[-1] + o if self.rep_residual and outputs[-1].size() == o.size() else o)
        return outputs[-1]

    forward = get_graph_rep
