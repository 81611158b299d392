"""RGIN layer / stack of the subgraph-isomorphism models on the MI355X hot path.

Same constructor arguments, parameter names/shapes (state_dict interchange) and forward() surface as
subgraph_isomorphism/models/rgin.py:16-260; the DGL ``update_all`` with its per-edge [E,H,H] weight gather
is replaced by the aggregate-then-transform HIP path (ops.rel_agg_transform).  GPU only.
"""
import torch as th
import torch.nn as nn

from .. import ops
from .act import map_activation_str_to_layer
from .init import init_weight

NODEFEAT, EDGETYPE, NODEOUTPUT = "node_feat", "edge_type", "node_out"  # constants.py:26-35


def _fused_slope(act):
    """Negative-side slope of an activation the fused kernels implement (0 for ReLU), else None."""
    if isinstance(act, nn.ReLU):
        return 0.0
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    return None


def dense_relation_weights(layer):
    """[R, in, out] weights from the basis / block-diagonal parameterisation (rgin.py:103-108, 114-117)."""
    if layer.regularizer in ("none", "basis"):
        if layer.num_bases < layer.num_rels:
            w = layer.weight.view(layer.num_bases, layer.input_dim * layer.hidden_dim)
            return th.matmul(layer.w_comp, w).view(layer.num_rels, layer.input_dim, layer.hidden_dim)
        return layer.weight
    si, so = layer.input_dim // layer.num_bases, layer.hidden_dim // layer.num_bases
    if layer.weight.is_cuda and layer.weight.dtype in (th.bfloat16, th.float32):
        return ops.bdd_dense(layer.weight, layer.num_rels, layer.num_bases, si, so)      # one launch (dn_bdd_compose)
    blocks = layer.weight.view(layer.num_rels, layer.num_bases, si, so)
    # block_diag per relation, differentiable: [R, B, si, B, so] with zeros off the diagonal
    eye = th.eye(layer.num_bases, dtype=blocks.dtype, device=blocks.device)
    dense = blocks.unsqueeze(3) * eye.view(1, layer.num_bases, 1, layer.num_bases, 1)
    return dense.reshape(layer.num_rels, layer.input_dim, layer.hidden_dim)


class RGINLayer(nn.Module):
    def __init__(
        self,
        input_dim,
        hidden_dim,
        num_rels=1,
        regularizer="basis",
        num_bases=-1,
        num_mlp_layers=2,
        self_loop=True,
        bias=True,
        batch_norm=False,
        act_func="relu",
        dropout=0.0,
    ):
        super(RGINLayer, self).__init__()
        assert regularizer in ["none", "basis", "bdd"]

        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_rels = num_rels
        self.regularizer = regularizer
        if regularizer == "none" or num_bases is None or num_bases > num_rels or num_bases <= 0:
            self.num_bases = num_rels
        else:
            self.num_bases = num_bases
        # parameter creation order follows rgin.py:42-88 so the RNG stream (and hence the initial weights under a
        # given torch.manual_seed) is identical to the reference
        if self_loop:
            self.loop_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        else:
            self.register_parameter("loop_weight", None)
        if bias:
            self.bias = nn.Parameter(th.empty(hidden_dim))
        else:
            self.register_parameter("bias", None)
        mlp = []
        for i in range(num_mlp_layers):
            mlp.append(nn.Linear(hidden_dim, hidden_dim))
            if i != num_mlp_layers - 1:
                if batch_norm:
                    mlp.append(nn.BatchNorm1d(hidden_dim))
                mlp.append(map_activation_str_to_layer(act_func))
        self.mlp = nn.Sequential(*mlp)
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)

        if regularizer == "none" or regularizer == "basis":
            self.weight = nn.Parameter(th.empty(self.num_bases, self.input_dim, self.hidden_dim))
            if self.num_bases < self.num_rels:
                self.w_comp = nn.Parameter(th.empty(self.num_rels, self.num_bases))
            else:
                self.register_parameter("w_comp", None)
        else:  # bdd
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

    def forward_residual(self, g, node_feat, edge_type):
        """node_feat + forward(g, node_feat, edge_type)[0] -- the representation nets' residual connection (rgin.py:243-245) -- from
        the layer's own launches where they can carry it (the fp32 layer function with dropout off), else the plain sum."""
        fuse = self.input_dim == self.hidden_dim and (self.drop.p == 0.0 or not self.training)
        out, edge_type = self.forward(g, node_feat, edge_type, _residual=fuse)
        if getattr(self, "_residual_done", False):
            self._residual_done = False
            return out, edge_type
        return node_feat + out, edge_type

    def forward(self, g, node_feat, edge_type, _residual=False):
        # side effects on the graph as in rgin.py:126-135,160
        g.ndata[NODEFEAT] = node_feat
        g.edata[EDGETYPE] = edge_type
        self._residual_done = False
        W = dense_relation_weights(self)
        if ops.fused_path_supported(node_feat, W):
            # bf16: message pass, self loop (rgin.py:140-142) and bias in ONE row-factorised MFMA pipeline
            index = g.row_index(edge_type, self.num_rels, self.self_loop, closing_hint=(node_feat.shape[1], node_feat.dtype))
            small = self._small_layer(node_feat, W, index, _residual)
            if small is not None:
                return self.drop(small), edge_type
            out = ops.rel_transform_fused(node_feat, W, self.bias if self.self_loop else None, index,
                                          W_loop=self.loop_weight if self.self_loop else None)
            if self.bias is not None and not self.self_loop:
                out = out + self.bias
        else:
            index = g.rel_index(edge_type, self.num_rels)
            out = ops.rel_agg_transform(node_feat, W, index)        # sum_e x[src] W[etype]  (rgin.py:102-120 + fn.sum)
            if self.self_loop:
                out = out + th.matmul(node_feat, self.loop_weight)  # rgin.py:140-142
            if self.bias is not None:
                out = out + self.bias
        if len(self.mlp) > 0:
            out, act_done = self._run_mlp(out)
        else:
            out, act_done = self.act(out), False
        if not act_done:
            out = self.act(out)                                     # activation after the MLP (twice if MLP empty)
        out = self.drop(out)
        return out, edge_type

    def _small_layer(self, node_feat, W, index, residual=False):
        """The whole layer -- conv, bias, Linear-act-Linear-act -- as ops.rgin_layer_small where it applies (bf16, the reference's default
        width 64 (config.py:456-461), a batch of small graphs, the MLP a plain Linear / activation chain with the layer's own `relu` /
        `leaky_relu`): seven launches a training step instead of eleven.  None: take the general route."""
        if not self.self_loop:
            return None
        mods = list(self.mlp)
        linears = [m for m in mods if isinstance(m, nn.Linear)]
        slope = _fused_slope(self.act)
        if slope is None or len(mods) != 3 or not all(isinstance(m, nn.Linear) or _fused_slope(m) == slope for m in mods):
            return None
        if ops.rgin_layer_small_ok(node_feat, W, self.loop_weight, self.bias, linears, index):
            return ops.rgin_layer_small(node_feat, W, self.loop_weight, self.bias, linears, slope, index)
        if ops.rgin_layer_wide_ok(node_feat, W, self.loop_weight, self.bias, linears, index):
            # H = 256 in bf16: the launches of the general route with ONE weight-gradient launch for the conv and both Linears
            return ops.rgin_layer_wide(node_feat, W, self.loop_weight, self.bias, linears, slope, index)
        if ops.rgin_layer_f32_ok(node_feat, W, self.loop_weight, self.bias, linears, index):
            # the reference's own precision: the same launches as the separate functions, ONE weight-gradient launch in the backward
            # (residual: node_feat + layer(node_feat) leaves the MLP launch; forward_residual is told through _residual_done)
            self._residual_done = bool(residual)
            return ops.rgin_layer_f32(node_feat, W, self.loop_weight, self.bias, linears, slope, index, residual=residual)
        return None

    def _run_mlp(self, out):
        """self.mlp(out), with every Linear (and the ReLU / leaky ReLU that follows it -- including the layer's final activation)
        sent through the fused MFMA Linear + bias + activation kernels when the dtype / width allow it: `relu` and the reference
        CLI's default `leaky_relu` (slope 1 / 5.5, config.py:329-335) take the same launches.  Returns (out, final_act_applied)."""
        mods = list(self.mlp)
        linears = [m for m in mods if isinstance(m, nn.Linear)]
        slope = _fused_slope(self.act)
        if (slope is not None and all(isinstance(m, nn.Linear) or _fused_slope(m) == slope for m in mods)
                and ops.relu_mlp_supported(out, linears)):
            return ops.relu_mlp(out, linears, slope), True          # Linear-act chain + final activation, fused end to end
        i, act_done = 0, False
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                nxt = mods[i + 1] if i + 1 < len(mods) else self.act
                fuse = isinstance(nxt, nn.ReLU)
                out = ops.linear_act(out, m.weight, m.bias, relu=fuse)
                if fuse:
                    if i + 1 < len(mods):
                        i += 1
                    else:
                        act_done = True
            elif isinstance(m, nn.BatchNorm1d) and m.training and ops.batch_norm_rows_supported(out) and out.shape[0] > 1:
                # --rep_rgin_batch_norm (rgin.py:53-54): statistics, running buffers, normalisation and the ReLU behind it on the HIP
                # BatchNorm kernels (dn_batchnorm_rows_*: three small launches each way) instead of a torch / MIOpen module + activation
                from ..graph_classification.models import hip_batch_norm_forward
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                out = hip_batch_norm_forward(m, out, fuse)
                if fuse:
                    i += 1
            else:
                out = m(out)
            i += 1
        return out, act_done

    def get_output_dim(self):
        return self.hidden_dim

    def extra_repr(self):
        return "\n".join([
            "in=%d, out=%d," % (self.input_dim, self.hidden_dim),
            "num_rels=%d, regularizer=%s, num_bases=%d," % (self.num_rels, self.regularizer, self.num_bases),
            "self_loop=%s, bias=%s," % (self.self_loop, self.bias is not None),
        ])


class RGINRepNet(nn.Module):
    """The representation stack RGIN builds in create_rep_net / get_graph_rep (rgin.py:179-260): a ModuleList of
    RGINLayer applied with optional residual, zero-mask (pattern side) or multiplicative gate (graph side).
    The embedding / prediction nets around it (basemodel.py) are outside the hot path."""

    def __init__(self, hid_dim, num_rels, num_layers=1, rep_residual=True, regularizer="basis", num_bases=-1,
                 num_mlp_layers=2, batch_norm=False, act_func="relu", dropout=0.0, name="graph"):
        super().__init__()
        self.rep_residual = rep_residual
        layers = nn.ModuleList()
        for i in range(num_layers):
            layers.add_module("%s_rgin_(%d)" % (name, i), RGINLayer(
                hid_dim, hid_dim, num_rels=num_rels, regularizer=regularizer, num_bases=num_bases,
                num_mlp_layers=num_mlp_layers, batch_norm=batch_norm, act_func=act_func, dropout=dropout))
        self.rgin = layers

    def get_pattern_rep(self, pattern, p_emb, mask=None):
        if mask is not None:
            p_zero_mask = ~mask
            outputs = [p_emb.masked_fill(p_zero_mask, 0.0)]
            etype = pattern.edata["label"]
            for layer in self.rgin:
                o, etype = layer(pattern, outputs[-1], etype)
                outputs.append(o.masked_fill(p_zero_mask, 0.0))
            return outputs[-1]
        return self.get_graph_rep(pattern, p_emb)

    def get_graph_rep(self, graph, g_emb, mask=None, gate=None):
        etype = graph.edata["label"]
        if mask is None and gate is None:
            outputs = [g_emb]
            for layer in self.rgin:
                if self.rep_residual and layer.input_dim == layer.hidden_dim:
                    o, etype = layer.forward_residual(graph, outputs[-1], etype)      # (= outputs[-1] + layer(...), fused where it can be)
                    outputs.append(o)
                    continue
                o, etype = layer(graph, outputs[-1], etype)
                outputs.append(outputs[-1] + o if self.rep_residual and outputs[-1].size() == o.size() else o)
            return outputs[-1]
        if gate is None:
            gate = mask.to(g_emb.dtype)
        elif mask is not None:
            gate = mask.to(g_emb.dtype) * gate
        outputs = [g_emb * gate]
        for layer in self.rgin:
            o, etype = layer(graph, outputs[-1], etype)
            o = o * gate
            outputs.append(outputs[-1] + o if self.rep_residual and outputs[-1].size() == o.size() else o)
        return outputs[-1]

    forward = get_graph_rep
