"""Dual (node + edge) message-passing layers of the subgraph-isomorphism models on the MI355X kernels.

Mirrors of  CompGCNLayer  subgraph_isomorphism/models/compgcn.py:104-283
            DMPLayer      subgraph_isomorphism/models/dmpnn.py:16-187
(constructor arguments, parameter names, `forward(graph, node_feat, edge_feat) -> (node_out, edge_out)`, the side
effects on graph.ndata / graph.edata that later layers read).  The reference evaluates a [E, H] x [H, H] product per
edge and direction inside `update_all`; every message here is linear in the weights, so the per-destination sums are
taken FIRST on the gather/segment-sum kernels (dn_gather_segsum_*: rows of x by source, rows of the edge features by
edge id, per-edge scalar = norm x direction mask) and the products run on [N, H] rows:

  CompGCN   agg[v] = (sum_{e->v, fwd} n_e c_e) W_in + (sum_{e->v, rev} n_e c_e) W_out,   c_e = comp(x[src_e], ef_e)
            'sub' never materialises c_e (two segment sums); 'mult' / 'corr' compose per edge on gathered rows
  DMP       node:  -(sum_{fwd} ef_e) W_in + (sum_{rev} ef_e) W_out + x W_nloop
            edge:  (x W_dst)[a_e] - (x W_src)[b_e] + ef_e W_eloop + 2 (1 + log2(1 + outdeg[dst_e])) ef_e (W_src - W_dst)
                   with (a, b) = (dst, src) on forward edges and (src, dst) on reversed ones
GPU only.
"""
import torch as th
import torch.nn as nn

from .. import ops
from .act import map_activation_str_to_layer
from .init import init_weight

NODEFEAT, EDGEFEAT = "node_feat", "edge_feat"
REVFLAG, NORM = "is_reversed", "norm"
INDEGREE, INNORM, OUTDEGREE, OUTNORM = "in_deg", "in_norm", "out_deg", "out_norm"


def _edge_index(g):
    """Cached ops.EdgeIndex of a BatchedGraph (CSR by destination + CSC by source)."""
    ix = getattr(g, "_dual_index", None)
    if ix is None:
        src, dst = g.all_edges()
        ix = ops.EdgeIndex(src, dst, g.number_of_nodes())
        try:
            g._dual_index = ix
        except Exception:
            pass
    return ix


def _dense(x, w):
    """x @ w with w [in, out]: matrix-core Linear when the width allows it, else rocBLAS."""
    # exact fp32 products here: these layers chain several products through norms and compositions, and are tiny
    return ops.linear_act(x, w.t(), exact=True) if w.shape[0] == w.shape[1] else th.matmul(x, w)


def _degrees(g, ix):
    if INDEGREE not in g.ndata or OUTDEGREE not in g.ndata:
        ind, outd = ops.degrees(ix.src, ix.dst, ix.num_nodes)
        g.ndata.setdefault(INDEGREE, ind.long())
        g.ndata.setdefault(OUTDEGREE, outd.long())
    return g.ndata[INDEGREE], g.ndata[OUTDEGREE]


def _recip_norm(deg, self_loop):
    deg = deg.float()
    if self_loop:
        return (deg + 1).reciprocal().unsqueeze(-1)
    return deg.reciprocal().masked_fill_(deg == 0, 1.0).unsqueeze(-1)


def _circular_correlation(a, b):
    n = a.shape[-1]
    return th.fft.irfft(th.conj(th.fft.rfft(a.float(), dim=-1)) * th.fft.rfft(b.float(), dim=-1), n=n, dim=-1).to(a.dtype)


class CompGCNLayer(nn.Module):
    def __init__(self, input_dim, hidden_dim, self_loop=True, comp_opt="mult", edge_norm="both", bias=True,
                 batch_norm=False, act_func="relu", dropout=0.0):
        super().__init__()
        assert edge_norm in ["none", "in", "out", "both"]
        self.input_dim, self.hidden_dim, self.edge_norm, self.comp_opt = input_dim, hidden_dim, edge_norm, comp_opt
        self.num_rels = 3 if self_loop else 2
        if self_loop:
            self.loop_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        else:
            self.register_parameter("loop_weight", None)
        if bias:
            self.bias = nn.Parameter(th.empty(hidden_dim))
        else:
            self.register_parameter("bias", None)
        self.bn = nn.BatchNorm1d(hidden_dim) if batch_norm else None
        self.in_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        self.out_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        self.rel_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        if self_loop:
            self.loop_rel = nn.Parameter(th.empty(1, input_dim))
        else:
            self.register_parameter("loop_rel", None)
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)
        for w in (self.in_weight, self.out_weight, self.rel_weight):
            init_weight(w, activation=act_func, init="uniform")
        if self_loop:
            init_weight(self.loop_weight, activation=act_func, init="uniform")
            init_weight(self.loop_rel, activation=act_func, init="uniform")
        if bias:
            nn.init.zeros_(self.bias)

    @property
    def self_loop(self):
        return hasattr(self, "loop_weight") and self.loop_weight is not None

    def _comp(self, head, relation):
        if self.comp_opt == "sub":
            return head - relation
        if self.comp_opt == "mult":
            return head * relation
        if self.comp_opt == "corr":
            return _circular_correlation(head, relation.expand_as(head) if relation.shape[0] == 1 else relation)
        raise NotImplementedError

    def _edge_scale(self, g, ix):
        """graph.edata[norm] of compgcn.py:200-209 ([E,1]) as a flat fp32 per-edge scalar, or None."""
        if self.edge_norm == "none":
            return None
        ind, outd = _degrees(g, ix)
        if self.edge_norm in ("in", "both") and INNORM not in g.ndata:
            g.ndata[INNORM] = _recip_norm(ind, self.self_loop)
        if self.edge_norm in ("out", "both") and OUTNORM not in g.ndata:
            g.ndata[OUTNORM] = _recip_norm(outd, self.self_loop)
        src, dst = ix.src.long(), ix.dst.long()
        if self.edge_norm == "in":
            n = g.ndata[INNORM][dst]
        elif self.edge_norm == "out":
            n = g.ndata[OUTNORM][src]
        else:
            n = (g.ndata[OUTNORM][src] * g.ndata[INNORM][dst]) ** 0.5
        g.edata[NORM] = n
        return n.reshape(-1).float().contiguous()

    def forward(self, graph, node_feat, edge_feat):
        g = graph
        g.ndata[NODEFEAT], g.edata[EDGEFEAT] = node_feat, edge_feat
        ix = _edge_index(g)
        scale = self._edge_scale(g, ix)
        rev = g.edata[REVFLAG].reshape(-1).bool() if REVFLAG in g.edata else None
        ones = th.ones(ix.num_edges, dtype=th.float32, device=node_feat.device) if (scale is None and rev is not None) else None
        base = scale if scale is not None else ones
        parts = [(self.in_weight, base if rev is None else base * (~rev).float())]
        if rev is not None:
            parts.append((self.out_weight, base * rev.float()))
        agg = None
        for w, sc in parts:
            if self.comp_opt == "sub":                               # sum n_e (x[src_e] - ef_e): two segment sums, no [E,H] temp
                a = ops.neighbor_sum(node_feat, ix, 0.0, sc) - ops.edge_sum(edge_feat, ix, sc)
            else:
                comp = self._comp(ops.gather_rows(node_feat, ix.src, (ix.out_ptr, ix.out_perm)), edge_feat)
                a = ops.edge_sum(comp, ix, sc)
            t = _dense(a, w)
            agg = t if agg is None else agg + t
        if self.self_loop:
            out = (agg + _dense(self._comp(node_feat, self.loop_rel), self.loop_weight)) * 0.3333333
        else:
            out = agg * 0.5
        if self.bias is not None:
            out = out + self.bias
        if self.bn is not None:
            out = self.bn(out)
        out = self.drop(self.act(out))
        return out, _dense(edge_feat, self.rel_weight)

    def get_output_dim(self):
        return self.hidden_dim

    def extra_repr(self):
        return "\n".join(["in=%s, out=%s," % (self.input_dim, self.hidden_dim), "comp_opt=%s," % self.comp_opt,
                          "edge_norm=%s, self_loop=%s, bias=%s," % (self.edge_norm, self.self_loop, self.bias is not None)])


class DMPLayer(nn.Module):
    def __init__(self, input_dim, hidden_dim, init_neigenv=4.0, init_eeigenv=4.0, bias=True, num_mlp_layers=2,
                 batch_norm=True, act_func="relu", dropout=0.0):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        names = ("in_weight", "out_weight", "src_weight", "dst_weight", "nloop_weight", "eloop_weight")
        for n in names:
            setattr(self, n, nn.Parameter(th.empty(input_dim, hidden_dim)))
        if bias:
            self.nbias, self.ebias = nn.Parameter(th.empty(hidden_dim)), nn.Parameter(th.empty(hidden_dim))
        else:
            self.register_parameter("nbias", None)
            self.register_parameter("ebias", None)

        def mlp():
            mods = []
            for i in range(num_mlp_layers):
                mods.append(nn.Linear(hidden_dim, hidden_dim))
                if i != num_mlp_layers - 1:
                    if batch_norm:
                        mods.append(nn.BatchNorm1d(hidden_dim))
                    mods.append(map_activation_str_to_layer(act_func))
            return nn.Sequential(*mods)

        self.nmlp, self.emlp = mlp(), mlp()
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)
        for n in names:
            init_weight(getattr(self, n), activation=act_func, init="uniform")
        for seq in (self.nmlp, self.emlp):
            for m in seq.modules():
                if isinstance(m, nn.Linear):
                    init_weight(m.weight, activation=act_func, init="uniform")
                    nn.init.zeros_(m.bias)
        if bias:
            nn.init.zeros_(self.nbias)
            nn.init.zeros_(self.ebias)
        with th.no_grad():                                           # "reparamerization tricks", dmpnn.py:80-87
            for n in ("in_weight", "out_weight", "nloop_weight"):
                getattr(self, n).data.div_(init_neigenv)
            for n in ("src_weight", "dst_weight", "eloop_weight"):
                getattr(self, n).data.div_(init_eeigenv)

    def _run(self, seq, h):
        if len(seq) == 0:
            return self.act(h)
        for m in seq:
            h = ops.linear_act(h, m.weight, m.bias, exact=True) if isinstance(m, nn.Linear) else m(h)
        return h

    def forward(self, graph, node_feat, edge_feat):
        g = graph
        g.ndata[NODEFEAT], g.edata[EDGEFEAT] = node_feat, edge_feat
        ix = _edge_index(g)
        if OUTDEGREE not in g.ndata:
            g.ndata[OUTDEGREE] = _degrees(g, ix)[1]
        rev = g.edata[REVFLAG].reshape(-1).bool() if REVFLAG in g.edata else None
        src, dst = ix.src.long(), ix.dst.long()
        # node side: -(sum_fwd ef) W_in + (sum_rev ef) W_out + x W_nloop
        if rev is None:
            agg = -_dense(ops.edge_sum(edge_feat, ix), self.in_weight)
            a_idx, b_idx = ix.dst, ix.src
        else:
            agg = (_dense(ops.edge_sum(edge_feat, ix, rev.float()), self.out_weight)
                   - _dense(ops.edge_sum(edge_feat, ix, (~rev).float()), self.in_weight))
            a_idx, b_idx = th.where(rev, src, dst).to(th.int32), th.where(rev, dst, src).to(th.int32)
        h = _dense(node_feat, self.nloop_weight) + agg
        if self.nbias is not None:
            h = h + self.nbias
        node_out = self.drop(self._run(self.nmlp, h))
        # edge side
        xd, xs = _dense(node_feat, self.dst_weight), _dense(node_feat, self.src_weight)
        edge_msg = ops.gather_rows(xd, a_idx) - ops.gather_rows(xs, b_idx)
        d = (1 + g.ndata[OUTDEGREE][dst].unsqueeze(-1).float()).log2()
        add = (2 * (1 + d)).to(edge_feat.dtype) * _dense(edge_feat, self.src_weight - self.dst_weight)
        e = _dense(edge_feat, self.eloop_weight) + add + edge_msg
        if self.ebias is not None:
            e = e + self.ebias
        return node_out, self.drop(self._run(self.emlp, e))

    def get_output_dim(self):
        return self.hidden_dim

    def extra_repr(self):
        return "in=%s, out=%s" % (self.input_dim, self.hidden_dim)
