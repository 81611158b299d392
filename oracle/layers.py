"""ORACLE (test infrastructure, not product code): fp32 CPU restatement of the reference's
message-passing layers in plain torch ops (autograd gives the reference gradients).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this.  The product path never does.

Parity status
  * SI ``RGINLayer`` / ``RGCNLayer``: PINNED -- checked against outputs + all gradients of the
    reference's own modules (run under the stand-ins of tests/golden/_ref_standins.py) for the
    regulariser x activation x self_loop x edge_norm grid in tests/golden/si_layers_*.npz.
  * GC ``GINConv`` / ``RGCNConv`` arithmetic: the algorithm lives in torch-geometric==2.0.2
    (README.md:26), which is NOT under /root/reference and not installed: **parity unpinned**
    for those two; restated from the published layer definitions (GIN: nn((1+eps) x_i + sum_j x_j);
    RGCNConv: sum_r aggr_{j in N_r(i)} x_j W_r + x_i root + b) and anchored on the reference's
    call sites gconv.py:197,212 and rgconv.py:17-18,40-41,96,121.
"""
import math

import torch as th
import torch.nn.functional as F

LEAKY_RELU_A = 1 / 5.5  # reference: SI constants.py:10, utils/act.py:27


def act_fn(name):
    """reference: SI utils/act.py:457-474 (subset reachable from RGIN/RGCN configs)."""
    if name == "none":
        return lambda x: x
    if name == "relu":
        return F.relu
    if name == "leaky_relu":
        return lambda x: F.leaky_relu(x, LEAKY_RELU_A)
    if name == "tanh":
        return th.tanh
    if name == "sigmoid":
        return th.sigmoid
    if name == "gelu":
        return F.gelu
    if name == "elu":
        return F.elu
    if name == "selu":
        return F.selu
    raise NotImplementedError(name)


def batch_norm_train(weight, bias, eps=1e-5):
    """Training-mode BatchNorm1d over the rows (torch.nn.BatchNorm1d as the reference's MLP holds it, SI models/rgin.py:53-54):
    y = (x - mean) / sqrt(biased var + eps) * weight + bias.  Returns the callable rgin_layer's mlp_bn takes; it records the batch
    statistics it saw (mean, UNBIASED var: what the running buffers are updated with)."""
    def f(x):
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        f.stats = (mean.detach(), x.var(0, unbiased=True).detach())
        return (x - mean) / th.sqrt(var + eps) * weight + bias
    return f


def segment_sum(rows, index, num_segments):
    """sum rows by destination -- DGL fn.sum / torch_scatter 'sum'."""
    out = th.zeros((num_segments,) + tuple(rows.shape[1:]), dtype=rows.dtype)
    return out.index_add(0, index, rows)


def relation_weights(weight, w_comp, regularizer, num_rels, num_bases, in_dim, out_dim):
    """Materialise dense [R, in, out] weights.  reference: SI models/rgin.py:103-108, 114-117."""
    if regularizer in ("none", "basis"):
        if w_comp is not None:
            return th.matmul(w_comp, weight.view(num_bases, in_dim * out_dim)).view(num_rels, in_dim, out_dim)
        return weight
    # bdd: block-diagonal with num_bases blocks
    si, so = in_dim // num_bases, out_dim // num_bases
    blocks = weight.view(num_rels, num_bases, si, so)
    dense = th.zeros(num_rels, in_dim, out_dim, dtype=weight.dtype)
    for b in range(num_bases):
        dense[:, b * si:(b + 1) * si, b * so:(b + 1) * so] = blocks[:, b]
    return dense


def _messages(x, src, etype, p, regularizer, num_rels, num_bases):
    """Per-edge transform-then-aggregate message, exactly the reference's formulation.
    reference: SI models/rgin.py:102-120 (== rgcn.py:100-122 before the norm)."""
    in_dim = x.shape[1]
    if regularizer in ("none", "basis"):
        out_dim = p["weight"].shape[2]
        W = relation_weights(p["weight"], p.get("w_comp"), regularizer, num_rels, num_bases, in_dim, out_dim)
        Wg = W.index_select(0, etype)
        return th.bmm(x[src].unsqueeze(1), Wg).squeeze(1)
    si = in_dim // num_bases
    so = p["weight"].shape[1] // (num_bases * si)
    Wg = p["weight"].index_select(0, etype).view(-1, si, so)
    return th.bmm(x[src].reshape(-1, 1, si), Wg).view(-1, num_bases * so)


def rgin_layer(x, src, dst, etype, p, regularizer="basis", num_rels=1, num_bases=-1,
               num_mlp_layers=2, act="relu", mlp_bn=None):
    """reference: SI models/rgin.py:137-160.  p: dict with weight, (w_comp), (loop_weight), (bias),
    mlp.{0,2,..}.{weight,bias}.  Activation after the MLP; twice if the MLP is empty (:147-151)."""
    if regularizer == "none" or num_bases is None or num_bases > num_rels or num_bases <= 0:
        num_bases = num_rels                                                   # (:38-41)
    f = act_fn(act)
    msg = _messages(x, src, etype, p, regularizer, num_rels, num_bases)
    out = segment_sum(msg, dst, x.shape[0])                                    # fn.sum (:98)
    if p.get("loop_weight") is not None:
        out = out + th.matmul(x, p["loop_weight"])                             # (:140-142)
    if p.get("bias") is not None:
        out = out + p["bias"]                                                  # (:145-146)
    if num_mlp_layers > 0:
        # Sequential(Linear, [BN], act, Linear, ...): module index steps by 2 (3 with BN) (:50-57)
        step = 3 if mlp_bn else 2
        for i in range(num_mlp_layers):
            out = F.linear(out, p["mlp.%d.weight" % (i * step)], p["mlp.%d.bias" % (i * step)])
            if i != num_mlp_layers - 1:
                if mlp_bn:
                    out = mlp_bn[i](out)
                out = f(out)
    else:
        out = f(out)
    return f(out)                                                              # (:151)


def rgcn_norms(src, dst, num_nodes, edge_norm, self_loop):
    """reference: SI models/rgcn.py:132-165."""
    in_deg = th.bincount(dst, minlength=num_nodes)
    out_deg = th.bincount(src, minlength=num_nodes)
    in_norm = out_norm = enorm = None
    if edge_norm in ("in", "both"):
        if self_loop:
            in_norm = (1.0 / (in_deg.float() + 1)).view(-1, 1)
        else:
            in_norm = (1.0 / in_deg.float()).masked_fill_(in_deg == 0, 0.0).view(-1, 1)
    if edge_norm in ("out", "both"):
        if self_loop:
            out_norm = (1.0 / (out_deg.float() + 1)).view(-1, 1)
        else:
            out_norm = (1.0 / out_deg.float()).masked_fill_(out_deg == 0, 0.0).view(-1, 1)
    if edge_norm == "in":
        enorm = in_norm[dst]
    elif edge_norm == "both":
        enorm = (out_norm[src] * in_norm[dst]) ** 0.5
    return in_norm, out_norm, enorm


def rgcn_layer(x, src, dst, etype, p, regularizer="basis", num_rels=1, num_bases=-1,
               edge_norm="in", act="relu", bn=None):
    """reference: SI models/rgcn.py:100-197."""
    if regularizer == "none" or num_bases is None or num_bases > num_rels or num_bases <= 0:
        num_bases = num_rels
    self_loop = p.get("loop_weight") is not None
    in_norm, out_norm, enorm = rgcn_norms(src, dst, x.shape[0], edge_norm, self_loop)
    msg = _messages(x, src, etype, p, regularizer, num_rels, num_bases)
    if edge_norm != "none":
        msg = msg * enorm                                                      # (:110-111)
    out = segment_sum(msg, dst, x.shape[0])
    if self_loop:
        loop_msg = th.matmul(x, p["loop_weight"])
        if edge_norm == "in":
            out = out + loop_msg * in_norm                                     # (:174-175)
        elif edge_norm == "both":
            out = out + loop_msg * (in_norm * out_norm) ** 0.5                 # (:178-179)
        else:
            out = out + loop_msg
    if p.get("bias") is not None:
        out = out + p["bias"]
    if bn is not None:
        out = bn(out)
    return act_fn(act)(out)


# ---- aggregate-then-transform form: same math, no [E,H,H] temporary; used for
# ---- the CPU baseline timing at sizes where the reference's own formulation cannot allocate.
def rgin_layer_agg_first(x, src, dst, etype, p, num_rels, act="relu", num_mlp_layers=2):
    N, H = x.shape
    f = act_fn(act)
    seg = dst * num_rels + etype
    A = segment_sum(x[src], seg, N * num_rels).view(N, num_rels * H)          # [N, R*H]
    out = A @ p["weight"].reshape(num_rels * H, -1)                            # sum_r A_r W_r
    if p.get("loop_weight") is not None:
        out = out + x @ p["loop_weight"]
    if p.get("bias") is not None:
        out = out + p["bias"]
    for i in range(num_mlp_layers):
        out = F.linear(out, p["mlp.%d.weight" % (2 * i)], p["mlp.%d.bias" % (2 * i)])
        if i != num_mlp_layers - 1:
            out = f(out)
    if num_mlp_layers == 0:
        out = f(out)
    return f(out)


# --------------------------------------------------------------------------------------
# GC side (PyG 2.0.2 layer definitions; parity unpinned, see module docstring)
# --------------------------------------------------------------------------------------
def gin_conv(x, src, dst, eps, nn):
    """GINConv: nn((1 + eps) * x_i + sum_{j->i} x_j).  call site: gconv.py:197,212."""
    return nn((1.0 + eps) * x + segment_sum(x[src], dst, x.shape[0]))


def rgcn_conv(x, src, dst, etype, weight, root, bias, aggr="mean"):
    """RGCNConv (no bases/blocks): sum_r aggr_{j in N_r(i)} x_j @ W_r + x_i @ root + bias.
    call sites: rgconv.py:17-18,40-41 (mean) and :96,121 (add)."""
    N = x.shape[0]
    R = weight.shape[0]
    out = th.zeros(N, weight.shape[2], dtype=x.dtype)
    for r in range(R):
        m = etype == r
        h = segment_sum(x[src[m]], dst[m], N)
        if aggr == "mean":
            cnt = th.bincount(dst[m], minlength=N).clamp(min=1).to(x.dtype).view(-1, 1)
            h = h / cnt
        out = out + h @ weight[r]
    out = out + x @ root
    if bias is not None:
        out = out + bias
    return out


def global_pool(x, batch, num_graphs, kind="add"):
    """global_add_pool / global_mean_pool / global_max_pool.  call sites gconv.py:53,95,148,210,213."""
    if kind == "add":
        return segment_sum(x, batch, num_graphs)
    if kind == "mean":
        cnt = th.bincount(batch, minlength=num_graphs).clamp(min=1).to(x.dtype).view(-1, 1)
        return segment_sum(x, batch, num_graphs) / cnt
    if kind == "max":
        out = th.full((num_graphs, x.shape[1]), -math.inf, dtype=x.dtype)
        out = out.scatter_reduce(0, batch.view(-1, 1).expand_as(x), x, reduce="amax", include_self=True)
        return out.masked_fill(out == -math.inf, 0.0)
    raise ValueError(kind)


def xavier_uniform_bound(shape, act):
    """Bound 'a' of the reference's custom Xavier-uniform.  reference: SI utils/init.py:52-75."""
    shape = tuple(shape) if len(shape) >= 2 else tuple(shape) + (1,)
    rf = 1
    for s in shape[2:]:
        rf *= s
    fan_in, fan_out = shape[1] * rf, shape[0] * rf
    if act in ("none",):
        gain = 1.0
    elif act in ("relu", "relu6", "elu", "selu", "celu", "gelu"):
        gain = math.sqrt(2.0)
    elif act in ("leaky_relu", "prelu"):
        gain = math.sqrt(2.0 / (1 + LEAKY_RELU_A ** 2))
    elif act == "tanh":
        gain = 5.0 / 3
    else:  # sigmoid / softmax family
        gain = 1.0
    return 1.7320508075688772 * gain * math.sqrt(2.0 / float(fan_in + fan_out))


def rgin_layer_rel_grouped(x, src, dst, etype, p, num_rels, act="relu", num_mlp_layers=2):
    """CPU-baseline port: the reference's per-edge `x[src] @ W[etype]` message (rgin.py:102-112) evaluated relation by
    relation (one [E_r, H] x [H, H] GEMM per relation instead of the [E, H, H] weight gather + bmm, which cannot be
    allocated at the benchmark size), summed by destination (fn.sum), then the reference's node update (rgin.py:137-154)."""
    N = x.shape[0]
    f = act_fn(act)
    order = th.argsort(etype, stable=True)
    counts = th.bincount(etype, minlength=num_rels).tolist()
    out = th.zeros(N, p["weight"].shape[2], dtype=x.dtype)
    pos = 0
    for r in range(num_rels):
        e = order[pos:pos + counts[r]]
        pos += counts[r]
        if e.numel():
            out = out.index_add(0, dst[e], x[src[e]] @ p["weight"][r])
    if p.get("loop_weight") is not None:
        out = out + x @ p["loop_weight"]
    if p.get("bias") is not None:
        out = out + p["bias"]
    for i in range(num_mlp_layers):
        out = F.linear(out, p["mlp.%d.weight" % (2 * i)], p["mlp.%d.bias" % (2 * i)])
        if i != num_mlp_layers - 1:
            out = f(out)
    if num_mlp_layers == 0:
        out = f(out)
    return f(out)


def gcn_norm(src, dst, edge_weight, num_nodes):
    """torch_geometric.nn.conv.gcn_conv.gcn_norm (improved=False, add_self_loops=True, fill_value=1): existing self loops
    keep their weight, every other node gets a weight-1 self loop; norm_e = deg^-1/2[src] * w_e * deg^-1/2[dst] with
    deg = sum of weights INTO each node.  Returns (src', dst', norm) with the self loops appended.  (parity unpinned)"""
    if edge_weight is None:
        edge_weight = th.ones(src.numel(), dtype=th.float32)
    keep = src != dst
    loop_w = th.ones(num_nodes, dtype=edge_weight.dtype)
    if (~keep).any():
        loop_w = loop_w.index_put((src[~keep],), edge_weight[~keep])
    ar = th.arange(num_nodes)
    s2, d2 = th.cat([src[keep], ar]), th.cat([dst[keep], ar])
    w2 = th.cat([edge_weight[keep], loop_w])
    deg = th.zeros(num_nodes, dtype=w2.dtype).index_add(0, d2, w2)
    dis = deg.pow(-0.5)
    dis = th.where(th.isinf(dis), th.zeros_like(dis), dis)
    return s2, d2, dis[s2] * w2 * dis[d2]


def gcn_conv(x, src, dst, edge_weight, weight, bias):
    """GCNConv: sum_j norm_ij (x_j W) + b, W applied first (lin has no bias).  call sites gconv.py:36-37,51-52."""
    s2, d2, norm = gcn_norm(src, dst, edge_weight, x.shape[0])
    xl = x @ weight.t()
    out = segment_sum(xl[s2] * norm.view(-1, 1), d2, x.shape[0])
    return out + bias if bias is not None else out


def sage_conv(x, src, dst, lin_l_w, lin_l_b, lin_r_w, aggr="mean"):
    """SAGEConv: lin_l(aggr_{j->i} x_j) + lin_r(x_i); aggr in mean | max | add; empty neighbourhoods give 0.
    call site gconv.py:129-132.  (parity unpinned)"""
    N = x.shape[0]
    if aggr == "max":
        agg = th.full((N, x.shape[1]), -math.inf, dtype=x.dtype)
        agg = agg.scatter_reduce(0, dst.view(-1, 1).expand(-1, x.shape[1]), x[src], reduce="amax", include_self=True)
        agg = th.where(th.isinf(agg), th.zeros_like(agg), agg)
    else:
        agg = segment_sum(x[src], dst, N)
        if aggr == "mean":
            agg = agg / th.bincount(dst, minlength=N).clamp(min=1).to(x.dtype).view(-1, 1)
    return F.linear(agg, lin_l_w, lin_l_b) + F.linear(x, lin_r_w)


def split_and_batchify_graph_feats(batched_graph_feats, graph_sizes, pre_pad=False):
    """reference: subgraph_isomorphism/utils/dl.py:51-81 (restated; th.cat of per-graph slices and zero pads)."""
    bsz = graph_sizes.size(0)
    sizes = graph_sizes.view(-1).tolist()
    mx = max(sizes)
    if min(sizes) == mx:
        return batched_graph_feats.view(bsz, mx, -1), th.ones((bsz, mx), dtype=th.bool)
    feats, mask = [], th.zeros((bsz, mx), dtype=th.bool)
    idx = 0
    for i, l in enumerate(sizes):
        pad = th.zeros((mx - l,) + tuple(batched_graph_feats.shape[1:]), dtype=batched_graph_feats.dtype)
        rows = batched_graph_feats[idx:idx + l]
        feats.extend([pad, rows] if pre_pad else [rows, pad])
        if pre_pad:
            mask[i, mx - l:] = True
        else:
            mask[i, :l] = True
        idx += l
    return th.cat(feats, 0).view(bsz, mx, -1), mask


# --------------------------------------------------------------------------------------
# f-4  dual message passing: CompGCNLayer / DMPLayer
# reference: subgraph_isomorphism/models/compgcn.py:104-283, models/dmpnn.py:16-187
# --------------------------------------------------------------------------------------
def _deg_norm(deg, self_loop):
    """(deg + 1)^-1 with a self loop, else deg^-1 with 1.0 where deg == 0 (compgcn.py:180-196)."""
    deg = deg.to(th.float32)
    if self_loop:
        return (deg + 1).reciprocal().unsqueeze(-1)
    return deg.reciprocal().masked_fill(deg == 0, 1.0).unsqueeze(-1)


def _circular_correlation(a, b):
    """irfft(conj(rfft(a)) * rfft(b)) along the feature axis (compgcn.py:216-220)."""
    n = a.shape[-1]
    return th.fft.irfft(th.conj(th.fft.rfft(a, dim=-1)) * th.fft.rfft(b, dim=-1), n=n, dim=-1)


def compose(head, relation, comp_opt):
    if comp_opt == "sub":
        return head - relation
    if comp_opt == "mult":
        return head * relation
    if comp_opt == "corr":
        return _circular_correlation(head, relation.expand_as(head) if relation.shape[0] == 1 else relation)
    raise NotImplementedError(comp_opt)


def compgcn_layer(x, ef, src, dst, rev, p, comp_opt="mult", edge_norm="both", act="relu"):
    """-> (node_out, edge_out).  rev: bool [E] or None (no REVFLAG on the graph); p: parameter dict (loop_weight absent =
    no self loop).  Messages per edge, evaluated exactly as the reference's message function."""
    N = x.shape[0]
    self_loop = p.get("loop_weight") is not None
    in_deg, out_deg = th.bincount(dst, minlength=N), th.bincount(src, minlength=N)
    comp = compose(x[src], ef, comp_opt)
    msg = comp @ p["in_weight"]
    if rev is not None:
        msg = th.where(rev.view(-1, 1), comp @ p["out_weight"], msg)                         # compgcn.py:227-230
    if edge_norm == "in":
        msg = msg * _deg_norm(in_deg, self_loop)[dst]
    elif edge_norm == "out":
        msg = msg * _deg_norm(out_deg, self_loop)[src]
    elif edge_norm == "both":
        msg = msg * (_deg_norm(out_deg, self_loop)[src] * _deg_norm(in_deg, self_loop)[dst]) ** 0.5
    agg = segment_sum(msg, dst, N)
    if self_loop:
        out = (agg + compose(x, p["loop_rel"], comp_opt) @ p["loop_weight"]) * 0.3333333      # :240-244
    else:
        out = agg * 0.5
    if p.get("bias") is not None:
        out = out + p["bias"]
    return act_fn(act)(out), ef @ p["rel_weight"]


def _mlp_or_act(h, p, prefix, num_mlp_layers, act):
    """Linear(-act-Linear)* when the layer has an MLP (NO activation after the last Linear), else the activation."""
    f = act_fn(act)
    if num_mlp_layers == 0:
        return f(h)
    idx = 0
    for i in range(num_mlp_layers):
        h = h @ p["%s.%d.weight" % (prefix, idx)].t() + p["%s.%d.bias" % (prefix, idx)]
        idx += 1
        if i != num_mlp_layers - 1:
            h = f(h)
            idx += 1                                    # batch_norm=False: Sequential index skips only the activation
    return h


def dmp_layer(x, ef, src, dst, rev, p, num_mlp_layers=2, act="relu"):
    """-> (node_out, edge_out) of DMPLayer (batch_norm=False), dmpnn.py:111-169."""
    N = x.shape[0]
    out_deg = th.bincount(src, minlength=N)
    edge_msg = x[dst] @ p["dst_weight"] - x[src] @ p["src_weight"]
    node_msg = -(ef @ p["in_weight"])
    if rev is not None:
        r = rev.view(-1, 1)
        edge_msg = th.where(r, x[src] @ p["dst_weight"] - x[dst] @ p["src_weight"], edge_msg)
        node_msg = th.where(r, ef @ p["out_weight"], node_msg)
    h = x @ p["nloop_weight"] + segment_sum(node_msg, dst, N)
    if p.get("nbias") is not None:
        h = h + p["nbias"]
    node_out = _mlp_or_act(h, p, "nmlp", num_mlp_layers, act)
    d = (1 + out_deg[dst].unsqueeze(-1).float()).log2()
    e = ef @ p["eloop_weight"] + 2 * (1 + d) * (ef @ (p["src_weight"] - p["dst_weight"])) + edge_msg
    if p.get("ebias") is not None:
        e = e + p["ebias"]
    return node_out, _mlp_or_act(e, p, "emlp", num_mlp_layers, act)
