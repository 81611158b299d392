"""Stand-ins for the third-party modules the reference imports but this image lacks.

TEST TOOLING ONLY.  Used by ``make_golden.py`` (in the authoring container, where
``/root/reference`` is mounted) so the reference's *own* Python can be imported and run
unmodified to produce golden vectors.  Nothing here is imported by the product package
and nothing here is reference source: it is a minimal re-implementation of the public
behaviour of

* ``python-igraph`` 0.9 ``Graph`` (only the dozen methods the reference touches),
* ``dgl`` (``DGLGraph.update_all/apply_edges`` for a message UDF + ``fn.sum`` reducer),
* ``numba.jit`` (identity decorator), ``torch._six.container_abcs``,
* the three ``torch_geometric`` names ``tu_data_processing.py`` imports,
* ``torch_geometric.nn`` 2.0.2 (the pin of README.md:26): ``GINConv``, ``RGCNConv``/``FastRGCNConv``, ``GCNConv``,
  ``SAGEConv``, ``global_{add,mean,max}_pool`` -- the published layer definitions in plain CPU torch, including what
  their constructors do to the RNG stream (``reset_parameters`` runs in every constructor, GINConv re-initialises its
  ``nn``, PyG ``Linear`` initialises itself once before the conv re-initialises it), so that the reference's GC models
  (models/gconv.py, models/rgconv.py) can be imported and run unmodified.

igraph semantics relied on (python-igraph docs): ``add_vertices``/``add_edges`` append in
order; attribute sequences are plain per-element lists, new elements get ``None``;
``incident(v, mode="in")`` returns edge ids; ``delete_vertices`` drops incident edges and
renumbers the survivors compactly, preserving order.
DGL semantics relied on: ``update_all(msg_udf, fn.sum(msg, out), upd_udf)`` calls the UDF
once over all edges (eid order), sums messages by destination, then calls the node UDF
once over all nodes.
"""
import collections.abc
import math
import sys
import types

import torch as th


# --------------------------------------------------------------------------- igraph
class _Elem:
    def __init__(self, seq, i):
        self._seq, self._i = seq, i

    def __getitem__(self, k):
        return self._seq._attrs[k][self._i]

    def __setitem__(self, k, v):
        self._seq._ensure(k)
        self._seq._attrs[k][self._i] = v

    @property
    def index(self):
        return self._i

    @property
    def source(self):
        return self._seq._g._edges[self._i][0]

    @property
    def target(self):
        return self._seq._g._edges[self._i][1]

    @property
    def tuple(self):
        return self._seq._g._edges[self._i]


class _Seq:
    def __init__(self, g, kind):
        self._g, self._kind, self._attrs = g, kind, {}

    def _n(self):
        return self._g._n if self._kind == "v" else len(self._g._edges)

    def _ensure(self, k):
        if k not in self._attrs:
            self._attrs[k] = [None] * self._n()

    def _grow(self, cnt):
        for k in self._attrs:
            self._attrs[k].extend([None] * cnt)

    def __len__(self):
        return self._n()

    def __getitem__(self, k):
        if isinstance(k, str):
            return list(self._attrs[k])
        return _Elem(self, k)

    def __setitem__(self, k, v):
        v = list(v)
        if len(v) != self._n():
            # igraph cycles shorter sequences; the reference never relies on it
            raise ValueError("attribute length %d != %d" % (len(v), self._n()))
        self._attrs[k] = v

    def __delitem__(self, k):
        del self._attrs[k]


class Graph:
    def __init__(self, directed=False):
        self._directed = directed
        self._n = 0
        self._edges = []
        self.vs = _Seq(self, "v")
        self.es = _Seq(self, "e")

    def add_vertices(self, n):
        self._n += n
        self.vs._grow(n)

    def add_edges(self, edges):
        edges = [(int(a), int(b)) for a, b in edges]
        for a, b in edges:
            if not (0 <= a < self._n and 0 <= b < self._n):
                raise ValueError("vertex id out of range")
        self._edges.extend(edges)
        self.es._grow(len(edges))

    def vcount(self):
        return self._n

    def ecount(self):
        return len(self._edges)

    def vertex_attributes(self):
        return list(self.vs._attrs.keys())

    def edge_attributes(self):
        return list(self.es._attrs.keys())

    def get_edgelist(self):
        return list(self._edges)

    def incident(self, v, mode="out"):
        mode = str(mode).lower()
        if mode == "in":
            return [e for e, (_, t) in enumerate(self._edges) if t == v]
        if mode == "out":
            return [e for e, (s, _) in enumerate(self._edges) if s == v]
        return [e for e, (s, t) in enumerate(self._edges) if s == v or t == v]

    def indegree(self):
        d = [0] * self._n
        for _, t in self._edges:
            d[t] += 1
        return d

    def outdegree(self):
        d = [0] * self._n
        for s, _ in self._edges:
            d[s] += 1
        return d

    def delete_vertices(self, vids):
        dead = set(int(v) for v in vids)
        keep_v = [v for v in range(self._n) if v not in dead]
        remap = {v: i for i, v in enumerate(keep_v)}
        keep_e = [e for e, (s, t) in enumerate(self._edges) if s not in dead and t not in dead]
        for k in self.vs._attrs:
            self.vs._attrs[k] = [self.vs._attrs[k][v] for v in keep_v]
        for k in self.es._attrs:
            self.es._attrs[k] = [self.es._attrs[k][e] for e in keep_e]
        self._edges = [(remap[self._edges[e][0]], remap[self._edges[e][1]]) for e in keep_e]
        self._n = len(keep_v)


Graph.__module__ = "igraph"  # the reference dispatches on str(graph.__class__)


# --------------------------------------------------------------------------- dgl
class _Sum:
    def __init__(self, msg, out):
        self.msg, self.out = msg, out


class _CopyFrom:
    def __init__(self, target, in_field, out_field):
        self.target, self.in_field, self.out_field = target, in_field, out_field


class _TargetCode:
    SRC, DST, EDGE = 0, 1, 2


class _Gathered:
    def __init__(self, data, idx):
        self._data, self._idx = data, idx

    def __getitem__(self, k):
        return self._data[k][self._idx]


class _EdgeBatch:
    def __init__(self, g):
        self.src = _Gathered(g.ndata, g._u)
        self.dst = _Gathered(g.ndata, g._v)
        self.data = g.edata


class _NodeBatch:
    def __init__(self, g):
        self.data = g.ndata


class FakeDGLGraph:
    """Just enough of dgl.DGLGraph for RGINLayer/RGCNLayer.forward."""

    def __init__(self, u=(), v=(), num_nodes=0):
        self._u = th.as_tensor(u, dtype=th.long).reshape(-1)
        self._v = th.as_tensor(v, dtype=th.long).reshape(-1)
        self._n = int(num_nodes)
        self.ndata, self.edata = {}, {}

    @staticmethod
    def _extend(store, old_count, add_count, data):
        """dgl add_nodes/add_edges feature rule: keys missing on either side are zero-filled."""
        data = dict(data or {})
        for k in list(store.keys()):
            old = store[k]
            new = data.pop(k, None)
            if new is None:
                new = th.zeros((add_count,) + tuple(old.shape[1:]), dtype=old.dtype)
            store[k] = th.cat([old, th.as_tensor(new).to(old.dtype)], dim=0)
        for k, new in data.items():
            new = th.as_tensor(new)
            old = th.zeros((old_count,) + tuple(new.shape[1:]), dtype=new.dtype)
            store[k] = th.cat([old, new], dim=0)

    def add_nodes(self, num, data=None):
        self._extend(self.ndata, self._n, int(num), data)
        self._n += int(num)

    def add_edges(self, u, v, data=None):
        u = th.as_tensor(u, dtype=th.long).reshape(-1)
        v = th.as_tensor(v, dtype=th.long).reshape(-1)
        self._extend(self.edata, int(self._u.numel()), int(u.numel()), data)
        self._u = th.cat([self._u, u])
        self._v = th.cat([self._v, v])

    def all_edges(self, form="uv", order="eid"):
        assert order == "eid"
        if form == "uv":
            return self._u, self._v
        return self._u, self._v, th.arange(self._u.numel())

    edges = all_edges

    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._u.numel())

    def in_degrees(self):
        return th.bincount(self._v, minlength=self._n)

    def out_degrees(self):
        return th.bincount(self._u, minlength=self._n)

    def apply_edges(self, func):
        if isinstance(func, _CopyFrom):
            idx = self._u if func.target == _TargetCode.SRC else self._v
            self.edata[func.out_field] = self.ndata[func.in_field][idx]
        else:
            self.edata.update(func(_EdgeBatch(self)))

    def incidence_matrix(self, typestr):
        """dgl incidence_matrix("in"): sparse [N, E], entry (dst(e), e) = 1, stored in eid order (so a row's _indices() list
        the in-edges by ascending eid)."""
        assert typestr == "in"
        E = int(self._u.numel())
        idx = th.stack([self._v, th.arange(E)]) if E else th.zeros((2, 0), dtype=th.long)
        return th.sparse_coo_tensor(idx, th.ones(E), (self._n, E))

    def remove_nodes(self, nids):
        """dgl remove_nodes: drops the nodes and their incident edges, survivors renumbered compactly in order."""
        dead = th.zeros(self._n, dtype=th.bool)
        dead[th.as_tensor(nids, dtype=th.long)] = True
        keep_v = ~dead
        remap = th.cumsum(keep_v.long(), 0) - 1
        keep_e = keep_v[self._u] & keep_v[self._v] if self._u.numel() else th.zeros(0, dtype=th.bool)
        self.ndata = {k: v[keep_v] for k, v in self.ndata.items()}
        self.edata = {k: v[keep_e] for k, v in self.edata.items()}
        self._u, self._v = remap[self._u[keep_e]], remap[self._v[keep_e]]
        self._n = int(keep_v.sum())

    def update_all(self, message_func, reduce_func, apply_node_func=None):
        msgs = message_func(_EdgeBatch(self))
        assert isinstance(reduce_func, _Sum)
        m = msgs[reduce_func.msg]
        agg = th.zeros((self._n,) + tuple(m.shape[1:]), dtype=m.dtype).index_add(0, self._v, m)
        self.ndata[reduce_func.out] = agg
        if apply_node_func is not None:
            self.ndata.update(apply_node_func(_NodeBatch(self)))
        self.ndata.pop(reduce_func.out)


class DGLGraph(FakeDGLGraph):
    """FakeDGLGraph under the class name the reference's convert_conjugate_graph dispatches on
    (SI utils/graph.py:75-81: str(graph.__class__) == "<class 'dgl.graph.DGLGraph'>")."""


DGLGraph.__module__ = "dgl.graph"


# --------------------------------------------------------------------------- torch_geometric.nn (2.0.2)
def _glorot(t):
    """torch_geometric.nn.inits.glorot: U(-a, a), a = sqrt(6 / (size(-2) + size(-1)))."""
    if t is not None:
        a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
        t.data.uniform_(-a, a)


def _pyg_reset(nn):
    """torch_geometric.nn.inits.reset (2.0.2): the DIRECT children that have reset_parameters, else the module itself."""
    def _one(item):
        if hasattr(item, "reset_parameters"):
            item.reset_parameters()
    if nn is not None:
        if hasattr(nn, "children") and len(list(nn.children())) > 0:
            for item in nn.children():
                _one(item)
        else:
            _one(nn)


class _PygLinear(th.nn.Module):
    """torch_geometric.nn.dense.linear.Linear: weight [out, in]; weight_initializer None -> kaiming_uniform(fan=in,
    a=sqrt(5)) i.e. U(-1/sqrt(in), 1/sqrt(in)); 'glorot' -> glorot; bias_initializer None -> U(-1/sqrt(in), 1/sqrt(in))."""

    def __init__(self, in_channels, out_channels, bias=True, weight_initializer=None, bias_initializer=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight_initializer, self.bias_initializer = weight_initializer, bias_initializer
        self.weight = th.nn.Parameter(th.Tensor(out_channels, in_channels))
        if bias:
            self.bias = th.nn.Parameter(th.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight_initializer == "glorot":
            _glorot(self.weight)
        else:
            bound = math.sqrt(6.0 / ((1.0 + 5.0) * self.in_channels))
            self.weight.data.uniform_(-bound, bound)
        if self.bias is not None:
            if self.bias_initializer == "zeros":
                self.bias.data.fill_(0.0)
            else:
                bound = 1.0 / math.sqrt(self.in_channels)
                self.bias.data.uniform_(-bound, bound)

    def forward(self, x):
        return th.nn.functional.linear(x, self.weight, self.bias)


def _scatter_sum(src, index, n):
    return th.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add(0, index, src)


def _scatter_mean(src, index, n):
    cnt = th.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
    return _scatter_sum(src, index, n) / cnt.view(-1, *([1] * (src.dim() - 1)))


class _ScatterMax(th.autograd.Function):
    """torch_scatter.scatter_max (CPU): empty segments give 0; the gradient goes to the FIRST maximal entry."""

    @staticmethod
    def forward(ctx, src, index, n):
        out = th.full((n,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype)
        arg = th.full(out.shape, -1, dtype=th.long)
        for e in range(src.shape[0]):                      # golden sizes are tiny
            i = int(index[e])
            better = src[e] > out[i]
            out[i] = th.where(better, src[e], out[i])
            arg[i] = th.where(better, th.full_like(arg[i], e), arg[i])
        out = th.where(arg < 0, th.zeros_like(out), out)
        ctx.save_for_backward(arg)
        ctx.rows = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        gs = th.zeros((ctx.rows + 1,) + tuple(g.shape[1:]), dtype=g.dtype)
        gs.scatter_add_(0, th.where(arg < 0, th.full_like(arg, ctx.rows), arg), g)
        return gs[:-1], None, None


def _aggregate(msg, index, n, aggr):
    if aggr == "add":
        return _scatter_sum(msg, index, n)
    if aggr == "mean":
        return _scatter_mean(msg, index, n)
    if aggr == "max":
        return _ScatterMax.apply(msg, index, n)
    raise ValueError(aggr)


def global_add_pool(x, batch, size=None):
    return _scatter_sum(x, batch, int(batch.max()) + 1 if size is None else size)


def global_mean_pool(x, batch, size=None):
    return _scatter_mean(x, batch, int(batch.max()) + 1 if size is None else size)


def global_max_pool(x, batch, size=None):
    return _ScatterMax.apply(x, batch, int(batch.max()) + 1 if size is None else size)


class GINConv(th.nn.Module):
    """x_i' = nn((1 + eps) x_i + sum_{j -> i} x_j); eps a Parameter iff train_eps is truthy, else a buffer."""

    def __init__(self, nn, eps=0.0, train_eps=False, **kwargs):
        super().__init__()
        self.aggr = "add"
        self.nn = nn
        self.initial_eps = eps
        if train_eps:
            self.eps = th.nn.Parameter(th.Tensor([eps]))
        else:
            self.register_buffer("eps", th.Tensor([eps]))
        self.reset_parameters()

    def reset_parameters(self):
        _pyg_reset(self.nn)
        self.eps.data.fill_(self.initial_eps)

    def forward(self, x, edge_index, size=None):
        out = _aggregate(x[edge_index[0]], edge_index[1], x.shape[0], self.aggr)
        out = out + (1 + self.eps) * x
        return self.nn(out)


class RGCNConv(th.nn.Module):
    """x_i' = sum_r aggr_{j in N_r(i)} x_j W_r + x_i root + bias (no bases / blocks on the reference's call sites);
    parameters created weight, root, bias; reset: glorot(weight), glorot(root), zeros(bias)."""

    def __init__(self, in_channels, out_channels, num_relations, num_bases=None, num_blocks=None, aggr="mean",
                 root_weight=True, bias=True, **kwargs):
        super().__init__()
        assert num_bases is None and num_blocks is None
        self.aggr = aggr
        self.in_channels, self.out_channels, self.num_relations = in_channels, out_channels, num_relations
        self.weight = th.nn.Parameter(th.Tensor(num_relations, in_channels, out_channels))
        if root_weight:
            self.root = th.nn.Parameter(th.Tensor(in_channels, out_channels))
        else:
            self.register_parameter("root", None)
        if bias:
            self.bias = th.nn.Parameter(th.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        _glorot(self.weight)
        _glorot(self.root)
        if self.bias is not None:
            self.bias.data.fill_(0.0)

    def forward(self, x, edge_index, edge_type=None):
        n = x.shape[0]
        out = th.zeros(n, self.out_channels, dtype=x.dtype)
        for r in range(self.num_relations):
            m = edge_type == r
            h = _aggregate(x[edge_index[0][m]], edge_index[1][m], n, self.aggr)
            out = out + h @ self.weight[r]
        if self.root is not None:
            out = out + x @ self.root
        if self.bias is not None:
            out = out + self.bias
        return out


class GCNConv(th.nn.Module):
    """x' = D^-1/2 (A + I) D^-1/2 (x W) + b over weighted edges (gcn_norm with add_remaining_self_loops, fill 1)."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, add_self_loops=True, normalize=True,
                 bias=True, **kwargs):
        super().__init__()
        assert not improved and not cached and add_self_loops and normalize
        self.aggr = "add"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = _PygLinear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        if bias:
            self.bias = th.nn.Parameter(th.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()
        if self.bias is not None:
            self.bias.data.fill_(0.0)

    def forward(self, x, edge_index, edge_weight=None):
        n = x.shape[0]
        row, col = edge_index[0], edge_index[1]
        if edge_weight is None:
            edge_weight = th.ones(row.numel(), dtype=x.dtype)
        # add_remaining_self_loops: existing loops are moved to the end, one per node, keeping their weight
        keep = row != col
        loop_w = th.ones(n, dtype=edge_weight.dtype)
        if not bool(keep.all()):
            loop_w = loop_w.index_put((row[~keep],), edge_weight[~keep])
        ar = th.arange(n)
        row, col = th.cat([row[keep], ar]), th.cat([col[keep], ar])
        w = th.cat([edge_weight[keep], loop_w])
        deg = _scatter_sum(w, col, n)
        dis = deg.pow(-0.5)
        dis = dis.masked_fill(dis == float("inf"), 0)
        norm = dis[row] * w * dis[col]
        h = self.lin(x)
        out = _scatter_sum(norm.view(-1, 1) * h[row], col, n)
        if self.bias is not None:
            out = out + self.bias
        return out


class SAGEConv(th.nn.Module):
    """x_i' = lin_l(aggr_{j -> i} x_j) + lin_r(x_i); lin_l with bias, lin_r without; the constructor re-initialises both."""

    def __init__(self, in_channels, out_channels, normalize=False, root_weight=True, bias=True, **kwargs):
        super().__init__()
        assert not normalize and root_weight
        self.aggr = kwargs.get("aggr", "mean")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = _PygLinear(in_channels, out_channels, bias=bias)
        self.lin_r = _PygLinear(in_channels, out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x, edge_index, size=None):
        out = _aggregate(x[edge_index[0]], edge_index[1], x.shape[0], self.aggr)
        return self.lin_l(out) + self.lin_r(x)


# --------------------------------------------------------------------------- numba
class _NumbaType:
    def __getitem__(self, k):
        return self

    def __call__(self, *a, **k):
        return self


def _jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not isinstance(a[0], _NumbaType):
        return a[0]
    return lambda f: f


def install():
    """Register the stand-ins in sys.modules (idempotent)."""
    if "igraph" not in sys.modules:
        ig = types.ModuleType("igraph")
        ig.Graph = Graph
        sys.modules["igraph"] = ig
    if "dgl" not in sys.modules:
        dgl = types.ModuleType("dgl")
        dgl.DGLGraph = FakeDGLGraph
        fn = types.ModuleType("dgl.function")
        fn.sum = lambda msg, out: _Sum(msg, out)
        fn.copy_u = lambda u, out: _CopyFrom(_TargetCode.SRC, u, out)
        fn.CopyMessageFunction = _CopyFrom
        fn.TargetCode = _TargetCode
        dgl.function = fn
        sys.modules["dgl"] = dgl
        sys.modules["dgl.function"] = fn
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")
        nb.jit = _jit
        nb.njit = _jit
        for name in ("int64", "int32", "float32", "float64", "boolean", "void"):
            setattr(nb, name, _NumbaType())
        sys.modules["numba"] = nb
    if "tensorboardX" not in sys.modules:
        tbx = types.ModuleType("tensorboardX")
        tbx.SummaryWriter = object
        sys.modules["tensorboardX"] = tbx
    if "torch._six" not in sys.modules:
        six = types.ModuleType("torch._six")
        six.container_abcs = collections.abc
        six.string_classes = (str, bytes)
        six.int_classes = int
        sys.modules["torch._six"] = six
    if "torch_geometric" not in sys.modules:
        pyg = types.ModuleType("torch_geometric")
        pyg_data = types.ModuleType("torch_geometric.data")
        pyg_ds = types.ModuleType("torch_geometric.datasets")
        pyg_data.InMemoryDataset = object
        pyg_data.download_url = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no network"))
        pyg_data.extract_zip = pyg_data.download_url
        pyg_ds.TUDataset = object
        pyg_nn = types.ModuleType("torch_geometric.nn")
        for cls in (GINConv, RGCNConv, GCNConv, SAGEConv):
            setattr(pyg_nn, cls.__name__, cls)
        pyg_nn.FastRGCNConv = RGCNConv
        pyg_nn.global_add_pool, pyg_nn.global_mean_pool, pyg_nn.global_max_pool = global_add_pool, global_mean_pool, global_max_pool
        pyg.data, pyg.datasets, pyg.nn = pyg_data, pyg_ds, pyg_nn
        sys.modules["torch_geometric"] = pyg
        sys.modules["torch_geometric.data"] = pyg_data
        sys.modules["torch_geometric.datasets"] = pyg_ds
        sys.modules["torch_geometric.nn"] = pyg_nn
