"""Stand-ins for the third-party modules the reference imports but this image lacks.

TEST TOOLING ONLY.  Used by ``make_golden.py`` (in the authoring container, where
``/root/reference`` is mounted) so the reference's *own* Python can be imported and run
unmodified to produce golden vectors.  Nothing here is imported by the product package
and nothing here is reference source: it is a minimal re-implementation of the public
behaviour of

* ``python-igraph`` 0.9 ``Graph`` (only the dozen methods the reference touches),
* ``dgl`` (``DGLGraph.update_all/apply_edges`` for a message UDF + ``fn.sum`` reducer),
* ``numba.jit`` (identity decorator), ``torch._six.container_abcs``,
* the three ``torch_geometric`` names ``tu_data_processing.py`` imports.

igraph semantics relied on (python-igraph docs): ``add_vertices``/``add_edges`` append in
order; attribute sequences are plain per-element lists, new elements get ``None``;
``incident(v, mode="in")`` returns edge ids; ``delete_vertices`` drops incident edges and
renumbers the survivors compactly, preserving order.
DGL semantics relied on: ``update_all(msg_udf, fn.sum(msg, out), upd_udf)`` calls the UDF
once over all edges (eid order), sums messages by destination, then calls the node UDF
once over all nodes.
"""
import collections.abc
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

    def update_all(self, message_func, reduce_func, apply_node_func=None):
        msgs = message_func(_EdgeBatch(self))
        assert isinstance(reduce_func, _Sum)
        m = msgs[reduce_func.msg]
        agg = th.zeros((self._n,) + tuple(m.shape[1:]), dtype=m.dtype).index_add(0, self._v, m)
        self.ndata[reduce_func.out] = agg
        if apply_node_func is not None:
            self.ndata.update(apply_node_func(_NodeBatch(self)))
        self.ndata.pop(reduce_func.out)


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
        pyg.data, pyg.datasets = pyg_data, pyg_ds
        sys.modules["torch_geometric"] = pyg
        sys.modules["torch_geometric.data"] = pyg_data
        sys.modules["torch_geometric.datasets"] = pyg_ds
