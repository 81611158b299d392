"""Batched-graph containers handed to the hot path (the boundary objects of SURVEY.md 8b).

* ``BatchedGraph`` offers the subset of the DGL graph API that ``GraphAdjModel.forward`` and the
  training loop touch (subgraph_isomorphism/models/basemodel.py:888-930, train.py:753-754,
  dataset.py:1605-1611): ``.to()``, ``.batch_size``, ``.batch_num_nodes()``, ``.ndata/.edata`` with the key
  strings of constants.py:12-35, ``number_of_nodes()``, ``in_degrees()`` ...  DGL itself is not used.
* ``GraphBatch`` is the PyG-style batch the GC models read (graph_neural_networks/main.py:39-41,
  dataset.py:130-137): ``x, edge_index, edge_attr, batch, y, is_dummy_node, is_dummy_edge``.

Both cache the device-side CSR/segment indices (ops.EdgeIndex / ops.RelIndex) built once per batch.
"""
import torch

from . import ops


import os as _os



class _IndexCache:
    def __init__(self):
        self._edge_index = None
        self._rel = []  # [(etype tensor, version, num_rels, RelIndex)]

    def clear(self):
        self._edge_index = None
        self._rel = []


class BatchedGraph:
    def __init__(self, src, dst, num_nodes, batch_num_nodes=None, batch_num_edges=None, ndata=None, edata=None,
                 node_ptr=None, edge_ptr=None):
        """node_ptr / edge_ptr (optional, [G + 1] each): the batch's graph boundaries when the caller already has them (e.g.
        from transforms.dummy_augment_si): kept as they are instead of being rebuilt from the counts (8 small launches)."""
        self._src = torch.as_tensor(src).reshape(-1)
        self._dst = torch.as_tensor(dst).reshape(-1)
        self._n = int(num_nodes)
        dev = self._src.device
        self._bnn = (torch.as_tensor([self._n]) if batch_num_nodes is None else torch.as_tensor(batch_num_nodes)).to(dev).long()
        self._bne = (torch.as_tensor([self._src.numel()]) if batch_num_edges is None
                     else torch.as_tensor(batch_num_edges)).to(dev).long()
        self.ndata = dict(ndata or {})
        self.edata = dict(edata or {})
        self._cache = _IndexCache()
        self._node_ptr = None if node_ptr is None else torch.as_tensor(node_ptr).to(dev).to(torch.int32).contiguous()
        self._edge_ptr = None if edge_ptr is None else torch.as_tensor(edge_ptr).to(dev).to(torch.int32).contiguous()

    # ---- DGL-like surface ---------------------------------------------------------------------
    @property
    def device(self):
        return self._src.device

    @property
    def batch_size(self):
        return int(self._bnn.numel())

    def batch_num_nodes(self):
        return self._bnn

    def batch_num_edges(self):
        return self._bne

    def number_of_nodes(self):
        return self._n

    num_nodes = number_of_nodes

    def number_of_edges(self):
        return int(self._src.numel())

    num_edges = number_of_edges

    def all_edges(self, form="uv", order="eid"):
        if order != "eid":
            raise NotImplementedError("only eid order is kept")
        if form == "uv":
            return self._src, self._dst
        return self._src, self._dst, torch.arange(self._src.numel(), device=self.device)

    edges = all_edges

    def in_degrees(self):
        if self._src.is_cuda:
            return ops.degrees(self._src.to(torch.int32), self._dst.to(torch.int32), self._n)[0].long()
        return torch.bincount(self._dst.long(), minlength=self._n)

    def out_degrees(self):
        if self._src.is_cuda:
            return ops.degrees(self._src.to(torch.int32), self._dst.to(torch.int32), self._n)[1].long()
        return torch.bincount(self._src.long(), minlength=self._n)

    def to(self, device, non_blocking=False):
        g = BatchedGraph(self._src.to(device, non_blocking=non_blocking), self._dst.to(device, non_blocking=non_blocking),
                         self._n, self._bnn.to(device), self._bne.to(device),
                         {k: v.to(device, non_blocking=non_blocking) for k, v in self.ndata.items()},
                         {k: v.to(device, non_blocking=non_blocking) for k, v in self.edata.items()},
                         node_ptr=self._node_ptr, edge_ptr=self._edge_ptr)
        return g

    def local_var(self):
        return self

    # ---- index structures consumed by the kernels -----------------------------------------------
    def node_ptr(self):
        """Graph boundaries of the batch (a property of the batch, not of an index: computed once and kept)."""
        if self._node_ptr is None:
            z = torch.zeros(1, dtype=torch.long, device=self._bnn.device)
            self._node_ptr = torch.cat([z, torch.cumsum(self._bnn, 0)]).to(torch.int32)
        return self._node_ptr

    def edge_ptr(self):
        if self._edge_ptr is None:
            z = torch.zeros(1, dtype=torch.long, device=self._bne.device)
            self._edge_ptr = torch.cat([z, torch.cumsum(self._bne, 0)]).to(torch.int32)
        return self._edge_ptr

    def edge_index(self):
        if self._cache._edge_index is None:
            self._cache._edge_index = ops.EdgeIndex(self._src, self._dst, self._n, node_ptr=self.node_ptr())
        return self._cache._edge_index

    def rel_index(self, etype, num_rels):
        for t, ver, r, ix in self._cache._rel:
            if t is etype and ver == etype._version and r == num_rels:
                return ix
        ix = ops.RelIndex(self._src, self._dst, etype, self._n, num_rels)
        self._cache._rel.append((etype, etype._version, num_rels, ix))
        if len(self._cache._rel) > 4:
            self._cache._rel.pop(0)
        return ix

    def row_index(self, etype, num_rels, self_loop, closing_hint=None):
        """closing_hint = (H, dtype) of the rows the caller is about to pass: a fresh batch's index is then built with everything
        its first step needs in one call (ops.RowIndex); a cached index is returned as it is."""
        for t, ver, r, ix in self._cache._rel:
            if t is etype and ver == etype._version and r == ("row", num_rels, self_loop):
                return ix
        ix = ops.RowIndexSet(self._src, self._dst, etype, self._n, num_rels, self_loop,
                             node_ptr=self.node_ptr(), edge_ptr=self.edge_ptr(), closing_hint=closing_hint)
        self._cache._rel.append((etype, etype._version, ("row", num_rels, self_loop), ix))
        if len(self._cache._rel) > 4:
            self._cache._rel.pop(0)
        return ix

    # ---- dgl.batch (dataset.py:1321-1328, 1609-1610) ----------------------------------------------
    @staticmethod
    def batch(graphs):
        """Disjoint union: concat features, offset node ids, keep per-graph sizes."""
        srcs, dsts, bnn, bne = [], [], [], []
        off = 0
        for g in graphs:
            srcs.append(g._src.long() + off)
            dsts.append(g._dst.long() + off)
            bnn.append(g._bnn)
            bne.append(g._bne)
            off += g._n
        nd = {k: torch.cat([g.ndata[k] for g in graphs], 0) for k in graphs[0].ndata} if graphs else {}
        ed = {k: torch.cat([g.edata[k] for g in graphs], 0) for k in graphs[0].edata} if graphs else {}
        return BatchedGraph(torch.cat(srcs), torch.cat(dsts), off, torch.cat(bnn), torch.cat(bne), nd, ed)


class GraphBatch:
    """PyG-style mini-batch (torch_geometric.data.Batch look-alike) for the GC models."""

    def __init__(self, x, edge_index, batch=None, edge_attr=None, y=None, is_dummy_node=None, is_dummy_edge=None,
                 ptr=None, num_graphs=None):
        self.x, self.edge_index, self.edge_attr, self.y = x, edge_index, edge_attr, y
        self.is_dummy_node, self.is_dummy_edge = is_dummy_node, is_dummy_edge
        if batch is None:
            batch = torch.zeros(x.shape[0], dtype=torch.long, device=x.device)
        self.batch = batch
        if ptr is None:
            ng = int(num_graphs) if num_graphs is not None else (int(batch.max().item()) + 1 if batch.numel() else 0)
            cnt = torch.bincount(batch, minlength=ng)
            ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=cnt.device), torch.cumsum(cnt, 0)])
        self.ptr = ptr
        self._cache = _IndexCache()

    @property
    def num_graphs(self):
        return int(self.ptr.numel() - 1)

    @property
    def num_nodes(self):
        return int(self.x.shape[0])

    def to(self, device, non_blocking=False):
        mv = lambda t: None if t is None else t.to(device, non_blocking=non_blocking)  # noqa: E731
        return GraphBatch(mv(self.x), mv(self.edge_index), mv(self.batch), mv(self.edge_attr), mv(self.y),
                          mv(self.is_dummy_node), mv(self.is_dummy_edge), mv(self.ptr))

    @staticmethod
    def collate(items):
        """PyG DataLoader collate (main.py:245-247): concat x, offset edge_index, build batch/ptr."""
        xs, eis, eas, ys, dn, de, bs = [], [], [], [], [], [], []
        off = 0
        ptr = [0]
        for i, d in enumerate(items):
            n = d.x.shape[0]
            xs.append(d.x)
            eis.append(d.edge_index + off)
            if d.edge_attr is not None:
                eas.append(d.edge_attr)
            if d.y is not None:
                ys.append(d.y.reshape(-1))
            if d.is_dummy_node is not None:
                dn.append(d.is_dummy_node)
            if d.is_dummy_edge is not None:
                de.append(d.is_dummy_edge)
            bs.append(torch.full((n,), i, dtype=torch.long, device=d.x.device))
            off += n
            ptr.append(off)
        cat = lambda l, dim=0: torch.cat(l, dim) if l else None  # noqa: E731
        dev = xs[0].device if xs else None
        return GraphBatch(cat(xs), cat(eis, 1), cat(bs), cat(eas), cat(ys), cat(dn), cat(de),
                          torch.tensor(ptr, dtype=torch.long, device=dev))


def graph_ptr_i32(data):
    """int32 node ranges per graph of a PyG-style batch (uses .ptr when present, else derives it from .batch)."""
    p = getattr(data, "ptr", None)
    hit = getattr(data, "_dn_ptr_i32", None)                  # (kept on the batch object: one conversion per batch, not per readout)
    if hit is not None and p is not None and hit[0] is p and hit[1] == p._version and hit[2].device == data.x.device:
        return hit[2]
    if p is None:
        b = data.batch
        ng = int(b.max().item()) + 1 if b.numel() else 0
        cnt = torch.bincount(b, minlength=ng)
        p = torch.cat([torch.zeros(1, dtype=torch.long, device=cnt.device), torch.cumsum(cnt, 0)])
    out = p.to(device=data.x.device, dtype=torch.int32)
    src = getattr(data, "ptr", None)
    if src is not None:
        try:
            data._dn_ptr_i32 = (src, src._version, out)
        except Exception:                                     # (a batch object without settable attributes: just no cache)
            pass
    return out


def _node_ptr_or_none(data):
    """Graph boundaries of a PyG-style batch (graph-local index builder, matrix-core neighbour sum), or None when the batch carries neither .ptr nor .batch."""
    if getattr(data, "ptr", None) is None and getattr(data, "batch", None) is None:
        return None
    return graph_ptr_i32(data)


def _edge_ptr_or_none(data, node_ptr):
    """Edge ranges per graph of a PyG-style batch whose edge list is the concatenation of the graphs' edge lists (what the
    collate of main.py:245-247 produces): graph of an edge = graph of its source, edge_ptr[g] = first edge of graph g.  No check
    here: the graph-local index builder validates every edge against its graph's node range on the device and declines (the
    general builder then runs) when the edges are not grouped by graph."""
    if node_ptr is None or data.edge_index.shape[1] == 0 or node_ptr.numel() < 2:
        return None
    src = data.edge_index[0].contiguous()
    bounds = node_ptr.to(src.dtype)
    gid = torch.bucketize(src, bounds[1:].contiguous(), right=True)
    return torch.searchsorted(gid, torch.arange(bounds.numel(), dtype=gid.dtype, device=gid.device)).to(torch.int32)


def edge_index_of(data):
    """Cached ops.EdgeIndex of a PyG-style batch (edge_index row 0 = src, row 1 = dst)."""
    cache = getattr(data, "_cache", None)
    if cache is None:
        cache = _IndexCache()
        try:
            data._cache = cache
        except Exception:
            pass
    if cache._edge_index is None or cache._edge_index.num_edges != data.edge_index.shape[1]:
        cache._edge_index = ops.EdgeIndex(data.edge_index[0], data.edge_index[1], data.x.shape[0], node_ptr=_node_ptr_or_none(data))
    return cache._edge_index


def row_index_of(data, etype, num_rels, self_loop, closing_hint=None):
    """Cached ops.RowIndexSet (bf16 matrix-core path) of a PyG-style batch.  closing_hint as BatchedGraph.row_index."""
    cache = getattr(data, "_cache", None)
    if cache is None:
        cache = _IndexCache()
        try:
            data._cache = cache
        except Exception:
            pass
    tag = ("row", num_rels, self_loop)
    for t, ver, r, ix in cache._rel:
        if (t is etype or (t.data_ptr() == etype.data_ptr() and t.numel() == etype.numel())) and ver == etype._version \
                and r == tag:
            return ix
    nptr = _node_ptr_or_none(data)
    ix = ops.RowIndexSet(data.edge_index[0], data.edge_index[1], etype, data.x.shape[0], num_rels, self_loop,
                         node_ptr=nptr, edge_ptr=_edge_ptr_or_none(data, nptr), closing_hint=closing_hint)
    cache._rel.append((etype, etype._version, tag, ix))
    if len(cache._rel) > 4:
        cache._rel.pop(0)
    return ix


def gcn_edge_index_of(data):
    """ops.EdgeIndex of the batch with GCN self loops (existing self loops dropped, one per node appended) + the boolean
    mask of kept edges.  Cached on the batch object."""
    cache = getattr(data, "_cache", None)
    if cache is None:
        cache = _IndexCache()
        try:
            data._cache = cache
        except Exception:
            pass
    hit = getattr(cache, "_gcn", None)
    if hit is None or hit[2] != data.edge_index.shape[1]:
        src, dst = data.edge_index[0], data.edge_index[1]
        keep = src != dst
        ar = torch.arange(data.x.shape[0], device=src.device, dtype=src.dtype)
        ix = ops.EdgeIndex(torch.cat([src[keep], ar]), torch.cat([dst[keep], ar]), data.x.shape[0], node_ptr=_node_ptr_or_none(data))
        # (once per batch, so that the conv's forward has no boolean indexing -- a nonzero + host sync per call, and a sort-based
        #  index_put in its backward -- and stays capturable: positions of the kept / dropped edges, the dropped edges' nodes)
        all_kept = bool(keep.all())
        ix.gcn_keep_idx = None if all_kept else keep.nonzero().view(-1)
        ix.gcn_drop_idx = None if all_kept else (~keep).nonzero().view(-1)
        ix.gcn_drop_src = None if all_kept else src[~keep]
        hit = (ix, keep, data.edge_index.shape[1])
        cache._gcn = hit
    return hit[0], hit[1]


def rel_index_of(data, etype, num_rels):
    cache = getattr(data, "_cache", None)
    if cache is None:
        cache = _IndexCache()
        try:
            data._cache = cache
        except Exception:
            pass
    for t, ver, r, ix in cache._rel:
        if (t is etype or (t.data_ptr() == etype.data_ptr() and t.numel() == etype.numel())) and ver == etype._version \
                and r == num_rels:
            return ix
    ix = ops.RelIndex(data.edge_index[0], data.edge_index[1], etype, data.x.shape[0], num_rels)
    cache._rel.append((etype, etype._version, num_rels, ix))
    if len(cache._rel) > 4:
        cache._rel.pop(0)
    return ix
