"""TU on-disk datasets either side of the device transforms.

Mirror of graph_classification/data_processing/tu_data_processing.py (the offline DUMMY_/LINE_/CONJ_ dataset builder)
and of the loader side graph_classification/graph_neural_networks/dataset.py, with the per-graph igraph loops replaced
by ONE batch of device tensors that goes through the HIP index builds (transforms.dummy_augment_gc / conjugate):

  load_graph_labels_from_TUDatadir   tu_data_processing.py:117-123
  load_graph_data_from_TUDatadir     :126-219   text -> batch (label shift :154-169, graph walk :173-181) [-> dummy]
  convert_conjugate_graph_forward    :222-338   whole batch at once; attributes gathered through rep_edge/shared_node
  save_graph_data / save_graph_labels :341-414  byte-identical files (tests/golden/tu_files.json)
  process_dataset                    :436-455   the __main__ pipeline without the download
  read_tu_data                       torch_geometric.io.read_tu_data (PyG 2.0.2, README.md:26) as called at dataset.py:150
  PYGDataset                         dataset.py:10-169 (raw_dir naming, label/attribute widths, set_dummy_flags)

Text parsing and formatting run on the host (numpy); everything between is int32 / float tensors on the GPU.
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import transforms
from .graph import GraphBatch

_KEYS = ("node_ptr", "edge_ptr", "src", "dst", "node_label", "edge_label", "node_id", "edge_id", "node_attr", "edge_attr",
         "is_dummy_node", "is_dummy_edge")


class TUBatch(dict):
    """All graphs of a TU dataset as one batch: node_ptr/edge_ptr [G+1], src/dst [E] global 0-based ids, labels, ids
    (the reference's vs["ID"] / es["ID"]), optional float64 attributes and dummy flags -- tensors on one device."""

    @property
    def num_graphs(self):
        return int(self["node_ptr"].numel() - 1)

    def cpu(self):
        return TUBatch({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self.items()})


# ------------------------------------------------------------------------------------------------ parsing
def _resolve(data_dir):
    raw = os.path.join(data_dir, "raw")
    return raw if os.path.exists(raw) else data_dir


def _column(path, dtype):
    with open(path) as f:
        txt = f.read()
    return np.array(txt.replace(",", " ").split(), dtype=dtype)


def _single_column(path, dtype):
    """One value per line, as the reference parses labels and attributes (int(line.strip()) / float(line.strip()),
    tu_data_processing.py:140-152): a line with several columns (ENZYMES-style multi-dimensional attributes) or a blank line
    raises ValueError there, and does here -- it must not be flattened into a longer 1-D array."""
    conv = int if np.issubdtype(dtype, np.integer) else float
    out = []
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            try:
                out.append(conv(line.strip()))
            except ValueError:
                raise ValueError("%s:%d: %r is not a single %s (the reference pipeline reads one value per line; "
                                 "multi-column attribute files are not supported)" % (path, ln, line.strip(), conv.__name__))
    return np.array(out, dtype=dtype)


def load_graph_labels_from_TUDatadir(data_dir):
    out = []
    for fn in sorted(os.listdir(data_dir)):
        if fn.endswith("_graph_labels.txt"):
            with open(os.path.join(data_dir, fn)) as f:
                out.extend(line.strip() for line in f)
    return out


def read_raw(data_dir):
    """The six raw arrays of :126-152 (files matched by suffix, several files of one kind concatenated in name order)."""
    data_dir = _resolve(data_dir)
    kinds = {"A": np.int64, "graph_indicator": np.int64, "node_labels": np.int64, "edge_labels": np.int64,
             "node_attributes": np.float64, "edge_attributes": np.float64}
    parts = {k: [] for k in kinds}
    for fn in sorted(os.listdir(data_dir)):
        for k, dt in kinds.items():
            if fn.endswith("_" + k + ".txt"):
                path = os.path.join(data_dir, fn)
                parts[k].append(_column(path, dt) if k == "A" else _single_column(path, dt))
    raw = {k: (np.concatenate(v) if v else np.zeros(0, dtype=kinds[k])) for k, v in parts.items()}
    raw["A"] = raw["A"].reshape(-1, 2)
    return raw


def _shift_labels(lab, count, dev):
    """:154-169: no labels -> all 1; otherwise shifted so that the minimum is 1."""
    if lab.size == 0:
        return torch.ones(count, dtype=torch.int32, device=dev)
    t = torch.from_numpy(lab).to(dev)
    return (t - t.min() + 1).to(torch.int32)


def _local_ids(ptr):
    n = int(ptr[-1])
    g = torch.repeat_interleave(torch.arange(ptr.numel() - 1, device=ptr.device), (ptr[1:] - ptr[:-1]).long())
    return (torch.arange(n, device=ptr.device) - ptr.long()[g]).to(torch.int32)


def load_graph_data_from_TUDatadir(data_dir, with_dummy=False, device="cuda"):
    raw = read_raw(data_dir)
    dev = torch.device(device)
    A = torch.from_numpy(raw["A"]).to(dev)
    gi = torch.from_numpy(raw["graph_indicator"]).to(dev)
    node_label = _shift_labels(raw["node_labels"], gi.numel(), dev)
    edge_label = _shift_labels(raw["edge_labels"], A.shape[0], dev)
    # graph walk (:173-181, 215-217): graphs gi[0], gi[0]+1, ... up to the graph of the LAST edge (later, edge-less graphs
    # are dropped by the reference); the nodes of graph gd are the next Counter(graph_indicator)[gd] rows
    g0 = int(gi[0])
    ge = gi[A[:, 0] - 1] - g0                                    # graph of every edge (both endpoints share it)
    G = int(ge[-1]) + 1 if ge.numel() else 0
    counts = torch.bincount(gi - g0, minlength=G)[:G] if gi.numel() else torch.zeros(G, dtype=torch.long, device=dev)
    zero = torch.zeros(1, dtype=torch.long, device=dev)
    node_ptr = torch.cat([zero, torch.cumsum(counts, 0)])
    edge_ptr = torch.cat([zero, torch.cumsum(torch.bincount(ge, minlength=G), 0)])
    N = int(node_ptr[-1])
    b = TUBatch(node_ptr=node_ptr.to(torch.int32), edge_ptr=edge_ptr.to(torch.int32),
                src=(A[:, 0] - 1).to(torch.int32), dst=(A[:, 1] - 1).to(torch.int32),
                node_label=node_label[:N].contiguous(), edge_label=edge_label,
                node_attr=torch.from_numpy(raw["node_attributes"][:N]).to(dev) if raw["node_attributes"].size else None,
                edge_attr=torch.from_numpy(raw["edge_attributes"]).to(dev) if raw["edge_attributes"].size else None,
                is_dummy_node=None, is_dummy_edge=None)
    if not with_dummy:
        b["node_id"], b["edge_id"] = _local_ids(b["node_ptr"]), _local_ids(b["edge_ptr"])
        return b
    a = transforms.dummy_augment_gc(b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], b["node_label"], b["edge_label"])
    out = TUBatch({k: a[k] for k in a})
    for key, flag in (("node_attr", "is_dummy_node"), ("edge_attr", "is_dummy_edge")):      # dummy ATTR = 0.0 (:190-197)
        if b[key] is None:
            out[key] = None
        else:
            v = torch.zeros(a[flag].numel(), dtype=torch.float64, device=dev)
            v[a[flag] == 0] = b[key]
            out[key] = v
    return out


def convert_conjugate_graph_forward(b):
    """L_Phi of every graph of the batch (dummy edges merged when the batch carries IS_DUMMY, plain line graph otherwise)."""
    line = b.get("is_dummy_edge") is None
    c = transforms.conjugate(b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], b["node_label"],
                             is_dummy_edge=None if line else b["is_dummy_edge"], mode="line" if line else "gc")
    rep, sh = c["rep_edge"].long(), c["shared_node"].long()
    pick = lambda t, i: None if t is None else t[i]  # noqa: E731
    return TUBatch(node_ptr=c["cnode_ptr"], edge_ptr=c["cedge_ptr"], src=c["csrc"], dst=c["cdst"],
                   node_label=b["edge_label"][rep], edge_label=b["node_label"][sh],
                   node_id=b["edge_id"][rep], edge_id=b["node_id"][sh],
                   node_attr=pick(b["edge_attr"], rep), edge_attr=pick(b["node_attr"], sh),
                   is_dummy_node=pick(b["is_dummy_edge"], rep), is_dummy_edge=pick(b["is_dummy_node"], sh))


# ------------------------------------------------------------------------------------------------ writer
def _prefix(data_dir, prefix):
    if prefix == "":
        prefix = os.path.basename(data_dir) + "_"
        if prefix == "raw_":
            prefix = os.path.basename(os.path.dirname(data_dir)) + "_"
    return prefix


def _put(path, values):
    with open(path, "w") as f:
        if len(values):
            f.write("\n".join(map(str, values)))
            f.write("\n")


def save_graph_labels(graph_labels, data_dir, prefix=""):
    _put(os.path.join(data_dir, _prefix(data_dir, prefix) + "graph_labels.txt"), list(graph_labels))


def save_graph_data(b, data_dir, prefix=""):
    prefix = _prefix(data_dir, prefix)
    h = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in b.items()}
    sizes = np.diff(h["node_ptr"].astype(np.int64))
    path = lambda name: os.path.join(data_dir, prefix + name + ".txt")  # noqa: E731
    _put(path("graph_indicator"), np.repeat(np.arange(1, sizes.size + 1), sizes).tolist())
    _put(path("A"), ["%d,%d" % (u + 1, v + 1) for u, v in zip(h["src"].tolist(), h["dst"].tolist())])
    _put(path("node_labels"), h["node_label"].tolist())
    _put(path("edge_labels"), h["edge_label"].tolist())
    if h.get("node_attr") is not None:
        _put(path("node_attributes"), h["node_attr"].astype(np.float64).tolist())
    if h.get("edge_attr") is not None:
        _put(path("edge_attributes"), h["edge_attr"].astype(np.float64).tolist())
    _put(path("node_ids"), h["node_id"].tolist())
    _put(path("edge_ids"), h["edge_id"].tolist())


def process_dataset(raw_dir, dataset, device="cuda"):
    """tu_data_processing.py:436-455 for an already downloaded <root>/<dataset>/raw: writes <root>/{DUMMY_,LINE_,CONJ_}<dataset>/raw."""
    labels = load_graph_labels_from_TUDatadir(raw_dir)
    plain = load_graph_data_from_TUDatadir(raw_dir, with_dummy=False, device=device)
    dummy = load_graph_data_from_TUDatadir(raw_dir, with_dummy=True, device=device)
    out = {}
    for pre, b in (("DUMMY_", dummy), ("LINE_", convert_conjugate_graph_forward(plain)),
                   ("CONJ_", convert_conjugate_graph_forward(dummy))):
        d = raw_dir.replace(dataset, pre + dataset)
        os.makedirs(d, exist_ok=True)
        save_graph_data(b, d)
        save_graph_labels(labels, d)
        out[pre] = d
    return out


# ------------------------------------------------------------------------------------------------ loader side
def _table(folder, prefix, name, dtype):
    path = os.path.join(folder, "%s_%s.txt" % (prefix, name))
    if not os.path.exists(path):
        return None
    with open(path) as f:
        first = f.readline()
    width = first.count(",") + 1
    return _column(path, dtype).reshape(-1, width)


def _one_hot(lab, dev):
    t = torch.from_numpy(lab).to(dev)
    t = t - t.min(dim=0)[0]
    return torch.cat([torch.nn.functional.one_hot(c) for c in t.unbind(-1)], dim=-1).to(torch.float32)


def read_tu_data(folder, prefix, device="cuda"):
    """(data, slices) as torch_geometric.io.read_tu_data returns them: x = [attributes | one-hot(labels - min)], self loops
    removed, edges coalesced (sorted by (row, col), attributes of duplicates added), per-graph local edge_index."""
    dev = torch.device(device)
    edge_index = torch.from_numpy(_table(folder, prefix, "A", np.int64)).to(dev).t().contiguous() - 1
    batch = torch.from_numpy(_table(folder, prefix, "graph_indicator", np.int64)).to(dev).view(-1) - 1
    cat = lambda parts: torch.cat(parts, dim=-1) if parts else None  # noqa: E731
    na, nl = _table(folder, prefix, "node_attributes", np.float32), _table(folder, prefix, "node_labels", np.int64)
    ea, el = _table(folder, prefix, "edge_attributes", np.float32), _table(folder, prefix, "edge_labels", np.int64)
    x = cat(([torch.from_numpy(na).to(dev)] if na is not None else []) + ([_one_hot(nl, dev)] if nl is not None else []))
    edge_attr = cat(([torch.from_numpy(ea).to(dev)] if ea is not None else []) + ([_one_hot(el, dev)] if el is not None else []))
    y = None
    gl = _table(folder, prefix, "graph_labels", np.int64)
    if gl is not None:
        _, y = torch.from_numpy(gl).to(dev).view(-1).unique(sorted=True, return_inverse=True)
    num_nodes = int(edge_index.max()) + 1 if x is None else x.shape[0]
    keep = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, keep]
    if edge_attr is not None:
        edge_attr = edge_attr[keep]
    key, order = torch.sort(edge_index[0] * num_nodes + edge_index[1], stable=True)
    first = torch.ones_like(key, dtype=torch.bool)
    first[1:] = key[1:] != key[:-1]
    if edge_attr is not None:
        seg = torch.cumsum(first.long(), 0) - 1
        edge_attr = torch.zeros((int(first.sum()), edge_attr.shape[1]), dtype=edge_attr.dtype, device=dev).index_add_(
            0, seg, edge_attr[order])
    edge_index = edge_index[:, order][:, first]
    zero = torch.zeros(1, dtype=torch.long, device=dev)
    node_slice = torch.cat([zero, torch.cumsum(torch.bincount(batch), 0)])
    row = edge_index[0]
    edge_slice = torch.cat([zero, torch.cumsum(torch.bincount(batch[row], minlength=node_slice.numel() - 1), 0)])
    edge_index = edge_index - node_slice[batch[row]].unsqueeze(0)
    slices = {"edge_index": edge_slice}
    if x is not None:
        slices["x"] = node_slice
    if edge_attr is not None:
        slices["edge_attr"] = edge_slice
    if y is not None:
        slices["y"] = torch.arange(y.numel() + 1, device=dev)
    return SimpleNamespace(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y), slices


class PYGDataset:
    """dataset.py:10-169 without PyG: reads <root>/<{,DUMMY_,LINE_,CONJ_}name>/raw (written by process_dataset or by the
    reference's tu_data_processing.py), keeps the collated tensors on `device`, hands out per-graph items for
    GraphBatch.collate.  No download (the reference raises for the derived datasets too, dataset.py:113-116)."""

    def __init__(self, root, name, use_node_attr=False, use_edge_attr=False, cleaned=False, add_dummy=False,
                 convert_conjugate=False, device="cuda"):
        self.root, self.name, self.cleaned = root, name, cleaned
        self.add_dummy, self.convert_conjugate = add_dummy, convert_conjugate
        self.data, self.slices = read_tu_data(self.raw_dir, self.data_name, device=device)
        self.data.is_dummy_node, self.data.is_dummy_edge = self.set_dummy_flags(self.data)
        if self.data.x is not None and not use_node_attr:
            self.data.x = self.data.x[:, self.num_node_attributes:]
        if self.data.edge_attr is not None and not use_edge_attr:
            self.data.edge_attr = self.data.edge_attr[:, self.num_edge_attributes:]

    @property
    def data_name(self):
        if self.add_dummy and self.convert_conjugate:
            return "CONJ_" + self.name
        if self.add_dummy:
            return "DUMMY_" + self.name
        if self.convert_conjugate:
            return "LINE_" + self.name
        return self.name

    @property
    def raw_dir(self):
        return os.path.join(self.root, self.data_name, "raw_cleaned" if self.cleaned else "raw")

    @property
    def num_node_labels(self):
        x = self.data.x
        if x is None:
            return 0
        for i in range(x.shape[1]):
            t = x[:, i:]
            if bool(((t == 0) | (t == 1)).all()) and bool((t.sum(dim=1) == 1).all()):
                return x.shape[1] - i
        return 0

    @property
    def num_node_attributes(self):
        return 0 if self.data.x is None else self.data.x.shape[1] - self.num_node_labels

    @property
    def num_edge_labels(self):
        ea = self.data.edge_attr
        if ea is None:
            return 0
        for i in range(ea.shape[1]):
            if float(ea[:, i:].sum()) == ea.shape[0]:
                return ea.shape[1] - i
        return 0

    @property
    def num_edge_attributes(self):
        return 0 if self.data.edge_attr is None else self.data.edge_attr.shape[1] - self.num_edge_labels

    @property
    def num_features(self):
        return 0 if self.data.x is None else self.data.x.shape[1]

    @property
    def num_classes(self):
        return int(self.data.y.max()) + 1

    def set_dummy_flags(self, data):
        """dataset.py:118-139 on the collated dataset: the first label column is the one-hot of label 0 = dummy."""
        if self.add_dummy:
            is_dummy_node = data.x[:, self.num_node_attributes].bool()
            if data.edge_attr is not None:
                is_dummy_edge = data.edge_attr[:, self.num_edge_attributes].bool()
            else:
                off = self.slices["x"][torch.bucketize(torch.arange(data.edge_index.shape[1], device=data.x.device),
                                                       self.slices["edge_index"], right=True) - 1]
                is_dummy_edge = is_dummy_node[data.edge_index[0] + off] | is_dummy_node[data.edge_index[1] + off]
        else:
            is_dummy_node = torch.zeros(data.x.shape[0], dtype=torch.bool, device=data.x.device)
            is_dummy_edge = torch.zeros(data.edge_index.shape[1], dtype=torch.bool, device=data.x.device)
        return is_dummy_node, is_dummy_edge

    def __len__(self):
        return int(self.slices["edge_index"].numel() - 1)

    def __getitem__(self, i):
        n0, n1 = int(self.slices["x"][i]), int(self.slices["x"][i + 1])
        e0, e1 = int(self.slices["edge_index"][i]), int(self.slices["edge_index"][i + 1])
        d = self.data
        return SimpleNamespace(x=d.x[n0:n1], edge_index=d.edge_index[:, e0:e1],
                               edge_attr=None if d.edge_attr is None else d.edge_attr[e0:e1], y=d.y[i:i + 1],
                               is_dummy_node=d.is_dummy_node[n0:n1], is_dummy_edge=d.is_dummy_edge[e0:e1])

    def batch(self, indices):
        """GraphBatch of the given graphs (what DataLoader(dataset, batch_size) yields, main.py:245-247)."""
        return GraphBatch.collate([self[int(i)] for i in indices])

    def __repr__(self):
        return "%s(%d)" % (self.name, len(self))
