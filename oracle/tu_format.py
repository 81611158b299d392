"""ORACLE (test infrastructure only) -- TU on-disk format either side of the transforms.

CPU restatement, plain Python / numpy, of
  load_graph_labels_from_TUDatadir   graph_classification/data_processing/tu_data_processing.py:117-123
  load_graph_data_from_TUDatadir     :126-219   (parsing, label shift, graph walk, dummy augmentation)
  convert_conjugate_graph_forward    :222-338   (attributes of the conjugate graph)
  save_graph_data / save_graph_labels :341-414  (the files of the DUMMY_/LINE_/CONJ_ datasets)
  the __main__ pipeline              :436-455
  PYGDataset.set_dummy_flags         graph_classification/graph_neural_networks/dataset.py:118-139
and of torch_geometric.io.read_tu_data, the loader the reference calls at dataset.py:150.  PyG is a third-party
dependency that is absent here (README.md:26 pins torch-geometric == 2.0.2); its published algorithm is restated:
read <prefix>_{A,graph_indicator,node_attributes,node_labels,edge_attributes,edge_labels,graph_labels}.txt,
labels -> one-hot of (label - min), x = [attributes | one-hot labels], y = rank of the graph label, remove self
loops, coalesce (sort by (row, col), drop duplicates: attributes of duplicates are ADDED), split by graph_indicator.

Pinned by tests/golden/tu_files.json (files written by the reference itself, tests/golden/make_golden.py); the
read_tu_data restatement has no reference run to pin it ("parity unpinned" for that one function).
Only tests/ may import this module.
"""
import os

import numpy as np

from . import transforms as OT

I64 = np.int64


# ------------------------------------------------------------------------------------ parsing (:117-152)
def load_graph_labels(data_dir):
    out = []
    for fn in sorted(os.listdir(data_dir)):
        if fn.endswith("_graph_labels.txt"):
            with open(os.path.join(data_dir, fn)) as f:
                out.extend(line.strip() for line in f)
    return out


def parse_tu_dir(data_dir):
    if os.path.exists(os.path.join(data_dir, "raw")):
        data_dir = os.path.join(data_dir, "raw")
    raw = dict(A=[], graph_indicator=[], node_labels=[], edge_labels=[], node_attributes=[], edge_attributes=[])
    for fn in sorted(os.listdir(data_dir)):
        path = os.path.join(data_dir, fn)
        if fn.endswith("_A.txt"):
            with open(path) as f:
                raw["A"].extend(tuple(map(int, line.strip().replace(" ", "").split(","))) for line in f)
        else:
            for key, conv in (("graph_indicator", int), ("node_labels", int), ("edge_labels", int),
                              ("node_attributes", float), ("edge_attributes", float)):
                if fn.endswith("_" + key + ".txt"):
                    with open(path) as f:
                        raw[key].extend(conv(line.strip()) for line in f)
    return raw


# ------------------------------------------------------------------------------------ graphs (:154-219)
def load_graph_data(raw, with_dummy=False):
    """Batch arrays (global 0-based ids) of the igraph list the reference builds."""
    b = OT.tu_raw_to_batch(raw["A"], raw["graph_indicator"], raw["node_labels"], raw["edge_labels"])
    G = len(b["node_ptr"]) - 1
    N, E = int(b["node_ptr"][-1]), len(b["src"])
    b["node_attr"] = np.asarray(raw["node_attributes"][:N], dtype=np.float64) if raw["node_attributes"] else None
    b["edge_attr"] = np.asarray(raw["edge_attributes"][:E], dtype=np.float64) if raw["edge_attributes"] else None
    if not with_dummy:
        b["node_id"] = np.concatenate([np.arange(b["node_ptr"][g + 1] - b["node_ptr"][g]) for g in range(G)] or [np.zeros(0, I64)]).astype(I64)
        b["edge_id"] = np.concatenate([np.arange(b["edge_ptr"][g + 1] - b["edge_ptr"][g]) for g in range(G)] or [np.zeros(0, I64)]).astype(I64)
        b["is_dummy_node"] = b["is_dummy_edge"] = None
        return b
    a = OT.dummy_augment_gc(b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], b["node_label"], b["edge_label"])
    # ATTR of the dummy vertex / dummy edges is 0.0 (:190-191, 196-197)
    for key, ptr, flag, old in (("node_attr", "node_ptr", "is_dummy_node", b["node_attr"]),
                                ("edge_attr", "edge_ptr", "is_dummy_edge", b["edge_attr"])):
        if old is None:
            a[key] = None
        else:
            v = np.zeros(len(a[flag]), dtype=np.float64)
            v[a[flag] == 0] = old
            a[key] = v
    return a


def conjugate_graph_data(b, line=False):
    """convert_conjugate_graph_forward over the batch: conj-vertex attributes are the edge attributes of its
    representative edge, conj-edge attributes the vertex attributes of the shared vertex (:235-246, 319-323)."""
    c = OT.conjugate(b["node_ptr"], b["edge_ptr"], b["src"], b["dst"], b["node_label"],
                     is_dummy_edge=None if line else b["is_dummy_edge"], mode="line" if line else "gc")
    rep, sh = c["rep_edge"], c["shared_node"]
    out = dict(node_ptr=c["cnode_ptr"], edge_ptr=c["cedge_ptr"], src=c["csrc"], dst=c["cdst"],
               node_label=b["edge_label"][rep], edge_label=b["node_label"][sh],
               node_id=b["edge_id"][rep], edge_id=b["node_id"][sh],
               node_attr=None if b["edge_attr"] is None else b["edge_attr"][rep],
               edge_attr=None if b["node_attr"] is None else b["node_attr"][sh])
    if b.get("is_dummy_edge") is not None:
        out["is_dummy_node"], out["is_dummy_edge"] = b["is_dummy_edge"][rep], b["is_dummy_node"][sh]
    else:
        out["is_dummy_node"] = out["is_dummy_edge"] = None
    return out


# ------------------------------------------------------------------------------------ writer (:341-414)
def _prefix(data_dir, prefix):
    if prefix == "":
        prefix = os.path.basename(data_dir) + "_"
        if prefix == "raw_":
            prefix = os.path.basename(os.path.dirname(data_dir)) + "_"
    return prefix


def save_graph_labels(graph_labels, data_dir, prefix=""):
    prefix = _prefix(data_dir, prefix)
    with open(os.path.join(data_dir, prefix + "graph_labels.txt"), "w") as f:
        for line in graph_labels:
            f.write(str(line))
            f.write("\n")


def save_graph_data(b, data_dir, prefix=""):
    prefix = _prefix(data_dir, prefix)
    G = len(b["node_ptr"]) - 1

    def put(name, values):
        with open(os.path.join(data_dir, prefix + name + ".txt"), "w") as f:
            for v in values:
                f.write(str(v))
                f.write("\n")

    gi = []
    for g in range(G):
        gi.extend([g + 1] * int(b["node_ptr"][g + 1] - b["node_ptr"][g]))
    put("graph_indicator", gi)
    put("A", ["%d,%d" % (int(u) + 1, int(v) + 1) for u, v in zip(b["src"], b["dst"])])
    put("node_labels", [int(x) for x in b["node_label"]])
    put("edge_labels", [int(x) for x in b["edge_label"]])
    if b.get("node_attr") is not None:
        put("node_attributes", [float(x) for x in b["node_attr"]])
    if b.get("edge_attr") is not None:
        put("edge_attributes", [float(x) for x in b["edge_attr"]])
    put("node_ids", [int(x) for x in b["node_id"]])
    put("edge_ids", [int(x) for x in b["edge_id"]])


def process_dataset(raw_dir, name):
    """The __main__ of tu_data_processing.py without the download: raw_dir = <root>/<name>/raw."""
    labels = load_graph_labels(raw_dir)
    raw = parse_tu_dir(raw_dir)
    plain, dummy = load_graph_data(raw, False), load_graph_data(raw, True)
    for pre, b in (("DUMMY_", dummy), ("LINE_", conjugate_graph_data(plain, line=True)),
                   ("CONJ_", conjugate_graph_data(dummy))):
        d = raw_dir.replace(name, pre + name)
        os.makedirs(d, exist_ok=True)
        save_graph_data(b, d)
        save_graph_labels(labels, d)


# ------------------------------------------------------------------------------------ PyG 2.0.2 read_tu_data
def _read_file(folder, prefix, name, conv):
    path = os.path.join(folder, "%s_%s.txt" % (prefix, name))
    if not os.path.exists(path):
        return None
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([conv(x) for x in line.split(",")])
    return rows


def _one_hot_cols(lab):
    lab = np.asarray(lab, dtype=I64)
    lab = lab - lab.min(axis=0)
    cols = []
    for j in range(lab.shape[1]):
        k = int(lab[:, j].max()) + 1
        oh = np.zeros((lab.shape[0], k), dtype=np.float32)
        oh[np.arange(lab.shape[0]), lab[:, j]] = 1.0
        cols.append(oh)
    return np.concatenate(cols, axis=1)


def read_tu_data(folder, prefix):
    A = np.asarray(_read_file(folder, prefix, "A", int), dtype=I64).reshape(-1, 2)
    edge_index = A.T - 1
    batch = np.asarray(_read_file(folder, prefix, "graph_indicator", int), dtype=I64).reshape(-1) - 1
    na, nl = _read_file(folder, prefix, "node_attributes", float), _read_file(folder, prefix, "node_labels", int)
    ea, el = _read_file(folder, prefix, "edge_attributes", float), _read_file(folder, prefix, "edge_labels", int)
    parts = ([np.asarray(na, dtype=np.float32)] if na is not None else []) + ([_one_hot_cols(nl)] if nl is not None else [])
    x = np.concatenate(parts, axis=1) if parts else None
    parts = ([np.asarray(ea, dtype=np.float32)] if ea is not None else []) + ([_one_hot_cols(el)] if el is not None else [])
    edge_attr = np.concatenate(parts, axis=1) if parts else None
    y = None
    gl = _read_file(folder, prefix, "graph_labels", int)
    if gl is not None:
        gl = np.asarray(gl, dtype=I64).reshape(-1)
        y = np.searchsorted(np.unique(gl), gl).astype(I64)
    num_nodes = int(edge_index.max()) + 1 if x is None else x.shape[0]
    # remove_self_loops
    keep = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, keep]
    if edge_attr is not None:
        edge_attr = edge_attr[keep]
    # coalesce: sort by row * N + col, merge duplicates (attributes added)
    key = edge_index[0] * num_nodes + edge_index[1]
    order = np.argsort(key, kind="stable")
    key, edge_index = key[order], edge_index[:, order]
    first = np.ones(len(key), dtype=bool)
    first[1:] = key[1:] != key[:-1]
    if edge_attr is not None:
        seg = np.cumsum(first) - 1
        merged = np.zeros((int(first.sum()), edge_attr.shape[1]), dtype=np.float32)
        np.add.at(merged, seg, edge_attr[order])
        edge_attr = merged
    edge_index = edge_index[:, first]
    # split
    node_slice = np.concatenate([[0], np.cumsum(np.bincount(batch))]).astype(I64)
    row = edge_index[0]
    edge_slice = np.concatenate([[0], np.cumsum(np.bincount(batch[row], minlength=len(node_slice) - 1))]).astype(I64)
    edge_index = edge_index - node_slice[batch[row]][None, :]
    slices = {"edge_index": edge_slice}
    if x is not None:
        slices["x"] = node_slice
    if edge_attr is not None:
        slices["edge_attr"] = edge_slice
    if y is not None:
        slices["y"] = np.arange(len(y) + 1, dtype=I64)
    return dict(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y), slices


def num_label_columns(x):
    """PYGDataset.num_node_labels (dataset.py:64-72): width of the trailing one-hot block."""
    if x is None:
        return 0
    for i in range(x.shape[1]):
        t = x[:, i:]
        if np.all((t == 0) | (t == 1)) and np.all(t.sum(axis=1) == 1):
            return x.shape[1] - i
    return 0


def num_edge_label_columns(edge_attr):
    """PYGDataset.num_edge_labels (dataset.py:80-87)."""
    if edge_attr is None:
        return 0
    for i in range(edge_attr.shape[1]):
        if edge_attr[:, i:].sum() == edge_attr.shape[0]:
            return edge_attr.shape[1] - i
    return 0


def set_dummy_flags(data, add_dummy):
    """dataset.py:118-139 on the whole collated dataset (local edge_index is only used as an index of is_dummy_node when
    the dataset has no edge_attr; with DUMMY_/CONJ_ files written by save_graph_data it always has)."""
    x, ei, ea = data["x"], data["edge_index"], data["edge_attr"]
    if add_dummy:
        is_dummy_node = x[:, x.shape[1] - num_label_columns(x)].astype(bool)
        is_dummy_edge = ea[:, ea.shape[1] - num_edge_label_columns(ea)].astype(bool)
    else:
        is_dummy_node = np.zeros(x.shape[0], dtype=bool)
        is_dummy_edge = np.zeros(ei.shape[1], dtype=bool)
    return is_dummy_node, is_dummy_edge
