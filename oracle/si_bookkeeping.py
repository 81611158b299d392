"""ORACLE (test infrastructure only) -- integer bookkeeping either side of L_Phi in subgraph isomorphism.

Plain-numpy restatement, pinned by tests/golden/si_bookkeeping.json (outputs of the reference's own functions):
  conjugate_subisomorphisms   subgraph_isomorphism/utils/graph.py:291-330  get_conjugate_subisomorphisms
  edgeseq_subisoweights       subgraph_isomorphism/dataset.py:63-108       compute_edgeseq_subisoweights
  nodeseq_subisoweights       dataset.py:54-60                             compute_nodeseq_subisoweights
  compute_norm                utils/graph.py:11-38   (DGL branch)
  largest_eigenvalues         utils/graph.py:41-71   (DGL branch)
  add_reversed_edges          train.py:323-345       (GraphAdj branch, one graph)
Only tests/ may import this module.

Shared structure of the two edge-level functions: the pattern's edges (in eid order) are cut into RUNS of consecutive
equal (u, v); a dict keyed by (u, v) keeps the labels of the LAST run of every key, in first-insertion order
(conjugate) or sorted-key order (weights).  For every subisomorphism and key the graph edges (g_u, g_v sorted) equal to
(sub[u], sub[v]) are scanned in order, and every (graph edge k, run label e) pair with e == g_el[k] counts.
"""
import numpy as np

I64 = np.int64


def _pattern_runs(p_u, p_v, p_el):
    """-> list of (u, v, labels) in first-insertion order, labels = those of the last run with that key."""
    order, labels = [], {}
    i, n = 0, len(p_el)
    while i < n:
        j = i + 1
        while j < n and p_u[j] == p_u[i] and p_v[j] == p_v[i]:
            j += 1
        key = (int(p_u[i]), int(p_v[i]))
        if key not in labels:
            order.append(key)
        labels[key] = [int(x) for x in p_el[i:j]]
        i = j
    return [(u, v, labels[(u, v)]) for (u, v) in order]


def _matches(g_u, g_v, u, v):
    return np.nonzero((g_u == u) & (g_v == v))[0]        # ascending k; g is (src, dst)-sorted so this is one range


def conjugate_subisomorphisms(p_u, p_v, p_el, g_u, g_v, g_el, subisomorphisms):
    """[S, len(p_el)] int64: column c (c-th distinct key) = the LAST graph edge k between the mapped endpoints whose label
    occurs in the key's run; 0 where nothing matches and in the columns past the number of distinct keys."""
    p_u, p_v, p_el, g_u, g_v, g_el = (np.asarray(a, dtype=I64) for a in (p_u, p_v, p_el, g_u, g_v, g_el))
    sub = np.asarray(subisomorphisms, dtype=I64).reshape(-1, sub_width(subisomorphisms))
    runs = _pattern_runs(p_u, p_v, p_el)
    out = np.zeros((len(sub), len(p_el)), dtype=I64)
    for i, s in enumerate(sub):
        for c, (u, v, labels) in enumerate(runs):
            for k in _matches(g_u, g_v, s[u], s[v]):
                if int(g_el[k]) in labels:
                    out[i, c] = k
    return out


def sub_width(subisomorphisms):
    a = np.asarray(subisomorphisms)
    return a.shape[1] if a.ndim == 2 else 0


def edgeseq_subisoweights(p_u, p_v, p_el, g_u, g_v, g_el, subisomorphisms):
    """[len(g_el)] int64: how many (subisomorphism, key, run label) triples hit each graph edge."""
    p_u, p_v, p_el, g_u, g_v, g_el = (np.asarray(a, dtype=I64) for a in (p_u, p_v, p_el, g_u, g_v, g_el))
    sub = np.asarray(subisomorphisms, dtype=I64).reshape(-1, sub_width(subisomorphisms))
    runs = _pattern_runs(p_u, p_v, p_el)
    w = np.zeros(len(g_el), dtype=I64)
    for s in sub:
        for (u, v, labels) in runs:
            for k in _matches(g_u, g_v, s[u], s[v]):
                w[k] += sum(1 for e in labels if e == int(g_el[k]))
    return w


def nodeseq_subisoweights(num_nodes, subisomorphisms):
    return np.bincount(np.asarray(subisomorphisms, dtype=I64).reshape(-1), minlength=num_nodes).astype(I64)


def compute_norm(src, dst, num_nodes, self_loop):
    """node_norm [N] float32 = 1/(in_deg+1) or 1/in_deg with 1.0 for isolated targets; edge_norm = node_norm[dst]."""
    in_deg = np.bincount(np.asarray(dst, dtype=I64), minlength=num_nodes).astype(np.float32)
    if self_loop:
        node_norm = (np.float32(1.0) / (in_deg + np.float32(1.0))).astype(np.float32)
    else:
        with np.errstate(divide="ignore"):
            node_norm = np.where(in_deg == 0, np.float32(1.0), np.float32(1.0) / in_deg).astype(np.float32)
    return node_norm, node_norm[np.asarray(dst, dtype=I64)]


def largest_eigenvalues(src, dst, num_nodes):
    src, dst = np.asarray(src, dtype=I64), np.asarray(dst, dtype=I64)
    in_deg = np.bincount(dst, minlength=num_nodes).astype(np.float32)
    out_deg = np.bincount(src, minlength=num_nodes).astype(np.float32)
    return float((out_deg[src] + in_deg[dst]).max()), float((in_deg[src] + out_deg[dst]).max())


def add_reversed_edges(src, dst, edge_label, max_ne, max_nel):
    """One graph: m reversed edges appended after the m originals (id = max_ne + e, label + max_nel, flag 1)."""
    src, dst, edge_label = (np.asarray(a, dtype=I64) for a in (src, dst, edge_label))
    m = len(src)
    return dict(src=np.concatenate([src, dst]), dst=np.concatenate([dst, src]),
                edge_id=np.concatenate([np.arange(m), max_ne + np.arange(m)]).astype(I64),
                edge_label=np.concatenate([edge_label, edge_label + max_nel]),
                is_reversed=np.concatenate([np.zeros(m, I64), np.ones(m, I64)]))
