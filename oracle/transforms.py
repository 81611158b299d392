"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's
integer graph transforms, on plain numpy arrays.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product path (``dummynode4graphlearning_amd``) never does.

Parity status: PINNED.  Every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors produced by running the reference's own Python (imported under the
stand-ins of ``tests/golden/_ref_standins.py``) in the authoring container, including the
paper's worked example ``figure/edge2vertex.png`` (KAT-1) and the SI-style KAT-2 of
SURVEY.md section 8(c).

A *batched graph* is a disjoint union in the layout dgl.batch / PyG collate produce:
``node_ptr [G+1]``, ``edge_ptr [G+1]`` (graph boundaries), ``src/dst [E]`` GLOBAL node ids,
edges of graph g contiguous in ``[edge_ptr[g], edge_ptr[g+1])`` in the reference's eid order.
All loops are deliberately the naive per-graph / per-edge loops of the reference.
"""
import numpy as np

I64 = np.int64


def _as(a):
    return np.ascontiguousarray(np.asarray(a, dtype=I64))


# --------------------------------------------------------------------------------------
# a-1  GC dummy augmentation
# reference: graph_classification/data_processing/tu_data_processing.py:186-200, 213-214
# --------------------------------------------------------------------------------------
def dummy_augment_gc(node_ptr, edge_ptr, src, dst, node_label, edge_label):
    """Per graph: vertices 0..n-1 real + vertex n dummy (label 0); edges = the m original
    edges in order, then interleaved (n,v),(v,n) for v = 0..n-1 (label 0, IS_DUMMY 1).
    Returns dict of arrays for the augmented batch (global ids)."""
    node_ptr, edge_ptr, src, dst = _as(node_ptr), _as(edge_ptr), _as(src), _as(dst)
    node_label, edge_label = _as(node_label), _as(edge_label)
    G = len(node_ptr) - 1
    o_src, o_dst, o_el, o_ed, o_eid = [], [], [], [], []
    o_nl, o_nd, o_nid = [], [], []
    new_node_ptr, new_edge_ptr = [0], [0]
    for g in range(G):
        n0, n1 = int(node_ptr[g]), int(node_ptr[g + 1])
        e0, e1 = int(edge_ptr[g]), int(edge_ptr[g + 1])
        n, m = n1 - n0, e1 - e0
        base = new_node_ptr[-1]
        # graph.vs["LABEL"] = node_labels + [0]; IS_DUMMY = [0]*n + [1]        (:188-189)
        o_nl.extend(node_label[n0:n1].tolist() + [0])
        o_nd.extend([0] * n + [1])
        o_nid.extend(range(n + 1))                                            # (:213)
        # m real edges (:192)
        for e in range(e0, e1):
            o_src.append(int(src[e]) - n0 + base)
            o_dst.append(int(dst[e]) - n0 + base)
        # 2n dummy edges, interleaved (n, v), (v, n) (:193)
        for v in range(n):
            o_src.append(base + n); o_dst.append(base + v)
            o_src.append(base + v); o_dst.append(base + n)
        o_el.extend(edge_label[e0:e1].tolist() + [0] * (2 * n))               # (:194)
        o_ed.extend([0] * m + [1] * (2 * n))                                  # (:195)
        o_eid.extend(range(m + 2 * n))                                        # (:214)
        new_node_ptr.append(base + n + 1)
        new_edge_ptr.append(new_edge_ptr[-1] + m + 2 * n)
    return dict(
        node_ptr=_as(new_node_ptr), edge_ptr=_as(new_edge_ptr), src=_as(o_src), dst=_as(o_dst),
        node_label=_as(o_nl), edge_label=_as(o_el), is_dummy_node=_as(o_nd), is_dummy_edge=_as(o_ed),
        node_id=_as(o_nid), edge_id=_as(o_eid),
    )


# --------------------------------------------------------------------------------------
# a-4  SI dummy augmentation (GraphAdj branch)
# reference: subgraph_isomorphism/train.py:404-474
# --------------------------------------------------------------------------------------
def dummy_augment_si(node_ptr, edge_ptr, src, dst, node_id, node_label, edge_id, edge_label,
                     max_nv, max_nvl, max_ne, max_nel, is_reversed=None):
    """Per graph: add node (id=max_nv, label=max_nvl, is_dummy=1); add 2n edges BLOCKED:
    all (u -> dummy) then all (dummy -> u); shared edge ids max_ne / max_ne+1; relation ids
    max_nel / max_nel+1; is_dummy 1; is_reversed 0 / 1."""
    node_ptr, edge_ptr, src, dst = _as(node_ptr), _as(edge_ptr), _as(src), _as(dst)
    node_id, node_label, edge_id, edge_label = _as(node_id), _as(node_label), _as(edge_id), _as(edge_label)
    rev_in = None if is_reversed is None else _as(is_reversed)
    G = len(node_ptr) - 1
    o_src, o_dst, o_el, o_ed, o_eid, o_rev = [], [], [], [], [], []
    o_nl, o_nd, o_nid = [], [], []
    new_node_ptr, new_edge_ptr = [0], [0]
    for g in range(G):
        n0, n1 = int(node_ptr[g]), int(node_ptr[g + 1])
        e0, e1 = int(edge_ptr[g]), int(edge_ptr[g + 1])
        n, m = n1 - n0, e1 - e0
        base = new_node_ptr[-1]
        o_nid.extend(node_id[n0:n1].tolist() + [max_nv])                      # (:419)
        o_nl.extend(node_label[n0:n1].tolist() + [max_nvl])                   # (:420)
        o_nd.extend([0] * n + [1])                                            # (:421)
        for e in range(e0, e1):
            o_src.append(int(src[e]) - n0 + base)
            o_dst.append(int(dst[e]) - n0 + base)
        # th.cat([u, v]), th.cat([v, u])  with u = arange(n), v = [n]*n       (:409-410, 424-426)
        for u in range(n):
            o_src.append(base + u); o_dst.append(base + n)
        for u in range(n):
            o_src.append(base + n); o_dst.append(base + u)
        o_eid.extend(edge_id[e0:e1].tolist() + [max_ne] * n + [max_ne + 1] * n)      # (:412-413)
        o_el.extend(edge_label[e0:e1].tolist() + [max_nel] * n + [max_nel + 1] * n)  # (:414-415)
        o_ed.extend([0] * m + [1] * (2 * n))                                         # (:430)
        old_rev = [0] * m if rev_in is None else rev_in[e0:e1].tolist()
        o_rev.extend(old_rev + [0] * n + [1] * n)                                    # (:431, 434-435)
        new_node_ptr.append(base + n + 1)
        new_edge_ptr.append(new_edge_ptr[-1] + m + 2 * n)
    return dict(
        node_ptr=_as(new_node_ptr), edge_ptr=_as(new_edge_ptr), src=_as(o_src), dst=_as(o_dst),
        node_id=_as(o_nid), node_label=_as(o_nl), is_dummy_node=_as(o_nd),
        edge_id=_as(o_eid), edge_label=_as(o_el), is_dummy_edge=_as(o_ed), is_reversed=_as(o_rev),
    )


# --------------------------------------------------------------------------------------
# a-2 / a-5  edge-to-vertex ("conjugate") transform L_Phi, one graph
# reference: tu_data_processing.py:223-338 (GC)  /  SI utils/graph.py:177-267 (igraph branch)
# --------------------------------------------------------------------------------------
def _conjugate_one(n, src, dst, eids, node_label, is_dummy_edge, mode):
    """src/dst LOCAL ids of one graph, in eid order.
    Returns (rep, cu, cv, shared):
      rep[k]    = local index of the edge whose attributes conj-vertex k copies,
      cu/cv[t]  = conj-edge endpoints (conj-vertex indices, after compaction),
      shared[t] = local index of the original vertex whose attributes conj-edge t copies.
    mode "gc": merge IS_DUMMY edges into one vertex, drop (Phi,Phi), dedupe on (uid,vid).
    mode "si": vertices keyed by eids, dedupe on (uid, label(shared), vid).
    mode "line": no merge, no dedupe (GC without IS_DUMMY attribute)."""
    m = len(src)
    if m == 0:
        return [], [], [], []
    # conj vertices: one slot per id, representative = first edge with that id
    num_slots = max(eids) + 1                                   # (:230 / :185)
    id2vertex = [None] * num_slots
    for e, eid in enumerate(eids):                               # (:232-236 / :187-191)
        if id2vertex[eid] is None:
            id2vertex[eid] = e
        else:
            id2vertex[eid] = min(id2vertex[eid], e)
    # sorted in-incidence lists (graph.incident(source, "in") sorted) (:266 / :219)
    in_edges = [[] for _ in range(n)]
    for e in range(m):
        in_edges[dst[e]].append(e)
    edges, edge_indices = [], []
    used_keys = set()
    for e in range(m):                                           # (:261-274 / :214-227)
        source = src[e]
        vid = eids[e]
        elabel = node_label[source]
        for ie in in_edges[source]:
            uid = eids[ie]
            key = (uid, elabel, vid)
            if mode == "line" or key not in used_keys:
                used_keys.add(key)
                edges.append((uid, vid))
                edge_indices.append(source)
    if mode == "gc":                                             # (:289-318)
        dummy_eids = [eids[e] for e in range(m) if is_dummy_edge[e]]
        if len(dummy_eids) > 0:
            for e in dummy_eids[1:]:
                id2vertex[e] = None
            phi = dummy_eids[0]
            dset = set(dummy_eids)
            used = {(phi, phi)}
            new_edges, new_idx = [], []
            for t in range(len(edges)):
                uid, vid = edges[t]
                if uid in dset:
                    uid = phi
                if vid in dset:
                    vid = phi
                if (uid, vid) not in used:
                    used.add((uid, vid))
                    new_edges.append((uid, vid))
                    new_idx.append(edge_indices[t])
            edges, edge_indices = new_edges, new_idx
    # delete unused vertex slots -> compact renumbering (:333-336 / :264-267)
    keep = [s for s in range(num_slots) if id2vertex[s] is not None]
    remap = {s: k for k, s in enumerate(keep)}
    rep = [id2vertex[s] for s in keep]
    cu = [remap[u] for u, _ in edges]
    cv = [remap[v] for _, v in edges]
    return rep, cu, cv, edge_indices


def conjugate(node_ptr, edge_ptr, src, dst, node_label, edge_id=None, is_dummy_edge=None, mode="gc"):
    """Batched L_Phi.  Output (global ids, batched layout):
      cnode_ptr [G+1], cedge_ptr [G+1], csrc, cdst [E'],
      rep_edge [N']   global index of the input edge each conj-vertex copies its attributes from,
      shared_node [E'] global index of the input vertex each conj-edge copies its attributes from.
    ``edge_id`` defaults to the local edge index (GC: es["ID"] = range, :214)."""
    node_ptr, edge_ptr, src, dst, node_label = _as(node_ptr), _as(edge_ptr), _as(src), _as(dst), _as(node_label)
    G = len(node_ptr) - 1
    if is_dummy_edge is None:
        is_dummy_edge = np.zeros(len(src), dtype=I64)
    is_dummy_edge = _as(is_dummy_edge)
    cnode_ptr, cedge_ptr = [0], [0]
    o_rep, o_cu, o_cv, o_sh = [], [], [], []
    for g in range(G):
        n0, n1 = int(node_ptr[g]), int(node_ptr[g + 1])
        e0, e1 = int(edge_ptr[g]), int(edge_ptr[g + 1])
        ls = (src[e0:e1] - n0).tolist()
        ld = (dst[e0:e1] - n0).tolist()
        eids = list(range(e1 - e0)) if edge_id is None else _as(edge_id)[e0:e1].tolist()
        rep, cu, cv, sh = _conjugate_one(
            n1 - n0, ls, ld, eids, node_label[n0:n1].tolist(), is_dummy_edge[e0:e1].tolist(), mode)
        vb = cnode_ptr[-1]
        o_rep.extend(e0 + r for r in rep)
        o_cu.extend(vb + u for u in cu)
        o_cv.extend(vb + v for v in cv)
        o_sh.extend(n0 + s for s in sh)
        cnode_ptr.append(vb + len(rep))
        cedge_ptr.append(cedge_ptr[-1] + len(cu))
    return dict(cnode_ptr=_as(cnode_ptr), cedge_ptr=_as(cedge_ptr), csrc=_as(o_cu), cdst=_as(o_cv),
                rep_edge=_as(o_rep), shared_node=_as(o_sh))


# --------------------------------------------------------------------------------------
# CSR build used by the gather/segment kernels (stable sort by (key)) -- the reference has
# no counterpart (DGL/torch-scatter do this internally); restated here only so the device
# index build has a bit-exact checker.
# --------------------------------------------------------------------------------------
def csr_by_key(key, num_segments):
    """Stable counting sort: returns (ptr [num_segments+1], perm [E]) with perm listing
    element indices grouped by key, ascending original index inside a group."""
    key = _as(key)
    perm = np.argsort(key, kind="stable").astype(I64)
    cnt = np.bincount(key, minlength=num_segments).astype(I64)
    ptr = np.zeros(num_segments + 1, dtype=I64)
    np.cumsum(cnt, out=ptr[1:])
    return ptr, perm


# --------------------------------------------------------------------------------------
# RGCN degree norms (int part) -- reference: SI models/rgcn.py:132-151
# --------------------------------------------------------------------------------------
def degrees(src, dst, num_nodes):
    src, dst = _as(src), _as(dst)
    return (np.bincount(dst, minlength=num_nodes).astype(I64),
            np.bincount(src, minlength=num_nodes).astype(I64))


# --------------------------------------------------------------------------------------
# a-1 (front half)  TU raw arrays -> batched COO
# reference: tu_data_processing.py:154-183, 216-218
# --------------------------------------------------------------------------------------
def tu_raw_to_batch(A, graph_indicator, node_labels=None, edge_labels=None):
    """A: [m,2] 1-based global ids grouped by graph.  Labels are shifted so the minimum is 1
    (:154-169).  Graph boundaries are found by walking A (:179-181); graphs that have no edge
    AFTER the last edge of the file are dropped (the walk ends at j == k)."""
    A = [(int(a), int(b)) for a, b in A]
    gi = [int(x) for x in graph_indicator]
    if node_labels is None or len(node_labels) == 0:
        nl = [1] * len(gi)
    else:
        nl = [int(x) for x in node_labels]
        mn = min(nl)
        if mn != 1:
            nl = [x - mn + 1 for x in nl]
    if edge_labels is None or len(edge_labels) == 0:
        el = [1] * len(A)
    else:
        el = [int(x) for x in edge_labels]
        mn = min(el)
        if mn != 1:
            el = [x - mn + 1 for x in el]
    counts = {}
    for x in gi:
        counts[x] = counts.get(x, 0) + 1
    node_ptr, edge_ptr = [0], [0]
    i = j = 0
    k = len(A)
    gd = gi[0]
    while j < k:
        while j < k and gd == gi[A[j][0] - 1] and gd == gi[A[j][1] - 1]:
            j += 1
        node_ptr.append(node_ptr[-1] + counts.get(gd, 0))
        edge_ptr.append(j)
        gd += 1
        i = j
    N = node_ptr[-1]
    src = [a - 1 for a, _ in A]
    dst = [b - 1 for _, b in A]
    return dict(node_ptr=_as(node_ptr), edge_ptr=_as(edge_ptr), src=_as(src), dst=_as(dst),
                node_label=_as(nl[:N]), edge_label=_as(el))
