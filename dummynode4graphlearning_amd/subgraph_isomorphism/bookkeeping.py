"""Integer bookkeeping either side of the edge-to-vertex transform in subgraph isomorphism, on the GPU.

Same names and argument meaning as the reference's numba / per-sample functions, vectorised over the subisomorphisms:

  get_conjugate_subisomorphisms   subgraph_isomorphism/utils/graph.py:291-330
  compute_edgeseq_subisoweights   subgraph_isomorphism/dataset.py:63-108
  compute_nodeseq_subisoweights   dataset.py:54-60
  compute_norm                    utils/graph.py:11-38      (graph given as src/dst tensors)
  compute_largest_eigenvalues     utils/graph.py:41-71
  add_reversed_edges              train.py:323-345          (GraphAdj branch, whole batch at once)

The reference bisects the (src, dst)-sorted graph edge list once per (subisomorphism, pattern key) in a Python/numba
loop; here the graph edges are sorted once by (src, dst, label) and every (subisomorphism, key, run label) query is one
row of two torch.searchsorted calls.  "Last matching edge wins" (conjugate) is the last element of the query's range
(stable sort keeps the edge order inside equal keys), the match weights are a difference array over the ranges.
"""
import numpy as np
import torch

from .. import ops


def _pattern_queries(p_u, p_v, p_el):
    """Host side (a pattern has a handful of edges): runs of consecutive equal (u, v) in eid order; per distinct key the
    labels of its LAST run, column = rank of the key by first appearance.  -> int64 arrays (u, v, label, column)."""
    p_u, p_v, p_el = (np.asarray(t.cpu() if torch.is_tensor(t) else t, dtype=np.int64) for t in (p_u, p_v, p_el))
    n = len(p_el)
    start = np.ones(n, dtype=bool)
    start[1:] = (p_u[1:] != p_u[:-1]) | (p_v[1:] != p_v[:-1])
    run_of = np.cumsum(start) - 1
    run_u, run_v = p_u[start], p_v[start]
    first, last = {}, {}
    for r, key in enumerate(zip(run_u.tolist(), run_v.tolist())):
        first.setdefault(key, len(first))
        last[key] = r
    qu, qv, ql, qc = [], [], [], []
    for key, r in last.items():
        for e in p_el[run_of == r].tolist():
            qu.append(key[0]); qv.append(key[1]); ql.append(e); qc.append(first[key])
    return tuple(np.asarray(a, dtype=np.int64) for a in (qu, qv, ql, qc))


def _ranges(p_u, p_v, p_el, g_u, g_v, g_el, subisomorphisms):
    dev = g_u.device
    sub = subisomorphisms.to(dev).long()
    sub = sub.reshape(-1, sub.shape[-1]) if sub.dim() == 2 else sub.reshape(0, 1)
    qu, qv, ql, qc = (torch.from_numpy(a).to(dev) for a in _pattern_queries(p_u, p_v, p_el))
    g_u, g_v, g_el = g_u.long(), g_v.long(), g_el.long()
    mod = int(max(int(torch.as_tensor(p_u).max()), int(torch.as_tensor(p_v).max()), int(g_u.max()), int(g_v.max()))) + 1
    nl = int(max(int(g_el.max()), int(ql.max()) if ql.numel() else 0)) + 1
    gkey, perm = torch.sort((g_u * mod + g_v) * nl + g_el, stable=True)
    qkey = ((sub[:, qu] * mod + sub[:, qv]) * nl + ql.unsqueeze(0)).contiguous()          # [S, Q]
    lo = torch.searchsorted(gkey, qkey, right=False)
    hi = torch.searchsorted(gkey, qkey, right=True)
    return sub, qc, perm, lo, hi


def get_conjugate_subisomorphisms(p_u, p_v, p_el, g_u, g_v, g_el, subisomorphisms):
    """[S, len(p_el)] int64 (see oracle/si_bookkeeping.py for the exact semantics, incl. the zero columns)."""
    sub, qc, perm, lo, hi = _ranges(p_u, p_v, p_el, g_u, g_v, g_el, subisomorphisms)
    S, P = sub.shape[0], int(torch.as_tensor(p_el).numel())
    out = torch.zeros((S, P), dtype=torch.long, device=perm.device)
    if S == 0 or qc.numel() == 0:
        return out
    last = perm[(hi - 1).clamp(min=0)]
    cand = torch.where(hi > lo, last, torch.zeros_like(last))
    return out.scatter_reduce(1, qc.unsqueeze(0).expand(S, -1), cand, reduce="amax", include_self=True)


def compute_edgeseq_subisoweights(p_u, p_v, p_el, g_u, g_v, g_el, subisomorphisms):
    sub, qc, perm, lo, hi = _ranges(p_u, p_v, p_el, g_u, g_v, g_el, subisomorphisms)
    E = perm.numel()
    diff = torch.zeros(E + 1, dtype=torch.long, device=perm.device)
    ones = torch.ones(lo.numel(), dtype=torch.long, device=perm.device)
    diff.index_add_(0, lo.reshape(-1), ones)
    diff.index_add_(0, hi.reshape(-1), -ones)
    w = torch.zeros(E, dtype=torch.long, device=perm.device)
    w[perm] = torch.cumsum(diff, 0)[:E]
    return w


def compute_nodeseq_subisoweights(num_nodes, subisomorphisms):
    return torch.bincount(subisomorphisms.reshape(-1).long(), minlength=int(num_nodes))


def compute_norm(src, dst, num_nodes, self_loop):
    """node_norm [N,1], edge_norm [E,1] float32 (in-degrees from dn_degrees_i32)."""
    in_deg, _ = ops.degrees(src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous(), int(num_nodes))
    in_deg = in_deg.float()
    if self_loop:
        node_norm = (in_deg + 1).reciprocal().unsqueeze(-1)
    else:
        node_norm = in_deg.reciprocal().masked_fill_(in_deg == 0, 1.0).unsqueeze(-1)
    return node_norm, node_norm[dst.long()]


def compute_largest_eigenvalues(src, dst, num_nodes):
    in_deg, out_deg = ops.degrees(src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous(), int(num_nodes))
    in_deg, out_deg = in_deg.float(), out_deg.float()
    u, v = src.long(), dst.long()
    return (out_deg[u] + in_deg[v]).max(), (in_deg[u] + out_deg[v]).max()


def add_reversed_edges(edge_ptr, src, dst, edge_id, edge_label, max_ne, max_nel):
    """Whole batch: per graph the m original edges followed by their m reversals (id = max_ne + e, label + max_nel,
    is_reversed = 1), which is what the per-sample add_edges call leaves behind after dgl.batch."""
    ep = edge_ptr.long()
    m = ep[1:] - ep[:-1]
    E = int(src.numel())
    g = torch.repeat_interleave(torch.arange(m.numel(), device=src.device), m)
    local = torch.arange(E, device=src.device) - ep[g]
    pos_o = 2 * ep[g] + local                      # new slot of the original edge
    pos_r = pos_o + m[g]                           # ... and of its reversal
    def mix(a, b):
        out = torch.empty(2 * E, dtype=a.dtype, device=a.device)
        out[pos_o], out[pos_r] = a, b
        return out
    return dict(edge_ptr=(2 * ep).to(edge_ptr.dtype), src=mix(src, dst), dst=mix(dst, src),
                edge_id=mix(edge_id, (max_ne + local).to(edge_id.dtype)), edge_label=mix(edge_label, edge_label + max_nel),
                is_reversed=mix(torch.zeros(E, dtype=torch.uint8, device=src.device),
                                torch.ones(E, dtype=torch.uint8, device=src.device)))
