"""Device-side graph transforms: dummy-node augmentation and the edge-to-vertex (conjugate) transform.

One-shot index builds on the GPU through libdn_hip.so, bit-exact with the reference's per-graph Python
loops (tests/test_gpu_transforms.py checks them against the oracle and the golden fixtures):

  dummy_augment_gc   <- load_graph_data_from_TUDatadir(with_dummy=True)
                        graph_classification/data_processing/tu_data_processing.py:186-200
  dummy_augment_si   <- add_dummy_nodes_edges (GraphAdj branch)   subgraph_isomorphism/train.py:404-474
  conjugate          <- convert_conjugate_graph_forward           tu_data_processing.py:223-338   (mode "gc"/"line")
                        convert_conjugate_graph (igraph branch)   subgraph_isomorphism/utils/graph.py:177-267 (mode "si")

A batched graph is given as int tensors on the GPU: node_ptr/edge_ptr [G+1], src/dst [E] global node ids,
edges of one graph contiguous in the reference's eid order.  Outputs are int32 device tensors.
"""
import ctypes

import torch

from ._lib import check, lib, ptr, require_gpu, stream_ptr

I32 = torch.int32
_MODES = {"gc": 0, "si": 1, "line": 2}


def _i32(t):
    return None if t is None else t.to(I32).contiguous()


def _u8(t):
    return None if t is None else t.to(torch.uint8).contiguous()


def dummy_augment_gc(node_ptr, edge_ptr, src, dst, node_label, edge_label):
    node_ptr, edge_ptr, src, dst, node_label, edge_label = map(_i32, (node_ptr, edge_ptr, src, dst, node_label, edge_label))
    dev = require_gpu(node_ptr, edge_ptr, src, dst, node_label, edge_label)
    G, N, E = node_ptr.numel() - 1, node_label.numel(), src.numel()
    No, Eo = N + G, E + 2 * N
    e32 = lambda n: torch.empty(n, dtype=I32, device=dev)  # noqa: E731
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)  # noqa: E731
    o = dict(node_ptr=e32(G + 1), edge_ptr=e32(G + 1), src=e32(Eo), dst=e32(Eo), node_label=e32(No), edge_label=e32(Eo),
             is_dummy_node=u8(No), is_dummy_edge=u8(Eo), node_id=e32(No), edge_id=e32(Eo))
    check(lib().dn_dummy_augment_gc_i32(G, N, E, ptr(node_ptr), ptr(edge_ptr), ptr(src), ptr(dst), ptr(node_label),
                                        ptr(edge_label), ptr(o["node_ptr"]), ptr(o["edge_ptr"]), ptr(o["src"]),
                                        ptr(o["dst"]), ptr(o["node_label"]), ptr(o["edge_label"]), ptr(o["is_dummy_node"]),
                                        ptr(o["is_dummy_edge"]), ptr(o["node_id"]), ptr(o["edge_id"]), stream_ptr()),
          "dn_dummy_augment_gc_i32")
    return o


def dummy_augment_si(node_ptr, edge_ptr, src, dst, node_id, node_label, edge_id, edge_label,
                     max_nv, max_nvl, max_ne, max_nel, is_reversed=None):
    node_ptr, edge_ptr, src, dst, node_id, node_label, edge_id, edge_label = map(
        _i32, (node_ptr, edge_ptr, src, dst, node_id, node_label, edge_id, edge_label))
    is_reversed = _u8(is_reversed)
    dev = require_gpu(node_ptr, edge_ptr, src, dst, node_id, node_label, edge_id, edge_label, is_reversed)
    G, N, E = node_ptr.numel() - 1, node_label.numel(), src.numel()
    No, Eo = N + G, E + 2 * N
    e32 = lambda n: torch.empty(n, dtype=I32, device=dev)  # noqa: E731
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)  # noqa: E731
    o = dict(node_ptr=e32(G + 1), edge_ptr=e32(G + 1), src=e32(Eo), dst=e32(Eo), node_id=e32(No), node_label=e32(No),
             edge_id=e32(Eo), edge_label=e32(Eo), is_dummy_node=u8(No), is_dummy_edge=u8(Eo), is_reversed=u8(Eo))
    check(lib().dn_dummy_augment_si_i32(G, N, E, ptr(node_ptr), ptr(edge_ptr), ptr(src), ptr(dst), ptr(node_id),
                                        ptr(node_label), ptr(edge_id), ptr(edge_label), ptr(is_reversed),
                                        int(max_nv), int(max_nvl), int(max_ne), int(max_nel),
                                        ptr(o["node_ptr"]), ptr(o["edge_ptr"]), ptr(o["src"]), ptr(o["dst"]),
                                        ptr(o["node_id"]), ptr(o["node_label"]), ptr(o["edge_id"]), ptr(o["edge_label"]),
                                        ptr(o["is_dummy_node"]), ptr(o["is_dummy_edge"]), ptr(o["is_reversed"]),
                                        stream_ptr()), "dn_dummy_augment_si_i32")
    return o


def conjugate(node_ptr, edge_ptr, src, dst, node_label, edge_id=None, is_dummy_edge=None, mode="gc"):
    """L_Phi of a batched graph.  Returns dict(cnode_ptr, cedge_ptr, csrc, cdst, rep_edge, shared_node):
    conj-vertex k copies the attributes of input edge rep_edge[k]; conj-edge t those of input vertex
    shared_node[t]."""
    node_ptr, edge_ptr, src, dst, node_label, edge_id = map(_i32, (node_ptr, edge_ptr, src, dst, node_label, edge_id))
    is_dummy_edge = _u8(is_dummy_edge)
    dev = require_gpu(node_ptr, edge_ptr, src, dst, node_label, edge_id, is_dummy_edge)
    m = _MODES[mode]
    G, N, E = node_ptr.numel() - 1, node_label.numel(), src.numel()
    if m == 0 and is_dummy_edge is None:
        is_dummy_edge = torch.zeros(E, dtype=torch.uint8, device=dev)
    e32 = lambda n: torch.empty(max(n, 1), dtype=I32, device=dev)  # noqa: E731
    L = lib()
    nb = L.dn_conjugate_workspace_bytes(G, N, E, -1)
    if nb == 0:
        check(-2, "dn_conjugate_workspace_bytes")
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    T = ctypes.c_int64(0)
    check(L.dn_conjugate_count_i32(N, E, ptr(src), ptr(dst), ctypes.byref(T), ptr(ws), ws.numel(), stream_ptr()),
          "dn_conjugate_count_i32")
    T = int(T.value)
    nb = L.dn_conjugate_workspace_bytes(G, N, E, T)
    if nb == 0:
        check(-2, "dn_conjugate_workspace_bytes")
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    cnode_ptr, cedge_ptr = e32(G + 1), e32(G + 1)
    csrc, cdst, shared, rep = e32(T), e32(T), e32(T), e32(E)
    counts = (ctypes.c_int64 * 2)()
    check(L.dn_conjugate_build_i32(m, G, N, E, T, ptr(node_ptr), ptr(edge_ptr), ptr(src), ptr(dst), ptr(node_label),
                                   ptr(edge_id), ptr(is_dummy_edge), ptr(cnode_ptr), ptr(cedge_ptr), ptr(csrc), ptr(cdst),
                                   ptr(rep), ptr(shared), counts, ptr(ws), ws.numel(), stream_ptr()),
          "dn_conjugate_build_i32")
    Nc, Ec = int(counts[0]), int(counts[1])
    return dict(cnode_ptr=cnode_ptr[:G + 1], cedge_ptr=cedge_ptr[:G + 1], csrc=csrc[:Ec], cdst=cdst[:Ec],
                rep_edge=rep[:Nc], shared_node=shared[:Ec], num_raw=T)
