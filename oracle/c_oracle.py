"""ctypes wrapper of oracle/dn_oracle.c (ORACLE: test infrastructure only).  Same array interface as
oracle/transforms.py; used where the pure-Python loops are too slow (full-size bit-exact checks)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libdn_oracle.so")
_lib = None
I64 = np.int64


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "dn_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = ctypes.CDLL(_LIB)
        _lib.dno_conjugate_raw_count.restype = ctypes.c_int64
    return _lib


def _a(x):
    return np.ascontiguousarray(np.asarray(x, dtype=I64))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def dummy_augment_gc(node_ptr, edge_ptr, src, dst, node_label, edge_label):
    node_ptr, edge_ptr, src, dst, node_label, edge_label = map(_a, (node_ptr, edge_ptr, src, dst, node_label, edge_label))
    G, N, E = len(node_ptr) - 1, len(node_label), len(src)
    z = lambda n: np.zeros(n, dtype=I64)  # noqa: E731
    o = dict(node_ptr=z(G + 1), edge_ptr=z(G + 1), src=z(E + 2 * N), dst=z(E + 2 * N), node_label=z(N + G),
             edge_label=z(E + 2 * N), is_dummy_node=z(N + G), is_dummy_edge=z(E + 2 * N), node_id=z(N + G), edge_id=z(E + 2 * N))
    lib().dno_dummy_augment_gc(ctypes.c_int64(G), _p(node_ptr), _p(edge_ptr), _p(src), _p(dst), _p(node_label), _p(edge_label),
                               *(_p(o[k]) for k in ("node_ptr", "edge_ptr", "src", "dst", "node_label", "edge_label",
                                                    "is_dummy_node", "is_dummy_edge", "node_id", "edge_id")))
    return o


def dummy_augment_si(node_ptr, edge_ptr, src, dst, node_id, node_label, edge_id, edge_label, max_nv, max_nvl, max_ne, max_nel,
                     is_reversed=None):
    node_ptr, edge_ptr, src, dst, node_id, node_label, edge_id, edge_label = map(
        _a, (node_ptr, edge_ptr, src, dst, node_id, node_label, edge_id, edge_label))
    rev = None if is_reversed is None else _a(is_reversed)
    G, N, E = len(node_ptr) - 1, len(node_label), len(src)
    z = lambda n: np.zeros(n, dtype=I64)  # noqa: E731
    o = dict(node_ptr=z(G + 1), edge_ptr=z(G + 1), src=z(E + 2 * N), dst=z(E + 2 * N), node_id=z(N + G), node_label=z(N + G),
             edge_id=z(E + 2 * N), edge_label=z(E + 2 * N), is_dummy_node=z(N + G), is_dummy_edge=z(E + 2 * N),
             is_reversed=z(E + 2 * N))
    c = ctypes.c_int64
    lib().dno_dummy_augment_si(c(G), _p(node_ptr), _p(edge_ptr), _p(src), _p(dst), _p(node_id), _p(node_label), _p(edge_id),
                               _p(edge_label), _p(rev), c(max_nv), c(max_nvl), c(max_ne), c(max_nel),
                               *(_p(o[k]) for k in ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id",
                                                    "edge_label", "is_dummy_node", "is_dummy_edge", "is_reversed")))
    return o


def conjugate(node_ptr, edge_ptr, src, dst, node_label, edge_id=None, is_dummy_edge=None, mode="gc"):
    node_ptr, edge_ptr, src, dst, node_label = map(_a, (node_ptr, edge_ptr, src, dst, node_label))
    eid = None if edge_id is None else _a(edge_id)
    G, N, E = len(node_ptr) - 1, len(node_label), len(src)
    dm = _a(is_dummy_edge) if is_dummy_edge is not None else np.zeros(E, dtype=I64)
    T = int(lib().dno_conjugate_raw_count(ctypes.c_int64(N), ctypes.c_int64(E), _p(src), _p(dst)))
    z = lambda n: np.zeros(max(n, 1), dtype=I64)  # noqa: E731
    cn, ce, cs, cd, rep, sh, cnt = z(G + 1), z(G + 1), z(T), z(T), z(E), z(T), z(2)
    lib().dno_conjugate(ctypes.c_int({"gc": 0, "si": 1, "line": 2}[mode]), ctypes.c_int64(G), _p(node_ptr), _p(edge_ptr), _p(src),
                        _p(dst), _p(node_label), _p(eid), _p(dm), _p(cn), _p(ce), _p(cs), _p(cd), _p(rep), _p(sh), _p(cnt))
    nv, ne = int(cnt[0]), int(cnt[1])
    return dict(cnode_ptr=cn[:G + 1], cedge_ptr=ce[:G + 1], csrc=cs[:ne], cdst=cd[:ne], rep_edge=rep[:nv], shared_node=sh[:ne],
                num_raw=T)
