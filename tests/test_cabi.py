"""The C-ABI library builds, loads and exports every symbol include/dn_hip.h declares (CPU only: no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dn_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dn_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from dummynode4graphlearning_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def test_header_and_binding_agree(lib):
    declared = _declared()
    assert len(declared) >= 25
    assert declared == lib.exported_symbols(), "include/dn_hip.h and _lib._SIGS list different entry points"


def test_library_exports_every_declared_symbol(lib):
    handle = ctypes.CDLL(lib.LIB_PATH)
    for name in _declared():
        assert hasattr(handle, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (dn_[a-z0-9_]+)", out))
    assert set(_declared()) <= exported


def test_signatures_are_plain_c(lib):
    """No torch / C++ types at the boundary: the header must compile as C."""
    src = '#include "dn_hip.h"\nint main(void) { return dn_version == 0; }\n'
    subprocess.run(["gcc", "-x", "c", "-std=c99", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-"],
                   input=src.encode(), check=True)


def test_argument_errors_are_reported_without_touching_the_gpu(lib):
    L = lib.lib()
    assert L.dn_version() >= 100
    rc = L.dn_gather_segsum_f32(None, 0, 0, None, None, None, 0, 0, None, None, 0.0, 0, None)   # H == 0
    assert rc == -1 and b"H must be > 0" in L.dn_last_error()
    rc = L.dn_rows_transform_bf16(None, None, 0x7fffffff, None, 100, 100, None, None, 0, None, None, 1, None, 0, 0.0, None)
    assert rc == -1 and b"unsupported widths" in L.dn_last_error()
    rc = L.dn_conjugate_build_i32(7, 0, 0, 0, 0, *([None] * 13), (ctypes.c_int64 * 2)(), None, 0, None)
    assert rc == -1 and b"bad mode" in L.dn_last_error()
    with pytest.raises(lib.DnHipError):
        lib.check(rc, "dn_conjugate_build_i32")
    # round-2 entry points: the same contract (arguments are checked before anything is launched)
    rc = L.dn_fold_tail_bf16(None, None, 5, 100, None, None, None, None, 0, None)
    assert rc == -1 and b"H must be 64, 128 or 256" in L.dn_last_error()
    rc = L.dn_fold_tail_bf16(None, None, 5, 256, None, None, None, None, 0, None)
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
    assert L.dn_fold_tail_bf16(None, None, 0, 256, None, None, None, None, 0, None) == 0           # nothing to do
    ok = ctypes.c_int32(7)
    assert L.dn_fold_tables_build_i32(0, 0, None, None, None, None, ctypes.byref(ok), None, 0, None) == 0 and ok.value == 0
    rc = L.dn_fold_tables_build_i32(10, 2, None, None, None, None, ctypes.byref(ok), None, 0, None)
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
    # round-4 entry points
    rc = L.dn_rows_close_bf16(None, 128, None, 0, None, None, None, None, 256, None, None, 8, *([None] * 7))
    assert rc == -1 and b"unsupported width" in L.dn_last_error()
    rc = L.dn_rows_close_bf16(None, 256, None, 0, None, None, None, None, 256, None, None, 8, *([None] * 7))
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
    rc = L.dn_rows_close_bf16(None, 256, None, 0, None, None, None, None, 256, None, None, 8, None, None, None, ctypes.c_void_p(16),
                              None, None, None)
    assert rc == -1 and b"absorbed fold" in L.dn_last_error()
    assert L.dn_rows_close_bf16(None, 256, None, 0, None, None, None, None, 256, None, None, 0, *([None] * 7)) == 0
    assert L.dn_close_units_capacity(2, 320, 4) == 2 * 2 + 10 + 1 + 4 * 9 + 2
    assert L.dn_close_units_workspace_bytes(2, 4) > 0
    rc = L.dn_close_units_build_i32(10, 5, 4, None, 1, 0, 0, None, None, 3, 0, 0, None, ctypes.c_void_p(16), None, 0, None, None, None, None, 0, None, 0,
                                    None)
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
    rc = L.dn_close_units_build_i32(100, 5, 4, None, 1, 0, 0, None, None, 3, 0, 0, None, ctypes.c_void_p(16), None, 0, None, None, None, None, 0, None, 0,
                                    None)
    assert rc == -1 and b"32-node windows" in L.dn_last_error()
    rc = L.dn_close_units_build_i32(10, 5, 4, None, 1, 0, 1, None, None, 3, 0, 0, None, ctypes.c_void_p(16), None, 0, None, None, None, None, 0, None, 0,
                                    None)
    assert rc == -1 and b"multiple of 8 workgroups" in L.dn_last_error()
    ok = ctypes.c_int32(0)
    rc = L.dn_fold_graph_tiles_build_i32(10, 2, None, None, None, None, None, ctypes.c_void_p(16), None)
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
    # round-6 entry points: graphs that span several tiles
    rc = L.dn_close_units_build_i32(10, 5, 4, None, 1, 1, 2, None, None, 3, 0, 0, None, ctypes.c_void_p(16), None, 0, None, None, None, None,
                                    0, None, 0, None)
    assert rc == -1 and b"orders 2 / 3" in L.dn_last_error()
    assert L.dn_fold_graph_tiles_multi_capacity(100, 8) == 3 + 8 + 1
    rc = L.dn_fold_graph_tiles_multi_build_i32(10, 2, None, None, None, 8, None, None, None, None, 0, ctypes.c_void_p(16), None)
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
    assert L.dn_bdd_compose(None, 0, 4, 16, 16, 2, None, None) == 0
    rc = L.dn_bdd_compose(None, 3, 4, 16, 16, 3, ctypes.c_void_p(16), ctypes.c_void_p(16), None) if False else L.dn_bdd_extract(
        ctypes.c_void_p(16), 3, 4, 16, 16, 3, ctypes.c_void_p(16), None)
    assert rc == -1 and b"elem_bytes" in L.dn_last_error()
    rc = L.dn_rows_selfsum_bf16(*([None] * 1), 256, *([None] * 4), 0, None, 6, 8, None, ctypes.c_void_p(16), None, 0, None, None, 0, 0, 0, None)
    assert rc == -1 and b"fold_info needs seg_part" in L.dn_last_error()
    rc = L.dn_overflow_rows_add_bf16(None, 100, None, 6, 8, None, None, 5, 0, 0, None, None)
    assert rc == -1 and b"unsupported width" in L.dn_last_error()
    rc = L.dn_overflow_rows_add_bf16(None, 256, None, 6, 8, None, None, 5, 0, 0, None, None)
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
    rc = L.dn_batchnorm_rows_f32(ctypes.c_void_p(16), 8, 64, None, None, 1e-5, ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16),
                                 ctypes.c_void_p(16), ctypes.c_void_p(16), None, 0.1, 0, None, ctypes.c_void_p(16), 1 << 20, None)
    assert rc == -1 and b"running_mean and running_var come together" in L.dn_last_error()


def test_local_index_and_async_table_entry_points_check_their_arguments(lib):
    """dn_row_index_build_local_i32 / the async table builders: sizes and pointers are checked before anything is launched"""
    L = lib.lib()
    assert L.dn_row_index_local_workspace_bytes(4, 10, 65, 20) == 0 and b"bad sizes" in L.dn_last_error()     # R > 64
    assert L.dn_row_index_local_workspace_bytes(-1, 10, 3, 20) == 0
    counts, rel, modes, st = (ctypes.c_int64 * 5)(), (ctypes.c_int32 * 4)(), (ctypes.c_int32 * 3)(), ctypes.c_int32(0)
    P16 = ctypes.c_void_p(16)
    rc = L.dn_row_index_build_local_i32(4, 10, 65, 20, P16, P16, P16, P16, P16, 1, 0.75, *([P16] * 10), counts, rel, modes,
                                        ctypes.byref(st), None, None, None, None, None, None, P16, 1 << 20, None)
    assert rc == -1 and b"more than 64 relations" in L.dn_last_error()
    rc = L.dn_row_index_build_local_i32(4, 10, 3, 20, None, P16, P16, P16, P16, 1, 0.75, *([P16] * 10), counts, rel, modes,
                                        ctypes.byref(st), None, None, None, None, None, None, P16, 1 << 20, None)
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
    rc = L.dn_slot_table_build_i32(10, 5, 6, P16, P16, 0, 0, None, None, None, None)
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
    rc = L.dn_slot_table_build_i32(10, 5, 1, P16, P16, 0, 0, None, P16, None, None)
    assert rc == -1 and b"bad sizes" in L.dn_last_error()
    rc = L.dn_sweep_tables_build_i32(100, P16, P16, P16, 10, 32, 8, 0, P16, None, None)
    assert rc == -1 and b"num_rels <= 64" in L.dn_last_error()
    rc = L.dn_sweep_tables_build_i32(4, P16, P16, P16, 10, 32, 0, 0, P16, None, None)
    assert rc == -1 and b"bad sizes" in L.dn_last_error()
    rc = L.dn_fold_tables_build_async_i32(10, 2, P16, P16, P16, P16, None, P16, 1 << 20, None)
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()


def test_conv_index_entry_point_checks_its_arguments(lib):
    """dn_conv_index_build_i32 (the whole per-batch index in one call): sizes, pointers and capacities are checked before the first
    launch (the workspace checks need a device for the scan's size query: tests/test_gpu_local_index.py)"""
    L = lib.lib()
    assert L.dn_conv_index_workspace_bytes(4, 10, 65, 20, 256, 0) == 0                                       # R > 64
    assert L.dn_conv_index_workspace_bytes(4, 10, 3, 20, 0, 0) == 0                                        # no workgroups
    G, N, R, E, wg = 4, 40, 6, 100, 256
    cap = L.dn_close_units_capacity(G, E + N, wg)
    counts, rel, modes, st = (ctypes.c_int64 * 5)(), (ctypes.c_int32 * (R + 1))(), (ctypes.c_int32 * R)(), ctypes.c_int32(0)
    absorb, plan = (ctypes.c_int32 * 4)(), (ctypes.c_int32 * 6)()
    P256 = ctypes.c_void_p(256)

    def call(G=G, host=counts, units=P256, cap=cap, sweep_s=8, sweep=P256, wgrad_rows=4096, kper=0, tcap=0, mt=None):
        return L.dn_conv_index_build_i32(G, N, R, E, P256, P256, P256, P256, P256, 1, 0.75, *([P256] * 10), host, rel, modes,
                                         ctypes.byref(st), P256, P256, P256, P256, P256, absorb, wg, 1, cap, P256, units, P256, P256,
                                         P256, P256, P256, P256, kper, tcap, *([mt] * 8), 32, sweep_s, sweep, sweep, 256, wgrad_rows, 64,
                                         P256, P256, plan, P256, 1 << 30, None)

    tc = L.dn_fold_graph_tiles_multi_capacity(N, 2 * wg)
    for kw, msg in ((dict(G=0), b"bad sizes"), (dict(host=None), b"NULL pointer"), (dict(units=None), b"NULL pointer"),
                    (dict(cap=cap - 1), b"unit table too small"), (dict(sweep=None), b"NULL pointer"),
                    (dict(wgrad_rows=100), b"bad chunk-table arguments"), (dict(kper=2, tcap=tc), b"chunked form needs"),
                    (dict(kper=2, tcap=tc - 1, mt=P256), b"chunked form needs"), (dict(kper=2, tcap=tc, mt=P256), b"unit table too small")):
        assert call(**kw) == -1 and msg in L.dn_last_error(), (kw, L.dn_last_error())


def test_graph_tile_sum_checks_its_arguments(lib):
    L = lib.lib()
    P16 = ctypes.c_void_p(16)
    rc = L.dn_graph_tile_sum_f32(P16, 10, 100, P16, P16, None, 5, P16, 1, 1.0, P16, P16, None)
    assert rc == -1 and b"H must be 64, 128 or 256" in L.dn_last_error()
    rc = L.dn_graph_tile_sum_f32(None, 10, 128, P16, P16, None, 5, P16, 1, 1.0, P16, P16, None)
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
    assert L.dn_graph_tile_sum_f32(None, 10, 128, None, None, None, 0, None, 0, 1.0, None, None, None) == 0      # no tiles


def test_graph_tiles_host_packs_whole_graphs(lib):
    """dn_graph_tiles_host is host code (no GPU): greedy runs of whole graphs with at most max_rows rows, larger graphs left out"""
    import numpy as np
    L = lib.lib()

    def pack(sizes, max_rows):
        npt = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        G = len(sizes)
        buf = np.empty((max(G, 1), 2), dtype=np.int32)
        n = ctypes.c_int64(0)
        assert L.dn_graph_tiles_host(npt.ctypes.data_as(ctypes.c_void_p), G, max_rows, buf.ctypes.data_as(ctypes.c_void_p), max(G, 1),
                                     ctypes.byref(n)) == 0
        return npt, buf[:n.value].copy()

    npt, t = pack([3, 0, 7, 100, 5, 55, 1, 65], 64)
    assert t.tolist() == [[0, 10], [110, 171]]
    rng = np.random.default_rng(0)
    for _ in range(50):
        sizes = rng.integers(0, 150, size=int(rng.integers(0, 60)))
        mr = int(rng.integers(1, 130))
        npt, t = pack(sizes, mr)
        small = (sizes <= mr) & (sizes > 0)
        covered = np.zeros(int(npt[-1]) + 1, dtype=bool)
        for a, b in t:
            assert 0 < b - a <= mr and a in npt and b in npt                 # whole graphs, within the limit
            assert not covered[a:b].any()
            covered[a:b] = True
        want = np.repeat(small, sizes)                                        # exactly the rows of the small graphs
        assert np.array_equal(covered[:int(npt[-1])], want)
        assert all(t[i][1] <= t[i + 1][0] for i in range(len(t) - 1))         # ascending, disjoint
    # greedy: two consecutive tiles cannot be merged unless a large or ... graph sits between them or the sum exceeds the limit
    npt, t = pack([10, 10, 10, 10], 25)
    assert t.tolist() == [[0, 20], [20, 40]]
    rc = L.dn_graph_tiles_host(None, 3, 64, None, 3, ctypes.byref(ctypes.c_int64(0)))
    assert rc == -1 and b"NULL pointer" in L.dn_last_error()
