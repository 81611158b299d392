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
    rc = L.dn_rows_transform_bf16(None, None, 0x7fffffff, None, 100, 100, None, None, 0, None, None, 1, None, None)
    assert rc == -1 and b"unsupported widths" in L.dn_last_error()
    rc = L.dn_conjugate_build_i32(7, 0, 0, 0, 0, *([None] * 13), (ctypes.c_int64 * 2)(), None, 0, None)
    assert rc == -1 and b"bad mode" in L.dn_last_error()
    with pytest.raises(lib.DnHipError):
        lib.check(rc, "dn_conjugate_build_i32")
