"""ctypes binding of libdn_hip.so (the C-ABI HIP library, include/dn_hip.h).

PyTorch is imported FIRST so that the library's `libamdhip64.so.7` dependency resolves to the HIP
runtime PyTorch already loaded (one runtime per process: device pointers and streams are only
meaningful inside the runtime that created them; `dn_runtime_probe` double-checks at first use).
There is no CPU fallback: if the library is missing or a tensor is not on a GPU, calls raise.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DN_HIP_LIB: another build of the same library (A/B runs of two kernel versions on one box, tools/ab_lib.sh); no fallback either way
LIB_PATH = os.environ.get("DN_HIP_LIB") or os.path.join(_HERE, "libdn_hip.so")

c_i32, c_i64, c_f32, c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
P = ctypes.c_void_p

class WgradJob(ctypes.Structure):
    """dn_wgrad_job of include/dn_hip.h (one weight gradient of a dn_rows_wgrad_multi_bf16 launch)."""
    _fields_ = [("A", P), ("A2", P), ("idx_a", P), ("G", P), ("G2", P), ("idx_g", P), ("mask_a_bits", P), ("na1", c_i32), ("ng1", c_i32),
                ("colsum_of", c_i32), ("first_rel", c_i32), ("row0", c_i32), ("act_slope", c_f32)]


# name -> (restype, argtypes); mirrors include/dn_hip.h one to one
_SIGS = {
    "dn_version": (ctypes.c_int, []),
    "dn_last_error": (ctypes.c_char_p, []),
    "dn_runtime_probe": (ctypes.c_int, [P]),
    "dn_gather_segsum_f32": (ctypes.c_int, [P, c_i64, c_i32, P, P, P, c_i64, c_i64, P, P, c_f32, c_i32, P]),
    "dn_gather_segsum_bf16": (ctypes.c_int, [P, c_i64, c_i32, P, P, P, c_i64, c_i64, P, P, c_f32, c_i32, P]),
    "dn_segment_sum_f32": (ctypes.c_int, [P, c_i32, P, c_i64, P, P]),
    "dn_segment_sum_bf16": (ctypes.c_int, [P, c_i32, P, c_i64, P, P]),
    "dn_segment_mean_f32": (ctypes.c_int, [P, c_i32, P, c_i64, P, P]),
    "dn_segment_mean_bf16": (ctypes.c_int, [P, c_i32, P, c_i64, P, P]),
    "dn_segment_max_f32": (ctypes.c_int, [P, c_i32, P, c_i64, P, P, P]),
    "dn_segment_max_bf16": (ctypes.c_int, [P, c_i32, P, c_i64, P, P, P]),
    "dn_segment_max_bwd_f32": (ctypes.c_int, [P, P, c_i32, P, c_i64, P, P]),
    "dn_segment_max_bwd_bf16": (ctypes.c_int, [P, P, c_i32, P, c_i64, P, P]),
    "dn_edge_dot_f32": (ctypes.c_int, [P, P, P, P, c_i32, c_i64, P, P]),
    "dn_edge_dot_bf16": (ctypes.c_int, [P, P, P, P, c_i32, c_i64, P, P]),
    "dn_gather_segmax_f32": (ctypes.c_int, [P, P, P, c_i64, c_i32, P, P, P]),
    "dn_gather_segmax_bf16": (ctypes.c_int, [P, P, P, c_i64, c_i32, P, P, P]),
    "dn_gather_segmax_bwd_f32": (ctypes.c_int, [P, P, P, P, P, c_i64, c_i32, P, P]),
    "dn_gather_segmax_bwd_bf16": (ctypes.c_int, [P, P, P, P, P, c_i64, c_i32, P, P]),
    "dn_graph_tile_sum_f32": (ctypes.c_int, [P, c_i64, c_i32, P, P, P, c_i64, P, c_i64, c_f32, P, P, P]),
    "dn_gather_rows_sum_f32": (ctypes.c_int, [P, c_i32, P, P, P, c_i32, c_i64, c_i32, c_f32, P, P]),
    "dn_graph_tiles_host": (ctypes.c_int, [P, c_i64, c_i32, P, c_i64, ctypes.POINTER(c_i64)]),
    "dn_csr_build_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "dn_csr_build_i32": (ctypes.c_int, [P, c_i64, c_i64, P, P, P, c_sz, P]),
    "dn_dummy_augment_gc_i32": (ctypes.c_int, [c_i64, c_i64, c_i64] + [P] * 16 + [P]),
    "dn_dummy_augment_si_i32": (ctypes.c_int, [c_i64, c_i64, c_i64] + [P] * 9 + [c_i32] * 4 + [P] * 11 + [P]),
    "dn_conjugate_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "dn_conjugate_count_i32": (ctypes.c_int, [c_i64, c_i64, P, P, ctypes.POINTER(c_i64), P, c_sz, P]),
    "dn_conjugate_build_i32": (ctypes.c_int, [c_i32, c_i64, c_i64, c_i64, c_i64] + [P] * 7 + [P] * 6 +
                               [ctypes.POINTER(c_i64), P, c_sz, P]),
    "dn_rel_index_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "dn_rel_index_build_i32": (ctypes.c_int, [c_i64, c_i64, c_i64] + [P] * 3 + [P] * 10 +
                               [ctypes.POINTER(c_i64), ctypes.POINTER(c_i32), P, c_sz, P]),
    "dn_row_index_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "dn_row_index_build_i32": (ctypes.c_int, [c_i64, c_i64, c_i64, P, P, P, c_i32, c_f32] + [P] * 10 +
                               [ctypes.POINTER(c_i64), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), P, c_sz, P]),
    "dn_row_index_local_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "dn_row_index_build_local_i32": (ctypes.c_int, [c_i64, c_i64, c_i64, c_i64, P, P, P, P, P, c_i32, c_f32] + [P] * 10 +
                                     [ctypes.POINTER(c_i64), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                      ctypes.POINTER(c_i32), P, P, P, P, P, ctypes.POINTER(c_i32), P, c_sz, P]),
    "dn_rows_wgrad_multi_bf16": (ctypes.c_int, [ctypes.POINTER(WgradJob), c_i32, c_i32, c_i64, P, c_i64, P, P, c_i32, P, P, P, c_sz, P]),
    "dn_rows_wgrad_multi_f32": (ctypes.c_int, [ctypes.POINTER(WgradJob), c_i32, c_i32, c_i64, P, c_i64, P, P, P, P, c_sz, P]),
    "dn_rows_chain2_f32": (ctypes.c_int, [P, c_i32, P, P, c_i32, P, P, P, P, c_i32, c_i64, P, P, c_i32, c_f32, P, P, P]),
    "dn_layer_graphs_fwd_bf16": (ctypes.c_int, [P, c_i32, P, P, P, c_i32, P, P, P, P, c_f32, P, P, P, P, P, c_i64, c_i64, P, P, P, P, P, P, P,
                                                P, P, P]),
    "dn_layer_graphs_bwd_bf16": (ctypes.c_int, [P, c_i32, P, P, c_i32, P, P, c_f32, P, P, P, P, P, P, P, c_i64, c_i64, P, P, P, P, P, P, P,
                                                P]),
    "dn_conv_graphs_max_nodes": (c_i32, []),
    "dn_conv_graphs_max_edges": (c_i32, []),
    "dn_conv_graphs_bf16": (ctypes.c_int, [P, c_i32, P, c_i32, P, P, c_i32, P, P, P, P, P, c_i64, c_i64, P, P, P, P, P, P]),
    "dn_conv_index_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64, c_i32, c_i64]),
    "dn_conv_index_build_i32": (ctypes.c_int, [c_i64, c_i64, c_i64, c_i64, P, P, P, P, P, c_i32, c_f32] + [P] * 10 +
                                [ctypes.POINTER(c_i64), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                 ctypes.POINTER(c_i32), P, P, P, P, P, ctypes.POINTER(c_i32)] +
                                [c_i32, c_i32, c_i64] + [P] * 8 + [c_i32, c_i64] + [P] * 8 + [c_i32, c_i32, P, P] + [c_i32, c_i32, c_i64, P, P] +
                                [ctypes.POINTER(c_i32), P, c_sz, P]),
    "dn_row_tables_build_i32": (ctypes.c_int, [c_i32, P, c_i32, c_i64, P, P, ctypes.c_uint64, P]),
    "dn_sweep_tables_build_i32": (ctypes.c_int, [c_i32, P, P, P, c_i64, c_i32, c_i32, ctypes.c_uint64, P, P, P]),
    "dn_slot_table_build_i32": (ctypes.c_int, [c_i64, c_i32, c_i32, P, P, c_i32, c_i32, P, P, P, P]),
    "dn_fold_tables_build_async_i32": (ctypes.c_int, [c_i64, c_i64, P, P, P, P, P, P, c_sz, P]),
    "dn_rows_wgrad_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32]),
    "dn_rows_wgrad_bf16": (ctypes.c_int, [P, P, c_i32, P, P, P, c_i32, P, c_i32, c_i32, c_i64, P, c_i64, P, P, c_i32,
                                          c_i32, P, P, P, P, P, c_f32, P, c_sz, P]),
    "dn_rows_transform_bf16": (ctypes.c_int, [P, P, c_i32, P, c_i32, c_i32, P, P, c_i32, P, P, c_i64, P, c_i32, c_f32, P]),
    "dn_relu_bwd_bf16": (ctypes.c_int, [P, P, P, c_i64, c_f32, P]),
    "dn_rows_chain2_bf16": (ctypes.c_int, [P, c_i32, P, P, c_i32, P, P, P, P, c_i32, c_i64, P, P, P, P, c_i32, c_f32, P]),
    "dn_rows_selfsum_bf16": (ctypes.c_int, [P, c_i32, P, P, P, P, c_i32, P, c_i32, c_i64, P, P, P, c_i32, P, P, c_i32, c_i32, c_i32, P]),
    "dn_overflow_rows_add_bf16": (ctypes.c_int, [P, c_i32, P, c_i32, c_i64, P, P, c_i32, c_i32, c_i32, P, P]),
    "dn_close_units_capacity": (c_i64, [c_i64, c_i64, c_i32]),
    "dn_close_units_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "dn_close_units_build_i32": (ctypes.c_int, [c_i64, c_i32, c_i32, P, c_i64, c_i32, c_i32, P, P, c_i64, c_i32, c_i32, P, P, P, c_i64, P,
                                                P, P, P, c_i32, P, c_sz, P]),
    "dn_fold_graph_tiles_build_i32": (ctypes.c_int, [c_i64, c_i64, P, P, P, P, P, P, P]),
    "dn_fold_graph_tiles_multi_capacity": (c_i64, [c_i64, c_i32]),
    "dn_fold_graph_tiles_multi_build_i32": (ctypes.c_int, [c_i64, c_i64, P, P, P, c_i32, P, P, P, P, c_i64, P, P]),
    "dn_rows_close_bf16": (ctypes.c_int, [P, c_i32, P, c_i32, P, P, P, P, c_i32, P, P, c_i64, P, P, P, P, P, P, P]),
    "dn_bdd_compose": (ctypes.c_int, [P, c_i64, c_i32, c_i32, c_i32, c_i32, P, P]),
    "dn_bdd_extract": (ctypes.c_int, [P, c_i64, c_i32, c_i32, c_i32, c_i32, P, P]),
    "dn_fold_tables_workspace_bytes": (c_sz, [c_i64]),
    "dn_fold_tables_build_i32": (ctypes.c_int, [c_i64, c_i64, P, P, P, P, ctypes.POINTER(ctypes.c_int32), P, c_sz, P]),
    "dn_fold_tail_bf16": (ctypes.c_int, [P, P, c_i64, c_i32, P, P, P, P, c_i32, P]),
    "dn_rows_transform_f32": (ctypes.c_int, [P, P, c_i32, P, c_i32, c_i32, P, P, c_i32, P, P, c_i64, P, c_i32, c_f32, P, c_i32, c_i32, c_i32, P]),
    "dn_rows_wgrad_f32": (ctypes.c_int, [P, P, c_i32, P, P, P, c_i32, P, c_i32, c_i32, c_i64, P, c_i64, P, P, c_i32, P, P, P,
                                         c_i32, c_f32, P, c_sz, P]),
    "dn_relu_bwd_f32": (ctypes.c_int, [P, P, P, c_i64, c_f32, P]),
    "dn_rows_gemm_f32": (ctypes.c_int, [P, P, P, c_i32, c_i32, c_i32, P, c_i64, P, P]),
    "dn_rows_gemm_bf16": (ctypes.c_int, [P, P, P, c_i32, c_i32, c_i32, P, c_i64, P, P]),
    "dn_rows_wgrad_any_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32]),
    "dn_rows_wgrad_any_f32": (ctypes.c_int, [P, P, c_i32, c_i32, c_i64, P, c_i64, P, P, P, P, c_sz, P]),
    "dn_rows_wgrad_any_bf16": (ctypes.c_int, [P, P, c_i32, c_i32, c_i64, P, c_i64, P, P, P, P, c_sz, P]),
    "dn_batchnorm_rows_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "dn_batchnorm_rows_f32": (ctypes.c_int, [P, c_i64, c_i32, P, P, c_f32, P, P, P, P, P, P, c_f32, c_i32, P, P, c_sz, P]),
    "dn_batchnorm_rows_bf16": (ctypes.c_int, [P, c_i64, c_i32, P, P, c_f32, P, P, P, P, P, P, c_f32, c_i32, P, P, c_sz, P]),
    "dn_batchnorm_rows_bwd_f32": (ctypes.c_int, [P, P, c_i64, c_i32, P, P, P, P, c_i32, P, P, P, P, c_sz, P]),
    "dn_batchnorm_rows_bwd_bf16": (ctypes.c_int, [P, P, c_i64, c_i32, P, P, P, P, c_i32, P, P, P, P, c_sz, P]),
    "dn_edge_norm_f32": (ctypes.c_int, [c_i32, c_i32, c_i64, c_i64] + [P] * 7 + [P]),
    "dn_degrees_i32": (ctypes.c_int, [c_i64, c_i64, P, P, P, P, P]),
}

_lib = None
_probed = False


class DnHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DnHipError(
                "libdn_hip.so not found at %s -- build it with `python -m dummynode4graphlearning_amd.csrc.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def exported_symbols():
    return sorted(_SIGS.keys())


def check(rc, what=""):
    if rc != 0:
        msg = lib().dn_last_error()
        raise DnHipError("%s failed (rc=%d): %s" % (what or "dn_hip call", rc, msg.decode() if msg else ""))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (the raw-handle query is ~10x cheaper than building a
    torch.cuda.Stream object per launch, which showed up in eager-mode profiles of small batches)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    """The product path runs on the GPU only; fail loudly otherwise."""
    global _probed
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise DnHipError("dummynode4graphlearning_amd kernels need GPU tensors (got device %s); "
                             "there is no CPU fallback" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise DnHipError("tensors on different devices: %s vs %s" % (t.device, dev))
        if not t.is_contiguous():
            raise DnHipError("non-contiguous tensor passed to a dn_hip kernel")
    if not _probed and dev is not None:
        for t in tensors:
            if t is not None and t.numel() > 0:
                check(lib().dn_runtime_probe(ptr(t)), "dn_runtime_probe")
                _probed = True
                break
    return dev


def source_digest():
    """sha256 (first 16 hex digits) over the sources that decide which launches a step makes and what they do: the HIP kernels,
    the C header and ops.py.  profiles/*_traffic.json records it; bench.py only quotes a committed traffic measurement whose
    digest equals the running tree's (a stale file must not label a later code change)."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")))
    files += [os.path.join(os.path.dirname(here), "include", "dn_hip.h"), os.path.join(here, "ops.py")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
