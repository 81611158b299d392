"""Host-side operators over the dn_hip C ABI: raw launches + torch.autograd.Function wrappers.

Everything here runs on the GPU through libdn_hip.so; PyTorch only supplies device memory, the
current stream and autograd bookkeeping.  There is deliberately no CPU path.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, lib, ptr, require_gpu, stream_ptr

I32 = torch.int32


class KernelTimer:
    """Optional HIP-event timing of individual kernel launches on the current stream (bench.py's roofline leg).
    Events are recorded on the stream the kernel is launched on; elapsed times are read after a synchronize."""

    def __init__(self):
        self.records = []  # (tag, start_event, end_event)
        self.enabled = True

    def launch(self, tag, fn):
        if not self.enabled:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.records.append((tag, a, b))
        return r

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, a, b in self.records:
            ms = a.elapsed_time(b)
            cnt, tot = out.get(tag, (0, 0.0))
            out[tag] = (cnt + 1, tot + ms)
        return out

    def reset(self):
        self.records = []


kernel_timer = None  # set to a KernelTimer by bench.py


def _suffix(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.bfloat16:
        return "bf16"
    raise _lib.DnHipError("dn_hip kernels support float32 and bfloat16 features, got %s" % t.dtype)


def _i32(t, name):
    if t is not None and t.dtype != I32:
        raise _lib.DnHipError("%s must be int32 on the device (got %s)" % (name, t.dtype))
    return t


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------------------------
# raw launches
# ----------------------------------------------------------------------------------------------
def gather_segsum(x, idx=None, ptr_=None, num_segments=None, scale=None, self_in=None, self_coef=0.0,
                  mean=False, out=None):
    """out[s] = self_coef*self_in[s] + sum_{i in [ptr[s],ptr[s+1])} scale[i] * x[idx[i]]  (dn_gather_segsum_*)."""
    require_gpu(x, idx, ptr_, scale, self_in, out)
    assert x.dim() == 2
    _i32(idx, "idx"), _i32(ptr_, "ptr")
    H = x.shape[1]
    M = idx.numel() if idx is not None else (x.shape[0] if ptr_ is None else None)
    if ptr_ is not None:
        S = ptr_.numel() - 1 if num_segments is None else int(num_segments)
        assert ptr_.numel() >= S + 1
        if M is None:
            M = 0  # contiguous rows: element i is row i; the kernel reads rows [ptr[0], ptr[S])
    else:
        S = M
    if scale is not None:
        assert scale.dtype == torch.float32 and idx is not None and scale.numel() == idx.numel()
    if self_in is not None:
        assert self_in.shape == (S, H) and self_in.dtype == x.dtype
    if out is None:
        out = torch.empty((S, H), dtype=x.dtype, device=x.device)
    else:
        assert out.shape == (S, H) and out.dtype == x.dtype and out.is_contiguous()
    fn = getattr(lib(), "dn_gather_segsum_" + _suffix(x))

    def _launch():
        check(fn(ptr(x), x.shape[0], H, ptr(idx), ptr(scale), ptr(ptr_), S, M, ptr(out), ptr(self_in),
                 float(self_coef), 1 if mean else 0, stream_ptr()), "dn_gather_segsum")

    if kernel_timer is not None:
        kernel_timer.launch("gather_segsum", _launch)
    else:
        _launch()
    return out


def csr_build(key, num_keys):
    """Stable grouping by integer key: returns (ptr [num_keys+1], perm [M]) int32  (dn_csr_build_i32)."""
    require_gpu(key)
    _i32(key, "key")
    M = key.numel()
    p = torch.empty(num_keys + 1, dtype=I32, device=key.device)
    perm = torch.empty(M, dtype=I32, device=key.device)
    nbytes = lib().dn_csr_build_workspace_bytes(M, num_keys)
    if nbytes == 0:
        check(-2, "dn_csr_build_workspace_bytes")
    ws = _ws(nbytes, key.device)
    check(lib().dn_csr_build_i32(ptr(key), M, num_keys, ptr(p), ptr(perm), ptr(ws), ws.numel(), stream_ptr()),
          "dn_csr_build_i32")
    return p, perm


def degrees(src, dst, num_nodes):
    require_gpu(src, dst)
    ind = torch.empty(num_nodes, dtype=I32, device=src.device)
    outd = torch.empty(num_nodes, dtype=I32, device=src.device)
    check(lib().dn_degrees_i32(num_nodes, src.numel(), ptr(_i32(src, "src")), ptr(_i32(dst, "dst")), ptr(ind), ptr(outd),
                               stream_ptr()), "dn_degrees_i32")
    return ind, outd


def edge_norm(mode, self_loop, src, dst, in_deg, out_deg):
    """RGCN norms (rgcn.py:132-165).  mode 'in' | 'both'.  Returns (in_norm [N,1], out_norm [N,1]|None, edge_norm [E])."""
    require_gpu(src, dst, in_deg, out_deg)
    N, E = in_deg.numel(), src.numel()
    in_norm = torch.empty(N, dtype=torch.float32, device=src.device)
    out_norm = torch.empty(N, dtype=torch.float32, device=src.device) if mode == "both" else None
    en = torch.empty(E, dtype=torch.float32, device=src.device)
    check(lib().dn_edge_norm_f32(1 if mode == "in" else 2, 1 if self_loop else 0, N, E, ptr(src), ptr(dst), ptr(in_deg),
                                 ptr(out_deg), ptr(in_norm), ptr(out_norm), ptr(en), stream_ptr()), "dn_edge_norm_f32")
    return in_norm.view(-1, 1), (out_norm.view(-1, 1) if out_norm is not None else None), en


WGRAD_CHUNK_ROWS = 4096


def make_row_chunks(rel_ptr_host, device, chunk_rows=None):
    """Split relation-major rows into chunks for dn_rows_wgrad_bf16: (chunks [C,4] int32, chunk_ptr [R+1] int32)."""
    chunk_rows = chunk_rows or WGRAD_CHUNK_ROWS
    chunks, cptr = [], [0]
    for r in range(len(rel_ptr_host) - 1):
        a, b = rel_ptr_host[r], rel_ptr_host[r + 1]
        while a < b:
            e = min(a + chunk_rows, b)
            chunks.append((r, a, e, 0))
            a = e
        cptr.append(len(chunks))
    ch = torch.tensor(chunks if chunks else [(0, 0, 0, 0)], dtype=I32).reshape(-1, 4)
    return ch.to(device), torch.tensor(cptr, dtype=I32).to(device), len(chunks)


def rows_wgrad(A, G, chunk_table, num_rels, idx_a=None, idx_g=None, out_dtype=None):
    """out[r] = sum_{p in relation r} A[idx_a[p]]^T G[idx_g[p]]  (dn_rows_wgrad_bf16; bf16 inputs, fp32 accumulate)."""
    chunks, chunk_ptr, nchunks = chunk_table
    require_gpu(A, G, idx_a, idx_g, chunks, chunk_ptr)
    assert A.dtype == torch.bfloat16 and G.dtype == torch.bfloat16
    Hi, Ho = A.shape[1], G.shape[1]
    out_dtype = out_dtype or A.dtype
    out = torch.empty((num_rels, Hi, Ho), dtype=out_dtype, device=A.device)
    ws = _ws(lib().dn_rows_wgrad_workspace_bytes(nchunks, Hi, Ho), A.device)

    def _launch():
        check(lib().dn_rows_wgrad_bf16(ptr(A), ptr(idx_a), ptr(G), ptr(idx_g), Hi, Ho, num_rels, ptr(chunks), nchunks,
                                       ptr(chunk_ptr), ptr(out), 1 if out_dtype == torch.float32 else 0, ptr(ws),
                                       ws.numel(), stream_ptr()), "dn_rows_wgrad_bf16")

    if kernel_timer is not None:
        kernel_timer.launch("rows_wgrad", _launch)
    else:
        _launch()
    return out


def wgrad_supported(A, G):
    return (A.dtype == torch.bfloat16 and G.dtype == torch.bfloat16 and A.shape[1] == G.shape[1]
            and A.shape[1] in (64, 128, 256))


# ----------------------------------------------------------------------------------------------
# index structures
# ----------------------------------------------------------------------------------------------
class EdgeIndex:
    """CSR by destination + CSC by source of one batched COO (int32, device resident).
    The one-shot build DGL / torch-scatter hide behind update_all / scatter."""

    def __init__(self, src, dst, num_nodes):
        require_gpu(src, dst)
        self.num_nodes, self.num_edges = int(num_nodes), int(src.numel())
        src, dst = src.to(I32).contiguous(), dst.to(I32).contiguous()
        self.src, self.dst = src, dst
        self.in_ptr, self.in_perm = csr_build(dst, num_nodes)
        self.out_ptr, self.out_perm = csr_build(src, num_nodes)
        # neighbour id lists in segment order (row gather of an int column == index_select plumbing)
        self.src_by_dst = gather_rows_i32(src, self.in_perm)
        self.dst_by_src = gather_rows_i32(dst, self.out_perm)


def gather_rows_i32(values, perm):
    return values.index_select(0, perm.long()) if values.numel() else values.clone()


class RelIndex:
    """(rel, dst)-segment index for aggregate-then-transform RGCN/RGIN (dn_rel_index_build_i32)."""

    def __init__(self, src, dst, etype, num_nodes, num_rels):
        require_gpu(src, dst, etype)
        dev = src.device
        N, R, E = int(num_nodes), int(num_rels), int(src.numel())
        self.num_nodes, self.num_rels, self.num_edges = N, R, E
        src, dst, etype = (t.to(I32).contiguous() for t in (src, dst, etype))
        e32 = lambda n: torch.empty(max(n, 1), dtype=I32, device=dev)  # noqa: E731
        self.perm1, self.src1, seg_ptr, seg_dst = e32(E), e32(E), e32(E + 1), e32(E)
        self.rel_ptr, self.dptr, sperm = e32(R + 1), e32(N + 1), e32(E)
        self.optr, self.operm, self.seg_by_src = e32(N + 1), e32(E), e32(E)
        nbytes = lib().dn_rel_index_workspace_bytes(N, R, E)
        if nbytes == 0:
            check(-2, "dn_rel_index_workspace_bytes")
        ws = _ws(nbytes, dev)
        host_P = ctypes.c_int64(0)
        host_rel = (ctypes.c_int32 * (R + 1))()
        check(lib().dn_rel_index_build_i32(N, R, E, ptr(src), ptr(dst), ptr(etype), ptr(self.perm1), ptr(self.src1),
                                           ptr(seg_ptr), ptr(seg_dst), ptr(self.rel_ptr), ptr(self.dptr), ptr(sperm),
                                           ptr(self.optr), ptr(self.operm), ptr(self.seg_by_src), ctypes.byref(host_P),
                                           host_rel, ptr(ws), ws.numel(), stream_ptr()), "dn_rel_index_build_i32")
        P = int(host_P.value)
        self.num_segments = P
        self.seg_ptr, self.seg_dst, self.sperm = seg_ptr[:P + 1], seg_dst[:P], sperm[:P]
        self.rel_ptr_host = [int(v) for v in host_rel]
        self.perm1, self.src1 = self.perm1[:E], self.src1[:E]
        self.operm, self.seg_by_src = self.operm[:E], self.seg_by_src[:E]
        self.chunk_table = make_row_chunks(self.rel_ptr_host, dev)


# ----------------------------------------------------------------------------------------------
# autograd operators
# ----------------------------------------------------------------------------------------------
class _NeighborSum(torch.autograd.Function):
    """agg[v] = self_coef * x[v] + sum_{e: dst(e)=v} w_e x[src(e)]; backward = same kernel on the CSC."""

    @staticmethod
    def forward(ctx, x, index, self_coef, edge_scale):
        x = x.contiguous()
        ctx.index, ctx.self_coef = index, float(self_coef)
        sc_in = sc_out = None
        if edge_scale is not None:
            sc_in = edge_scale.index_select(0, index.in_perm.long())
            sc_out = edge_scale.index_select(0, index.out_perm.long())
        ctx.sc_out = sc_out
        return gather_segsum(x, index.src_by_dst, index.in_ptr, index.num_nodes, scale=sc_in,
                             self_in=x if self_coef != 0.0 else None, self_coef=self_coef)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        ix = ctx.index
        gx = gather_segsum(g, ix.dst_by_src, ix.out_ptr, ix.num_nodes, scale=ctx.sc_out,
                           self_in=g if ctx.self_coef != 0.0 else None, self_coef=ctx.self_coef)
        return gx, None, None, None


def neighbor_sum(x, index, self_coef=0.0, edge_scale=None):
    """GIN aggregation (gconv.py:212): (self_coef) x_i + sum_{j->i} x_j; edge_scale is a constant weight."""
    return _NeighborSum.apply(x, index, self_coef, edge_scale)


def _grouped_mm(A, W, rel_ptr_host, transpose_w=False):
    """Y[rows of relation r] = A[rows] @ W[r] (or W[r]^T): plain per-relation library GEMMs on
    contiguous row ranges (segments are relation-major)."""
    out_dim = W.shape[1] if transpose_w else W.shape[2]
    Y = torch.empty((A.shape[0], out_dim), dtype=A.dtype, device=A.device)
    for r in range(W.shape[0]):
        a, b = rel_ptr_host[r], rel_ptr_host[r + 1]
        if b > a:
            torch.mm(A[a:b], W[r].t() if transpose_w else W[r], out=Y[a:b])
    return Y


class _RelAggTransform(torch.autograd.Function):
    """agg[v] = sum_r ( sum_{e in r, dst=v} s_e x[src_e] ) W_r   (two gather passes + per-relation GEMMs)."""

    @staticmethod
    def forward(ctx, x, W, index, edge_scale):
        x = x.contiguous()
        W = W.contiguous()
        ix = index
        sc1 = sc_src = None
        if edge_scale is not None:
            sc1 = edge_scale.index_select(0, ix.perm1.long())
            sc_src = edge_scale.index_select(0, ix.operm.long())
        A = gather_segsum(x, ix.src1, ix.seg_ptr, ix.num_segments, scale=sc1)          # [P, in]
        Y = _grouped_mm(A, W, ix.rel_ptr_host)                                          # [P, out]
        agg = gather_segsum(Y, ix.sperm, ix.dptr, ix.num_nodes)                         # [N, out]
        ctx.index, ctx.sc_src = ix, sc_src
        ctx.save_for_backward(A, W)
        return agg

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        A, W = ctx.saved_tensors
        ix = ctx.index
        gY = gather_segsum(g, ix.seg_dst, None)                                         # [P, out] row gather
        gx = gW = None
        if ctx.needs_input_grad[0]:
            gA = _grouped_mm(gY, W, ix.rel_ptr_host, transpose_w=True)                  # [P, in]
            gx = gather_segsum(gA, ix.seg_by_src, ix.optr, ix.num_nodes, scale=ctx.sc_src)
        if ctx.needs_input_grad[1]:
            if wgrad_supported(A, g):
                # MFMA split-K kernel; gathers the g rows itself (idx_g = segment destinations)
                gW = rows_wgrad(A, g, ix.chunk_table, W.shape[0], idx_g=ix.seg_dst, out_dtype=W.dtype)
            else:
                gW = torch.zeros_like(W)
                for r in range(W.shape[0]):
                    a, b = ix.rel_ptr_host[r], ix.rel_ptr_host[r + 1]
                    if b > a:
                        torch.mm(A[a:b].t(), gY[a:b], out=gW[r])
        return gx, gW, None, None


def rel_agg_transform(x, W, index, edge_scale=None):
    """Relation-wise message pass: sum over in-edges of x[src] @ W[etype] (rgin.py:102-120,159)."""
    return _RelAggTransform.apply(x, W, index, edge_scale)


class _SegmentReduce(torch.autograd.Function):
    """Per-graph readout over contiguous rows: sum | mean | max."""

    @staticmethod
    def forward(ctx, x, ptr_, kind):
        x = x.contiguous()
        require_gpu(x, ptr_)
        S, H = ptr_.numel() - 1, x.shape[1]
        out = torch.empty((S, H), dtype=x.dtype, device=x.device)
        sfx = _suffix(x)
        ctx.kind, ctx.rows = kind, x.shape[0]
        if kind == "max":
            arg = torch.empty((S, H), dtype=I32, device=x.device)
            check(getattr(lib(), "dn_segment_max_" + sfx)(ptr(x), H, ptr(ptr_), S, ptr(out), ptr(arg), stream_ptr()),
                  "dn_segment_max")
            ctx.save_for_backward(ptr_, arg)
        else:
            check(getattr(lib(), "dn_segment_%s_%s" % (kind, sfx))(ptr(x), H, ptr(ptr_), S, ptr(out), stream_ptr()),
                  "dn_segment_" + kind)
            ctx.save_for_backward(ptr_)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if ctx.kind == "max":
            ptr_, arg = ctx.saved_tensors
            gin = torch.empty((ctx.rows, g.shape[1]), dtype=g.dtype, device=g.device)
            check(getattr(lib(), "dn_segment_max_bwd_" + _suffix(g))(ptr(g), ptr(arg), g.shape[1], ptr(ptr_),
                                                                     ptr_.numel() - 1, ptr(gin), stream_ptr()),
                  "dn_segment_max_bwd")
            return gin, None, None
        (ptr_,) = ctx.saved_tensors
        S = ptr_.numel() - 1
        # broadcast each graph's row back to its nodes: a row gather keyed by the node's graph id
        seg_of_row = torch.repeat_interleave(torch.arange(S, device=g.device, dtype=I32),
                                             (ptr_[1:] - ptr_[:-1]).long(), output_size=ctx.rows)
        scale = None
        if ctx.kind == "mean":
            cnt = (ptr_[1:] - ptr_[:-1]).to(torch.float32).clamp(min=1.0)
            scale = (1.0 / cnt).index_select(0, seg_of_row.long())
        gin = gather_segsum(g, seg_of_row, None, scale=scale)
        return gin, None, None


def segment_reduce(x, graph_ptr, kind="sum"):
    """global_add_pool / global_mean_pool / global_max_pool over the batch's contiguous node ranges."""
    assert kind in ("sum", "mean", "max")
    return _SegmentReduce.apply(x, graph_ptr, kind)
