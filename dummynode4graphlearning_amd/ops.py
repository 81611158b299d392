"""Host-side operators over the dn_hip C ABI: raw launches + torch.autograd.Function wrappers.

Everything here runs on the GPU through libdn_hip.so; PyTorch only supplies device memory, the
current stream and autograd bookkeeping.  There is deliberately no CPU path.
"""
import ctypes
import os as _os
import threading as _threading

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


def graph_tiles(node_ptr, max_rows=64):
    """Greedy runs of whole graphs with at most max_rows rows each: (tiles [T, 2] int32 on the device, covered rows, rest) where
    rest = the rows of graphs larger than max_rows as a sorted int64 tensor (they keep the plain gather)."""
    npt = node_ptr.detach().cpu().tolist()
    tiles, rest, beg = [], [], None
    for g in range(len(npt) - 1):
        a, b = npt[g], npt[g + 1]
        if b - a > max_rows:
            if beg is not None:
                tiles.append((beg, a)); beg = None
            rest.append((a, b))
            continue
        if beg is None:
            beg = a
        elif b - beg > max_rows:
            tiles.append((beg, a)); beg = a
    if beg is not None:
        tiles.append((beg, npt[-1]))
    tiles = [(a, b) for a, b in tiles if b > a]
    t = torch.tensor(tiles, dtype=I32, device=node_ptr.device).reshape(-1, 2)
    covered = sum(b - a for a, b in tiles)
    r = (torch.cat([torch.arange(a, b) for a, b in rest]) if rest else torch.zeros(0, dtype=torch.long)).to(node_ptr.device)
    return t, covered, r


def graph_tile_records(tiles, ptr_):
    """[T, 4] int32 records {first row, end row, ptr[first row], ptr[end row]} of graph_tiles' row ranges for one CSR."""
    return torch.cat([tiles, ptr_[tiles.long()].to(I32)], 1).contiguous()


def gather_rows_sum(x, idx, ptr_, rows, workgroup_per_row, self_coef, out):
    """out[s] = self_coef * x[s] + sum of x[idx[i]] over s's list, for the listed rows only (dn_gather_rows_sum_f32)."""
    require_gpu(x, idx, ptr_, rows, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous() and rows.dtype == I32 and rows.is_contiguous()
    records = rows.dim() == 2                                       # [n, 4] records {row, ptr[row], ptr[row + 1], 0}
    assert not records or rows.shape[1] == 4

    def _launch():
        check(lib().dn_gather_rows_sum_f32(ptr(x), int(x.shape[1]), ptr(ptr_), ptr(idx), ptr(rows), 1 if records else 0,
                                           int(rows.shape[0]), 1 if workgroup_per_row else 0, float(self_coef), ptr(out),
                                           stream_ptr()), "dn_gather_rows_sum_f32")
    if kernel_timer is not None:
        kernel_timer.launch("gather_rows_sum", _launch)
    else:
        _launch()
    return out


def graph_tile_sum(x, idx, ptr_, tiles, self_coef=0.0, out=None, seg=None, bad=None):
    """out[v] = self_coef * x[v] + sum of x[idx[i]] over v's list, for the rows of `tiles` only (dn_graph_tile_sum_f32: the tile's
    adjacency as a dense bf16 matrix, the rows as three bf16 planes, matrix cores).  Rows outside the tiles are left untouched.
    tiles: graph_tiles' [T, 2] row ranges or graph_tile_records' [T, 4]; seg (optional): the row of every index entry."""
    require_gpu(x, idx, ptr_, tiles)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] in (64, 128, 256)
    assert idx.dtype == I32 and ptr_.dtype == I32 and tiles.dtype == I32 and tiles.is_contiguous()
    if tiles.shape[1] == 2:
        tiles = graph_tile_records(tiles, ptr_)
    assert seg is None or (seg.dtype == I32 and seg.numel() == idx.numel())
    if out is None:
        out = torch.empty_like(x)
    if bad is None:                                                 # (a caller that launches repeatedly keeps ONE flag: the kernel only ORs)
        bad = torch.zeros(1, dtype=I32, device=x.device)

    def _launch():
        check(lib().dn_graph_tile_sum_f32(ptr(x), int(x.shape[0]), int(x.shape[1]), ptr(ptr_), ptr(idx),
                                          ptr(seg) if seg is not None else None, int(idx.numel()), ptr(tiles),
                                          int(tiles.shape[0]), float(self_coef), ptr(out), ptr(bad), stream_ptr()),
              "dn_graph_tile_sum_f32")
    if kernel_timer is not None:
        kernel_timer.launch("graph_tile_sum", _launch)
    else:
        _launch()
    return out, bad


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


WGRAD_CHUNK_ROWS = int(_os.environ.get("DN_WGRAD_CHUNK", "4096"))            # step of the chunk tables cut without a look at the relation sizes
# Chunk tables cut WITH the relation sizes (wgrad_chunk_rows, the one-call index, the dense tables) take the smallest chunk for which
# the launch is ONE round of workgroups, up to this many rows (config 5: 775 chunks of 4,096 rows = 3.03 rounds of 256 workgroups and
# 203 MB of partial products became 245 chunks of 12,864 rows and 64 MB: conv weight gradient + reduce 534 -> 473 us, round 6)
WGRAD_CHUNK_CAP = max(WGRAD_CHUNK_ROWS, int(_os.environ.get("DN_WGRAD_CHUNK_CAP", "65536")))
# fp32 matrix products: False = 3-term bf16 split on the fast MFMA path (1e-5-level agreement with exact f32, inside the
# reference's 1e-4 bar), True = exact f32 MFMA (1/16 of the bf16 rate; the checker).  F32_EXACT is the PROCESS default (read once
# from the environment; assignable); `with f32_exact(...)` overrides it for the calling THREAD only, and every autograd function
# below records the mode its forward ran in and runs its backward in the same mode -- whatever thread autograd uses for it and
# whether or not the `with` block has been left by then.
F32_EXACT = _os.environ.get("DN_F32_EXACT", "0") == "1"
_f32_tls = _threading.local()


def f32_mode():
    """The fp32 arithmetic a kernel launched NOW from this thread uses: the thread's f32_exact override, else ops.F32_EXACT."""
    m = getattr(_f32_tls, "mode", None)
    return F32_EXACT if m is None else m


class f32_exact:
    """Context manager: run the fp32 matrix kernels inside it on the exact-f32 MFMA (True) or on the bf16 split (False).
    Thread-local; backward passes of work recorded inside it keep the mode (see F32_EXACT)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.old = getattr(_f32_tls, "mode", None)
        _f32_tls.mode = self.on
        return self

    def __exit__(self, *exc):
        _f32_tls.mode = self.old
        return False


def _backward_in_forward_mode(backward):
    """Decorator of an autograd backward: run it in the fp32 mode its forward recorded (ctx.f32_mode)."""
    def wrapped(ctx, *grads):
        with f32_exact(ctx.f32_mode):
            return backward(ctx, *grads)
    wrapped.__doc__ = backward.__doc__
    return wrapped


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


INT32_MAX = 0x7fffffff


def rows_wgrad(A, G, chunk_table, num_rels, idx_a=None, idx_g=None, out_dtype=None, A2=None, G2=None, colsum_of=0,
               mask_a=None, a_out=None, mask_a_bits=None, colsum_lp=False, slope=0.0, colsum_rel=None):
    """out[r] = sum_{p in relation r} Acat[idx_a[p]]^T Gcat[idx_g[p]]  (dn_rows_wgrad_bf16; bf16 in, fp32 accumulate).
    Acat = [A; A2], Gcat = [G; G2] (virtual concatenations).  colsum_of = 1|2 additionally returns the fp32 per-relation
    column sums [R, H] of operand A|G (the bias gradient); colsum_lp: return them in out's dtype instead (bf16 output: written
    by the reduce launch itself, so the bias gradient needs no cast launch); colsum_rel (bf16): the column sums of that relation's
    rows only, the other rows of the result are zeros."""
    chunks, chunk_ptr, nchunks = chunk_table
    require_gpu(A, G, idx_a, idx_g, chunks, chunk_ptr, A2, G2, mask_a, a_out, mask_a_bits)
    assert A.dtype == G.dtype and A.dtype in (torch.bfloat16, torch.float32)
    is_f32 = A.dtype == torch.float32
    assert mask_a_bits is None or (not is_f32 and mask_a_bits.dtype == torch.uint8
                                   and mask_a_bits.shape == (A.shape[0], A.shape[1] // 8))
    if is_f32:
        out_dtype = torch.float32
    Hi, Ho = A.shape[1], G.shape[1]
    out_dtype = out_dtype or A.dtype
    out = torch.empty((num_rels, Hi, Ho), dtype=out_dtype, device=A.device)
    ws = _ws(lib().dn_rows_wgrad_workspace_bytes(nchunks, Hi, Ho), A.device)
    colsum = torch.empty((num_rels, Hi), dtype=torch.float32, device=A.device) if colsum_of else None
    # the column sums again in the output dtype (the bias gradient as the parameter wants it): written by the reduce launch
    colsum_lp = (torch.empty((num_rels, Hi), dtype=out_dtype, device=A.device)
                 if (colsum_lp and colsum_of and not is_f32 and out_dtype != torch.float32) else None)

    def _launch():
        na1 = A.shape[0] if A2 is not None else INT32_MAX
        ng1 = G.shape[0] if G2 is not None else INT32_MAX
        if is_f32:
            check(lib().dn_rows_wgrad_f32(ptr(A), ptr(A2), na1, ptr(idx_a), ptr(G), ptr(G2), ng1, ptr(idx_g), Hi, Ho, num_rels,
                                          ptr(chunks), nchunks, ptr(chunk_ptr), ptr(out), int(colsum_of), ptr(colsum),
                                          ptr(mask_a), ptr(a_out), 1 if f32_mode() else 0, float(slope), ptr(ws), ws.numel(), stream_ptr()),
                  "dn_rows_wgrad_f32")
        else:
            check(lib().dn_rows_wgrad_bf16(ptr(A), ptr(A2), na1, ptr(idx_a), ptr(G), ptr(G2), ng1, ptr(idx_g), Hi, Ho, num_rels,
                                           ptr(chunks), nchunks, ptr(chunk_ptr), ptr(out),
                                           1 if out_dtype == torch.float32 else 0,
                                           int(colsum_of) | (((int(colsum_rel) + 1) << 8) if (colsum_of and colsum_rel is not None) else 0),
                                           ptr(colsum), ptr(mask_a),
                                           ptr(a_out), ptr(mask_a_bits), ptr(colsum_lp), float(slope), ptr(ws), ws.numel(), stream_ptr()),
                  "dn_rows_wgrad_bf16")

    if kernel_timer is not None:
        kernel_timer.launch("rows_wgrad", _launch)
    else:
        _launch()
    if colsum_of:
        return out, (colsum_lp if colsum_lp is not None else colsum)
    return out


def rows_wgrad_multi(jobs, chunk_table, num_rels, H, out_dtype):
    """Several weight gradients in ONE launch + ONE reduce (dn_rows_wgrad_multi_bf16 / _f32 on the bf16 split, H = 64 / 128).  jobs: dicts with A, G
    (and optionally A2, G2, idx_a, idx_g, mask_a_bits, colsum_of, slope), first_rel, row0 -- relations numbered through, rows laid
    end to end in one virtual row space that chunk_table covers.  -> (out [num_rels, H, H], colsum [num_rels, H]) in out_dtype."""
    chunks, chunk_ptr, nchunks = chunk_table
    dev = jobs[0]["A"].device
    arr = (_lib.WgradJob * len(jobs))()
    keep = []
    for k, jb in enumerate(jobs):
        A, G = jb["A"], jb["G"]
        require_gpu(A, G, jb.get("A2"), jb.get("G2"), jb.get("idx_a"), jb.get("idx_g"), jb.get("mask_a_bits"))
        assert A.dtype == G.dtype == jobs[0]["A"].dtype and A.dtype in (torch.bfloat16, torch.float32)
        assert A.shape[1] == G.shape[1] == H and A.is_contiguous() and G.is_contiguous()
        keep.append(jb)
        pv = lambda t: (t.data_ptr() if t is not None else None)  # noqa: E731
        arr[k].A, arr[k].A2, arr[k].idx_a = pv(A), pv(jb.get("A2")), pv(jb.get("idx_a"))
        arr[k].G, arr[k].G2, arr[k].idx_g = pv(G), pv(jb.get("G2")), pv(jb.get("idx_g"))
        arr[k].mask_a_bits = pv(jb.get("mask_a_bits"))
        arr[k].na1 = A.shape[0] if jb.get("A2") is not None else INT32_MAX
        arr[k].ng1 = G.shape[0] if jb.get("G2") is not None else INT32_MAX
        arr[k].colsum_of, arr[k].first_rel, arr[k].row0 = int(jb.get("colsum_of", 0)), int(jb["first_rel"]), int(jb["row0"])
        arr[k].act_slope = float(jb.get("slope", 0.0))
    is_f32 = jobs[0]["A"].dtype == torch.float32
    if is_f32:                                                              # (a job's mask is then the saved activation itself: float [rows, H])
        assert out_dtype == torch.float32 and not f32_mode()
        assert all(jb.get("mask_a_bits") is None or (jb["mask_a_bits"].dtype == torch.float32 and jb["mask_a_bits"].shape == jb["A"].shape
                                                     and jb["mask_a_bits"].is_contiguous()) for jb in jobs)
    out = torch.empty((num_rels, H, H), dtype=out_dtype, device=dev)
    colsum = torch.empty((num_rels, H), dtype=torch.float32, device=dev)
    colsum_lp = torch.empty((num_rels, H), dtype=out_dtype, device=dev) if out_dtype != torch.float32 else None
    ws = _ws(lib().dn_rows_wgrad_workspace_bytes(nchunks, H, H), dev)

    def _launch():
        if is_f32:                                                          # fp32 rows on the 3-term bf16 split
            check(lib().dn_rows_wgrad_multi_f32(arr, len(jobs), H, num_rels, ptr(chunks), nchunks, ptr(chunk_ptr), ptr(out), ptr(colsum),
                                                ptr(ws), ws.numel(), stream_ptr()), "dn_rows_wgrad_multi_f32")
            return
        check(lib().dn_rows_wgrad_multi_bf16(arr, len(jobs), H, num_rels, ptr(chunks), nchunks, ptr(chunk_ptr), ptr(out),
                                             1 if out_dtype == torch.float32 else 0, ptr(colsum), ptr(colsum_lp), ptr(ws), ws.numel(),
                                             stream_ptr()), "dn_rows_wgrad_multi_bf16")
    if kernel_timer is not None:
        kernel_timer.launch("rows_wgrad_multi", _launch)
    else:
        _launch()
    return out, (colsum_lp if colsum_lp is not None else colsum)


# split-K chunks of the ONE weight-gradient launch of an H = 64 / 128 layer (_RginLayerSmallFn, _RginLayerF32Fn): two rounds' worth of
# workgroups -- a workgroup walks its chunk tile by tile, a dependent chain (22 tiles of 32 rows at config 3 with 256 chunks), and two
# of these small workgroups share a CU: config 3 fp32 0.105 -> 0.100 ms, bf16 0.069 -> 0.068 (1,024: the same)
_SMALL_WG = 512


def wgrad_chunk_rows(rel_ptr_host, workgroups=256):
    """Rows per split-K chunk of the weight gradient over relation-major rows: the smallest multiple of 64 (>= 256) for which the
    chunks of all relations -- every relation ends in a partial chunk -- fit ONE round of `workgroups`; batches too large for
    that (over 256 x WGRAD_CHUNK_CAP rows) keep WGRAD_CHUNK_CAP and run several rounds.  (A 1/8 shard of config 5 cut by rows / 256
    alone made 270 chunks: a second round for 14 workgroups, 138 us instead of 80.)"""
    sizes = [int(b) - int(a) for a, b in zip(rel_ptr_host[:-1], rel_ptr_host[1:]) if int(b) > int(a)]
    total = sum(sizes)
    if total == 0:
        return 256
    c = max(256, -(-total // workgroups // 64) * 64)
    while c < WGRAD_CHUNK_CAP and sum(-(-n // c) for n in sizes) > workgroups:
        c += 64
    return min(c, WGRAD_CHUNK_CAP)


def build_row_tables(rel_ptr_dev, num_rels, num_rows, step, want_ptr=False, skip_mask=0):
    """Tile (step = 32) / chunk tables of relation-major rows built ON THE DEVICE (dn_row_tables_build_i32): returns
    (table [M, 4] int32, M) or (table, piece_ptr [num_rels + 1], M) with M = the upper bound rows / step + num_rels --
    unused entries are empty pieces, which every consumer skips.  skip_mask: bit r leaves relation r out."""
    require_gpu(rel_ptr_dev)
    _i32(rel_ptr_dev, "rel_ptr")
    dev = rel_ptr_dev.device
    M = int(num_rows) // int(step) + int(num_rels) + 1
    table = torch.empty((M, 4), dtype=I32, device=dev)
    pptr = torch.empty(int(num_rels) + 1, dtype=I32, device=dev) if want_ptr else None
    check(lib().dn_row_tables_build_i32(int(num_rels), ptr(rel_ptr_dev), int(step), M, ptr(table), ptr(pptr), int(skip_mask),
                                        stream_ptr()), "dn_row_tables_build_i32")
    return (table, pptr, M) if want_ptr else (table, M)


SWEEP_ENABLED = _os.environ.get("DN_SWEEP", "1") != "0"
SWEEP_WG_PER_GROUP = 32                  # dn_rows_transform_bf16 at H = 256: 256 persistent workgroups = 8 XCDs x 32 CUs
SWEEP_MIN_TILES_PER_WG = 32              # smaller launches fit the L2s anyway


def build_sweep_tables(rel_ptr_dev, num_rels, row_in, row_out, num_nodes, num_rows, skip_mask=0, wg_per_group=SWEEP_WG_PER_GROUP,
                       want_info=False):
    """L2-blocked tile order of relation-major rows for the persistent transform launch (dn_sweep_tables_build_i32; see
    include/dn_hip.h): (table [8 * wg_per_group * S, 4] int32, 8 * wg_per_group * S) -- the pair rows_transform takes.  S is
    sized from what the host knows (rows / 32 + one partial tile per group and relation, 10 % head-room for uneven groups and the
    pure workgroups' quota); when
    a group needs more the builder writes the plain order into the same table (still valid), no read-back."""
    require_gpu(rel_ptr_dev, row_in, row_out)
    _i32(rel_ptr_dev, "rel_ptr"), _i32(row_in, "row_in"), _i32(row_out, "row_out")
    G = 8 * int(wg_per_group)
    S = int(1.10 * (int(num_rows) // 32 + 8 * int(num_rels)) / G) + 4     # (head-room: uneven groups, the pure workgroups' larger quota)
    table = torch.empty((G * S, 4), dtype=I32, device=rel_ptr_dev.device)
    info = torch.empty(2, dtype=I32, device=rel_ptr_dev.device) if want_info else None
    check(lib().dn_sweep_tables_build_i32(int(num_rels), ptr(rel_ptr_dev), ptr(row_in), ptr(row_out), int(num_nodes),
                                          int(wg_per_group), S, int(skip_mask), ptr(table), ptr(info), stream_ptr()),
          "dn_sweep_tables_build_i32")
    return ((table, G * S), info) if want_info else (table, G * S)


def make_row_tiles(rel_ptr_host, device, tile_rows=32):
    """Tile table for dn_rows_transform_bf16: [T,4] int32 rows {rel, beg, end, 0}, tiles never cross relations."""
    import numpy as np
    parts = []
    for r in range(len(rel_ptr_host) - 1):
        a, b = rel_ptr_host[r], rel_ptr_host[r + 1]
        if b > a:
            beg = np.arange(a, b, tile_rows, dtype=np.int64)
            parts.append(np.stack([np.full_like(beg, r), beg, np.minimum(beg + tile_rows, b), np.zeros_like(beg)], 1))
    tl = np.concatenate(parts, 0) if parts else np.zeros((0, 4), dtype=np.int64)
    return torch.from_numpy(tl.astype(np.int32)).to(device), int(tl.shape[0])


def rows_transform(X, Wn, tile_table, num_rows, idx=None, X2=None, bias=None, relu=False, mask_pos=None, tag="dense",
                   out=None, w_kn=False, slope=0.0, W_loop=None, loop_rel=-1, bias_rel=-1):
    """Y[p] = epi(Xcat[idx[p]] @ Wn[rel(p)]^T), zeroed where mask_pos[p] <= 0  (dn_rows_transform_bf16 / _f32).
    w_kn: Wn[r] is given [in][out] (the parameter's own layout; the bf16 H = 256 ring kernel and the fp32 kernels) instead of
    [out][in].  slope: `relu` / `mask_pos` as leaky ReLU (0 = ReLU): relu gives max(v, 0) + slope * min(v, 0), mask_pos
    multiplies by slope instead of zeroing.  fp32 only: W_loop [H, H] serves the tiles of relation `loop_rel` (the self-loop
    parameter, not concatenated behind Wn); bias_rel >= 0: `bias` is one row [H] added to that relation's tiles only."""
    tiles, ntiles = tile_table
    require_gpu(X, Wn, tiles, idx, X2, bias, mask_pos, W_loop)
    assert X.dtype == Wn.dtype and X.dtype in (torch.bfloat16, torch.float32) and Wn.dim() == 3
    assert bias is None or bias.dtype == X.dtype
    assert not w_kn or X.dtype == torch.float32 or (Wn.shape[1] == Wn.shape[2] and Wn.shape[1] in (64, 128, 256))
    assert (W_loop is None and bias_rel < 0) or X.dtype == torch.float32
    assert W_loop is None or (W_loop.dtype == X.dtype and W_loop.shape == Wn.shape[1:] and W_loop.is_contiguous() and loop_rel >= 0)
    assert Wn.is_contiguous() and (bias is None or bias.is_contiguous())
    Ho, Hi = Wn.shape[1], Wn.shape[2]
    assert X.shape[1] == Hi
    if out is None:
        Y = torch.empty((num_rows, Ho), dtype=X.dtype, device=X.device)
    else:
        assert out.is_contiguous() and out.dtype == X.dtype and out.shape[0] >= num_rows and out.shape[1] == Ho
        Y = out[:num_rows]

    def _launch():
        n1 = X.shape[0] if X2 is not None else INT32_MAX
        if X.dtype == torch.float32:
            check(lib().dn_rows_transform_f32(ptr(X), ptr(X2), n1, ptr(idx), Hi, Ho, ptr(Wn), ptr(bias), 1 if relu else 0,
                                              ptr(mask_pos), ptr(tiles), ntiles, ptr(Y), 1 if f32_mode() else 0, float(slope),
                                              ptr(W_loop), int(loop_rel) if W_loop is not None else -1, int(bias_rel),
                                              1 if w_kn else 0, stream_ptr()),
                  "dn_rows_transform_f32")
        else:
            check(lib().dn_rows_transform_bf16(ptr(X), ptr(X2), n1, ptr(idx), Hi, Ho, ptr(Wn), ptr(bias), 1 if relu else 0,
                                               ptr(mask_pos), ptr(tiles), ntiles, ptr(Y), 1 if w_kn else 0, float(slope), stream_ptr()),
                  "dn_rows_transform_bf16")

    if kernel_timer is not None:
        kernel_timer.launch("rows_transform:" + tag, _launch)
    else:
        _launch()
    return Y


MFMA_DTYPES = (torch.bfloat16, torch.float32)


SELFSUM_SLOTS = 6
SELFSUM_ENABLED = _os.environ.get("DN_SELFSUM", "1") != "0"


# DN_OVER_INSIDE_ROWS: up to this many nodes dn_rows_selfsum_bf16 finishes the nodes with more rows than slots itself (0: never).
# Config-3-shaped batches, H = 64 bf16, step with the walk inside / as its own launch: 25.6 k nodes 0.0985 / 0.1067 ms, 38 k 0.114 /
# 0.118, 51 k 0.123 / 0.127, 77 k 0.151 / 0.150, 102 k 0.182 / 0.175, 410 k 0.533 / 0.485
OVERFLOW_INSIDE_MAX_ROWS = int(_os.environ.get("DN_OVER_INSIDE_ROWS", "65536"))
# ... and only while no node's list is longer than this: the walk inside the slot kernel takes a list one entry at a time (dependent
# loads) while the rest of the workgroup waits at a barrier, so one hub node with thousands of rows would stall its tile's pipeline;
# dn_overflow_rows_add_bf16 ranks 64 entries per round trip
OVERFLOW_INSIDE_MAX_LIST = int(_os.environ.get("DN_OVER_INSIDE_LIST", "64"))


def rows_selfsum(x, Wn, bias, S, S2, slots, out=None, seg=None, lists=None, w_kn=False):
    """out[v] = x[v] @ Wn^T (+ bias) + sum_k Scat[slots[v, k]]  (dn_rows_selfsum_bf16; Scat = S rows then S2 rows).
    seg = (fold_info int32 [ceil(N/32), 12], seg_part fp32 [n_part, H]): also write the per-(segment, tile) column sums of x (the
    folded pre-aggregation, see the header).  lists = (list_ptr, list_rows, num_edge_rows, drop_beg, drop_end, overflow[, longest
    list]): the per-node row lists the slot table was built from -- nodes with more rows than slots (-2 in their last slot) are finished
    from them inside the launch (small batches whose longest list is known and short) or by a second small launch
    (dn_overflow_rows_add_bf16).  w_kn: Wn is given [in][out] (the parameter's own layout) instead."""
    require_gpu(x, Wn, bias, S, S2, slots)
    if seg is not None:
        require_gpu(*seg)
        assert seg[0].dtype == I32 and seg[0].shape == ((x.shape[0] + 31) // 32, 12) and seg[0].is_contiguous()
        assert seg[1].dtype == torch.float32 and seg[1].shape[1] == x.shape[1] and seg[1].is_contiguous()
    N, H = x.shape
    assert x.dtype == torch.bfloat16 and Wn.shape == (H, H) and slots.shape == (N, SELFSUM_SLOTS) and slots.dtype == I32
    x, Wn, slots = x.contiguous(), Wn.contiguous(), slots.contiguous()
    if out is None:
        out = torch.empty((N, H), dtype=x.dtype, device=x.device)
    n1 = int(S.shape[0]) if S2 is not None else 0x7fffffff

    # small batches: the nodes with more rows than slots are finished inside the launch (one launch less); large ones by the
    # overflow launch below (the in-kernel walk stalls the tile pipeline: + 130 us at config 5)
    inside = (lists is not None and N <= OVERFLOW_INSIDE_MAX_ROWS and S is not None and S.numel() > 0
              and len(lists) > 6 and lists[6] is not None and lists[6] <= OVERFLOW_INSIDE_MAX_LIST)
    lin = lists if inside else (None, None, 0, 0, 0, None)

    def _launch():
        check(lib().dn_rows_selfsum_bf16(ptr(x), H, ptr(Wn), ptr(bias), ptr(S) if S is not None and S.numel() else None,
                                         ptr(S2), n1, ptr(slots), SELFSUM_SLOTS, N, ptr(out),
                                         ptr(seg[0]) if seg else None, ptr(seg[1]) if seg else None, 1 if w_kn else 0,
                                         ptr(lin[0]), ptr(lin[1]), int(lin[2]), int(lin[3]), int(lin[4]), stream_ptr()),
              "dn_rows_selfsum_bf16")
    if kernel_timer is not None:
        kernel_timer.launch("rows_selfsum", _launch)
    else:
        _launch()
    if lists is not None and not inside:
        lp, lr, ner, db, de, over = lists[:6]
        require_gpu(lp, lr, over)
        _i32(lp, "list_ptr"), _i32(lr, "list_rows")
        assert lp.numel() >= N + 1 and S2 is None and over.dtype == torch.uint8 and over.numel() >= N

        def _launch2():
            check(lib().dn_overflow_rows_add_bf16(ptr(S) if S is not None and S.numel() else None, H, ptr(over), SELFSUM_SLOTS, N,
                                                  ptr(lp), ptr(lr), int(ner), int(db), int(de), ptr(out), stream_ptr()),
                  "dn_overflow_rows_add_bf16")
        if kernel_timer is not None:
            kernel_timer.launch("overflow_rows_add", _launch2)
        else:
            _launch2()
    return out


CHAIN2_ENABLED = _os.environ.get("DN_CHAIN2", "1") != "0"


def rows_chain2(x, W1n, b1, relu1, W2n, b2, relu2, mask0_bits=None, mask1_bits=None, want_bits=False, w_kn=(False, False),
                slope=0.0):
    """(Y1, Y2[, bits1, bits2]) with Y1 = epi1(m0(x) @ W1n^T), Y2 = epi2(Y1 @ W2n^T) in one pass over the rows
    (dn_rows_chain2_bf16).  mask*_bits: uint8 [N, H/8] keep-masks (input / stage-1 output); want_bits: also return the
    "> 0" bit tensors of Y1 and Y2 (the ReLU masks the backward needs, 1/16 of the activations).  w_kn[i]: weight i is given
    [in][out] instead of [out][in] (the backward chain on the Linear weights as they are: no transposed copies)."""
    x, W1n, W2n = x.contiguous(), W1n.contiguous(), W2n.contiguous()
    require_gpu(x, W1n, b1, W2n, b2, mask0_bits, mask1_bits)
    N, H = x.shape
    assert x.dtype == torch.bfloat16 and W1n.shape == (H, H) and W2n.shape == (H, H)
    for m in (mask0_bits, mask1_bits):
        assert m is None or (m.dtype == torch.uint8 and m.shape == (N, H // 8))
    Y1, Y2 = torch.empty_like(x), torch.empty_like(x)
    bits1 = torch.empty((N, H // 8), dtype=torch.uint8, device=x.device) if want_bits else None
    bits2 = torch.empty((N, H // 8), dtype=torch.uint8, device=x.device) if want_bits else None

    def _launch():
        check(lib().dn_rows_chain2_bf16(ptr(x), H, ptr(W1n), ptr(b1), 1 if relu1 else 0, ptr(mask0_bits), ptr(mask1_bits),
                                        ptr(W2n), ptr(b2), 1 if relu2 else 0, N, ptr(Y1), ptr(Y2), ptr(bits1), ptr(bits2),
                                        (1 if w_kn[0] else 0) | (2 if w_kn[1] else 0), float(slope), stream_ptr()),
              "dn_rows_chain2_bf16")
    if kernel_timer is not None:
        kernel_timer.launch("rows_chain2", _launch)
    else:
        _launch()
    return (Y1, Y2, bits1, bits2) if want_bits else (Y1, Y2)


def rows_chain2_f32(x, W1n, b1, relu1, W2n, b2, relu2, mask0=None, mask1=None, w_kn=(False, False), slope=0.0, residual=None):
    """(Y1, Y2) of two dense layers in one pass over fp32 rows (dn_rows_chain2_f32, 3-term split, H = 64 / 128): Y1 = epi1(m0(x) @ W1n^T)
    masked by mask1, Y2 = epi2(Y1 @ W2n^T); mask0 / mask1: float [N, H] saved activations (keep where > 0, else x slope)."""
    x, W1n, W2n = x.contiguous(), W1n.contiguous(), W2n.contiguous()
    require_gpu(x, W1n, b1, W2n, b2, mask0, mask1)
    N, H = x.shape
    require_gpu(residual)
    assert x.dtype == torch.float32 and H in (64, 128) and W1n.shape == (H, H) and W2n.shape == (H, H) and not f32_mode()
    for m in (mask0, mask1):
        assert m is None or (m.dtype == torch.float32 and m.shape == (N, H) and m.is_contiguous())
    Y1, Y2 = torch.empty_like(x), torch.empty_like(x)
    assert residual is None or (residual.dtype == torch.float32 and residual.shape == x.shape and residual.is_contiguous())
    Ysum = torch.empty_like(x) if residual is not None else None      # residual given: also returns Y2 + residual

    def _launch():
        check(lib().dn_rows_chain2_f32(ptr(x), H, ptr(W1n), ptr(b1), 1 if relu1 else 0, ptr(mask0), ptr(mask1), ptr(W2n), ptr(b2),
                                       1 if relu2 else 0, N, ptr(Y1), ptr(Y2), (1 if w_kn[0] else 0) | (2 if w_kn[1] else 0), float(slope),
                                       ptr(residual), ptr(Ysum), stream_ptr()), "dn_rows_chain2_f32")
    if kernel_timer is not None:
        kernel_timer.launch("rows_chain2", _launch)
    else:
        _launch()
    return (Y1, Y2, Ysum) if residual is not None else (Y1, Y2)


def build_slot_table(list_ptr, list_rows, num_nodes, num_edge_rows, K=SELFSUM_SLOTS, drop=(0, 0), drop_enable=None):
    """Fixed-width view of per-node row lists for dn_rows_selfsum_bf16 (dn_slot_table_build_i32: one launch, no read-back):
    (slots [N, K] int32, overflow [N] uint8).  Rows >= num_edge_rows (the self-loop rows) and rows in drop = (beg, end) are left out (drop_enable: a
    device flag that switches the range off when 0); a node with more than K rows keeps its first K-1 and gets -2 in its last
    slot and 1 in its overflow byte -- a small launch after the closing launch finishes it from the list (pass lists to rows_selfsum)."""
    require_gpu(list_ptr, list_rows)
    dev = list_rows.device
    N, P = int(num_nodes), int(num_edge_rows)
    list_ptr, list_rows = list_ptr.to(I32).contiguous(), list_rows.to(I32).contiguous()
    slots = torch.empty((N, K), dtype=I32, device=dev)
    over = torch.empty(max(N, 1), dtype=torch.uint8, device=dev)
    check(lib().dn_slot_table_build_i32(N, P, K, ptr(list_ptr), ptr(list_rows), int(drop[0]), int(drop[1]), ptr(drop_enable),
                                        ptr(slots), ptr(over), stream_ptr()), "dn_slot_table_build_i32")
    return slots, over


# The closing launch at H = 256 as a stream of 32-row units (csrc/dn_close.hip): no slot limit, no overflow launch.
# DN_CLOSE_RING=0 keeps dn_rows_selfsum_bf16 + dn_overflow_rows_add_bf16 at every width (e.g. to trace a non-finite row: the
# unit kernel turns a NaN / Inf of one product row into NaN for that column of the whole 32-node tile).
CLOSE_RING_ENABLED = _os.environ.get("DN_CLOSE_RING", "1") != "0"


class CloseUnits:
    """Tables of dn_rows_close_bf16 for one direction of a RowIndex (dn_close_units_build_i32)."""
    __slots__ = ("unit_ptr", "units", "ent_row", "ent_mask", "num_wg", "num_nodes", "num_tiles", "agg", "order", "num_segments")


# DN_CLOSE_ORDER=0: the closing launch walks the batch upwards as one front (workgroup w: tiles w, w + G, ...) instead of every XCD
# walking its eighth of the batch downwards -- towards the rows the transform launch in front of it handled last
CLOSE_XCD_ORDER = _os.environ.get("DN_CLOSE_ORDER", "1") != "0"


def _close_order(num_wg):
    return 1 if (CLOSE_XCD_ORDER and num_wg % 8 == 0) else 0


def _num_cus(dev):
    return int(torch.cuda.get_device_properties(dev).multi_processor_count)


def build_graph_tiles(seg_ptr, seg_nodes, num_nodes, ok=None, add_idx=None):
    """Tiles = the graphs of a batch, for the absorbed fold (dn_fold_graph_tiles_build_i32, one launch, no read-back):
    -> (tile_ptr [S + 1], fold_info [S, 12], ok [1] device flag: non-zero when every graph has at most 32 nodes, the
    segments are contiguous ascending runs and -- add_idx [S] given -- every segment's target row lies inside its own tile:
    the AGG unit of a tile is a read-modify-write by the workgroup that stored the tile)."""
    require_gpu(seg_ptr, seg_nodes, add_idx)
    dev = seg_ptr.device
    S = int(seg_ptr.numel()) - 1
    tile_ptr = torch.empty(S + 1, dtype=I32, device=dev)
    info = torch.empty((max(S, 1), 12), dtype=I32, device=dev)
    if ok is None:
        ok = torch.zeros(1, dtype=I32, device=dev)
    assert add_idx is None or (add_idx.dtype == I32 and add_idx.numel() == S and add_idx.is_contiguous())
    check(lib().dn_fold_graph_tiles_build_i32(int(num_nodes), S, ptr(seg_ptr), ptr(seg_nodes), ptr(add_idx), ptr(tile_ptr),
                                              ptr(info), ptr(ok), stream_ptr()), "dn_fold_graph_tiles_build_i32")
    return tile_ptr, info[:S], ok


# DN_CLOSE_SINGLE=0 (experiments): never use the graphs-as-tiles tables of round 4, also where every graph fits one tile
CLOSE_SINGLE_ENABLED = _os.environ.get("DN_CLOSE_SINGLE", "1") != "0"
# DN_CLOSE_MULTI=0: graphs over 32 nodes keep the fp32 partial rows + dn_fold_tail_bf16 (rounds 4-5) instead of the multi-tile absorbed fold
CLOSE_MULTI_ENABLED = _os.environ.get("DN_CLOSE_MULTI", "1") != "0"


# DN_CLOSE_CHUNK_TILES: 32-node tiles per chunk of the chunked closing launch (a chunk = a run of whole graphs that one workgroup
# takes in one piece; every workgroup gets the same number of chunks, dealt in the front order of the single-tile tables)
CLOSE_CHUNK_TILES = int(_os.environ.get("DN_CLOSE_CHUNK_TILES", "16"))


def close_chunks(num_nodes, num_wg):
    """Chunks of a batch for the chunked closing launch: K per workgroup, about CLOSE_CHUNK_TILES tiles each."""
    k = int(round(num_nodes / 32.0 / num_wg / max(CLOSE_CHUNK_TILES, 1)))
    return num_wg * max(1, min(k, 16383 // num_wg))


def build_graph_tiles_multi(seg_ptr, seg_nodes, num_nodes, ok=None, add_idx=None, num_chunks=None):
    """Tiles of a batch of graphs of ANY size for the absorbed fold (dn_fold_graph_tiles_multi_build_i32: three launches, no
    read-back): the batch cut into num_chunks chunks at graph boundaries (one per workgroup of the closing launch), every chunk
    into consecutive 32-node tiles that run across its graphs.
    -> (tile_ptr [cap + 1], fold_info [cap, 12], chunk_tile [C + 1], chunk_graph [C + 1], cap, ok [1] device flag); the number of
    tiles that exist is chunk_tile[C], on the device; cap bounds it."""
    require_gpu(seg_ptr, seg_nodes, add_idx)
    dev = seg_ptr.device
    S, N = int(seg_ptr.numel()) - 1, int(num_nodes)
    C = int(num_chunks) if num_chunks else close_chunks(N, _num_cus(dev))
    cap = int(lib().dn_fold_graph_tiles_multi_capacity(N, C))
    chunk_tile = torch.empty(C + 1, dtype=I32, device=dev)
    chunk_graph = torch.empty(C + 1, dtype=I32, device=dev)
    tile_ptr = torch.empty(cap + 1, dtype=I32, device=dev)
    info = torch.empty((cap, 12), dtype=I32, device=dev)
    if ok is None:
        ok = torch.zeros(1, dtype=I32, device=dev)
    assert add_idx is None or (add_idx.dtype == I32 and add_idx.numel() == S and add_idx.is_contiguous())
    check(lib().dn_fold_graph_tiles_multi_build_i32(N, S, ptr(seg_ptr), ptr(seg_nodes), ptr(add_idx), C, ptr(chunk_tile), ptr(chunk_graph),
                                                    ptr(tile_ptr), ptr(info), cap, ptr(ok), stream_ptr()),
          "dn_fold_graph_tiles_multi_build_i32")
    return tile_ptr, info, chunk_tile, chunk_graph, cap, ok


def build_close_units(list_ptr, list_rows, num_nodes, num_edge_rows, drop=(0, 0), drop_enable=None, num_wg=None, tile_ptr=None,
                      agg=False, order=None, multi=None):
    """Per tile (32-node windows, or the node ranges tile_ptr gives) the distinct kept rows of its nodes' lists + membership
    masks, and the per-workgroup unit records the closing launch streams (three launches, no read-back).  Same filter as
    build_slot_table.  agg: append every workgroup's AGG units (the absorbed fold, tile_ptr from build_graph_tiles).  order: 0 =
    workgroup w takes tiles w, w + G, ...; 1 = every XCD walks its eighth of the batch downwards (include/dn_hip.h); None = 1 when
    the number of workgroups allows it.  multi = (chunk_tile, chunk_graph, cap, num_segments) of build_graph_tiles_multi (with its
    tile_ptr; its num_chunks = K num_wg): orders 2 / 3 -- the chunks are dealt to the workgroups as orders 0 / 1 deal tiles, so a
    graph's tiles stay in one workgroup's stream."""
    require_gpu(list_ptr, list_rows, tile_ptr)
    dev = list_rows.device
    N, P, L = int(num_nodes), int(num_edge_rows), int(list_rows.numel())
    list_ptr, list_rows = list_ptr.to(I32).contiguous(), list_rows.to(I32).contiguous()
    cu = CloseUnits()
    cu.num_wg = int(num_wg) if num_wg else _num_cus(dev)
    cu.num_nodes, cu.agg = N, bool(agg)
    cu.order = _close_order(cu.num_wg) if order is None else int(order)
    cu.num_tiles = T = (int(tile_ptr.numel()) - 1) if tile_ptr is not None else (N + 31) // 32
    cu.num_segments = T
    tg = tf = None
    kper = 0
    if multi is not None:
        tg, tf, cap_t, nseg = multi
        require_gpu(tg, tf)
        kper = (int(tg.numel()) - 1) // cu.num_wg
        assert agg and tile_ptr is not None and cap_t == T and tg.numel() == kper * cu.num_wg + 1 == tf.numel() and kper >= 1
        cu.order, cu.num_segments = 2 + _close_order(cu.num_wg), int(nseg)
    assert not agg or tile_ptr is not None
    cap = int(lib().dn_close_units_capacity(T, L, cu.num_wg))
    cu.unit_ptr = torch.empty(cu.num_wg + 1, dtype=I32, device=dev)
    cu.units = torch.empty((cap, 4), dtype=I32, device=dev)
    cu.ent_row = torch.empty(max(L, 1), dtype=I32, device=dev)
    cu.ent_mask = torch.empty(max(L, 1), dtype=I32, device=dev)
    ws = _ws(lib().dn_close_units_workspace_bytes(T, cu.num_wg), dev)
    check(lib().dn_close_units_build_i32(N, P, cu.num_wg, ptr(tile_ptr), T, 1 if agg else 0, cu.order, ptr(list_ptr), ptr(list_rows), L,
                                         int(drop[0]), int(drop[1]), ptr(drop_enable), ptr(cu.unit_ptr), ptr(cu.units), cap,
                                         ptr(cu.ent_row), ptr(cu.ent_mask), ptr(tg), ptr(tf), kper, ptr(ws), ws.numel(), stream_ptr()),
          "dn_close_units_build_i32")
    return cu


def rows_close(x, W, bias, S, cu, out=None, seg=None, w_kn=False, agg=None):
    """out[v] = x[v] @ W_loop (+ bias) + sum of the rows of S that cu lists for v  (dn_rows_close_bf16, H = 256).
    W: [out, in] (w_kn False, the transposed copy) or [in, out] as the parameter stores it (w_kn True).  seg as in rows_selfsum
    (fp32 partial rows for dn_fold_tail_bf16).  agg = (fold_info [T, 12], W_agg [H, H] in W's layout, aux [S, H] out, agg_idx [S]):
    the absorbed fold -- cu built with build_graph_tiles' (or build_graph_tiles_multi's) tiles and agg=True; aux[j] = the column sum
    of segment j (bf16), out[agg_idx[j]] += aux[j] @ W_agg inside the same launch."""
    require_gpu(x, W, bias, S, cu.unit_ptr, cu.units, cu.ent_row, cu.ent_mask)
    N, H = x.shape
    assert H == 256 and x.dtype == torch.bfloat16 and W.shape == (H, H) and W.dtype == x.dtype and N == cu.num_nodes
    assert S is None or (S.dtype == x.dtype and S.shape[1] == H and S.is_contiguous())
    assert (agg is not None) == cu.agg and not (agg is not None and seg is not None)
    x, W = x.contiguous(), W.contiguous()
    fi = sp = wa = ax = ai = None
    if seg is not None:
        require_gpu(*seg)
        fi, sp = seg
        assert fi.dtype == I32 and fi.shape == (cu.num_tiles, 12) and fi.is_contiguous()
        assert sp.dtype == torch.float32 and sp.shape[1] == H and sp.is_contiguous()
    if agg is not None:
        require_gpu(*agg)
        fi, wa, ax, ai = agg
        assert fi.dtype == I32 and fi.shape == (cu.num_tiles, 12) and fi.is_contiguous()
        assert wa.dtype == x.dtype and wa.shape == (H, H) and wa.is_contiguous()
        assert ax.dtype == x.dtype and ax.shape == (cu.num_segments, H) and ax.is_contiguous()
        assert ai.dtype == I32 and ai.numel() == cu.num_segments and ai.is_contiguous()
    if out is None:
        out = torch.empty((N, H), dtype=x.dtype, device=x.device)

    def _launch():
        check(lib().dn_rows_close_bf16(ptr(x), H, ptr(W), 1 if w_kn else 0, ptr(bias), ptr(S) if S is not None and S.numel() else None,
                                       ptr(cu.unit_ptr), ptr(cu.units), cu.num_wg, ptr(cu.ent_row), ptr(cu.ent_mask), N, ptr(out),
                                       ptr(fi), ptr(sp), ptr(wa), ptr(ax), ptr(ai), stream_ptr()), "dn_rows_close_bf16")
    if kernel_timer is not None:
        kernel_timer.launch("rows_close", _launch)
    else:
        _launch()
    return out


def wgrad_supported(A, G):
    return A.dtype == G.dtype and A.dtype in MFMA_DTYPES and A.shape[1] == G.shape[1] and A.shape[1] in (64, 128, 256)


# ----------------------------------------------------------------------------------------------
# index structures
# ----------------------------------------------------------------------------------------------
HUB_SPLIT = 64   # segments longer than this are split into chunks summed by separate lane groups


class _SplitCSR:
    """CSR whose long segments ("hubs": the dummy node of a big graph has in-degree n) are taken out of the main pass and
    summed in HUB_SPLIT-entry chunks by separate lane groups, then folded in chunk order (deterministic).  One lane group
    walking a 600-entry segment would otherwise outlast the whole launch (cdna_hip_programming.md, scatter/gather:
    'split lists longer than ... into chunks summed by separate waves')."""

    def __init__(self, ptr, idx, num_segments):
        self.ptr, self.idx, self.num_segments = ptr, idx, int(num_segments)
        self.hub_ids = None
        if idx.numel() == 0:
            return
        deg = (ptr[1:] - ptr[:-1]).long()
        if int(deg.max()) <= HUB_SPLIT:
            return
        dev = ptr.device
        hub = deg > HUB_SPLIT
        hub_ids = torch.nonzero(hub).reshape(-1)
        seg_of_entry = torch.repeat_interleave(torch.arange(self.num_segments, device=dev), deg)
        keep = ~hub[seg_of_entry]
        # main pass: hubs become empty segments
        deg_main = torch.where(hub, torch.zeros_like(deg), deg)
        self.ptr_main = torch.cat([deg_main.new_zeros(1), torch.cumsum(deg_main, 0)]).to(I32)
        self.keep = keep
        # (positions once, so that a scaled sum has no boolean indexing per call: nonzero + host sync, not capturable)
        self.keep_pos, self.hub_pos = torch.nonzero(keep).reshape(-1), torch.nonzero(~keep).reshape(-1)
        self.idx_main = idx[keep].contiguous()
        # hub pass: entries of hub h in HUB_SPLIT-sized chunks
        hdeg = deg[hub_ids]
        nchunk = (hdeg + HUB_SPLIT - 1) // HUB_SPLIT
        self.idx_hub = idx[~keep].contiguous()
        hub_base = torch.cat([hdeg.new_zeros(1), torch.cumsum(hdeg, 0)])[:-1]
        chunk_hub = torch.repeat_interleave(torch.arange(hub_ids.numel(), device=dev), nchunk)
        first_chunk = torch.cat([nchunk.new_zeros(1), torch.cumsum(nchunk, 0)])
        chunk_in_hub = torch.arange(int(first_chunk[-1]), device=dev) - first_chunk[:-1][chunk_hub]
        cbeg = hub_base[chunk_hub] + chunk_in_hub * HUB_SPLIT
        cend = torch.minimum(cbeg + HUB_SPLIT, (hub_base + hdeg)[chunk_hub])
        self.chunk_ptr = torch.cat([cbeg, cend[-1:]]).to(I32)                 # chunks are contiguous in idx_hub
        self.fold_ptr = first_chunk.to(I32)                                   # chunk ranges per hub
        self.hub_ids = hub_ids

    def segsum(self, x, scale=None, self_in=None, self_coef=0.0):
        if self.hub_ids is None:
            return gather_segsum(x, self.idx, self.ptr, self.num_segments, scale=scale, self_in=self_in, self_coef=self_coef)
        sc_main = sc_hub = None
        if scale is not None:
            sc_main, sc_hub = scale.index_select(0, self.keep_pos), scale.index_select(0, self.hub_pos)
        out = gather_segsum(x, self.idx_main, self.ptr_main, self.num_segments, scale=sc_main, self_in=self_in,
                            self_coef=self_coef)
        part = gather_segsum(x, self.idx_hub, self.chunk_ptr, self.chunk_ptr.numel() - 1, scale=sc_hub)
        hub = gather_segsum(part, None, self.fold_ptr)                        # per-hub sum of its chunk partials
        out.index_add_(0, self.hub_ids, hub)                                  # distinct rows: order-independent
        return out


class EdgeIndex:
    """CSR by destination + CSC by source of one batched COO (int32, device resident).
    The one-shot build DGL / torch-scatter hide behind update_all / scatter."""

    def __init__(self, src, dst, num_nodes, node_ptr=None):
        """node_ptr (optional, [G + 1]): the batch's graph boundaries -- with it, fp32 neighbour sums of small graphs run on the
        matrix cores (tile_plan / dn_graph_tile_sum_f32)."""
        require_gpu(src, dst)
        self.num_nodes, self.num_edges = int(num_nodes), int(src.numel())
        self._node_ptr, self._plan = node_ptr, None
        src, dst = src.to(I32).contiguous(), dst.to(I32).contiguous()
        self.src, self.dst = src, dst
        self.in_ptr, self.in_perm = csr_build(dst, num_nodes)
        self.out_ptr, self.out_perm = csr_build(src, num_nodes)
        # neighbour id lists in segment order (row gather of an int column == index_select plumbing)
        self.src_by_dst = gather_rows_i32(src, self.in_perm)
        self.dst_by_src = gather_rows_i32(dst, self.out_perm)
        self.fwd = _SplitCSR(self.in_ptr, self.src_by_dst, num_nodes)
        self.bwd = _SplitCSR(self.out_ptr, self.dst_by_src, num_nodes)
        self._max_bwd = None

    def tile_plan(self):
        """Plan of the matrix-core neighbour sum (dn_graph_tile_sum_f32) for this batch, or None: tiles = greedy runs of whole
        graphs with at most 64 rows (packed by dn_graph_tiles_host from one small copy of the graph boundaries), the rows of larger
        graphs as two row lists per direction for dn_gather_rows_sum_f32 (lists of up to 64 entries: a lane group per row; hubs: a
        workgroup per row).  Built once per batch."""
        if self._plan is None:
            self._plan = _build_tile_plan(self) or False
        return self._plan or None

    def max_backward_index(self):
        """(tptr, tslot, seg_of_slot): CSR slots grouped by the row they gather, and each slot's destination."""
        if self._max_bwd is None:
            tptr, tslot = csr_build(self.src_by_dst, self.num_nodes)
            deg = (self.in_ptr[1:] - self.in_ptr[:-1]).long()
            seg = torch.repeat_interleave(torch.arange(self.num_nodes, device=deg.device, dtype=I32), deg,
                                          output_size=self.num_edges)
            self._max_bwd = (tptr, tslot, seg.contiguous())
        return self._max_bwd


def gather_rows_i32(values, perm):
    return values.index_select(0, perm.long()) if values.numel() else values.clone()


def _check_edge_types(etype, num_rels):
    """Edge types outside [0, num_rels) would index past the mode table / alias another relation's sort key inside the
    index builds (the reference raises IndexError from weight.index_select(0, etype), rgin.py:109): fail loudly instead.
    One fused reduction + a single scalar read-back per index build."""
    if etype.numel() == 0:
        return
    lo, hi = torch.aminmax(etype)
    lo, hi = int(lo), int(hi)
    if lo < 0 or hi >= int(num_rels):
        raise _lib.DnHipError("edge type out of [0, %d): min %d, max %d (with dummy edges the relation count grows by the "
                              "dummy labels, e.g. max_ngel + 2 in the SI flow)" % (int(num_rels), lo, hi))


class RelIndex:
    """(rel, dst)-segment index for aggregate-then-transform RGCN/RGIN (dn_rel_index_build_i32)."""

    def __init__(self, src, dst, etype, num_nodes, num_rels):
        require_gpu(src, dst, etype)
        dev = src.device
        N, R, E = int(num_nodes), int(num_rels), int(src.numel())
        self.num_nodes, self.num_rels, self.num_edges = N, R, E
        _check_edge_types(etype, R)
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
        # chunk / tile tables on the device: split-K chunks of the matrix-core weight gradient, and the tables of the any-width
        # grouped products (dn_rows_gemm_* / dn_rows_wgrad_any_*)
        rel_ptr_d = torch.tensor(self.rel_ptr_host, dtype=I32).to(dev, non_blocking=True)
        self.chunk_table = build_row_tables(rel_ptr_d, R, P, WGRAD_CHUNK_ROWS, want_ptr=True)
        self.gemm_tiles = build_row_tables(rel_ptr_d, R, P, 64)
        self.gemm_chunks = build_row_tables(rel_ptr_d, R, P, 1024, want_ptr=True)


# ----------------------------------------------------------------------------------------------
# autograd operators
# ----------------------------------------------------------------------------------------------
# DN_TILE_SUM=0: graph-local neighbour sums always take the plain gather (dn_gather_segsum_*)
TILE_SUM_ENABLED = _os.environ.get("DN_TILE_SUM", "1") != "0"
TILE_SUM_ROWS = 64
TILE_SUM_MIN_ROWS = int(_os.environ.get("DN_TILE_SUM_MIN_ROWS", "4096"))


class _TilePlan:
    __slots__ = ("dirs", "bad", "checked", "covered")


def _build_tile_plan(index):
    import numpy as np
    node_ptr = index._node_ptr
    if node_ptr is None or int(node_ptr.numel()) < 2 or index.num_nodes == 0:
        return None
    dev = index.src.device
    npt = np.ascontiguousarray(node_ptr.detach().cpu().numpy().astype(np.int32))
    G, N = int(npt.shape[0]) - 1, index.num_nodes
    if int(npt[0]) != 0 or int(npt[-1]) != N or np.any(np.diff(npt) < 0):
        return None
    sizes = np.diff(npt)
    small = sizes <= TILE_SUM_ROWS
    if not np.any(small & (sizes > 0)):
        return None
    buf = np.empty((max(G, 1), 2), dtype=np.int32)
    n = ctypes.c_int64(0)
    check(lib().dn_graph_tiles_host(npt.ctypes.data_as(ctypes.c_void_p), G, TILE_SUM_ROWS, buf.ctypes.data_as(ctypes.c_void_p),
                                    max(G, 1), ctypes.byref(n)), "dn_graph_tiles_host")
    tl = buf[:int(n.value)]
    beg, end = tl[:, 0], tl[:, 1]
    keep = end > beg
    tiles = torch.from_numpy(tl.copy()).to(dev)
    rest = torch.from_numpy(np.flatnonzero(np.repeat(~small, sizes)).astype(np.int32)).to(dev)
    plan = _TilePlan()
    plan.bad = torch.zeros(1, dtype=I32, device=dev)
    plan.checked = False
    plan.covered = int((end[keep] - beg[keep]).sum())
    assert plan.covered + int(rest.numel()) == N
    plan.dirs = {}
    for d, ptr_, idx in (("f", index.in_ptr, index.src_by_dst), ("b", index.out_ptr, index.dst_by_src)):
        rec = graph_tile_records(tiles, ptr_)
        if rest.numel():
            rl = rest.long()
            deg = ptr_[rl + 1] - ptr_[rl]
            recs = torch.stack([rest, ptr_[rl], ptr_[rl + 1], torch.zeros_like(rest)], 1)       # {row, first entry, end entry, 0}
            short, long_ = recs[deg <= 64].contiguous(), recs[deg > 64].contiguous()
        else:
            short = long_ = rest.reshape(0, 4)
        plan.dirs[d] = (rec, short, long_, ptr_, idx)
    return plan


def _tile_sum_ok(x, index, edge_scale):
    # (batches of a few hundred rows are launch-bound: the plain kernel's one launch beats tile + row-list launches there --
    #  config 1, 617 rows: 0.45 vs 0.51 ms per GIN step; config 2, 20 k rows: 0.86 vs 0.83)
    return (TILE_SUM_ENABLED and edge_scale is None and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] in (64, 128, 256)
            and x.shape[0] == index.num_nodes and index.num_nodes >= TILE_SUM_MIN_ROWS and index.tile_plan() is not None)


def _tile_neighbor_sum(x, index, direction, self_coef):
    """self_coef * x[v] + sum over v's list of x rows: tiles of small graphs on the matrix cores, the rows of larger graphs from
    their row lists.  Same sums as the plain gather up to fp32 summation order (2e-6)."""
    plan = index.tile_plan()
    rec, short, long_, ptr_, idx = plan.dirs[direction]
    out = torch.empty_like(x)
    graph_tile_sum(x, idx, ptr_, rec, self_coef=self_coef, out=out, bad=plan.bad)
    if short.numel():
        gather_rows_sum(x, idx, ptr_, short, False, self_coef, out)
    if long_.numel():
        gather_rows_sum(x, idx, ptr_, long_, True, self_coef, out)
    if not plan.checked:                                            # once per batch: an edge that leaves its tile?
        if int(plan.bad.item()) != 0:
            index._plan = False
            return None
        plan.checked = True
    return out


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
        ctx.save_for_backward(x if (edge_scale is not None and ctx.needs_input_grad[3]) else x.new_empty(0))
        if _tile_sum_ok(x, index, edge_scale):
            out = _tile_neighbor_sum(x, index, "f", float(self_coef))
            if out is not None:
                return out
        return index.fwd.segsum(x, scale=sc_in, self_in=x if self_coef != 0.0 else None, self_coef=self_coef)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        ix = ctx.index
        gx = None
        if ctx.needs_input_grad[0]:
            gx = _tile_neighbor_sum(g, ix, "b", ctx.self_coef) if _tile_sum_ok(g, ix, ctx.sc_out) else None
            if gx is None:
                gx = ix.bwd.segsum(g, scale=ctx.sc_out, self_in=g if ctx.self_coef != 0.0 else None, self_coef=ctx.self_coef)
        gs = None
        if ctx.needs_input_grad[3]:
            (x,) = ctx.saved_tensors
            gs = edge_dot(x, ix.src, g, ix.dst)                 # d out / d w_e = < x[src_e], g[dst_e] >
        return gx, None, None, gs


def neighbor_sum(x, index, self_coef=0.0, edge_scale=None):
    """(self_coef) x_i + sum_{j->i} w_ij x_j  (GIN aggregation gconv.py:212; GCN propagate with w = gcn norm).
    edge_scale [E] fp32 in original edge order; it receives a gradient (dn_edge_dot_*) when it requires one."""
    return _NeighborSum.apply(x, index, self_coef, edge_scale)


class _EdgeSum(torch.autograd.Function):
    """agg[v] = sum_{e: dst(e)=v} w_e ef[e]   (edge rows summed onto their destination: the reduce half of a dual
    message pass, compgcn.py:272 / dmpnn.py:163); backward: g_ef[e] = w_e g[dst(e)] (a row gather)."""

    @staticmethod
    def forward(ctx, ef, index, edge_scale):
        ef = ef.contiguous()
        ctx.index = index
        sc_in = edge_scale.index_select(0, index.in_perm.long()) if edge_scale is not None else None
        ctx.save_for_backward(edge_scale if edge_scale is not None else ef.new_empty(0))
        ctx.has_scale = edge_scale is not None
        return gather_segsum(ef, index.in_perm, index.in_ptr, index.num_nodes, scale=sc_in)

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return gather_segsum(g.contiguous(), ctx.index.dst, None, scale=scale if ctx.has_scale else None), None, None


def edge_sum(ef, index, edge_scale=None):
    """sum of (scaled) edge rows per destination node; edge_scale [E] fp32 (no gradient) in edge order."""
    return _EdgeSum.apply(ef, index, edge_scale)


class _GatherRows(torch.autograd.Function):
    """out[i] = x[idx[i]]; backward = deterministic segment sum over the rows that gathered each x row (CSR of idx)."""

    @staticmethod
    def forward(ctx, x, idx, csr):
        ctx.csr, ctx.n = csr, x.shape[0]
        return gather_segsum(x.contiguous(), idx, None)

    @staticmethod
    def backward(ctx, g):
        ptr_, perm = ctx.csr
        return gather_segsum(g.contiguous(), perm, ptr_, ctx.n), None, None


def gather_rows(x, idx, csr=None):
    """x[idx] with a reproducible backward.  csr = csr_build(idx, x.shape[0]) may be passed in when it is cached."""
    idx = idx.to(I32).contiguous()
    if csr is None:
        csr = csr_build(idx, x.shape[0])
    return _GatherRows.apply(x, idx, csr)


def edge_dot(a, ia, b, ib):
    """out[e] = <a[ia[e]], b[ib[e]]>  (dn_edge_dot_*), fp32."""
    require_gpu(a, ia, b, ib)
    E = ia.numel() if ia is not None else a.shape[0]
    out = torch.empty(E, dtype=torch.float32, device=a.device)
    check(getattr(lib(), "dn_edge_dot_" + _suffix(a))(ptr(a), ptr(ia), ptr(b), ptr(ib), a.shape[1], E, ptr(out), stream_ptr()),
          "dn_edge_dot")
    return out


class _NeighborMax(torch.autograd.Function):
    """out[v, h] = max_{j->v} x[j, h] (0 for isolated v); backward routes each gradient entry to its arg-max source."""

    @staticmethod
    def forward(ctx, x, index):
        x = x.contiguous()
        N, H = index.num_nodes, x.shape[1]
        out = torch.empty((N, H), dtype=x.dtype, device=x.device)
        arg = torch.empty((N, H), dtype=I32, device=x.device)
        check(getattr(lib(), "dn_gather_segmax_" + _suffix(x))(ptr(x), ptr(index.src_by_dst), ptr(index.in_ptr), N, H,
                                                               ptr(out), ptr(arg), stream_ptr()), "dn_gather_segmax")
        ctx.index, ctx.rows = index, x.shape[0]
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        (arg,) = ctx.saved_tensors
        tptr, tslot, seg = ctx.index.max_backward_index()
        gin = torch.empty((ctx.rows, g.shape[1]), dtype=g.dtype, device=g.device)
        check(getattr(lib(), "dn_gather_segmax_bwd_" + _suffix(g))(ptr(g), ptr(arg), ptr(tptr), ptr(tslot), ptr(seg),
                                                                   ctx.rows, g.shape[1], ptr(gin), stream_ptr()),
              "dn_gather_segmax_bwd")
        return gin, None


def neighbor_max(x, index):
    require_gpu(x)
    return _NeighborMax.apply(x, index)


def rows_gemm(A, W, tile_table, transpose_w=False, bias=None):
    """Y[rows of relation r] = A[rows] @ W[r] (or W[r]^T) (+ bias[r]) for ANY widths in one launch (dn_rows_gemm_*; rows
    relation-major, 64-row tiles from build_row_tables).  W: [R, K, N], or [R, N, K] with transpose_w; bias: [R, N]."""
    tiles, ntiles = tile_table
    A, W = A.contiguous(), W.contiguous()
    require_gpu(A, W, tiles, bias)
    if bias is not None:
        bias = bias.contiguous()
        assert bias.dtype == A.dtype and bias.numel() == W.shape[0] * (W.shape[1] if transpose_w else W.shape[2])
    K = A.shape[1]
    N = W.shape[1] if transpose_w else W.shape[2]
    assert A.dtype == W.dtype and (W.shape[2] if transpose_w else W.shape[1]) == K
    Y = torch.empty((A.shape[0], N), dtype=A.dtype, device=A.device)
    if A.shape[0] == 0:
        return Y

    def _launch():
        check(getattr(lib(), "dn_rows_gemm_" + _suffix(A))(ptr(A), ptr(W), ptr(bias), K, N, 1 if transpose_w else 0, ptr(tiles),
                                                          ntiles, ptr(Y), stream_ptr()), "dn_rows_gemm")
    if kernel_timer is not None:
        kernel_timer.launch("rows_gemm", _launch)
    else:
        _launch()
    return Y


def rows_wgrad_any(A, G, chunk_table, num_rels, want_colsum=False):
    """out[r] = sum_{p in relation r} A[p]^T G[p] for ANY widths (dn_rows_wgrad_any_*); want_colsum: also the fp32 column sums
    of A per relation ([R, K]: the bias gradient when A is a Linear layer's output gradient), from the same launches."""
    chunks, chunk_ptr, nchunks = chunk_table
    A, G = A.contiguous(), G.contiguous()
    require_gpu(A, G, chunks, chunk_ptr)
    K, N = A.shape[1], G.shape[1]
    out = torch.empty((num_rels, K, N), dtype=A.dtype, device=A.device)
    cs = torch.empty((num_rels, K), dtype=torch.float32, device=A.device) if want_colsum else None
    ws = _ws(lib().dn_rows_wgrad_any_workspace_bytes(nchunks, K, N), A.device)
    check(getattr(lib(), "dn_rows_wgrad_any_" + _suffix(A))(ptr(A), ptr(G), K, N, num_rels, ptr(chunks), nchunks, ptr(chunk_ptr),
                                                           ptr(out), ptr(cs), ptr(ws), ws.numel(), stream_ptr()), "dn_rows_wgrad_any")
    return (out, cs) if want_colsum else out


class _RelAggTransform(torch.autograd.Function):
    """agg[v] = sum_r ( sum_{e in r, dst=v} s_e x[src_e] ) W_r   (two gather passes + one relation-grouped GEMM launch)."""

    @staticmethod
    def forward(ctx, x, W, index, edge_scale):
        ctx.f32_mode = f32_mode()
        x = x.contiguous()
        W = W.contiguous()
        ix = index
        sc1 = sc_src = None
        if edge_scale is not None:
            sc1 = edge_scale.index_select(0, ix.perm1.long())
            sc_src = edge_scale.index_select(0, ix.operm.long())
        A = gather_segsum(x, ix.src1, ix.seg_ptr, ix.num_segments, scale=sc1)          # [P, in]
        Y = rows_gemm(A, W, ix.gemm_tiles)                                              # [P, out]
        agg = gather_segsum(Y, ix.sperm, ix.dptr, ix.num_nodes)                         # [N, out]
        ctx.index, ctx.sc_src = ix, sc_src
        ctx.save_for_backward(A, W)
        return agg

    @staticmethod
    @_backward_in_forward_mode
    def backward(ctx, g):
        g = g.contiguous()
        A, W = ctx.saved_tensors
        ix = ctx.index
        gY = gather_segsum(g, ix.seg_dst, None)                                         # [P, out] row gather
        gx = gW = None
        if ctx.needs_input_grad[0]:
            gA = rows_gemm(gY, W, ix.gemm_tiles, transpose_w=True)                      # [P, in]
            gx = gather_segsum(gA, ix.seg_by_src, ix.optr, ix.num_nodes, scale=ctx.sc_src)
        if ctx.needs_input_grad[1]:
            if wgrad_supported(A, g):
                # MFMA split-K kernel; gathers the g rows itself (idx_g = segment destinations)
                gW = rows_wgrad(A, g, ix.chunk_table, W.shape[0], idx_g=ix.seg_dst, out_dtype=W.dtype)
            else:
                gW = rows_wgrad_any(A, gY, ix.gemm_chunks, W.shape[0])
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
        # broadcast each graph's row back to its nodes: a row gather keyed by the node's graph id (+ 1 / count for the mean)
        seg_of_row, scale = _segment_rows(ptr_, ctx.rows, ctx.kind == "mean")
        gin = gather_segsum(g, seg_of_row, None, scale=scale)
        return gin, None, None


_segment_rows_cache = {}


def _segment_rows(ptr_, rows, want_scale):
    """(graph id of every row [rows] int32, 1 / rows-of-its-graph per row or None) for a batch's graph_ptr -- five small launches,
    kept per ptr tensor (id, version): every readout of every layer and step over the same batch object reuses them."""
    key = (id(ptr_), ptr_._version, int(rows), ptr_.data_ptr())
    hit = _segment_rows_cache.get(key)
    if hit is None or hit[0]() is not ptr_:
        S = ptr_.numel() - 1
        lens = ptr_[1:] - ptr_[:-1]
        seg = torch.repeat_interleave(torch.arange(S, device=ptr_.device, dtype=I32), lens.long(), output_size=int(rows))
        if len(_segment_rows_cache) > 64:
            _segment_rows_cache.clear()
        import weakref
        hit = [weakref.ref(ptr_), seg, None]
        _segment_rows_cache[key] = hit
    if want_scale and hit[2] is None:
        cnt = (ptr_[1:] - ptr_[:-1]).to(torch.float32).clamp(min=1.0)
        hit[2] = (1.0 / cnt).index_select(0, hit[1].long())
    return hit[1], (hit[2] if want_scale else None)


def segment_reduce(x, graph_ptr, kind="sum"):
    """global_add_pool / global_mean_pool / global_max_pool over the batch's contiguous node ranges."""
    assert kind in ("sum", "mean", "max")
    return _SegmentReduce.apply(x, graph_ptr, kind)


# ----------------------------------------------------------------------------------------------
# fused relation-wise message pass on the matrix cores (bf16): row factorisation + MFMA kernels
# ----------------------------------------------------------------------------------------------
# DN_LOCAL_INDEX=0: always the general (sort-based) row-index builder, also for batches with graph boundaries
LOCAL_INDEX_ENABLED = _os.environ.get("DN_LOCAL_INDEX", "1") != "0"
# DN_CONV_INDEX=0: never dn_conv_index_build_i32 -- row index, closing tables, sweep orders and chunk table by their own calls
CONV_INDEX_ENABLED = _os.environ.get("DN_CONV_INDEX", "1") != "0"


class RowIndex:
    """Relation-major ROW FACTORISATION of  out[v] = sum_{e: dst(e)=v} x[src(e)] W[etype(e)]  (+ x[v] W_loop).

    Every row p has ONE input row and its product Y[p] = in_row(p) @ W[rel(p)] is added to one or more outputs:
      EDGE relation (few shared endpoints):  one row per edge          in = x[src],            out -> dst
      AGG  relation (few distinct dst, e.g. u -> dummy):  row per dst   in = sum_e x[src_e] (aux, pre-aggregated), out -> dst
      TF   relation (few distinct src, e.g. dummy -> u):  row per src   in = x[src],            out -> every dst_e
      self loop (optional, relation id R):   row per node              in = x[v],              out -> v
    so the matrix cores transform min(#edges, #distinct dst, #distinct src) rows per relation, and the dummy
    relations (2n of the m+2n edges of a dummy-augmented graph) collapse to one row per graph each.
    The backward pass is the mirror image (in <-> out lists, W transposed)."""

    EDGE, AGG, TF = 0, 1, 2

    def __init__(self, src, dst, etype, num_nodes, num_rels, self_loop=True, edge_frac=0.75, node_ptr=None, edge_ptr=None,
                 closing_hint=None):
        """One C-ABI call + device-side tile tables.  With the batch's graph boundaries (node_ptr / edge_ptr, [G+1] each) the
        graph-local builder runs (dn_row_index_build_local_i32: one wavefront rank-sorts one graph in LDS, one scan); without
        them, or when the batch does not qualify (a graph of 8192 edges or more, more than 64 relations), the general one
        (dn_row_index_build_i32: stable radix sorts + scans over the whole batch).  Both produce the same tables bit for bit.
        closing_hint = (H, dtype) of the rows the index will serve: for (256, bfloat16) with graph boundaries the WHOLE per-batch
        index -- row index, unit streams of both closing launches, sweep orders, weight-gradient chunk table -- is one call with one
        read-back (dn_conv_index_build_i32); whatever that call could not serve is built on first use as before."""
        require_gpu(src, dst, etype)
        dev = src.device
        N, R, E = int(num_nodes), int(num_rels), int(src.numel())
        self.num_nodes, self.num_rels, self.num_edges, self.self_loop = N, R, E, bool(self_loop)
        try_local = node_ptr is not None and edge_ptr is not None and R <= 64 and LOCAL_INDEX_ENABLED
        if try_local:                                                 # (graphs over the LDS limit: do not even try)
            try_local = E == 0 or E < 8192 * (int(node_ptr.numel()) - 1)
        if not try_local:
            _check_edge_types(etype, R)                               # (the local builder validates on the device)
        src, dst, etype = (t.to(I32).contiguous() for t in (src, dst, etype))
        e32 = lambda n: torch.empty(max(int(n), 1), dtype=I32, device=dev)  # noqa: E731
        row_in, row_out = e32(E + N), e32(E + N)
        aux_f_ptr, aux_f_idx, aux_b_ptr, aux_b_idx = e32(E + 1), e32(E), e32(E + 1), e32(E)
        dst_ptr, dst_rows, src_ptr, src_rows = e32(N + 2), e32(2 * E + N), e32(N + 2), e32(2 * E + N)
        counts = (ctypes.c_int64 * 5)()
        host_rel = (ctypes.c_int32 * (R + 1))()
        host_modes = (ctypes.c_int32 * R)()
        self.built_by = "general"
        self.max_graph = None                                         # local builder: (nodes, edges) of the batch's largest graph
        self.raw = None                                               # ... and (src, dst, etype, node_ptr, edge_ptr) int32
        self._absorb = None                                           # local builder: {direction: (tile_ptr, fold_info, verdict)}
        rel_dev = None
        pre = None                                                    # (tables dn_conv_index_build_i32 left behind, see below)
        if try_local:
            require_gpu(node_ptr, edge_ptr)
            rel_dev = e32(R + 2)                                      # the relation offsets as the device builder leaves them
            node_ptr, edge_ptr = node_ptr.to(I32).contiguous(), edge_ptr.to(I32).contiguous()
            G = int(node_ptr.numel()) - 1
            # the absorbed-fold verdicts + graph tiles of both directions come out of the same call (and the same read-back)
            gt_bufs = [(e32(G + 1), torch.empty((max(G, 1), 12), dtype=I32, device=dev)) for _ in range(2)]
            host_absorb = (ctypes.c_int32 * 4)()                       # fold verdicts f / b, the largest graph's nodes / edges
            assert G >= 0 and int(edge_ptr.numel()) == G + 1
            one_call = (CONV_INDEX_ENABLED and CLOSE_SINGLE_ENABLED and closing_hint is not None and closing_hint[0] == 256
                        and closing_hint[1] == torch.bfloat16 and self_loop and G >= 1 and N >= 1 and CLOSE_RING_ENABLED
                        and CLOSE_AGG_ENABLED and FOLD_ENABLED)
            nbytes = 0 if one_call else lib().dn_row_index_local_workspace_bytes(G, N, R, E)
            if one_call:
                num_wg = _num_cus(dev)
                # (a graph over 32 nodes: the same call builds the chunked tiles and their unit streams instead; which, is decided on the device)
                kper = close_chunks(N, num_wg) // num_wg if CLOSE_MULTI_ENABLED else 0
                tcap = int(lib().dn_fold_graph_tiles_multi_capacity(N, kper * num_wg)) if kper else 0
                nbytes = lib().dn_conv_index_workspace_bytes(G, N, R, E, num_wg, tcap)
            if one_call and nbytes:
                cap = int(lib().dn_close_units_capacity(max(G, tcap), E + N, num_wg))
                # the chunked form's eight tables out of ONE allocation (most batches never look at them: views are made on demand)
                mt_sizes = (kper * num_wg + 1, kper * num_wg + 1, tcap + 1, 12 * max(tcap, 1))
                mt_off, acc = [], 0
                for _ in range(2):
                    for n_ in mt_sizes:
                        mt_off.append(acc)
                        acc += (n_ + 3) // 4 * 4                        # (16-byte aligned pieces)
                mt_arena = e32(acc) if kper else None
                mt_ptr = [ctypes.c_void_p(mt_arena.data_ptr() + 4 * o) if kper else None for o in mt_off]

                def mt_views(k):
                    o = mt_off[4 * k:4 * k + 4]
                    return (mt_arena[o[0]:o[0] + mt_sizes[0]], mt_arena[o[1]:o[1] + mt_sizes[1]], mt_arena[o[2]:o[2] + mt_sizes[2]],
                            mt_arena[o[3]:o[3] + mt_sizes[3]].view(max(tcap, 1), 12))
                cus = []
                for _ in range(2):
                    cu = CloseUnits()
                    cu.num_wg, cu.num_nodes, cu.agg, cu.num_tiles, cu.order, cu.num_segments = num_wg, N, True, G, _close_order(num_wg), G
                    cu.unit_ptr, cu.units = e32(num_wg + 1), torch.empty((cap, 4), dtype=I32, device=dev)
                    cu.ent_row, cu.ent_mask = e32(E + N), e32(E + N)
                    cus.append(cu)
                Gw = 8 * SWEEP_WG_PER_GROUP
                want_sweep = SWEEP_ENABLED and E // 32 >= Gw * SWEEP_MIN_TILES_PER_WG
                S = int(1.10 * (E // 32 + 8 * R) / Gw) + 4 if want_sweep else 0
                sweeps = [torch.empty((Gw * S, 4), dtype=I32, device=dev) for _ in range(2)] if want_sweep else [None, None]
                chunk_cap = (E + N) // 256 + R + 3
                chunk_tab, chunk_pp = torch.empty((chunk_cap, 4), dtype=I32, device=dev), e32(R + 2)
                host_plan = (ctypes.c_int32 * 6)()
                ws = _ws(nbytes, dev)
                status = ctypes.c_int32(0)
                check(lib().dn_conv_index_build_i32(
                    G, N, R, E, ptr(node_ptr), ptr(edge_ptr), ptr(src), ptr(dst), ptr(etype), 1, float(edge_frac), ptr(row_in),
                    ptr(row_out), ptr(aux_f_ptr), ptr(aux_f_idx), ptr(aux_b_ptr), ptr(aux_b_idx), ptr(dst_ptr), ptr(dst_rows),
                    ptr(src_ptr), ptr(src_rows), counts, host_rel, host_modes, ctypes.byref(status), ptr(rel_dev), ptr(gt_bufs[0][0]),
                    ptr(gt_bufs[0][1]), ptr(gt_bufs[1][0]), ptr(gt_bufs[1][1]), host_absorb, num_wg, cus[0].order, cap, ptr(cus[0].unit_ptr),
                    ptr(cus[0].units), ptr(cus[0].ent_row), ptr(cus[0].ent_mask), ptr(cus[1].unit_ptr), ptr(cus[1].units),
                    ptr(cus[1].ent_row), ptr(cus[1].ent_mask), kper, tcap, *mt_ptr, SWEEP_WG_PER_GROUP, S, ptr(sweeps[0]), ptr(sweeps[1]), 256,
                    WGRAD_CHUNK_CAP, chunk_cap, ptr(chunk_tab), ptr(chunk_pp), host_plan, ptr(ws), ws.numel(), stream_ptr()),
                    "dn_conv_index_build_i32")
                if status.value == 0:
                    self.built_by = "local"
                    self._absorb = {d: (gt_bufs[k][0], gt_bufs[k][1], int(host_absorb[k])) for k, d in enumerate(("f", "b"))}
                    self.max_graph = (int(host_absorb[2]), int(host_absorb[3]))
                    # (the sweep tables were sized by a bound; the builder laid them out with the slots they need)
                    sw = [(sweeps[k][:Gw * host_plan[4 + k]], Gw * int(host_plan[4 + k])) if want_sweep and host_plan[4 + k] > 0 else None
                          for k in range(2)]
                    pre = (cus, sw, (chunk_tab, chunk_pp, int(host_plan[3])), int(host_plan[0]), int(host_plan[1]), mt_views, tcap)
            elif nbytes:
                ws = _ws(nbytes, dev)
                status = ctypes.c_int32(0)
                check(lib().dn_row_index_build_local_i32(G, N, R, E, ptr(node_ptr), ptr(edge_ptr), ptr(src), ptr(dst), ptr(etype),
                                                         1 if self_loop else 0, float(edge_frac), ptr(row_in), ptr(row_out),
                                                         ptr(aux_f_ptr), ptr(aux_f_idx), ptr(aux_b_ptr), ptr(aux_b_idx),
                                                         ptr(dst_ptr), ptr(dst_rows), ptr(src_ptr), ptr(src_rows), counts, host_rel,
                                                         host_modes, ctypes.byref(status), ptr(rel_dev), ptr(gt_bufs[0][0]),
                                                         ptr(gt_bufs[0][1]), ptr(gt_bufs[1][0]), ptr(gt_bufs[1][1]), host_absorb,
                                                         ptr(ws), ws.numel(), stream_ptr()),
                      "dn_row_index_build_local_i32")
                if status.value == 0:
                    self.built_by = "local"
                    self._absorb = {d: (gt_bufs[k][0], gt_bufs[k][1], int(host_absorb[k])) for k, d in enumerate(("f", "b"))}
                    self.max_graph = (int(host_absorb[2]), int(host_absorb[3]))
            if self.built_by == "local":                              # the batch itself, for the launches that take graphs whole
                self.raw = (src, dst, etype, node_ptr, edge_ptr)
            if self.built_by != "local":
                _check_edge_types(etype, R)
        if self.built_by == "general":
            nbytes = lib().dn_row_index_workspace_bytes(N, R, E)
            if nbytes == 0:
                check(-2, "dn_row_index_workspace_bytes")
            ws = _ws(nbytes, dev)
            check(lib().dn_row_index_build_i32(N, R, E, ptr(src), ptr(dst), ptr(etype), 1 if self_loop else 0, float(edge_frac),
                                               ptr(row_in), ptr(row_out), ptr(aux_f_ptr), ptr(aux_f_idx), ptr(aux_b_ptr),
                                               ptr(aux_b_idx), ptr(dst_ptr), ptr(dst_rows), ptr(src_ptr), ptr(src_rows), counts,
                                               host_rel, host_modes, ptr(ws), ws.numel(), stream_ptr()), "dn_row_index_build_i32")
        P, n_agg, n_tf, n_agg_e, n_tf_e = (int(v) for v in counts)
        self.modes = [int(m) for m in host_modes]
        rel_ptr = [int(v) for v in host_rel]
        P_all = P + (N if self_loop else 0)
        if self_loop:
            rel_ptr.append(P_all)
        self.num_aux_f, self.num_aux_b = n_agg, n_tf
        self.aux_f_ptr, self.aux_f_idx = aux_f_ptr[:n_agg + 1], aux_f_idx[:max(n_agg_e, 1)][:n_agg_e]
        self.aux_b_ptr, self.aux_b_idx = aux_b_ptr[:n_tf + 1], aux_b_idx[:max(n_tf_e, 1)][:n_tf_e]
        n_f = (E - n_agg_e) + n_agg + (N if self_loop else 0)      # forward list entries; the rest sits in the discard segment
        n_b = (E - n_tf_e) + n_tf + (N if self_loop else 0)
        self.dst_ptr, self.dst_rows = dst_ptr[:N + 1], dst_rows[:n_f]
        self.src_ptr, self.src_rows = src_ptr[:N + 1], src_rows[:n_b]
        self.row_in, self.row_out = row_in[:P_all], row_out[:P_all]
        self.num_rows, self.num_edge_rows = P_all, P
        self.num_all_rels = R + (1 if self_loop else 0)
        self.rel_ptr_host = rel_ptr
        # tile / chunk tables on the device (the 18 relation offsets go up in one small copy; no host loops)
        if self.built_by == "local":                                 # (already on the device: no upload)
            rel_ptr_d = rel_dev[:len(rel_ptr)]
        else:
            rel_ptr_d = torch.tensor(rel_ptr, dtype=I32).to(dev, non_blocking=True)
        self.rel_ptr_dev = rel_ptr_d                                # (reused by the fold tables: one upload per batch)
        self._tile_table = self._edge_tile_table = None             # built on first use: the folded bf16 path needs neither
        self._slots, self._units, self._fold = {}, {}, {}
        # one workgroup per CU (the LDS-DMA ring fills a CU's LDS) for the split-K weight gradient whatever the batch size:
        # the smallest chunk that keeps ALL relations' chunks (each relation ends in a partial one) within one round of 256
        if pre is not None:
            self.chunk_table = pre[2]
            if pre[3] and pre[4]:                                    # both directions served: nothing is left for the first step
                cands = {d: _fold_candidate(self, d) for d in ("f", "b")}
                assert cands["f"] is not None and cands["b"] is not None and cands["f"][3] == cands["b"][3] == G
                for k, d in enumerate(("f", "b")):
                    info = _make_fold_info(self, d, cands[d])
                    if pre[3 + k] == 2:                              # a graph over 32 nodes: the call left the chunked form behind
                        ct, cg, tpm, fim = pre[5](k)
                        info.graph_tiles = (tpm, fim)
                        info.multi = (ct, cg, pre[6], G)
                        pre[0][k].order, pre[0][k].num_tiles = 2 + _close_order(pre[0][k].num_wg), pre[6]
                    else:
                        info.graph_tiles = (self._absorb[d][0], self._absorb[d][1])
                    if pre[1][k] is not None and _sweep_wanted(self):
                        info.sweep_tiles = pre[1][k]
                    self._fold[d] = info
                    self._units[d] = pre[0][k]
        else:
            self.chunk_table = build_row_tables(rel_ptr_d, self.num_all_rels, P_all, wgrad_chunk_rows(rel_ptr), want_ptr=True)


def _conv_tiles(ix, fold, xs):
    """Tile table of the edge rows (minus the folded relation) for the conv's transform launch: the L2-blocked sweep order when
    the persistent H = 256 bf16 launch will walk it and the batch is large enough for the order to matter (built once per
    index and direction, on first use), else the plain relation-major tiles."""
    return _conv_tiles_for(ix, fold, xs.shape[1], xs.dtype)


def _sweep_wanted(ix):
    """Is the batch large enough for the L2-blocked order to matter (smaller launches fit the L2s anyway)?"""
    return (SWEEP_ENABLED and ix.num_rels <= 64
            and ix.num_edge_rows // 32 >= 8 * SWEEP_WG_PER_GROUP * SWEEP_MIN_TILES_PER_WG)


def _conv_tiles_for(ix, fold, H, dtype):
    P, R = ix.num_edge_rows, ix.num_rels
    if not (dtype == torch.bfloat16 and H == 256 and _sweep_wanted(ix)):
        if fold.main_tiles is None:                                  # (built on first use: a batch on the sweep order never needs them)
            fold.main_tiles = build_row_tables(ix.rel_ptr_dev, ix.num_rels, ix.num_edge_rows, 32, skip_mask=1 << fold.rel)
        return fold.main_tiles
    if fold.sweep_tiles is None:
        fold.sweep_tiles = build_sweep_tables(ix.rel_ptr_dev, R, ix.row_in, ix.row_out, ix.num_nodes, P, skip_mask=1 << fold.rel)
    return fold.sweep_tiles


def prepare_closing(ix, H, dtype):
    """Every table message_pass would build lazily on a batch's first step -- the closing tables of both directions (slot tables
    or unit streams, fold tables) and the sweep tile orders -- so that a loop can account the per-batch index cost outside the
    step (bench.py's index_build_ms, the overlapped fresh-batch leg)."""
    kind = "units" if (CLOSE_RING_ENABLED and H == 256 and dtype == torch.bfloat16) else "slots"
    _closing_tables(ix, kind)
    for d in ("f", "b"):
        if ix._fold[d] is not None:
            _conv_tiles_for(ix, ix._fold[d], H, dtype)


def _row_index_tile_table(ix):
    """Tile table of ALL rows (the self loop as relation R): the path without the fused closing launch."""
    if ix._tile_table is None:
        ix._tile_table = build_row_tables(ix.rel_ptr_dev, ix.num_all_rels, ix.num_rows, 32)
    return ix._tile_table


def _row_index_edge_tile_table(ix):
    """Tile table of the edge rows (relations 0 .. R-1): closing launch without a folded relation."""
    if not ix.self_loop:
        return _row_index_tile_table(ix)
    if ix._edge_tile_table is None:
        ix._edge_tile_table = build_row_tables(ix.rel_ptr_dev, ix.num_rels, ix.num_edge_rows, 32)
    return ix._edge_tile_table


RowIndex.tile_table = property(_row_index_tile_table)
RowIndex.edge_tile_table = property(_row_index_edge_tile_table)


def _fold_candidate(ix, direction):
    """(relation, first row, end row, #aux lists) of the one collapsed relation whose pre-aggregation the closing launch could
    absorb in this direction (AGG forward / TF backward), or None -- decided from what the host already knows."""
    mode = RowIndex.AGG if direction == "f" else RowIndex.TF
    n_aux = ix.num_aux_f if direction == "f" else ix.num_aux_b
    rels = [r for r, m in enumerate(ix.modes) if m == mode and ix.rel_ptr_host[r + 1] > ix.rel_ptr_host[r]]
    if not (FOLD_ENABLED and ix.self_loop and len(rels) == 1 and ix.num_rels <= 64 and n_aux > 0):
        return None
    r = rels[0]
    beg, end = ix.rel_ptr_host[r], ix.rel_ptr_host[r + 1]
    return (r, beg, end, n_aux) if end - beg == n_aux else None


# DN_CLOSE_AGG=0: keep the fp32 partial rows + dn_fold_tail_bf16 even where every graph fits one tile of the unit stream
CLOSE_AGG_ENABLED = _os.environ.get("DN_CLOSE_AGG", "1") != "0"


def _fold_add_idx(ix, direction, cand):
    """The node every segment's product is added to: the folded relation's rows' outputs (forward) / inputs (backward)."""
    return (ix.row_out if direction == "f" else ix.row_in)[cand[1]:cand[2]].contiguous()


def _make_fold_info(ix, direction, cand):
    r, beg, end, n_aux = cand
    info = _Fold()
    info.rel, info.beg, info.end, info.n = r, beg, end, n_aux
    info.fold_info = info.part_ptr = info.graph_tiles = info.multi = None
    info.num_parts = int(2 * n_aux + ix.num_nodes // 32 + 1)   # upper bound of part_ptr[-1] without a read-back: every segment
    #                                                            starts one partial row, every tile boundary inside one another
    info.main_tiles = None                               # plain relation-major tiles, built on first use (_conv_tiles_for)
    info.sweep_tiles = None                              # built on the first H = 256 launch (_conv_tiles)
    info.add_idx = _fold_add_idx(ix, direction, cand)
    return info


def _queue_fold_tables(ix, direction, n_aux, flag):
    """dn_fold_tables_build_async_i32 for one direction (32-node tiles, fp32 partial rows): -> (fold_info, part_ptr); verdict in flag."""
    N, dev = ix.num_nodes, ix.row_in.device
    aux_ptr, aux_idx = (ix.aux_f_ptr, ix.aux_f_idx) if direction == "f" else (ix.aux_b_ptr, ix.aux_b_idx)
    fold_info = torch.empty(((N + 31) // 32, 12), dtype=I32, device=dev)
    part_ptr = torch.empty(n_aux + 1, dtype=I32, device=dev)
    ws = _ws(lib().dn_fold_tables_workspace_bytes(n_aux), dev)
    check(lib().dn_fold_tables_build_async_i32(N, n_aux, ptr(aux_ptr), ptr(aux_idx), ptr(fold_info), ptr(part_ptr), ptr(flag),
                                               ptr(ws), ws.numel(), stream_ptr()), "dn_fold_tables_build_async_i32")
    return fold_info, part_ptr


def _late_fold_tables(ix, direction, info):
    """The 32-node-tile fold tables (fp32 partial rows) of a direction whose fold had been ABSORBED so far (a second closing kind on
    the same index).  The builder repeats the contiguity test the graph tiles passed, so its verdict must be "valid": it is read back
    and checked here instead of being handed on as a device flag that the slot builder honours and the overflow finishers (which drop
    the folded rows unconditionally) would not."""
    flag = torch.zeros(1, dtype=I32, device=ix.row_in.device)
    info.fold_info, info.part_ptr = _queue_fold_tables(ix, direction, info.n, flag)
    if int(flag.item()) == 0:
        raise _lib.DnHipError("fold tables of direction %r are invalid although the graph tiles of the same segments were valid" % direction)


def _closing_tables(ix, kind="slots"):
    """Tables of the closing launches of BOTH directions of a RowIndex.  Every builder leaves its verdict on the device; the host
    reads the verdicts of both directions in ONE copy (it picks the launch sequence by them).
      kind "units" (dn_rows_close_bf16, H = 256): first dn_fold_graph_tiles_build_i32 -- are the graphs small enough for the
        ABSORBED fold? -- one copy; only a direction that fails gets the 32-node-tile fold tables (dn_fold_tables_build_async_i32)
        and a second copy.  The unit streams are built behind the verdicts (their tiles depend on them).
      kind "slots" (dn_rows_selfsum_bf16): the fold tables and the slot tables (which read the verdict on the device) are queued
        in front of the one copy.
    A second kind on the same index reuses the verdicts and builds only what it misses."""
    have = ix._slots if kind == "slots" else ix._units
    if have:
        return
    N, P, dev, K = ix.num_nodes, ix.num_edge_rows, ix.row_in.device, SELFSUM_SLOTS
    first = not ix._fold
    lists = {d: tuple(t.to(I32).contiguous() for t in ((ix.dst_ptr, ix.dst_rows) if d == "f" else (ix.src_ptr, ix.src_rows)))
             for d in ("f", "b")}
    dirs = ("f", "b")

    def add_idx_of(direction, cand):
        return _fold_add_idx(ix, direction, cand)

    def make_info(direction, cand):
        return _make_fold_info(ix, direction, cand)

    tabs = {}
    if first:
        cands = {d: _fold_candidate(ix, d) for d in dirs}
        flags = torch.zeros(4, dtype=I32, device=dev)                # [parts_f, parts_b, graph_tiles_f, graph_tiles_b]
        gts, parts = {}, {}
        h = hp = [0, 0, 0, 0]
        if kind == "units" and CLOSE_AGG_ENABLED and ix._absorb is not None:
            # the graph-local index builder answered "can the fold be absorbed?" itself (same candidate rule, same test) and its
            # verdicts came back with its one read-back: no launch, no synchronisation here
            h = [0, 0, 0, 0]
            for k, d in enumerate(dirs):
                if cands[d] is not None:
                    tp, fi, verdict = ix._absorb[d]                      # (bit 0: every block within 32 nodes; bit 1: valid without that limit)
                    gts[d] = (tp[:cands[d][3] + 1], fi[:cands[d][3]])
                    h[2 + k] = verdict & 1
        elif kind == "units" and CLOSE_AGG_ENABLED:
            for k, d in enumerate(dirs):
                if cands[d] is not None:
                    aux_ptr, aux_idx = (ix.aux_f_ptr, ix.aux_f_idx) if d == "f" else (ix.aux_b_ptr, ix.aux_b_idx)
                    gts[d] = build_graph_tiles(aux_ptr[:cands[d][3] + 1].contiguous(), aux_idx, N, ok=flags[2 + k:],
                                               add_idx=add_idx_of(d, cands[d]))
            if gts:
                h = flags.cpu().tolist()                             # synchronisation 1: can the fold be absorbed?
        if not CLOSE_SINGLE_ENABLED:
            h = [h[0], h[1], 0, 0]
        need_parts = [d for k, d in enumerate(dirs) if cands[d] is not None and h[2 + k] == 0]
        multi = {}
        if kind == "units" and CLOSE_AGG_ENABLED and CLOSE_MULTI_ENABLED and need_parts:
            # a graph over 32 nodes: the absorbed fold over MULTI-TILE graphs (every graph's tiles in one workgroup's stream) where the
            # segments pass the same test without the size limit; one read-back for both directions
            mflags = torch.zeros(2, dtype=I32, device=dev)
            for k, d in enumerate(dirs):
                if d in need_parts:
                    aux_ptr, aux_idx = (ix.aux_f_ptr, ix.aux_f_idx) if d == "f" else (ix.aux_b_ptr, ix.aux_b_idx)
                    multi[d] = build_graph_tiles_multi(aux_ptr[:cands[d][3] + 1].contiguous(), aux_idx, N, ok=mflags[k:],
                                                       add_idx=add_idx_of(d, cands[d]))
            hm = mflags.cpu().tolist()
            multi = {d: multi[d] for k, d in enumerate(dirs) if d in multi and hm[k] != 0}
            need_parts = [d for d in need_parts if d not in multi]
        for d in need_parts:
            parts[d] = _queue_fold_tables(ix, d, cands[d][3], flags[dirs.index(d):])
        if kind == "slots":
            for k, d in enumerate(dirs):
                drop, enable = ((cands[d][1], cands[d][2]), flags[k:]) if cands[d] is not None else ((0, 0), None)
                tabs[d] = build_slot_table(*lists[d], N, P, K, drop=drop, drop_enable=enable)
        if need_parts:
            hp = flags.cpu().tolist()                                # synchronisation 2 (the only one for kind "slots")
        for k, d in enumerate(dirs):
            info = None
            if cands[d] is not None and h[2 + k] != 0:
                info = make_info(d, cands[d])
                info.graph_tiles = (gts[d][0], gts[d][1])
            elif d in multi:
                info = make_info(d, cands[d])
                info.graph_tiles = (multi[d][0], multi[d][1])
                info.multi = (multi[d][2], multi[d][3], multi[d][4], cands[d][3])
            elif cands[d] is not None and hp[k] != 0:
                info = make_info(d, cands[d])
                info.fold_info, info.part_ptr = parts[d]
            ix._fold[d] = info
    elif kind == "slots":
        for d in dirs:
            info = ix._fold[d]
            if info is not None and info.fold_info is None:          # absorbed so far: the slot kernel needs the partial-row tables
                _late_fold_tables(ix, d, info)
            drop = (info.beg, info.end) if info is not None else (0, 0)
            tabs[d] = build_slot_table(*lists[d], N, P, K, drop=drop)
    longest = {d: None for d in dirs}
    if kind == "slots" and 0 < N <= OVERFLOW_INSIDE_MAX_ROWS:
        # the longest list of each direction (one small read-back per batch, small batches only): rows_selfsum walks overflowing
        # lists inside the launch only while they are short
        mx = torch.stack([(lists[d][0][1:N + 1] - lists[d][0][:N]).max() for d in dirs]).tolist()
        longest = {d: int(mx[k]) for k, d in enumerate(dirs)}
    for d in dirs:
        info = ix._fold[d]
        drop = (info.beg, info.end) if info is not None else (0, 0)
        if (kind == "units" and info is not None and info.graph_tiles is None and CLOSE_AGG_ENABLED and CLOSE_SINGLE_ENABLED
                and ix._absorb is not None and (ix._absorb[d][2] & 1)):                                # (the slot tables came first: the builder's verdict still stands)
            info.graph_tiles = (ix._absorb[d][0][:info.n + 1], ix._absorb[d][1][:info.n])
        if kind == "slots":
            slots, over = tabs[d]
            ix._slots[d] = (slots, (*lists[d], P, drop[0], drop[1], over, longest[d]))
        elif info is not None and info.graph_tiles is not None:     # every graph inside one tile: the fold is absorbed
            ix._units[d] = build_close_units(*lists[d], N, P, drop=drop, tile_ptr=info.graph_tiles[0], agg=True, multi=info.multi)
        else:
            if info is not None and info.fold_info is None:
                _late_fold_tables(ix, d, info)
            ix._units[d] = build_close_units(*lists[d], N, P, drop=drop)


def _row_index_slots(ix, direction):
    """Slot tables of a RowIndex for the fused closing launch ('f': rows into each destination, 'b': rows out of each source).
    The rows of a FOLDED relation (_row_index_fold) are left out: they are added by the tail launches."""
    _closing_tables(ix, "slots")
    return ix._slots[direction]


def _row_index_close_units(ix, direction):
    """The same lists as the unit stream of dn_rows_close_bf16 (H = 256)."""
    _closing_tables(ix, "units")
    return ix._units[direction]


def _close_kind(x):
    """Which closing launch serves rows of this width: the unit stream (H = 256) or the slot kernel."""
    return "units" if (CLOSE_RING_ENABLED and x.shape[1] == 256 and x.dtype == torch.bfloat16) else "slots"


# The collapsed relation of a dummy-augmented batch (u -> dummy forward, dummy -> u backward) needs the SUM of a graph's rows as
# its input row.  As a launch of its own (gather_segsum over the aux lists) that is a second full read of x (config 5: 0.5 GB,
# ~100 us per direction); the closing launch reads every x row anyway, so it can produce those sums on the side.
FOLD_ENABLED = _os.environ.get("DN_FOLD", "1") != "0"


class _Fold:
    """Tables of one folded relation: rows [beg, end) of the row set, one per segment (graph)."""
    __slots__ = ("rel", "beg", "end", "n", "fold_info", "part_ptr", "num_parts", "main_tiles", "sweep_tiles", "add_idx",
                 "graph_tiles", "multi")


def _row_index_fold(ix, direction, kind="slots"):
    """The relation whose pre-aggregation the closing launch can absorb, or None: exactly ONE collapsed relation in this
    direction (AGG forward / TF backward), its aux lists contiguous ascending node ranges (a graph's nodes), bf16 self-loop path
    (checked on the device by dn_fold_tables_build_async_i32; the verdict comes back with the closing tables of `kind`)."""
    _closing_tables(ix, kind)
    return ix._fold[direction]


RowIndex.slots = _row_index_slots
RowIndex.close_units = _row_index_close_units


class RowIndexSet:
    """Holder of a batch's RowIndex (`parts` = [(first node, end node, RowIndex)], one part) and of the Y buffer its launches
    share.  (Rounds 1-2 could cut a batch into cache-resident sub-batches here; under-filled launches lost more than the
    Infinity Cache returned at every split -- DESIGN.md section 4 -- and the splitting was removed in round 3.)"""

    def __init__(self, src, dst, etype, num_nodes, num_rels, self_loop, node_ptr=None, edge_ptr=None, closing_hint=None):
        N = int(num_nodes)
        self.num_nodes, self.num_rels, self.self_loop = N, int(num_rels), bool(self_loop)
        self.parts = [(0, N, RowIndex(src, dst, etype, N, num_rels, self_loop=self_loop, node_ptr=node_ptr, edge_ptr=edge_ptr,
                                      closing_hint=closing_hint))]
        self.max_rows = max(ix.num_rows for _, _, ix in self.parts)
        self.num_rows = sum(ix.num_rows for _, _, ix in self.parts)
        self.num_all_rels = self.num_rels + (1 if self_loop else 0)
        self._ybuf = {}

    def ybuf(self, H, dtype, dev):
        key = (H, dtype, str(dev))
        b = self._ybuf.get(key)
        if b is None:
            b = torch.empty((self.max_rows, H), dtype=dtype, device=dev)
            self._ybuf[key] = b
        return b


def _selfsum_ok(ix, x):
    return SELFSUM_ENABLED and ix.self_loop and x.dtype == torch.bfloat16


def fold_tail(part, part_ptr, num_segments, Wn, idx, out, w_kn=False):
    """aux[j] = sum of partial rows part_ptr[j] .. part_ptr[j+1] in order;  out[idx[j]] += aux[j] @ Wn^T  (dn_fold_tail_bf16;
    w_kn: Wn given [in][out] instead).  Returns aux (bf16)."""
    require_gpu(part, part_ptr, Wn, idx, out)
    H = part.shape[1]
    assert part.dtype == torch.float32 and part_ptr.dtype == I32 and idx.dtype == I32 and idx.numel() == num_segments
    assert out.dtype == torch.bfloat16 and Wn.dtype == torch.bfloat16 and Wn.shape == (H, H) and out.shape[1] == H
    assert out.is_contiguous() and Wn.is_contiguous() and part.is_contiguous()
    aux = torch.empty((num_segments, H), dtype=torch.bfloat16, device=part.device)

    def _launch():
        check(lib().dn_fold_tail_bf16(ptr(part), ptr(part_ptr), int(num_segments), H, ptr(Wn), ptr(idx), ptr(aux), ptr(out),
                                      1 if w_kn else 0, stream_ptr()), "dn_fold_tail_bf16")
    if kernel_timer is not None:
        kernel_timer.launch("fold_tail", _launch)
    else:
        _launch()
    return aux


class PassWeights:
    """Weights of ONE direction of a message pass: rel [R, H, H] for the edge relations, loop [H, H] for the self loop (or None).
    kn = False: every matrix is [n][k] (rows = output columns, the form the MFMA kernels read contiguously: W^T for the forward
    pass, W itself for the input-gradient pass); kn = True: [k][n] -- the forward pass straight on the parameters `weight` /
    `loop_weight` as the reference stores them (rgin.py:61-67), which the H = 256 bf16 kernels gather themselves, so the step
    has no `cat` / `transpose().contiguous()` launches."""
    __slots__ = ("rel", "loop", "kn", "_all")

    def __init__(self, rel, loop, kn=False):
        self.rel, self.loop, self.kn, self._all = rel.contiguous(), (loop.contiguous() if loop is not None else None), bool(kn), None

    def all_nk(self):
        """[R (+1), n, k] in one contiguous tensor (loop last): the kernels without a kn mode (fp32, H != 256, no closing launch)."""
        if self._all is None:
            w = self.rel if self.loop is None else torch.cat([self.rel, self.loop.unsqueeze(0)], 0)
            self._all = w.transpose(1, 2).contiguous() if self.kn else w.contiguous()
        return self._all

    def nk(self):
        """The same weights with kn = False."""
        if not self.kn:
            return self
        a = self.all_nk()
        return PassWeights(a if self.loop is None else a[:-1], None if self.loop is None else a[-1], False)


# DN_KN_SMALL=0: at H = 64 / 128 (bf16) the weights are concatenated and transposed in front of the forward launches, as before round 5
KN_SMALL_ENABLED = _os.environ.get("DN_KN_SMALL", "1") != "0"


def _kn_ok(xs):
    """The launches that take [k][n] weights: bf16 H = 256 (ring transform, unit-stream closing launch, fold tail), bf16 H = 64 /
    128 (register-staged transform, slot kernel, fold tail) and the fp32 transform (any of its widths; it also takes the self-loop
    matrix and the bias where the parameters lie)."""
    if xs.dtype == torch.float32:
        return xs.shape[1] in (64, 128, 256)
    if xs.dtype == torch.bfloat16 and xs.shape[1] in (64, 128):
        return KN_SMALL_ENABLED                      # round 5: the register-staged transform and the slot kernel read [k][n] too
    return xs.dtype == torch.bfloat16 and xs.shape[1] == 256 and CLOSE_RING_ENABLED


def _closing_launch(xs, W_loop, bias, Y, ix, direction, out, seg=None, w_kn=False):
    """The closing launch over one RowIndex: the unit stream at H = 256 (dn_rows_close_bf16), else the slot kernel + its
    overflow launch (dn_rows_selfsum_bf16, dn_overflow_rows_add_bf16)."""
    if _close_kind(xs) == "units":
        return rows_close(xs, W_loop, bias, Y, ix.close_units(direction), out=out, seg=seg, w_kn=w_kn)
    slots, lists = ix.slots(direction)
    return rows_selfsum(xs, W_loop, bias, Y, None, slots, out=out, seg=seg, lists=lists, w_kn=w_kn)


def _message_pass_folded(xs, pw, bias, ix, direction, ybuf, out, idx_rows):
    """message_pass with the collapsed relation's pre-aggregation absorbed by the closing launch: transform of every other
    relation -> closing launch (+ per-graph column sums of xs) -> tail launch (combine the sums, transform the one row per graph,
    add each product to its node).  Where every graph fits one tile of the unit stream (H = 256) the closing launch does the
    tail's work itself (its AGG units): two launches per direction.  Same sums as the unfolded path up to bf16 rounding of the
    collapsed rows."""
    kind = _close_kind(xs)
    fold = _row_index_fold(ix, direction, kind)
    P, H = ix.num_edge_rows, xs.shape[1]
    Y = rows_transform(xs, pw.rel, _conv_tiles(ix, fold, xs), P, idx=idx_rows, tag="conv", out=ybuf, w_kn=pw.kn)
    if kind == "units" and ix.close_units(direction).agg:
        aux = torch.empty((fold.n, H), dtype=xs.dtype, device=xs.device)
        rows_close(xs, pw.loop, bias, Y[:P], ix.close_units(direction), out=out, w_kn=pw.kn,
                   agg=(fold.graph_tiles[1], pw.rel[fold.rel], aux, fold.add_idx))
        return aux
    part = torch.empty((fold.num_parts, H), dtype=torch.float32, device=xs.device)
    _closing_launch(xs, pw.loop, bias, Y[:P], ix, direction, out, seg=(fold.fold_info, part), w_kn=pw.kn)
    return fold_tail(part, fold.part_ptr, fold.n, pw.rel[fold.rel], fold.add_idx, out, w_kn=pw.kn)


# DN_CONV_GRAPHS=0: H = 64 bf16 batches of small graphs keep the row-factorised launches (transform, closing launch, fold tail)
CONV_GRAPHS_ENABLED = _os.environ.get("DN_CONV_GRAPHS", "1") != "0"


def conv_graphs_ok(xs, pw, ix):
    """Can dn_conv_graphs_bf16 take this pass?  bf16 rows of width 64, square weights with a self loop, a batch the graph-local
    index builder has seen (it reports the largest graph), every graph within 64 nodes / 1024 edges, at most 16 relations."""
    if not (CONV_GRAPHS_ENABLED and xs.dtype == torch.bfloat16 and xs.shape[1] == 64 and pw.loop is not None and ix.self_loop
            and ix.raw is not None and ix.max_graph is not None and ix.num_rels <= 16 and tuple(pw.rel.shape[1:]) == (64, 64)):
        return False
    return 0 < ix.max_graph[0] <= 64 and ix.max_graph[1] <= 1024 and ix.num_edges > 0       # (no edges: nothing to point the launch at)


def conv_graphs(xs, pw, bias, ix, direction, out):
    """One direction of the conv over ix's batch as ONE launch (dn_conv_graphs_bf16, one workgroup per graph, every relation edge
    by edge).  Returns the pre-aggregated rows the weight gradient's collapsed relation takes (per-graph column sums of xs, written
    by the same launch where the direction's aux lists are the graphs' segments; else one gather launch)."""
    src, dst, etype, node_ptr, edge_ptr = ix.raw
    key_in, key_out = (src, dst) if direction == "f" else (dst, src)
    aux_idx, aux_ptr, n_aux = (ix.aux_f_idx, ix.aux_f_ptr, ix.num_aux_f) if direction == "f" else (ix.aux_b_idx, ix.aux_b_ptr, ix.num_aux_b)
    G, N, H = int(node_ptr.numel()) - 1, ix.num_nodes, 64
    require_gpu(xs, pw.rel, pw.loop, bias, out)
    assert xs.is_contiguous() and out.is_contiguous() and out.shape == xs.shape and pw.rel.is_contiguous() and pw.loop.is_contiguous()
    if getattr(ix, "_cg_err", None) is None:
        ix._cg_err = torch.zeros(16, dtype=I32, device=xs.device)
    cand = _fold_candidate(ix, direction) if n_aux else None
    inside = (cand is not None and cand[3] == G and n_aux == G and ix._absorb is not None and ix._absorb[direction][2] != 0)
    aux = torch.empty((n_aux, H), dtype=xs.dtype, device=xs.device) if inside else None
    sp = aux_ptr[:n_aux + 1].contiguous() if inside else None

    def _launch():
        check(lib().dn_conv_graphs_bf16(ptr(xs), H, ptr(pw.rel), 1 if pw.kn else 0, ptr(pw.loop), ptr(bias), ix.num_rels, ptr(node_ptr),
                                        ptr(edge_ptr), ptr(key_in), ptr(key_out), ptr(etype), G, N, ptr(out), ptr(sp),
                                        ptr(aux_idx) if inside else None, ptr(aux), ptr(ix._cg_err), stream_ptr()),
              "dn_conv_graphs_bf16")
    if kernel_timer is not None:
        kernel_timer.launch("conv_graphs", _launch)
    else:
        _launch()
    if n_aux and not inside:
        aux = gather_segsum(xs, aux_idx, aux_ptr, n_aux)
    return aux


def _graphs_aux(ix, direction, xs):
    """(inside, aux tensor or None, seg_ptr or None, seg_nodes or None) for the whole-graph launches: the direction's aux lists are the
    graphs' segments (one collapsed relation, one row per graph) -> the launch writes the column sums itself."""
    aux_idx, aux_ptr, n_aux = (ix.aux_f_idx, ix.aux_f_ptr, ix.num_aux_f) if direction == "f" else (ix.aux_b_idx, ix.aux_b_ptr, ix.num_aux_b)
    G = int(ix.raw[3].numel()) - 1
    cand = _fold_candidate(ix, direction) if n_aux else None
    inside = (cand is not None and cand[3] == G and n_aux == G and ix._absorb is not None and ix._absorb[direction][2] != 0)
    if not inside:
        return False, None, None, None
    return True, torch.empty((n_aux, xs.shape[1]), dtype=xs.dtype, device=xs.device), aux_ptr[:n_aux + 1].contiguous(), aux_idx


def layer_graphs_fwd(x, W, W_loop, bias, w1, b1, w2, b2, slope, ix):
    """A whole RGIN layer forward on ix's batch of small graphs in ONE launch (dn_layer_graphs_fwd_bf16).
    -> (h conv rows, h1, h2, bits1, bits2, aux)."""
    src, dst, etype, node_ptr, edge_ptr = ix.raw
    G, N, H = int(node_ptr.numel()) - 1, ix.num_nodes, 64
    require_gpu(x, W, W_loop, bias, w1, b1, w2, b2)
    assert x.is_contiguous() and W.is_contiguous() and W_loop.is_contiguous() and w1.is_contiguous() and w2.is_contiguous()
    if getattr(ix, "_cg_err", None) is None:
        ix._cg_err = torch.zeros(16, dtype=I32, device=x.device)
    inside, aux, sp, sn = _graphs_aux(ix, "f", x)
    h, h1, h2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    bits1 = torch.empty((N, H // 8), dtype=torch.uint8, device=x.device)
    bits2 = torch.empty((N, H // 8), dtype=torch.uint8, device=x.device)

    def _launch():
        check(lib().dn_layer_graphs_fwd_bf16(ptr(x), H, ptr(W), ptr(W_loop), ptr(bias), ix.num_rels, ptr(w1), ptr(b1), ptr(w2), ptr(b2),
                                             float(slope), ptr(node_ptr), ptr(edge_ptr), ptr(src), ptr(dst), ptr(etype), G, N, ptr(h), ptr(h1),
                                             ptr(h2), ptr(bits1), ptr(bits2), ptr(sp), ptr(sn), ptr(aux), ptr(ix._cg_err), stream_ptr()),
              "dn_layer_graphs_fwd_bf16")
    if kernel_timer is not None:
        kernel_timer.launch("layer_graphs_fwd", _launch)
    else:
        _launch()
    if ix.num_aux_f and not inside:
        aux = gather_segsum(x, ix.aux_f_idx, ix.aux_f_ptr, ix.num_aux_f)
    return h, h1, h2, bits1, bits2, aux


def layer_graphs_bwd(g, W, W_loop, w1, w2, slope, bits1, bits2, ix):
    """... and its input gradients in ONE launch (dn_layer_graphs_bwd_bf16).  -> (g1, g0, gx, aux_b)."""
    src, dst, etype, node_ptr, edge_ptr = ix.raw
    G, N, H = int(node_ptr.numel()) - 1, ix.num_nodes, 64
    require_gpu(g, W, W_loop, w1, w2, bits1, bits2)
    assert g.is_contiguous() and W.is_contiguous() and W_loop.is_contiguous() and w1.is_contiguous() and w2.is_contiguous()
    g1, g0, gx = torch.empty_like(g), torch.empty_like(g), torch.empty_like(g)
    inside, aux_b, sp, sn = _graphs_aux(ix, "b", g)

    def _launch():
        check(lib().dn_layer_graphs_bwd_bf16(ptr(g), H, ptr(W), ptr(W_loop), ix.num_rels, ptr(w1), ptr(w2), float(slope), ptr(bits1),
                                             ptr(bits2), ptr(node_ptr), ptr(edge_ptr), ptr(src), ptr(dst), ptr(etype), G, N, ptr(g1), ptr(g0),
                                             ptr(gx), ptr(sp), ptr(sn), ptr(aux_b), ptr(ix._cg_err), stream_ptr()),
              "dn_layer_graphs_bwd_bf16")
    if kernel_timer is not None:
        kernel_timer.launch("layer_graphs_bwd", _launch)
    else:
        _launch()
    if ix.num_aux_b and not inside:
        aux_b = gather_segsum(g0, ix.aux_b_idx, ix.aux_b_ptr, ix.num_aux_b)
    return g1, g0, gx, aux_b


def message_pass(xs, pw, bias, ix, direction, ybuf, out, add_in=None):
    """One direction of the row-factorised pass over one RowIndex -- the launches that ARE the layer's gather-scatter:
         'f':  out[v] = sum_{rows p -> v} (in_row(p) @ W[rel p])          pw = the weights (PassWeights), [k][n] or W^T
         'b':  out[u] = sum_{rows p <- u} (g_row(p)  @ W[rel p]^T)        pw = W per relation as it is ([n = in][k = out])
       = pre-aggregation of the collapsed relations (gather_segsum) -> gathered-row transform on the matrix cores ->
       closing launch.  With a self loop in bf16 the closing launch is dn_rows_close_bf16 / dn_rows_selfsum_bf16 (self-loop
       transform + bias + per-node sum of the edge rows in one pass); otherwise the self-loop rows go through the transform like
       any relation and a per-node gather_segsum closes.  Returns the pre-aggregated rows (kept for the weight gradient)."""
    if direction == "f":
        aux_idx, aux_ptr, n_aux, idx_rows, lst, lptr = ix.aux_f_idx, ix.aux_f_ptr, ix.num_aux_f, ix.row_in, ix.dst_rows, ix.dst_ptr
    else:
        aux_idx, aux_ptr, n_aux, idx_rows, lst, lptr = ix.aux_b_idx, ix.aux_b_ptr, ix.num_aux_b, ix.row_out, ix.src_rows, ix.src_ptr
    f32_direct = xs.dtype == torch.float32 and _kn_ok(xs) and pw.rel.shape[1] == pw.rel.shape[2]
    assert add_in is None or f32_direct        # (add_in: rows added to the result by the closing gather -- the fp32 path only)
    if conv_graphs_ok(xs, pw, ix):
        return conv_graphs(xs, pw, bias, ix, direction, out)
    if pw.kn and not (f32_direct or (_kn_ok(xs) and _selfsum_ok(ix, xs))):
        pw = pw.nk()
    if _selfsum_ok(ix, xs) and _row_index_fold(ix, direction, _close_kind(xs)) is not None:
        return _message_pass_folded(xs, pw, bias, ix, direction, ybuf, out, idx_rows)
    aux = gather_segsum(xs, aux_idx, aux_ptr, n_aux) if n_aux else None
    if _selfsum_ok(ix, xs):
        P = ix.num_edge_rows
        Y = (rows_transform(xs, pw.rel, ix.edge_tile_table, P, idx=idx_rows, X2=aux, tag="conv", out=ybuf, w_kn=pw.kn)
             if P else ybuf[:0])
        _closing_launch(xs, pw.loop, bias, Y[:P], ix, direction, out, w_kn=pw.kn)
        return aux
    if f32_direct:
        # fp32: the transform reads weight / loop_weight / h_bias where the parameters lie (no cat, no transposed copy, no padded
        # bias matrix); the self-loop rows are relation R of the tile table
        R = pw.rel.shape[0]
        Y = rows_transform(xs, pw.rel, ix.tile_table, ix.num_rows, idx=idx_rows, X2=aux, bias=bias, tag="conv", out=ybuf,
                           w_kn=pw.kn, W_loop=pw.loop, loop_rel=R, bias_rel=R if bias is not None else -1)
        gather_segsum(Y, lst, lptr, ix.num_nodes, out=out, self_in=add_in, self_coef=1.0 if add_in is not None else 0.0)
        return aux
    Wmat = pw.all_nk()
    bias_all = None
    if bias is not None:
        bias_all = torch.zeros((Wmat.shape[0], Wmat.shape[1]), dtype=xs.dtype, device=xs.device)
        bias_all[-1] = bias                                                  # only self-loop rows carry the bias
    Y = rows_transform(xs, Wmat, ix.tile_table, ix.num_rows, idx=idx_rows, X2=aux, bias=bias_all, tag="conv", out=ybuf)
    gather_segsum(Y, lst, lptr, ix.num_nodes, out=out)
    return aux


class _RowTransformFn(torch.autograd.Function):
    """out = sum over in-edges of x[src] @ W[etype]  (+ x @ W_loop + bias when the index has the self loop), evaluated
    over the parts of a RowIndexSet (one part).  W [R, in, out] and W_loop [in, out] are the layer's own tensors: nothing is
    concatenated or transposed on the bf16 H = 256 path."""

    @staticmethod
    def forward(ctx, x, W, W_loop, bias, index_set):
        ctx.f32_mode = f32_mode()
        x = x.contiguous()
        H_out = W.shape[2]
        pw = PassWeights(W, W_loop, kn=True)
        if not _kn_ok(x):
            pw = pw.nk()                                                     # [R', out, in], one cat + transposed copy
        out = torch.empty((x.shape[0], H_out), dtype=x.dtype, device=x.device)
        ybuf = index_set.ybuf(H_out, x.dtype, x.device)
        auxs = []
        for n0, n1, ix in index_set.parts:
            aux = message_pass(x[n0:n1], pw, bias, ix, "f", ybuf, out[n0:n1])
            auxs.append(aux if aux is not None else x.new_empty(0))      # a few MB: kept for the weight gradient
        ctx.index_set, ctx.has_bias, ctx.has_loop = index_set, bias is not None, W_loop is not None
        ctx.save_for_backward(x, W, W_loop if W_loop is not None else x.new_empty(0), *auxs)
        return out

    @staticmethod
    @_backward_in_forward_mode
    def backward(ctx, g):
        iset = ctx.index_set
        g = g.contiguous()
        x, W, W_loop = ctx.saved_tensors[:3]
        auxs = ctx.saved_tensors[3:]
        R_all = W.shape[0] + (1 if ctx.has_loop else 0)
        pw = PassWeights(W, W_loop if ctx.has_loop else None, kn=False)     # the input-gradient pass reads W as it is
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3])
        gx = torch.empty_like(x) if need_x else None
        gW32 = cs32 = None
        ybuf = iset.ybuf(W.shape[1], g.dtype, g.device)
        single = len(iset.parts) == 1
        for part, (n0, n1, ix) in enumerate(iset.parts):
            gs, xs = g[n0:n1], x[n0:n1]
            if need_x:
                aux_b = message_pass(gs, pw, None, ix, "b", ybuf, gx[n0:n1])
            else:
                aux_b = gather_segsum(gs, ix.aux_b_idx, ix.aux_b_ptr, ix.num_aux_b) if ix.num_aux_b else None
            if need_w:
                aux = auxs[part] if ix.num_aux_f else None
                # bias gradient = column sum of g over the self-loop rows (one per node), folded into the same kernel
                gw, cs = rows_wgrad(xs, gs, ix.chunk_table, R_all, idx_a=ix.row_in, idx_g=ix.row_out, A2=aux,
                                    G2=aux_b, out_dtype=W.dtype if single else torch.float32, colsum_of=2, colsum_lp=single,
                                    colsum_rel=R_all - 1)                  # (gb = cs32[-1] below: the other relations' sums are not used)
                gW32 = gw if gW32 is None else gW32.add_(gw)
                cs32 = cs if cs32 is None else cs32.add_(cs)
        gW = gL = gb = None
        if need_w:
            gAll = gW32.to(W.dtype)                                          # [R (+1), in, out]: the two gradients are views of it
            gW = gAll[:W.shape[0]]
            if ctx.has_loop:
                gL = gAll[W.shape[0]]
            if ctx.has_bias:
                gb = cs32[-1].to(g.dtype)
        return gx, gW, gL, gb, None


class _BddDenseFn(torch.autograd.Function):
    """Block-diagonal relation weights [R, B * si * so] -> dense [R, B * si, B * so] (dn_bdd_compose), gradient = the diagonal
    blocks of the dense gradient (dn_bdd_extract): one launch each way (rgin.py:114-120's per-block products as dense ones)."""

    @staticmethod
    def forward(ctx, weight, R, B, si, so):
        w = weight.contiguous()
        require_gpu(w)
        assert w.numel() == R * B * si * so and w.dtype in (torch.bfloat16, torch.float32)
        dense = torch.empty((R, B * si, B * so), dtype=w.dtype, device=w.device)
        check(lib().dn_bdd_compose(ptr(w), R, B, si, so, w.element_size(), ptr(dense), stream_ptr()), "dn_bdd_compose")
        ctx.dims, ctx.shape = (R, B, si, so), weight.shape
        return dense

    @staticmethod
    def backward(ctx, g):
        R, B, si, so = ctx.dims
        g = g.contiguous()
        gb = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
        check(lib().dn_bdd_extract(ptr(g), R, B, si, so, g.element_size(), ptr(gb), stream_ptr()), "dn_bdd_extract")
        return gb, None, None, None, None


def bdd_dense(weight, num_rels, num_bases, submat_in, submat_out):
    """Dense [R, B * si, B * so] view of block-diagonal relation weights, differentiable (GPU, bf16 / fp32)."""
    return _BddDenseFn.apply(weight, int(num_rels), int(num_bases), int(submat_in), int(submat_out))


def fused_path_supported(x, W):
    """bf16 (storage bf16 / fp32 accumulate) and fp32 (exact-f32 MFMA) run the row-factorised matrix-core pipeline when the
    layer is square with H in {64, 128, 256}; everything else takes the generic two-pass path."""
    return (x.is_cuda and x.dtype == W.dtype and x.dtype in MFMA_DTYPES and W.dim() == 3
            and W.shape[1] == W.shape[2] and W.shape[1] in (64, 128, 256) and x.shape[1] == W.shape[1])


def rel_transform_fused(x, W, bias, index_set, W_loop=None):
    """Fused row-factorised path.  Either W = [R, in, out] with the self-loop weight given separately (W_loop [in, out] -- the
    layer's own parameters, nothing concatenated), or, without W_loop, W = [R (+1), in, out] with the self-loop weight LAST when
    index_set.self_loop.  bias is added on the self-loop rows (requires the self loop).  index_set: RowIndexSet (or a RowIndex)."""
    if isinstance(index_set, RowIndex):
        one = index_set
        index_set = RowIndexSet.__new__(RowIndexSet)
        index_set.num_nodes, index_set.num_rels, index_set.self_loop = one.num_nodes, one.num_rels, one.self_loop
        index_set.parts, index_set.max_rows, index_set.num_rows = [(0, one.num_nodes, one)], one.num_rows, one.num_rows
        index_set.num_all_rels, index_set._ybuf = one.num_all_rels, {}
    assert bias is None or index_set.self_loop
    if W_loop is None and index_set.self_loop:
        assert W.shape[0] == index_set.num_all_rels
        W, W_loop = W[:-1], W[-1]                                            # views: their gradients flow back into W's
    assert W.shape[0] == index_set.num_rels and (W_loop is not None) == index_set.self_loop
    return _RowTransformFn.apply(x, W, W_loop, bias, index_set)


# ----------------------------------------------------------------------------------------------
# dense Linear(+ReLU) of the post-aggregate MLP on the same MFMA kernels (bf16)
# ----------------------------------------------------------------------------------------------
_dense_tables = {}


def _dense_table(n_rows, dev):
    """Tile / chunk tables of a single-relation, identity-indexed row set (cached per size and device)."""
    key = (int(n_rows), str(dev))
    t = _dense_tables.get(key)
    if t is None:
        # split-K chunks sized for ONE round of 256 workgroups whatever the row count (a 20 k-row GC batch with 4096-row chunks
        # would run its weight gradient on five workgroups; 378 chunks of a 1 M-row batch would run one and a half rounds)
        # (at least 128 rows a chunk, not 256: a 16 k-row GC batch then has 125 workgroups walking 4 tiles each instead of 63 walking
        #  8 -- GIN steps under replay 0.334 -> 0.318 / 0.636 -> 0.625 / 0.812 -> 0.787 ms; 64 rows: the H = 256 partial tiles cost more)
        chunk = max(128, min(WGRAD_CHUNK_CAP, -(-int(n_rows) // 256 // 64) * 64))
        t = (make_row_tiles([0, int(n_rows)], dev), make_row_chunks([0, int(n_rows)], dev, chunk_rows=chunk))
        if len(_dense_tables) > 8:
            _dense_tables.clear()
        _dense_tables[key] = t
    return t


class _LinearActFn(torch.autograd.Function):
    """y = relu?(x @ weight^T + bias) with weight [out, in] (nn.Linear layout): forward and both backward products on
    dn_rows_transform_bf16 / dn_rows_wgrad_bf16, bias and ReLU fused in the forward epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, exact=None):
        x = x.contiguous()
        tiles, _ = _dense_table(x.shape[0], x.device)
        # exact: None = module default, True / False, or "fwd" = exact products where errors propagate (forward, input
        # gradient), the 3-term split for the weight gradient (a leaf: its 1e-5 relative error goes nowhere)
        ctx.exact = f32_mode() if exact is None else bool(exact)
        ctx.exact_w = False if exact == "fwd" else ctx.exact
        with f32_exact(ctx.exact):
            y = rows_transform(x, weight.contiguous().unsqueeze(0), tiles, x.shape[0],
                               bias=None if bias is None else bias.contiguous().view(1, -1), relu=relu)
        ctx.relu, ctx.has_bias = bool(relu), bias is not None
        ctx.save_for_backward(x, weight, y if relu else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, g):
        with f32_exact(ctx.exact):
            return _LinearActFn._backward(ctx, g) + (None,)

    @staticmethod
    def _backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        g = g.contiguous()
        tiles, chunks = _dense_table(x.shape[0], x.device)
        gx = gw = gb = None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.relu and not need_w:
            g = relu_bwd(g, y)
        if need_w:
            # g^T x -> [out, in] and colsum(g); the ReLU mask is applied while the rows are staged and the masked rows are
            # written out for the input-gradient launch (no separate elementwise pass)
            gm = torch.empty_like(g) if ctx.relu else None
            with f32_exact(ctx.exact_w):
                gw, cs = rows_wgrad(g, x, chunks, 1, out_dtype=weight.dtype, colsum_of=1, mask_a=y if ctx.relu else None,
                                    a_out=gm)
            g = gm if ctx.relu else g
            gw = gw[0]
            if ctx.has_bias:
                gb = cs[0].to(g.dtype)
        if ctx.needs_input_grad[0]:
            kn = g.dtype == torch.float32                       # the fp32 kernels read Linear.weight [out = k][in = n] as it is
            gx = rows_transform(g, (weight if kn else weight.t()).contiguous().unsqueeze(0), tiles, x.shape[0], w_kn=kn)  # g @ weight
        return gx, gw, gb, None


def relu_bwd(g, y, slope=0.0):
    """(y > 0) ? g : slope * g  (dn_relu_bwd_bf16 / _f32; slope 0 = ReLU)."""
    require_gpu(g, y)
    out = torch.empty_like(g)
    fn = lib().dn_relu_bwd_f32 if g.dtype == torch.float32 else lib().dn_relu_bwd_bf16
    check(fn(ptr(g), ptr(y), ptr(out), g.numel(), float(slope), stream_ptr()), "dn_relu_bwd")
    return out


class _ReluMlpFn(torch.autograd.Function):
    """y_L = act(lin_L(... act(lin_1(x)))) with every Linear followed by the activation -- ReLU (slope 0) or leaky ReLU (the
    reference MLP + final activation, rgin.py:50-57,147-151, for act_func "relu" and for its CLI default "leaky_relu", slope
    1 / 5.5: config.py:329-335, utils/act.py:466).
    Two layers in bf16 (the reference default): ONE forward launch (dn_rows_chain2_bf16) that also emits both activation masks
    as bit tensors (sign of the output = sign of the pre-activation for either activation), and three backward launches: weight
    gradient of layer 2 (outer mask applied from its bits while the rows are staged), the whole input-gradient chain (mask,
    dgrad 2, mask, dgrad 1), weight gradient of layer 1.
    Otherwise: one fused Linear+bias+activation launch per layer forward; per layer backward a weight/bias-gradient launch and
    an input-gradient launch whose epilogue applies the activation mask of the layer below."""

    @staticmethod
    def forward(ctx, x, slope, *wb):
        ctx.f32_mode = f32_mode()
        n = len(wb) // 2
        x = x.contiguous()
        ctx.slope = slope = float(slope)
        if n == 2 and CHAIN2_ENABLED and x.dtype == torch.bfloat16:
            h1, h2, bits1, bits2 = rows_chain2(x, wb[0], wb[1], True, wb[2], wb[3], True, want_bits=True, slope=slope)
            ctx.n, ctx.chain = n, True
            ctx.has_bias = [wb[1] is not None, wb[3] is not None]
            ctx.save_for_backward(x, h1, bits1, bits2, wb[0], wb[2])      # the output itself is not kept: only its sign bits
            return h2
        if (n == 2 and CHAIN2_F32_ENABLED and x.dtype == torch.float32 and x.shape[1] in (64, 128) and not f32_mode()
                and all(t is None or t.dtype == torch.float32 for t in wb)):
            # fp32 (the reference's precision) at H = 64 / 128: both Linears in one launch each way (dn_rows_chain2_f32); the saved
            # activations are the masks
            h1, h2 = rows_chain2_f32(x, wb[0], wb[1], True, wb[2], wb[3], True, slope=slope)
            ctx.n, ctx.chain = n, "f32"
            ctx.has_bias = [wb[1] is not None, wb[3] is not None]
            ctx.save_for_backward(x, h1, h2, wb[0], wb[2])
            return h2
        tiles, _ = _dense_table(x.shape[0], x.device)
        acts = [x]
        for i in range(n):
            w, b = wb[2 * i], wb[2 * i + 1]
            acts.append(rows_transform(acts[-1], w.contiguous().unsqueeze(0), tiles, x.shape[0],
                                       bias=None if b is None else b.contiguous().view(1, -1), relu=True, slope=slope))
        ctx.n, ctx.chain = n, False
        ctx.has_bias = [wb[2 * i + 1] is not None for i in range(n)]
        ctx.save_for_backward(*acts, *[wb[2 * i] for i in range(n)])
        return acts[-1]

    @staticmethod
    @_backward_in_forward_mode
    def backward(ctx, gout):
        n, slope = ctx.n, ctx.slope
        saved = ctx.saved_tensors
        g = gout.contiguous()
        grads = [None] * (2 + 2 * n)
        if ctx.chain == "f32":
            x0, h1, h2, w1, w2 = saved
            _, chunks = _dense_table(x0.shape[0], x0.device)
            gw2, cs2 = rows_wgrad(g, h1, chunks, 1, out_dtype=w2.dtype, colsum_of=1, mask_a=h2, slope=slope)
            g1, g0 = rows_chain2_f32(g, w2, None, False, w1, None, False, mask0=h2, mask1=h1, w_kn=(True, True), slope=slope)
            gw1, cs1 = rows_wgrad(g1, x0, chunks, 1, out_dtype=w1.dtype, colsum_of=1)
            grads[2], grads[4] = gw1[0], gw2[0]
            if ctx.has_bias[0]:
                grads[3] = cs1[0].to(g.dtype)
            if ctx.has_bias[1]:
                grads[5] = cs2[0].to(g.dtype)
            grads[0] = g0 if ctx.needs_input_grad[0] else None
            return tuple(grads)
        if ctx.chain:
            x0, h1, bits1, bits2, w1, w2 = saved
            _, chunks = _dense_table(x0.shape[0], x0.device)
            gw2, cs2 = rows_wgrad(g, h1, chunks, 1, out_dtype=w2.dtype, colsum_of=1, mask_a_bits=bits2, colsum_lp=True, slope=slope)
            g1, g0 = rows_chain2(g, w2, None, False, w1, None, False, mask0_bits=bits2, mask1_bits=bits1, w_kn=(True, True),
                                 slope=slope)
            gw1, cs1 = rows_wgrad(g1, x0, chunks, 1, out_dtype=w1.dtype, colsum_of=1, colsum_lp=True)
            grads[2], grads[4] = gw1[0], gw2[0]
            if ctx.has_bias[0]:
                grads[3] = cs1[0].to(g.dtype)
            if ctx.has_bias[1]:
                grads[5] = cs2[0].to(g.dtype)
            grads[0] = g0 if ctx.needs_input_grad[0] else None
            return tuple(grads)
        acts, ws = saved[:n + 1], saved[n + 1:]
        tiles, chunks = _dense_table(acts[0].shape[0], acts[0].device)
        for i in range(n - 1, -1, -1):
            w = ws[i]
            if i == n - 1:
                # outermost activation: masked while the rows are staged for the weight gradient, masked rows saved for the
                # input-gradient launch -- no separate elementwise pass
                gm = torch.empty_like(g)
                gw, cs = rows_wgrad(g, acts[i], chunks, 1, out_dtype=w.dtype, colsum_of=1, mask_a=acts[n], a_out=gm, slope=slope)
                g = gm
            else:
                gw, cs = rows_wgrad(g, acts[i], chunks, 1, out_dtype=w.dtype, colsum_of=1)   # g^T a_{i} ; colsum(g)
            grads[2 + 2 * i] = gw[0]
            if ctx.has_bias[i]:
                grads[3 + 2 * i] = cs[0].to(g.dtype)
            if i > 0 or ctx.needs_input_grad[0]:
                kn = g.dtype == torch.float32                   # fp32 kernels read Linear.weight [out = k][in = n] as it is
                g = rows_transform(g, (w if kn else w.t()).contiguous().unsqueeze(0), tiles, g.shape[0],
                                   mask_pos=acts[i] if i > 0 else None, slope=slope, w_kn=kn)   # masked for the activation below
        grads[0] = g if ctx.needs_input_grad[0] else None
        return tuple(grads)


# DN_LAYER_GRAPHS=0: inside _RginLayerSmallFn the conv and the MLP chain stay separate launches (dn_conv_graphs_bf16 + dn_rows_chain2_bf16)
LAYER_GRAPHS_ENABLED = _os.environ.get("DN_LAYER_GRAPHS", "1") != "0"
# DN_LAYER_SMALL=0: an H = 64 bf16 RGIN layer on small graphs stays a chain of separate autograd functions (conv, MLP)
LAYER_SMALL_ENABLED = _os.environ.get("DN_LAYER_SMALL", "1") != "0"


def rgin_layer_small_ok(x, W, W_loop, bias, linears, index_set):
    """Can a whole RGIN layer (conv + bias + 2-layer MLP + activations) run as _RginLayerSmallFn?  bf16, H = 64, square, self loop,
    two square Linears, a batch dn_conv_graphs_bf16 takes."""
    if not (LAYER_SMALL_ENABLED and CHAIN2_ENABLED and W_loop is not None and len(linears) == 2 and len(index_set.parts) == 1):
        return False
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 64 and W.dtype == x.dtype
            and all(l.weight.dtype == x.dtype and tuple(l.weight.shape) == (64, 64) for l in linears)):
        return False
    return conv_graphs_ok(x, PassWeights(W, W_loop, kn=True), index_set.parts[0][2])


class _RginLayerSmallFn(torch.autograd.Function):
    """act(lin2(act(lin1(conv(x))))) of an RGIN layer at the reference's default width on a batch of small graphs (rgin.py:102-160 +
    50-57, BASELINE config 3) as SEVEN launches a step: forward = the conv (dn_conv_graphs_bf16) + the MLP chain
    (dn_rows_chain2_bf16); backward = the chain's input gradients, the conv's input gradient (the same launch on the transposed
    weights), ONE weight-gradient launch for the conv's R + 1 matrices and both Linears (dn_rows_wgrad_multi_bf16) + its reduce --
    where the separate autograd functions need 11 launches plus the casts between them."""

    @staticmethod
    def forward(ctx, x, slope, index_set, W, W_loop, bias, w1, b1, w2, b2):
        ctx.f32_mode = f32_mode()
        x = x.contiguous()
        ix = index_set.parts[0][2]
        if LAYER_GRAPHS_ENABLED:
            h, h1, h2, bits1, bits2, aux = layer_graphs_fwd(x, W.contiguous(), W_loop.contiguous(), bias, w1.contiguous(), b1,
                                                            w2.contiguous(), b2, float(slope), ix)
        else:
            h = torch.empty_like(x)
            aux = conv_graphs(x, PassWeights(W, W_loop, kn=True), bias, ix, "f", h)
            h1, h2, bits1, bits2 = rows_chain2(h, w1, b1, True, w2, b2, True, want_bits=True, slope=float(slope))
        ctx.ix, ctx.slope = ix, float(slope)
        ctx.has = (bias is not None, b1 is not None, b2 is not None, aux is not None)
        ctx.save_for_backward(x, h, h1, bits1, bits2, W, W_loop, w1, w2, aux if aux is not None else x.new_empty(0))
        return h2

    @staticmethod
    @_backward_in_forward_mode
    def backward(ctx, gout):
        ix, slope = ctx.ix, ctx.slope
        x, h, h1, bits1, bits2, W, W_loop, w1, w2, aux = ctx.saved_tensors
        g = gout.contiguous()
        N, H, R = x.shape[0], 64, W.shape[0]
        if LAYER_GRAPHS_ENABLED:
            g1, g0, gx, aux_b = layer_graphs_bwd(g, W.contiguous(), W_loop.contiguous(), w1.contiguous(), w2.contiguous(), slope, bits1, bits2,
                                                 ix)
        else:
            g1, g0 = rows_chain2(g, w2, None, False, w1, None, False, mask0_bits=bits2, mask1_bits=bits1, w_kn=(True, True), slope=slope)
            gx = torch.empty_like(x)
            aux_b = conv_graphs(g0, PassWeights(W, W_loop, kn=False), None, ix, "b", gx)
        # the conv's rows (relation-major, the self loop as relation R), then the two Linears' dense rows, in one virtual row space
        if getattr(ix, "_layer_chunks", None) is None:
            P_all = ix.num_rows
            vptr_host = list(ix.rel_ptr_host) + [P_all + N, P_all + 2 * N]
            vptr = torch.cat([ix.rel_ptr_dev[:R + 2], torch.tensor([P_all + N, P_all + 2 * N], dtype=I32, device=x.device)])
            ix._layer_chunks = build_row_tables(vptr, R + 3, P_all + 2 * N, wgrad_chunk_rows(vptr_host, _SMALL_WG), want_ptr=True)
        jobs = [dict(A=x, A2=aux if ctx.has[3] else None, idx_a=ix.row_in, G=g0, G2=aux_b, idx_g=ix.row_out, colsum_of=2, first_rel=0, row0=0),
                dict(A=g1, G=h, colsum_of=1, first_rel=R + 1, row0=ix.num_rows),
                dict(A=g, G=h1, colsum_of=1, mask_a_bits=bits2, slope=slope, first_rel=R + 2, row0=ix.num_rows + N)]
        gw, cs = rows_wgrad_multi(jobs, ix._layer_chunks, R + 3, H, W.dtype)
        return (gx, None, None, gw[:R], gw[R], cs[R] if ctx.has[0] else None, gw[R + 1], cs[R + 1] if ctx.has[1] else None, gw[R + 2],
                cs[R + 2] if ctx.has[2] else None)


# DN_LAYER_WIDE=0: an H = 256 bf16 RGIN layer stays a chain of separate autograd functions (conv, MLP), three weight-gradient launches
LAYER_WIDE_ENABLED = _os.environ.get("DN_LAYER_WIDE", "1") != "0"
# ... as it does above this many rows (the conv's + the two Linears'): every weight gradient of the layer is HBM-bound, so ONE launch
# saves the fixed costs of two (partial tiles written and reduced, ring fill and drain: 40 us of a 0.53-ms step on an eighth of
# config 5 -- one rank's share at 8 GPUs) and nothing else; on the whole batch (6.2 M rows) the mixed launch measured 25-55 us SLOWER
# than the three -- the gathered rows' share of Infinity-Cache hits does not survive the Linears' streams (docs/LAB_NOTES.md)
WIDE_LAYER_MAX_ROWS = int(_os.environ.get("DN_WIDE_MAX_ROWS", str(2 << 20)))


WIDE_DENSE_WEIGHT = float(_os.environ.get("DN_WIDE_DENSE_WEIGHT", "2"))


def wide_layer_chunks(rel_ptr_host, n_nodes, device, workgroups=256, dense_weight=None):
    """Split-K chunk table of _RginLayerWideFn's ONE weight-gradient launch: the conv's relation-major rows (rel_ptr_host: R + 1
    relations, the self loop last), then the two Linears' n_nodes rows each, in one virtual row space.  A workgroup's time is its
    tile count x what a tile of its kind costs, and a tile of rows in row order -- streamed from HBM, no second reader -- costs about
    twice a tile of gathered rows (which the L2s and the Infinity Cache serve in part): the Linears' chunks hold 1 / dense_weight of
    the rows of a conv chunk, and the chunk size is the smallest for which everything fits ONE round of `workgroups`.
    -> (chunks [C, 4] int32, chunk_ptr [R + 4] int32, C), as make_row_chunks."""
    w = float(dense_weight if dense_weight is not None else WIDE_DENSE_WEIGHT)
    conv = [int(b) - int(a) for a, b in zip(rel_ptr_host[:-1], rel_ptr_host[1:])]
    P_all, n = int(rel_ptr_host[-1]), int(n_nodes)
    dense_step = lambda c: max(64, int(c / w) // 32 * 32)                   # noqa: E731
    count = lambda c: sum(-(-m // c) for m in conv if m > 0) + 2 * (-(-n // dense_step(c)))   # noqa: E731
    c = max(256, -(-int((sum(conv) + 2 * w * n) // workgroups) // 64) * 64)
    while c < WGRAD_CHUNK_CAP and count(c) > workgroups:
        c += 64
    c = min(c, WGRAD_CHUNK_CAP)
    steps = [c] * len(conv) + [dense_step(c)] * 2
    vptr = [int(v) for v in rel_ptr_host] + [P_all + n, P_all + 2 * n]
    chunks, cptr = [], [0]
    for r, step in enumerate(steps):
        a, b = vptr[r], vptr[r + 1]
        while a < b:
            e = min(a + step, b)
            chunks.append((r, a, e, 0))
            a = e
        cptr.append(len(chunks))
    ch = torch.tensor(chunks if chunks else [(0, 0, 0, 0)], dtype=I32).reshape(-1, 4)
    return ch.to(device), torch.tensor(cptr, dtype=I32).to(device), len(chunks)


def rgin_layer_wide_ok(x, W, W_loop, bias, linears, index_set):
    """Can a whole RGIN layer run as _RginLayerWideFn?  bf16, H = 256, square, self loop, two square Linears, one index part on the
    closing-launch path (the benchmarked configuration: BASELINE config 5)."""
    if not (LAYER_WIDE_ENABLED and CHAIN2_ENABLED and W_loop is not None and len(linears) == 2 and len(index_set.parts) == 1):
        return False
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256 and W.dtype == x.dtype
            and tuple(W.shape[1:]) == (256, 256) and x.shape[0] > 0
            and all(l.weight.dtype == x.dtype and tuple(l.weight.shape) == (256, 256) for l in linears)):
        return False
    ix = index_set.parts[0][2]
    return _selfsum_ok(ix, x) and 0 < ix.num_rows and ix.num_rows + 2 * x.shape[0] <= WIDE_LAYER_MAX_ROWS


class _RginLayerWideFn(torch.autograd.Function):
    """act(lin2(act(lin1(conv(x))))) of an RGIN layer at H = 256 in bf16 (rgin.py:102-160 + 50-57, BASELINE config 5) with the
    launches of the separate functions (_RowTransformFn: transform + closing launch per direction; _ReluMlpFn: one chain launch per
    direction) and ONE weight-gradient launch for the conv's R + 1 matrices and both Linears (dn_rows_wgrad_multi_bf16 at H = 256)
    at the end of the backward pass: the Linears' rows, which no one reads twice, stream from HBM under the conv's matrix work on
    rows the L2s serve, and one round of partial tiles is written and reduced instead of three."""

    @staticmethod
    def forward(ctx, x, slope, index_set, W, W_loop, bias, w1, b1, w2, b2):
        ctx.f32_mode = f32_mode()
        x = x.contiguous()
        ix = index_set.parts[0][2]
        pw = PassWeights(W, W_loop, kn=True)
        if not _kn_ok(x):
            pw = pw.nk()
        h = torch.empty_like(x)
        aux = message_pass(x, pw, bias, ix, "f", index_set.ybuf(256, x.dtype, x.device), h)
        h1, h2, bits1, bits2 = rows_chain2(h, w1, b1, True, w2, b2, True, want_bits=True, slope=float(slope))
        ctx.index_set, ctx.slope = index_set, float(slope)
        ctx.has = (bias is not None, b1 is not None, b2 is not None, aux is not None)
        ctx.save_for_backward(x, h, h1, bits1, bits2, W, W_loop, w1, w2, aux if aux is not None else x.new_empty(0))
        return h2

    @staticmethod
    @_backward_in_forward_mode
    def backward(ctx, gout):
        iset, slope = ctx.index_set, ctx.slope
        ix = iset.parts[0][2]
        x, h, h1, bits1, bits2, W, W_loop, w1, w2, aux = ctx.saved_tensors
        g = gout.contiguous()
        N, H, R = x.shape[0], 256, W.shape[0]
        g1, g0 = rows_chain2(g, w2, None, False, w1, None, False, mask0_bits=bits2, mask1_bits=bits1, w_kn=(True, True), slope=slope)
        gx = torch.empty_like(x)
        aux_b = message_pass(g0, PassWeights(W, W_loop, kn=False), None, ix, "b", iset.ybuf(H, g.dtype, g.device), gx)
        # the conv's rows (relation-major, the self loop as relation R), then the two Linears' dense rows, in one virtual row space
        if getattr(ix, "_layer_chunks", None) is None:
            ix._layer_chunks = wide_layer_chunks(list(ix.rel_ptr_host)[:R + 2], N, x.device)
        # (bias gradient = column sum of g0 over the self-loop rows: relation R of the first job)
        jobs = [dict(A=x, A2=aux if ctx.has[3] else None, idx_a=ix.row_in, G=g0, G2=aux_b, idx_g=ix.row_out,
                     colsum_of=2 | ((R + 1) << 8), first_rel=0, row0=0),
                dict(A=g1, G=h, colsum_of=1, first_rel=R + 1, row0=ix.num_rows),
                dict(A=g, G=h1, colsum_of=1, mask_a_bits=bits2, slope=slope, first_rel=R + 2, row0=ix.num_rows + N)]
        gw, cs = rows_wgrad_multi(jobs, ix._layer_chunks, R + 3, H, W.dtype)
        return (gx if ctx.needs_input_grad[0] else None, None, None, gw[:R], gw[R], cs[R] if ctx.has[0] else None, gw[R + 1],
                cs[R + 1] if ctx.has[1] else None, gw[R + 2], cs[R + 2] if ctx.has[2] else None)


def rgin_layer_wide(x, W, W_loop, bias, linears, slope, index_set):
    """The whole RGIN layer of rgin_layer_wide_ok's case as one autograd function (_RginLayerWideFn)."""
    return _RginLayerWideFn.apply(x, float(slope), index_set, W, W_loop, bias, linears[0].weight, linears[0].bias,
                                  linears[1].weight, linears[1].bias)


LAYER_F32_ENABLED = _os.environ.get("DN_LAYER_F32", "1") != "0"
CHAIN2_F32_ENABLED = _os.environ.get("DN_CHAIN2_F32", "1") != "0"          # 0: the fp32 layer function keeps one launch per Linear


def rgin_layer_f32_ok(x, W, W_loop, bias, linears, index_set):
    """Can a whole fp32 RGIN layer run as _RginLayerF32Fn?  The reference's precision on the bf16 split (not the exact-f32 mode),
    H = 64 / 128, square, self loop, two square Linears, one part."""
    if not (LAYER_F32_ENABLED and W_loop is not None and len(linears) == 2 and len(index_set.parts) == 1 and not f32_mode()):
        return False
    H = x.shape[1] if x.dim() == 2 else 0
    ok = lambda t: t is None or (t.is_cuda and t.dtype == torch.float32)  # noqa: E731
    return (x.is_cuda and x.dtype == torch.float32 and H in (64, 128) and x.shape[0] > 0 and tuple(W.shape[1:]) == (H, H)
            and W.dtype == x.dtype and tuple(W_loop.shape) == (H, H) and ok(W_loop) and ok(bias)
            and all(ok(l.weight) and ok(l.bias) and tuple(l.weight.shape) == (H, H) for l in linears)
            and index_set.parts[0][2].num_rows > 0 and _kn_ok(x))


class _RginLayerF32Fn(torch.autograd.Function):
    """act(lin2(act(lin1(conv(x))))) of an RGIN layer in the reference's own precision (fp32 on the 3-term bf16 split) at H = 64 / 128
    (rgin.py:102-160 + 50-57; BASELINE config 3 as the reference runs it) as ONE autograd function: the forward launches are those of
    _RowTransformFn + _ReluMlpFn; the backward masks the incoming gradient once (dn_relu_bwd_f32), runs the two input-gradient launches
    (the inner mask in the first one's epilogue), the conv's input-gradient pass, and then ONE weight-gradient launch + ONE reduce for
    the conv's R + 1 matrices and both Linears (dn_rows_wgrad_multi_f32) where the separate functions take three of each."""

    @staticmethod
    def forward(ctx, x, slope, index_set, W, W_loop, bias, w1, b1, w2, b2, residual=False):
        ctx.f32_mode = f32_mode()
        x = x.contiguous()
        ix = index_set.parts[0][2]
        N, H = x.shape
        slope = float(slope)
        h = torch.empty_like(x)
        aux = message_pass(x, PassWeights(W, W_loop, kn=True), bias, ix, "f", index_set.ybuf(H, x.dtype, x.device), h)
        ctx.residual = bool(residual)
        out = None
        if CHAIN2_F32_ENABLED and residual:             # x + layer(x): the sum leaves the MLP launch next to the activation
            h1, h2, out = rows_chain2_f32(h, w1, b1, True, w2, b2, True, slope=slope, residual=x)
        elif CHAIN2_F32_ENABLED:
            h1, h2 = rows_chain2_f32(h, w1, b1, True, w2, b2, True, slope=slope)
        else:
            tiles, _ = _dense_table(N, x.device)
            h1 = rows_transform(h, w1.contiguous().unsqueeze(0), tiles, N, bias=None if b1 is None else b1.contiguous().view(1, -1), relu=True,
                                slope=slope)
            h2 = rows_transform(h1, w2.contiguous().unsqueeze(0), tiles, N, bias=None if b2 is None else b2.contiguous().view(1, -1), relu=True,
                                slope=slope)
        ctx.index_set, ctx.slope = index_set, slope
        ctx.has = (bias is not None, b1 is not None, b2 is not None, aux is not None)
        ctx.save_for_backward(x, h, h1, h2, W, W_loop, w1, w2, aux if aux is not None else x.new_empty(0))
        if residual:
            if out is None:
                out = h2 + x
            return out
        return h2

    @staticmethod
    @_backward_in_forward_mode
    def backward(ctx, gout):
        iset, slope = ctx.index_set, ctx.slope
        ix = iset.parts[0][2]
        x, h, h1, h2, W, W_loop, w1, w2, aux = ctx.saved_tensors
        N, H, R = x.shape[0], x.shape[1], W.shape[0]
        g = gout.contiguous()
        if CHAIN2_F32_ENABLED:
            # outer mask, dgrad 2, inner mask, dgrad 1 in ONE launch; the weight gradient of Linear 2 masks g itself (mask = the saved output)
            gm1, g0 = rows_chain2_f32(g, w2, None, False, w1, None, False, mask0=h2, mask1=h1, w_kn=(True, True), slope=slope)
            job2 = dict(A=g, G=h1, colsum_of=1, mask_a_bits=h2, slope=slope)
        else:
            tiles, _ = _dense_table(N, x.device)
            gm2 = relu_bwd(g, h2, slope)                                    # the outer activation's mask
            gm1 = rows_transform(gm2, w2.contiguous().unsqueeze(0), tiles, N, mask_pos=h1, slope=slope, w_kn=True)   # masked for the inner one
            g0 = rows_transform(gm1, w1.contiguous().unsqueeze(0), tiles, N, w_kn=True)
            job2 = dict(A=gm2, G=h1, colsum_of=1)
        gx = torch.empty_like(x)
        # (residual: the gradient of x + layer(x) is g + the layer's input gradient -- g rides in the conv's closing gather)
        aux_b = message_pass(g0, PassWeights(W, W_loop, kn=False), None, ix, "b", iset.ybuf(H, x.dtype, x.device), gx,
                             add_in=g if ctx.residual else None)
        # the conv's rows (relation-major, the self loop as relation R), then the two Linears' dense rows, in one virtual row space
        if getattr(ix, "_layer_chunks", None) is None:
            P_all = ix.num_rows
            vptr_host = list(ix.rel_ptr_host) + [P_all + N, P_all + 2 * N]
            vptr = torch.cat([ix.rel_ptr_dev[:R + 2], torch.tensor([P_all + N, P_all + 2 * N], dtype=I32, device=x.device)])
            ix._layer_chunks = build_row_tables(vptr, R + 3, P_all + 2 * N, wgrad_chunk_rows(vptr_host, _SMALL_WG), want_ptr=True)
        jobs = [dict(A=x, A2=aux if ctx.has[3] else None, idx_a=ix.row_in, G=g0, G2=aux_b, idx_g=ix.row_out, colsum_of=2, first_rel=0, row0=0),
                dict(A=gm1, G=h, colsum_of=1, first_rel=R + 1, row0=ix.num_rows),
                dict(first_rel=R + 2, row0=ix.num_rows + N, **job2)]
        gw, cs = rows_wgrad_multi(jobs, ix._layer_chunks, R + 3, H, torch.float32)
        return (gx, None, None, gw[:R], gw[R], cs[R] if ctx.has[0] else None, gw[R + 1], cs[R + 1] if ctx.has[1] else None, gw[R + 2],
                cs[R + 2] if ctx.has[2] else None, None)


def rgin_layer_f32(x, W, W_loop, bias, linears, slope, index_set, residual=False):
    """residual: return x + layer(x) (the representation nets' residual connection) from the same launches."""
    return _RginLayerF32Fn.apply(x, float(slope), index_set, W, W_loop, bias, linears[0].weight, linears[0].bias, linears[1].weight,
                                 linears[1].bias, bool(residual))


def rgin_layer_small(x, W, W_loop, bias, linears, slope, index_set):
    return _RginLayerSmallFn.apply(x, float(slope), index_set, W, W_loop, bias, linears[0].weight, linears[0].bias, linears[1].weight,
                                   linears[1].bias)


def relu_mlp_supported(x, linears):
    return (x.is_cuda and x.dtype in MFMA_DTYPES and x.dim() == 2 and x.shape[0] > 0 and len(linears) > 0
            and all(l.weight.dtype == x.dtype and l.weight.shape[0] == l.weight.shape[1] == x.shape[1]
                    and x.shape[1] in (64, 128, 256) for l in linears))


def relu_mlp(x, linears, slope=0.0):
    """act(lin_L(... act(lin_1(x)))) through _ReluMlpFn; slope 0 = ReLU, > 0 = leaky ReLU."""
    wb = []
    for l in linears:
        wb += [l.weight, l.bias]
    return _ReluMlpFn.apply(x, float(slope), *wb)


def linear_act(x, weight, bias=None, relu=False, exact=None):
    """nn.Linear (+ ReLU) on the MFMA kernels when supported (bf16 / fp32, square 64/128/256), else torch.  exact: force the
    exact-f32 MFMA (True) or the bf16 split (False) for fp32 operands; None = the module default (ops.F32_EXACT)."""
    if (x.is_cuda and x.dtype == weight.dtype and x.dtype in MFMA_DTYPES and x.dim() == 2
            and weight.shape[0] == weight.shape[1] and weight.shape[0] in (64, 128, 256) and x.shape[1] == weight.shape[1]
            and x.shape[0] > 0):
        return _LinearActFn.apply(x, weight, bias, relu, exact)
    y = torch.nn.functional.linear(x, weight, bias)
    return torch.relu(y) if relu else y


# ----------------------------------------------------------------------------------------------
# dense side of the GC models: BatchNorm over the nodes of the batch, Linear of any width
# ----------------------------------------------------------------------------------------------
class _BatchNormRowsFn(torch.autograd.Function):
    """Training-mode BatchNorm over the rows of [N, C] (dn_batchnorm_rows_*): returns (y, mean, biased var)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, running_mean=None, running_var=None, momentum=0.0, relu=False, batches_tracked=None):
        x = x.contiguous()
        N, C = x.shape
        if batches_tracked is not None:
            require_gpu(batches_tracked)
            assert batches_tracked.dtype == torch.int64 and batches_tracked.numel() == 1
        if running_mean is not None:
            assert running_mean.dtype == torch.float32 and running_var.dtype == torch.float32
            assert running_mean.is_contiguous() and running_var.is_contiguous() and running_mean.numel() == C
        w32 = weight.detach().float().contiguous() if weight is not None else None
        b32 = bias.detach().float().contiguous() if bias is not None else None
        y = torch.empty_like(x)
        mean, var, rstd = (torch.empty(C, dtype=torch.float32, device=x.device) for _ in range(3))
        ws = _ws(lib().dn_batchnorm_rows_workspace_bytes(N, C), x.device)
        check(getattr(lib(), "dn_batchnorm_rows_" + _suffix(x))(ptr(x), N, C, ptr(w32), ptr(b32), float(eps), ptr(y), ptr(mean), ptr(var),
                                                               ptr(rstd), ptr(running_mean), ptr(running_var), float(momentum),
                                                               1 if relu else 0, ptr(batches_tracked), ptr(ws), ws.numel(),
                                                               stream_ptr()),
              "dn_batchnorm_rows")
        ctx.relu = bool(relu)
        ctx.save_for_backward(x, mean, rstd, w32 if w32 is not None else x.new_empty(0),
                              b32 if (relu and b32 is not None) else x.new_empty(0))
        ctx.has_w, ctx.has_b = weight is not None, bias is not None
        ctx.wdtype = weight.dtype if weight is not None else None
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)                      # (no zero tensors for the statistics outputs' "gradients": 2 fills a call)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, mean, rstd, w32, b32 = ctx.saved_tensors
        if dy is None:
            return (None,) * 9
        dy = dy.contiguous()
        N, C = x.shape
        dx = torch.empty_like(x)
        s1, s2 = (torch.empty(C, dtype=torch.float32, device=x.device) for _ in range(2))
        ws = _ws(lib().dn_batchnorm_rows_workspace_bytes(N, C), x.device)
        check(getattr(lib(), "dn_batchnorm_rows_bwd_" + _suffix(x))(ptr(dy), ptr(x), N, C, ptr(mean), ptr(rstd),
                                                                   ptr(w32) if ctx.has_w else None,
                                                                   ptr(b32) if (ctx.relu and ctx.has_b) else None,
                                                                   1 if ctx.relu else 0, ptr(dx), ptr(s1), ptr(s2), ptr(ws),
                                                                   ws.numel(), stream_ptr()), "dn_batchnorm_rows_bwd")
        gw = s2.to(ctx.wdtype) if ctx.has_w else None
        gb = s1.to(ctx.wdtype if ctx.has_w else dy.dtype) if ctx.has_b else None
        return dx, gw, gb, None, None, None, None, None, None


def batch_norm_rows_supported(x):
    return (x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[0] >= 1
            and x.shape[1] % 4 == 0 and 4 <= x.shape[1] <= 1024)


def batch_norm_rows(x, weight, bias, eps=1e-5, running_mean=None, running_var=None, momentum=0.0, relu=False, batches_tracked=None):
    """(y, batch mean, biased batch variance) of training-mode BatchNorm over the rows of x; relu: y = ReLU(BatchNorm(x)) in the
    same launches, forward and backward.  running_mean / running_var (fp32 buffers, optional) are updated in place by the
    statistics launch: r = (1 - momentum) r + momentum * new, unbiased variance; batches_tracked (int64 [1], optional): + 1."""
    return _BatchNormRowsFn.apply(x, weight, bias, eps, running_mean, running_var, momentum, relu, batches_tracked)


_single_rel_tables = {}


def _single_rel_table(n_rows, dev):
    """Device tile / chunk tables of ONE relation of n_rows rows for the any-width products (cached per size and device)."""
    key = (int(n_rows), str(dev))
    t = _single_rel_tables.get(key)
    if t is None:
        rp = torch.tensor([0, int(n_rows)], dtype=I32).to(dev)
        # (at least 64 rows a chunk: the any-width weight gradient walks a chunk in dependent 16-row steps -- load, barrier, multiply --
        #  so a 128-row chunk was eight round trips of latency: GIN steps under replay 0.355 -> 0.334 ms / 0.647 -> 0.636 / 0.838 -> 0.816)
        chunk = max(64, min(2048, -(-int(n_rows) // 256 // 64) * 64))
        t = (build_row_tables(rp, 1, n_rows, 64), build_row_tables(rp, 1, n_rows, chunk, want_ptr=True))
        if len(_single_rel_tables) > 8:
            _single_rel_tables.clear()
        _single_rel_tables[key] = t
    return t


class _LinearAnyFn(torch.autograd.Function):
    """y = x @ weight^T (+ bias) for any in / out widths on dn_rows_gemm_* / dn_rows_wgrad_any_* (weight [out, in], nn.Linear
    layout).  The library GEMMs torch picks for a [20 k x 5] x [5 x 128] product and its [128 x 20 k] x [20 k x 5] weight
    gradient take 20-80 us each; these are single small launches."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.f32_mode = f32_mode()
        x = x.contiguous()
        tiles, _ = _single_rel_table(x.shape[0], x.device)
        y = rows_gemm(x, weight.contiguous().unsqueeze(0), tiles, transpose_w=True,          # W[0] is [N, K] = [out, in]
                      bias=None if bias is None else bias.contiguous().view(1, -1))
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @_backward_in_forward_mode
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        tiles, chunks = _single_rel_table(x.shape[0], x.device)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = rows_gemm(g, weight.contiguous().unsqueeze(0), tiles)                       # g [P, out] @ W [out, in]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or need_b:                                                # g^T x -> [out, in]; colsum(g) -> bias
            gw, cs = rows_wgrad_any(g, x, chunks, 1, want_colsum=True)
            gw = gw[0]
            if need_b:
                gb = cs[0].to(g.dtype)
        return gx, gw, gb


def linear_any(x, weight, bias=None, exact=None):
    """nn.Linear on the HIP path whatever its widths: matrix cores when square 64 / 128 / 256, the any-width kernels otherwise.
    exact: fp32 arithmetic of the matrix-core path (None = module default, True = exact f32, False = bf16 split)."""
    if (x.is_cuda and x.dim() == 2 and x.shape[0] > 0 and x.dtype == weight.dtype and x.dtype in MFMA_DTYPES):
        if weight.shape[0] == weight.shape[1] and weight.shape[0] in (64, 128, 256) and x.shape[1] == weight.shape[1]:
            return _LinearActFn.apply(x, weight, bias, False, exact)
        return _LinearAnyFn.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)
