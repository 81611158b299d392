"""Host logic of the persistent message-pass launch (ops.PipeIndex -> dn_rows_pipe_bf16), on the CPU: the work tables are
executed by a small SIMULATOR of the kernel's protocol (roles walking their tile sequences, the done / cdone counters, the ring
of batch slots that is overwritten every `depth` batches) and must (1) never deadlock, (2) never overwrite a ring slot that
still has unread rows, (3) reproduce  out[v] = x[v] W_self + b + sum_{e: dst(e) = v} x[src(e)] W[etype(e)]  exactly (the self
loop travels through the ring as one more relation).
No GPU code runs here (the table builder only uses torch index ops + numpy); the GPU parity tests are in test_gpu_pipe.py."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch


def _fake_row_index(src, dst, et, N, R):
    """A RowIndex look-alike on the CPU with every edge as its own row (relation-major, stable), self-loop rows appended:
    exactly the fields PipeIndex reads."""
    order = np.argsort(et, kind="stable")
    P = len(src)
    row_in = np.concatenate([src[order], np.arange(N)]).astype(np.int32)
    row_out = np.concatenate([dst[order], np.arange(N)]).astype(np.int32)
    rel_ptr = np.concatenate([[0], np.cumsum(np.bincount(et, minlength=R))]).tolist() + [P + N]

    def lists(key):
        k = np.concatenate([key[order], np.arange(N)])
        o = np.argsort(k, kind="stable")
        ptr = np.concatenate([[0], np.cumsum(np.bincount(k, minlength=N))])
        return torch.from_numpy(ptr.astype(np.int32)), torch.from_numpy(o.astype(np.int32))

    dp, dr = lists(dst)
    sp, sr = lists(src)
    return SimpleNamespace(num_nodes=N, num_edge_rows=P, num_rows=P + N, num_rels=R, num_all_rels=R + 1, self_loop=True,
                           rel_ptr_host=rel_ptr, row_in=torch.from_numpy(row_in),
                           row_out=torch.from_numpy(row_out), dst_ptr=dp, dst_rows=dr, src_ptr=sp, src_rows=sr), order


def _random_batch(rng, G, R, n_lo, n_hi, empty_frac=0.1):
    node_ptr, src, dst, et = [0], [], [], []
    for _ in range(G):
        n = int(rng.integers(n_lo, n_hi + 1))
        m = 0 if rng.random() < empty_frac else int(rng.integers(1, 3 * n + 1))
        base = node_ptr[-1]
        src.extend((base + rng.integers(0, n, size=m)).tolist())
        dst.extend((base + rng.integers(0, n, size=m)).tolist())
        et.extend(rng.integers(0, R, size=m).tolist())
        node_ptr.append(base + n)
    return (np.asarray(node_ptr, dtype=np.int64), np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64),
            np.asarray(et, dtype=np.int64))


def simulate(pipe, direction, x, W, bias):
    """Run the tables the way the kernel does.  Roles advance one tile at a time in round-robin; a role blocks on the same
    conditions as the kernel.  Returns out (float64) or raises on deadlock / ring misuse."""
    tiles = pipe.tiles.numpy()
    roles = pipe.roles.numpy()
    bt = pipe.batches.numpy()
    row_idx = pipe.row_idx[direction].numpy()
    lptr, lloc = pipe.list_ptr[direction].numpy(), pipe.list_local[direction].numpy()
    N, H = x.shape
    B = pipe.num_batches
    ring = np.full((pipe.ring_rows, H), np.nan)
    ring_owner = np.full(pipe.ring_rows, -1)          # batch whose rows currently sit in a ring row
    done, cdone = np.zeros(B, dtype=np.int64), np.zeros(B, dtype=np.int64)
    out = np.full((N, H), np.nan)
    cur = {r: int(roles[r, 1]) for r in range(roles.shape[0]) if roles[r, 0] in (0, 1)}
    seen_rows, seen_nodes = np.zeros(pipe.ix.num_rows, dtype=np.int64), np.zeros(N, dtype=np.int64)
    pending = {}
    progress = True
    while progress:
        progress = False
        for r in list(cur):
            t = cur[r]
            if t >= roles[r, 2]:
                if pending.get(r, -1) >= 0:
                    done[pending[r]] += 1
                    pending[r] = -1
                    progress = True
                del cur[r]
                continue
            beg, end, b, rf = (int(v) for v in tiles[t])
            rel, first, last = rf & 0xffff, (rf >> 16) & 1, (rf >> 17) & 1
            rowbase, ringoff, need_c, wait_b, need_t = (int(v) for v in bt[b][:5])
            if roles[r, 0] == 0:
                if pending.get(r, -1) >= 0:                      # the kernel signals a unit one tile late, before it may block
                    done[pending[r]] += 1
                    pending[r] = -1
                    progress = True
                if first and wait_b >= 0 and cdone[wait_b] < bt[wait_b][2]:
                    continue                                     # ring slot still being read
                for p in range(beg, end):
                    rr = ringoff + (p - rowbase)
                    assert 0 <= p - rowbase < pipe.slot_rows
                    if ring_owner[rr] >= 0:
                        assert cdone[ring_owner[rr]] >= bt[ring_owner[rr]][2], "ring row overwritten before it was consumed"
                    ring[rr] = x[row_idx[p]] @ W[rel]
                    ring_owner[rr] = b
                    seen_rows[p] += 1
                if last:
                    pending[r] = b
            else:
                if first and done[b] < need_t:
                    continue
                assert done[b] >= need_t, "closing tile ran before its batch was complete"
                for v in range(beg, end):
                    acc = np.zeros(H) + (bias if bias is not None else 0.0)
                    for i in range(lptr[v], lptr[v + 1]):
                        rr = ringoff + int(lloc[i])
                        assert ring_owner[rr] == b, "list entry points outside its batch"
                        acc = acc + ring[rr]
                    out[v] = acc
                    seen_nodes[v] += 1
                cdone[b] += 1
            cur[r] = t + 1
            progress = True
    assert not cur, "deadlock: %d roles stuck" % len(cur)
    assert (seen_rows == 1).all() and (seen_nodes == 1).all()
    assert (cdone == bt[:B, 2]).all() and (done == bt[:B, 4]).all()
    return out


@pytest.mark.parametrize("seed,G,R,n_lo,n_hi,kw", [
    (0, 40, 5, 3, 12, dict(batch_nodes=24, depth=2, roles_per_group=8)),          # many tiny batches, shallow ring
    (1, 9, 3, 20, 90, dict(batch_nodes=64, depth=3, roles_per_group=6)),          # graphs larger than a batch / a tile
    (2, 200, 16, 31, 31, dict(batch_nodes=384, depth=6, roles_per_group=64)),     # config-5-shaped, production parameters
    (3, 5, 4, 1, 3, dict(batch_nodes=8, depth=2, roles_per_group=4)),             # fewer graphs than groups
    (4, 30, 12, 4, 9, dict(batch_nodes=32, depth=2, roles_per_group=5)),          # more relations than roles
])
def test_pipe_tables_execute_correctly(seed, G, R, n_lo, n_hi, kw):
    from dummynode4graphlearning_amd import ops
    rng = np.random.default_rng(seed)
    node_ptr, src, dst, et = _random_batch(rng, G, R, n_lo, n_hi)
    N, H = int(node_ptr[-1]), 4
    ix, order = _fake_row_index(src, dst, et, N, R)
    pipe = ops.PipeIndex(ix, torch.from_numpy(node_ptr), **kw)
    assert pipe.num_groups == min(8, G)
    x = rng.standard_normal((N, H))
    W = rng.standard_normal((R + 1, H, H))
    bias = rng.standard_normal(H)
    for direction in ("f", "b"):
        got = simulate(pipe, direction, x, W, bias)
        a, b = (src, dst) if direction == "f" else (dst, src)
        want = x @ W[R] + bias
        for e in range(len(src)):
            want[b[e]] += x[a[e]] @ W[et[e]]
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)


def test_pipe_tables_without_edges():
    from dummynode4graphlearning_amd import ops
    node_ptr = np.array([0, 3, 7], dtype=np.int64)
    z = np.zeros(0, dtype=np.int64)
    ix, _ = _fake_row_index(z, z, z, 7, 2)
    pipe = ops.PipeIndex(ix, torch.from_numpy(node_ptr), batch_nodes=4, depth=2, roles_per_group=4)
    x = np.arange(14, dtype=np.float64).reshape(7, 2)
    W = np.stack([np.eye(2)] * 3)
    out = simulate(pipe, "f", x, W, None)
    np.testing.assert_allclose(out, x)
