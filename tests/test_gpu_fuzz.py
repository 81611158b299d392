"""Randomised parity sweep: device transforms and layers against the oracle on random batches (degenerate shapes included).
Runs for DN_FUZZ_SECONDS (default 15 s) so that it stays cheap in the regular suite; set it to minutes for a soak run."""
import os
import shutil
import time

import numpy as np
import pytest
import torch

from oracle import layers as OL
from oracle import transforms as OT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BUDGET = float(os.environ.get("DN_FUZZ_SECONDS", "15"))


def _rand_batch(rng):
    """Random batch in the reference's layout: graphs contiguous, multi-edges / self loops / isolated nodes / empty graphs."""
    G = int(rng.integers(1, 40))
    node_ptr, edge_ptr, src, dst = [0], [0], [], []
    for _ in range(G):
        n = int(rng.integers(1, 14))
        m = int(rng.integers(0, 4 * n)) if rng.random() > 0.1 else 0
        base = node_ptr[-1]
        src.extend((base + rng.integers(0, n, size=m)).tolist())
        dst.extend((base + rng.integers(0, n, size=m)).tolist())
        node_ptr.append(base + n)
        edge_ptr.append(edge_ptr[-1] + m)
    N, E = node_ptr[-1], edge_ptr[-1]
    a = lambda v: np.asarray(v, dtype=np.int64)  # noqa: E731
    return dict(node_ptr=a(node_ptr), edge_ptr=a(edge_ptr), src=a(src), dst=a(dst), node_label=rng.integers(1, 5, size=N),
                edge_label=rng.integers(1, 4, size=E), node_id=np.concatenate([np.arange(node_ptr[g + 1] - node_ptr[g]) for g in range(G)]),
                edge_id=np.concatenate([np.arange(edge_ptr[g + 1] - edge_ptr[g]) for g in range(G)] + [np.zeros(0, np.int64)]).astype(np.int64))


def _dev(b, keys):
    return [torch.from_numpy(np.ascontiguousarray(b[k])).to(DEV) for k in keys]


def test_randomised_parity_sweep():
    from dummynode4graphlearning_amd import BatchedGraph, transforms
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGCNLayer, RGINLayer
    rng = np.random.default_rng(int(os.environ.get("DN_FUZZ_SEED", "12345")))
    t_end = time.time() + BUDGET
    rounds = 0
    while time.time() < t_end or rounds < 3:
        b = _rand_batch(rng)
        gk = ("node_ptr", "edge_ptr", "src", "dst", "node_label", "edge_label")
        # GC augmentation + conjugate (both modes)
        got = transforms.dummy_augment_gc(*_dev(b, gk))
        ref = OT.dummy_augment_gc(*(b[k] for k in gk))
        for k in ref:
            assert np.array_equal(got[k].cpu().numpy().astype(np.int64), ref[k]), ("gc", k)
        for mode, flag in (("gc", ref["is_dummy_edge"]), ("line", None)):
            src_b = ref if mode == "gc" else b
            cj = transforms.conjugate(*_dev(src_b, ("node_ptr", "edge_ptr", "src", "dst", "node_label")),
                                      is_dummy_edge=None if flag is None else torch.from_numpy(flag).to(DEV), mode=mode)
            rj = OT.conjugate(src_b["node_ptr"], src_b["edge_ptr"], src_b["src"], src_b["dst"], src_b["node_label"],
                              is_dummy_edge=flag, mode=mode)
            for k in ("cnode_ptr", "cedge_ptr", "csrc", "cdst", "rep_edge", "shared_node"):
                assert np.array_equal(cj[k].cpu().numpy().astype(np.int64), rj[k]), (mode, k)
        # SI augmentation (+ conjugate with merged edge ids)
        sk = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
        vocab = (20, 6, 80, 5)
        gs = transforms.dummy_augment_si(*_dev(b, sk), *vocab)
        rs = OT.dummy_augment_si(*(b[k] for k in sk), *vocab)
        for k in rs:
            assert np.array_equal(gs[k].cpu().numpy().astype(np.int64), rs[k]), ("si", k)
        cj = transforms.conjugate(gs["node_ptr"], gs["edge_ptr"], gs["src"], gs["dst"], gs["node_label"], edge_id=gs["edge_id"], mode="si")
        rj = OT.conjugate(rs["node_ptr"], rs["edge_ptr"], rs["src"], rs["dst"], rs["node_label"], edge_id=rs["edge_id"], mode="si")
        for k in ("csrc", "cdst", "rep_edge", "shared_node"):
            assert np.array_equal(cj[k].cpu().numpy().astype(np.int64), rj[k]), ("si conj", k)
        # layers on the SI-augmented batch: fp32 against the oracle (1e-4), bf16 loosely, random options
        N, R = len(rs["node_label"]), int(rs["edge_label"].max()) + 1 if len(rs["edge_label"]) else 1
        H = int(rng.choice([16, 64, 128]))
        kind = rng.choice(["rgin", "rgcn"])
        self_loop = bool(rng.integers(0, 2))
        torch.manual_seed(int(rng.integers(0, 1 << 30)))
        # (the reference CLI's defaults -- leaky_relu, bdd with 4 blocks: config.py:116,148,332,404 -- are in the mix)
        act = str(rng.choice(["relu", "leaky_relu"]))
        reg, nb = ("bdd", 4) if rng.integers(0, 3) == 0 else ("basis", -1)
        if kind == "rgin":
            nm = int(rng.choice([0, 2]))
            layer = RGINLayer(H, H, num_rels=R, regularizer=reg, num_bases=nb, num_mlp_layers=nm, self_loop=self_loop, act_func=act)
        else:
            norm = str(rng.choice(["none", "in", "both"]))
            layer = RGCNLayer(H, H, num_rels=R, regularizer=reg, num_bases=nb, edge_norm=norm, self_loop=self_loop, act_func=act)
        x = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
        coef = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
        s_t, d_t, e_t = (torch.from_numpy(rs[k]) for k in ("src", "dst", "edge_label"))
        p = {k: v.detach().clone().requires_grad_(True) for k, v in layer.named_parameters()}
        xr = x.clone().requires_grad_(True)
        if kind == "rgin":
            ref_o = OL.rgin_layer(xr, s_t, d_t, e_t, p, regularizer=reg, num_rels=R, num_bases=nb, num_mlp_layers=nm, act=act)
        else:
            ref_o = OL.rgcn_layer(xr, s_t, d_t, e_t, p, regularizer=reg, num_rels=R, num_bases=nb, edge_norm=norm, act=act)
        (ref_o * coef).sum().backward()
        dl = layer.to(DEV)
        xd = x.to(DEV).requires_grad_(True)
        with_ptr = bool(rng.integers(0, 2))               # the batch's graph boundaries -> the graph-local index builder

        def batch_graph():
            if not with_ptr:
                return BatchedGraph(s_t.to(DEV), d_t.to(DEV), N)
            return BatchedGraph(s_t.to(DEV), d_t.to(DEV), N, torch.from_numpy(np.diff(rs["node_ptr"])), torch.from_numpy(np.diff(rs["edge_ptr"])),
                                node_ptr=torch.from_numpy(rs["node_ptr"]).to(DEV), edge_ptr=torch.from_numpy(rs["edge_ptr"]).to(DEV))
        out, _ = dl(batch_graph(), xd, e_t.to(DEV))
        (out * coef.to(DEV)).sum().backward()
        desc = (kind, H, act, reg, "self_loop=%s" % self_loop, "N=%d E=%d R=%d" % (N, len(rs["src"]), R), "round %d" % rounds)
        scale = float(ref_o.detach().abs().max().clamp(min=1e-6))
        e_out = float((out.detach().cpu() - ref_o.detach()).abs().max()) / scale
        assert e_out < 1e-4, desc + ("out", e_out)
        gscale = float(xr.grad.abs().max().clamp(min=1e-6))
        gerr = (xd.grad.cpu() - xr.grad).abs() / gscale
        e_gx = float(gerr.max())
        if e_gx >= 1e-4:
            # a pre-activation within fp32 rounding of 0 may sit on opposite sides of a ReLU kink in the two evaluations: the
            # forward value (~0) still matches, but that element's whole gradient path switches.  Such an event at node v touches
            # v and the SOURCES of v's in-edges (all of a graph when v is its dummy node); a wrong kernel touches rows that no
            # small set of nodes explains.  Require that at most 2 such neighbourhoods cover every bad row, and report it.
            bad = np.nonzero((gerr.max(dim=1).values >= 1e-4).numpy())[0]
            print("fuzz: ReLU-kink event?", desc, "grad_x max err %.2e on %d of %d rows" % (e_gx, len(bad), N))
            left = set(bad.tolist())
            nb = [set([v]) for v in range(N)]
            for u, v in zip(rs["src"].tolist(), rs["dst"].tolist()):
                nb[v].add(u)
            for _ in range(2):
                if left:
                    best = max(range(N), key=lambda v: len(nb[v] & left))
                    left -= nb[best]
            assert not left, desc + ("grad_x", e_gx, len(bad), "rows not explained by two ReLU-kink neighbourhoods", sorted(left)[:8])
        # every parameter gradient (not in a round with a kink event: one flipped element is one of only ~N terms of a bias
        # gradient's column -- several per cent at these batch sizes)
        ptol = 1e-4
        for k, v in (dl.named_parameters() if e_gx < 1e-4 else ()):
            rg = p[k].grad
            if rg is None:
                assert v.grad is None or float(v.grad.abs().max()) == 0.0, desc + (k,)
                continue
            e_p = float((v.grad.cpu() - rg).abs().max()) / float(rg.abs().max().clamp(min=1e-6))
            assert e_p < ptol, desc + (k, e_p)
        if H in (64, 128) and N > 0:
            bl = dl.to(torch.bfloat16)
            xb = x.to(DEV).to(torch.bfloat16).requires_grad_(True)
            ob, _ = bl(batch_graph(), xb, e_t.to(DEV))
            ob.float().sum().backward()
            assert bool(torch.isfinite(ob.float()).all()) and bool(torch.isfinite(xb.grad.float()).all())
            err = float((ob.detach().float().cpu() - ref_o.detach()).norm() / ref_o.detach().norm().clamp(min=1e-6))
            assert err < 6e-2, (kind, H, "bf16", err)
        rounds += 1
    print("fuzz rounds:", rounds)


def _rand_batch_256(rng):
    """Batches for the H = 256 kernels: up to a few hundred graphs of 1 .. 44 nodes -- all within 31 nodes most of the time (every
    graph + its dummy node inside one 32-node tile: the absorbed fold), some over (partial rows + tail launch)."""
    G = int(rng.integers(1, 160))
    cap = 31 if rng.random() < 0.6 else 44
    node_ptr, edge_ptr, src, dst = [0], [0], [], []
    for _ in range(G):
        n = int(rng.integers(1, cap + 1))
        m = int(rng.integers(0, 3 * n)) if rng.random() > 0.05 else 0
        base = node_ptr[-1]
        src.append(base + rng.integers(0, n, size=m))
        dst.append(base + rng.integers(0, n, size=m))
        node_ptr.append(base + n)
        edge_ptr.append(edge_ptr[-1] + m)
    N, E = node_ptr[-1], edge_ptr[-1]
    a = lambda v: np.asarray(v, dtype=np.int64)  # noqa: E731
    R0 = int(rng.integers(1, 15))
    return dict(node_ptr=a(node_ptr), edge_ptr=a(edge_ptr), src=np.concatenate(src).astype(np.int64), dst=np.concatenate(dst).astype(np.int64),
                node_label=rng.integers(1, 5, size=N), edge_label=rng.integers(0, R0, size=E), node_id=np.zeros(N, np.int64),
                edge_id=np.zeros(E, np.int64)), R0


def _rgin_ref_by_relation(x, src, dst, et, p, reg, R, nb, nm, act):
    """oracle.layers.rgin_layer with the messages summed relation by relation (the same sums: the reference's [E, H, H] weight
    gather -- rgin.py:109 -- is 0.5 MB per edge at H = 256 in fp64)."""
    H = x.shape[1]
    if reg == "none" or nb is None or nb > R or nb <= 0:
        nb = R
    W = OL.relation_weights(p["weight"], p.get("w_comp"), reg, R, nb, H, H)
    f = OL.act_fn(act)
    out = x @ p["loop_weight"] + p["bias"]
    for r in range(R):
        m = (et == r).nonzero().reshape(-1)
        if m.numel():
            out = out.index_add(0, dst[m], x[src[m]] @ W[r])
    for i in range(nm):
        out = torch.nn.functional.linear(out, p["mlp.%d.weight" % (2 * i)], p["mlp.%d.bias" % (2 * i)])
        if i != nm - 1:
            out = f(out)
    if nm == 0:
        out = f(out)
    return f(out)


def test_randomised_h256_bf16_sweep():
    """The benchmarked kernels under random batches: RGINLayer at H = 256 in bf16 (ring transform, unit-stream closing launch with
    AGG units or partial rows + tail, ring MLP chains with bit masks, LDS-DMA weight gradients) on SI-augmented batches WITH their
    graph boundaries (graph-local index builder), against the oracle in fp64 on the same bf16 parameters and inputs: output, input
    gradient and EVERY parameter gradient (relative L2: bf16 storage noise and the ReLU decisions it flips bound what can be
    asked), everything finite, and the step bit-identical when repeated."""
    from dummynode4graphlearning_amd import BatchedGraph, ops, transforms
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    rng = np.random.default_rng(int(os.environ.get("DN_FUZZ_SEED", "12345")) + 2)
    t_end = time.time() + BUDGET
    rounds, absorbed_rounds, worst = 0, 0, {}
    sk = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    while time.time() < t_end or rounds < 4:
        b, R0 = _rand_batch_256(rng)
        gs = transforms.dummy_augment_si(*_dev(b, sk), 50, 6, 10, R0)
        R, H = R0 + 2, 256
        N = int(gs["node_label"].numel())
        act = str(rng.choice(["relu", "leaky_relu"]))
        reg, nb = ("bdd", 4) if (rng.integers(0, 3) == 0 and R >= 4) else ("basis", -1)    # (num_bases > R is reset to R: rgin.py:38-41)
        nm = int(rng.choice([0, 2, 2]))
        torch.manual_seed(int(rng.integers(0, 1 << 30)))
        layer = RGINLayer(H, H, num_rels=R, regularizer=reg, num_bases=nb, num_mlp_layers=nm, self_loop=True, act_func=act)
        layer = layer.to(DEV).to(torch.bfloat16)
        x0 = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32)).to(DEV).to(torch.bfloat16)
        coef = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32)).to(DEV).to(torch.bfloat16)
        et = gs["edge_label"].long()
        with_ptr = rng.random() < 0.8
        kw = dict(node_ptr=gs["node_ptr"], edge_ptr=gs["edge_ptr"]) if with_ptr else {}
        bnn = (gs["node_ptr"][1:] - gs["node_ptr"][:-1]).long()
        bne = (gs["edge_ptr"][1:] - gs["edge_ptr"][:-1]).long()
        g = BatchedGraph(gs["src"], gs["dst"], N, bnn, bne, **kw)
        runs = []
        for _ in range(2):
            for q in layer.parameters():
                q.grad = None
            x = x0.clone().requires_grad_(True)
            out, _ = layer(g, x, et)
            out.backward(coef)
            runs.append([out.detach().clone(), x.grad.clone()] + [q.grad.clone() for q in layer.parameters()])
        desc = ("N=%d E=%d R=%d" % (N, int(et.numel()), R), act, reg, "mlp %d" % nm, "ptr=%s" % with_ptr, "round %d" % rounds)
        for a_, b_ in zip(*runs):
            assert torch.equal(a_, b_), desc
            assert bool(torch.isfinite(a_.float()).all()), desc
        ix = g.row_index(et, R, True).parts[0][2]
        if with_ptr and int(et.numel()) > 0:
            assert ix.built_by == "local", desc
        absorbed_rounds += int(all(d in ix._units and ix._units[d].agg for d in "fb"))
        p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in layer.named_parameters()}
        xr = x0.double().cpu().requires_grad_(True)
        ref = _rgin_ref_by_relation(xr, gs["src"].long().cpu(), gs["dst"].long().cpu(), et.cpu(), p64, reg, R, nb, nm, act)
        ref.backward(coef.double().cpu())

        def l2(a_, b_):
            return float((a_.double().cpu() - b_).norm() / b_.norm().clamp(min=1e-9))
        errs = {"out": l2(runs[0][0], ref.detach()), "grad_x": l2(runs[0][1], xr.grad)}
        for (k, v), gq in zip(layer.named_parameters(), runs[0][2:]):
            if p64[k].grad is not None and float(p64[k].grad.norm()) > 0:
                errs[k] = l2(gq, p64[k].grad)
        for k, e in errs.items():
            worst[k] = max(worst.get(k, 0.0), e)
            # (no storage points in this reference: a bf16 rounding of every stored tensor + the ReLU decisions it flips; on a batch
            #  of a few hundred nodes ONE flipped decision moves every gradient by percents -- round 73 of the fixed stream, N = 165:
            #  out 2.9e-3, every gradient 4-6 %, the relation weights 11 %, with the round-5 kernels as with these)
            assert e < (1e-2 if k == "out" else (1e-1 if N >= 400 else 2e-1)), "%s: %s %.3e (all: %s)" % (" ".join(desc), k, e, " ".join("%s %.2e" % kv for kv in errs.items()))
        rounds += 1
    print("h256 fuzz rounds: %d (%d with both folds absorbed); worst relative L2: %s" % (
        rounds, absorbed_rounds, " ".join("%s %.1e" % kv for kv in sorted(worst.items()))))
    assert rounds >= 4


def test_randomised_gc_and_dual_sweep():
    """GIN aggregation + readouts (graph_classification) and the dual message-passing layers on random batches."""
    from dummynode4graphlearning_amd import BatchedGraph, ops
    from dummynode4graphlearning_amd.subgraph_isomorphism import CompGCNLayer, DMPLayer
    rng = np.random.default_rng(int(os.environ.get("DN_FUZZ_SEED", "12345")) + 1)
    t_end = time.time() + BUDGET
    rounds = 0
    while time.time() < t_end or rounds < 3:
        b = _rand_batch(rng)
        N, E = int(b["node_ptr"][-1]), len(b["src"])
        H = int(rng.choice([8, 32, 64, 128]))
        s_t, d_t = torch.from_numpy(b["src"]), torch.from_numpy(b["dst"])
        x = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
        coef = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
        # GIN aggregation (1 + eps) x_i + sum_j x_j, with a per-edge weight half of the time
        eps = float(rng.standard_normal()) * 0.3
        w = torch.from_numpy(rng.random(E).astype(np.float32)) if rng.random() < 0.5 else None
        ix = ops.EdgeIndex(s_t.to(DEV), d_t.to(DEV), N)
        xd = x.to(DEV).requires_grad_(True)
        agg = ops.neighbor_sum(xd, ix, 1.0 + eps, None if w is None else w.to(DEV))
        (agg * coef.to(DEV)).sum().backward()
        xr = x.clone().requires_grad_(True)
        msg = xr[s_t] * (w.view(-1, 1) if w is not None else 1.0)
        ref = (1.0 + eps) * xr + torch.zeros(N, H).index_add(0, d_t, msg)
        (ref * coef).sum().backward()
        sc = float(ref.detach().abs().max().clamp(min=1e-6))
        assert float((agg.detach().cpu() - ref.detach()).abs().max()) / sc < 1e-5
        assert float((xd.grad.cpu() - xr.grad).abs().max()) / float(xr.grad.abs().max().clamp(min=1e-6)) < 1e-5
        # readouts
        gptr = torch.from_numpy(b["node_ptr"]).to(DEV).int()
        batch = torch.repeat_interleave(torch.arange(len(b["node_ptr"]) - 1), torch.from_numpy(np.diff(b["node_ptr"])))
        for kind, okind in (("sum", "add"), ("mean", "mean"), ("max", "max")):
            got = ops.segment_reduce(x.to(DEV), gptr, kind)
            refp = OL.global_pool(x, batch, len(b["node_ptr"]) - 1, okind)
            assert float((got.cpu() - refp).abs().max()) < 1e-5 * max(1.0, float(refp.abs().max())), kind
        # dual layers (fp32, H a matrix-core width half of the time)
        if E > 0:
            Hd = int(rng.choice([16, 64]))
            kind = str(rng.choice(["compgcn", "dmp"]))
            rev = torch.from_numpy(rng.random(E) < 0.5) if rng.random() < 0.6 else None
            torch.manual_seed(int(rng.integers(0, 1 << 30)))
            if kind == "compgcn":
                comp, norm, sl = str(rng.choice(["sub", "mult", "corr"])), str(rng.choice(["none", "in", "out", "both"])), bool(rng.integers(0, 2))
                layer = CompGCNLayer(Hd, Hd, self_loop=sl, comp_opt=comp, edge_norm=norm, act_func="tanh")
            else:
                nm = int(rng.choice([0, 1, 2]))
                layer = DMPLayer(Hd, Hd, num_mlp_layers=nm, batch_norm=False, act_func="tanh")
            xx = torch.from_numpy(rng.standard_normal((N, Hd)).astype(np.float32))
            ef = torch.from_numpy(rng.standard_normal((E, Hd)).astype(np.float32))
            p = {k: v.detach().clone() for k, v in layer.named_parameters()}
            if kind == "compgcn":
                rn, re = OL.compgcn_layer(xx, ef, s_t, d_t, rev, p, comp_opt=comp, edge_norm=norm, act="tanh")
            else:
                rn, re = OL.dmp_layer(xx, ef, s_t, d_t, rev, p, num_mlp_layers=nm, act="tanh")
            g = BatchedGraph(s_t.to(DEV), d_t.to(DEV), N)
            if rev is not None:
                g.edata["is_reversed"] = rev.to(DEV)
            no, eo = layer.to(DEV)(g, xx.to(DEV), ef.to(DEV))
            for a, r_, nm_ in ((no, rn, "node"), (eo, re, "edge")):
                err = float((a.detach().cpu() - r_.detach()).abs().max()) / float(r_.detach().abs().max().clamp(min=1e-6))
                assert err < 1e-4, (kind, Hd, nm_, err)
        rounds += 1
    print("fuzz (gc + dual) rounds:", rounds)


def test_randomised_tu_files_and_bookkeeping(tmp_path):
    """f-2 / f-3 on random inputs: whole DUMMY_/LINE_/CONJ_ datasets byte for byte against the oracle writer (itself pinned to
    reference-written files), and the conjugate-subisomorphism / match-weight bookkeeping against its restatement."""
    from dummynode4graphlearning_amd import tu_io
    from dummynode4graphlearning_amd.subgraph_isomorphism import bookkeeping as BK
    from oracle import si_bookkeeping as OB
    from oracle import tu_format as TF
    rng = np.random.default_rng(int(os.environ.get("DN_FUZZ_SEED", "12345")) + 2)
    t_end = time.time() + BUDGET
    rounds = 0
    while time.time() < t_end or rounds < 2:
        name = "FZ%d" % rounds
        raw = os.path.join(str(tmp_path), name, "raw")
        os.makedirs(raw)
        G = int(rng.integers(1, 9))
        A, gi, nl, base = [], [], [], 0
        for g in range(G):
            n = int(rng.integers(1, 9))
            m = int(rng.integers(1, 3 * n + 1)) if (g == G - 1 or rng.random() > 0.15) else 0
            A += [(base + int(rng.integers(0, n)) + 1, base + int(rng.integers(0, n)) + 1) for _ in range(m)]
            gi += [g + 1] * n
            nl += [int(v) for v in rng.integers(0, 4, size=n)]
            base += n
        wr = lambda fn, rows: open(os.path.join(raw, "%s_%s.txt" % (name, fn)), "w").write("".join(str(r) + "\n" for r in rows))  # noqa: E731
        wr("A", ["%d, %d" % e for e in A]); wr("graph_indicator", gi); wr("node_labels", nl)
        if rng.random() < 0.6:
            wr("edge_labels", [int(v) for v in rng.integers(0, 3, size=len(A))])
        with_attr = rng.random() < 0.5 and len(A) > len(gi)          # the reference indexes edge attributes by a node offset
        if with_attr:
            wr("node_attributes", [repr(round(float(v), 3)) for v in rng.standard_normal(len(gi))])
            wr("edge_attributes", [repr(round(float(v), 3)) for v in rng.standard_normal(len(A))])
        wr("graph_labels", [int(v) for v in rng.integers(0, 2, size=G)])
        tu_io.process_dataset(raw, name)
        ref_root = os.path.join(str(tmp_path), "ref%d" % rounds)
        ref_raw = os.path.join(ref_root, name, "raw")
        os.makedirs(ref_raw)
        for fn in os.listdir(raw):
            with open(os.path.join(raw, fn)) as f, open(os.path.join(ref_raw, fn), "w") as o:
                o.write(f.read())
        TF.process_dataset(ref_raw, name)
        for pre in ("DUMMY_", "LINE_", "CONJ_"):
            d_got, d_ref = raw.replace(name, pre + name), ref_raw.replace(name, pre + name)
            assert sorted(os.listdir(d_got)) == sorted(os.listdir(d_ref)), pre
            for fn in os.listdir(d_ref):
                with open(os.path.join(d_got, fn)) as f, open(os.path.join(d_ref, fn)) as r:
                    assert f.read() == r.read(), (pre, fn)
        for d in [ref_root] + [os.path.join(str(tmp_path), pre + name) for pre in ("", "DUMMY_", "LINE_", "CONJ_")]:
            shutil.rmtree(d, ignore_errors=True)                 # a soak run writes tens of thousands of datasets
        # bookkeeping
        pn, gn = int(rng.integers(2, 6)), int(rng.integers(5, 30))
        pm = int(rng.integers(1, 10))
        p_u, p_v, p_el = rng.integers(0, pn, size=pm), rng.integers(0, pn, size=pm), rng.integers(0, 3, size=pm)
        S_ = int(rng.integers(1, 9))
        sub = np.stack([rng.permutation(gn)[:pn] for _ in range(S_)])
        g_u = np.concatenate([rng.integers(0, gn, size=40), sub[:, p_u].reshape(-1)])
        g_v = np.concatenate([rng.integers(0, gn, size=40), sub[:, p_v].reshape(-1)])
        g_el = np.concatenate([rng.integers(0, 3, size=40), np.tile(p_el, S_)])
        o = np.lexsort((g_v, g_u))
        g_u, g_v, g_el = g_u[o], g_v[o], g_el[o]
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int64)).to(DEV)  # noqa: E731
        a = [t(v) for v in (p_u, p_v, p_el, g_u, g_v, g_el, sub)]
        assert np.array_equal(BK.get_conjugate_subisomorphisms(*a).cpu().numpy(), OB.conjugate_subisomorphisms(p_u, p_v, p_el, g_u, g_v, g_el, sub))
        assert np.array_equal(BK.compute_edgeseq_subisoweights(*a).cpu().numpy(), OB.edgeseq_subisoweights(p_u, p_v, p_el, g_u, g_v, g_el, sub))
        rounds += 1
    print("fuzz (tu files + bookkeeping) rounds:", rounds)
