"""Data-parallel plumbing on the CPU with the gloo backend, world_size 2: graph sharding, the flat gradient bucket across
optimizer.zero_grad() steps, BatchNorm statistics over the global batch, and bench.py's own rank launcher."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dummynode4graphlearning_amd.parallel import FlatGradBucket, shard_graphs
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        bucket = FlatGradBucket(model.parameters())
        # a "batch" of 8 graphs of different sizes; every rank takes its contiguous shard
        sizes_n = torch.tensor([3, 9, 4, 4, 7, 2, 6, 5])
        sizes_e = torch.tensor([4, 20, 6, 5, 15, 2, 9, 7])
        shards = shard_graphs(sizes_n, sizes_e, world)
        assert shards[0][0] == 0 and shards[-1][1] == 8 and all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
        g0, g1 = shards[rank]
        data = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10.0
        bucket.zero()
        loss = model(data[g0:g1]).square().sum() / 8.0                # global-batch mean: scale by 1/global
        loss.backward()
        assert model[0].weight.grad.data_ptr() == bucket.flat.data_ptr()    # grads ARE slices of the bucket
        bucket.average = False
        bucket.all_reduce()
        # reference: the whole batch on one process
        torch.manual_seed(0)
        ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        (ref(data).square().sum() / 8.0).backward()
        err = max(float((p.grad - q.grad).abs().max()) for p, q in zip(model.parameters(), ref.parameters()))
        q.put((rank, err, shards))
    finally:
        dist.destroy_process_group()


def test_sharded_gradients_match_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, shards in res:
        assert err < 1e-6, (rank, err)
    assert res[0][2] == res[1][2]


def test_shard_graphs_balances_nodes_plus_edges():
    from dummynode4graphlearning_amd.parallel import shard_graphs
    n = torch.full((1000,), 31)
    e = torch.full((1000,), 122)
    sh = shard_graphs(n, e, 8)
    assert [b - a for a, b in sh] == [125] * 8
    n = torch.tensor([100, 1, 1, 1, 1, 1, 1, 1])
    sh = shard_graphs(n, torch.zeros(8, dtype=torch.long), 2)
    assert sh[0] == (0, 1) and sh[1] == (1, 8)
    assert shard_graphs(torch.tensor([5]), torch.tensor([5]), 4)[-1][1] == 1


def _run_world(target, world=2, timeout=180):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


def _steps_worker(rank, world, port, q):
    """the reference loop's optimizer.zero_grad() sets .grad to None (torch >= 2.0), autograd then
    allocates fresh gradient tensors -- the bucket must carry THOSE, step after step, and the replicas must stay equal."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dummynode4graphlearning_amd.parallel import FlatGradBucket
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        model[2].to(torch.float64)                                    # mixed dtypes: one bucket per dtype
        unused = torch.nn.Parameter(torch.ones(4))                     # never receives a gradient
        params = list(model.parameters()) + [unused]
        bucket = FlatGradBucket(params)
        assert len(bucket.buckets()) == 2
        opt = torch.optim.SGD(params, lr=0.1)
        torch.manual_seed(1)
        data = torch.randn(8, 6)
        torch.manual_seed(0)
        ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        ref[2].to(torch.float64)
        ropt = torch.optim.SGD(ref.parameters(), lr=0.1)
        half = slice(0, 4) if rank == 0 else slice(4, 8)
        for step in range(3):
            opt.zero_grad()                                           # set_to_none=True: breaks the aliasing on purpose
            loss = model[2](torch.relu(model[0](data[half])).double()).square().mean()
            loss.backward()
            assert model[0].weight.grad.data_ptr() != bucket.flat.data_ptr() or step == 0
            bucket.all_reduce()                                       # average of the two half-batch means = global mean
            assert model[0].weight.grad.data_ptr() == bucket.buckets()[0].data_ptr()
            opt.step()
            ropt.zero_grad()
            ref[2](torch.relu(ref[0](data)).double()).square().mean().backward()
            ropt.step()
        err = max(float((p.double() - r.double()).abs().max()) for p, r in zip(model.parameters(), ref.parameters()))
        flat = torch.cat([p.detach().double().reshape(-1) for p in model.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        q.put((rank, err, float((gathered[0] - gathered[1]).abs().max()), float(unused.grad.abs().max())))
    finally:
        dist.destroy_process_group()


def test_bucket_survives_zero_grad_and_replicas_stay_identical():
    for rank, err, spread, unused_grad in _run_world(_steps_worker):
        assert err < 1e-6, (rank, err)                  # 3 SGD steps == the single-process run on the whole batch
        assert spread == 0.0                            # replicas bit-identical after the steps
        assert unused_grad == 0.0


def _overlap_worker(rank, world, port, q):
    """OverlappedGradReducer: one bucket per layer, each all-reduced from INSIDE backward the moment its last gradient lands."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dummynode4graphlearning_amd.parallel import OverlappedGradReducer

        def make():
            torch.manual_seed(0)
            return torch.nn.Sequential(torch.nn.Linear(6, 7), torch.nn.ReLU(), torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        model, ref = make(), make()
        unused = torch.nn.Parameter(torch.ones(4))                     # a group whose hooks never fire: it goes FIRST (the groups
        layers = [model[0], model[2], model[4]]                        # leave last-to-first, so it holds nobody back)
        reducer = OverlappedGradReducer([[unused]] + [l.parameters() for l in layers])
        opt = torch.optim.SGD(list(model.parameters()) + [unused], lr=0.1)
        ropt = torch.optim.SGD(ref.parameters(), lr=0.1)
        torch.manual_seed(1)
        data = torch.randn(8, 6)
        half = slice(0, 4) if rank == 0 else slice(4, 8)
        logs = []
        for step in range(3):
            opt.zero_grad()                                           # set_to_none=True
            model(data[half]).square().mean().backward()
            logs.append(list(reducer.launched))                       # what left during backward, before finish()
            reducer.finish()
            opt.step()
            ropt.zero_grad()
            ref(data).square().mean().backward()
            ropt.step()
        err = max(float((p - r).abs().max()) for p, r in zip(model.parameters(), ref.parameters()))
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        q.put((rank, err, float((gathered[0] - gathered[1]).abs().max()), logs, float(unused.grad.abs().max())))
    finally:
        dist.destroy_process_group()


def test_overlapped_reducer_launches_each_layers_collective_inside_backward():
    for rank, err, spread, logs, unused_grad in _run_world(_overlap_worker):
        assert err < 1e-6, (rank, err)                  # 3 SGD steps == the single-process run on the whole batch
        assert spread == 0.0
        assert unused_grad == 0.0
        for log in logs:
            # the three layer groups left in reverse layer order, from inside backward(); when the last layer's bucket left, the 4
            # parameters of the two layers below had no gradient yet (their backward had not run), then 2, then 0
            assert [g for g, _ in log] == [3, 2, 1], log
            assert [w - 1 for _, w in log] == [4, 2, 0], log           # (-1: the never-used parameter stays pending)


def _overlap_uneven_worker(rank, world, port, q):
    """A parameter of the LAST group gets a gradient on rank 0 only (a data-dependent branch): the ranks must still issue their
    collectives in the same order -- rank 1 holds every group back until finish() instead of sending layer 1 first."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dummynode4graphlearning_amd.parallel import OverlappedGradReducer
        torch.manual_seed(0)
        l0, l1 = torch.nn.Linear(5, 5), torch.nn.Linear(5, 5)          # equally sized buckets: a swapped order would not even fail
        extra = torch.nn.Parameter(torch.full((5,), 0.5))              # part of the last group, used on rank 0 only
        reducer = OverlappedGradReducer([l0.parameters(), list(l1.parameters()) + [extra]])
        torch.manual_seed(1 + rank)
        x = torch.randn(4, 5)
        orders = []
        for step in range(2):
            for p in list(l0.parameters()) + list(l1.parameters()) + [extra]:
                p.grad = None
            def loss():
                y = l1(torch.relu(l0(x)))
                return ((y + extra) if rank == 0 else y).square().mean()
            ps = list(l0.parameters()) + list(l1.parameters())
            want = {id(p): g.detach().clone() for p, g in zip(ps, torch.autograd.grad(loss(), ps))}   # (no accumulation: no hooks)
            loss().backward()
            during = [g for g, _ in reducer.launched]
            reducer.finish()
            orders.append(during)
            # averaged gradients: gather both ranks' local ones and compare
            for p in list(l0.parameters()) + list(l1.parameters()):
                both = [torch.zeros_like(want[id(p)]) for _ in range(world)]
                dist.all_gather(both, want[id(p)])
                assert torch.allclose(p.grad, (both[0] + both[1]) / 2, atol=1e-6)
        raised = False
        for p in list(l0.parameters()) + list(l1.parameters()) + [extra]:
            p.grad = None
        (l1(torch.relu(l0(x))) + extra).square().mean().backward()     # every group complete: all collectives have left ...
        try:
            (l1(torch.relu(l0(x))) + extra).square().mean().backward() # ... so a second backward without finish() must refuse
        except RuntimeError as e:
            raised = "finish()" in str(e)
        try:                                                           # ... and the reducer stays unusable ...
            reducer.finish()
            raised = False
        except RuntimeError as e:
            raised = raised and "reset()" in str(e)
        reducer.reset()                                                # ... until it is re-armed without communicating
        # micro-batch accumulation: two backward() calls under no_sync() only accumulate, the third reduces the SUM
        ps = list(l0.parameters()) + list(l1.parameters())
        torch.manual_seed(7 + rank)
        mbs = [torch.randn(4, 5) for _ in range(3)]
        f = lambda xb: (l1(torch.relu(l0(xb))) + extra).square().mean()
        want = [sum(gs) for gs in zip(*[torch.autograd.grad(f(xb), ps) for xb in mbs])]
        for p in ps + [extra]:
            p.grad = None
        with reducer.no_sync():
            for xb in mbs[:-1]:
                f(xb).backward()
                assert reducer.launched == []
        f(mbs[-1]).backward()
        reducer.finish()
        acc_ok = True
        for p, w in zip(ps, want):
            both = [torch.zeros_like(w) for _ in range(world)]
            dist.all_gather(both, w.detach())
            acc_ok = acc_ok and bool(torch.allclose(p.grad, (both[0] + both[1]) / 2, atol=1e-6))
        q.put((rank, orders, raised and acc_ok))
    finally:
        dist.destroy_process_group()


def test_overlapped_reducer_keeps_one_collective_order_on_every_rank():
    res = {r: (o, raised) for r, o, raised in _run_world(_overlap_uneven_worker)}
    assert res[0][0] == [[1, 0], [1, 0]]                # rank 0: both groups left inside backward, last group first
    assert res[1][0] == [[], []]                        # rank 1: `extra` never got a gradient -> everything waits for finish()
    assert res[0][1] and res[1][1]


def _syncbn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dummynode4graphlearning_amd.parallel import FlatGradBucket, SyncBatchNorm1d, convert_sync_batchnorm, dp_loss_scale
        dt = torch.float64 if os.environ.get("DN_TEST_F64") == "1" else torch.float32
        def make():
            torch.manual_seed(0)
            return torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.BatchNorm1d(8), torch.nn.ReLU(), torch.nn.Linear(8, 2)).to(dt)
        ref, model = make(), convert_sync_batchnorm(make())
        assert isinstance(model[1], SyncBatchNorm1d) and list(model.state_dict()) == list(ref.state_dict())
        torch.manual_seed(3)
        rows = (torch.randn(23, 6) * 3.0 + 5.0).to(dt)                 # large mean: the naive sum-of-squares form would cancel
        graph_of_row = torch.sort(torch.randint(0, 6, (23,)))[0]       # 6 "graphs" of uneven size; rank 0 takes 4, rank 1 takes 2
        y = torch.randn(6, 2).to(dt)
        def loss_of(net, sel_rows, sel_graphs):
            h = net(rows[sel_rows])
            pooled = torch.zeros(6, 2, dtype=dt).index_add(0, graph_of_row[sel_rows], h)[sel_graphs]
            return (pooled - y[sel_graphs]).square().mean()            # mean over the graphs of the (sub-)batch
        mine = [0, 1, 2, 3] if rank == 0 else [4, 5]
        sel = torch.isin(graph_of_row, torch.tensor(mine))
        bucket = FlatGradBucket(model.parameters())
        model.train(), ref.train()
        bucket.zero()
        (loss_of(model, sel, mine) * dp_loss_scale(len(mine), 6)).backward()
        bucket.all_reduce()
        loss_of(ref, torch.ones(23, dtype=torch.bool), list(range(6))).backward()
        # (the Linear bias in front of the BatchNorm has an exactly-zero true gradient: absolute floor on the scale)
        gerr = max(float((p.grad - r.grad).abs().max() / r.grad.abs().max().clamp(min=1e-3))
                   for p, r in zip(model.parameters(), ref.parameters()))
        serr = max(float((model.state_dict()[k].double() - ref.state_dict()[k].double()).abs().max())
                   for k in ("1.running_mean", "1.running_var"))
        model.eval(), ref.eval()
        eerr = float((model(rows) - ref(rows)).detach().abs().max())
        q.put((rank, gerr, serr, eerr, int(model[1].num_batches_tracked)))
    finally:
        dist.destroy_process_group()


def test_sync_batchnorm_equals_single_process_statistics():
    """SURVEY 8e caveat 1: the GC models' BatchNorm1d under data parallelism -- global-batch statistics, gradients (through
    the statistics) and running buffers equal the single-process run on the whole batch, with uneven shards."""
    for f64, tol in (("1", 1e-12), ("0", 2e-4)):        # fp64: the algebra is exact; fp32: rounding only
        os.environ["DN_TEST_F64"] = f64                 # inherited by the spawned ranks
        try:
            for rank, gerr, serr, eerr, nb in _run_world(_syncbn_worker):
                assert gerr < tol, (f64, rank, gerr)
                assert serr < max(tol, 1e-5) and eerr < max(tol, 1e-5) and nb == 1, (f64, rank, serr, eerr, nb)
        finally:
            os.environ.pop("DN_TEST_F64", None)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment must create two ranks itself and report n_gpus 2
    (--dry-run: gloo rendezvous on the CPU, no product code)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["rank_sum"] == 3.0 and lines[0]["steps"] == 3
    # config 5 is ONE global batch cut by parallel.shard_graphs: the two ranks report complementary graph ranges
    assert lines[0]["scaling"] == "strong" and lines[0]["shard_graphs"] == [[0, 16384], [16384, 32768]]
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run"],
                         env=dict(env, RANK="0", WORLD_SIZE="2"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)


def test_bench_shards_of_the_global_batch_are_disjoint_complete_and_balanced():
    """bench.shard_of: the 8 ranks of the config-5 run take 4,096 graphs each; on graphs of unequal sizes the ranges
    still tile the batch and carry equal nodes + edges to within one graph; a shard is a self-contained batch (ids relative)."""
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from dummynode4graphlearning_amd import synthetic
    raw = synthetic.config5(5, 32768)
    ranges = [bench.shard_of(raw, r, 8)[1] for r in range(8)]
    assert ranges == [(4096 * r, 4096 * (r + 1)) for r in range(8)]
    sub, (g0, g1) = bench.shard_of(raw, 3, 8)
    assert sub["node_ptr"][0] == 0 and sub["edge_ptr"][0] == 0 and len(sub["node_ptr"]) == g1 - g0 + 1
    assert sub["src"].min() >= 0 and sub["dst"].max() < sub["node_ptr"][-1] and len(sub["src"]) == sub["edge_ptr"][-1]
    e0 = int(raw["edge_ptr"][g0])
    assert np.array_equal(sub["src"], raw["src"][e0:e0 + len(sub["src"])] - raw["node_ptr"][g0])
    # unequal graphs (config 4's NCI1-shaped sizes, relabelled as an SI batch is not needed: only the split is under test)
    from dummynode4graphlearning_amd.parallel import shard_graphs
    r4 = synthetic.config4(seed=4, graphs=1000)
    bnn = r4["node_ptr"][1:] - r4["node_ptr"][:-1]
    bne = r4["edge_ptr"][1:] - r4["edge_ptr"][:-1]
    w = (bnn + 1) + (bne + 2 * bnn)
    sh = shard_graphs(torch.from_numpy(bnn + 1), torch.from_numpy(bne + 2 * bnn), 8)
    assert sh[0][0] == 0 and sh[-1][1] == 1000 and all(sh[i][1] == sh[i + 1][0] for i in range(7))
    loads = [int(w[a:b].sum()) for a, b in sh]
    assert max(loads) - min(loads) <= 2 * int(w.max())


def test_bucket_pack_after_set_to_none_gathers_the_steps_gradients():
    """zero(set_to_none=True) -- optimizer.zero_grad()'s default -- then backward WRITES fresh gradient tensors; pack() gathers them
    into the flat buffers in one multi-tensor copy and re-aliases .grad (mixed dtypes, an unused parameter); no process group."""
    from dummynode4graphlearning_amd.parallel import FlatGradBucket
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(4, 3))
    b = torch.nn.Parameter(torch.randn(5))
    c = torch.nn.Parameter(torch.randn(2, 2, dtype=torch.float64))
    unused = torch.nn.Parameter(torch.randn(3))
    bucket = FlatGradBucket([a, b, c, unused])
    for step in range(3):
        bucket.zero(set_to_none=True)
        assert all(p.grad is None for p in (a, b, c, unused))
        x = torch.randn(3)
        loss = (a @ x).sum() * (step + 1) + (b ** 2).sum() + (c.double() ** 3).sum()
        loss.backward()
        want = {id(p): p.grad.detach().clone() for p in (a, b, c)}
        assert a.grad.data_ptr() != bucket.buckets()[0].data_ptr()              # fresh tensors, not views of the bucket
        bucket.pack()
        f32, f64 = bucket.buckets()
        assert torch.equal(f32[:12].view(4, 3), want[id(a)]) and torch.equal(f32[12:17], want[id(b)])
        assert torch.equal(f32[17:20], torch.zeros(3))                           # the unused parameter contributes zeros
        assert torch.equal(f64.view(2, 2), want[id(c)])
        assert a.grad.data_ptr() == f32.data_ptr() and unused.grad is not None  # .grad aliases the bucket again
        bucket.all_reduce()                                                      # (a world of one: identity)
        assert torch.equal(f32[:12].view(4, 3), want[id(a)])
