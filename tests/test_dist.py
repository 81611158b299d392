"""Data-parallel plumbing on the CPU with the gloo backend, world_size 2: graph sharding + flat gradient bucket."""
import os
import socket

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
