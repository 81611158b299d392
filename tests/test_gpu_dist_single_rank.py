"""The RCCL code path on real hardware with the one GPU a test box has: a world-size-1 `nccl` process group runs the same
collectives (flat-bucket all-reduce, SyncBatchNorm's all-gather / all-reduce) as an 8-GPU job, and the result must equal the
plain single-process step.  (World-size-2 behaviour is covered on the CPU with gloo in tests/test_dist.py.)"""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture
def nccl_world_of_one():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from dummynode4graphlearning_amd import parallel
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    parallel.RUN_COLLECTIVES_IN_A_WORLD_OF_ONE = True            # really issue the RCCL calls
    yield
    parallel.RUN_COLLECTIVES_IN_A_WORLD_OF_ONE = False
    dist.destroy_process_group()
    for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)


def test_rccl_bucket_and_syncbn_match_plain_step(nccl_world_of_one):
    from dummynode4graphlearning_amd import BatchedGraph, GraphBatch, parallel, synthetic, transforms
    from dummynode4graphlearning_amd import graph_classification as GC
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    # ---- SI layer (HIP kernels) + flat bucket over RCCL, two steps with optimizer.zero_grad() in between
    raw = synthetic.config3(seed=4, graphs=32)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(DEV) for k in keys), raw["max_nv"], raw["max_nvl"],
                                      raw["max_ne"], raw["max_nel"])
    N, R = int(aug["node_label"].numel()), raw["num_rels"]
    g = BatchedGraph(aug["src"], aug["dst"], N)
    et = aug["edge_label"].long()
    torch.manual_seed(0)
    a = RGINLayer(64, 64, num_rels=R).to(DEV)
    b = RGINLayer(64, 64, num_rels=R).to(DEV)
    b.load_state_dict(a.state_dict())
    bucket = parallel.FlatGradBucket(a.parameters())
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda *a_, **k_: (calls.append("all_reduce"), orig(*a_, **k_))[1]
    oa, ob = torch.optim.SGD(a.parameters(), lr=0.05), torch.optim.SGD(b.parameters(), lr=0.05)
    x = torch.randn(N, 64, device=DEV)
    for _ in range(2):
        oa.zero_grad()
        a(g, x, et)[0].square().mean().backward()
        bucket.all_reduce()                                   # RCCL all-reduce (average over a world of one)
        oa.step()
        ob.zero_grad()
        b(g, x, et)[0].square().mean().backward()
        ob.step()
    dist.all_reduce = orig
    assert len(calls) == 2, "the bucket did not issue its RCCL all-reduce"
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
    # ---- GC GIN with SyncBatchNorm1d: collectives on the device, equal to torch's BatchNorm1d
    rng = np.random.default_rng(1)
    n_g = rng.integers(5, 20, size=24)
    ptr = np.concatenate([[0], np.cumsum(n_g)])
    xs = torch.from_numpy(rng.standard_normal((ptr[-1], 8)).astype(np.float32))
    ei = np.concatenate([np.stack([rng.integers(0, n, size=3 * n) + ptr[i], rng.integers(0, n, size=3 * n) + ptr[i]]) for i, n in enumerate(n_g)], 1)
    batch = torch.repeat_interleave(torch.arange(24), torch.from_numpy(n_g))
    data = GraphBatch(xs, torch.from_numpy(ei).long(), batch, y=torch.from_numpy(rng.integers(0, 2, size=24))).to(DEV)
    args = SimpleNamespace(num_features=8, hidden_dim=64, num_classes=2, dropout_ratio=0.0, additional=None, epochs=1, device=DEV,
                           dummy_weight=0)
    torch.manual_seed(1)
    ref = GC.GIN(args).to(DEV).train()
    torch.manual_seed(1)
    syn = parallel.convert_sync_batchnorm(GC.GIN(args)).to(DEV).train()
    assert any(isinstance(m, parallel.SyncBatchNorm1d) for m in syn.modules())
    F.nll_loss(ref(data), data.y).backward()
    F.nll_loss(syn(data), data.y).backward()
    for (k, p), (_, q) in zip(ref.named_parameters(), syn.named_parameters()):
        if p.grad is not None and float(p.grad.abs().max()) > 1e-5:
            err = float((p.grad - q.grad).abs().max() / p.grad.abs().max())
            assert err < 1e-3, (k, err)
    for (k, u), (_, v) in zip(ref.state_dict().items(), syn.state_dict().items()):
        if "running_" in k:
            torch.testing.assert_close(u, v, rtol=1e-4, atol=1e-5, msg=k)


def test_rccl_average_of_a_bf16_bucket(nccl_world_of_one):
    """bench.py's config-5 bucket is bf16 and FlatGradBucket averages inside the collective on RCCL (ReduceOp.AVG): the call must
    be accepted for bf16 and, in a world of one, leave the gradients bit for bit."""
    from dummynode4graphlearning_amd import parallel
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(256, 256, device=DEV).to(torch.bfloat16)) for _ in range(3)]
    bucket = parallel.FlatGradBucket(params)
    want = []
    for p in params:
        p.grad = torch.randn_like(p)
        want.append(p.grad.clone())
    bucket.all_reduce()
    torch.cuda.synchronize()
    for p, w in zip(params, want):
        assert p.grad.data_ptr() != w.data_ptr() and torch.equal(p.grad, w)
    assert params[0].grad.data_ptr() == bucket.flat.data_ptr()


def test_overlapped_reducer_over_rccl_on_a_three_layer_stack(nccl_world_of_one):
    """parallel.OverlappedGradReducer on a 3-layer RGIN stack (the reference's depth, config.py rgin_num_layers): every layer's
    bucket leaves for RCCL from INSIDE backward, in reverse layer order, and the step equals the plain one."""
    from dummynode4graphlearning_amd import BatchedGraph, parallel, synthetic, transforms
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    raw = synthetic.config3(seed=6, graphs=24)
    keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
    aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(DEV) for k in keys), raw["max_nv"], raw["max_nvl"],
                                      raw["max_ne"], raw["max_nel"])
    N, R = int(aug["node_label"].numel()), raw["num_rels"]
    g = BatchedGraph(aug["src"], aug["dst"], N)
    et = aug["edge_label"].long()

    def stack():
        torch.manual_seed(0)
        return torch.nn.ModuleList([RGINLayer(64, 64, num_rels=R, regularizer="bdd", num_bases=4, act_func="leaky_relu") for _ in range(3)]).to(DEV)

    def run(layers, x):
        h = x
        for l in layers:
            h = l(g, h, et)[0]
        return h.square().mean()
    a, b = stack(), stack()
    reducer = parallel.OverlappedGradReducer([l.parameters() for l in a])
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda *a_, **k_: (calls.append("all_reduce"), orig(*a_, **k_))[1]
    oa, ob = torch.optim.SGD(a.parameters(), lr=0.05), torch.optim.SGD(b.parameters(), lr=0.05)
    x = torch.randn(N, 64, device=DEV)
    try:
        for _ in range(2):
            oa.zero_grad()
            run(a, x).backward()
            assert [gi for gi, _ in reducer.launched] == [2, 1, 0]          # all three left before backward() returned
            assert [w for _, w in reducer.launched][0] > 0 and reducer.launched[-1][1] == 0
            reducer.finish()
            oa.step()
            ob.zero_grad()
            run(b, x).backward()
            ob.step()
    finally:
        dist.all_reduce = orig
        reducer.remove()
    assert len(calls) == 6
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)


@pytest.mark.parametrize("world", [2, 8])
def test_shards_of_a_global_batch_reproduce_the_full_batch_step(world):
    """What `bench.py --gpus N` relies on (no data-path collective): the config-5 layer run on each rank's shard of ONE
    global batch (bench.shard_of = parallel.shard_graphs on the augmented sizes) gives the full-batch outputs row for row, and the
    shards' parameter gradients add up to the full batch's -- here with all shards on the one GPU, bf16 pipeline, 1,024 graphs."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    dev = torch.device(DEV)
    H, R, graphs = 256, 16, 1024
    torch.manual_seed(3)
    layer = RGINLayer(H, H, num_rels=R, regularizer="basis", num_bases=-1, num_mlp_layers=2, act_func="relu").to(dev).to(torch.bfloat16)
    params = [p for p in layer.parameters()]

    def run(gb, x, go):
        for p in params:
            p.grad = None
        xs = x.clone().requires_grad_(True)
        out, _ = layer(gb, xs, gb.edata["label"])
        out.backward(go)
        return out.detach().float(), xs.grad.float(), [p.grad.float().clone() for p in params]

    full, raw, _ = bench.build_batch(dev, 5, graphs, "config5")
    N = full.number_of_nodes()
    gen = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(N, H, device=dev, generator=gen).to(torch.bfloat16)
    go = torch.randn(N, H, device=dev, generator=gen).to(torch.bfloat16)
    out_f, gx_f, gp_f = run(full, x, go)
    outs, gxs, gps, covered = [], [], None, 0
    for r in range(world):
        shard, _, _ = bench.build_batch(dev, 5, graphs, "config5", shard=(r, world))
        n = shard.number_of_nodes()
        o, gx, gp = run(shard, x[covered:covered + n], go[covered:covered + n])      # shards are contiguous graph (= node) ranges
        outs.append(o); gxs.append(gx)
        gps = gp if gps is None else [a + b for a, b in zip(gps, gp)]
        covered += n
    assert covered == N
    out_s, gx_s = torch.cat(outs), torch.cat(gxs)
    # the same rows go through the same arithmetic; only tile boundaries (bf16 rounding points of the per-graph fold sums) move
    assert float((out_s - out_f).abs().max() / out_f.abs().max()) < 2e-2
    assert float((out_s - out_f).norm() / out_f.norm()) < 2e-3
    assert float((gx_s - gx_f).norm() / gx_f.norm()) < 2e-3
    for a, b, p in zip(gps, gp_f, params):
        assert float((a - b).norm() / b.norm().clamp_min(1e-6)) < 1e-2, tuple(p.shape)      # sums of bf16-rounded per-shard gradients
