#!/usr/bin/env python3
"""bench.py -- edges/s forward+backward of one dummy-augmented RGIN conv layer on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
      N > 1 and no RANK in the environment: bench.py starts `python -m torch.distributed.run --nproc-per-node N` on itself
      (a fresh child process tree, before this process touches the GPU) -- one rank per GPU over RCCL; launched by
      torch.distributed.run directly (the driver's form), it checks --gpus == WORLD_SIZE.
  python bench.py --workload config4 --gpus N            BASELINE config 4: GIN hidden 256 on NCI1-shaped dummy graphs,
      512 graphs per GPU cut from one global batch by parallel.shard_graphs, SyncBatchNorm statistics, flat-bucket all-reduce

Workload (config.workload): BASELINE.json configs[4] = SURVEY.md 8(d) config 5, the one the metric is quoted on
(fits one GPU): synthetic SI-style batch of 32 768 graphs x (30+1) nodes, 62 real + 60 dummy edges, R = 16
=> N = 1 015 808, E = 3 997 696; RGINLayer(256, 256, basis, full) in bf16 storage / fp32 accumulate.
STRONG scaling (default, what SURVEY 8(d)/8(e) define for config 5): the ONE global batch (seed 5, identical on every
rank) is cut into contiguous graph ranges by parallel.shard_graphs -- 4 096 graphs per GPU at N = 8 -- every rank builds
its own index and runs its shard with no data-path collective; the only collective is the flat gradient all-reduce (RCCL).
value = global E / max-over-ranks step time.  `--scaling weak` keeps round 2's form (every rank its own full batch, seed 5 +
rank).  A step = layer forward + backward (dx and all parameter gradients) + all-reduce, inputs resident in HBM; the
one-shot index build of the batch is outside the timed region and reported separately.  At N = 1 the line also carries
config.strong_scaling_proxy: the step and the fresh-batch index build of ONE eighth of the batch on this GPU and the
efficiency t(batch) / (8 t(eighth)) an 8-GPU run could reach before the all-reduce.

Rank 0 prints ONE JSON line with `roofline` (gather/segment-sum kernel, HIP events on the launch stream) and
`cpu_baseline` (oracle port timed on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def shard_of(raw, rank, world):
    """Rank `rank`'s contiguous graph range of the global batch `raw` (parallel.shard_graphs on the sizes AFTER the dummy
    augmentation: n + 1 nodes, m + 2 n edges per graph), as a batch of its own (ids relative to the shard)."""
    from dummynode4graphlearning_amd.parallel import shard_graphs
    bnn = raw["node_ptr"][1:] - raw["node_ptr"][:-1]
    bne = raw["edge_ptr"][1:] - raw["edge_ptr"][:-1]
    g0, g1 = shard_graphs(torch.from_numpy(bnn + 1), torch.from_numpy(bne + 2 * bnn), world)[rank]
    if (g0, g1) == (0, len(bnn)):
        return raw, (g0, g1)
    n0, n1, e0, e1 = (int(raw[k][g]) for k, g in (("node_ptr", g0), ("node_ptr", g1), ("edge_ptr", g0), ("edge_ptr", g1)))
    sub = dict(raw)
    sub.update(node_ptr=raw["node_ptr"][g0:g1 + 1] - n0, edge_ptr=raw["edge_ptr"][g0:g1 + 1] - e0,
               src=raw["src"][e0:e1] - n0, dst=raw["dst"][e0:e1] - n0, edge_label=raw["edge_label"][e0:e1],
               edge_id=raw["edge_id"][e0:e1], node_id=raw["node_id"][n0:n1], node_label=raw["node_label"][n0:n1])
    return {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in sub.items()}, (g0, g1)


def build_batch(dev, seed, graphs, workload, shard=None):
    """shard = (rank, world): only that rank's graphs of the global batch go to the device."""
    from dummynode4graphlearning_amd import BatchedGraph, synthetic, transforms
    raw = {"config5": synthetic.config5, "config3": synthetic.config3, "proteins": synthetic.proteins_si}[workload](seed, graphs)
    if shard is not None:
        raw, _ = shard_of(raw, *shard)
    t = {k: torch.from_numpy(v).to(dev) for k, v in raw.items() if isinstance(v, np.ndarray)}
    aug_ms = []
    for _ in range(3):                                   # first call: code-object load + allocator growth; then steady state
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        aug = transforms.dummy_augment_si(t["node_ptr"], t["edge_ptr"], t["src"], t["dst"], t["node_id"], t["node_label"],
                                          t["edge_id"], t["edge_label"], raw["max_nv"], raw["max_nvl"], raw["max_ne"],
                                          raw["max_nel"])
        torch.cuda.synchronize()
        aug_ms.append((time.perf_counter() - t0) * 1e3)
    aug_ms = (aug_ms[0], min(aug_ms[1:]))
    N = int(aug["node_label"].numel())
    bnn = (aug["node_ptr"][1:] - aug["node_ptr"][:-1]).long()
    bne = (aug["edge_ptr"][1:] - aug["edge_ptr"][:-1]).long()
    g = BatchedGraph(aug["src"], aug["dst"], N, bnn, bne, ndata={"id": aug["node_id"], "label": aug["node_label"]},
                     edata={"id": aug["edge_id"], "label": aug["edge_label"]}, node_ptr=aug["node_ptr"], edge_ptr=aug["edge_ptr"])
    return g, raw, aug_ms


def cpu_baseline(raw, H, R, budget_s=12.0, sample_graphs=2048):
    """Oracle port of RGINLayer (per-edge messages grouped by relation, fp32, torch CPU) on a bounded sample."""
    from oracle import layers as OL
    from oracle import transforms as OT
    G = min(sample_graphs, len(raw["node_ptr"]) - 1)
    n1, e1 = int(raw["node_ptr"][G]), int(raw["edge_ptr"][G])
    aug = OT.dummy_augment_si(raw["node_ptr"][:G + 1], raw["edge_ptr"][:G + 1], raw["src"][:e1], raw["dst"][:e1],
                              raw["node_id"][:n1], raw["node_label"][:n1], raw["edge_id"][:e1], raw["edge_label"][:e1],
                              raw["max_nv"], raw["max_nvl"], raw["max_ne"], raw["max_nel"])
    cores = min(os.cpu_count() or 1, 64)     # torch CPU GEMMs stop scaling (and start thrashing) far below 256 threads
    torch.set_num_threads(cores)
    N, E = len(aug["node_label"]), len(aug["src"])
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, generator=g).requires_grad_(True)
    p = {"weight": (torch.randn(R, H, H, generator=g) * 0.05).requires_grad_(True),
         "loop_weight": (torch.randn(H, H, generator=g) * 0.05).requires_grad_(True),
         "bias": torch.zeros(H, requires_grad=True),
         "mlp.0.weight": (torch.randn(H, H, generator=g) * 0.05).requires_grad_(True), "mlp.0.bias": torch.zeros(H, requires_grad=True),
         "mlp.2.weight": (torch.randn(H, H, generator=g) * 0.05).requires_grad_(True), "mlp.2.bias": torch.zeros(H, requires_grad=True)}
    src, dst, et = (torch.from_numpy(aug[k]) for k in ("src", "dst", "edge_label"))
    gout = torch.randn(N, H, generator=g)

    def one():
        out = OL.rgin_layer_rel_grouped(x, src, dst, et, p, R, act="relu", num_mlp_layers=2)
        out.backward(gout)
        x.grad = None
        for t in p.values():
            t.grad = None

    one()
    t0 = time.perf_counter()
    it = 0
    while True:
        one()
        it += 1
        if time.perf_counter() - t0 > budget_s or it >= 50:
            break
    dt = (time.perf_counter() - t0) / it
    return {"value": E / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": "first %d graphs of the same batch (N=%d, E=%d), fp32, %d iterations of "
                      "oracle.layers.rgin_layer_rel_grouped fwd+bwd on %d torch threads (host has %d logical CPUs)"
                      % (G, N, E, it, cores, os.cpu_count() or 1)}


def _free_port():
    import socket
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        return sck.getsockname()[1]


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: run N ranks of this file under torch.distributed.run as a CHILD process tree and
    exit with its code.  Called before anything in this process has touched the GPU (no torch.cuda call yet); the ranks
    are fresh interpreters, nothing is re-exec'ed."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """CPU-only plumbing check (tests/test_dist.py): the ranks rendezvous over gloo, prove the world with one all-reduce
    and rank 0 prints the line's skeleton.  Nothing of the product path runs (it has no CPU fallback)."""
    if world > 1 or "RANK" in os.environ:
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    if dist.is_initialized():
        dist.all_reduce(t)
    # the split every rank would take (host-side bookkeeping only: parallel.shard_graphs on the synthetic batch's sizes)
    shards = None
    if args.workload in ("config5", "config3", "proteins") and args.scaling == "strong":
        from dummynode4graphlearning_amd import synthetic
        graphs = args.graphs or {"config5": 32768, "config3": 512, "proteins": 16384}[args.workload]
        raw = {"config5": synthetic.config5, "config3": synthetic.config3, "proteins": synthetic.proteins_si}[args.workload](
            {"config5": 5, "config3": 3, "proteins": 2}[args.workload], graphs)
        _, (g0, g1) = shard_of(raw, rank, world)
        mine = torch.tensor([g0, g1], dtype=torch.int64)
        allr = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        if dist.is_initialized():
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        shards = [[int(a[0]), int(a[1])] for a in allr]
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rank_sum": float(t.item()), "steps": args.steps,
                          "warmup": args.warmup, "workload": args.workload, "scaling": args.scaling,
                          "shard_graphs": shards}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def run_config4(args, rank, world, dev, emit=True):
    """BASELINE config 4: NCI1-shaped graphs + dummy nodes, GIN hidden 256 (default 2 layers), data parallel.
    One global batch of 512 x world graphs is cut into contiguous shards by parallel.shard_graphs (balanced by nodes +
    edges); every rank augments and runs its shard; BatchNorm statistics span the global batch (SyncBatchNorm1d); one
    flat-bucket gradient all-reduce per step.  A step = zero_grad -> forward -> nll_loss -> backward -> all-reduce."""
    from types import SimpleNamespace
    import torch.nn.functional as F
    from dummynode4graphlearning_amd import GraphBatch, synthetic, transforms
    from dummynode4graphlearning_amd import graph_classification as GC
    from dummynode4graphlearning_amd.parallel import FlatGradBucket, convert_sync_batchnorm, dp_loss_scale, shard_graphs
    per_gpu = args.graphs or 512
    H = args.hidden or 256
    raw = synthetic.config4(seed=4, graphs=per_gpu * world)                      # identical on every rank
    bnn = raw["node_ptr"][1:] - raw["node_ptr"][:-1]
    bne = raw["edge_ptr"][1:] - raw["edge_ptr"][:-1]
    g0, g1 = shard_graphs(torch.from_numpy(bnn + 1), torch.from_numpy(bne + 2 * bnn), world)[rank]   # sizes after augmentation
    n0, n1, e0, e1 = (int(raw[k][g]) for k, g in (("node_ptr", g0), ("node_ptr", g1), ("edge_ptr", g0), ("edge_ptr", g1)))
    sub = dict(node_ptr=raw["node_ptr"][g0:g1 + 1] - n0, edge_ptr=raw["edge_ptr"][g0:g1 + 1] - e0, src=raw["src"][e0:e1] - n0,
               dst=raw["dst"][e0:e1] - n0, node_label=raw["node_label"][n0:n1], edge_label=raw["edge_label"][e0:e1])
    aug = transforms.dummy_augment_gc(*(torch.from_numpy(np.ascontiguousarray(sub[k])).to(dev) for k in
                                        ("node_ptr", "edge_ptr", "src", "dst", "node_label", "edge_label")))
    Fdim = 38                                                                     # 37 node labels + dummy label 0, one-hot
    x = F.one_hot(aug["node_label"].long(), Fdim).float()
    node_ptr = aug["node_ptr"].long()
    G = g1 - g0
    batch = torch.repeat_interleave(torch.arange(G, device=dev), node_ptr[1:] - node_ptr[:-1])
    y = torch.from_numpy(np.random.default_rng(40).integers(0, 2, size=per_gpu * world)[g0:g1]).to(dev)
    data = GraphBatch(x, torch.stack([aug["src"].long(), aug["dst"].long()]), batch, y=y, ptr=node_ptr)
    N, E = int(x.shape[0]), int(aug["src"].numel())
    margs = SimpleNamespace(num_features=Fdim, hidden_dim=H, num_classes=2, dropout_ratio=0.0, num_relations=2,
                            additional=None, epochs=1, device=dev, dummy_weight=0)
    torch.manual_seed(1234)                                                       # same initial replica on every rank
    model = convert_sync_batchnorm(GC.GIN(margs)).to(dev).train()
    bucket = FlatGradBucket(model.parameters())
    scale = dp_loss_scale(G, per_gpu * world, world)

    def compute():
        bucket.zero(set_to_none=True)
        loss = F.nll_loss(model(data), data.y) * scale
        loss.backward()
        bucket.pack()
        return loss

    # one rank: forward + loss + backward replayed from a HIP graph (the step is ~150 launches of 5-30 us); with more ranks the
    # SyncBatchNorm collectives sit inside forward / backward, so the launches stay eager
    graphed = None
    if world == 1 and not args.no_graph:
        from dummynode4graphlearning_amd import StepGraph
        graphed = StepGraph(compute, warmup=max(args.warmup, 2))
        if not graphed.captured:
            graphed = None

    def step():
        loss = graphed() if graphed is not None else compute()
        bucket.all_reduce()
        return loss

    for _ in range(max(args.warmup, 2)):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tot = torch.tensor([float(E), float(N)], device=dev, dtype=torch.float64)
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot)
    if rank == 0:
        ms = float(tt.item()) / args.steps * 1e3
        gE, gN = float(tot[0].item()), float(tot[1].item())
        # algorithmic bytes of the model's ONE GIN conv (default config: 2 layers = first_h + one GINConv), fwd + bwd
        alg = 2.0 * (gE * H * 4 + gN * H * 4 + 8.0 * gE)
        line = {
            "metric": "edges/sec fwd+bwd on dummy-augmented GIN model step (config 4)", "value": gE / (ms * 1e-3),
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config4: GIN(hidden=%d, 2 layers) training step (fwd + nll_loss + bwd + all-reduce) on "
                                   "%d NCI1-shaped dummy graphs per GPU (global N=%d, E=%d), SyncBatchNorm, %s"
                                   % (H, per_gpu, int(gN), int(gE), "HIP-graph replay" if graphed is not None else "eager launches"),
                       "global_edges": int(gE), "parallelism": "dp%d" % world, "grad_bucket_bytes": bucket.bytes(),
                       "shard_graphs": [g0, g1], "hip_graph": graphed is not None},
            "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": alg / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), "traffic": None,
                         "note": "whole training step of a 16 k-node batch per GPU: launch-bound, not a kernel roofline"},
        }
        if emit:
            print(json.dumps(line), flush=True)
        return line
    return None


def rgin_leg(dev, workload, dtype, steps, warmup, act="relu", regularizer="basis"):
    """One more configuration on the N = 1 line (`secondary`): the RGINLayer step of `workload` in `dtype` -- batch, per-batch
    index, captured step (forward + backward + bucket pack, replayed) and the conv's gather-scatter launches replayed on their
    own, exactly as the main leg measures config 5 -- reduced to {ms, edges_per_s, launches_per_step, index_build_ms, roofline}."""
    from dummynode4graphlearning_amd import ops
    from dummynode4graphlearning_amd.parallel import FlatGradBucket
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    from dummynode4graphlearning_amd.subgraph_isomorphism.rgin import dense_relation_weights
    H, R, graphs, seed = {"config3": (64, 8, 512, 3), "proteins": (256, 16, 16384, 2), "config5": (256, 16, 32768, 5)}[workload]
    s = 2 if dtype == torch.bfloat16 else 4
    g, raw, _ = build_batch(dev, seed, graphs, workload)
    N, E = g.number_of_nodes(), g.number_of_edges()
    et = g.edata["label"]
    torch.manual_seed(1234)
    layer = RGINLayer(H, H, num_rels=R, regularizer=regularizer, num_bases=4 if regularizer == "bdd" else -1, num_mlp_layers=2,
                      act_func=act).to(dev).to(dtype)
    bucket = FlatGradBucket(layer.parameters())
    gen = torch.Generator(device=dev).manual_seed(100)
    x = torch.randn(N, H, device=dev, generator=gen).to(dtype).requires_grad_(True)
    gout = torch.randn(N, H, device=dev, generator=gen).to(dtype)

    def build_index():
        g._cache.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix = g.row_index(et, R, True, closing_hint=(H, dtype))
        if dtype == torch.bfloat16:
            for _, _, part in ix.parts:
                ops.prepare_closing(part, H, dtype)
        torch.cuda.synchronize()
        return ix, (time.perf_counter() - t0) * 1e3

    index, _ = build_index()
    index, index_ms = build_index()

    def compute():
        bucket.zero(set_to_none=True)
        x.grad = None
        out, _ = layer(g, x, et)
        out.backward(gout)
        bucket.pack()

    def conv():
        with torch.no_grad():
            W = dense_relation_weights(layer)
            fw = ops.PassWeights(W, layer.loop_weight, kn=True)
            if not ops._kn_ok(x):
                fw = fw.nk()
            bw = ops.PassWeights(W, layer.loop_weight, kn=False)
            ybuf = index.ybuf(H, dtype, dev)
            for n0, n1, ix in index.parts:
                ops.message_pass(x[n0:n1], fw, layer.bias, ix, "f", ybuf, cg_out[n0:n1])
                ops.message_pass(gout[n0:n1], bw, None, ix, "b", ybuf, cg_out[n0:n1])

    def graphed(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                fn()
            gr.replay()
            torch.cuda.synchronize()
            return gr.replay, True
        except Exception as exc:
            sys.stderr.write("[bench] HIP graph capture failed in a secondary leg (%s); timing eager launches\n" % exc)
            torch.cuda.synchronize()
            return fn, False

    def timed(fn, reps, warm):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for _ in range(2):
        compute()
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.kernel_timer = timer
    compute()                                           # eager pass: the library launches of the whole step
    ops.kernel_timer = None
    step_launches = len(timer.records)
    step, is_graph = graphed(compute)
    # (small steps: more replays for the same GPU time, as for the strong-scaling proxy)
    mult = 8 if N < 200000 else 1
    ms = timed(step, mult * steps, mult * warmup)
    cg_out = torch.empty((N, H), dtype=dtype, device=dev)
    conv()
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.kernel_timer = timer
    conv()
    ops.kernel_timer = None
    n_launch = len(timer.records)
    replay_conv, _ = graphed(conv)
    kms = timed(replay_conv, mult * max(steps, 5), 2)
    alg = 2.0 * (E * H * s + N * H * s + 8.0 * E)
    ach = alg / (kms * 1e-3) / 1e9
    return {"workload": "%s: RGINLayer(%d,%d,R=%d,%s,%s) fwd+bwd, %d graphs, N=%d, E=%d, %s, SI dummy augmentation"
                        % (workload, H, H, R, regularizer, act, graphs, N, E, "bf16" if dtype == torch.bfloat16 else "f32"),
            "dtype": "bf16" if dtype == torch.bfloat16 else "f32", "ms": ms, "edges_per_s": E / (ms * 1e-3),
            "index_build_ms": index_ms, "hip_graph": is_graph, "launches_per_step": step_launches, "conv_launches_per_step": n_launch,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None, "launches_per_step": n_launch, "kernel_ms_per_step": kms, "alg_bytes_per_step": alg,
                         "note": "the conv's gather-scatter launches (both directions) replayed on their own, HIP events"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config5", choices=["config5", "config3", "config4", "proteins"],
                    help="proteins: the config-5 layer (H = 256, R = 16, bf16) on 16384 PROTEINS-shaped graphs (mean 39, up to 620 nodes): "
                         "graphs over 32 nodes, where the closing launch keeps partial rows + the fold tail")
    ap.add_argument("--dry-run", action="store_true", help="CPU-only launch check: rendezvous over gloo, no product code")
    ap.add_argument("--graphs", type=int, default=0, help="graphs in the global batch (strong) / per GPU (weak); 0 = the workload's own size")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="config5 / config3: strong = ONE global batch cut by parallel.shard_graphs (default); weak = a full batch per rank")
    ap.add_argument("--no-proxy", action="store_true", help="skip the one-GPU strong-scaling proxy (an eighth of the batch)")
    ap.add_argument("--steady", action="store_true",
                    help="profiling aid: the replayed step and the roofline leg only (no fresh-batch leg, GIN leg, proxy, CPU baseline)")
    ap.add_argument("--dtype", default="", choices=["", "bf16", "f32"])
    ap.add_argument("--hidden", type=int, default=0, help="override the workload's hidden size (experiments only)")
    ap.add_argument("--act", default="relu", choices=["relu", "leaky_relu"],
                    help="activation of the layer's MLP (the reference CLI's default is leaky_relu, config.py:329-335)")
    ap.add_argument("--regularizer", default="basis", choices=["basis", "bdd"],
                    help="relation-weight parameterisation (the reference CLI's default is bdd with 4 bases, config.py:145-158)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config 3 / config 4 / proteins legs of the default line")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    args = ap.parse_args()
    if args.steady:
        args.no_proxy = args.no_cpu_baseline = True

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args))             # nothing in this process has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # torch.distributed.run (even with one rank)
    if world > 1 or launched:
        dist.init_process_group("nccl", device_id=dev)
    if args.workload == "config4":
        run_config4(args, rank, world, dev)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    from dummynode4graphlearning_amd import ops
    from dummynode4graphlearning_amd.parallel import FlatGradBucket
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer

    if args.workload == "config5":
        H, R, graphs, dtype = 256, 16, args.graphs or 32768, torch.bfloat16
    elif args.workload == "proteins":
        H, R, graphs, dtype = 256, 16, args.graphs or 16384, torch.bfloat16
    else:
        H, R, graphs, dtype = 64, 8, args.graphs or 512, torch.float32
    if args.hidden:
        H = args.hidden
    if args.dtype:
        dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    s = 2 if dtype == torch.bfloat16 else 4

    strong = args.scaling == "strong"
    seed0 = {"config5": 5, "config3": 3, "proteins": 2}[args.workload]
    if strong:                                  # ONE global batch, this rank's contiguous graph range of it
        g, raw, aug_ms = build_batch(dev, seed0, graphs, args.workload, shard=(rank, world))
    else:                                       # weak: every rank its own full batch
        g, raw, aug_ms = build_batch(dev, seed0 + rank, graphs, args.workload)
    N, E = g.number_of_nodes(), g.number_of_edges()
    etype = g.edata["label"]
    torch.manual_seed(1234)
    layer = RGINLayer(H, H, num_rels=R, regularizer=args.regularizer, num_bases=4 if args.regularizer == "bdd" else -1,
                      num_mlp_layers=2, act_func=args.act).to(dev).to(dtype)
    bucket = FlatGradBucket(layer.parameters())

    fused = H in (64, 128, 256)                 # row-factorised MFMA pipeline (bf16 and exact-f32) vs generic two-pass path
    use_graph = not args.no_graph

    def prepare(gb, seed):
        """Index build (timed, twice), inputs and the captured step of one batch on this GPU.
        -> dict(step=callable, graph=CUDAGraph or None, index=..., index_ms=(first, steady), x=..., gout=...)."""
        et = gb.edata["label"]
        n = gb.number_of_nodes()
        gen = torch.Generator(device=dev).manual_seed(seed)
        xb = torch.randn(n, H, device=dev, generator=gen).to(dtype).requires_grad_(True)
        gob = torch.randn(n, H, device=dev, generator=gen).to(dtype)

        def build_index():
            gb._cache.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix = gb.row_index(et, R, True, closing_hint=(H, dtype)) if fused else gb.rel_index(et, R)
            if fused and dtype == torch.bfloat16:   # the closing tables and tile orders are part of the per-batch index cost
                for _, _, part in ix.parts:
                    ops.prepare_closing(part, H, dtype)
            torch.cuda.synchronize()
            return ix, (time.perf_counter() - t0) * 1e3

        ix, first_ms = build_index()            # first call: allocator growth + code-object load
        ix, steady_ms = build_index()           # steady state (what a training loop pays per new batch)

        def compute():
            bucket.zero(set_to_none=True)       # optimizer.zero_grad()'s default (train.py:836): gradients are written, not added
            xb.grad = None
            out, _ = layer(gb, xb, et)
            out.backward(gob)
            bucket.pack()                       # the step's gradients -> the flat bucket (one launch), inside the captured step

        def step_eager():
            compute()
            bucket.all_reduce()

        # The step is launch-bound from Python (a few hundred small launches): capture it once into a HIP graph and replay it.
        # The gradient all-reduce (RCCL) stays outside the graph, on the same stream right after the replay.
        for _ in range(max(args.warmup, 2)):
            step_eager()
        torch.cuda.synchronize()
        gr = None
        if use_graph:
            gr = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                compute()                               # warm the private pool on the capture stream
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            # thread_local: the RCCL watchdog thread of a multi-rank run may touch the runtime while this thread captures
            try:
                with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                    compute()
            except Exception as exc:                    # never lose the measurement to a capture problem: run eagerly
                sys.stderr.write("[bench] HIP graph capture failed (%s); timing eager launches\n" % exc)
                torch.cuda.synchronize()
                gr = None

        def step():
            if gr is not None:
                gr.replay()
                bucket.all_reduce()
            else:
                step_eager()

        return dict(step=step, graph=gr, index=ix, index_ms=(first_ms, steady_ms), x=xb, gout=gob)

    main = prepare(g, 100 + rank)
    step, graph, index, x, gout = main["step"], main["graph"], main["index"], main["x"], main["gout"]
    index_first_ms, index_ms = main["index_ms"]

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    e_glob = float(E)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        te = torch.tensor([float(E)], device=dev, dtype=torch.float64)
        dist.all_reduce(te)
        e_glob = float(te.item())               # strong: the global batch's edges; weak: world x E

    # roofline leg: the conv's gather-scatter launches alone (gather/segment-sum + gathered-row transform of the forward
    # and of the input-gradient pass, exactly the launches the step makes), replayed as their own HIP graph and timed with
    # HIP events on the launch stream
    # (measured HERE, right behind the timed steps and in their memory state: behind the proxy and the fresh-batch leg -- a second
    #  batch, a second index, a few GB of allocator traffic -- the same four launches ran 2-6 % slower on the same box than
    #  in the step they belong to, whose own kernel trace agrees with this placement: profiles/r05_kernel_stats_step.csv vs _leg.csv)
    def conv_gather_scatter():
        with torch.no_grad():
            from dummynode4graphlearning_amd.subgraph_isomorphism.rgin import dense_relation_weights
            W = dense_relation_weights(layer)                       # (basis with num_bases == R: the parameter itself)
            if fused:
                # exactly what _RowTransformFn issues: the forward pass on the parameters as they are stored ([k][n]: no cat /
                # transpose launches at H = 256 bf16), the input-gradient pass on W itself
                fw = ops.PassWeights(W, layer.loop_weight, kn=True)
                if not ops._kn_ok(x):
                    fw = fw.nk()
                bw = ops.PassWeights(W, layer.loop_weight, kn=False)
                ybuf = index.ybuf(H, dtype, dev)
                for n0, n1, ix in index.parts:
                    ops.message_pass(x[n0:n1], fw, layer.bias, ix, "f", ybuf, cg_out[n0:n1])
                    ops.message_pass(gout[n0:n1], bw, None, ix, "b", ybuf, cg_out[n0:n1])
            else:
                A = ops.gather_segsum(x, index.src1, index.seg_ptr, index.num_segments)
                ops.gather_segsum(A, index.sperm, index.dptr, N)
                gy = ops.gather_segsum(gout, index.seg_dst, None)
                ops.gather_segsum(gy, index.seg_by_src, index.optr, N)

    cg_out = torch.empty((N, H), dtype=dtype, device=dev)
    conv_gather_scatter()
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.kernel_timer = timer
    conv_gather_scatter()                               # eager pass: counts the launches (times include launch gaps)
    ops.kernel_timer = None
    n_launch = len(timer.records)
    def graphed(fn):
        """fn replayed from a HIP graph; falls back to the eager callable if the capture fails."""
        try:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                fn()
            gr.replay()
            torch.cuda.synchronize()
            return gr.replay
        except Exception as exc:
            sys.stderr.write("[bench] HIP graph capture failed (%s); timing eager launches\n" % exc)
            torch.cuda.synchronize()
            return fn

    replay_conv = graphed(conv_gather_scatter)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(args.steps, 5)
    e0.record()
    for _ in range(reps):
        replay_conv()
    e1.record()
    torch.cuda.synchronize()
    kernel_ms_step = e0.elapsed_time(e1) / reps
    ms_per_step = dt / args.steps * 1e3

    # strong-scaling proxy (rank 0, one GPU): ONE eighth of the global batch (the shard rank 0 of an 8-GPU run takes) on this
    # GPU -- step under HIP-graph replay incl. bucket.pack(), fresh-batch index build -- and the efficiency an 8-GPU run could
    # reach before the gradient all-reduce: t(batch) / (8 t(eighth)).  No multi-GPU curve is measured here.
    proxy = None
    if rank == 0 and world == 1 and strong and not args.no_proxy and graphs >= 8:
        g8, raw8, aug8 = build_batch(dev, seed0, graphs, args.workload, shard=(0, 8))
        p8 = prepare(g8, 100)
        # (8 x the warm-up and the steps of the main leg: the same GPU time -- 20 steps of 0.55 ms behind a second of batch and
        #  index building measured the clocks' ramp as much as the step: 0.549-0.573 ms from run to run)
        for _ in range(8 * args.warmup):
            p8["step"]()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(8 * args.steps):
            p8["step"]()
        torch.cuda.synchronize()
        ms8 = (time.perf_counter() - t1) / (8 * args.steps) * 1e3
        ms1 = dt / args.steps * 1e3
        proxy = {"shard_graphs": int(g8.batch_size), "shard_edges": g8.number_of_edges(), "shard_ms_per_step": ms8,
                 "shard_index_build_ms": p8["index_ms"][1], "shard_dummy_augment_ms": aug8[1],
                 "batch_ms_per_step": ms1, "efficiency_at_8": ms1 / (8.0 * ms8), "predicted_speedup_at_8": ms1 / ms8,
                 "note": "one GPU, no collective: t(batch) / (8 t(eighth)); the 2.5 MB gradient all-reduce of an 8-GPU step is not in it"}
        del p8, g8
        # ... and with the collective: the RCCL all-reduce of the step's gradient bucket MEASURED in a world of one on this GPU
        # (launch + kernel latency of the call the 8-GPU step makes; nothing crosses a link), plus what a ring over xGMI adds for
        # these bytes (2 (W-1)/W x bytes over one 153 GB/s link at 80 %: MI355X_MICROARCH.md) -- a MODEL, no curve was measured
        if not dist.is_initialized():
            try:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
                flat = bucket.flat
                for _ in range(5):
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG)
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                ea.record()
                for _ in range(50):
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG)
                eb.record()
                torch.cuda.synchronize()
                ar1 = ea.elapsed_time(eb) / 50
                dist.destroy_process_group()
                ring = 2.0 * (7.0 / 8.0) * bucket.bytes() / (0.8 * 153e9) * 1e3
                proxy.update({"allreduce_world1_ms": ar1, "allreduce_ring_model_8gpu_ms": ar1 + ring,
                              "predicted_speedup_at_8_incl_allreduce": ms1 / (ms8 + ar1 + ring)})
            except Exception as exc:                    # (never lose the line to the rendezvous)
                sys.stderr.write("[bench] world-of-one all-reduce not measured: %s\n" % exc)

    # fresh-batch leg (rank 0, one GPU): a training loop sees a new batch every step, so it pays the dummy augmentation and the
    # index build per step.  Sequentially that is `edges_per_s_incl_index_build`; here the NEXT batch's augmentation + index
    # build run on a side stream while the current step replays (what the reference's DataLoader workers do for dgl.batch on
    # the CPU): per-step wall time of the overlapped loop.
    overlapped_ms = None
    if rank == 0 and world == 1 and graph is not None and fused and dtype == torch.bfloat16 and not args.steady:
        from dummynode4graphlearning_amd import transforms as _tr
        side2 = torch.cuda.Stream()                     # (a high-priority side stream changes nothing: 0.88 vs 0.89 G edges/s)
        traw = {k: torch.from_numpy(v).to(dev) for k, v in raw.items() if isinstance(v, np.ndarray)}

        def next_batch_index():
            with torch.cuda.stream(side2):
                aug = _tr.dummy_augment_si(traw["node_ptr"], traw["edge_ptr"], traw["src"], traw["dst"], traw["node_id"],
                                           traw["node_label"], traw["edge_id"], traw["edge_label"], raw["max_nv"], raw["max_nvl"],
                                           raw["max_ne"], raw["max_nel"])
                g._cache.clear()
                ix = g.row_index(etype, R, True, closing_hint=(H, dtype))
                for _, _, part in ix.parts:
                    ops.prepare_closing(part, H, dtype)
                return aug, ix

        for _ in range(2):
            step()
            next_batch_index()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()                                      # replay of the captured step (its own index tables are kept alive by the graph)
            next_batch_index()                          # host-side syncs of the build overlap the replay
        torch.cuda.synchronize()
        overlapped_ms = (time.perf_counter() - t1) / args.steps * 1e3


    # roofline: algorithmic bytes of the layer's gather-scatter forward+backward (SURVEY.md 8d:
    # 2*(E*H*s + N*H*s + 8*E)) over the time of the launches that implement it
    launches_per_step = n_launch
    alg_bytes_step = 2.0 * (E * H * s + N * H * s + 8.0 * E)
    achieved = alg_bytes_step / (kernel_ms_step * 1e-3) / 1e9 if kernel_ms_step > 0 else 0.0

    # HBM traffic of those launches comes from PMC counters, which cannot be read from inside the process: it is taken
    # from the committed rocprofv3 measurement of this exact workload (profiles/rNN_traffic.json), else null
    traffic = None
    from dummynode4graphlearning_amd._lib import source_digest
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):                    # newest committed measurement of this exact workload AND code
        try:
            with open(os.path.join(ROOT, "profiles", tag + "_traffic.json")) as f:
                tj = json.load(f)
            if ((tj["workload"], tj["N"], tj["E"], tj["H"], tj["dtype"]) == (args.workload, N, E, H, "bf16" if dtype == torch.bfloat16 else "f32")
                    and tj.get("source_sha16") == source_digest()):
                traffic = tj["conv_gather_scatter_hbm_bytes_per_step"]
                break
        except Exception:
            continue

    # secondary line (rank 0, N = 1): the pure gather -> segment-sum kernel on a GIN conv (no relation transform), forward
    # + backward, on a PROTEINS-shaped dummy-augmented batch (SURVEY 8d config 2 x 32 graphs, fp32 H = 128)
    gin = None
    if rank == 0 and world == 1 and not args.steady:
        from dummynode4graphlearning_amd import synthetic as syn, transforms as tr
        r2 = syn.config2(graphs=16384)
        t2 = {k: torch.from_numpy(v).to(dev) for k, v in r2.items()}
        a2 = tr.dummy_augment_gc(t2["node_ptr"], t2["edge_ptr"], t2["src"], t2["dst"], t2["node_label"], t2["edge_label"])
        N2, E2, H2 = int(a2["node_label"].numel()), int(a2["src"].numel()), 128
        ei = ops.EdgeIndex(a2["src"], a2["dst"], N2, node_ptr=a2["node_ptr"])
        x2 = torch.randn(N2, H2, device=dev, requires_grad=True)
        go2 = torch.randn(N2, H2, device=dev)

        def gin_fb():
            x2.grad = None
            ops.neighbor_sum(x2, ei, 1.0).backward(go2)

        gin_fb()
        torch.cuda.synchronize()
        replay_gin = graphed(gin_fb)
        e0.record()
        for _ in range(20):
            replay_gin()
        e1.record()
        torch.cuda.synchronize()
        gms = e0.elapsed_time(e1) / 20
        # honest roofline of this kernel: COMPULSORY HBM bytes (every x / grad row read once, every output row written once,
        # the int32 index once, per direction) -- the E*H*s gathered bytes of the SURVEY formula are graph-local re-reads that
        # the L2 serves, so pricing them against the HBM peak would give a "fraction" above 1
        gcomp = 2.0 * (N2 * H2 * 4 + N2 * H2 * 4 + 8.0 * E2)
        ggath = 2.0 * (E2 * H2 * 4)
        gin = {"workload": "GIN conv gather+segment-sum fwd+bwd, 16384 PROTEINS-shaped dummy graphs, N=%d E=%d H=128 fp32" % (N2, E2),
               "ms": gms, "edges_per_s": E2 / (gms * 1e-3),
               "roofline": {"bound": "hbm", "achieved": gcomp / (gms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": gcomp / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, "compulsory_bytes": gcomp},
               "l2_gather_GBps": ggath / (gms * 1e-3) / 1e9,
               "note": "frac = compulsory HBM bytes / time / 8 TB/s; the gathered rows (l2_gather_GBps) are served by L2"}
        del x2, go2, ei, replay_gin

    # the other configurations of BASELINE.json on the same line (rank 0, N = 1, default workload only): config 3 in the reference's
    # own precision and in bf16, one rank of config 4, and the config-5 layer on PROTEINS-shaped graphs.  They run LAST: nothing
    # above is timed behind them.
    secondary = {}
    if rank == 0 and world == 1 and args.workload == "config5" and not args.steady and not args.no_secondary and not args.graphs:
        from types import SimpleNamespace
        for key, (wl, dt_) in (("config3_f32", ("config3", torch.float32)), ("config3_bf16", ("config3", torch.bfloat16)),
                               ("proteins_bf16", ("proteins", torch.bfloat16))):
            try:
                secondary[key] = rgin_leg(dev, wl, dt_, args.steps, args.warmup)
            except Exception as exc:                    # (never lose the line to a secondary leg)
                sys.stderr.write("[bench] secondary leg %s failed: %r\n" % (key, exc))
            torch.cuda.empty_cache()
        try:
            c4 = run_config4(SimpleNamespace(graphs=0, hidden=0, no_graph=False, warmup=args.warmup, steps=8 * args.steps), 0, 1, dev,
                             emit=False)
            secondary["config4_one_rank"] = {"workload": c4["config"]["workload"], "dtype": c4["dtype"], "ms": c4["ms_per_step"],
                                             "edges_per_s": c4["value"], "hip_graph": c4["config"]["hip_graph"],
                                             "launches_per_step": None, "roofline": c4["roofline"]}
        except Exception as exc:
            sys.stderr.write("[bench] secondary leg config4 failed: %r\n" % (exc,))

    if rank == 0:
        line = {
            "metric": "edges/sec fwd+bwd on dummy-augmented RGIN conv", "value": e_glob / (ms_per_step * 1e-3),
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "bf16" if dtype == torch.bfloat16 else "f32", "data": "synthetic",
            "config": {"workload": "%s: RGINLayer(%d,%d,R=%d,%s,%s) fwd+bwd, %s (rank 0: N=%d, E=%d), SI dummy augmentation"
                                   % (args.workload, H, H, R, args.regularizer, args.act,
                                      ("ONE global batch of %d graphs cut by parallel.shard_graphs over %d GPU(s)" % (graphs, world))
                                      if strong else ("%d graphs per GPU" % graphs), N, E),
                       "global_edges": int(e_glob), "parallelism": "dp%d" % world,
                       "shard_graphs_rank0": int(g.batch_size), "strong_scaling_proxy": proxy,
                       "rows_P": getattr(index, "num_rows", None) or index.num_segments, "index_build_ms": index_ms,
                       "index_build_first_call_ms": index_first_ms, "dummy_augment_ms": aug_ms[1],
                       "dummy_augment_first_call_ms": aug_ms[0],
                       # a training loop sees a NEW batch every step: dummy augmentation + index build + step, per fresh batch
                       "edges_per_s_incl_index_build": e_glob / ((ms_per_step + index_ms + aug_ms[1]) * 1e-3),
                       "fresh_batch_overlapped_ms_per_step": overlapped_ms,
                       "edges_per_s_fresh_batch_overlapped": (E / (overlapped_ms * 1e-3)) if overlapped_ms else None,
                       "grad_bucket_bytes": bucket.bytes(), "hip_graph": graph is not None,
                       "sub_batches": len(index.parts) if hasattr(index, "parts") else 1},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "rows_transform_ring_kernel + rows_close_ring_kernel (the conv's launches, both directions)",
                         "launches_per_step": launches_per_step, "kernel_ms_per_step": kernel_ms_step,
                         "alg_bytes_per_step": alg_bytes_step},
        }
        if gin is not None:
            line["secondary"] = {"gin_conv_gather": gin}
        if secondary:
            line.setdefault("secondary", {}).update(secondary)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(raw, H, R)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
