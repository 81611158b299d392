#!/usr/bin/env python3
"""GPU timing of a whole SI representation net -- RGINRepNet, 3 layers, residual, H = 64, R = 8 on BASELINE config 3 (512 graphs x 50
nodes, E = 102,400) -- forward + mse + backward (no optimizer), eager and under HIP-graph replay, fp32 (the reference's precision) and
bf16: what a reference training step spends in the rep net (subgraph_isomorphism/models/rgin.py:179-260).
usage: python tools/si_model_bench.py [--dtype f32|bf16]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dummynode4graphlearning_amd import BatchedGraph, synthetic, transforms  # noqa: E402
from dummynode4graphlearning_amd.subgraph_isomorphism import RGINRepNet  # noqa: E402

dev = torch.device("cuda:0")
raw = synthetic.config3()
keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(dev) for k in keys), raw["max_nv"], raw["max_nvl"], raw["max_ne"], raw["max_nel"])
N, E = int(aug["node_label"].numel()), int(aug["src"].numel())
bnn = (aug["node_ptr"][1:] - aug["node_ptr"][:-1]).long()
bne = (aug["edge_ptr"][1:] - aug["edge_ptr"][:-1]).long()
for dt in ([torch.float32, torch.bfloat16] if "--dtype" not in sys.argv else [torch.float32 if sys.argv[sys.argv.index("--dtype") + 1] == "f32" else torch.bfloat16]):
    g = BatchedGraph(aug["src"], aug["dst"], N, bnn, bne, edata={"label": aug["edge_label"]}, node_ptr=aug["node_ptr"], edge_ptr=aug["edge_ptr"])
    torch.manual_seed(5)
    net = RGINRepNet(64, 8, num_layers=3, regularizer="basis", act_func="relu").to(dev).to(dt)
    x = torch.randn(N, 64, device=dev).to(dt)
    tgt = torch.randn(N, 64, device=dev).to(dt)

    def step():
        for p in net.parameters():
            p.grad = None
        F.mse_loss(net.get_graph_rep(g, x), tgt).backward()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50
    gr = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(gr):
            step()
        gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            gr.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / 100
    except Exception as exc:
        rep = float("nan")
        print("   (HIP graph capture failed: %s)" % str(exc).splitlines()[0])
    print("config 3 RGINRepNet 3 layers H=64 %s, N=%d E=%d: eager %.3f ms (%.1f M edges/s), HIP-graph replay %.3f ms (%.1f M edges/s)"
          % ("fp32" if dt == torch.float32 else "bf16", N, E, eager * 1e3, E / eager / 1e6, rep * 1e3, E / rep / 1e6), flush=True)
