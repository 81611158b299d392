#!/usr/bin/env python3
"""GPU timing of whole GC model training steps (forward + nll_loss + backward, no optimizer) on BASELINE configs 1, 2 and 4
(one rank's share), eager and under HIP-graph replay -- the numbers BASELINE.md section 2 puts next to the reference's CPU timing."""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dummynode4graphlearning_amd import GraphBatch, synthetic, transforms  # noqa: E402
from dummynode4graphlearning_amd import graph_classification as GC  # noqa: E402

dev = torch.device("cuda:0")


def batch_of(raw, labels, F_):
    aug = transforms.dummy_augment_gc(*(torch.from_numpy(raw[k]).to(dev) for k in ("node_ptr", "edge_ptr", "src", "dst", "node_label", "edge_label")))
    x = F.one_hot(aug["node_label"].long(), labels + 1).float()
    if x.shape[1] < F_:
        x = torch.cat([torch.rand(x.shape[0], F_ - x.shape[1], device=dev), x], 1)
    node_ptr = aug["node_ptr"].long()
    G = node_ptr.numel() - 1
    batch = torch.repeat_interleave(torch.arange(G, device=dev), node_ptr[1:] - node_ptr[:-1])
    y = torch.randint(0, 2, (G,), device=dev)
    return GraphBatch(x, torch.stack([aug["src"].long(), aug["dst"].long()]), batch, y=y, ptr=node_ptr,
                      is_dummy_node=aug["is_dummy_node"].bool(), is_dummy_edge=aug["is_dummy_edge"].bool())


def run(name, raw, labels, F_, H, layers, kind="GIN"):
    data = batch_of(raw, labels, F_)
    extra = {"num_layers": layers, "train_eps": False}
    if kind == "GraphSAGE":
        extra["aggregation"] = "max"                # the segment-max kernels (dn_gather_segmax_*)
    args = SimpleNamespace(num_features=data.x.shape[1], hidden_dim=H, num_classes=2, dropout_ratio=0.0,
                           additional=extra, epochs=1, device=dev, dummy_weight=1.0 if kind == "GCN" else 0)   # GCN: the trainable dummy-edge weight (dn_edge_dot_*)
    torch.manual_seed(0)
    model = getattr(GC, kind)(args).to(dev).train()

    def step():
        for p in model.parameters():
            p.grad = None
        F.nll_loss(model(data), data.y).backward()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 30
    try:
        from dummynode4graphlearning_amd import StepGraph
        sg = StepGraph(step, warmup=1, fallback=False)      # (the product's capture: BatchNorm's two-launch form rides on its ticket word)
        sg()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            sg()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / 50
    except Exception as exc:
        rep = float("nan")
        print("   (HIP graph capture failed: %s)" % exc)
    E, N = data.edge_index.shape[1], data.x.shape[0]
    print("%s %s %d-layer H=%d step, N=%d E=%d: eager %.3f ms (%.1f M edges/s), HIP-graph replay %.3f ms (%.1f M edges/s)"
          % (name, kind, layers, H, N, E, eager * 1e3, E / eager / 1e6, rep * 1e3, E / rep / 1e6))


if "--f1" in sys.argv:              # the f-1 models: GCN with the trainable dummy-edge weight, GraphSAGE with max aggregation
    run("config 2", synthetic.config2(), 3, 5, 128, 2, kind="GCN")
    run("config 2", synthetic.config2(), 3, 5, 128, 2, kind="GraphSAGE")
else:
    run("config 1", synthetic.config1(), 7, 8, 64, 3)
    run("config 2", synthetic.config2(), 3, 5, 128, 2)
    run("config 4 (one rank)", synthetic.config4(), 37, 38, 256, 2)
