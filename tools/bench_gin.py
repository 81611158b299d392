#!/usr/bin/env python3
"""Secondary measurement (not the bench.py contract line): GIN conv gather/segment-sum on TU-shaped dummy-augmented
batches, forward + backward, HIP-graph replay.  (The CPU baseline of the contract lives in
bench.py's cpu_baseline leg only: tools never touch oracle/.)

  python tools/bench_gin.py [--config 2|4] [--graphs 512]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--graphs", type=int, default=512)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    from dummynode4graphlearning_amd import ops, synthetic, transforms
    dev = torch.device("cuda:0")
    raw = synthetic.config2(graphs=a.graphs) if a.config == 2 else synthetic.config4(graphs=a.graphs)
    H = 128 if a.config == 2 else 256
    t = {k: torch.from_numpy(v).to(dev) for k, v in raw.items()}
    aug = transforms.dummy_augment_gc(t["node_ptr"], t["edge_ptr"], t["src"], t["dst"], t["node_label"], t["edge_label"])
    N, E = int(aug["node_label"].numel()), int(aug["src"].numel())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index = ops.EdgeIndex(aug["src"], aug["dst"], N)
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    x = torch.randn(N, H, device=dev, requires_grad=True)
    gout = torch.randn(N, H, device=dev)

    def fb():
        x.grad = None
        ops.neighbor_sum(x, index, 1.0).backward(gout)

    for _ in range(3):
        fb()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fb()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    alg = 2.0 * (E * H * 4 + N * H * 4 + 8.0 * E)
    deg = np.bincount(aug["dst"].cpu().numpy(), minlength=N)
    print(json.dumps({"workload": "config%d GIN conv gather fwd+bwd, %d graphs, N=%d E=%d H=%d fp32" % (a.config, a.graphs, N, E, H),
                      "ms": ms, "edges_per_s": E / (ms * 1e-3), "alg_GBps": alg / (ms * 1e-3) / 1e9,
                      "max_in_degree": int(deg.max()), "hubs_split": 0 if index.fwd.hub_ids is None else int(index.fwd.hub_ids.numel()),
                      "index_build_ms": build_ms}))


if __name__ == "__main__":
    main()
