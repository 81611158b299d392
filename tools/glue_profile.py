#!/usr/bin/env python3
"""Which torch ops (not our launches) run in one eager config-5 layer step: python tools/glue_profile.py [--graphs 4096]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    a = ap.parse_args()
    import bench
    from torch.profiler import profile, ProfilerActivity
    from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
    from dummynode4graphlearning_amd import parallel
    dev = torch.device("cuda:0")
    gb, raw, _ = bench.build_batch(dev, 5, a.graphs, "config5")
    H, R = 256, 16
    torch.manual_seed(0)
    layer = RGINLayer(H, H, num_rels=R, regularizer="basis", num_bases=-1, num_mlp_layers=2).to(dev).to(torch.bfloat16)
    bucket = parallel.FlatGradBucket(layer.parameters())
    et = gb.edata["label"]
    n = gb.number_of_nodes()
    x = torch.randn(n, H, device=dev).to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(n, H, device=dev).to(torch.bfloat16)

    def step():
        bucket.zero(set_to_none=True)
        x.grad = None
        out, _ = layer(gb, x, et)
        out.backward(go)
        bucket.pack()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages() if e.device_time_total > 0 or getattr(e, "cuda_time_total", 0) > 0]
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))


if __name__ == "__main__":
    main()
