#!/usr/bin/env python3
"""Timing of the two dense (MLP) weight gradients of config 5 alone: python tools/mlp_wgrad_exp.py [--graphs 32768]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1015808)
    a = ap.parse_args()
    from dummynode4graphlearning_amd import ops
    dev = torch.device("cuda:0")
    N, H = a.rows, 256
    torch.manual_seed(0)
    g = torch.randn(N, H, device=dev).to(torch.bfloat16)
    h = torch.randn(N, H, device=dev).to(torch.bfloat16)
    bits = (torch.rand(N, H // 8, device=dev) * 255).to(torch.uint8)
    _, chunks = ops._dense_table(N, dev)
    t_plain = timed(lambda: ops.rows_wgrad(g, h, chunks, 1, out_dtype=torch.bfloat16, colsum_of=1))
    t_mask = timed(lambda: ops.rows_wgrad(g, h, chunks, 1, out_dtype=torch.bfloat16, colsum_of=1, mask_a_bits=bits))
    print("dense wgrad %d rows: plain %.1f us, with bit mask %.1f us (incl. reduce)" % (N, t_plain, t_mask), flush=True)


if __name__ == "__main__":
    main()
