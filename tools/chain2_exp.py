#!/usr/bin/env python3
"""Timing of the two MLP chain launches of config 5 alone (dn_rows_chain2_bf16: forward with sign-bit outputs, backward with both
masks): python tools/chain2_exp.py [--rows 1015808]   (DN_HIP_LIB selects the build)"""
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
    ap.add_argument("--slope", type=float, default=0.0)
    a = ap.parse_args()
    from dummynode4graphlearning_amd import ops
    dev = torch.device("cuda:0")
    N, H = a.rows, 256
    torch.manual_seed(0)
    x = torch.randn(N, H, device=dev).to(torch.bfloat16)
    g = torch.randn(N, H, device=dev).to(torch.bfloat16)
    W1 = (torch.randn(H, H, device=dev) / 16).to(torch.bfloat16)
    W2 = (torch.randn(H, H, device=dev) / 16).to(torch.bfloat16)
    b1 = torch.randn(H, device=dev).to(torch.bfloat16)
    b2 = torch.randn(H, device=dev).to(torch.bfloat16)
    h1, h2, bits1, bits2 = ops.rows_chain2(x, W1, b1, True, W2, b2, True, want_bits=True, slope=a.slope)
    t_f = timed(lambda: ops.rows_chain2(x, W1, b1, True, W2, b2, True, want_bits=True, slope=a.slope))
    t_n = timed(lambda: ops.rows_chain2(x, W1, b1, True, W2, b2, True, slope=a.slope))
    t_l = timed(lambda: ops.rows_chain2(x, W1, None, False, W2, None, False, slope=a.slope))
    print("  forward without sign-bit outputs %.1f us; without bias / activation too %.1f us" % (t_n, t_l))
    t_b = timed(lambda: ops.rows_chain2(g, W2, None, False, W1, None, False, mask0_bits=bits2, mask1_bits=bits1, w_kn=(True, True), slope=a.slope))
    print("chain2 %d rows slope %.3f: forward %.1f us, backward %.1f us" % (N, a.slope, t_f, t_b), flush=True)


if __name__ == "__main__":
    main()
