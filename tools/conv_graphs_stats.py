#!/usr/bin/env python3
"""Where does dn_conv_graphs_bf16 spend its time?  Build with DN_BUILD_EXTRA=-DDN_CG_STATS; workgroup 0 leaves 100 MHz clock stamps
behind: start, weights in registers, image in LDS, edges bucketed, products done, rows stored.  usage (GPU box):
DN_BUILD_EXTRA=-DDN_CG_STATS python -m dummynode4graphlearning_amd.csrc.build --force && python tools/conv_graphs_stats.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dummynode4graphlearning_amd import ops, synthetic, transforms
dev = "cuda:0"
raw = synthetic.config3(seed=3, graphs=512)
keys = ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")
aug = transforms.dummy_augment_si(*(torch.from_numpy(raw[k]).to(dev) for k in keys), raw["max_nv"], raw["max_nvl"], raw["max_ne"], raw["max_nel"])
N, R, H = int(aug["node_label"].numel()), 8, 64
ix = ops.RowIndex(aug["src"], aug["dst"], aug["edge_label"], N, R, self_loop=True, node_ptr=aug["node_ptr"], edge_ptr=aug["edge_ptr"])
ix._cg_err = torch.zeros(16, dtype=torch.int32, device=dev)
x = torch.randn(N, H, device=dev).to(torch.bfloat16)
W = (torch.randn(R, H, H, device=dev) / 8).to(torch.bfloat16); Wl = (torch.randn(H, H, device=dev) / 8).to(torch.bfloat16)
out = torch.empty_like(x)
for kn in (True, False):
    pw = ops.PassWeights(W, Wl, kn=kn)
    for it in range(5):
        ops.conv_graphs(x, pw, None, ix, "f" if kn else "b", out)
        torch.cuda.synchronize()
    st = ix._cg_err[2:14].cpu().numpy().view(np.int64)
    print("kn", kn, "us per phase (weights, image, bucketing, products, stores):", [round((st[k + 1] - st[k]) / 100.0, 2) for k in range(5)])
