#!/usr/bin/env python3
"""Which Python lines launch the small torch kernels (fills, adds, copies) of a GC model step: torch.profiler with stacks over one
eager step of the config-4 GIN model.   python tools/small_ops_trace.py"""
import os, sys, collections
from types import SimpleNamespace
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dummynode4graphlearning_amd import GraphBatch, synthetic, transforms  # noqa: E402
from dummynode4graphlearning_amd import graph_classification as GC  # noqa: E402

dev = torch.device("cuda:0")
raw = synthetic.config4(seed=4, graphs=512) if hasattr(synthetic, "config4") else None
aug = transforms.dummy_augment_gc(*(torch.from_numpy(np.ascontiguousarray(raw[k])).to(dev) for k in
                                    ("node_ptr", "edge_ptr", "src", "dst", "node_label", "edge_label")))
x = F.one_hot(aug["node_label"].long(), 38).float()
node_ptr = aug["node_ptr"].long()
G = node_ptr.numel() - 1
batch = torch.repeat_interleave(torch.arange(G, device=dev), node_ptr[1:] - node_ptr[:-1])
y = torch.randint(0, 2, (G,), device=dev)
data = GraphBatch(x, torch.stack([aug["src"].long(), aug["dst"].long()]), batch, y=y, ptr=node_ptr)
margs = SimpleNamespace(num_features=38, hidden_dim=256, num_classes=2, dropout_ratio=0.0, num_relations=2, additional=None, epochs=1,
                        device=dev, dummy_weight=0)
model = GC.GIN(margs).to(dev).train()


def step():
    for p in model.parameters():
        p.grad = None
    F.nll_loss(model(data), data.y).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

WATCH = ("fill_", "zero_", "zeros", "add_", "add.", "copy_", "mul", "cat", "cumsum", "sub", "div", "_to_copy", "arange", "repeat_interleave",
         "index_select", "gather", "scatter", "index_add", "ones", "full", "clone", "new_zeros", "empty_like")
small = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(w in name for w in WATCH):
            st = traceback.extract_stack()
            ours = [f for f in st if "dummynode4graphlearning_amd" in f.filename]
            where = "%s:%d %s" % (os.path.relpath(ours[-1].filename, ROOT), ours[-1].lineno, ours[-1].line) if ours else "(autograd engine / caller)"
            small[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    step()
torch.cuda.synchronize()
for (name, where), c in sorted(small.items(), key=lambda kv: -kv[1]):
    print("%3d x %-28s %s" % (c, name, where[:150]))
