#!/usr/bin/env python3
"""CPU timing of the REFERENCE's own layers / models in the authoring container (needs /root/reference; never runs on the GPU
box): subgraph_isomorphism/models/rgin.py:RGINLayer imported unmodified under the stand-ins of tests/golden/_ref_standins.py
(DGL update_all = call the reference's message UDF over all edges, index_add_ by destination, call its update UDF), and
graph_classification/.../models/gconv.py:GIN under the torch_geometric.nn stand-ins.  fwd + backward, best of 3 after a warm-up.
Output goes into BASELINE.md section 2."""
import importlib
import importlib.util
import os
import sys
import time
import types
from types import SimpleNamespace

import numpy as np
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _ref_standins as S  # noqa: E402
from dummynode4graphlearning_amd import synthetic  # noqa: E402
from oracle import transforms as OT  # noqa: E402

S.install()
REF = "/root/reference"
th.set_num_threads(os.cpu_count() or 1)


def best(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def si_rgin(cfg, H, regularizer, num_bases, act="relu"):
    SI = os.path.join(REF, "subgraph_isomorphism")
    sys.path.insert(0, SI)
    if "models" not in sys.modules:
        pkg = types.ModuleType("models")
        pkg.__path__ = [os.path.join(SI, "models")]
        sys.modules["models"] = pkg
    rgin = importlib.import_module("models.rgin")
    raw = cfg()
    aug = OT.dummy_augment_si(*(raw[k] for k in ("node_ptr", "edge_ptr", "src", "dst", "node_id", "node_label", "edge_id", "edge_label")),
                              raw["max_nv"], raw["max_nvl"], raw["max_ne"], raw["max_nel"])
    N, E, R = len(aug["node_label"]), len(aug["src"]), raw["num_rels"]
    th.manual_seed(0)
    layer = rgin.RGINLayer(H, H, num_rels=R, regularizer=regularizer, num_bases=num_bases, num_mlp_layers=2, act_func=act)
    g = S.FakeDGLGraph(aug["src"], aug["dst"], N)
    x = th.randn(N, H, requires_grad=True)
    et = th.from_numpy(aug["edge_label"])

    def step():
        layer.zero_grad()
        x.grad = None
        o, _ = layer(g, x, et)
        o.sum().backward()
    t = best(step)
    return N, E, t


def gc_gin(cfg, F_, H, labels, layers):
    mdir = os.path.join(REF, "graph_classification", "graph_neural_networks", "models")
    spec = importlib.util.spec_from_file_location("_ref_gconv", os.path.join(mdir, "gconv.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    raw = cfg()
    aug = OT.dummy_augment_gc(raw["node_ptr"], raw["edge_ptr"], raw["src"], raw["dst"], raw["node_label"], raw["edge_label"])
    N, E, G = len(aug["node_label"]), len(aug["src"]), len(aug["node_ptr"]) - 1
    x = th.nn.functional.one_hot(th.from_numpy(aug["node_label"]), labels + 1).float()
    if x.shape[1] < F_:
        x = th.cat([th.rand(N, F_ - x.shape[1]), x], 1)
    batch = th.repeat_interleave(th.arange(G), th.from_numpy(np.diff(aug["node_ptr"])))
    data = SimpleNamespace(x=x, edge_index=th.from_numpy(np.stack([aug["src"], aug["dst"]])), batch=batch,
                           y=th.randint(0, 2, (G,)))
    args = SimpleNamespace(num_features=x.shape[1], hidden_dim=H, num_classes=2, dropout_ratio=0.0, additional={"num_layers": layers, "train_eps": False},
                           epochs=1, device="cpu", dummy_weight=0)
    th.manual_seed(0)
    model = mod.GIN(args).train()

    def step():
        model.zero_grad()
        th.nn.functional.nll_loss(model(data), data.y).backward()
    return N, E, best(step)


if __name__ == "__main__":
    print("host: %d logical CPUs, torch %s, %d threads" % (os.cpu_count(), th.__version__, th.get_num_threads()))
    N, E, t = gc_gin(synthetic.config1, 8, 64, 7, 3)
    print("config 1  GIN 3-layer H=64 model step (32 MUTAG-shaped dummy graphs, N=%d, E=%d): %.2f ms -> %.2f M edges/s" % (N, E, t * 1e3, E / t / 1e6))
    N, E, t = gc_gin(synthetic.config2, 5, 128, 3, 2)
    print("config 2  GIN 2-layer H=128 model step (512 PROTEINS-shaped dummy graphs, N=%d, E=%d): %.1f ms -> %.2f M edges/s" % (N, E, t * 1e3, E / t / 1e6))
    for reg, nb, act in (("basis", -1, "relu"), ("bdd", 4, "relu"), ("bdd", 4, "leaky_relu")):
        N, E, t = si_rgin(synthetic.config3, 64, reg, nb, act)
        print("config 3  RGINLayer H=64 R=8 %s %s fwd+bwd (N=%d, E=%d): %.0f ms -> %.3f M edges/s" % (reg, act, N, E, t * 1e3, E / t / 1e6))
