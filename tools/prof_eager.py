import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from dummynode4graphlearning_amd import BatchedGraph, synthetic, transforms
from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer
dev = torch.device("cuda:0")
raw = synthetic.config3()
t = {k: torch.from_numpy(v).to(dev) for k, v in raw.items() if isinstance(v, np.ndarray)}
aug = transforms.dummy_augment_si(t["node_ptr"], t["edge_ptr"], t["src"], t["dst"], t["node_id"], t["node_label"], t["edge_id"], t["edge_label"], raw["max_nv"], raw["max_nvl"], raw["max_ne"], raw["max_nel"])
N = int(aug["node_label"].numel()); H = 64; R = raw["num_rels"]
DT = torch.float32 if "--f32" in sys.argv else torch.bfloat16
layer = RGINLayer(H, H, num_rels=R, regularizer="basis", num_bases=-1, num_mlp_layers=2, act_func="relu").to(dev).to(DT)
bnn = (aug["node_ptr"][1:] - aug["node_ptr"][:-1]).long(); bne = (aug["edge_ptr"][1:] - aug["edge_ptr"][:-1]).long()
g = BatchedGraph(aug["src"], aug["dst"], N, bnn, bne, node_ptr=aug["node_ptr"], edge_ptr=aug["edge_ptr"])   # (graph boundaries: the whole-graph launches)
et = aug["edge_label"].long()
x = torch.randn(N, H, device=dev).to(DT).requires_grad_(True)
gout = torch.randn(N, H, device=dev).to(DT)
def step():
    out, _ = layer(g, x, et)
    out.backward(gout)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
