import sys, json, numpy as np, torch
sys.path.insert(0, "/root/repo")
from dummynode4graphlearning_amd import ops, BatchedGraph
from dummynode4graphlearning_amd.subgraph_isomorphism import RGINLayer, RGCNLayer
DEV="cuda:0"
z=np.load("/root/repo/tests/golden/si_layers.npz"); meta=json.loads(bytes(z["meta"]).decode())
def rm(a,b):
    a,b=a.detach().double().cpu(),b.detach().double().cpu(); return float((a-b).abs().max()/b.abs().max().clamp(min=1e-12))
for m in meta:
    if m["hidden_dim"] < 64: continue
    tag=m["tag"]
    for ex in (False, True):
        ops.F32_EXACT=ex
        kw=dict(num_rels=m["num_rels"],regularizer=m["regularizer"],num_bases=m["num_bases"],self_loop=m["self_loop"],act_func=m["act_func"])
        layer=RGINLayer(m["input_dim"],m["hidden_dim"],num_mlp_layers=m["num_mlp_layers"],**kw) if m["kind"]=="rgin" else RGCNLayer(m["input_dim"],m["hidden_dim"],edge_norm=m["edge_norm"],**kw)
        layer.load_state_dict({k[len(tag)+7:]:torch.from_numpy(z[k]) for k in z.files if k.startswith(tag+"/param/")})
        layer=layer.to(DEV).train()
        u,v,t=(torch.from_numpy(z[tag+"/"+k]).to(DEV) for k in ("u","v","t"))
        x=torch.from_numpy(z[tag+"/x"]).to(DEV).requires_grad_(True)
        out,_=layer(BatchedGraph(u,v,m["N"]),x,t)
        (out*torch.from_numpy(z[tag+"/coef"]).to(DEV)).sum().backward()
        errs={"out":rm(out,torch.from_numpy(z[tag+"/out"])),"gx":rm(x.grad,torch.from_numpy(z[tag+"/grad_x"]))}
        for k,p in layer.named_parameters():
            ref=z[tag+"/grad/"+k]
            if ref.size and np.abs(ref).max()>0: errs[k]=rm(p.grad,torch.from_numpy(ref))
        print(tag,m["kind"],m["hidden_dim"],m["regularizer"],"exact" if ex else "split",{k:"%.1e"%e for k,e in errs.items()})
